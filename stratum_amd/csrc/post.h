// post.h — the step after the path: the display transform and the image metric the
// reference runs on the renderer's output, as HIP kernels.
//   tonemap  : src/Shaders/kernels/tonemap.hlsl (reduce_max :106-153, main :155-226), modes of tonemap.h:8-21
//   compare  : src/Shaders/kernels/image_compare.hlsl:13-46 (what ImageComparer shows, ImageComparer.cpp:61-90)
// Arithmetic follows the arithmetic contract (include/sthip_detmath.h): pow() is det_powf, 2^exposure is
// det_expf(exposure * ln 2). Not restated: the exposure smoothing over frames (gExposureAlpha with gPrevMax,
// tonemap.hlsl:170-180) — a render call here has no previous frame.
#pragma once

#include "device_math.h"
#include "shading.h"

#define TONEMAP_MAX_QUANTIZATION 16384.0f  // gMaxQuantization, tonemap.hlsl:104

enum TonemapMode {  // tonemap.h:8-21
  eRaw,
  eReinhard,
  eReinhardExtended,
  eReinhardLuminance,
  eReinhardLuminanceExtended,
  eUncharted2,
  eFilmic,
  eACES,
  eACESApprox,
  eViridisR,
  eViridisLengthRGB,
  eTonemapModeCount
};

DEV f3 operator/(f3 a, f3 b) { return F3(a.x / b.x, a.y / b.y, a.z / b.z); }

DEV f3 viridis_quintic(float x) {  // common.h:114-123
  const float x1x = 1, x1y = x, x1z = x * x, x1w = x * x * x;
  const float x2x = x1y * x1w, x2y = x1z * x1w;
  return F3((x1x * 0.280268003f + x1y * -0.143510503f + x1z * 2.225793877f + x1w * -14.815088879f) + (x2x * 25.212752309f + x2y * -11.772589584f),
            (x1x * -0.002117546f + x1y * 1.617109353f + x1z * -1.909305070f + x1w * 2.701152864f) + (x2x * -1.685288385f + x2y * 0.178738871f),
            (x1x * 0.300805501f + x1y * 2.614650302f + x1z * -12.019139090f + x1w * 28.933559110f) + (x2x * -33.491294770f + x2y * 13.762053843f));
}
DEV float saturate1(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
DEV f3 saturate3(f3 v) { return F3(saturate1(v.x), saturate1(v.y), saturate1(v.z)); }
DEV float rgb_to_srgb1(float c) { return c <= 0.0031308f ? c * 12.92f : det_powf(c * 1.055f, 1 / 2.4f) - 0.055f; }  // common.h:103-109

DEV f3 tonemap_reinhard(f3 c) {
  const float l = luminance3(c);
  const f3 tc = c / (F3s(1.0f) + c);
  const f3 a = c / (1 + l);
  return F3(lerp1(a.x, tc.x, tc.x), lerp1(a.y, tc.y, tc.y), lerp1(a.z, tc.z, tc.z));
}
DEV f3 tonemap_reinhard_extended(f3 c, f3 max_c) {
  const f3 m = F3(max_c.x == 0 ? 1.0f : max_c.x, max_c.y == 0 ? 1.0f : max_c.y, max_c.z == 0 ? 1.0f : max_c.z);  // lerp(max_c, 1, max_c == 0)
  return c / (F3s(1.0f) + c) * (F3s(1.0f) + c / (m * m));
}
DEV f3 tonemap_reinhard_luminance(f3 c) {
  const float l = luminance3(c);
  const float l1 = l / (1 + l);
  return c * (l1 / l);
}
DEV f3 tonemap_reinhard_luminance_extended(f3 c, float max_l) {
  const float l = luminance3(c);
  const float l1 = (l / (1 + l)) * (1 + l / pow2f(max_l == 0 ? 1 : max_l));
  return c * (l1 / l);
}
DEV float uncharted2_partial1(float x) {
  const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
  return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}
DEV f3 tonemap_uncharted2(f3 c, float max_l) {
  const float d = uncharted2_partial1(max_l == 0 ? 1 : max_l);
  return F3(uncharted2_partial1(c.x) / d, uncharted2_partial1(c.y) / d, uncharted2_partial1(c.z) / d);
}
DEV float filmic1(float c) {
  c = fmaxf(0.0f, c - 0.004f);
  return (c * (6.2f * c + 0.5f)) / (c * (6.2f * c + 1.7f) + 0.06f);
}
DEV float rtt_and_odt_fit1(float v) {
  const float a = v * (v + 0.0245786f) - 0.000090537f;
  const float b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
  return a / b;
}
DEV f3 aces_fitted(f3 v) {
  const f3 i = F3(0.59719f * v.x + 0.35458f * v.y + 0.04823f * v.z, 0.07600f * v.x + 0.90834f * v.y + 0.01566f * v.z, 0.02840f * v.x + 0.13383f * v.y + 0.83777f * v.z);
  const f3 f = F3(rtt_and_odt_fit1(i.x), rtt_and_odt_fit1(i.y), rtt_and_odt_fit1(i.z));
  return saturate3(F3(1.60475f * f.x + -0.53108f * f.y + -0.07367f * f.z, -0.10208f * f.x + 1.10813f * f.y + -0.00605f * f.z, -0.00327f * f.x + -0.07276f * f.y + 1.07602f * f.z));
}
DEV float aces_approx1(float v) {
  v *= 0.6f;
  return saturate1((v * (2.51f * v + 0.03f)) / (v * (2.43f * v + 0.59f) + 0.14f));
}

// reduce_max, tonemap.hlsl:106-153: per-channel and luminance maxima, quantised so that InterlockedMax works on uints
__global__ void k_tonemap_reduce_max(const float4* input, const float4* albedo, uint32_t n, uint32_t modulate, uint32_t* gmax) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float4 v = input[i];
    if (modulate) {
      const float4 a = albedo[i];
      v.x *= a.x;
      v.y *= a.y;
      v.z *= a.z;
    }
    v.w = luminance3(xyz(v));
    if (v.x != v.x || v.y != v.y || v.z != v.z || v.w != v.w || v.w <= 0) continue;
    const float q[4] = {v.x * TONEMAP_MAX_QUANTIZATION, v.y * TONEMAP_MAX_QUANTIZATION, v.z * TONEMAP_MAX_QUANTIZATION, v.w * TONEMAP_MAX_QUANTIZATION};
    for (int k = 0; k < 4; k++) {
      const float c = fminf(fmaxf(q[k], 0.0f), 4294967295.0f);
      atomicMax(&gmax[k], c >= 4294967295.0f ? 0xFFFFFFFFu : (uint32_t)c);
    }
  }
}

// main, tonemap.hlsl:155-226
// prev: the previous call's state (max r, g, b, luminance; luminance moments), state_out: this call's (gMax bytes 16..39)
struct TonemapState {
  float v[6];
};
__global__ void k_tonemap(const float4* input, const float4* albedo, float4* output, uint32_t n, uint32_t mode, uint32_t modulate, uint32_t gamma, float exposure, const uint32_t* gmax,
                          float exposure_alpha, TonemapState prev, float* state_out) {
  f3 cur_max = F3((float)gmax[0] / TONEMAP_MAX_QUANTIZATION, (float)gmax[1] / TONEMAP_MAX_QUANTIZATION, (float)gmax[2] / TONEMAP_MAX_QUANTIZATION);
  float cur_max_l = (float)gmax[3] / TONEMAP_MAX_QUANTIZATION;
  float m0 = cur_max_l, m1 = cur_max_l * cur_max_l;  // cur_moments, tonemap.hlsl:168
  if (exposure_alpha > 0 && exposure_alpha < 1) {     // :169-178
    if (prev.v[4] == prev.v[4] && prev.v[5] == prev.v[5] && prev.v[4] > 0) {
      const float sa = sqrtf(exposure_alpha);
      m0 = lerp1(prev.v[4], m0, sa);
      m1 = lerp1(prev.v[5], m1, sa);
    }
    if (prev.v[0] == prev.v[0] && prev.v[1] == prev.v[1] && prev.v[2] == prev.v[2] && prev.v[3] == prev.v[3] && prev.v[3] > 0) {
      cur_max = F3(lerp1(prev.v[0], cur_max.x, exposure_alpha), lerp1(prev.v[1], cur_max.y, exposure_alpha), lerp1(prev.v[2], cur_max.z, exposure_alpha));
      cur_max_l = lerp1(prev.v[3], cur_max_l, exposure_alpha);
    }
  }
  if (state_out && blockIdx.x == 0 && threadIdx.x == 0) {
    state_out[0] = cur_max.x;
    state_out[1] = cur_max.y;
    state_out[2] = cur_max.z;
    state_out[3] = cur_max_l;
    state_out[4] = m0;
    state_out[5] = m1;
  }
  const float scale = det_expf(exposure * 0.693147180559945f);  // pow(2, gExposure)
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    f3 radiance = xyz(input[i]);
    if (modulate) radiance = radiance * (F3s(1e-2f) + xyz(albedo[i]));
    radiance = radiance * scale;
    switch (mode) {
      case eReinhard: radiance = tonemap_reinhard(radiance); break;
      case eReinhardExtended: radiance = tonemap_reinhard_extended(radiance, cur_max); break;
      case eReinhardLuminance: radiance = tonemap_reinhard_luminance(radiance); break;
      case eReinhardLuminanceExtended: radiance = tonemap_reinhard_luminance_extended(radiance, cur_max_l); break;
      case eUncharted2: radiance = tonemap_uncharted2(radiance, cur_max_l); break;
      case eFilmic: radiance = F3(filmic1(radiance.x), filmic1(radiance.y), filmic1(radiance.z)); break;
      case eACES: radiance = aces_fitted(radiance); break;
      case eACESApprox: radiance = F3(aces_approx1(radiance.x), aces_approx1(radiance.y), aces_approx1(radiance.z)); break;
      case eViridisR: radiance = viridis_quintic(saturate1(luminance3(radiance))); break;
      case eViridisLengthRGB: radiance = viridis_quintic(saturate1(luminance3(radiance) / (cur_max_l == 0 ? 1.0f : cur_max_l))); break;
      default: break;
    }
    if (gamma) radiance = F3(rgb_to_srgb1(radiance.x), rgb_to_srgb1(radiance.y), rgb_to_srgb1(radiance.z));
    output[i] = make_float4(radiance.x, radiance.y, radiance.z, 1.0f);
  }
}

// image_compare.hlsl:13-46. The reference sums a wave with WaveActiveSum (order and wave shape are the
// hardware's); pinned here as: 64 consecutive pixels (row-major) per group, summed in pixel order, truncated to
// uint after scaling by the quantisation, groups added with an integer atomic (order independent).
__global__ void __launch_bounds__(64) k_image_compare(const float4* image1, const float4* image2, uint32_t n, uint32_t metric, uint32_t quantization, uint32_t* out) {
  __shared__ float err[64];
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  float error = 0;
  if (i < n) {
    const f3 c1 = xyz(image1[i]), c2 = xyz(image2[i]);
    const f3 d = c1 - c2;
    if (metric == 0) {  // eSMAPE
      const f3 q = F3(fabsf(d.x) / (fabsf(c1.x) + fabsf(c2.x)), fabsf(d.y) / (fabsf(c1.y) + fabsf(c2.y)), fabsf(d.z) / (fabsf(c1.z) + fabsf(c2.z)));
      error = q.x + q.y + q.z;
    } else if (metric == 1) {  // eMSE
      error = d.x * d.x + d.y * d.y + d.z * d.z;
    } else {  // eAverage
      error = d.x + d.y + d.z;
    }
    error /= (float)(3u * n);
  }
  err[threadIdx.x] = error;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0;
    for (int k = 0; k < 64; k++) s += err[k];
    const float valf = s * (float)quantization;
    const uint32_t val = valf >= 4294967295.0f ? 0xFFFFFFFFu : (valf > 0 ? (uint32_t)valf : 0u);
    const uint32_t prev = atomicAdd(&out[0], val);
    if (valf >= 4294967295.0f || 0xFFFFFFFFu - val < prev) out[1] = 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// temporal accumulation, kernels/temporal_accumulation.hlsl:59-145 (host: Denoiser.cpp:73-77,176-213): reprojects
// the previous frame's accumulated colour / moments through gPrevUVs with per-tap validity tests, then blends the
// new sample in with alpha = n_new / n. Both specialisations (gReprojection on / off) and gDemodulateAlbedo.
// As upstream: the normal test compares against cos(degrees(2)) (degrees, not radians: 114.59 rad).
// ---------------------------------------------------------------------------------------------------------------
#define ACCUMULATE_INLINE_VIEWS 4u
struct AccumulateParams {
  uint32_t width, height, view_count, reprojection, demodulate_albedo;
  float history_limit;
  uint32_t instance_count;
  const sthip_ViewData* views;  // device array, or null: the views travel in `inline_views` (up to ACCUMULATE_INLINE_VIEWS: no staging, no wait)
  sthip_ViewData inline_views[4];
  const float4* radiance;
  const float4* albedo;
  const sthip_VisibilityInfo* visibility;
  const sthip_DepthInfo* depth;
  const float2* prev_uvs;
  const sthip_VisibilityInfo* prev_visibility;
  const sthip_DepthInfo* prev_depth;
  const float4* prev_accum_color;
  const float2* prev_accum_moments;
  const uint32_t* instance_index_map;  // may be null: identity
  float4* accum_color;
  float2* accum_moments;
};
DEV bool bad4(float4 c) { return c.x != c.x || c.y != c.y || c.z != c.z || c.w != c.w || isinf(c.x) || isinf(c.y) || isinf(c.z) || isinf(c.w); }
__global__ void __launch_bounds__(256) k_accumulate(AccumulateParams p) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.width * p.height) return;
  const uint32_t x = i % p.width, y = i / p.width;
  int view_index = -1;
  const sthip_ViewData* views = p.views ? p.views : p.inline_views;
  for (uint32_t v = 0; v < p.view_count; v++) {
    const sthip_ViewData& vw = views[v];
    if ((int)x >= vw.image_min[0] && (int)y >= vw.image_min[1] && (int)x < vw.image_max[0] && (int)y < vw.image_max[1]) {
      view_index = (int)v;
      break;
    }
  }
  if (view_index < 0) return;
  const sthip_ViewData& view = views[view_index];
  float4 color_prev = make_float4(0, 0, 0, 0);
  float2 moments_prev = make_float2(0, 0);
  float sum_w = 0;
  if (p.reprojection) {
    const sthip_VisibilityInfo vis = p.visibility[i];
    const sthip_DepthInfo depth = p.depth[i];
    if ((vis.instance_primitive_index & 0xFFFFu) != STHIP_INVALID_INSTANCE) {
      const float2 uv = p.prev_uvs[i];
      const float pos_x = (float)view.image_min[0] + uv.x * (float)(view.image_max[0] - view.image_min[0]) - 0.5f;
      const float pos_y = (float)view.image_min[1] + uv.y * (float)(view.image_max[1] - view.image_min[1]) - 0.5f;
      // (a position that no int can hold selects no tap; the conversion of such a float is implementation-defined)
      const bool finite_pos = fabsf(pos_x) < 1e9f && fabsf(pos_y) < 1e9f;
      const int px = finite_pos ? (int)pos_x : -0x40000000, py = finite_pos ? (int)pos_y : -0x40000000;  // int2 p = pos_prev: truncation
      const float wx = pos_x - floorf(pos_x), wy = pos_y - floorf(pos_y);    // frac()
      float sd, cd;
      det_sincosf(2.0f * 57.2957795130823f, &sd, &cd);  // cos(degrees(2))
      const f3 n = unpack_normal_octahedron(vis.packed_normal);
      const uint32_t inst = vis.instance_primitive_index & 0xFFFFu;
      const uint32_t mapped = p.instance_index_map ? (inst < p.instance_count ? p.instance_index_map[inst] : 0xFFFFFFFFu) : inst;
      const float dz = sqrtf(depth.dz_dxy[0] * depth.dz_dxy[0] + depth.dz_dxy[1] * depth.dz_dxy[1]);
      for (int yy = 0; yy <= 1; yy++)
        for (int xx = 0; xx <= 1; xx++) {
          const int qx = px + xx, qy = py + yy;
          if (!(qx >= view.image_min[0] && qy >= view.image_min[1] && qx < view.image_max[0] && qy < view.image_max[1])) continue;
          const size_t q = (size_t)qy * p.width + qx;
          const sthip_VisibilityInfo pv = p.prev_visibility[q];
          if (mapped != (pv.instance_primitive_index & 0xFFFFu)) continue;
          if (dot3(n, unpack_normal_octahedron(pv.packed_normal)) < cd) continue;
          if (fabsf(depth.prev_z - p.prev_depth[q].z) >= 1.5f * dz) continue;
          const float4 c = p.prev_accum_color[q];
          if (c.w <= 0 || bad4(c)) continue;
          const float wc = (xx == 0 ? (1 - wx) : wx) * (yy == 0 ? (1 - wy) : wy);
          color_prev.x += c.x * wc;
          color_prev.y += c.y * wc;
          color_prev.z += c.z * wc;
          color_prev.w += c.w * wc;
          const float2 m = p.prev_accum_moments[q];
          moments_prev.x += m.x * wc;
          moments_prev.y += m.y * wc;
          sum_w += wc;
        }
    }
  } else {
    color_prev = p.prev_accum_color[i];
    if (color_prev.x != color_prev.x || color_prev.y != color_prev.y || color_prev.z != color_prev.z || isinf(color_prev.x) || isinf(color_prev.y) || isinf(color_prev.z))
      color_prev = make_float4(0, 0, 0, 0);
    else
      moments_prev = p.prev_accum_moments[i];
    sum_w = 1;
  }
  float4 color_curr = p.radiance[i];
  if (p.demodulate_albedo) {
    const float4 a = p.albedo[i];
    color_curr.x /= (1e-2f + a.x);
    color_curr.y /= (1e-2f + a.y);
    color_curr.z /= (1e-2f + a.z);
  }
  if (isinf(color_curr.x) || isinf(color_curr.y) || isinf(color_curr.z) || color_curr.x != color_curr.x || color_curr.y != color_curr.y || color_curr.z != color_curr.z)
    color_curr = make_float4(0, 0, 0, 0);
  if (isinf(moments_prev.x) || isinf(moments_prev.y) || moments_prev.x != moments_prev.x || moments_prev.y != moments_prev.y) moments_prev = make_float2(0, 0);
  const float l = luminance3(xyz(color_curr));
  if (sum_w > 0 && color_prev.w > 0) {
    const float inv_sum = 1 / sum_w;
    color_prev.x *= inv_sum;
    color_prev.y *= inv_sum;
    color_prev.z *= inv_sum;
    color_prev.w *= inv_sum;
    moments_prev.x *= inv_sum;
    moments_prev.y *= inv_sum;
    float n = color_prev.w + color_curr.w;
    if (p.history_limit > 0 && n > p.history_limit) n = p.history_limit;
    const float alpha = saturate1(color_curr.w / n);
    p.accum_color[i] = make_float4(lerp1(color_prev.x, color_curr.x, alpha), lerp1(color_prev.y, color_curr.y, alpha), lerp1(color_prev.z, color_curr.z, alpha), n);
    p.accum_moments[i] = make_float2(lerp1(moments_prev.x, l, alpha), lerp1(moments_prev.y, l * l, alpha));
  } else {
    p.accum_color[i] = color_curr;
    p.accum_moments[i] = make_float2(l, l * l);
  }
}

