// post.h — the step after the path: the display transform and the image metric the
// reference runs on the renderer's output, as HIP kernels.
//   tonemap  : src/Shaders/kernels/tonemap.hlsl (reduce_max :106-153, main :155-226), modes of tonemap.h:8-21
//   compare  : src/Shaders/kernels/image_compare.hlsl:13-46 (what ImageComparer shows, ImageComparer.cpp:61-90)
// Arithmetic follows the arithmetic contract (include/sthip_detmath.h): pow() is det_powf, 2^exposure is
// det_expf(exposure * ln 2). Not restated: the exposure smoothing over frames (gExposureAlpha with gPrevMax,
// tonemap.hlsl:170-180) — a render call here has no previous frame.
#pragma once

#include "device_math.h"

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
__global__ void k_tonemap(const float4* input, const float4* albedo, float4* output, uint32_t n, uint32_t mode, uint32_t modulate, uint32_t gamma, float exposure, const uint32_t* gmax) {
  const f3 cur_max = F3((float)gmax[0] / TONEMAP_MAX_QUANTIZATION, (float)gmax[1] / TONEMAP_MAX_QUANTIZATION, (float)gmax[2] / TONEMAP_MAX_QUANTIZATION);
  const float cur_max_l = (float)gmax[3] / TONEMAP_MAX_QUANTIZATION;
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
