// post_oracle.cpp — CPU restatement of the steps after the path. TEST INFRASTRUCTURE: only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; nothing the product ships links it.
//   orc_tonemap        kernels/tonemap.hlsl:21-226 (curves :21-102, reduce_max :106-153, main :155-226),
//                      colour helpers of common.h:66-69 (luminance), :107-113 (rgb_to_srgb), :115-123 (viridis)
//   orc_image_compare  kernels/image_compare.hlsl:13-46
//   orc_accumulate     kernels/temporal_accumulation.hlsl:59-145 (both gReprojection specialisations)
// Parity: pow/exp follow the arithmetic contract (include/sthip_detmath.h); the wave sum of image_compare is pinned
// to 64 consecutive pixels added in order (see include/sthip.h). The reference's own tests hold no vectors for these
// kernels, so this part of the oracle is "parity unpinned" against the reference and pinned only by its own
// known-answer tests (tests/test_post.py: closed-form values of every curve, hand-computed metric values).
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../include/sthip.h"
#include "../include/sthip_detmath.h"

namespace {

float lum(const float* c) { return c[0] * 0.2126f + c[1] * 0.7152f + c[2] * 0.0722f; }
float sat(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
float u2p(float x) {  // tonemap_uncharted2_partial1, tonemap.hlsl:44-52
  const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
  return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}
void viridis(float x, float* out) {  // common.h:115-123
  const float p[6] = {1.0f, x, x * x, x * x * x, 0, 0};
  const float x4 = p[1] * p[3], x5 = p[2] * p[3];
  static const float K[3][6] = {{0.280268003f, -0.143510503f, 2.225793877f, -14.815088879f, 25.212752309f, -11.772589584f},
                                {-0.002117546f, 1.617109353f, -1.909305070f, 2.701152864f, -1.685288385f, 0.178738871f},
                                {0.300805501f, 2.614650302f, -12.019139090f, 28.933559110f, -33.491294770f, 13.762053843f}};
  for (int c = 0; c < 3; c++) {
    float d4 = p[0] * K[c][0];
    d4 = d4 + p[1] * K[c][1];
    d4 = d4 + p[2] * K[c][2];
    d4 = d4 + p[3] * K[c][3];
    const float d2 = x4 * K[c][4] + x5 * K[c][5];
    out[c] = d4 + d2;
  }
}


// bitfield.h:76-93 unpack_normal_octahedron (as in stratum_oracle.cpp)
void unpack_oct(uint32_t packed, float out[3]) {
  const float px = det_f16tof32(packed & 0xFFFFu), py = det_f16tof32(packed >> 16);
  float v[3] = {px, py, 1.0f - (fabsf(px) + fabsf(py))};
  if (v[2] < 0) {
    const float qx = (1.0f - fabsf(v[1])) * (v[0] >= 0 ? 1.0f : -1.0f);
    const float qy = (1.0f - fabsf(v[0])) * (v[1] >= 0 ? 1.0f : -1.0f);
    v[0] = qx;
    v[1] = qy;
  }
  const float len = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  out[0] = v[0] / len;
  out[1] = v[1] / len;
  out[2] = v[2] / len;
}
bool notfinite(float x) { return x != x || isinf(x); }
float mixf(float a, float b, float t) { return a + t * (b - a); }

}  // namespace

extern "C" {

// temporal_accumulation.hlsl:59-145; all pointers of the descriptor are host pointers here
void orc_accumulate(const sthip_accumulate_desc* d) {
  const uint32_t W = d->width, H = d->height;
  float sd, cd;
  det_sincosf(2.0f * 57.2957795130823f, &sd, &cd);  // cos(degrees(2)), :88
  for (uint32_t y = 0; y < H; y++)
    for (uint32_t x = 0; x < W; x++) {
      const size_t i = (size_t)y * W + x;
      const sthip_ViewData* view = nullptr;
      for (uint32_t v = 0; v < d->view_count && !view; v++) {
        const sthip_ViewData& c = d->gViews[v];
        if ((int)x >= c.image_min[0] && (int)y >= c.image_min[1] && (int)x < c.image_max[0] && (int)y < c.image_max[1]) view = &c;
      }
      if (!view) continue;
      float cp[4] = {0, 0, 0, 0}, mp[2] = {0, 0}, sum_w = 0;
      if (d->reprojection) {
        const sthip_VisibilityInfo vis = d->gVisibility[i];
        const sthip_DepthInfo depth = d->gDepth[i];
        const uint32_t inst = vis.instance_primitive_index & 0xFFFFu;
        if (inst != STHIP_INVALID_INSTANCE) {
          const float pos_x = (float)view->image_min[0] + d->gPrevUVs[2 * i] * (float)(view->image_max[0] - view->image_min[0]) - 0.5f;
          const float pos_y = (float)view->image_min[1] + d->gPrevUVs[2 * i + 1] * (float)(view->image_max[1] - view->image_min[1]) - 0.5f;
          const bool finite_pos = fabsf(pos_x) < 1e9f && fabsf(pos_y) < 1e9f;
          const int px = finite_pos ? (int)pos_x : -0x40000000, py = finite_pos ? (int)pos_y : -0x40000000;
          const float wx = pos_x - floorf(pos_x), wy = pos_y - floorf(pos_y);
          float n[3];
          unpack_oct(vis.packed_normal, n);
          const uint32_t mapped = d->gInstanceIndexMap ? (inst < d->instance_count ? d->gInstanceIndexMap[inst] : 0xFFFFFFFFu) : inst;
          const float dz = sqrtf(depth.dz_dxy[0] * depth.dz_dxy[0] + depth.dz_dxy[1] * depth.dz_dxy[1]);
          for (int yy = 0; yy <= 1; yy++)
            for (int xx = 0; xx <= 1; xx++) {
              const int qx = px + xx, qy = py + yy;
              if (!(qx >= view->image_min[0] && qy >= view->image_min[1] && qx < view->image_max[0] && qy < view->image_max[1])) continue;
              const size_t q = (size_t)qy * W + qx;
              const sthip_VisibilityInfo pv = d->gPrevVisibility[q];
              if (mapped != (pv.instance_primitive_index & 0xFFFFu)) continue;
              float pn[3];
              unpack_oct(pv.packed_normal, pn);
              if (n[0] * pn[0] + n[1] * pn[1] + n[2] * pn[2] < cd) continue;
              if (fabsf(depth.prev_z - d->gPrevDepth[q].z) >= 1.5f * dz) continue;
              const float* c = d->gPrevAccumColor + 4 * q;
              if (c[3] <= 0 || notfinite(c[0]) || notfinite(c[1]) || notfinite(c[2]) || notfinite(c[3])) continue;
              const float wc = (xx == 0 ? (1 - wx) : wx) * (yy == 0 ? (1 - wy) : wy);
              for (int k = 0; k < 4; k++) cp[k] += c[k] * wc;
              mp[0] += d->gPrevAccumMoments[2 * q] * wc;
              mp[1] += d->gPrevAccumMoments[2 * q + 1] * wc;
              sum_w += wc;
            }
        }
      } else {
        memcpy(cp, d->gPrevAccumColor + 4 * i, 16);
        if (notfinite(cp[0]) || notfinite(cp[1]) || notfinite(cp[2])) {
          cp[0] = cp[1] = cp[2] = cp[3] = 0;
        } else {
          mp[0] = d->gPrevAccumMoments[2 * i];
          mp[1] = d->gPrevAccumMoments[2 * i + 1];
        }
        sum_w = 1;
      }
      float cc[4];
      memcpy(cc, d->gRadiance + 4 * i, 16);
      if (d->demodulate_albedo)
        for (int k = 0; k < 3; k++) cc[k] /= (1e-2f + d->gAlbedo[4 * i + k]);
      if (notfinite(cc[0]) || notfinite(cc[1]) || notfinite(cc[2])) cc[0] = cc[1] = cc[2] = cc[3] = 0;
      if (notfinite(mp[0]) || notfinite(mp[1])) mp[0] = mp[1] = 0;
      const float l = lum(cc);
      float* oc = d->gAccumColor + 4 * i;
      float* om = d->gAccumMoments + 2 * i;
      if (sum_w > 0 && cp[3] > 0) {
        const float inv_sum = 1 / sum_w;
        for (int k = 0; k < 4; k++) cp[k] *= inv_sum;
        mp[0] *= inv_sum;
        mp[1] *= inv_sum;
        float n = cp[3] + cc[3];
        if (d->history_limit > 0 && n > d->history_limit) n = d->history_limit;
        const float alpha = sat(cc[3] / n);
        for (int k = 0; k < 3; k++) oc[k] = mixf(cp[k], cc[k], alpha);
        oc[3] = n;
        om[0] = mixf(mp[0], l, alpha);
        om[1] = mixf(mp[1], l * l, alpha);
      } else {
        memcpy(oc, cc, 16);
        om[0] = l;
        om[1] = l * l;
      }
    }
}

// gInput/gAlbedo/gOutput: RGBA32F of width*height; out_max[4] = the maxima main() sees
void orc_tonemap_state(const float* input, const float* albedo, float* output, uint32_t width, uint32_t height, uint32_t mode, uint32_t modulate, uint32_t gamma, float exposure, float* out_max,
                       float exposure_alpha, float* exposure_state);
void orc_tonemap(const float* input, const float* albedo, float* output, uint32_t width, uint32_t height, uint32_t mode, uint32_t modulate, uint32_t gamma, float exposure, float* out_max) {
  orc_tonemap_state(input, albedo, output, width, height, mode, modulate, gamma, exposure, out_max, 0.0f, nullptr);
}
// exposure_state: in = gPrevMax bytes 16..39 (the previous frame's blended maxima and luminance moments), out = this frame's
void orc_tonemap_state(const float* input, const float* albedo, float* output, uint32_t width, uint32_t height, uint32_t mode, uint32_t modulate, uint32_t gamma, float exposure, float* out_max,
                       float exposure_alpha, float* exposure_state) {
  const size_t n = (size_t)width * height;
  uint32_t mx[4] = {0, 0, 0, 0};
  {
    for (size_t i = 0; i < n; i++) {  // reduce_max (BDPT.cpp:788-801 runs it for every mode)
      float v[4] = {input[4 * i], input[4 * i + 1], input[4 * i + 2], 0};
      if (modulate)
        for (int c = 0; c < 3; c++) v[c] *= albedo[4 * i + c];
      v[3] = lum(v);
      if (v[0] != v[0] || v[1] != v[1] || v[2] != v[2] || v[3] != v[3] || v[3] <= 0) continue;
      for (int c = 0; c < 4; c++) {
        float q = v[c] * 16384.0f;
        q = q < 0 ? 0 : q;
        const uint32_t u = q >= 4294967295.0f ? 0xFFFFFFFFu : (uint32_t)q;
        if (u > mx[c]) mx[c] = u;
      }
    }
  }
  float cmax[4];
  for (int c = 0; c < 4; c++) cmax[c] = (float)mx[c] / 16384.0f;
  if (out_max) memcpy(out_max, cmax, sizeof(cmax));
  {  // tonemap.hlsl:168-182: exposure smoothing over frames
    float m0 = cmax[3], m1 = cmax[3] * cmax[3];
    if (exposure_state && exposure_alpha > 0 && exposure_alpha < 1) {
      const float* pv = exposure_state;
      if (pv[4] == pv[4] && pv[5] == pv[5] && pv[4] > 0) {
        const float sa = sqrtf(exposure_alpha);
        m0 = pv[4] + sa * (m0 - pv[4]);
        m1 = pv[5] + sa * (m1 - pv[5]);
      }
      if (pv[0] == pv[0] && pv[1] == pv[1] && pv[2] == pv[2] && pv[3] == pv[3] && pv[3] > 0)
        for (int c = 0; c < 4; c++) cmax[c] = pv[c] + exposure_alpha * (cmax[c] - pv[c]);
    }
    if (exposure_state) {
      memcpy(exposure_state, cmax, 16);
      exposure_state[4] = m0;
      exposure_state[5] = m1;
    }
  }
  const float gain = det_expf(exposure * 0.693147180559945f);
  for (size_t i = 0; i < n; i++) {  // main
    float r[3] = {input[4 * i], input[4 * i + 1], input[4 * i + 2]};
    if (modulate)
      for (int c = 0; c < 3; c++) r[c] *= (1e-2f + albedo[4 * i + c]);
    for (int c = 0; c < 3; c++) r[c] *= gain;
    const float l = lum(r);
    switch (mode) {
      case STHIP_TONEMAP_REINHARD:
        for (int c = 0; c < 3; c++) {
          const float tc = r[c] / (1.0f + r[c]);
          const float a = r[c] / (1 + l);
          r[c] = a + tc * (tc - a);
        }
        break;
      case STHIP_TONEMAP_REINHARD_EXTENDED:
        for (int c = 0; c < 3; c++) {
          const float m = cmax[c] == 0 ? 1.0f : cmax[c];
          r[c] = r[c] / (1.0f + r[c]) * (1.0f + r[c] / (m * m));
        }
        break;
      case STHIP_TONEMAP_REINHARD_LUMINANCE: {
        const float l1 = l / (1 + l);
        for (int c = 0; c < 3; c++) r[c] = r[c] * (l1 / l);
        break;
      }
      case STHIP_TONEMAP_REINHARD_LUMINANCE_EXTENDED: {
        const float m = cmax[3] == 0 ? 1 : cmax[3];
        const float l1 = (l / (1 + l)) * (1 + l / (m * m));
        for (int c = 0; c < 3; c++) r[c] = r[c] * (l1 / l);
        break;
      }
      case STHIP_TONEMAP_UNCHARTED2: {
        const float d = u2p(cmax[3] == 0 ? 1 : cmax[3]);
        for (int c = 0; c < 3; c++) r[c] = u2p(r[c]) / d;
        break;
      }
      case STHIP_TONEMAP_FILMIC:
        for (int c = 0; c < 3; c++) {
          const float x = fmaxf(0.0f, r[c] - 0.004f);
          r[c] = (x * (6.2f * x + 0.5f)) / (x * (6.2f * x + 1.7f) + 0.06f);
        }
        break;
      case STHIP_TONEMAP_ACES: {
        static const float Min[3][3] = {{0.59719f, 0.35458f, 0.04823f}, {0.07600f, 0.90834f, 0.01566f}, {0.02840f, 0.13383f, 0.83777f}};
        static const float Mout[3][3] = {{1.60475f, -0.53108f, -0.07367f}, {-0.10208f, 1.10813f, -0.00605f}, {-0.00327f, -0.07276f, 1.07602f}};
        float v[3], f[3];
        for (int c = 0; c < 3; c++) v[c] = Min[c][0] * r[0] + Min[c][1] * r[1] + Min[c][2] * r[2];
        for (int c = 0; c < 3; c++) f[c] = (v[c] * (v[c] + 0.0245786f) - 0.000090537f) / (v[c] * (0.983729f * v[c] + 0.4329510f) + 0.238081f);
        for (int c = 0; c < 3; c++) r[c] = sat(Mout[c][0] * f[0] + Mout[c][1] * f[1] + Mout[c][2] * f[2]);
        break;
      }
      case STHIP_TONEMAP_ACES_APPROX:
        for (int c = 0; c < 3; c++) {
          const float v = r[c] * 0.6f;
          r[c] = sat((v * (2.51f * v + 0.03f)) / (v * (2.43f * v + 0.59f) + 0.14f));
        }
        break;
      case STHIP_TONEMAP_VIRIDIS_R: viridis(sat(l), r); break;
      case STHIP_TONEMAP_VIRIDIS_LENGTH_RGB: viridis(sat(l / (cmax[3] == 0 ? 1.0f : cmax[3])), r); break;
      default: break;
    }
    if (gamma)
      for (int c = 0; c < 3; c++) r[c] = r[c] <= 0.0031308f ? r[c] * 12.92f : det_powf(r[c] * 1.055f, 1 / 2.4f) - 0.055f;
    output[4 * i] = r[0];
    output[4 * i + 1] = r[1];
    output[4 * i + 2] = r[2];
    output[4 * i + 3] = 1.0f;
  }
}

void orc_image_compare(const float* image1, const float* image2, uint32_t width, uint32_t height, uint32_t metric, uint32_t quantization, uint32_t* sum_out, uint32_t* overflow_out) {
  const uint32_t n = width * height;
  uint32_t acc = 0, ovf = 0;
  for (uint32_t g = 0; g < n; g += 64) {
    float s = 0;
    for (uint32_t i = g; i < g + 64; i++) {
      float e = 0;
      if (i < n) {
        const float* a = image1 + 4 * (size_t)i;
        const float* b = image2 + 4 * (size_t)i;
        float t[3];
        for (int c = 0; c < 3; c++) {
          const float d = a[c] - b[c];
          t[c] = metric == STHIP_COMPARE_SMAPE ? fabsf(d) / (fabsf(a[c]) + fabsf(b[c])) : (metric == STHIP_COMPARE_MSE ? d * d : d);
        }
        e = (t[0] + t[1] + t[2]) / (float)(3u * n);
      }
      s += e;
    }
    const float valf = s * (float)quantization;
    const uint32_t val = valf >= 4294967295.0f ? 0xFFFFFFFFu : (valf > 0 ? (uint32_t)valf : 0u);
    if (valf >= 4294967295.0f || 0xFFFFFFFFu - val < acc) ovf = 1;
    acc += val;
  }
  *sum_out = acc;
  if (overflow_out) *overflow_out = ovf;
}

}  // extern "C"
