/* sthip_detmath.h — the deterministic-arithmetic contract of the boundary.
 *
 * The reference evaluates sin/cos/pow/log/f32tof16 through vendor shader
 * intrinsics compiled with fast-math (Shader.cpp:111), i.e. their results are
 * implementation-defined (SURVEY.md §8c "parity unpinned", items 3-4). Parity at
 * 1e-4 relative L2 on a 1-spp framebuffer needs every branch to be taken
 * identically on the CPU and the GPU, so the boundary pins those intrinsics to
 * the definitions in this header. Both sides of the ABI (the HIP kernels and
 * the CPU oracle) evaluate them from this one text; everything else (integrator,
 * traversal, shading) is written independently on each side.
 *
 * Rules for code on either side of the boundary:
 *   - IEEE binary32, round-to-nearest-even, denormals kept, no fast-math;
 *   - no implicit contraction (-ffp-contract=off); a fused multiply-add
 *     happens only where fmaf() is written;
 *   - +, -, *, /, sqrtf are correctly rounded (hipcc default
 *     -fhip-fp32-correctly-rounded-divide-sqrt; SSE2 on the host).
 * Polynomials follow the Cephes single-precision library (Moshier), which is
 * the published algorithm these definitions restate.
 */
#ifndef STHIP_DETMATH_H
#define STHIP_DETMATH_H

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define STHIP_HD __host__ __device__ __forceinline__
#else
#define STHIP_HD static inline
#endif

STHIP_HD uint32_t det_f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
STHIP_HD float det_u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

#define DET_PI 3.14159274101257324f       /* (float)M_PI */
#define DET_2PI 6.28318548202514648f      /* (float)(2*M_PI) */
#define DET_INV_PI 0.318309873342514038f  /* (float)(1/M_PI) */
#define DET_2PI2 19.7392082214355469f     /* (float)(2*M_PI*M_PI), environment.h:73,92 */
#define DET_INV_4PI 0.0795774683356285095f /* (float)(1/(4*M_PI)), common.h:153 */

/* sin and cos of x (|x| up to a few thousand), ~1-2 ulp.
 * Cody-Waite reduction by pi/2 in three fmaf steps, Cephes sinf/cosf kernels. */
STHIP_HD void det_sincosf(float x, float* s_out, float* c_out) {
  const float kf = floorf(x * 0.636619772367581343f + 0.5f);
  const int k = (int)kf;
  float r = fmaf(kf, -1.5703125f, x);
  r = fmaf(kf, -4.837512969970703125e-4f, r);
  r = fmaf(kf, -7.54978995489188216e-8f, r);
  const float z = r * r;
  float sp = -1.9515295891e-4f;
  sp = sp * z + 8.3321608736e-3f;
  sp = sp * z - 1.6666654611e-1f;
  const float s = sp * z * r + r;
  float cp = 2.443315711809948e-5f;
  cp = cp * z - 1.388731625493765e-3f;
  cp = cp * z + 4.166664568298827e-2f;
  const float c = cp * z * z - 0.5f * z + 1.0f;
  switch (k & 3) {
    case 0: *s_out = s; *c_out = c; break;
    case 1: *s_out = c; *c_out = -s; break;
    case 2: *s_out = -s; *c_out = -c; break;
    default: *s_out = -c; *c_out = s; break;
  }
}

/* natural log of a positive normal float (Cephes logf) */
STHIP_HD float det_logf(float x) {
  const uint32_t u = det_f2u(x);
  int e = (int)((u >> 23) & 0xFF) - 126;
  float m = det_u2f((u & 0x007FFFFFu) | 0x3F000000u); /* [0.5, 1) */
  if (m < 0.707106781186547524f) {
    e -= 1;
    m = m + m - 1.0f;
  } else {
    m = m - 1.0f;
  }
  const float z = m * m;
  float y = 7.0376836292e-2f;
  y = y * m - 1.1514610310e-1f;
  y = y * m + 1.1676998740e-1f;
  y = y * m - 1.2420140846e-1f;
  y = y * m + 1.4249322787e-1f;
  y = y * m - 1.6668057665e-1f;
  y = y * m + 2.0000714765e-1f;
  y = y * m - 2.4999993993e-1f;
  y = y * m + 3.3333331174e-1f;
  y = y * m * z;
  const float fe = (float)e;
  y = y + -2.12194440e-4f * fe;
  y = y + -0.5f * z;
  float r = m + y;
  r = r + 0.693359375f * fe;
  return r;
}

/* log2 of a positive normal float (texture LOD, image_value.h:86,94) */
STHIP_HD float det_log2f(float x) { return det_logf(x) * 1.44269504088896341f; }
// tan as the quotient of the pinned sin and cos (only the hash grid's cell size needs it, hashgrid.hlsli:12)
STHIP_HD float det_tanf(float x) {
  float s, c;
  det_sincosf(x, &s, &c);
  return s / c;
}

/* e^x for |x| < 87 (Cephes expf) */
STHIP_HD float det_expf(float x) {
  const float n = floorf(1.44269504088896341f * x + 0.5f);
  x = x - n * 0.693359375f;
  x = x - n * -2.12194440e-4f;
  const float z = x * x;
  float p = 1.9875691500e-4f;
  p = p * x + 1.3981999507e-3f;
  p = p * x + 8.3334519073e-3f;
  p = p * x + 4.1665795894e-2f;
  p = p * x + 1.6666665459e-1f;
  p = p * x + 5.0000001201e-1f;
  p = p * z + x + 1.0f;
  const int ni = (int)n;
  return p * det_u2f((uint32_t)(ni + 127) << 23);
}

/* pow(a, b) for a > 0 */
STHIP_HD float det_powf(float a, float b) { return det_expf(b * det_logf(a)); }

/* atan(x), Cephes atanf: range reduction at tan(pi/8), tan(3pi/8) */
STHIP_HD float det_atanf(float xx) {
  float x = fabsf(xx), y;
  if (x > 2.414213562373095f) {
    y = 1.5707963267948966192f;
    x = -(1.0f / x);
  } else if (x > 0.4142135623730950f) {
    y = 0.7853981633974483096f;
    x = (x - 1.0f) / (x + 1.0f);
  } else {
    y = 0.0f;
  }
  const float z = x * x;
  float p = 8.05374449538e-2f;
  p = p * z - 1.38776856032e-1f;
  p = p * z + 1.99777106478e-1f;
  p = p * z - 3.33329491539e-1f;
  y = y + (p * z * x + x);
  return xx < 0.0f ? -y : y;
}

/* atan2(y, x) as the reference's stable_atan2 uses it (common.h:134-136): x == 0 is decided by the caller's
 * rule (y == 0 -> 0, else +-pi/2), everything else goes through atan(y / x) and the quadrant */
STHIP_HD float det_atan2f(float y, float x) {
  if (x == 0.0f) return y == 0.0f ? 0.0f : (y < 0.0f ? -1.5707963267948966192f : 1.5707963267948966192f);
  if (y == 0.0f) return x < 0.0f ? DET_PI : 0.0f;
  const float z = det_atanf(y / x);
  if (x > 0.0f) return z;
  return y < 0.0f ? z - DET_PI : z + DET_PI;
}

/* asin / acos on [-1, 1], Cephes asinf / acosf */
STHIP_HD float det_asinf(float xx) {
  const float a = fabsf(xx);
  if (a < 1.0e-4f) return xx;
  float x, z;
  const int big = a > 0.5f;
  if (big) {
    z = 0.5f * (1.0f - a);
    x = sqrtf(z);
  } else {
    x = a;
    z = x * x;
  }
  float p = 4.2163199048e-2f;
  p = p * z + 2.4181311049e-2f;
  p = p * z + 4.5470025998e-2f;
  p = p * z + 7.4953002686e-2f;
  p = p * z + 1.6666752422e-1f;
  float r = p * z * x + x;
  if (big) r = 1.5707963267948966192f - (r + r);
  return xx < 0.0f ? -r : r;
}
STHIP_HD float det_acosf(float x) {
  if (x > 0.5f) return 2.0f * det_asinf(sqrtf(0.5f * (1.0f - x)));
  if (x < -0.5f) return DET_PI - 2.0f * det_asinf(sqrtf(0.5f * (1.0f + x)));
  return 1.5707963267948966192f - det_asinf(x);
}

/* pow(x, 5) as written by the reference's own pow5 (common.h:45-49): pow4(x)*x */
STHIP_HD float det_pow5f(float x) {
  const float x2 = x * x;
  return (x2 * x2) * x;
}

/* f32 -> f16 bits, round-to-nearest-even, and back (HLSL f32tof16 / f16tof32,
 * bitfield.h:56-75). Software on both sides so the rounding mode is explicit. */
STHIP_HD uint32_t det_f32tof16(float f) {
  const uint32_t x = det_f2u(f);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const uint32_t ax = x & 0x7FFFFFFFu;
  if (ax >= 0x7F800000u) return sign | 0x7C00u | ((ax > 0x7F800000u) ? 0x200u : 0u); /* inf / nan */
  if (ax >= 0x477FF000u) return sign | 0x7C00u;                                       /* overflow -> inf */
  if (ax < 0x33000001u) return sign;                                                  /* underflow -> 0 */
  if (ax < 0x38800000u) {                                                             /* f16 subnormal */
    const uint32_t shift = 126u - (ax >> 23);          /* 14..24 */
    const uint32_t mant = (ax & 0x7FFFFFu) | 0x800000u; /* 24-bit */
    const uint32_t half = mant >> shift;
    const uint32_t rem = mant & ((1u << shift) - 1u);
    const uint32_t mid = 1u << (shift - 1u);
    uint32_t h = half;
    if (rem > mid || (rem == mid && (half & 1u))) h += 1u;
    return sign | h;
  }
  uint32_t h = ((ax - 0x38000000u) >> 13);
  const uint32_t rem = ax & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h += 1u;
  return sign | h;
}

STHIP_HD float det_f16tof32(uint32_t h) {
  const uint32_t sign = (h & 0x8000u) << 16;
  const uint32_t e = (h >> 10) & 0x1Fu;
  const uint32_t m = h & 0x3FFu;
  if (e == 0u) {
    if (m == 0u) return det_u2f(sign);
    /* subnormal: value = m * 2^-24, exact in f32 */
    const float v = (float)m * 5.9604644775390625e-8f;
    return det_u2f(det_f2u(v) | sign);
  }
  if (e == 31u) return det_u2f(sign | 0x7F800000u | (m << 13));
  return det_u2f(sign | ((e + 112u) << 23) | (m << 13));
}

#endif /* STHIP_DETMATH_H */
