// device_math.h — scalar/vector helpers of the HIP path tracer.
//
// Arithmetic rules (include/sthip_detmath.h): IEEE binary32, compiled with -ffp-contract=off,
// fused multiply-adds only where fmaf() is written. Each helper states its evaluation order
// because the order is part of the boundary's contract (the CPU oracle must reach the same bits).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sthip_detmath.h"
#include "../../include/sthip_wire.h"

#define DEV __device__ __forceinline__

struct f3 {
  float x, y, z;
};
DEV f3 F3(float x, float y, float z) {
  f3 r;
  r.x = x;
  r.y = y;
  r.z = z;
  return r;
}
DEV f3 F3s(float s) { return F3(s, s, s); }
DEV f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
DEV f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
DEV f3 operator*(float s, f3 a) { return F3(s * a.x, s * a.y, s * a.z); }
DEV f3 operator/(f3 a, float s) { return F3(a.x / s, a.y / s, a.z / s); }
DEV f3 operator-(f3 a) { return F3(-a.x, -a.y, -a.z); }
DEV float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV f3 cross3(f3 a, f3 b) { return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV float len_sqr(f3 a) { return dot3(a, a); }
DEV float length3(f3 a) { return sqrtf(dot3(a, a)); }
// HLSL normalize pinned as one division and three multiplies
DEV f3 normalize3(f3 a) {
  const float inv = 1.0f / sqrtf(dot3(a, a));
  return a * inv;
}
DEV float pow2f(float x) { return x * x; }
DEV bool all_le0(f3 a) { return a.x <= 0 && a.y <= 0 && a.z <= 0; }
DEV bool any_gt0(f3 a) { return a.x > 0 || a.y > 0 || a.z > 0; }
DEV bool any_nan(f3 a) { return a.x != a.x || a.y != a.y || a.z != a.z; }
DEV float sgnf(float x) { return (float)((x > 0) - (x < 0)); }
DEV float lerp1(float a, float b, float t) { return a + t * (b - a); }
DEV f3 lerp3(f3 a, f3 b, float t) { return a + (b - a) * t; }
DEV float luminance3(f3 c) { return dot3(c, F3(0.2126f, 0.7152f, 0.0722f)); }  // common.h:66-68
DEV float comp3(f3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

DEV f3 ld3(const float* p) { return F3(p[0], p[1], p[2]); }
DEV f3 xyz(float4 v) { return F3(v.x, v.y, v.z); }

// TransformData, transform.h:9-23: rows dotted left to right
struct Xf {
  float4 r0, r1, r2;
};
DEV Xf load_xf(const sthip_TransformData* t, uint32_t i) {
  const float4* p = reinterpret_cast<const float4*>(t + i);
  Xf x;
  x.r0 = p[0];
  x.r1 = p[1];
  x.r2 = p[2];
  return x;
}
DEV f3 xf_vector(const Xf& t, f3 v) {
  return F3(t.r0.x * v.x + t.r0.y * v.y + t.r0.z * v.z, t.r1.x * v.x + t.r1.y * v.y + t.r1.z * v.z, t.r2.x * v.x + t.r2.y * v.y + t.r2.z * v.z);
}
DEV f3 xf_point(const Xf& t, f3 v) {
  return F3(t.r0.x * v.x + t.r0.y * v.y + t.r0.z * v.z + t.r0.w, t.r1.x * v.x + t.r1.y * v.y + t.r1.z * v.z + t.r1.w,
            t.r2.x * v.x + t.r2.y * v.y + t.r2.z * v.z + t.r2.w);
}
DEV float xf_at(const Xf& t, int i, int j) {
  const float4 r = i == 0 ? t.r0 : (i == 1 ? t.r1 : t.r2);
  return j == 0 ? r.x : (j == 1 ? r.y : (j == 2 ? r.z : r.w));
}
// tmul, transform.h:88-104: lhs * [rhs; 0 0 0 1]
DEV Xf xf_mul(const Xf& a, const Xf& b) {
  Xf r;
  float4* rows[3] = {&r.r0, &r.r1, &r.r2};
  const float4 ar[3] = {a.r0, a.r1, a.r2};
  for (int i = 0; i < 3; i++) {
    rows[i]->x = ar[i].x * b.r0.x + ar[i].y * b.r1.x + ar[i].z * b.r2.x;
    rows[i]->y = ar[i].x * b.r0.y + ar[i].y * b.r1.y + ar[i].z * b.r2.y;
    rows[i]->z = ar[i].x * b.r0.z + ar[i].y * b.r1.z + ar[i].z * b.r2.z;
    rows[i]->w = (ar[i].x * b.r0.w + ar[i].y * b.r1.w + ar[i].z * b.r2.w) + ar[i].w;
  }
  return r;
}

// common.h:125-132
DEV void make_orthonormal(f3 N, f3& T, f3& B) {
  if (N.x != N.y || N.x != N.z)
    T = F3(N.z - N.y, N.x - N.z, N.y - N.x);
  else
    T = F3(N.z - N.y, N.x + N.z, -N.y - N.x);
  T = normalize3(T);
  B = cross3(N, T);
}
// common.h:154-161
DEV f3 sample_cos_hemisphere(float u1, float u2) {
  const float phi = DET_2PI * u2;
  float s, c;
  det_sincosf(phi, &s, &c);
  const float r = sqrtf(u1);
  const float x = r * c, y = r * s;
  return F3(x, y, sqrtf(fmaxf(0.f, 1.0f - (x * x + y * y))));
}
DEV float cosine_hemisphere_pdfW(float cos_theta) { return fmaxf(cos_theta, 0.f) / DET_PI; }
// common.h:184-190
DEV float ray_plane(f3 origin, f3 dir, f3 normal) {
  const float denom = dot3(normal, dir);
  if (fabsf(denom) > 0) return -dot3(origin, normal) / denom;
  return __builtin_inff();
}
