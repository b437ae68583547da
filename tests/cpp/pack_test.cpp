// pack_test.cpp — the conservative plane rounding of the 48-byte node (stratum_amd/csrc/bvh_build.h: pack_plane / pack_node):
// whatever byte replaces the low mantissa byte of a packed plane, a lower plane must read <= the true value and an upper
// plane >= it, no byte may produce a NaN or an infinity, the error stays within 511 ulp, and the child references come back
// from the low bytes exactly. Host-only (no GPU).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "../../stratum_amd/csrc/bvh_build.h"

static float as_float(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static uint32_t as_uint(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
static long long ordered(float f) {  // monotone integer key of a float
  const uint32_t u = as_uint(f);
  return (u >> 31) ? -(long long)(u & 0x7FFFFFFFu) : (long long)(u & 0x7FFFFFFFu);
}

int main() {
  std::mt19937 rng(1234);
  std::vector<float> values = {0.0f, -0.0f, 1.0f, -1.0f, 1e-45f, -1e-45f, 3.5e-43f, -3.5e-43f, 1e-38f, -1e-38f, 15.029797f, -15.029563f, 1e30f, -1e30f, 3.3e38f, -3.3e38f, INFINITY, -INFINITY, NAN};
  for (int i = 0; i < 200000; i++) {
    const uint32_t u = rng();
    const float f = as_float(u);
    values.push_back(f);
    values.push_back((float)((int)(rng() % 2000001) - 1000000) * 1e-4f);
  }
  long long worst = 0;
  for (float v : values)
    for (int upper = 0; upper < 2; upper++) {
      const uint32_t packed = sthip::pack_plane(v, upper != 0, 0xA5u);
      if ((packed & 0xFFu) != 0xA5u) {
        printf("FAIL: byte not stored\n");
        return 1;
      }
      float clamped = v;
      if (!(clamped > -3.0e38f)) clamped = -3.0e38f;
      if (clamped > 3.0e38f) clamped = 3.0e38f;
      for (uint32_t b = 0; b < 256; b++) {
        const float r = as_float((packed & ~0xFFu) | b);
        if (!std::isfinite(r)) {
          printf("FAIL: %g (upper %d) byte %u reads %g\n", v, upper, b, r);
          return 1;
        }
        if (upper ? !(r >= clamped) : !(r <= clamped)) {
          printf("FAIL: %g (upper %d) byte %u reads %.9g: not conservative\n", v, upper, b, r);
          return 1;
        }
        const long long err = std::llabs(ordered(r) - ordered(clamped));
        if (err > worst) worst = err;
      }
    }
  if (worst > 511) {
    printf("FAIL: worst error %lld ulp\n", worst);
    return 1;
  }
  // whole nodes: references round-trip, z planes exact
  for (int i = 0; i < 20000; i++) {
    BvhNode n;
    for (int k = 0; k < 4; k++) {
      n.n0xy[k] = (float)((int)(rng() % 40001) - 20000) * 1e-3f;
      n.n1xy[k] = (float)((int)(rng() % 40001) - 20000) * 1e-3f;
      n.nz[k] = (float)((int)(rng() % 40001) - 20000) * 1e-3f;
    }
    n.ref[0] = rng();
    n.ref[1] = rng();
    n.ref[2] = n.ref[3] = 0;
    const BvhNodePacked q = sthip::pack_node(n);
    uint32_t r0 = 0, r1 = 0;
    for (int k = 0; k < 4; k++) {
      r0 |= (as_uint(q.n0xy[k]) & 0xFFu) << (8 * k);
      r1 |= (as_uint(q.n1xy[k]) & 0xFFu) << (8 * k);
      if (q.nz[k] != n.nz[k]) {
        printf("FAIL: z plane changed\n");
        return 1;
      }
      const bool upper = (k & 1) != 0;
      if (upper ? !(q.n0xy[k] >= n.n0xy[k] && q.n1xy[k] >= n.n1xy[k]) : !(q.n0xy[k] <= n.n0xy[k] && q.n1xy[k] <= n.n1xy[k])) {
        printf("FAIL: node plane not conservative\n");
        return 1;
      }
    }
    if (r0 != n.ref[0] || r1 != n.ref[1]) {
      printf("FAIL: reference did not round-trip\n");
      return 1;
    }
  }
  printf("PACK OK: worst plane error %lld ulp\n", worst);
  return 0;
}
