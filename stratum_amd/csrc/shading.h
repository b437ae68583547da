// shading.h — device restatement of the reference's shading code for the HIP wavefront kernels.
// Citations are relative to /root/reference/src/Shaders.
#pragma once

#include "device_math.h"

// ---------------------------------------------------------------------------------------------
// scene views as the kernels see them (device pointers to the arrays of sthip_scene_desc)
// ---------------------------------------------------------------------------------------------
struct DeviceScene {
  const sthip_PackedVertexData* vertices;
  const uint8_t* indices;
  const sthip_InstanceData* instances;
  const sthip_TransformData* xf;
  const sthip_TransformData* inv_xf;
  const sthip_TransformData* motion_xf;
  const uint8_t* materials;
  const uint32_t* lights;
  uint32_t instance_count;
  uint32_t light_count;
  // Texture2D<float4> gImages[] (bdpt.hlsl:33): mip chains packed into one float4 buffer
  const struct DeviceImage* images;
  const float4* image_texels;
  uint32_t image_count;
  const float* distributions;  // StructuredBuffer<float> gDistributions (environment map tables, dist2.h)
  uint32_t distribution_count;
  const uint32_t* volume_words;  // ByteAddressBuffer gVolumes[]: the NanoVDB grids back to back
  const DeviceVolume* volumes;   // their parsed headers (media.h)
  uint32_t volume_count;
  // the leaf triangles of the acceleration structure (bvh.h: BvhTri, three float4: the positions) and, in the same order, what
  // else a hit needs of its three vertices (bvh.h: BvhTriShade, four float4: normals and uvs): a hit carries its leaf index
  // (RayHit::leaf), so k_shade reads both at once instead of walking hit -> instance -> indices -> vertices
  const float4* leaf_tris;
  const float4* leaf_shade;
};

#define STHIP_MAX_MIPS 16
struct DeviceImage {
  uint32_t offset[STHIP_MAX_MIPS];  // first texel of each level in image_texels
  uint16_t w[STHIP_MAX_MIPS], h[STHIP_MAX_MIPS];
  uint32_t levels;
  uint32_t pad[3];
};

// scene.h:29-47
struct Inst {
  uint4 p;
  DEV uint32_t type() const { return p.x & 0xFu; }
  DEV uint32_t material_address() const { return p.x >> 4; }
  DEV uint32_t light_index() const { return p.y & 0xFFFu; }
  DEV uint32_t prim_count() const { return (p.y >> 12) & 0xFFFFu; }
  DEV uint32_t index_stride() const { return p.y >> 28; }
  DEV uint32_t first_vertex() const { return p.z; }
  DEV uint32_t indices_byte_offset() const { return p.w; }
  DEV float radius() const { return __uint_as_float(p.z); }  // sphere instances, scene.h:43
};
DEV Inst load_inst(const DeviceScene& sc, uint32_t i) {
  Inst r;
  r.p = *reinterpret_cast<const uint4*>(sc.instances + i);
  return r;
}

// scene.h:139-161 load_tri_: the index buffer is padded by 8 bytes at upload so Load2 never runs off the end
DEV void load_tri(const DeviceScene& sc, const Inst& in, uint32_t prim, uint32_t& i0, uint32_t& i1, uint32_t& i2) {
  const uint32_t stride = in.index_stride();
  const uint32_t off = in.indices_byte_offset() + prim * 3u * stride;
  if (stride == 2u) {
    const uint32_t aligned = off & ~3u;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(sc.indices + aligned);
    const uint32_t w0 = w[0], w1 = w[1];
    if (aligned == off) {
      i0 = w0 & 0xffffu;
      i1 = w0 >> 16;
      i2 = w1 & 0xffffu;
    } else {
      i0 = w0 >> 16;
      i1 = w1 & 0xffffu;
      i2 = w1 >> 16;
    }
  } else {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(sc.indices + off);
    i0 = w[0];
    i1 = w[1];
    i2 = w[2];
  }
  i0 += in.first_vertex();
  i1 += in.first_vertex();
  i2 += in.first_vertex();
}

// ---------------------------------------------------------------------------------------------
// R1/R2 — rng.hlsli:22-47
// ---------------------------------------------------------------------------------------------
DEV uint32_t pcg4d_x(uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
  x = x * 1664525u + 1013904223u;
  y = y * 1664525u + 1013904223u;
  z = z * 1664525u + 1013904223u;
  w = w * 1664525u + 1013904223u;
  x += y * w;
  y += z * x;
  z += x * y;
  w += y * z;
  x ^= x >> 16;
  y ^= y >> 16;
  z ^= z >> 16;
  w ^= w >> 16;
  x += y * w;
  // the remaining three updates of pcg4d only feed .yzw, which rng_next_uint discards (rng.hlsli:35-38)
  return x;
}
struct Rng {
  uint32_t x, y, seed, counter;
  DEV uint32_t next_uint() {
    counter++;
    return pcg4d_x(x, y, seed, counter);
  }
  DEV float next_float() { return det_u2f(0x3f800000u | (next_uint() >> 9)) - 1.0f; }
};

// ---------------------------------------------------------------------------------------------
// R3 — bitfield.h:56-93
// ---------------------------------------------------------------------------------------------
DEV uint32_t pack_normal_octahedron(f3 v) {
  const float s = 1.0f / (fabsf(v.x) + fabsf(v.y) + fabsf(v.z));
  float px = v.x * s, py = v.y * s;
  if (v.z <= 0) {
    const float qx = (1.0f - fabsf(py)) * (px >= 0 ? 1.0f : -1.0f);
    const float qy = (1.0f - fabsf(px)) * (py >= 0 ? 1.0f : -1.0f);
    px = qx;
    py = qy;
  }
  return det_f32tof16(px) | (det_f32tof16(py) << 16);
}
DEV f3 unpack_normal_octahedron(uint32_t packed) {
  const float px = det_f16tof32(packed & 0xFFFFu), py = det_f16tof32(packed >> 16);
  f3 v = F3(px, py, 1.0f - (fabsf(px) + fabsf(py)));
  if (v.z < 0) {
    const float qx = (1.0f - fabsf(v.y)) * (v.x >= 0 ? 1.0f : -1.0f);
    const float qy = (1.0f - fabsf(v.x)) * (v.y >= 0 ? 1.0f : -1.0f);
    v.x = qx;
    v.y = qy;
  }
  return normalize3(v);
}

// ---------------------------------------------------------------------------------------------
// T3 — intersection.hlsli:44-62
// ---------------------------------------------------------------------------------------------
DEV f3 ray_offset(f3 pos, f3 n) {
  const float int_scale = 256.0f;
  const float origin = 1 / 32.0f;
  const float float_scale = 1 / 65536.0f;
  int32_t ox = (int32_t)(int_scale * n.x), oy = (int32_t)(int_scale * n.y), oz = (int32_t)(int_scale * n.z);
  if (pos.x < 0) ox = -ox;
  if (pos.y < 0) oy = -oy;
  if (pos.z < 0) oz = -oz;
  const float pix = det_u2f((uint32_t)((int32_t)det_f2u(pos.x) + ox));
  const float piy = det_u2f((uint32_t)((int32_t)det_f2u(pos.y) + oy));
  const float piz = det_u2f((uint32_t)((int32_t)det_f2u(pos.z) + oz));
  return F3(fabsf(pos.x) < origin ? pos.x + n.x * float_scale : pix, fabsf(pos.y) < origin ? pos.y + n.y * float_scale : piy,
            fabsf(pos.z) < origin ? pos.z + n.z * float_scale : piz);
}

// ---------------------------------------------------------------------------------------------
// W7/S1 — ShadingData, shading_data.h:10-37 and shading_data.hlsli:2-73.
// uv_screen_size / mean_curvature feed the texture LOD through ray cones (path.hlsli:224-244); in the untextured
// kernel instantiation nothing reads them and the compiler drops their computation.
// ---------------------------------------------------------------------------------------------
struct ShadingData {
  f3 position;
  uint32_t flags;
  uint32_t packed_geometry_normal, packed_shading_normal, packed_tangent;
  float shape_area;
  float u, v;
  float uv_screen_size, mean_curvature;
  DEV f3 geometry_normal() const { return unpack_normal_octahedron(packed_geometry_normal); }
  DEV f3 shading_normal() const { return unpack_normal_octahedron(packed_shading_normal); }
  DEV f3 tangent() const { return unpack_normal_octahedron(packed_tangent); }
};
// the local frame (shading_data.h:29-37) unpacked once; to_world/to_local then match the reference's
// per-call unpacking bit for bit because unpacking is deterministic
struct Frame3 {
  f3 n, t, b;
  DEV f3 to_world(f3 w) const { return w.x * t + w.y * b + w.z * n; }
  DEV f3 to_local(f3 w) const { return F3(dot3(w, t), dot3(w, b), dot3(w, n)); }
};
DEV Frame3 make_frame(const ShadingData& sd) {
  Frame3 f;
  f.n = sd.shading_normal();
  f.t = sd.tangent();
  f.b = cross3(f.n, f.t);
  return f;
}

// The ShadingData of a point on a triangle from its three vertices (positions, normals, uvs) — shading_data.hlsli:2-73
DEV void triangle_shading_data(const Xf& xf, ShadingData& r, f3 p0, f3 p1, f3 p2, f3 n0, f3 n1, f3 n2, float u0, float v0, float u1, float v1, float u2, float v2, float b1, float b2, bool flip_uvs);

// ... of triangle `prim` of an instance, through the scene arrays (instance -> indices -> vertices): light sampling
DEV void make_triangle_shading_data(const DeviceScene& sc, ShadingData& r, uint32_t inst_index, const Inst& in, uint32_t prim, float b1, float b2, bool flip_uvs = false) {
  const Xf xf = load_xf(sc.xf, inst_index);
  uint32_t i0, i1, i2;
  load_tri(sc, in, prim, i0, i1, i2);
  const float4* q0 = reinterpret_cast<const float4*>(sc.vertices + i0);
  const float4* q1 = reinterpret_cast<const float4*>(sc.vertices + i1);
  const float4* q2 = reinterpret_cast<const float4*>(sc.vertices + i2);
  const float4 a0 = q0[0], a1 = q0[1], c0 = q1[0], c1 = q1[1], e0 = q2[0], e1 = q2[1];
  triangle_shading_data(xf, r, xyz(a0), xyz(c0), xyz(e0), xyz(a1), xyz(c1), xyz(e1), a0.w, a1.w, c0.w, c1.w, e0.w, e1.w, b1, b2, flip_uvs);
}
// ... of a HIT: the leaf triangle the traversal found (its positions are the vertices' own, object space) and the record
// beside it (k_fill_tri_shade: the same vertices' normals and uvs) — the same twenty-four floats, two dependent loads earlier
DEV void make_hit_shading_data(const DeviceScene& sc, ShadingData& r, uint32_t inst_index, uint32_t leaf, float b1, float b2, bool flip_uvs = false) {
  const Xf xf = load_xf(sc.xf, inst_index);
  const float4* t = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sc.leaf_tris) + (size_t)leaf * 48u);
  const float4* h = sc.leaf_shade + (size_t)leaf * 4u;
  const float4 a0 = t[0], c0 = t[1], e0 = t[2];
  const float4 a1 = h[0], c1 = h[1], e1 = h[2], uu = h[3];
  triangle_shading_data(xf, r, xyz(a0), xyz(c0), xyz(e0), xyz(a1), xyz(c1), xyz(e1), uu.x, a1.w, uu.y, c1.w, uu.z, e1.w, b1, b2, flip_uvs);
}
DEV void triangle_shading_data(const Xf& xf, ShadingData& r, f3 p0, f3 p1, f3 p2, f3 n0, f3 n1, f3 n2, float u0, float v0, float u1, float v1, float u2, float v2, float b1, float b2, bool flip_uvs) {
  // :64-73
  const f3 v1v0 = p1 - p0, v2v0 = p2 - p0;
  const f3 local_position = p0 + v1v0 * b1 + v2v0 * b2;
  r.position = xf_point(xf, local_position);
  // :2-63
  r.u = u0 + (u1 - u0) * b1 + (u2 - u0) * b2;
  r.v = v0 + (v1 - v0) * b1 + (v2 - v0) * b2;
  if (flip_uvs) r.v = 1 - r.v;  // gFlipTriangleUVs, shading_data.hlsli:5-6
  const f3 dPds = xf_vector(xf, p0 - p2);
  const f3 dPdt = xf_vector(xf, p1 - p2);
  f3 geometry_normal = cross3(dPds, dPdt);
  const float area2 = length3(geometry_normal);
  geometry_normal = geometry_normal / area2;
  r.packed_geometry_normal = pack_normal_octahedron(geometry_normal);
  r.shape_area = area2 / 2;

  const float duvds0 = u2 - u0, duvds1 = v2 - v0;
  const float duvdt0 = u2 - u1, duvdt1 = v2 - v1;
  const float det = duvds0 * duvdt1 - duvdt0 * duvds1;
  const float inv_det = 1 / det;
  const float dsdu = duvdt1 * inv_det;
  const float dtdu = -duvds1 * inv_det;
  const float dsdv = duvdt0 * inv_det;
  const float dtdv = -duvds0 * inv_det;
  f3 dPdu, dPdv;
  if (det != 0) {
    dPdu = -(dPds * dsdu + dPdt * dtdu);
    dPdv = -(dPds * dsdv + dPdt * dtdv);
    r.uv_screen_size = 1 / fmaxf(length3(dPdu), length3(dPdv));
  } else {
    make_orthonormal(geometry_normal, dPdu, dPdv);
    r.uv_screen_size = 1;
  }
  f3 shading_normal = n0 + (n1 - n0) * b1 + (n2 - n0) * b2;
  if ((shading_normal.x == 0 && shading_normal.y == 0 && shading_normal.z == 0) || any_nan(shading_normal)) {
    r.packed_shading_normal = r.packed_geometry_normal;
    r.packed_tangent = pack_normal_octahedron(normalize3(dPdu));
    r.mean_curvature = 0;
  } else {
    shading_normal = normalize3(xf_vector(xf, shading_normal));
    const f3 tangent = normalize3(dPdu - shading_normal * dot3(shading_normal, dPdu));
    r.packed_shading_normal = pack_normal_octahedron(shading_normal);
    r.packed_tangent = pack_normal_octahedron(tangent);
    if (dot3(shading_normal, geometry_normal) < 0) r.packed_geometry_normal = pack_normal_octahedron(-geometry_normal);
    // :56-61
    const f3 dNds = n2 - n0, dNdt = n2 - n1;
    const f3 dNdu = dNds * dsdu + dNdt * dtdu;
    const f3 dNdv = dNds * dsdv + dNdt * dtdv;
    const f3 bitangent = normalize3(cross3(shading_normal, tangent));
    r.mean_curvature = (dot3(dNdu, tangent) + dot3(dNdv, bitangent)) / 2;
  }
  r.flags = 0;
}

// common.h:134-147
DEV void cartesian_to_spherical_uv(f3 v, float& u, float& vv) {
  const float theta = det_atan2f(v.z, v.x);
  u = theta * DET_INV_PI * .5f + .5f;
  vv = det_acosf(fminf(fmaxf(v.y, -1.f), 1.f)) * DET_INV_PI;
}
DEV f3 spherical_uv_to_cartesian(float u, float v) {
  u = u * 2 - 1;
  u *= DET_PI;
  v *= DET_PI;
  float su, cu, sv, cv;
  det_sincosf(u, &su, &cu);
  det_sincosf(v, &sv, &cv);
  return F3(sv * cu, cv, sv * su);
}

// shading_data.hlsli:93-105 (the angles of dpdu / dpdv are the uv themselves, as the reference writes them)
DEV void make_sphere_shading_data(const DeviceScene& sc, ShadingData& r, uint32_t inst_index, const Inst& in, f3 local_position) {
  const Xf t = load_xf(sc.xf, inst_index);
  const f3 normal = normalize3(xf_vector(t, local_position));
  r.position = xf_point(t, local_position);
  r.packed_geometry_normal = r.packed_shading_normal = pack_normal_octahedron(normal);
  const float radius = in.radius();
  r.shape_area = 4 * DET_PI * radius * radius;
  r.mean_curvature = 1 / radius;
  cartesian_to_spherical_uv(normalize3(local_position), r.u, r.v);
  float su, cu, sv, cv;
  det_sincosf(r.u, &su, &cu);
  det_sincosf(r.v, &sv, &cv);
  const f3 dpdu = xf_vector(t, F3(-su * sv, 0, cu * sv));
  const f3 dpdv = xf_vector(t, F3(cu * cv, -sv, su * cv));
  r.packed_tangent = pack_normal_octahedron(normalize3(dpdu - normal * dot3(normal, dpdu)));
  r.uv_screen_size = 1 / fmaxf(length3(dpdu), length3(dpdv));
  r.flags = 0;
}

// ---------------------------------------------------------------------------------------------
// S3/M1-M4 — DisneyMaterial: materials/disney_material.hlsli, disney_{diffuse,metal,glass,clearcoat}.hlsli, microfacet.h
// ---------------------------------------------------------------------------------------------
struct MaterialEvalRecord {
  f3 f;
  float pdf_fwd, pdf_rev;
};
struct MaterialSampleRecord {
  f3 dir_out;
  float pdf_fwd, pdf_rev, eta, roughness;
};

DEV float schlick_fresnel1(float F0, float cos_theta) { return F0 + (1 - F0) * det_pow5f(fmaxf(1.0f - cos_theta, 0.0f)); }
DEV f3 schlick_fresnel3(f3 F0, float cos_theta) { return F0 + (F3s(1.0f) - F0) * det_pow5f(fmaxf(1.0f - cos_theta, 0.0f)); }
DEV float fresnel_dielectric3(float n_dot_i, float n_dot_t, float eta) {
  const float rs = (n_dot_i - eta * n_dot_t) / (n_dot_i + eta * n_dot_t);
  const float rp = (eta * n_dot_i - n_dot_t) / (eta * n_dot_i + n_dot_t);
  return (rs * rs + rp * rp) / 2;
}
DEV float fresnel_dielectric(float n_dot_i, float eta) {
  const float n_dot_t_sq = 1 - (1 - n_dot_i * n_dot_i) / (eta * eta);
  if (n_dot_t_sq < 0) return 1;
  const float n_dot_t = sqrtf(n_dot_t_sq);
  return fresnel_dielectric3(fabsf(n_dot_i), n_dot_t, eta);
}
DEV float Dm(float ax, float ay, f3 h) {
  const float ax2 = ax * ax, ay2 = ay * ay;
  const f3 h2 = h * h;
  const float hh = h2.x / ax2 + h2.y / ay2 + h2.z;
  return 1 / (DET_PI * ax * ay * hh * hh);
}
DEV float G1(float ax, float ay, f3 w) {
  const float ax2 = ax * ax, ay2 = ay * ay;
  const f3 w2 = w * w;
  const float lambda = (sqrtf(1 + (w2.x * ax2 + w2.y * ay2) / w2.z) - 1) / 2;
  return 1 / (1 + lambda);
}
DEV float R0f(float eta) {
  const float num = eta - 1, denom = eta + 1;
  return (num * num) / (denom * denom);
}
DEV float Dc(float alpha_g, float h_lz) {
  const float a2 = alpha_g * alpha_g;
  return (a2 - 1) / (DET_PI * det_logf(a2) * (1 + (a2 - 1) * h_lz * h_lz));
}
DEV float Gc(f3 w) {
  const float wx = w.x * 0.25f, wy = w.y * 0.25f;
  const float lambda = (sqrtf(1 + (wx * wx + wy * wy) / (w.z * w.z)) - 1) / 2;
  return 1 / (1 + lambda);
}
DEV f3 reflect3(f3 i, f3 n) { return i - 2 * dot3(n, i) * n; }
DEV f3 refract3(f3 i, f3 n, float eta) {
  const float ni = dot3(n, i);
  const float k = 1 - eta * eta * (1 - ni * ni);
  if (k < 0) return F3s(0.0f);
  return eta * i - (eta * ni + sqrtf(k)) * n;
}
// microfacet.h:76-106
DEV f3 sample_visible_normals(f3 local_dir_in, float ax, float ay, float r0, float r1) {
  const bool inside = local_dir_in.z < 0;
  if (inside) local_dir_in = -local_dir_in;
  const f3 hemi_dir_in = normalize3(F3(ax * local_dir_in.x, ay * local_dir_in.y, local_dir_in.z));
  const float r = sqrtf(r0);
  const float phi = DET_2PI * r1;
  float sphi, cphi;
  det_sincosf(phi, &sphi, &cphi);
  const float t1 = r * cphi;
  float t2 = r * sphi;
  const float s = (1 + hemi_dir_in.z) / 2;
  t2 = (1 - s) * sqrtf(1 - t1 * t1) + s * t2;
  const f3 disk_N = F3(t1, t2, sqrtf(fmaxf(0.0f, 1 - t1 * t1 - t2 * t2)));
  f3 T1, T2;
  make_orthonormal(hemi_dir_in, T1, T2);
  const f3 hemi_N = disk_N.x * T1 + disk_N.y * T2 + disk_N.z * hemi_dir_in;
  f3 N = normalize3(F3(ax * hemi_N.x, ay * hemi_N.y, fmaxf(0.f, hemi_N.z)));
  if (inside) N = -N;
  return N;
}

struct DisneyMaterial {
  float4 d0, d1, d2;  // disney_data.h:1-20
  DEV f3 base_color() const { return xyz(d0); }
  DEV float emission() const { return d0.w; }
  DEV float metallic() const { return d1.x; }
  DEV float roughness() const { return d1.y; }
  DEV float anisotropic() const { return d1.z; }
  DEV float subsurface() const { return d1.w; }
  DEV float clearcoat() const { return d2.x; }
  DEV float clearcoat_gloss() const { return d2.y; }
  DEV float transmission() const { return d2.z; }
  DEV float eta() const { return d2.w; }
  DEV float alpha() const { return roughness() * roughness(); }

  // disney_material.hlsli:46-79 with no textures bound; records are 72 B at 4-byte alignment
  DEV void load(const DeviceScene& sc, uint32_t address) {
    const float* p = reinterpret_cast<const float*>(sc.materials + address);
    d0 = make_float4(p[0], p[1], p[2], p[3]);
    d1 = make_float4(p[5], p[6], p[7], p[8]);
    d2 = make_float4(p[10], p[11], p[12], p[13]);
  }

  // the same record out of the copy of gMaterialData a k_shade block holds in LDS (kernels.h: shade_lds)
  typedef __attribute__((address_space(3))) const float LdsFloat;
  DEV void load_lds(const LdsFloat* table, uint32_t address) {
    const LdsFloat* p = table + (address >> 2);
    d0 = make_float4(p[0], p[1], p[2], p[3]);
    d1 = make_float4(p[5], p[6], p[7], p[8]);
    d2 = make_float4(p[10], p[11], p[12], p[13]);
  }

  // sample_image, image_value.h:81-97: SampleLevel(gStaticSampler, uv, lod) as repeat addressing + trilinear
  // filtering over the box-filtered mip chain
  static DEV float4 texel(const DeviceScene& sc, const DeviceImage& im, uint32_t level, int x, int y) {
    const int w = (int)im.w[level], h = (int)im.h[level];
    x = ((x % w) + w) % w;
    y = ((y % h) + h) % h;
    return sc.image_texels[im.offset[level] + (uint32_t)y * (uint32_t)w + (uint32_t)x];
  }
  static DEV float4 bilinear(const DeviceScene& sc, const DeviceImage& im, uint32_t level, float u, float v) {
    const float x = u * (float)im.w[level] - 0.5f, y = v * (float)im.h[level] - 0.5f;
    const float x0 = floorf(x), y0 = floorf(y);
    const float fx = x - x0, fy = y - y0;
    const int ix = (int)x0, iy = (int)y0;
    const float4 c00 = texel(sc, im, level, ix, iy), c10 = texel(sc, im, level, ix + 1, iy);
    const float4 c01 = texel(sc, im, level, ix, iy + 1), c11 = texel(sc, im, level, ix + 1, iy + 1);
    float4 r;
    r.x = lerp1(lerp1(c00.x, c10.x, fx), lerp1(c01.x, c11.x, fx), fy);
    r.y = lerp1(lerp1(c00.y, c10.y, fx), lerp1(c01.y, c11.y, fx), fy);
    r.z = lerp1(lerp1(c00.z, c10.z, fx), lerp1(c01.z, c11.z, fx), fy);
    r.w = lerp1(lerp1(c00.w, c10.w, fx), lerp1(c01.w, c11.w, fx), fy);
    return r;
  }
  static DEV float4 sample_image(const DeviceScene& sc, uint32_t index, float u, float v, float uv_screen_size, bool ray_cones) {
    const DeviceImage& im = sc.images[index];
    float lod = 0;
    if (ray_cones && uv_screen_size > 0) lod = det_log2f(fmaxf(uv_screen_size * fmaxf((float)im.w[0], (float)im.h[0]), 1e-6f));
    const float top = (float)(im.levels - 1);
    lod = fminf(fmaxf(lod, 0.0f), top);
    const float l0 = floorf(lod);
    const uint32_t i0 = (uint32_t)l0, i1 = min(i0 + 1, im.levels - 1);
    const float f = lod - l0;
    const float4 a = bilinear(sc, im, i0, u, v), b = bilinear(sc, im, i1, u, v);
    return make_float4(lerp1(a.x, b.x, f), lerp1(a.y, b.y, f), lerp1(a.z, b.z, f), lerp1(a.w, b.w, f));
  }
  static DEV float4 eval_image_value4(const DeviceScene& sc, const float* p, float u, float v, float uvs, bool ray_cones) {  // image_value.h:194-198
    const float4 value = make_float4(p[0], p[1], p[2], p[3]);
    const uint32_t image_index = reinterpret_cast<const uint32_t*>(p)[4];
    if (image_index >= STHIP_IMAGE_COUNT) return value;
    if (!(value.x > 0 || value.y > 0 || value.z > 0 || value.w > 0)) return make_float4(0, 0, 0, 0);
    const float4 t = sample_image(sc, image_index, u, v, uvs, ray_cones);
    return make_float4(value.x * t.x, value.y * t.y, value.z * t.z, value.w * t.w);
  }
  // disney_material.hlsli:46-79 with image values and the normal map (flip_bitangent is never set on this path)
  DEV void load_textured(const DeviceScene& sc, uint32_t address, float u, float v, float uvs, uint32_t& packed_shading_normal, uint32_t& packed_tangent, uint32_t sampling_flags) {
    const float* p = reinterpret_cast<const float*>(sc.materials + address);
    const bool ray_cones = (sampling_flags >> STHIP_eRayCones) & 1u;
    d0 = eval_image_value4(sc, p, u, v, uvs, ray_cones);
    d1 = eval_image_value4(sc, p + 5, u, v, uvs, ray_cones);
    d2 = eval_image_value4(sc, p + 10, u, v, uvs, ray_cones);
    const uint32_t bump_index = reinterpret_cast<const uint32_t*>(p)[16];
    const float bump_strength = p[17];
    if (((sampling_flags >> STHIP_eNormalMaps) & 1u) && bump_index < STHIP_IMAGE_COUNT && bump_strength > 0) {
      const float4 t = sample_image(sc, bump_index, u, v, uvs, ray_cones);
      f3 bump = F3(1.0f * t.x, 1.0f * t.y, 1.0f * t.z) * 2 - F3s(1.0f);
      if ((sampling_flags >> STHIP_eFlipNormalMaps) & 1u) bump.y = -bump.y;
      bump = normalize3(F3(bump.x * bump_strength, bump.y * bump_strength, bump.z > 0 ? bump.z : 1.0f));
      f3 n = unpack_normal_octahedron(packed_shading_normal);
      f3 t3 = unpack_normal_octahedron(packed_tangent);
      n = normalize3(t3 * bump.x + cross3(n, t3) * bump.y + n * bump.z);
      t3 = normalize3(t3 - n * dot3(n, t3));
      packed_shading_normal = pack_normal_octahedron(n);
      packed_tangent = pack_normal_octahedron(t3);
    }
  }
  DEV f3 Le() const { return base_color() * emission(); }
  DEV f3 albedo() const { return base_color(); }
  DEV bool can_eval() const { return emission() <= 0 && any_gt0(base_color()); }
  DEV bool is_specular() const { return (metallic() > 0.999f || transmission() > 0.999f) && roughness() <= 1e-2f; }

  DEV f3 diffuse_eval(f3 dir_in, f3 dir_out) const {  // disney_diffuse.hlsli:1-17
    const float hdotwo = fabsf(dot3(normalize3(dir_in + dir_out), dir_out));
    const float FSS90 = roughness() * hdotwo * hdotwo;
    const float FD90 = 0.5f + 2 * FSS90;
    const float ndotwi5 = det_pow5f(1 - fabsf(dir_in.z));
    const float ndotwo5 = det_pow5f(1 - fabsf(dir_out.z));
    const float FDwi = 1 + (FD90 - 1) * ndotwi5;
    const float FDwo = 1 + (FD90 - 1) * ndotwo5;
    const f3 f_base_diffuse = (base_color() / DET_PI) * FDwi * FDwo;
    const float FSSwi = 1 + (FSS90 - 1) * ndotwi5;
    const float FSSwo = 1 + (FSS90 - 1) * ndotwo5;
    const f3 f_subsurface = (1.25f * base_color() / DET_PI) * (FSSwi * FSSwo * (1 / (fabsf(dir_in.z) + fabsf(dir_out.z)) - 0.5f) + 0.5f);
    return lerp3(f_base_diffuse, f_subsurface, subsurface()) * fabsf(dir_out.z);
  }
  DEV void alphas(float& ax, float& ay) const {
    const float aspect = sqrtf(1 - 0.9f * anisotropic());
    ax = fmaxf(0.0001f, alpha() / aspect);
    ay = fmaxf(0.0001f, alpha() * aspect);
  }
  static DEV float glass_reflect_pdf(float F, float D, float G_in, float cos_theta_in) { return (F * D * G_in) / (4 * fabsf(cos_theta_in)); }
  static DEV float glass_refract_pdf(float F, float D, float G_in, float cos_theta_in, float h_dot_in, float h_dot_out, float eta) {
    const float sqrt_denom = h_dot_in + eta * h_dot_out;
    const float dh_dout = eta * eta * h_dot_out / (sqrt_denom * sqrt_denom);
    return (1 - F) * D * G_in * fabsf(dh_dout * h_dot_in / cos_theta_in);
  }
  static DEV f3 glass_eval_reflect(f3 base_color, float F, float D, float G, float cos_theta_in) { return base_color * (F * D * G) / (4 * fabsf(cos_theta_in)); }
  static DEV f3 glass_eval_refract(f3 base_color, float F, float D, float G, float cos_theta_in, float h_dot_in, float h_dot_out, float local_eta, bool adjoint) {
    const float sqrt_denom = h_dot_in + local_eta * h_dot_out;
    const float eta_factor = adjoint ? (1 / (local_eta * local_eta)) : 1;
    const f3 sq = F3(sqrtf(base_color.x), sqrtf(base_color.y), sqrtf(base_color.z));
    return sq * (eta_factor * (1 - F) * D * G * fabsf(h_dot_out * h_dot_in)) / (fabsf(cos_theta_in) * sqrt_denom * sqrt_denom);
  }
  static DEV float metal_eval_pdf(float D, float G_in, float cos_theta_in) { return D * G_in / (4 * fabsf(cos_theta_in)); }
  static DEV f3 metal_eval(f3 base_color, float D, float G, f3 dir_in, float h_dot_out) {
    return base_color * schlick_fresnel3(base_color, fabsf(h_dot_out)) * D * G / (4 * fabsf(dir_in.z));
  }
  static DEV float clearcoat_eval_pdf(float D, f3 h, float hdotwo) { return D * fabsf(h.z) / (4 * fabsf(hdotwo)); }
  static DEV float clearcoat_eval(float D, f3 dir_in, f3 dir_out, float hdotwo) {
    const float Fc = schlick_fresnel1(R0f(1.5f), hdotwo);
    return Fc * D * Gc(dir_in) * Gc(dir_out) / (4 * fabsf(dir_in.z));
  }

  // disney_material.hlsli:141-200
  DEV void eval(MaterialEvalRecord& r, f3 dir_in, f3 dir_out, bool adjoint) const {
    r.f = F3s(0.0f);
    r.pdf_fwd = r.pdf_rev = 0;
    if (emission() > 0) return;
    const float one_minus_metallic = 1 - metallic();
    const float w_diffuse = (1 - transmission()) * one_minus_metallic;
    const float w_metal = metallic();
    const float w_glass = transmission() * one_minus_metallic;
    const float w_clearcoat = 0.25f * clearcoat();
    const bool transmit = dir_in.z * dir_out.z < 0;
    if (w_glass > 0 || w_metal > 0 || w_clearcoat > 0) {
      const float local_eta = dir_in.z < 0 ? 1 / eta() : eta();
      f3 h = normalize3(transmit ? (dir_in + dir_out * local_eta) : (dir_in + dir_out));
      if (h.z * dir_in.z < 0) h = -h;
      const float h_dot_in = dot3(h, dir_in);
      const float h_dot_out = dot3(h, dir_out);
      float ax, ay;
      alphas(ax, ay);
      const float D = Dm(ax, ay, h);
      const float G_in = G1(ax, ay, dir_in);
      const float G_out = G1(ax, ay, dir_out);
      const float F = fresnel_dielectric(h_dot_in, local_eta);
      if (transmit) {
        if (w_glass > 0) {
          r.f = w_glass * glass_eval_refract(base_color(), F, D, G_in * G_out, dir_in.z, h_dot_in, h_dot_out, local_eta, adjoint);
          r.pdf_fwd = w_glass * glass_refract_pdf(F, D, G_in, dir_in.z, h_dot_in, h_dot_out, local_eta);
          r.pdf_rev = w_glass * glass_refract_pdf(fresnel_dielectric(h_dot_out, 1 / local_eta), D, G_out, dir_out.z, h_dot_out, h_dot_in, 1 / local_eta);
        }
      } else {
        if (w_glass > 0) {
          r.f = r.f + w_glass * glass_eval_reflect(base_color(), F, D, G_in * G_out, dir_in.z);
          r.pdf_fwd += w_glass * glass_reflect_pdf(F, D, G_in, dir_in.z);
          r.pdf_rev += w_glass * glass_reflect_pdf(fresnel_dielectric(h_dot_out, local_eta), D, G_out, dir_out.z);
        }
        if (w_metal > 0) {
          r.f = r.f + w_metal * metal_eval(base_color(), D, G_in * G_out, dir_in, dot3(h, dir_out));
          r.pdf_fwd += w_metal * metal_eval_pdf(D, G_in, dir_in.z);
          r.pdf_rev += w_metal * metal_eval_pdf(D, G_out, dir_out.z);
        }
        if (w_clearcoat > 0) {
          const float D_c = Dc((1 - clearcoat_gloss()) * 0.1f + clearcoat_gloss() * 0.001f, h.z);
          r.f = r.f + F3s(w_clearcoat * clearcoat_eval(D_c, dir_in, dir_out, h_dot_out));
          r.pdf_fwd += w_clearcoat * clearcoat_eval_pdf(D_c, h, h_dot_out);
          r.pdf_rev += w_clearcoat * clearcoat_eval_pdf(D_c, h, h_dot_in);
        }
      }
    }
    if (!transmit && w_diffuse > 0) {
      r.pdf_fwd += w_diffuse * cosine_hemisphere_pdfW(fabsf(dir_out.z));
      r.pdf_rev += w_diffuse * cosine_hemisphere_pdfW(fabsf(dir_in.z));
      r.f = r.f + w_diffuse * diffuse_eval(dir_in, dir_out);
    }
  }

  // disney_material.hlsli:201-315
  DEV f3 sample(MaterialSampleRecord& r, f3 rnd, f3 dir_in, f3& beta, bool adjoint) const {
    if (emission() > 0) {
      beta = F3s(0.0f);
      r.pdf_fwd = r.pdf_rev = 0;
      r.eta = 0;
      r.roughness = 0;
      r.dir_out = F3s(0.0f);
      return F3s(0.0f);
    }
    const float one_minus_metallic = 1 - metallic();
    const float w_diffuse = (1 - transmission()) * one_minus_metallic;
    const float w_metal = metallic();
    const float w_glass = transmission() * one_minus_metallic;
    const float w_clearcoat = 0.25f * clearcoat();
    const bool lobes = w_glass > 0 || w_metal > 0 || w_clearcoat > 0;  // D, F, G only feed these lobes
    float ax, ay;
    alphas(ax, ay);
    const float alpha_c = (1 - clearcoat_gloss()) * 0.1f + clearcoat_gloss() * 0.001f;
    const float local_eta = dir_in.z < 0 ? 1 / eta() : eta();
    const float G_in = lobes ? G1(ax, ay, dir_in) : 0.0f;
    f3 h = F3s(0.0f);
    float h_dot_in = 0, D = 0, F = 0;
    r.eta = 0;
    r.roughness = roughness();
    if (rnd.z < w_glass + w_metal) {
      h = sample_visible_normals(dir_in, ax, ay, rnd.x, rnd.y);
      h_dot_in = dot3(h, dir_in);
      D = Dm(ax, ay, h);
      F = fresnel_dielectric(h_dot_in, local_eta);
      if (rnd.z < w_glass) {
        const float h_dot_out_sq = 1 - (1 - h_dot_in * h_dot_in) / (local_eta * local_eta);
        if (h_dot_out_sq <= 0 || rnd.z / w_glass <= F) {
          r.dir_out = reflect3(-dir_in, h);
        } else {
          r.dir_out = refract3(-dir_in, h, 1 / local_eta);
          r.eta = local_eta;
          const float G_out = G1(ax, ay, r.dir_out);
          const float h_dot_out = dot3(h, r.dir_out);
          r.pdf_fwd = w_glass * glass_refract_pdf(F, D, G_in, dir_in.z, h_dot_in, h_dot_out, local_eta);
          r.pdf_rev = w_glass * glass_refract_pdf(fresnel_dielectric(h_dot_out, 1 / local_eta), D, G_out, r.dir_out.z, h_dot_out, h_dot_in, 1 / local_eta);
          const f3 f = w_glass * glass_eval_refract(base_color(), F, D, G_in * G_out, dir_in.z, h_dot_in, h_dot_out, local_eta, adjoint);
          beta = beta * F3(f.x / r.pdf_fwd, f.y / r.pdf_fwd, f.z / r.pdf_fwd);
          return f;
        }
      } else {
        r.dir_out = reflect3(-dir_in, h);
      }
    } else {
      if (rnd.z < w_glass + w_metal + w_clearcoat) {
        const float alpha2 = alpha_c * alpha_c;
        const float cos_phi = sqrtf((1 - det_powf(alpha2, 1 - rnd.x)) / (1 - alpha2));
        const float sin_phi = sqrtf(1 - fmaxf(cos_phi * cos_phi, 0.0f));
        const float theta = DET_2PI * rnd.y;
        float st, ct;
        det_sincosf(theta, &st, &ct);
        h = F3(sin_phi * ct, sin_phi * st, cos_phi);
        if (dir_in.z < 0) h = -h;
        r.dir_out = reflect3(-dir_in, h);
        r.roughness = alpha_c;
      } else {
        r.dir_out = sample_cos_hemisphere(rnd.x, rnd.y);
        if (dir_in.z < 0) r.dir_out = -r.dir_out;
        r.roughness = 1;
        if (lobes) h = normalize3(dir_in + r.dir_out);
      }
      if (lobes) {
        h_dot_in = dot3(h, dir_in);
        D = Dm(ax, ay, h);
        F = fresnel_dielectric(h_dot_in, local_eta);
      }
    }
    r.pdf_fwd = 0;
    r.pdf_rev = 0;
    f3 f = F3s(0.0f);
    if (lobes) {
      const float G_out = G1(ax, ay, r.dir_out);
      const float h_dot_out = dot3(h, r.dir_out);
      if (w_glass > 0) {
        r.pdf_fwd += w_glass * glass_reflect_pdf(F, D, G_in, dir_in.z);
        r.pdf_rev += w_glass * glass_reflect_pdf(fresnel_dielectric(h_dot_out, local_eta), D, G_out, r.dir_out.z);
        f = f + w_glass * glass_eval_reflect(base_color(), F, D, G_in * G_out, dir_in.z);
      }
      if (w_metal > 0) {
        r.pdf_fwd += w_metal * metal_eval_pdf(D, G_in, dir_in.z);
        r.pdf_rev += w_metal * metal_eval_pdf(D, G_out, r.dir_out.z);
        f = f + w_metal * metal_eval(base_color(), D, G_in * G_out, dir_in, h_dot_out);
      }
      if (w_clearcoat > 0) {
        const float D_c = Dc(alpha_c, h.z);
        r.pdf_fwd += w_clearcoat * clearcoat_eval_pdf(D_c, h, h_dot_out);
        r.pdf_rev += w_clearcoat * clearcoat_eval_pdf(D_c, h, h_dot_in);
        f = f + F3s(w_clearcoat * clearcoat_eval(D_c, dir_in, r.dir_out, h_dot_out));
      }
    }
    if (w_diffuse > 0) {
      r.pdf_fwd += w_diffuse * cosine_hemisphere_pdfW(fabsf(r.dir_out.z));
      r.pdf_rev += w_diffuse * cosine_hemisphere_pdfW(fabsf(dir_in.z));
      f = f + w_diffuse * diffuse_eval(dir_in, r.dir_out);
    }
    beta = beta * F3(f.x / r.pdf_fwd, f.y / r.pdf_fwd, f.z / r.pdf_fwd);
    return f;
  }
};

// dist2.h:6-20 (upper_bound), :29-57 (dist2d_pdf / dist2d_sample) over gDistributions
DEV uint32_t dist_upper_bound(const float* data, uint32_t first, uint32_t last, float value) {
  int count = (int)(last - first);
  while (count > 0) {
    uint32_t it = first;
    const int step = count / 2;
    it += (uint32_t)step;
    if (value >= data[it]) {
      first = ++it;
      count -= step + 1;
    } else
      count = step;
  }
  return first;
}
DEV int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
DEV float dist2d_pdf(const float* data, uint32_t pdf_marginals, uint32_t pdf_rows, uint32_t w, uint32_t h, float u, float v) {
  const int x = (int)fminf(fmaxf(u * (float)w, 0.0f), (float)(w - 1));
  const int y = (int)fminf(fmaxf(v * (float)h, 0.0f), (float)(h - 1));
  const float pdf_y = data[pdf_marginals + y];
  const float pdf_x = data[pdf_rows + y * w + x];
  return pdf_y * pdf_x * (float)w * (float)h;
}
DEV void dist2d_sample(const float* data, uint32_t cdf_marginals, uint32_t cdf_rows, uint32_t w, uint32_t h, float rx, float ry, float& u, float& v) {
  const uint32_t y_ptr = dist_upper_bound(data, cdf_marginals, cdf_marginals + h + 1, ry) - cdf_marginals;
  const int y_offset = clampi((int)y_ptr - 1, 0, (int)h - 1);
  float dy = ry - data[cdf_marginals + y_offset];
  if ((data[cdf_marginals + y_offset + 1] - data[cdf_marginals + y_offset]) > 0) dy /= (data[cdf_marginals + y_offset + 1] - data[cdf_marginals + y_offset]);
  const int row_offset = y_offset * ((int)w + 1);
  const uint32_t x_ptr = dist_upper_bound(data, cdf_rows + row_offset, cdf_rows + row_offset + w + 1, rx) - cdf_rows;
  const int x_offset = clampi((int)x_ptr - row_offset - 1, 0, (int)w - 1);
  float dx = rx - data[cdf_rows + row_offset + x_offset];
  if (data[cdf_rows + row_offset + x_offset + 1] - data[cdf_rows + row_offset + x_offset] > 0) dx /= (data[cdf_rows + row_offset + x_offset + 1] - data[cdf_rows + row_offset + x_offset]);
  u = ((float)x_offset + dx) / (float)w;
  v = ((float)y_offset + dy) / (float)h;
}

// environment.h:8-95. Record in gMaterialData: ImageValue3 (float3 value, uint image_index) and, when an image is
// bound, the offsets of marginal_pdf, row_pdf, marginal_cdf, row_cdf in gDistributions (:17-22,37-45).
struct Environment {
  f3 value;
  uint32_t image_index;
  uint32_t marginal_pdf, row_pdf, marginal_cdf, row_cdf;
  DEV bool has_image(const DeviceScene& sc) const { return image_index < sc.image_count; }
  DEV void load(const DeviceScene& sc, uint32_t address) {
    const uint32_t* a = reinterpret_cast<const uint32_t*>(sc.materials + address);  // 4-byte aligned (72-byte records before it)
    value = F3(__uint_as_float(a[0]), __uint_as_float(a[1]), __uint_as_float(a[2]));
    image_index = a[3];
    marginal_pdf = row_pdf = marginal_cdf = row_cdf = 0;
    if (has_image(sc)) {
      marginal_pdf = a[4];
      row_pdf = a[5];
      marginal_cdf = a[6];
      row_cdf = a[7];
    }
  }
  DEV f3 lookup(const DeviceScene& sc, float u, float v) const { return xyz(DisneyMaterial::sample_image(sc, image_index, u, v, 0.0f, false)); }
  DEV f3 eval(const DeviceScene& sc, f3 dir_out) const {
    if (!has_image(sc)) return value;
    float u, v;
    cartesian_to_spherical_uv(dir_out, u, v);
    return lookup(sc, u, v) * value;
  }
  // sample_texel / sample_texel_pdf, bdpt_util.hlsli:85-180 (eSampleEnvironmentMapDirectly): a descent through the mip
  // chain from the 2 x 1 level towards level 1 (at most 10 levels), at each level one of the 2 x 2 children in
  // proportion to luminance x sin(theta). Texel loads outside a level return zero, as Texture2D::Load does.
  static DEV float texel_weight(const DeviceScene& sc, const DeviceImage& im, uint32_t level, uint32_t x, uint32_t y, float inv_h) {
    float sn, cs;
    det_sincosf(DET_PI * ((float)y + 0.5f) * inv_h, &sn, &cs);
    if (x >= im.w[level] || y >= im.h[level]) return 0.0f * sn;
    const float4 t = sc.image_texels[im.offset[level] + (size_t)y * im.w[level] + x];
    return luminance3(F3(t.x, t.y, t.z)) * sn;
  }
  static DEV bool texel_level(const DeviceScene& sc, const DeviceImage& im, uint32_t level, uint32_t cx, uint32_t cy, float p[4]) {
    const float inv_h = 1 / (float)im.h[level];
    p[0] = p[1] = p[2] = p[3] = 0;
    if (im.w[level] > 1) {
      p[0] = texel_weight(sc, im, level, cx, cy, inv_h);
      p[1] = texel_weight(sc, im, level, cx + 1, cy, inv_h);
    }
    if (im.h[level] > 1) {
      p[2] = texel_weight(sc, im, level, cx, cy + 1, inv_h);
      p[3] = texel_weight(sc, im, level, cx + 1, cy + 1, inv_h);
    }
    const float sum = ((p[0] + p[1]) + p[2]) + p[3];
    if (sum < 1e-6f) return false;
    for (int j = 0; j < 4; j++) p[j] /= sum;
    return true;
  }
  DEV void sample_texel(const DeviceScene& sc, float rx, float ry, float& pdf, float& u, float& v) const {
    const DeviceImage& im = sc.images[image_index];
    const uint32_t level_count = im.levels;
    pdf = 1;
    uint32_t cx = 0, cy = 0, lw = 1, lh = 1;
    for (uint32_t i = 1; i < min(10u + 1u, level_count - 1); i++) {
      const uint32_t level = level_count - 1 - i;
      const uint32_t w = im.w[level], h = im.h[level];
      cx *= w / lw;
      cy *= h / lh;
      float p[4];
      if (!texel_level(sc, im, level, cx, cy, p)) continue;
      for (int j = 0; j < 4; j++) {
        if (rx < p[j]) {
          cx += (uint32_t)(j & 1);
          cy += (uint32_t)(j >> 1);
          pdf *= p[j];
          rx /= p[j];
          break;
        }
        rx -= p[j];
      }
      lw = w;
      lh = h;
    }
    pdf *= (float)(lw * lh);
    u = ((float)cx + rx) / (float)lw;
    v = ((float)cy + ry) / (float)lh;
  }
  DEV float sample_texel_pdf(const DeviceScene& sc, float u, float v) const {
    const DeviceImage& im = sc.images[image_index];
    const uint32_t level_count = im.levels;
    float pdf = 1;
    uint32_t lw = 1, lh = 1;
    for (uint32_t i = 1; i < min(10u + 1u, level_count - 1); i++) {
      const uint32_t level = level_count - 1 - i;
      const uint32_t w = im.w[level], h = im.h[level];
      const uint32_t cx = (uint32_t)(floorf((float)w * u / 2) * 2), cy = (uint32_t)(floorf((float)h * v / 2) * 2);
      float p[4];
      if (!texel_level(sc, im, level, cx, cy, p)) continue;
      const uint32_t dx = (uint32_t)(u * (float)w) - cx, dy = (uint32_t)(v * (float)h) - cy;  // saturate() of an unsigned: 0 .. 1
      const uint32_t ox = dx > 1 ? 1 : dx, oy = dy > 1 ? 1 : dy;
      pdf *= p[oy * 2 + ox];
      lw = w;
      lh = h;
    }
    return pdf * (float)(lw * lh);
  }
  DEV f3 sample(const DeviceScene& sc, float rx, float ry, f3& dir_out, float& pdf, bool direct = false) const {
    if (direct && has_image(sc)) {  // environment.h:66-67
      float u, v;
      sample_texel(sc, rx, ry, pdf, u, v);
      dir_out = spherical_uv_to_cartesian(u, v);
      pdf /= (DET_2PI2 * sqrtf(1 - dir_out.y * dir_out.y));
      return value * lookup(sc, u, v);
    }
    if (!has_image(sc)) {
      // sample_uniform_sphere's (phi, theta) go through spherical_uv_to_cartesian as if they were uv (as upstream)
      dir_out = spherical_uv_to_cartesian(2 * DET_PI * ry, det_acosf(2 * rx - 1));
      pdf = DET_INV_4PI;
      return value;
    }
    const uint32_t w = sc.images[image_index].w[0], h = sc.images[image_index].h[0];
    float u, v;
    dist2d_sample(sc.distributions, marginal_cdf, row_cdf, w, h, rx, ry, u, v);
    pdf = dist2d_pdf(sc.distributions, marginal_pdf, row_pdf, w, h, u, v);
    dir_out = spherical_uv_to_cartesian(u, v);
    pdf /= (DET_2PI2 * sqrtf(1 - dir_out.y * dir_out.y));
    return value * lookup(sc, u, v);
  }
  DEV float eval_pdf(const DeviceScene& sc, f3 dir_out, bool direct = false) const {
    if (!has_image(sc)) return DET_INV_4PI;
    float u, v;
    cartesian_to_spherical_uv(dir_out, u, v);
    if (direct) return sample_texel_pdf(sc, u, v) / (DET_2PI2 * sqrtf(1 - dir_out.y * dir_out.y));  // environment.h:84-85
    const uint32_t w = sc.images[image_index].w[0], h = sc.images[image_index].h[0];
    const float pdf = dist2d_pdf(sc.distributions, marginal_pdf, row_pdf, w, h, u, v);
    return pdf / (DET_2PI2 * sqrtf(1 - dir_out.y * dir_out.y));
  }
};

// path.hlsli:8-15
DEV float mis2(bool use_mis, float a, float b) {
  if (!use_mis) return 0.5f;
  const float a2 = a * a;
  return a2 / (a2 + b * b);
}
// path.hlsli:67-98 with gShadingNormalFix off, adjoint = false (view paths)
DEV float shading_normal_correction(float ndotin, float ndotout, float ngdotin, float ngdotout, float ngdotns = 1.0f, bool terminator_fix = false, bool adjoint = false) {
  if (sgnf(ngdotout * ngdotin) != sgnf(ndotin * ndotout)) return 0;
  float G = 1;
  if (terminator_fix) {  // eShadingNormalShadowFix, path.hlsli:84-86
    G = fminf(1.0f, fabsf(adjoint ? ngdotin / (ndotin * ngdotns) : ngdotout / (ndotout * ngdotns)));
    G = -(pow2f(G) * G) + pow2f(G) + G;
  }
  if (adjoint) {  // light paths: the non-symmetry of shading normals, path.hlsli:90-95
    const float num = ngdotout * ndotin;
    const float denom = ndotout * ngdotin;
    if (fabsf(denom) > 1e-5f) G *= fabsf(num / denom);
  }
  return G;
}
// path.hlsli:29-36: dE (or dL) of a vertex from the previous vertex's
DEV float connection_dVC(float dVC, float pdfA_rev, float prev_pdfA_fwd, bool specular) { return ((specular ? 0.0f : 1.0f) + dVC * pow2f(pdfA_rev)) / pow2f(prev_pdfA_fwd); }
