// medium.h — the Medium of materials/medium.hlsli on the device: Henyey-Greenstein phase function and delta tracking
// against the density grid's root maximum (NanoVDB reader: media.h).
#pragma once

#include "shading.h"
#include "traverse.h"

struct Medium {
  f3 density_scale, albedo_scale;
  float anisotropy, attenuation_unit;
  uint32_t density_volume_index, albedo_volume_index;
  DEV void load(const DeviceScene& sc, uint32_t address) {  // medium.hlsli:12-19, Material.hpp:80-87 (4-byte aligned)
    const uint32_t* a = reinterpret_cast<const uint32_t*>(sc.materials + address);
    density_scale = F3(__uint_as_float(a[0]), __uint_as_float(a[1]), __uint_as_float(a[2]));
    anisotropy = __uint_as_float(a[3]);
    albedo_scale = F3(__uint_as_float(a[4]), __uint_as_float(a[5]), __uint_as_float(a[6]));
    attenuation_unit = __uint_as_float(a[7]);
    density_volume_index = a[8];
    albedo_volume_index = a[9];
  }
  DEV bool can_eval() const { return any_gt0(density_scale); }
  DEV bool is_specular() const { return fabsf(anisotropy) > 0.999f; }
  DEV float phase(f3 dir_in, f3 dir_out) const {  // medium.hlsli:26-34
    return DET_INV_4PI * (1 - anisotropy * anisotropy) / det_powf(1 + anisotropy * anisotropy + 2 * anisotropy * dot3(dir_in, dir_out), 1.5f);
  }
  // medium.hlsli:35-56; dir_out in the space dir_in is given in (world); pdf_fwd = pdf_rev = f
  DEV f3 sample(float r0, float r1, f3 dir_in, float& pdf, float& roughness) const {
    f3 dir_out;
    if (fabsf(anisotropy) < 1e-3f) {
      const float z = 1 - 2 * r0;
      float sn, cs;
      det_sincosf(DET_2PI * r1, &sn, &cs);
      const float rr = sqrtf(fmaxf(0.0f, 1 - z * z));
      dir_out = F3(rr * cs, rr * sn, z);
    } else {
      const float tmp = (anisotropy * anisotropy - 1) / (2 * r0 * anisotropy - (anisotropy + 1));
      const float cos_elevation = (tmp * tmp - (1 + anisotropy * anisotropy)) / (2 * anisotropy);
      const float sin_elevation = sqrtf(fmaxf(1 - cos_elevation * cos_elevation, 0.0f));
      float sn, cs;
      det_sincosf(DET_2PI * r1, &sn, &cs);
      f3 t, b;
      make_orthonormal(dir_in, t, b);
      dir_out = t * (sin_elevation * cs) + b * (sin_elevation * sn) + dir_in * cos_elevation;
    }
    pdf = phase(dir_in, dir_out);
    roughness = 1 - fabsf(anisotropy);
    return dir_out;
  }
  static DEV float grid_value(const DeviceScene& sc, uint32_t volume, f3 pos_index) {
    NvdbView g;
    g.v = sc.volumes[volume];
    g.w = sc.volume_words + g.v.first_word;
    return g.value((int32_t)floorf(pos_index.x), (int32_t)floorf(pos_index.y), (int32_t)floorf(pos_index.z));
  }
  // delta_track, medium.hlsli:74-127. origin / direction in the object space of the volume instance (= the grid's world
  // space). Returns true with the scatter position (grid world space, as upstream returns it) on a real collision; a
  // null collision ends the walk (upstream returns after the first one).
  DEV bool delta_track(const DeviceScene& sc, Rng& rng, f3 origin, f3 direction, float t_max, f3& beta, f3& dir_pdf, f3& nee_pdf, bool can_scatter, uint32_t max_null_collisions,
                       f3& scatter_p) const {
    const DeviceVolume v = sc.volumes[density_volume_index];
    const f3 majorant = density_scale * v.root_max;
    const uint32_t channel = rng.next_uint() % 3u;
    const float maj_c = channel == 0 ? majorant.x : (channel == 1 ? majorant.y : majorant.z);
    if (maj_c < 1e-6f) return false;
    origin = nvdb_world_to_index(v, origin);
    direction = nvdb_world_to_index_dir(v, direction);
    for (uint32_t iteration = 0; iteration < max_null_collisions && any_gt0(beta); iteration++) {
      const float r0 = rng.next_float(), r1 = rng.next_float();
      const float t = attenuation_unit * -det_logf(1 - r0) / maj_c;
      if (t < t_max) {
        origin = origin + direction * t;
        t_max -= t;
        const f3 local_density = density_scale * grid_value(sc, density_volume_index, origin);
        const f3 local_albedo = albedo_scale * (albedo_volume_index == 0xFFFFFFFFu ? 1.0f : grid_value(sc, albedo_volume_index, origin));
        const f3 local_sigma_s = local_density * local_albedo;
        const f3 local_sigma_a = local_density * (F3s(1.0f) - local_albedo);
        const f3 local_sigma_t = local_sigma_s + local_sigma_a;
        const f3 real_prob = F3(local_sigma_t.x / majorant.x, local_sigma_t.y / majorant.y, local_sigma_t.z / majorant.z);
        const float max_maj = fmaxf(fmaxf(majorant.x, majorant.y), majorant.z);
        const f3 tr = F3(det_expf(-majorant.x * t), det_expf(-majorant.y * t), det_expf(-majorant.z * t)) / max_maj;
        const float rp_c = channel == 0 ? real_prob.x : (channel == 1 ? real_prob.y : real_prob.z);
        if (can_scatter && r1 < rp_c) {  // real particle
          beta = beta * (tr * local_sigma_s);
          dir_pdf = dir_pdf * (tr * majorant * real_prob);
          scatter_p = nvdb_index_to_world(v, origin);
          return true;
        } else {  // fake particle
          beta = beta * (tr * (majorant - local_sigma_t));
          dir_pdf = dir_pdf * (tr * majorant * (F3s(1.0f) - real_prob));
          nee_pdf = nee_pdf * (tr * majorant);
          return false;
        }
      } else {  // transmitted without scattering
        const f3 tr = F3(det_expf(-majorant.x * t_max), det_expf(-majorant.y * t_max), det_expf(-majorant.z * t_max));
        beta = beta * tr;
        nee_pdf = nee_pdf * tr;
        dir_pdf = dir_pdf * tr;
        break;
      }
    }
    return false;
  }
};
DEV float average3(f3 x) { return (x.x + x.y + x.z) / 3; }

// The boundary a ray crosses when the traversal reports a volume instance (intersection.hlsli:93-113,160-165 and
// make_volume_shading_data, shading_data.hlsli:106-110): world position and the octahedral-packed face normal.
DEV void volume_boundary(const DeviceScene& sc, uint32_t inst_index, const Inst& in, f3 seg_origin, f3 direction, float t, f3& position, uint32_t& packed_normal) {
  const Xf inv = load_xf(sc.inv_xf, inst_index), xf = load_xf(sc.xf, inst_index);
  const float im[12] = {inv.r0.x, inv.r0.y, inv.r0.z, inv.r0.w, inv.r1.x, inv.r1.y, inv.r1.z, inv.r1.w, inv.r2.x, inv.r2.y, inv.r2.z, inv.r2.w};
  const f3 oo = obj_point(im, seg_origin), od = obj_vector(im, direction);
  const DeviceVolume v = sc.volumes[in.p.z];
  float tt;
  f3 face = F3s(0.0f);
  volume_test(v, oo, od, 0.0f, __builtin_inff(), tt, &face);  // the committed candidate again, for its face
  packed_normal = pack_normal_octahedron(normalize3(xf_vector(xf, nvdb_index_to_world_dir(v, face))));
  position = xf_point(xf, oo + od * t);
}
