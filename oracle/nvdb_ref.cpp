// nvdb_ref.cpp — reference build (oracle/_ref, build container only): a small NanoVDB fog volume made with the
// NanoVDB 32.3.3 the reference vendors (src/extern/nanovdb), and known answers read from it through the very PNanoVDB
// functions the reference's shaders call (medium.hlsli:58-71,85-88; intersection.hlsli:93-113). The grid bytes and the
// answers become the fixture tests/golden/fog_sphere.npz (tests/golden/make_nvdb_golden.py); nothing of NanoVDB itself is
// copied into this repository.
//   usage: nvdb_ref <radius_voxels> <voxel_size> <half_width> <grid.nvdb> <probe.bin>
#include <nanovdb/util/Primitives.h>
#define PNANOVDB_C
#include <nanovdb/PNanoVDB.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  const float radius = (float)atof(argv[1]);
  const double voxel = atof(argv[2]);
  const double half_width = atof(argv[3]);
  auto handle = nanovdb::createFogVolumeSphere<float>(radius * (float)voxel, nanovdb::Vec3f(0.1f, -0.05f, 0.2f), voxel, half_width, nanovdb::Vec3d(0), "density");
  FILE* f = fopen(argv[4], "wb");
  fwrite(handle.data(), 1, handle.size(), f);
  fclose(f);

  pnanovdb_buf_t buf = pnanovdb_make_buf((pnanovdb_uint32_t*)handle.data(), handle.size() / 4);
  pnanovdb_grid_handle_t grid = {{0}};
  pnanovdb_tree_handle_t tree = pnanovdb_grid_get_tree(buf, grid);
  pnanovdb_root_handle_t root = pnanovdb_tree_get_root(buf, tree);
  pnanovdb_readaccessor_t acc;
  pnanovdb_readaccessor_init(&acc, root);
  std::vector<int32_t> out_i;
  std::vector<float> out_f;
  // header answers: bbox, root maximum, grid type
  const pnanovdb_coord_t bmin = pnanovdb_root_get_bbox_min(buf, root), bmax = pnanovdb_root_get_bbox_max(buf, root);
  out_i.insert(out_i.end(), {bmin.x, bmin.y, bmin.z, bmax.x, bmax.y, bmax.z, (int32_t)pnanovdb_grid_get_grid_type(buf, grid), 0});
  out_f.push_back(pnanovdb_read_float(buf, pnanovdb_root_get_max_address(PNANOVDB_GRID_TYPE_FLOAT, buf, root)));
  // value lookups at pseudo-random index coordinates around the sphere (and far outside: background)
  uint32_t s = 12345u;
  auto next = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
  const int N = 6000;
  const int span = (int)radius + 8;
  for (int k = 0; k < N; k++) {
    pnanovdb_coord_t c;
    const int wide = (k % 16 == 0) ? 5000 : span;  // some far away: other root keys
    c.x = (int)(next() % (2 * wide + 1)) - wide;
    c.y = (int)(next() % (2 * wide + 1)) - wide;
    c.z = (int)(next() % (2 * wide + 1)) - wide;
    const pnanovdb_address_t a = pnanovdb_readaccessor_get_value_address(PNANOVDB_GRID_TYPE_FLOAT, buf, &acc, &c);
    out_i.insert(out_i.end(), {c.x, c.y, c.z});
    out_f.push_back(pnanovdb_read_float(buf, a));
  }
  // the four map functions at a few points
  const int M = 16;
  for (int k = 0; k < M; k++) {
    pnanovdb_vec3_t p = {(float)(next() % 2001) * 0.001f - 1.0f, (float)(next() % 2001) * 0.001f - 1.0f, (float)(next() % 2001) * 0.001f - 1.0f};
    const pnanovdb_vec3_t a = pnanovdb_grid_world_to_indexf(buf, grid, &p), b = pnanovdb_grid_world_to_index_dirf(buf, grid, &p);
    const pnanovdb_vec3_t c = pnanovdb_grid_index_to_worldf(buf, grid, &a), d = pnanovdb_grid_index_to_world_dirf(buf, grid, &b);
    out_f.insert(out_f.end(), {p.x, p.y, p.z, a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, d.x, d.y, d.z});
  }
  f = fopen(argv[5], "wb");
  const int32_t counts[4] = {(int32_t)out_i.size(), (int32_t)out_f.size(), N, M};
  fwrite(counts, 4, 4, f);
  fwrite(out_i.data(), 4, out_i.size(), f);
  fwrite(out_f.data(), 4, out_f.size(), f);
  fclose(f);
  printf("%zu bytes, %d probes\n", handle.size(), N);
  return 0;
}
