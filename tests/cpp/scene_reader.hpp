// scene_reader.hpp — reads the scene description stratum_amd.scene.dump_description writes and builds the node graph
// from it the way a loader would (Scene.hpp:30-37: one node per primitive carrying a TransformData and a MeshPrimitive /
// SpherePrimitive / Medium). Shared by host_test.cpp and multi_host.cpp.
#pragma once
#include <fstream>
#include <string>
#include <vector>

#include "../../stratum_amd/host/stratum_hip.hpp"

struct Reader {
  std::ifstream f;
  explicit Reader(const char* p) : f(p, std::ios::binary) {
    if (!f) throw std::runtime_error(std::string("cannot open ") + p);
  }
  template <typename T>
  T get() {
    T v;
    f.read((char*)&v, sizeof(T));
    return v;
  }
  template <typename T>
  std::vector<T> vec(size_t n) {
    std::vector<T> v(n);
    f.read((char*)v.data(), n * sizeof(T));
    return v;
  }
};


struct LoadedScene {
  stm::Node* scene_node = nullptr;
  stm::component_ptr<stm::Scene> scene;
  stm::ViewData view;
  stm::TransformData view_xf;
  uint32_t W = 0, H = 0;
};

// builds Scene (+ images, materials, meshes, primitives, environment, media) under `app_node`; leaves `r` positioned
// behind the camera block (the packed arrays the `pack` mode compares follow)
inline LoadedScene load_scene(Reader& r, stm::Node& app_node) {
  using namespace stm;
  LoadedScene L;
  Node& scene_node = app_node.make_child("Scene");
  auto scene = scene_node.make_component<Scene>();

  // images
  const uint32_t n_img = r.get<uint32_t>();
  std::vector<component_ptr<Image>> images;
  for (uint32_t i = 0; i < n_img; i++) {
    const uint32_t w = r.get<uint32_t>(), h = r.get<uint32_t>();
    auto im = scene_node.make_child("image").make_component<Image>();
    im->width = w;
    im->height = h;
    im->pixels = r.vec<float>((size_t)w * h * 4);
    images.push_back(im);
  }
  auto image_of = [&](uint32_t index) { return index < images.size() ? images[index] : component_ptr<Image>(); };
  const uint32_t n_img1 = r.get<uint32_t>();
  std::vector<component_ptr<Image1>> images1;
  for (uint32_t i = 0; i < n_img1; i++) {
    const uint32_t w = r.get<uint32_t>(), h = r.get<uint32_t>();
    auto im = scene_node.make_child("image1").make_component<Image1>();
    im->width = w;
    im->height = h;
    im->pixels = r.vec<float>((size_t)w * h);
    images1.push_back(im);
  }
  // materials
  const uint32_t n_mat = r.get<uint32_t>();
  std::vector<component_ptr<Material>> materials;
  for (uint32_t i = 0; i < n_mat; i++) {
    const sthip_MaterialRecord rec = r.get<sthip_MaterialRecord>();
    auto m = scene_node.make_child("material").make_component<Material>();
    for (int k = 0; k < 3; k++) {
      std::memcpy(m->values[k].value, rec.values[k].value, 16);
      m->values[k].image = image_of(rec.values[k].image_index);
    }
    if (rec.alpha_mask_index < images1.size()) m->alpha_mask = images1[rec.alpha_mask_index];
    m->bump_image = image_of(rec.bump_index);
    m->bump_strength = rec.bump_strength;
    materials.push_back(m);
  }
  // meshes
  const uint32_t n_mesh = r.get<uint32_t>();
  std::vector<component_ptr<Mesh>> meshes;
  for (uint32_t i = 0; i < n_mesh; i++) {
    const uint32_t nv = r.get<uint32_t>(), nt = r.get<uint32_t>(), stride = r.get<uint32_t>();
    auto mesh = scene_node.make_child("mesh").make_component<Mesh>();
    mesh->positions = r.vec<stm::float3>(nv);
    mesh->normals = r.vec<stm::float3>(nv);
    mesh->uvs = r.vec<stm::float2>(nv);
    mesh->indices = r.vec<uint32_t>((size_t)nt * 3);
    mesh->index_stride = stride;
    meshes.push_back(mesh);
  }
  // instances: one node each, carrying a TransformData and a MeshPrimitive (Scene.hpp:30-33)
  const uint32_t n_inst = r.get<uint32_t>();
  for (uint32_t i = 0; i < n_inst; i++) {
    const uint32_t mesh = r.get<uint32_t>(), mat = r.get<uint32_t>();
    const TransformData t = r.get<TransformData>();
    Node& n = scene_node.make_child("prim" + std::to_string(i));
    n.make_component<TransformData>(t);
    n.make_component<MeshPrimitive>(MeshPrimitive{materials.at(mat), meshes.at(mesh)});
  }
  // sphere primitives (Scene.hpp:34-37) and the environment component (environment.h)
  const uint32_t n_sph = r.get<uint32_t>();
  for (uint32_t i = 0; i < n_sph; i++) {
    const uint32_t mat = r.get<uint32_t>();
    const float radius = r.get<float>();
    const TransformData t = r.get<TransformData>();
    Node& n = scene_node.make_child("sphere" + std::to_string(i));
    n.make_component<TransformData>(t);
    n.make_component<SpherePrimitive>(SpherePrimitive{materials.at(mat), radius});
  }
  const uint32_t env_kind = r.get<uint32_t>();
  if (env_kind) {
    float value[3];
    for (float& v : value) v = r.get<float>();
    const uint32_t image = r.get<uint32_t>();
    scene_node.make_child("environment").make_component<Environment>(make_environment(env_kind == 2 ? images.at(image) : component_ptr<Image>(), value[0], value[1], value[2]));
  }
  // media (Material.hpp:72-87): NanoVDB buffers and one Medium component per volume instance
  std::vector<std::shared_ptr<std::vector<uint8_t>>> volumes;
  const uint32_t n_vol = r.get<uint32_t>();
  for (uint32_t i = 0; i < n_vol; i++) {
    const uint64_t bytes = r.get<uint64_t>();
    volumes.push_back(std::make_shared<std::vector<uint8_t>>(r.vec<uint8_t>((size_t)bytes)));
  }
  const uint32_t n_med = r.get<uint32_t>();
  for (uint32_t i = 0; i < n_med; i++) {
    Medium med;
    for (float& v : med.density_scale) v = r.get<float>();
    med.anisotropy = r.get<float>();
    for (float& v : med.albedo_scale) v = r.get<float>();
    med.attenuation_unit = r.get<float>();
    const uint32_t dv = r.get<uint32_t>(), av = r.get<uint32_t>();
    med.density_buffer = volumes.at(dv);
    if (av != 0xFFFFFFFFu) med.albedo_buffer = volumes.at(av);
    const TransformData t = r.get<TransformData>();
    Node& n = scene_node.make_child("medium" + std::to_string(i));
    n.make_component<TransformData>(t);
    n.make_component<Medium>(med);
  }
  const ViewData view = r.get<ViewData>();
  const TransformData view_xf = r.get<TransformData>();
  const uint32_t W = r.get<uint32_t>(), H = r.get<uint32_t>();

  L.scene_node = &scene_node;
  L.scene = scene;
  L.view = view;
  L.view_xf = view_xf;
  L.W = W;
  L.H = H;
  return L;
}
