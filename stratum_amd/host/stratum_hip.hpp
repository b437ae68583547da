// stratum_hip.hpp — C++ host side above the C ABI (include/sthip.h): the slice of Stratum's host
// interface that feeds and drives the path-tracing hot path, with the reference's own names, argument
// meaning and error behaviour (exceptions), so that host code written against Stratum keeps compiling:
//
//   stm::NodeGraph / Node / component_ptr<T> / Node::Event     src/Node/NodeGraph.hpp:13-360
//   stm::TransformData helpers, make_perspective, quatf        src/Shaders/transform.h, quatf.h
//   stm::Material, Mesh, MeshPrimitive, Camera                 src/Node/Material.hpp, Scene.hpp:15-37
//   stm::node_to_world                                         src/Node/Scene.cpp:108-117
//   stm::Application (OnUpdate / OnRenderWindow only)          src/Node/Application.hpp:11-29
//   stm::Scene  (update() packs SceneData, data())             src/Node/Scene.hpp:44-69, Scene.cpp:299-684
//   stm::BDPT   (update(), render(), prev_result())            src/Node/BDPT.hpp:13-24, BDPT.cpp:35-127,341-838
//
// What is NOT here: Vulkan (Device, CommandBuffer, Image, Buffer), loaders, GUI, window. The
// CommandBuffer& parameters of the reference's signatures become an opaque stm::CommandBuffer that carries
// the HIP stream. The reference needs Eigen for its vector types; this header uses plain arrays (Eigen is
// not in the image), so float3 etc. are minimal structs.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <queue>
#include <stdexcept>
#include <string>
#include <tuple>
#include <typeindex>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include <dlfcn.h>

#include "../../include/sthip.h"

namespace stm {

// ------------------------------------------------------------------------------------------------
// NodeGraph.hpp:13-360
// ------------------------------------------------------------------------------------------------
class Node;
class NodeGraph;

template <typename T>
class component_ptr {
 public:
  component_ptr() = default;
  component_ptr(std::nullptr_t) {}
  component_ptr(Node* n, T* c) : mNode(n), mComponent(c) {}
  component_ptr(const Node* n, T* c) : mNode(const_cast<Node*>(n)), mComponent(c) {}
  Node& node() const { return *mNode; }
  operator bool() const { return mComponent != nullptr; }
  T& operator*() const { return *mComponent; }
  T* operator->() const { return mComponent; }
  T* get() const { return mComponent; }
  void reset() {
    mNode = nullptr;
    mComponent = nullptr;
  }

 private:
  Node* mNode = nullptr;
  T* mComponent = nullptr;
};

class NodeGraph {
 public:
  bool empty() const { return mNodes.empty(); }
  bool contains(const Node* ptr) const { return mNodes.count(ptr) != 0; }
  inline Node& emplace(const std::string& name);
  inline void erase(Node& node);
  template <typename T>
  size_t component_count() const {
    auto it = mComponentMap.find(typeid(T));
    return it == mComponentMap.end() ? 0 : it->second.mComponents.size();
  }
  inline ~NodeGraph();

 private:
  friend class Node;
  struct component_map {
    void (*mDestructor)(const void*);
    std::unordered_map<const Node*, void*> mComponents;
  };
  std::unordered_map<const Node*, std::unique_ptr<Node>> mNodes;
  std::unordered_map<std::type_index, component_map> mComponentMap;
};

class Node {
 public:
  enum EventPriority : uint32_t { eFirst = 0, eAlmostFirst = 0x3FFFFFFF, eDefault = 0x7FFFFFFF, eAlmostLast = 0xBFFFFFFD, eLast = 0xFFFFFFFF };

  template <typename... Args>
  class Event {
   public:
    using function_t = std::function<void(Args...)>;
    void clear() { mListeners.clear(); }
    bool empty() const { return mListeners.empty(); }
    void add_listener(const Node& node, function_t&& fn, uint32_t priority = EventPriority::eDefault) {
      mListeners.emplace_back(&node, std::move(fn), priority);
      std::stable_sort(mListeners.begin(), mListeners.end(), [](const auto& a, const auto& b) { return std::get<2>(a) < std::get<2>(b); });
      if (!mNodeGraph) mNodeGraph = &node.node_graph();
    }
    void erase(const Node& node) {
      for (auto it = mListeners.begin(); it != mListeners.end();) it = (std::get<0>(*it) == &node) ? mListeners.erase(it) : it + 1;
    }
    void operator()(Args... args) const {
      auto tmp = mListeners;  // listeners may add/remove listeners
      for (const auto& l : tmp)
        if (mNodeGraph->contains(std::get<0>(l))) std::get<1>(l)(args...);
    }

   private:
    const NodeGraph* mNodeGraph = nullptr;
    std::vector<std::tuple<const Node*, function_t, uint32_t>> mListeners;
  };

  Node(const Node&) = delete;
  ~Node() {
    for (const std::type_index& t : mComponents) {
      auto& cm = mNodeGraph.mComponentMap.at(t);
      auto it = cm.mComponents.find(this);
      if (it != cm.mComponents.end()) {
        cm.mDestructor(it->second);
        cm.mComponents.erase(it);
      }
    }
  }

  const std::string& name() const { return mName; }
  NodeGraph& node_graph() const { return mNodeGraph; }
  Node* parent() const { return mParent; }
  // children in insertion order (the reference keeps edges in an unordered_multimap, NodeGraph.hpp:152, so its
  // instance order follows the hash function; a deterministic order is kept here so that packing is reproducible)
  const std::vector<Node*>& children() const { return mChildren; }
  void set_parent(Node& parent) {
    if (mParent == &parent) return;
    clear_parent();
    parent.mChildren.push_back(this);
    mParent = &parent;
  }
  void clear_parent() {
    if (!mParent) return;
    auto& c = mParent->mChildren;
    c.erase(std::remove(c.begin(), c.end(), this), c.end());
    mParent = nullptr;
  }
  Node& root() {
    Node* r = this;
    while (r->mParent) r = r->mParent;
    return *r;
  }
  Node& make_child(const std::string& name) {
    Node& n = mNodeGraph.emplace(name);
    n.set_parent(*this);
    return n;
  }

  // make_component<T>(args...): T(args...), T(Node*, args...) or T(Node&, args...), NodeGraph.hpp:247-268
  template <typename T, typename... Args>
  component_ptr<T> make_component(Args&&... args) {
    if (mComponents.count(typeid(T))) throw std::logic_error("Cannot make multiple components of the same type within the same node");
    auto it = mNodeGraph.mComponentMap.find(typeid(T));
    if (it == mNodeGraph.mComponentMap.end())
      it = mNodeGraph.mComponentMap.emplace(typeid(T), NodeGraph::component_map{[](const void* p) { delete reinterpret_cast<const T*>(p); }, {}}).first;
    T* ptr;
    if constexpr (std::is_constructible_v<T, Node&, Args...>)
      ptr = new T(*this, std::forward<Args>(args)...);
    else if constexpr (std::is_constructible_v<T, Node*, Args...>)
      ptr = new T(this, std::forward<Args>(args)...);
    else
      ptr = new T(std::forward<Args>(args)...);
    mComponents.emplace(typeid(T));
    it->second.mComponents[this] = ptr;
    return component_ptr<T>(this, ptr);
  }
  template <typename T>
  component_ptr<T> find() const {
    auto it = mNodeGraph.mComponentMap.find(typeid(T));
    if (it == mNodeGraph.mComponentMap.end()) return {};
    auto c = it->second.mComponents.find(this);
    return c == it->second.mComponents.end() ? component_ptr<T>() : component_ptr<T>(this, reinterpret_cast<T*>(c->second));
  }
  template <typename T>
  component_ptr<T> find_in_ancestor() const {
    for (const Node* n = this; n; n = n->mParent)
      if (auto c = n->find<T>()) return c;
    return {};
  }
  template <typename T>
  component_ptr<T> find_in_descendants() const {
    component_ptr<T> r;
    for_each_descendant<T>([&](const component_ptr<T>& c) {
      if (!r) r = c;
    });
    return r;
  }
  // breadth-first over this node and its descendants, NodeGraph.hpp:329-346
  template <typename T, typename F>
  void for_each_descendant(F&& fn) const {
    std::queue<const Node*> q;
    q.push(this);
    while (!q.empty()) {
      const Node* n = q.front();
      q.pop();
      if (auto c = n->find<T>()) fn(c);
      for (Node* c : n->children()) q.push(c);
    }
  }

 private:
  friend class NodeGraph;
  Node(NodeGraph& g, const std::string& name) : mNodeGraph(g), mName(name), mParent(nullptr) {}
  NodeGraph& mNodeGraph;
  std::string mName;
  Node* mParent;
  std::unordered_set<std::type_index> mComponents;
  std::vector<Node*> mChildren;
};

inline Node& NodeGraph::emplace(const std::string& name) {
  Node* ptr = new Node(*this, name);
  mNodes.emplace(ptr, std::unique_ptr<Node>(ptr));
  return *ptr;
}
inline void NodeGraph::erase(Node& node) {
  const std::vector<Node*> kids = node.children();
  for (Node* c : kids) c->clear_parent();
  node.clear_parent();
  mNodes.erase(&node);
}
inline NodeGraph::~NodeGraph() {
  for (auto& n : mNodes) {
    n.second->mParent = nullptr;
    n.second->mChildren.clear();
  }
  mNodes.clear();
}

// ------------------------------------------------------------------------------------------------
// vector / transform helpers (Common/hlsl_compat.hpp aliases Eigen; plain structs here)
// ------------------------------------------------------------------------------------------------
struct float2 {
  float x = 0, y = 0;
};
struct float3 {
  float x = 0, y = 0, z = 0;
  float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct quatf {
  float3 xyz;
  float w = 1;
};
inline quatf quatf_identity() { return quatf{{0, 0, 0}, 1}; }
inline quatf angle_axis(float angle, float3 axis) {  // quatf.h:27-29
  const float s = std::sin(angle / 2);
  return quatf{{axis.x * s, axis.y * s, axis.z * s}, std::cos(angle / 2)};
}

using TransformData = sthip_TransformData;
using ProjectionData = sthip_ProjectionData;
using ViewData = sthip_ViewData;
using InstanceData = sthip_InstanceData;
using PackedVertexData = sthip_PackedVertexData;
using BDPTPushConstants = sthip_BDPTPushConstants;
using VisibilityInfo = sthip_VisibilityInfo;
using DepthInfo = sthip_DepthInfo;

// make_transform, transform.h:47-52: translation * rotation(quaternion) * scaling, the standard rotation matrix
// Eigen produces on the reference's host side
inline TransformData make_transform(float3 t, quatf r, float3 s) {
  TransformData o;
  const float sqw = r.w * r.w, sqx = r.xyz.x * r.xyz.x, sqy = r.xyz.y * r.xyz.y, sqz = r.xyz.z * r.xyz.z;
  const float invs = 1 / (sqx + sqy + sqz + sqw);
  float m[3][3];
  m[0][0] = (sqx - sqy - sqz + sqw) * invs;
  m[1][1] = (-sqx + sqy - sqz + sqw) * invs;
  m[2][2] = (-sqx - sqy + sqz + sqw) * invs;
  float tmp1 = r.xyz.x * r.xyz.y, tmp2 = r.xyz.z * r.w;
  m[1][0] = 2 * (tmp1 + tmp2) * invs;
  m[0][1] = 2 * (tmp1 - tmp2) * invs;
  tmp1 = r.xyz.x * r.xyz.z;
  tmp2 = r.xyz.y * r.w;
  m[2][0] = 2 * (tmp1 - tmp2) * invs;
  m[0][2] = 2 * (tmp1 + tmp2) * invs;
  tmp1 = r.xyz.y * r.xyz.z;
  tmp2 = r.xyz.x * r.w;
  m[2][1] = 2 * (tmp1 + tmp2) * invs;
  m[1][2] = 2 * (tmp1 - tmp2) * invs;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) o.m[i][j] = m[i][j] * s[j];  // rotation * scaling
    o.m[i][3] = t[i];
  }
  return o;
}
inline TransformData transform_identity() { return make_transform({0, 0, 0}, quatf_identity(), {1, 1, 1}); }
// transform.h:88-104
inline TransformData tmul(const TransformData& a, const TransformData& b) {
  TransformData r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 4; j++) {
      float s = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
      if (j == 3) s += a.m[i][3];
      r.m[i][j] = s;
    }
  return r;
}
// TransformData::inverse(), transform.h:26-31: the reference takes Eigen's 4x4 inverse. Pinned here (and in
// stratum_amd/scene.py) as the adjugate of the 3x3 block over its determinant, evaluated in double in this
// exact order, then 0 - (inv * t); identity in, identity out.
inline TransformData inverse(const TransformData& t) {
  const double a00 = t.m[0][0], a01 = t.m[0][1], a02 = t.m[0][2];
  const double a10 = t.m[1][0], a11 = t.m[1][1], a12 = t.m[1][2];
  const double a20 = t.m[2][0], a21 = t.m[2][1], a22 = t.m[2][2];
  const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
  const double det = a00 * c00 + a01 * c01 + a02 * c02;
  const double i00 = c00 / det, i01 = (a02 * a21 - a01 * a22) / det, i02 = (a01 * a12 - a02 * a11) / det;
  const double i10 = c01 / det, i11 = (a00 * a22 - a02 * a20) / det, i12 = (a02 * a10 - a00 * a12) / det;
  const double i20 = c02 / det, i21 = (a01 * a20 - a00 * a21) / det, i22 = (a00 * a11 - a01 * a10) / det;
  const double tx = t.m[0][3], ty = t.m[1][3], tz = t.m[2][3];
  TransformData r;
  r.m[0][0] = (float)i00, r.m[0][1] = (float)i01, r.m[0][2] = (float)i02, r.m[0][3] = (float)(0.0 - (i00 * tx + i01 * ty + i02 * tz));
  r.m[1][0] = (float)i10, r.m[1][1] = (float)i11, r.m[1][2] = (float)i12, r.m[1][3] = (float)(0.0 - (i10 * tx + i11 * ty + i12 * tz));
  r.m[2][0] = (float)i20, r.m[2][1] = (float)i21, r.m[2][2] = (float)i22, r.m[2][3] = (float)(0.0 - (i20 * tx + i21 * ty + i22 * tz));
  return r;
}
// transform.h:159-168
inline ProjectionData make_perspective(float fovy, float aspect, float2 offset, float znear) {
  ProjectionData r{};
  r.scale[1] = 1 / std::tan(fovy / 2);
  r.scale[0] = aspect * r.scale[1];
  r.offset[0] = offset.x;
  r.offset[1] = offset.y;
  r.near_plane = znear;
  r.far_plane = 0;
  r.vertical_fov = fovy;
  return r;
}
inline float3 back_project(const ProjectionData& p, float2 v) {  // transform.h:136-147
  float3 r;
  if (p.vertical_fov < 0) {
    r.x = (v.x - p.offset[0]) / p.scale[0];
    r.y = (v.y - p.offset[1]) / p.scale[1];
  } else {
    const float s = p.near_plane > 0 ? 1.f : (p.near_plane < 0 ? -1.f : 0.f);
    r.x = p.near_plane * (v.x * s - p.offset[0]) / p.scale[0];
    r.y = p.near_plane * (v.y * s - p.offset[1]) / p.scale[1];
  }
  r.z = p.near_plane;
  return r;
}

// node_to_world, Scene.cpp:108-117
inline TransformData node_to_world(const Node& node) {
  TransformData transform = transform_identity();
  for (const Node* p = &node; p != nullptr; p = p->parent())
    if (auto c = p->find<TransformData>()) transform = tmul(*c, transform);
  return transform;
}

// ------------------------------------------------------------------------------------------------
// scene components: Material.hpp:13-38, Scene.hpp:15-37
// ------------------------------------------------------------------------------------------------
// an RGBA32F texture (what the reference holds as an Image::View of a Texture2D<float4>)
struct Image {
  uint32_t width = 0, height = 0;
  std::vector<float> pixels;  // width * height * 4, row 0 first
};
// a one-channel float texture (Texture2D<float>): what Material::alpha_mask holds (R8Unorm coverage upstream, Scene.cpp:182)
struct Image1 {
  uint32_t width = 0, height = 0;
  std::vector<float> pixels;  // width * height, row 0 first
};
// MaterialResources, image_value.h:34-66: images get their gImages / gImage1s index the first time a material stores them
struct MaterialResources {
  std::vector<const Image*> image4s;
  std::vector<const Image1*> image1s;
  uint32_t get_index(const component_ptr<Image1>& image) {
    if (!image) return ~0u;
    for (size_t i = 0; i < image1s.size(); i++)
      if (image1s[i] == image.get()) return (uint32_t)i;
    image1s.push_back(image.get());
    return (uint32_t)image1s.size() - 1;
  }
  std::vector<const std::vector<uint8_t>*> volumes;  // volume_data_map: NanoVDB buffers in first-use order (gVolumes)
  uint32_t get_index(const std::shared_ptr<std::vector<uint8_t>>& buffer) {  // image_value.h:49-55
    if (!buffer) return ~0u;
    for (size_t i = 0; i < volumes.size(); i++)
      if (volumes[i] == buffer.get()) return (uint32_t)i;
    volumes.push_back(buffer.get());
    return (uint32_t)volumes.size() - 1;
  }
  std::vector<std::pair<const std::vector<float>*, uint32_t>> distribution_data_map;  // table -> offset in gDistributions
  uint32_t distribution_data_size = 0;
  uint32_t get_index(const component_ptr<Image>& image) {
    if (!image) return ~0u;
    for (size_t i = 0; i < image4s.size(); i++)
      if (image4s[i] == image.get()) return (uint32_t)i;
    image4s.push_back(image.get());
    return (uint32_t)image4s.size() - 1;
  }
  uint32_t get_index(const std::vector<float>& data) {  // image_value.h:56-66
    for (const auto& e : distribution_data_map)
      if (e.first == &data) return e.second;
    const uint32_t r = distribution_data_size;
    distribution_data_map.emplace_back(&data, r);
    distribution_data_size += (uint32_t)data.size();
    return r;
  }
};
struct ImageValue4 {
  float value[4] = {0, 0, 0, 0};
  component_ptr<Image> image;  // null: constant value
};
struct Material {
  ImageValue4 values[3];
  component_ptr<Image1> alpha_mask;  // Material.hpp:14
  component_ptr<Image> bump_image;
  float bump_strength = 1;
  float* base_color() { return values[0].value; }
  float& emission() { return values[0].value[3]; }
  float& metallic() { return values[1].value[0]; }
  float& roughness() { return values[1].value[1]; }
  float& anisotropic() { return values[1].value[2]; }
  float& subsurface() { return values[1].value[3]; }
  float& clearcoat() { return values[2].value[0]; }
  float& clearcoat_gloss() { return values[2].value[1]; }
  float& transmission() { return values[2].value[2]; }
  float& eta() { return values[2].value[3]; }
  // Material::store, Material.hpp:32-38
  void store(std::vector<uint32_t>& bytes, MaterialResources& resources) const {
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 4; j++) {
        uint32_t u;
        std::memcpy(&u, &values[i].value[j], 4);
        bytes.push_back(u);
      }
      bytes.push_back(resources.get_index(values[i].image));
    }
    bytes.push_back(resources.get_index(alpha_mask));
    bytes.push_back(resources.get_index(bump_image));
    uint32_t u;
    std::memcpy(&u, &bump_strength, 4);
    bytes.push_back(u);
  }
};
// what the reference's Mesh + copy_vertices produce for the path (positions, normals, uvs, indices)
struct Mesh {
  std::vector<float3> positions, normals;
  std::vector<float2> uvs;
  std::vector<uint32_t> indices;
  uint32_t index_stride = 4;  // 2 or 4 bytes
};
struct MeshPrimitive {
  component_ptr<Material> mMaterial;
  component_ptr<Mesh> mMesh;
};
struct SpherePrimitive {  // Scene.hpp:34-37
  component_ptr<Material> mMaterial;
  float mRadius = 1;
};
// Material.hpp:72-87: a heterogeneous medium over NanoVDB float grids; the buffers are the bytes of a nanovdb::GridHandle
struct Medium {
  float density_scale[3] = {1, 1, 1};
  float anisotropy = 0;
  float albedo_scale[3] = {1, 1, 1};
  float attenuation_unit = 1;
  std::shared_ptr<std::vector<uint8_t>> density_buffer, albedo_buffer;
  void store(std::vector<uint32_t>& bytes, MaterialResources& pool) const {
    auto f = [&](float v) { uint32_t u; std::memcpy(&u, &v, 4); bytes.push_back(u); };
    for (float v : density_scale) f(v);
    f(anisotropy);
    for (float v : albedo_scale) f(v);
    f(attenuation_unit);
    bytes.push_back(pool.get_index(density_buffer));
    bytes.push_back(pool.get_index(albedo_buffer));
  }
  // GridData::mWorldBBox (6 doubles at byte 560 of the grid): what worldBBox() returns
  bool world_bbox(double box[6]) const {
    if (!density_buffer || density_buffer->size() < 608) return false;
    std::memcpy(box, density_buffer->data() + 560, 48);
    return true;
  }
};

// dist2.h:80-154 build_distributions for a lat-long RGBA32F image: f(x, y) = luminance * sin(pi (y + 0.5) / H) is a
// double (float luminance times double sin), every running sum is stored as float, exactly as the reference's loops
inline void build_distributions(const Image& img, std::vector<float>& pdf_marginals, std::vector<float>& pdf_rows, std::vector<float>& cdf_marginals, std::vector<float>& cdf_rows) {
  const uint32_t W = img.width, H = img.height;
  pdf_marginals.assign(H, 0.f);
  pdf_rows.assign((size_t)W * H, 0.f);
  cdf_marginals.assign(H + 1, 0.f);
  cdf_rows.assign((size_t)(W + 1) * H, 0.f);
  const float invHeight = 1 / (float)H;
  auto f = [&](uint32_t x, uint32_t y) -> double {
    const float* c = &img.pixels[4 * ((size_t)y * W + x)];
    const float lum = (c[0] * 0.2126f + c[1] * 0.7152f) + c[2] * 0.0722f;
    return lum * std::sin(M_PI * (y + 0.5f) * invHeight);
  };
  for (uint32_t y = 0; y < H; y++) {
    float* row = &cdf_rows[(size_t)y * (W + 1)];
    row[0] = 0;
    for (uint32_t x = 0; x < W; x++) row[x + 1] = (float)(row[x] + f(x, y));
    const float integral = row[W];
    if (integral > 0) {
      for (uint32_t x = 0; x < W; x++) row[x] /= integral;
      for (uint32_t x = 0; x < W; x++) pdf_rows[(size_t)y * W + x] = (float)(f(x, y) / integral);
    } else {
      for (uint32_t x = 0; x < W; x++) {
        pdf_rows[(size_t)y * W + x] = float(1) / float(W);
        row[x] = float(x) / float(W);
      }
      row[W] = 1;
    }
  }
  cdf_marginals[0] = 0;
  for (uint32_t y = 0; y < H; y++) cdf_marginals[y + 1] = cdf_marginals[y] + cdf_rows[(size_t)y * (W + 1) + W];
  const float total_values = cdf_marginals[H];
  if (total_values > 0) {
    for (uint32_t y = 0; y < H; y++) cdf_marginals[y] /= total_values;
    cdf_marginals[H] = 1;
    for (uint32_t y = 0; y < H; y++) pdf_marginals[y] = cdf_rows[(size_t)y * (W + 1) + W] / total_values;
  } else {
    for (uint32_t y = 0; y < H; y++) {
      pdf_marginals[y] = float(1) / float(H);
      cdf_marginals[y] = float(y) / float(H);
    }
    cdf_marginals[H] = 1;
  }
  for (uint32_t y = 0; y < H; y++) cdf_rows[(size_t)y * (W + 1) + W] = 1;
}

// environment.h:8-25,96-150: constant radiance, or a lat-long image scaled by `value` with its sampling tables
struct Environment {
  float value[3] = {0, 0, 0};
  component_ptr<Image> image;
  std::vector<float> marginal_pdf, row_pdf, marginal_cdf, row_cdf;
  bool is_zero() const { return value[0] == 0 && value[1] == 0 && value[2] == 0; }
  void store(std::vector<uint32_t>& bytes, MaterialResources& resources) const {
    for (int j = 0; j < 3; j++) {
      uint32_t u;
      std::memcpy(&u, &value[j], 4);
      bytes.push_back(u);
    }
    bytes.push_back(resources.get_index(image));
    if (image) {
      bytes.push_back(resources.get_index(marginal_pdf));
      bytes.push_back(resources.get_index(row_pdf));
      bytes.push_back(resources.get_index(marginal_cdf));
      bytes.push_back(resources.get_index(row_cdf));
    }
  }
};
// load_environment, environment.h:99-150, for an image that is already in memory
inline Environment make_environment(const component_ptr<Image>& image, float r = 1, float g = 1, float b = 1) {
  Environment e;
  e.value[0] = r, e.value[1] = g, e.value[2] = b;
  e.image = image;
  if (image) build_distributions(*image, e.marginal_pdf, e.row_pdf, e.marginal_cdf, e.row_cdf);
  return e;
}
struct Rect2D {
  int32_t x = 0, y = 0;
  uint32_t width = 0, height = 0;
};
struct Camera {
  ProjectionData mProjection{};
  Rect2D mImageRect;
  ViewData view() const {  // Scene.hpp:19-28
    ViewData v{};
    v.projection = mProjection;
    v.image_min[0] = mImageRect.x;
    v.image_min[1] = mImageRect.y;
    v.image_max[0] = mImageRect.x + (int32_t)mImageRect.width;
    v.image_max[1] = mImageRect.y + (int32_t)mImageRect.height;
    const float3 a = back_project(mProjection, {1, 1}), b = back_project(mProjection, {-1, -1});
    float ex = a.x - b.x, ey = a.y - b.y;
    if (mProjection.vertical_fov >= 0) {
      ex /= mProjection.near_plane;
      ey /= mProjection.near_plane;
    }
    v.projection.sensor_area = std::fabs(ex * ey);
    return v;
  }
};

// the HIP stream stands in for the Vulkan command buffer the reference threads through update()/render()
struct CommandBuffer {
  void* hip_stream = nullptr;
};

// Application.hpp:11-29: only the two events the renderer hooks
class Application {
 public:
  Node::Event<CommandBuffer&, float> OnUpdate;
  Node::Event<CommandBuffer&> OnRenderWindow;
  explicit Application(Node& node) : mNode(node) {}
  Node& node() const { return mNode; }
  void run_frame(CommandBuffer& cb, float dt = 0) {  // Application.cpp:64,69
    OnUpdate(cb, dt);
    OnRenderWindow(cb);
  }

 private:
  Node& mNode;
};

// ------------------------------------------------------------------------------------------------
// Scene: Scene.hpp:44-77, Scene::update Scene.cpp:299-684 (mesh instances only)
// ------------------------------------------------------------------------------------------------
class Scene {
 public:
  struct SceneData {
    std::vector<PackedVertexData> mVertices;
    std::vector<uint8_t> mIndices;
    std::vector<uint32_t> mMaterialData;
    std::vector<InstanceData> mInstances;
    std::vector<TransformData> mInstanceTransforms, mInstanceInverseTransforms, mInstanceMotionTransforms;
    std::vector<uint32_t> mLightInstanceMap;
    std::vector<Node*> mInstanceNodes;
    MaterialResources mResources;
    std::vector<sthip_image_desc> mImageDescs, mImage1Descs;
    std::vector<sthip_volume_desc> mVolumeDescs;                          // gVolumes
    std::vector<std::pair<const Medium*, uint32_t>> mMediumInstances;     // Medium component -> its volume instance (mInstanceTransformMap)
    std::vector<float> mDistributionData;  // gDistributions, Scene.cpp:670-683
    uint32_t mEnvironmentMaterialAddress = ~0u;
    uint32_t mMaterialCount = 0;
    uint32_t mEmissivePrimitiveCount = 0;
    sthip_scene_desc desc() const {
      sthip_scene_desc d{};
      d.gVertices = mVertices.data();
      d.vertex_count = (uint32_t)mVertices.size();
      d.gIndices = mIndices.data();
      d.indices_bytes = (uint32_t)mIndices.size();
      d.gInstances = mInstances.data();
      d.instance_count = (uint32_t)mInstances.size();
      d.gInstanceTransforms = mInstanceTransforms.data();
      d.gInstanceInverseTransforms = mInstanceInverseTransforms.data();
      d.gInstanceMotionTransforms = mInstanceMotionTransforms.data();
      d.gMaterialData = mMaterialData.data();
      d.material_bytes = (uint32_t)(mMaterialData.size() * 4);
      d.gLightInstances = mLightInstanceMap.data();
      d.light_count = (uint32_t)mLightInstanceMap.size();
      d.gImages = mImageDescs.data();
      d.image_count = (uint32_t)mImageDescs.size();
      d.gImage1s = mImage1Descs.data();
      d.image1_count = (uint32_t)mImage1Descs.size();
      d.gDistributions = mDistributionData.empty() ? nullptr : mDistributionData.data();
      d.distribution_count = (uint32_t)mDistributionData.size();
      d.gVolumes = mVolumeDescs.empty() ? nullptr : mVolumeDescs.data();
      d.volume_count = (uint32_t)mVolumeDescs.size();
      return d;
    }
  };

  explicit Scene(Node& node) : mNode(node) {
    if (auto app = node.find_in_ancestor<Application>()) app->OnUpdate.add_listener(node, [this](CommandBuffer& cb, float dt) { update(cb, dt); }, Node::EventPriority::eDefault);
  }
  Node& node() const { return mNode; }
  const std::shared_ptr<SceneData>& data() const { return mSceneData; }
  void mark_dirty() { mDirty = true; }

  // Scene.cpp:299-684 for MeshPrimitive instances: same traversal order (breadth-first), same packing
  void update(CommandBuffer&, float) {
    if (!mDirty && mSceneData) return;
    // the previous frame's object-to-world transform of every instance's node (mInstanceTransformMap, Scene.cpp:398-427):
    // the motion transform is prev_object_to_world x world_to_object (make_instance_motion_transform, scene.h:49)
    std::unordered_map<const Node*, TransformData> prevTransforms;
    if (mSceneData)
      for (size_t i = 0; i < mSceneData->mInstanceNodes.size(); i++) prevTransforms.emplace(mSceneData->mInstanceNodes[i], mSceneData->mInstanceTransforms[i]);
    auto prev_of = [&](const Node& node, const TransformData& current) {
      auto it = prevTransforms.find(&node);
      return it == prevTransforms.end() ? current : it->second;
    };
    auto sd = std::make_shared<SceneData>();
    std::unordered_map<const Material*, uint32_t> materialMap;
    std::unordered_map<const Mesh*, std::pair<uint32_t, uint32_t>> meshMap;  // first_vertex, indices_byte_offset (shared by instances of one mesh)
    mNode.root().for_each_descendant<MeshPrimitive>([&](const component_ptr<MeshPrimitive>& prim) {
      if (!prim->mMesh || !prim->mMaterial) return;
      const Mesh& mesh = *prim->mMesh;
      // process_material, Scene.cpp:387-396
      auto mit = materialMap.find(prim->mMaterial.get());
      if (mit == materialMap.end()) {
        mit = materialMap.emplace(prim->mMaterial.get(), (uint32_t)(sd->mMaterialData.size() * sizeof(uint32_t))).first;
        prim->mMaterial->store(sd->mMaterialData, sd->mResources);
        sd->mMaterialCount++;
      }
      // copy_vertices + concatenation, copy_vertices.hlsl:29-37, Scene.cpp:461-485,643-658
      auto meshit = meshMap.find(&mesh);
      if (meshit == meshMap.end()) {
        meshit = meshMap.emplace(&mesh, std::make_pair((uint32_t)sd->mVertices.size(), (uint32_t)sd->mIndices.size())).first;
        for (size_t i = 0; i < mesh.positions.size(); i++) {
          PackedVertexData v{};
          v.position[0] = mesh.positions[i].x, v.position[1] = mesh.positions[i].y, v.position[2] = mesh.positions[i].z;
          if (i < mesh.normals.size()) v.normal[0] = mesh.normals[i].x, v.normal[1] = mesh.normals[i].y, v.normal[2] = mesh.normals[i].z;
          if (i < mesh.uvs.size()) v.u = mesh.uvs[i].x, v.v = mesh.uvs[i].y;
          sd->mVertices.push_back(v);
        }
        for (uint32_t idx : mesh.indices) {
          if (mesh.index_stride == 2) {
            const uint16_t s = (uint16_t)idx;
            sd->mIndices.insert(sd->mIndices.end(), (const uint8_t*)&s, (const uint8_t*)&s + 2);
          } else {
            sd->mIndices.insert(sd->mIndices.end(), (const uint8_t*)&idx, (const uint8_t*)&idx + 4);
          }
        }
        while (sd->mIndices.size() & 3) sd->mIndices.push_back(0);  // align_up(size, 4), Scene.cpp:507
      }
      const uint32_t triCount = (uint32_t)(mesh.indices.size() / 3);
      if (triCount > 0xFFFF) throw std::invalid_argument("MeshPrimitive has more than 65535 triangles (16-bit primitive index, scene.h:23-24,37)");
      const TransformData transform = node_to_world(prim.node());
      // make_instance_triangles, scene.h:51-61; process_instance, Scene.cpp:398-427
      InstanceData inst{};
      inst.packed[0] = STHIP_INSTANCE_TYPE_TRIANGLES | (mit->second << 4);
      inst.packed[1] = 0xFFFu | (triCount << 12) | (mesh.index_stride << 28);
      inst.packed[2] = meshit->second.first;
      inst.packed[3] = meshit->second.second;
      const uint32_t instance_index = (uint32_t)sd->mInstances.size();
      if (prim->mMaterial->emission() > 0) {
        inst.packed[1] = (inst.packed[1] & ~0xFFFu) | ((uint32_t)sd->mLightInstanceMap.size() & 0xFFFu);
        sd->mLightInstanceMap.push_back(instance_index);
        sd->mEmissivePrimitiveCount += triCount;
      }
      sd->mInstances.push_back(inst);
      sd->mInstanceNodes.push_back(&prim.node());
      const TransformData inv = inverse(transform);
      sd->mInstanceTransforms.push_back(transform);
      sd->mInstanceInverseTransforms.push_back(inv);
      sd->mInstanceMotionTransforms.push_back(tmul(prev_of(prim.node(), transform), inv));  // make_instance_motion_transform(inv, prevObjectToWorld), scene.h:49
    });
    // sphere instances, Scene.cpp:511-553 (after every mesh instance)
    mNode.root().for_each_descendant<SpherePrimitive>([&](const component_ptr<SpherePrimitive>& prim) {
      if (!prim->mMaterial) return;
      auto mit = materialMap.find(prim->mMaterial.get());
      if (mit == materialMap.end()) {
        mit = materialMap.emplace(prim->mMaterial.get(), (uint32_t)(sd->mMaterialData.size() * sizeof(uint32_t))).first;
        prim->mMaterial->store(sd->mMaterialData, sd->mResources);
        sd->mMaterialCount++;
      }
      if (prim->mMaterial->emission() > 0) sd->mEmissivePrimitiveCount++;
      TransformData transform = node_to_world(prim.node());
      // :520-521: the radius is scaled by the DETERMINANT of the 3x3 block, the instance keeps only the translation
      const float(*m)[4] = transform.m;
      const float det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
      const float r = prim->mRadius * det;
      TransformData t{};
      t.m[0][0] = t.m[1][1] = t.m[2][2] = 1;
      t.m[0][3] = m[0][3], t.m[1][3] = m[1][3], t.m[2][3] = m[2][3];
      InstanceData inst{};  // make_instance_sphere, scene.h:62-70
      inst.packed[0] = STHIP_INSTANCE_TYPE_SPHERE | (mit->second << 4);
      inst.packed[1] = 0xFFFu;
      std::memcpy(&inst.packed[2], &r, 4);
      const uint32_t instance_index = (uint32_t)sd->mInstances.size();
      if (prim->mMaterial->emission() * (4 * M_PI * r * r) > 0) {
        inst.packed[1] = (inst.packed[1] & ~0xFFFu) | ((uint32_t)sd->mLightInstanceMap.size() & 0xFFFu);
        sd->mLightInstanceMap.push_back(instance_index);
      }
      sd->mInstances.push_back(inst);
      sd->mInstanceNodes.push_back(&prim.node());
      const TransformData inv = inverse(t);
      sd->mInstanceTransforms.push_back(t);
      sd->mInstanceInverseTransforms.push_back(inv);
      sd->mInstanceMotionTransforms.push_back(tmul(prev_of(prim.node(), t), inv));
    });
    // media, Scene.cpp:556-590 (after every sphere): one volume instance per Medium component
    mNode.root().for_each_descendant<Medium>([&](const component_ptr<Medium>& vol) {
      if (!vol || !vol->density_buffer) return;
      const uint32_t material_address = (uint32_t)(sd->mMaterialData.size() * sizeof(uint32_t));
      sd->mMaterialCount++;
      vol->store(sd->mMaterialData, sd->mResources);
      InstanceData inst{};  // make_instance_volume, scene.h:71-79
      inst.packed[0] = STHIP_INSTANCE_TYPE_VOLUME | (material_address << 4);
      inst.packed[1] = 0xFFFu;
      inst.packed[2] = sd->mResources.get_index(vol->density_buffer);
      const TransformData transform = node_to_world(vol.node());
      sd->mMediumInstances.emplace_back(vol.get(), (uint32_t)sd->mInstances.size());
      sd->mInstances.push_back(inst);
      sd->mInstanceNodes.push_back(&vol.node());
      const TransformData inv = inverse(transform);
      sd->mInstanceTransforms.push_back(transform);
      sd->mInstanceInverseTransforms.push_back(inv);
      sd->mInstanceMotionTransforms.push_back(tmul(prev_of(vol.node(), transform), inv));
    });
    for (const auto* v : sd->mResources.volumes) sd->mVolumeDescs.push_back(sthip_volume_desc{v->data(), (uint64_t)v->size()});
    // environment material, Scene.cpp:631-640: the first Environment whose value is not zero
    mNode.root().for_each_descendant<Environment>([&](const component_ptr<Environment>& environment) {
      if (environment && !environment->is_zero() && sd->mEnvironmentMaterialAddress == ~0u) {
        sd->mEnvironmentMaterialAddress = (uint32_t)(sd->mMaterialData.size() * sizeof(uint32_t));
        sd->mMaterialCount++;
        environment->store(sd->mMaterialData, sd->mResources);
      }
    });
    for (const Image* im : sd->mResources.image4s) sd->mImageDescs.push_back(sthip_image_desc{im->pixels.data(), im->width, im->height});
    for (const Image1* im : sd->mResources.image1s) sd->mImage1Descs.push_back(sthip_image_desc{im->pixels.data(), im->width, im->height});
    sd->mDistributionData.resize(sd->mResources.distribution_data_size);
    for (const auto& e : sd->mResources.distribution_data_map) std::copy(e.first->begin(), e.first->end(), sd->mDistributionData.begin() + e.second);  // Scene.cpp:679-680
    mSceneData = sd;
    mDirty = false;
  }

 private:
  Node& mNode;
  std::shared_ptr<SceneData> mSceneData;
  bool mDirty = true;
};

// ------------------------------------------------------------------------------------------------
// BDPT: BDPT.hpp:13-24. Errors surface as exceptions like in the reference (Shader.cpp:115, Instance.cpp:25).
// ------------------------------------------------------------------------------------------------
class BDPT {
 public:
  struct Frame {  // the outputs BDPT::render leaves in its FrameResources (BDPT.cpp:546-605)
    uint32_t width = 0, height = 0;
    std::vector<float> mRadiance, mAlbedo, mPrevUVs;
    std::vector<float> mTonemapResult;  // gOutput of the tone-map block, RGBA32F (BDPT.cpp:558)
    float mTonemapMax[4] = {0, 0, 0, 0};
    std::vector<VisibilityInfo> mVisibility;
    std::vector<DepthInfo> mDepth;
    uint64_t mRayCount[2] = {0, 0};
    std::vector<float> mDebugImage;  // gDebugImage (BDPT.cpp:560): RGBA32F, only with a debug mode
  };

  explicit BDPT(Node& node, int device = 0) : mNode(node) {
    if (sthip_create(device, &mCtx) != STHIP_OK) throw std::runtime_error(std::string("sthip_create: ") + sthip_last_error(nullptr));
    // one frame per call: the reservoir-reuse hash grids of a frame are the next frame's "previous" ones (BDPT.cpp:621-627)
    (void)sthip_set_option(mCtx, "reuse_grids_persist", 1);
    // BDPT.cpp:55-76
    mSamplingFlags = (1u << STHIP_eRemapThreads) | (1u << STHIP_eRayCones) | (1u << STHIP_eSampleBSDFs) | (1u << STHIP_eCoherentRR) | (1u << STHIP_eNormalMaps) |
                     (1u << STHIP_eNEE) | (1u << STHIP_eMIS) | (1u << STHIP_eDeferShadowRays);
    std::memset(&mPushConstants, 0, sizeof(mPushConstants));
    mPushConstants.gMinPathVertices = 4;
    mPushConstants.gMaxPathVertices = 8;
    mPushConstants.gMaxDiffuseVertices = 2;
    mPushConstants.gMaxNullCollisions = 64;
    mPushConstants.gEnvironmentSampleProbability = 0.5f;
    mPushConstants.gLightPresampleTileSize = 1024;
    mPushConstants.gLightPresampleTileCount = 128;
    mPushConstants.gLightPathCount = 64;
    mPushConstants.gReservoirM = 16;
    mPushConstants.gReservoirMaxM = 64;
    mPushConstants.gReservoirSpatialM = 4;
    mPushConstants.gHashGridBucketCount = 200000;
    mPushConstants.gHashGridMinBucketRadius = 0.1f;
    mPushConstants.gHashGridBucketPixelRadius = 6;
    if (auto app = node.find_in_ancestor<Application>())
      app->OnUpdate.add_listener(node, [this](CommandBuffer& cb, float dt) { update(cb, dt); }, Node::EventPriority::eAlmostLast);  // BDPT.cpp:38
  }
  virtual ~BDPT() { sthip_destroy(mCtx); }
  BDPT(const BDPT&) = delete;

  // the instance arguments BDPT's constructor reads (BDPT.cpp:78-127): `minPathVertices`, `maxPathVertices`,
  // `maxDiffuseVertices`, `maxNullCollisions`, `environmentSampleProbability`, `lightPresampleTileSize`,
  // `lightPresampleTileCount`, and `bdptFlag` = [~|!]name with the flag's display name lower-cased, spaces removed
  void set_argument(const std::string& key, const std::string& value) {
    if (key == "bdptFlag") return set_flag(value);
    if (key == "environmentSampleProbability") {
      mPushConstants.gEnvironmentSampleProbability = std::stof(value);
      return;
    }
    static const std::pair<const char*, uint32_t BDPTPushConstants::*> fields[] = {
        {"minPathVertices", &BDPTPushConstants::gMinPathVertices},
        {"maxPathVertices", &BDPTPushConstants::gMaxPathVertices},
        {"maxDiffuseVertices", &BDPTPushConstants::gMaxDiffuseVertices},
        {"maxNullCollisions", &BDPTPushConstants::gMaxNullCollisions},
        {"lightPresampleTileSize", &BDPTPushConstants::gLightPresampleTileSize},
        {"lightPresampleTileCount", &BDPTPushConstants::gLightPresampleTileCount},
        {"lightPathCount", &BDPTPushConstants::gLightPathCount},
        {"reservoirM", &BDPTPushConstants::gReservoirM},
        {"reservoirMaxM", &BDPTPushConstants::gReservoirMaxM},
        {"reservoirSpatialM", &BDPTPushConstants::gReservoirSpatialM},
        {"hashGridBucketCount", &BDPTPushConstants::gHashGridBucketCount},
    };
    for (const auto& f : fields)
      if (key == f.first) mPushConstants.*(f.second) = (uint32_t)std::stoul(value);
  }
  void set_flag(std::string arg) {
    if (arg.empty()) return;
    bool on = true;
    if (arg[0] == '~' || arg[0] == '!') {
      on = false;
      arg = arg.substr(1);
    }
    for (char& c : arg) c = (char)std::tolower((unsigned char)c);
    static const char* names[STHIP_eBDPTFlagCount] = {
        "performancecounters", "remapthreads", "coherentrr", "coherentsampling", "fliptriangleuvs", "flipnormalmaps", "alphatest", "normalmaps",
        "shadingnormalshadowfix", "raycones", "samplebsdfs", "nee", "neereservoirs", "neereservoirreuse", "mis", "samplelightpower",
        "uniformspheresampling", "presamplelights", "defershadowrays", "connecttoviews", "connecttolightpaths", "lightvertexcache", "lvcreservoirs",
        "lvcreservoirreuse", "jitterhashgridlookups", "sampleenvironmentmapdirectly"};
    for (uint32_t i = 0; i < STHIP_eBDPTFlagCount; i++)
      if (arg == names[i]) {  // unknown names are ignored, as upstream
        if (on)
          mSamplingFlags |= 1u << i;
        else
          mSamplingFlags &= ~(1u << i);
      }
  }

  Node& node() const { return mNode; }
  uint32_t& sampling_flags() { return mSamplingFlags; }
  BDPTPushConstants& push_constants() { return mPushConstants; }
  const Frame& prev_result() const { return mPrevFrame; }  // BDPT.hpp:18
  bool last_update_was_transforms_only() const { return mLastUpdateWasTransformsOnly; }
  // tone-map state the reference keeps on its pipeline objects (BDPT.cpp:44-54,190-193,304-309)
  uint32_t& tonemap_mode() { return mTonemapMode; }
  float& exposure() { return mExposure; }
  float& exposure_alpha() { return mExposureAlpha; }
  bool& gamma_correction() { return mGammaCorrection; }
  // "Export" -> "Save" (BDPT.cpp:313-337): the last frame's radiance as a Radiance .hdr file
  void export_hdr(const std::string& path) const {
    if (mPrevFrame.mRadiance.empty()) throw std::runtime_error("BDPT::export_hdr: no frame rendered yet");
    if (sthip_write_hdr(path.c_str(), mPrevFrame.width, mPrevFrame.height, mPrevFrame.mRadiance.data()) != STHIP_OK)
      throw std::runtime_error("BDPT::export_hdr: cannot write " + path);
  }

  // BDPT::update (BDPT.cpp:341-421): (re)bind the scene when Scene::update produced new SceneData
  virtual void update(CommandBuffer& cb, float) {
    auto scene = mNode.find_in_ancestor<Scene>();
    if (!scene) scene = mNode.root().find_in_descendants<Scene>();
    if (!scene || !scene->data() || scene->data().get() == mBound) return;
    (void)sthip_set_stream(mCtx, cb.hip_stream);
    const sthip_scene_desc d = scene->data()->desc();
    // Only instances moved since the bound SceneData (same geometry, materials, images, volumes): the bottom levels in HBM
    // are still right, as the reference's cached BLASes are (Scene.cpp:435-459); rebuild the top level only.
    bool updated = false;
    if (mBoundData && same_geometry(*mBoundData, *scene->data())) {
      const int rc = sthip_scene_update_transforms(mCtx, d.gInstanceTransforms, d.gInstanceInverseTransforms, d.gInstanceMotionTransforms, d.instance_count);
      if (rc == STHIP_OK)
        updated = true;
      else if (rc != STHIP_ERR_UNSUPPORTED)
        throw std::runtime_error(std::string("sthip_scene_update_transforms: ") + sthip_last_error(mCtx));
    }
    mLastUpdateWasTransformsOnly = updated;
    if (!updated && sthip_scene_upload(mCtx, &d) != STHIP_OK) throw std::runtime_error(std::string("sthip_scene_upload: ") + sthip_last_error(mCtx));
    mBound = scene->data().get();
    mBoundData = scene->data();
    mPushConstants.gLightCount = d.light_count;                    // BDPT.cpp:396
    mPushConstants.gEnvironmentMaterialAddress = scene->data()->mEnvironmentMaterialAddress;  // BDPT.cpp:393
  }

  // what a frame hands to sthip_render: the view arrays, the push constants and the scene flags as BDPT::render resolves
  // them (BDPT.cpp:444-503). Shared by the single-device render below and the multi-device one (stratum_hip_multi.hpp).
  struct FrameSetup {
    std::vector<ViewData> v;
    std::vector<TransformData> t, ti;
    std::vector<uint32_t> view_media;
    sthip_frame_desc f{};
    BDPTPushConstants pc{};
    uint32_t scene_flags = 0;
  };
  void prepare_frame(uint32_t width, uint32_t height, const std::vector<std::pair<ViewData, TransformData>>& views, FrameSetup& fs) const {
    if (!mBound) throw std::runtime_error("BDPT::render: no scene bound (Scene::update / BDPT::update have not run)");
    std::vector<ViewData>& v = fs.v;
    std::vector<TransformData>&t = fs.t, &ti = fs.ti;
    for (const auto& p : views) {
      v.push_back(p.first);
      t.push_back(p.second);
      ti.push_back(inverse(p.second));  // BDPT.cpp:448-452
    }
    sthip_frame_desc& f = fs.f;
    f.gViews = v.data();
    f.gViewTransforms = t.data();
    f.gInverseViewTransforms = ti.data();
    f.gPrevViews = mPrevViews.size() == v.size() ? mPrevViews.data() : nullptr;
    f.gPrevInverseViewTransforms = mPrevInverseViewTransforms.size() == ti.size() ? mPrevInverseViewTransforms.data() : nullptr;
    f.view_count = (uint32_t)views.size();
    BDPTPushConstants& pc = fs.pc;
    pc = mPushConstants;
    pc.gOutputExtent[0] = width;
    pc.gOutputExtent[1] = height;
    pc.gViewCount = f.view_count;
    if (!((mSamplingFlags >> STHIP_eLVC) & 1u)) pc.gLightPathCount = width * height;  // BDPT.cpp:469-470: with the cache on it stays the user's value
    uint32_t& scene_flags = fs.scene_flags;  // BDPT.cpp:486-503
    scene_flags = 0;
    if (pc.gEnvironmentMaterialAddress != ~0u)
      scene_flags |= STHIP_BDPT_FLAG_HAS_ENVIRONMENT;
    else
      pc.gEnvironmentSampleProbability = 0;
    if (pc.gLightCount)
      scene_flags |= STHIP_BDPT_FLAG_HAS_EMISSIVES;
    else
      pc.gEnvironmentSampleProbability = 1;
    // media: BDPT.cpp:456-466 (the volume instance each camera is inside of) and :497-500
    std::vector<uint32_t>& view_media = fs.view_media;
    view_media.assign(views.size(), 0xFFFFu);
    if (!mBoundData->mMediumInstances.empty()) {
      scene_flags |= STHIP_BDPT_FLAG_HAS_MEDIA;
      for (const auto& mi : mBoundData->mMediumInstances) {
        double box[6];
        if (!mi.first->world_bbox(box)) continue;
        const TransformData& inv = mBoundData->mInstanceInverseTransforms[mi.second];
        for (size_t i = 0; i < views.size(); i++) {
          const TransformData& vt = views[i].second;
          double p[3];
          for (int a = 0; a < 3; a++) p[a] = (double)inv.m[a][0] * vt.m[0][3] + (double)inv.m[a][1] * vt.m[1][3] + (double)inv.m[a][2] * vt.m[2][3] + inv.m[a][3];
          if (p[0] >= box[0] && p[1] >= box[1] && p[2] >= box[2] && p[0] <= box[3] && p[1] <= box[4] && p[2] <= box[5]) view_media[i] = mi.second;
        }
      }
      f.gViewMediumInstances = view_media.data();
    } else {
      pc.gMaxNullCollisions = 0;
    }
  }

  // BDPT::render (BDPT.cpp:423-838) for the hot path: one sample per pixel per call, seed = frame number (:480)
  virtual void render(CommandBuffer& cb, uint32_t width, uint32_t height, const std::vector<std::pair<ViewData, TransformData>>& views, uint32_t seed_count = 1) {
    FrameSetup fs;
    prepare_frame(width, height, views, fs);
    const sthip_frame_desc& f = fs.f;
    const BDPTPushConstants& pc = fs.pc;
    const uint32_t scene_flags = fs.scene_flags;
    Frame fr;
    fr.width = width;
    fr.height = height;
    const size_t n = (size_t)width * height;
    fr.mRadiance.assign(4 * n, 0.f);
    fr.mAlbedo.assign(4 * n, 0.f);
    fr.mPrevUVs.assign(2 * n, 0.f);
    fr.mVisibility.assign(n, VisibilityInfo{});
    fr.mDepth.assign(n, DepthInfo{});
    sthip_outputs o{};
    o.gRadiance = fr.mRadiance.data();
    o.gAlbedo = fr.mAlbedo.data();
    o.gVisibility = fr.mVisibility.data();
    o.gDepth = fr.mDepth.data();
    o.gPrevUVs = fr.mPrevUVs.data();
    o.gRayCount = fr.mRayCount;
    if (mDebugMode != STHIP_DEBUG_NONE) {  // BDPT.cpp:526-541 (gDebugMode), :560: the image persists from frame to frame
      if (mDebugImage.size() != 4 * n) mDebugImage.assign(4 * n, 0.f);
      o.debug_mode = mDebugMode;
      o.gDebugImage = mDebugImage.data();
    }
    (void)sthip_set_stream(mCtx, cb.hip_stream);
    // BDPT.cpp:474,482-483: a frame whose first camera moved since the last one reuses nothing (gReservoirSpatialM = 0) unless the
    // denoiser reprojects; setting the option drops the grids the last frame left (include/sthip.h)
    const bool changed = !mPrevInverseViewTransforms.empty() && !fs.ti.empty() && std::memcmp(&mPrevInverseViewTransforms[0], &fs.ti[0], sizeof(TransformData)) != 0;
    if (changed && !mReprojection) (void)sthip_set_option(mCtx, "reuse_grids_persist", 1);
    if (sthip_render(mCtx, &pc, mSamplingFlags, scene_flags, &f, mFrameNumber, seed_count, &o) != STHIP_OK)
      throw std::runtime_error(std::string("sthip_render: ") + sthip_last_error(mCtx));
    if (mDebugMode != STHIP_DEBUG_NONE) fr.mDebugImage = mDebugImage;
    finish_frame(std::move(fr), fs, seed_count);
  }
  // BDPTDebugMode (bdpt.h:177-193; the inspector's "Debug mode", BDPT.cpp:253-262) with mPushConstants.gDebugViewPathLength /
  // gDebugLightPathLength; the image starts from zero when the mode or the frame size changes
  void set_debug_mode(uint32_t mode) {
    if (mode != mDebugMode) mDebugImage.clear();
    mDebugMode = mode < STHIP_DEBUG_MODE_COUNT ? mode : (uint32_t)STHIP_DEBUG_NONE;
  }
  uint32_t debug_mode() const { return mDebugMode; }
  // Denoiser::reprojection() (BDPT.cpp:472-473): with it a moving camera keeps the previous frame's reuse grids
  void set_reprojection(bool on) { mReprojection = on; }

 protected:
  uint32_t frame_number() const { return mFrameNumber; }
  uint32_t sampling_flags() const { return mSamplingFlags; }
  // the bookkeeping of a frame that has been handed to the device(s): the next frame's seed and its "previous views"
  // (BDPT.cpp:480,563). finish_frame does it itself unless told that the frame was noted when it was submitted (frames in flight).
  void note_submitted(const FrameSetup& fs, uint32_t seed_count) {
    mFrameNumber += seed_count;
    mPrevViews = fs.v;
    mPrevInverseViewTransforms = fs.ti;
  }
  // what follows the path in BDPT::render: the tone map block and the frame bookkeeping
  void finish_frame(Frame fr, const FrameSetup& fs, uint32_t seed_count, bool noted_at_submit = false) {
    const uint32_t width = fr.width, height = fr.height;
    const size_t n = (size_t)width * height;
    if (fr.mAlbedo.size() != 4 * n) fr.mAlbedo.assign(4 * n, 0.f);
    // tone map (BDPT.cpp:783-815); without a denoiser gModulateAlbedo stays off (:779-780 only run when one exists)
    fr.mTonemapResult.assign(4 * n, 0.f);
    sthip_tonemap_desc tm{};
    tm.width = width;
    tm.height = height;
    tm.mode = mTonemapMode;
    tm.modulate_albedo = 0;
    tm.gamma_correction = mGammaCorrection ? 1u : 0u;
    tm.exposure = mExposure;
    tm.exposure_alpha = mExposureAlpha;       // gExposureAlpha, BDPT.cpp:51,192,307
    tm.exposure_state = mTonemapState;        // mPrevFrame->mTonemapMax bytes 16..39 -> gPrevMax, BDPT.cpp:810-811
    tm.gInput = fr.mDebugImage.size() == 4 * n ? fr.mDebugImage.data() : fr.mRadiance.data();  // BDPT.cpp:764: with a debug mode the debug image is what is shown
    tm.gAlbedo = fr.mAlbedo.data();
    tm.gOutput = fr.mTonemapResult.data();
    tm.out_max = fr.mTonemapMax;
    if (sthip_tonemap(mCtx, &tm) != STHIP_OK) throw std::runtime_error(std::string("sthip_tonemap: ") + sthip_last_error(mCtx));
    if (!noted_at_submit) note_submitted(fs, seed_count);
    mPrevFrame = std::move(fr);
  }

 public:
  void reset_frame_number(uint32_t n = 0) { mFrameNumber = n; }

 protected:
  Node& mNode;
  sthip_ctx* mCtx = nullptr;
  const void* mBound = nullptr;
  std::shared_ptr<Scene::SceneData> mBoundData;
  bool mLastUpdateWasTransformsOnly = false;
  template <typename T>
  static bool same_bytes(const std::vector<T>& a, const std::vector<T>& b) {
    return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(T)) == 0);
  }
  static bool same_geometry(const Scene::SceneData& a, const Scene::SceneData& b) {
    if (!same_bytes(a.mVertices, b.mVertices) || !same_bytes(a.mIndices, b.mIndices) || !same_bytes(a.mInstances, b.mInstances) || !same_bytes(a.mMaterialData, b.mMaterialData) ||
        !same_bytes(a.mLightInstanceMap, b.mLightInstanceMap) || !same_bytes(a.mDistributionData, b.mDistributionData))
      return false;
    if (a.mResources.image4s != b.mResources.image4s || a.mResources.image1s != b.mResources.image1s || a.mResources.volumes != b.mResources.volumes) return false;  // the same objects
    return a.mEnvironmentMaterialAddress == b.mEnvironmentMaterialAddress;
  }
  uint32_t mSamplingFlags = 0;
  BDPTPushConstants mPushConstants;
  uint32_t mFrameNumber = 0;
  uint32_t mDebugMode = STHIP_DEBUG_NONE;
  bool mReprojection = false;
  std::vector<float> mDebugImage;
  uint32_t mTonemapMode = STHIP_TONEMAP_RAW;  // BDPT.cpp:48
  float mExposure = 0;
  float mExposureAlpha = 0;
  float mTonemapState[6] = {0, 0, 0, 0, 0, 0};
  bool mGammaCorrection = true;
  std::vector<ViewData> mPrevViews;
  std::vector<TransformData> mPrevInverseViewTransforms;
  Frame mPrevFrame;
};

// ---------------------------------------------------------------------------------------------
// Plugin loading: the component main.cpp attaches per `--plugin=<lib>;<fn>;<fn>...` argument
// (src/main.cpp:11-24,148-149) over src/Common/dynamic_library.hpp:7-59. Same contract: the library is opened
// with RTLD_NOW when the component is made and stays loaded for the component's life; invoke<R, Args...>(name, args...)
// resolves `name` once (cached) and calls it as R(*)(Args...); a library that does not load throws runtime_error, a
// missing symbol invalid_argument.
// ---------------------------------------------------------------------------------------------
class dynamic_library {
 public:
  // RTLD_NODELETE: components a plugin function makes carry deleters whose code lives in the plugin, and a node graph
  // destroys its components in no particular order — dlclose() then only drops the handle, the code stays mapped
  explicit dynamic_library(const std::string& filename) : mHandle(dlopen(filename.c_str(), RTLD_NOW | RTLD_NODELETE)) {
    if (!mHandle) {
      const char* why = dlerror();
      throw std::runtime_error("Failed to load " + filename + (why ? std::string(": ") + why : std::string()));
    }
  }
  dynamic_library(const dynamic_library&) = delete;
  dynamic_library& operator=(const dynamic_library&) = delete;
  ~dynamic_library() {
    if (mHandle) dlclose(mHandle);
  }
  template <typename return_t, typename... Args>
  return_t invoke(const std::string& name, Args... args) {
    auto it = mFunctionPtrs.find(name);
    if (it == mFunctionPtrs.end()) it = mFunctionPtrs.emplace(name, dlsym(mHandle, name.c_str())).first;
    if (!it->second) throw std::invalid_argument("Could not find function " + name);
    using fn_t = return_t (*)(Args...);
    return reinterpret_cast<fn_t>(it->second)(std::forward<Args>(args)...);
  }

 private:
  void* mHandle;
  std::unordered_map<std::string, void*> mFunctionPtrs;
};

// `plugin_info` = "<library>;<function>;<function>...": a child node named after the library's stem under `dst` carries the
// dynamic_library component, and every named function is called with that child node (main.cpp:11-24)
inline void load_plugins(const std::string& plugin_info, Node& dst) {
  const size_t first = plugin_info.find(';');
  const std::string filename = plugin_info.substr(0, first);
  std::string stem = filename.substr(filename.find_last_of('/') == std::string::npos ? 0 : filename.find_last_of('/') + 1);
  if (stem.find_last_of('.') != std::string::npos && stem.find_last_of('.') > 0) stem = stem.substr(0, stem.find_last_of('.'));
  auto plugin = dst.make_child(stem).make_component<dynamic_library>(filename);
  for (size_t at = first; at != std::string::npos;) {
    const size_t next = plugin_info.find(';', at + 1);
    const std::string fn = plugin_info.substr(at + 1, next == std::string::npos ? std::string::npos : next - at - 1);
    plugin->invoke<void, Node&>(fn, plugin.node());
    at = next;
  }
}

}  // namespace stm
