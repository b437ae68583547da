// host_test.cpp — drives the C++ host side (stratum_amd/host/stratum_hip.hpp) the way Stratum's main.cpp
// drives its components (main.cpp:59-66,97-149): build a node graph, let Application fire OnUpdate /
// OnRenderWindow, read the renderer's result.
//   host_test pack   <scene.bin>                       (no GPU) Scene::update must reproduce the packed arrays
//   host_test render <scene.bin> <out.bin> <seeds> [tonemap_mode exposure out.hdr]   (GPU)    BDPT::update + render, writes RGBA32F radiance
#include <cstdio>
#include <fstream>
#include <iostream>

#include "../../stratum_amd/host/stratum_hip.hpp"
#include "scene_reader.hpp"

using namespace stm;

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: host_test pack|render scene.bin [out.bin seeds]\n");
    return 2;
  }
  try {
    const std::string mode = argv[1];
    Reader r(argv[2]);
    NodeGraph graph;
    Node& root = graph.emplace("Instance");
    auto app = root.make_child("Application").make_component<Application>();
    LoadedScene L = load_scene(r, app.node());
    Node& scene_node = *L.scene_node;
    auto scene = L.scene;
    const ViewData view = L.view;
    const TransformData view_xf = L.view_xf;
    const uint32_t W = L.W, H = L.H;

    CommandBuffer cb;
    if (mode == "pack") {
      scene->update(cb, 0);
      const auto& sd = *scene->data();
      auto check = [&](const char* name, const void* got, size_t got_bytes) {
        const uint64_t n = r.get<uint64_t>();
        const std::vector<uint8_t> want = r.vec<uint8_t>(n);
        if (n != got_bytes || std::memcmp(want.data(), got, n) != 0) {
          std::printf("MISMATCH %s: %zu bytes packed, %llu expected\n", name, got_bytes, (unsigned long long)n);
          std::exit(1);
        }
      };
      check("gVertices", sd.mVertices.data(), sd.mVertices.size() * sizeof(PackedVertexData));
      check("gIndices", sd.mIndices.data(), sd.mIndices.size());
      check("gInstances", sd.mInstances.data(), sd.mInstances.size() * sizeof(InstanceData));
      check("gInstanceTransforms", sd.mInstanceTransforms.data(), sd.mInstanceTransforms.size() * sizeof(TransformData));
      check("gInstanceInverseTransforms", sd.mInstanceInverseTransforms.data(), sd.mInstanceInverseTransforms.size() * sizeof(TransformData));
      check("gInstanceMotionTransforms", sd.mInstanceMotionTransforms.data(), sd.mInstanceMotionTransforms.size() * sizeof(TransformData));
      check("gMaterialData", sd.mMaterialData.data(), sd.mMaterialData.size() * 4);
      check("gLightInstances", sd.mLightInstanceMap.data(), sd.mLightInstanceMap.size() * 4);
      check("gDistributions", sd.mDistributionData.data(), sd.mDistributionData.size() * 4);
      // event order: Scene (eDefault) before BDPT (eAlmostLast) — checked without a device through a probe
      std::vector<int> order;
      Node& probe = app.node().make_child("probe");
      app->OnUpdate.add_listener(probe, [&](CommandBuffer&, float) { order.push_back(2); }, Node::EventPriority::eAlmostLast);
      app->OnUpdate.add_listener(probe, [&](CommandBuffer&, float) { order.push_back(1); }, Node::EventPriority::eFirst);
      app->run_frame(cb);
      if (order != std::vector<int>{1, 2}) {
        std::printf("MISMATCH event priority order\n");
        return 1;
      }
      std::printf("PACK OK %zu instances %zu lights\n", sd.mInstances.size(), sd.mLightInstanceMap.size());
      return 0;
    }
    if (mode == "render" && argc >= 5) {
      const uint32_t seeds = (uint32_t)std::atoi(argv[4]);
      // init_renderer<BDPT>, main.cpp:59-66
      auto renderer = app.node().make_child("BDPT").make_component<BDPT>();
      for (int a = 8; a < argc; a++) {  // instance arguments as --key=value (BDPT.cpp:78-127)
        const std::string kv = argv[a];
        const size_t eq = kv.find('=');
        if (kv.rfind("--", 0) == 0 && eq != std::string::npos) renderer->set_argument(kv.substr(2, eq - 2), kv.substr(eq + 1));
      }
      if (argc >= 8) {  // tone-map settings as the GUI would set them
        renderer->tonemap_mode() = (uint32_t)std::atoi(argv[5]);
        renderer->exposure() = (float)std::atof(argv[6]);
      }
      app->OnRenderWindow.add_listener(renderer.node(), [&](CommandBuffer& c) { renderer->render(c, W, H, {{view, view_xf}}, seeds); });
      app->run_frame(cb);  // OnUpdate: Scene::update, then BDPT::update (eAlmostLast); OnRenderWindow: BDPT::render
      const auto& fr = renderer->prev_result();
      std::ofstream out(argv[3], std::ios::binary);
      out.write((const char*)fr.mRadiance.data(), fr.mRadiance.size() * 4);
      out.write((const char*)fr.mVisibility.data(), fr.mVisibility.size() * sizeof(VisibilityInfo));
      out.write((const char*)fr.mRayCount, 16);
      out.write((const char*)fr.mTonemapResult.data(), fr.mTonemapResult.size() * 4);
      if (argc >= 8) renderer->export_hdr(argv[7]);
      std::printf("RENDER OK %ux%u rays %llu\n", W, H, (unsigned long long)fr.mRayCount[0]);
      // --frames=N: N - 1 more frames of the unchanged scene and camera (the frame number is the seed, BDPT.cpp:480); the last one
      // is written beside the first — what frame-to-frame state (the reservoir-reuse grids) makes of it
      for (int a = 8; a < argc; a++) {
        const std::string kv = argv[a];
        if (kv.rfind("--frames=", 0) != 0) continue;
        const int frames = std::atoi(kv.c_str() + 9);
        for (int i = 1; i < frames; i++) app->run_frame(cb);
        const auto& frn = renderer->prev_result();
        std::ofstream outn(std::string(argv[3]) + ".last", std::ios::binary);
        outn.write((const char*)frn.mRadiance.data(), frn.mRadiance.size() * 4);
        outn.write((const char*)frn.mRayCount, 16);
        std::printf("FRAMES %d\n", frames);
      }
      // --move=dx,dy,dz (must come last): every node whose transform is not the identity moves, the scene is marked
      // dirty, and a second frame is rendered: Scene::update makes a new SceneData (motion transforms from the previous
      // one), BDPT::update sees that only transforms changed and rebuilds the top level alone
      for (int a = 8; a < argc; a++) {
        const std::string kv = argv[a];
        if (kv.rfind("--move=", 0) != 0) continue;
        float dx = 0, dy = 0, dz = 0;
        std::sscanf(kv.c_str() + 7, "%f,%f,%f", &dx, &dy, &dz);
        scene_node.for_each_descendant<TransformData>([&](const component_ptr<TransformData>& t) {
          static const float I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
          if (std::memcmp(t.get(), I, sizeof(I)) == 0) return;
          t->m[0][3] += dx;
          t->m[1][3] += dy;
          t->m[2][3] += dz;
        });
        scene->mark_dirty();
        app->run_frame(cb);
        const auto& fr2 = renderer->prev_result();
        std::ofstream out2(std::string(argv[3]) + ".moved", std::ios::binary);
        out2.write((const char*)fr2.mRadiance.data(), fr2.mRadiance.size() * 4);
        out2.write((const char*)fr2.mPrevUVs.data(), fr2.mPrevUVs.size() * 4);
        std::printf("MOVED transforms_only=%d\n", renderer->last_update_was_transforms_only() ? 1 : 0);
      }
      return 0;
    }
    std::fprintf(stderr, "bad arguments\n");
    return 2;
  } catch (const std::exception& e) {
    std::printf("EXCEPTION %s\n", e.what());
    return 3;
  }
}
