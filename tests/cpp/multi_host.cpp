// multi_host.cpp — the multi-GPU driver of the C++ host side (stratum_amd/host/stratum_hip_multi.hpp).
//   multi_host layout <W> <H> <world> <tile_w> <tile_h>      (no GPU) prints, per rank, the slot count and a checksum of
//        the slot -> pixel map, then packs a synthetic frame by ownership, assembles it on the host and checks identity
//   multi_host render <scene.bin> <out.bin> <seeds> <devices,comma-separated>   (GPU) the node graph of host_test with
//        stm::MultiDeviceBDPT as the renderer: every rank renders its tiles, RCCL gathers them, rank 0 assembles
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../stratum_amd/host/stratum_hip_multi.hpp"
#include "scene_reader.hpp"

using namespace stm;

int main(int argc, char** argv) {
  try {
    if (argc >= 7 && std::string(argv[1]) == "layout") {
      const ShardLayout L{(uint32_t)atoi(argv[2]), (uint32_t)atoi(argv[3]), (uint32_t)atoi(argv[4]), (uint32_t)atoi(argv[5]), (uint32_t)atoi(argv[6])};
      std::vector<float> frame((size_t)L.width * L.height * 4), back(frame.size());
      for (size_t i = 0; i < frame.size(); i++) frame[i] = (float)((i * 2654435761ull) % 1000003ull) + 1.0f;
      std::vector<std::vector<float>> packed(L.world);
      std::vector<const void*> ptrs;
      for (uint32_t r = 0; r < L.world; r++) {
        const uint32_t n = L.slot_count(r);
        packed[r].assign((size_t)n * 4, 0.f);
        uint64_t sum = 0;
        for (uint32_t s = 0; s < n; s++) {
          uint32_t x, y;
          if (L.slot_pixel(r, s, x, y)) {
            if (L.owner_of_pixel(x, y) != r) {
              std::printf("FAIL: slot %u of rank %u maps to a pixel of rank %u\n", s, r, L.owner_of_pixel(x, y));
              return 1;
            }
            std::memcpy(&packed[r][4 * (size_t)s], &frame[4 * ((size_t)y * L.width + x)], 16);
            sum = sum * 1099511628211ull + ((uint64_t)y * L.width + x + 1);
          } else {
            sum = sum * 1099511628211ull;
          }
        }
        std::printf("rank %u slots %u checksum %llu\n", r, n, (unsigned long long)sum);
        ptrs.push_back(packed[r].data());
      }
      L.assemble(ptrs, back.data());
      if (std::memcmp(back.data(), frame.data(), frame.size() * 4) != 0) {
        std::printf("FAIL: the assembled frame differs\n");
        return 1;
      }
      std::printf("LAYOUT OK\n");
      return 0;
    }
    if (argc >= 6 && std::string(argv[1]) == "render") {
      std::vector<int> devices;
      std::stringstream ss(argv[5]);
      for (std::string tok; std::getline(ss, tok, ',');) devices.push_back(atoi(tok.c_str()));
      Reader rd(argv[2]);
      NodeGraph graph;
      Node& root = graph.emplace("Instance");
      auto app = root.make_child("Application").make_component<Application>();
      LoadedScene L = load_scene(rd, app.node());
      auto renderer = app.node().make_child("BDPT").make_component<MultiDeviceBDPT>(devices);
      for (int a = 6; a < argc; a++) {  // instance arguments as --key=value (BDPT.cpp:78-127)
        const std::string kv = argv[a];
        const size_t eq = kv.find('=');
        if (kv.rfind("--", 0) != 0 || eq == std::string::npos) continue;
        if (kv.substr(2, eq - 2) == "splitSeeds")  // the seed-split replica mode: whole frames over disjoint seed ranges + one ncclReduce(sum)
          renderer->split_seeds(kv.substr(eq + 1) != "0");
        else
          renderer->set_argument(kv.substr(2, eq - 2), kv.substr(eq + 1));
      }
      const uint32_t seeds = (uint32_t)atoi(argv[4]);
      app->OnRenderWindow.add_listener(renderer.node(), [&](CommandBuffer& c) { renderer->render(c, L.W, L.H, {{L.view, L.view_xf}}, seeds); });
      CommandBuffer cb;
      app->run_frame(cb);  // OnUpdate: Scene::update, then MultiDeviceBDPT::update on every GPU; OnRenderWindow: the sharded render
      const auto& fr = renderer->prev_result();
      std::ofstream out(argv[3], std::ios::binary);
      out.write((const char*)fr.mRadiance.data(), (std::streamsize)(fr.mRadiance.size() * 4));
      out.write((const char*)fr.mRayCount, 16);
      std::printf("RENDER OK world %zu rays %llu %llu\n", renderer->world(), (unsigned long long)fr.mRayCount[0], (unsigned long long)fr.mRayCount[1]);
      return 0;
    }
    std::fprintf(stderr, "usage: multi_host layout W H world tw th | render scene.bin out.bin seeds devices\n");
    return 2;
  } catch (const std::exception& e) {
    std::printf("ERROR: %s\n", e.what());
    return 1;
  }
}
