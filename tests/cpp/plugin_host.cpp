// plugin_host.cpp — plays the part of Stratum's main.cpp for ONE thing: loading the renderer as a plugin
// (src/main.cpp:11-24,148-149 over src/Common/dynamic_library.hpp:41-59). Usage:
//   plugin_host "<path>/libstratum_hip_plugin.so;stratum_hip_register" [expect-missing-symbol]
// Builds Instance -> Application, hands the --plugin string to stm::load_plugins, then checks what the entry point left
// in the graph. Exit 0 = the renderer component exists under the plugin's node and is subscribed to OnUpdate;
// exit 3 = the library loaded and the entry point ran but no HIP device exists (sthip_create failed: the product has no
// CPU backend) — the outcome a box without a GPU must produce; anything else is a failure of the boundary.
#include <cstdio>
#include <iostream>

#include "../../stratum_amd/host/stratum_hip.hpp"

using namespace stm;

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: plugin_host '<lib>;<fn>[;<fn>...]'\n");
    return 2;
  }
  NodeGraph graph;
  Node& root = graph.emplace("Instance");
  auto app = root.make_child("Application").make_component<Application>();
  try {
    load_plugins(argv[1], app.node());
  } catch (const std::invalid_argument& e) {
    std::printf("MISSING SYMBOL: %s\n", e.what());
    return 4;
  } catch (const std::runtime_error& e) {
    const std::string what = e.what();
    if (what.find("sthip_create") != std::string::npos) {
      std::printf("ENTRY RAN, NO DEVICE: %s\n", e.what());
      return 3;
    }
    std::printf("LOAD FAILED: %s\n", e.what());
    return 5;
  }
  // the entry point was called with the plugin's own child node under Application (main.cpp:16,22)
  component_ptr<BDPT> renderer;
  component_ptr<dynamic_library> lib;
  app.node().for_each_descendant<BDPT>([&](const component_ptr<BDPT>& c) { renderer = c; });
  app.node().for_each_descendant<dynamic_library>([&](const component_ptr<dynamic_library>& c) { lib = c; });
  if (!lib || !renderer) {
    std::printf("FAIL: the plugin did not leave a renderer in the graph\n");
    return 1;
  }
  if (&renderer.node() != &lib.node()) {
    std::printf("FAIL: the renderer is not on the plugin's node\n");
    return 1;
  }
  if (app->OnUpdate.empty()) {
    std::printf("FAIL: the renderer did not subscribe to Application::OnUpdate\n");
    return 1;
  }
  std::printf("PLUGIN OK: %s carries BDPT, OnUpdate subscribed\n", lib.node().name().c_str());
  return 0;
}
