// plugin.cpp — the entry point Stratum's plugin loader calls (--plugin=libstratum_hip_plugin.so;stratum_hip_register,
// src/main.cpp:11-24,148-149; src/Common/dynamic_library.hpp:41-59): `void fn(stm::Node&)` on the main thread at
// start-up, with the plugin's own child node under Application. It installs the HIP renderer component.
#include "stratum_hip.hpp"

extern "C" void stratum_hip_register(stm::Node& node) {
  auto renderer = node.make_component<stm::BDPT>();
  (void)renderer;
}
