// SLPX_SETUP_TIMING=1: seconds per setup phase on stderr (setup = everything that happens
// once per sparsity pattern: AD structure, tape compilation, KKT plan, symbolic LDLT, upload).
#pragma once

#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace slpx {

struct SetupLap {
  bool on = std::getenv("SLPX_SETUP_TIMING") != nullptr;
  std::chrono::steady_clock::time_point prev = std::chrono::steady_clock::now();
  void operator()(const char* what) {
    const auto now = std::chrono::steady_clock::now();
    if (on) std::fprintf(stderr, "slpx setup: %-36s %.4f s\n", what, std::chrono::duration<double>(now - prev).count());
    prev = now;
  }
};

}  // namespace slpx
