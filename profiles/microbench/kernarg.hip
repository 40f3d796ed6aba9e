// What the FIRST use of a kernel argument costs a wave on gfx950, in wall-clock ticks of 10 ns
// (s_memrealtime): the kernarg segment is written by the host for every launch, so its first
// read is never cached.  Variants:
//   value    : a 4-byte by-value argument (first line of the kernarg segment)
//   big      : the LAST word of a 768-byte by-value struct (what a generated tape kernel with its
//              model numbers passed by value did: tape_jit.cpp)
//   pointer  : a device pointer argument, then one word behind it (kernarg -> device memory)
//   preload  : the same with the pointer preloaded into SGPRs by the command processor
//              (-mllvm -amdgpu-kernarg-preload-count=2), where the toolchain / firmware do it
//   hipcc -O3 --offload-arch=gfx950 kernarg.hip -o kernarg_bin && ./kernarg_bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

struct Big {
  double c[64];
  unsigned w[64];
};

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }

#define MEASURE(expr)                                                  \
  const unsigned long long t0 = now();                                 \
  asm volatile("" ::: "memory");                                       \
  const unsigned v = (expr);                                           \
  asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(v) : "memory");            \
  const unsigned long long t1 = now();                                 \
  if (threadIdx.x == 0) out[blockIdx.x] = (t1 - t0) | (static_cast<unsigned long long>(v & 1u) << 63)

__global__ __launch_bounds__(64) void k_value(unsigned a, unsigned long long* out) { MEASURE(a); }
__global__ __launch_bounds__(64) void k_big(Big b, unsigned long long* out) { MEASURE(b.w[63]); }
__global__ __launch_bounds__(64) void k_pointer(const unsigned* __restrict__ p, unsigned long long* out) {
  MEASURE(p[0]);
}

template <typename Launch>
void run(const char* name, int blocks, unsigned long long* d_out, Launch&& launch) {
  std::vector<unsigned long long> h(blocks);
  std::vector<double> med;
  double worst = 0;
  for (int rep = 0; rep < 20; ++rep) {
    launch();
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d_out, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    for (auto& x : h) x &= ~(1ull << 63);
    std::sort(h.begin(), h.end());
    if (rep >= 4) {
      med.push_back(10.0 * h[blocks / 2]);
      worst = std::max(worst, 10.0 * h[blocks - 1]);
    }
  }
  std::sort(med.begin(), med.end());
  std::printf("%-28s %4d workgroups: median of the waves %6.0f ns, slowest wave %6.0f ns\n", name, blocks,
              med[med.size() / 2], worst);
}

int main() {
  unsigned long long* d_out = nullptr;
  unsigned* d_p = nullptr;
  hipMalloc(&d_out, 4096 * sizeof(unsigned long long));
  hipMalloc(&d_p, 4096);
  hipMemset(d_p, 0, 4096);
  Big b{};
  for (int blocks : {1, 256, 2048}) {
    run("value", blocks, d_out, [&] { hipLaunchKernelGGL(k_value, dim3(blocks), dim3(64), 0, 0, 3u, d_out); });
    run("last word of 768 B by value", blocks, d_out, [&] { hipLaunchKernelGGL(k_big, dim3(blocks), dim3(64), 0, 0, b, d_out); });
    run(PRELOAD_NAME, blocks, d_out, [&] { hipLaunchKernelGGL(k_pointer, dim3(blocks), dim3(64), 0, 0, d_p, d_out); });
  }
  return 0;
}
