// From a kernel's result in pinned host memory to the start of the next kernel the host launches because of it —
// the gap an interior-point iteration pays 1.4 times: kernel A spins ~20 us and leaves
// its end time in pinned memory, the host spins on that word and launches B, whose first lane leaves its start time.
// Variants: B with 16 / 256 / 768 bytes of by-value arguments; B launched ahead of A's end in the same stream (the
// in-order floor); B launched on the host's signal but gated — already resident, spinning on a pinned word.
//   hipcc -O3 --offload-arch=gfx950 -o launch_gap_bin launch_gap.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                        \
  do {                                                                  \
    hipError_t e_ = (x);                                                \
    if (e_ != hipSuccess) {                                             \
      std::printf("%s: %s\n", #x, hipGetErrorString(e_));               \
      return 1;                                                         \
    }                                                                   \
  } while (0)

struct Big256 { unsigned long long w[32]; };
struct Big768 { unsigned long long w[96]; };

__global__ void kernel_a(volatile unsigned long long* host_end, unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    __threadfence_system();
    *host_end = wall_clock64();
  }
}
__global__ void kernel_b16(unsigned long long* start, unsigned long long) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *start = wall_clock64();
}
__global__ void kernel_b256(unsigned long long* start, Big256 a) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *start = wall_clock64() + (a.w[31] & 0);
}
__global__ void kernel_b768(unsigned long long* start, Big768 a) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *start = wall_clock64() + (a.w[95] & 0);
}
// the same pair with workgroups that carry LDS like the step kernel's (57-160 KB of dynamic LDS, 1024 threads)
__global__ void kernel_a_lds(volatile unsigned long long* host_end, unsigned long long ticks) {
  extern __shared__ unsigned char lds[];
  lds[threadIdx.x] = 1;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    __threadfence_system();
    *host_end = wall_clock64() + (lds[5] & 0);
  }
}
__global__ void kernel_b_lds(unsigned long long* start, unsigned long long) {
  extern __shared__ unsigned char lds[];
  lds[threadIdx.x] = 1;
  if (threadIdx.x == 0 && blockIdx.x == 0) *start = wall_clock64() + (lds[0] & 0);
}
// resident before the host decides: leaves when the host opens the gate (a pinned word), then stamps
template <int SLEEP>
__global__ void kernel_gated(unsigned long long* start, volatile unsigned long long* gate, unsigned long long ticket) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(const_cast<unsigned long long*>(gate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < ticket && ++spins < (1u << 22))
      __builtin_amdgcn_s_sleep(SLEEP);
    if (blockIdx.x == gridDim.x - 1) *start = wall_clock64();
  }
}
// one workgroup asks the host, the others ask a word in device memory that workgroup passes the answer to
__global__ void kernel_gated_relay(unsigned long long* start, volatile unsigned long long* gate, unsigned long long ticket,
                                   unsigned long long* relay) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    if (blockIdx.x == 0) {
      while (__hip_atomic_load(const_cast<unsigned long long*>(gate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < ticket && ++spins < (1u << 22))
        __builtin_amdgcn_s_sleep(2);
      __hip_atomic_store(relay, ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(relay, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ticket && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(2);
    }
    if (blockIdx.x == gridDim.x - 1) *start = wall_clock64();
  }
}

int main() {
  unsigned long long *h_end = nullptr, *h_gate = nullptr, *d_start = nullptr;
  CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_end), 64));
  CHECK(hipHostMalloc(reinterpret_cast<void**>(&h_gate), 64));
  CHECK(hipMalloc(reinterpret_cast<void**>(&d_start), 64));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const unsigned long long ticks = 2000;  // 20 us at 100 MHz
  Big256 a256{};
  Big768 a768{};
  auto stats = [](std::vector<double>& v) {
    std::sort(v.begin(), v.end());
    std::printf("median %.2f us, min %.2f, p90 %.2f\n", v[v.size() / 2], v.front(), v[v.size() * 9 / 10]);
  };
  unsigned long long* d_relay = nullptr;
  CHECK(hipMalloc(reinterpret_cast<void**>(&d_relay), 64));
  CHECK(hipMemset(d_relay, 0, 64));
  unsigned long long ticket = 0;
  // (variants 10-12: B enqueued while A — 60 us here — is already running, 0 / 15 / 35 us after A's launch call returned)
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel_a_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel_b_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  // (variants 13-18: LDS-carrying workgroups on one or both sides, B enqueued at once / 25 us into A)
  for (int variant = 0; variant < 19; ++variant) {
    std::vector<double> gaps;
    for (int it = 0; it < 300; ++it) {
      *h_end = 0;
      unsigned long long start = 0;
      if (variant >= 13) {
        const int k = variant - 13;  // 0,1: A lds, B small; 2,3: A small, B lds; 4,5: both lds
        const bool a_lds = k < 2 || k >= 4, b_lds = k >= 2, late = (k & 1) != 0;
        if (a_lds) hipLaunchKernelGGL(kernel_a_lds, dim3(137), dim3(1024), 100 * 1024, s, h_end, 5000ull);
        else hipLaunchKernelGGL(kernel_a, dim3(1), dim3(64), 0, s, h_end, 5000ull);
        const auto t0 = std::chrono::steady_clock::now();
        while (late && std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 25.0) {
        }
        if (b_lds) hipLaunchKernelGGL(kernel_b_lds, dim3(137), dim3(1024), 100 * 1024, s, d_start, 0ull);
        else hipLaunchKernelGGL(kernel_b16, dim3(137), dim3(1024), 0, s, d_start, 0ull);
      } else if (variant >= 10) {
        hipLaunchKernelGGL(kernel_a, dim3(1), dim3(64), 0, s, h_end, 6000ull);
        const auto t0 = std::chrono::steady_clock::now();
        const double wait_us = variant == 10 ? 0.0 : variant == 11 ? 15.0 : 35.0;
        while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < wait_us) {
        }
        hipLaunchKernelGGL(kernel_b16, dim3(137), dim3(1024), 0, s, d_start, 0ull);
      } else if (variant >= 4) {
        // gated: B is launched right behind A and waits for the host's word
        ++ticket;
        hipLaunchKernelGGL(kernel_a, dim3(1), dim3(64), 0, s, h_end, ticks);
        const int grid = (variant == 4 || variant == 6 || variant == 8) ? 137 : 30;
        if (variant <= 5) hipLaunchKernelGGL(kernel_gated<4>, dim3(grid), dim3(1024), 0, s, d_start, h_gate, ticket);
        else if (variant <= 7) hipLaunchKernelGGL(kernel_gated<32>, dim3(grid), dim3(1024), 0, s, d_start, h_gate, ticket);
        else hipLaunchKernelGGL(kernel_gated_relay, dim3(grid), dim3(1024), 0, s, d_start, h_gate, ticket, d_relay);
        while (*reinterpret_cast<volatile unsigned long long*>(h_end) == 0) {
        }
        *reinterpret_cast<volatile unsigned long long*>(h_gate) = ticket;
      } else if (variant == 3) {
        hipLaunchKernelGGL(kernel_a, dim3(1), dim3(64), 0, s, h_end, ticks);
        hipLaunchKernelGGL(kernel_b16, dim3(137), dim3(1024), 0, s, d_start, 0ull);
      } else {
        hipLaunchKernelGGL(kernel_a, dim3(1), dim3(64), 0, s, h_end, ticks);
        while (*reinterpret_cast<volatile unsigned long long*>(h_end) == 0) {
        }
        if (variant == 0) hipLaunchKernelGGL(kernel_b16, dim3(137), dim3(1024), 0, s, d_start, 0ull);
        if (variant == 1) hipLaunchKernelGGL(kernel_b256, dim3(137), dim3(1024), 0, s, d_start, a256);
        if (variant == 2) hipLaunchKernelGGL(kernel_b768, dim3(137), dim3(1024), 0, s, d_start, a768);
      }
      CHECK(hipStreamSynchronize(s));
      CHECK(hipMemcpy(&start, d_start, 8, hipMemcpyDeviceToHost));
      if (it >= 20) gaps.push_back(static_cast<double>(static_cast<long long>(start - *h_end)) / 100.0);
    }
    const char* names[] = {"host sees A's result, launches B (16 bytes of arguments)", "... B with 256 bytes of arguments",
                           "... B with 768 bytes of arguments", "B launched behind A at once (in-order floor)",
                           "B launched behind A at once, gated on a pinned word the host writes when it sees A's result (137 workgroups ask, s_sleep 4)",
                           "... 30 workgroups ask, s_sleep 4", "... 137 workgroups ask, s_sleep 32", "... 30 workgroups ask, s_sleep 32",
                           "... one of 137 workgroups asks and passes the answer on through device memory", "... one of 30 workgroups asks and passes it on",
                           "B enqueued right behind A's launch call (A runs 60 us)", "B enqueued 15 us after A's launch call", "B enqueued 35 us after A's launch call",
                           "A: 137 workgroups with 100 KB of LDS, 50 us; B without LDS, enqueued at once", "... B enqueued 25 us into A",
                           "A without LDS; B: 137 workgroups with 100 KB of LDS, enqueued at once", "... B enqueued 25 us into A",
                           "A and B with 100 KB of LDS, B enqueued at once", "... B enqueued 25 us into A"};
    std::printf("%-130s: ", names[variant]);
    stats(gaps);
  }
  return 0;
}
