// Cost of executing straight-line code for the first time on a CU (instruction-cache misses),
// gfx950.  One wave runs a dependent chain of N v_fma_f64 (8 bytes each): N = 256 (2 KB of
// code) ... 8192 (64 KB); each kernel is launched cold (after a different big kernel ran) and
// again right after itself.
//   hipcc -O3 --offload-arch=gfx950 icache.hip -o icache_bin && ./icache_bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int N>
__global__ void chain(double* out, long long* clk, double a, double b) {
  double x = out[threadIdx.x];
  asm volatile("" ::: "memory");
  long long t0 = clock64();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; ++i) x = __builtin_fma(x, a, b);
  asm volatile("" ::"v"(x) : "memory");
  long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int N>
void run(double* out, long long* clk, const char* name) {
  long long c[2][2];
  for (int rep = 0; rep < 2; ++rep) {
    // evict: a different 64 KB kernel on every CU
    chain<8192 + N><<<512, 64>>>(out, clk, 1.0, 0.0);
    (void)hipDeviceSynchronize();
    chain<N><<<1, 64>>>(out, clk, 1.0000001, 1e-9);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&c[rep][0], clk, sizeof(long long), hipMemcpyDeviceToHost);
    chain<N><<<1, 64>>>(out, clk, 1.0000001, 1e-9);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&c[rep][1], clk, sizeof(long long), hipMemcpyDeviceToHost);
  }
  std::printf("%-10s %5d instr (%3d KB): after another kernel %7lld clocks (%.1f / instr), relaunched at once %7lld (%.1f / instr)\n", name, N,
              N * 8 / 1024, c[1][0], double(c[1][0]) / N, c[1][1], double(c[1][1]) / N);
}

int main() {
  double* out;
  long long* clk;
  (void)hipMalloc(&out, 1024 * sizeof(double));
  (void)hipMalloc(&clk, 1024 * sizeof(long long));
  (void)hipMemset(out, 0, 1024 * sizeof(double));
  run<256>(out, clk, "chain");
  run<1024>(out, clk, "chain");
  run<4096>(out, clk, "chain");
  return 0;
}
