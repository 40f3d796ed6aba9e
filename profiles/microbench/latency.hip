// Dependent-operation latencies on gfx950 (one wave unless stated), in core clocks (s_memtime).
//   hipcc -O3 --offload-arch=gfx950 latency.hip -o latency && ./latency
// Calibrates the model behind the supernodal LDLT design: what a dependent f64
// operation, a v_readlane broadcast, an LDS round trip and a workgroup barrier cost.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

constexpr int N = 256;

__global__ void k_fma(double* out, long long* clk, double a, double b) {
  double x = out[threadIdx.x];
  asm volatile("" ::: "memory");
  long long t0 = clock64();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; ++i) x = __builtin_fma(x, a, b);
  asm volatile("" :: "v"(x) : "memory");
  long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) clk[0] = t1 - t0;
}
__global__ void k_fma_indep(double* out, long long* clk, double a, double b) {
  double x0 = out[threadIdx.x], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  asm volatile("" ::: "memory");
  long long t0 = clock64();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < N / 4; ++i) {
    x0 = __builtin_fma(x0, a, b);
    x1 = __builtin_fma(x1, a, b);
    x2 = __builtin_fma(x2, a, b);
    x3 = __builtin_fma(x3, a, b);
  }
  asm volatile("" :: "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "memory");
  long long t1 = clock64();
  out[threadIdx.x] = x0 + x1 + x2 + x3;
  if (threadIdx.x == 0) clk[0] = t1 - t0;
}
__global__ void k_rcp(double* out, long long* clk) {
  double x = out[threadIdx.x];
  asm volatile("" ::: "memory");
  long long t0 = clock64();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rcp(x);
  asm volatile("" :: "v"(x) : "memory");
  long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) clk[0] = t1 - t0;
}
__global__ void k_readlane_fma(double* out, long long* clk, double a) {
  double x = out[threadIdx.x];
  asm volatile("" ::: "memory");
  long long t0 = clock64();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; ++i) x = __builtin_fma(readlane_f64(x, i & 63), a, x);
  asm volatile("" :: "v"(x) : "memory");
  long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) clk[0] = t1 - t0;
}
__global__ void k_fma_f32(float* out, long long* clk, float a, float b) {
  float x = out[threadIdx.x];
  asm volatile("" ::: "memory");
  long long t0 = clock64();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; ++i) x = __builtin_fmaf(x, a, b);
  asm volatile("" :: "v"(x) : "memory");
  long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) clk[0] = t1 - t0;
}
__global__ void k_lds(double* out, long long* clk) {
  __shared__ unsigned int idx[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) idx[i] = (i * 37 + 11) & 1023;
  __syncthreads();
  unsigned int p = threadIdx.x;
  asm volatile("" ::: "memory");
  long long t0 = clock64();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; ++i) p = idx[p];
  asm volatile("" :: "v"(p) : "memory");
  long long t1 = clock64();
  out[threadIdx.x] = p;
  if (threadIdx.x == 0) clk[0] = t1 - t0;
}
__global__ void k_lds_rw(double* out, long long* clk) {  // store -> load of another lane's value (wave-local)
  __shared__ double buf[64];
  double x = out[threadIdx.x];
  buf[threadIdx.x] = x;
  asm volatile("" ::: "memory");
  long long t0 = clock64();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; ++i) {
    buf[threadIdx.x] = x;
    x = buf[(threadIdx.x + 1) & 63] + 1.0;
  }
  asm volatile("" :: "v"(x) : "memory");
  long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) clk[0] = t1 - t0;
}
__global__ void k_barrier(double* out, long long* clk) {
  asm volatile("" ::: "memory");
  long long t0 = clock64();
  asm volatile("" ::: "memory");
#pragma unroll 1
  for (int i = 0; i < N; ++i) __syncthreads();
  long long t1 = clock64();
  if (threadIdx.x == 0) clk[0] = t1 - t0;
  out[threadIdx.x] = 0;
}
__global__ void k_dpp_sum(double* out, long long* clk) {
  double x = out[threadIdx.x];
  asm volatile("" ::: "memory");
  long long t0 = clock64();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0xB1, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0xB1, 0xf, 0xf, true);
    x += __hiloint2double(hi, lo);
  }
  asm volatile("" :: "v"(x) : "memory");
  long long t1 = clock64();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) clk[0] = t1 - t0;
}

int main() {
  double* out;
  long long* clk;
  hipMalloc(&out, 1024 * sizeof(double));
  hipMalloc(&clk, sizeof(long long));
  hipMemset(out, 0, 1024 * sizeof(double));
  auto report = [&](const char* name, int n) {
    long long c = 0;
    hipMemcpy(&c, clk, sizeof(c), hipMemcpyDeviceToHost);
    std::printf("%-46s %8.1f clocks each\n", name, double(c) / n);
  };
  for (int rep = 0; rep < 2; ++rep) {  // second pass: code warm
    k_fma<<<1, 64>>>(out, clk, 1.0000001, 1e-9); hipDeviceSynchronize(); if (rep) report("v_fma_f64 dependent chain", N);
    k_fma_indep<<<1, 64>>>(out, clk, 1.0000001, 1e-9); hipDeviceSynchronize(); if (rep) report("v_fma_f64, 4 independent chains (per instr)", N);
    k_fma<<<1, 1024>>>(out, clk, 1.0000001, 1e-9); hipDeviceSynchronize(); if (rep) report("v_fma_f64 dependent, 16 waves on the CU", N);
    k_fma_f32<<<1, 64>>>((float*)out, clk, 1.0000001f, 1e-9f); hipDeviceSynchronize(); if (rep) report("v_fma_f32 dependent chain", N);
    hipMemset(out, 0x3f, 1024 * sizeof(double));
    k_rcp<<<1, 64>>>(out, clk); hipDeviceSynchronize(); if (rep) report("v_rcp_f64 dependent chain", N);
    k_readlane_fma<<<1, 64>>>(out, clk, 1e-9); hipDeviceSynchronize(); if (rep) report("v_readlane x2 -> v_fma_f64 dependent", N);
    k_dpp_sum<<<1, 64>>>(out, clk); hipDeviceSynchronize(); if (rep) report("2 x v_mov_dpp -> v_add_f64 dependent", N);
    k_lds<<<1, 64>>>(out, clk); hipDeviceSynchronize(); if (rep) report("LDS pointer chase (ds_read_b32 -> address)", N);
    k_lds_rw<<<1, 64>>>(out, clk); hipDeviceSynchronize(); if (rep) report("LDS store -> load other lane -> add", N);
    k_barrier<<<1, 64>>>(out, clk); hipDeviceSynchronize(); if (rep) report("__syncthreads, 1 wave", N);
    k_barrier<<<1, 256>>>(out, clk); hipDeviceSynchronize(); if (rep) report("__syncthreads, 4 waves", N);
    k_barrier<<<1, 1024>>>(out, clk); hipDeviceSynchronize(); if (rep) report("__syncthreads, 16 waves", N);
  }
  return 0;
}
