// sn_finish_wave / chain_solve_wave (csrc/ldlt_kernels.h) in isolation: clocks per call for a
// w-wide chain with nr rows, one wave, LDS-resident trapezoid — the numbers behind the
// supernodal level design.
//   hipcc -O3 -std=c++23 --offload-arch=gfx950 -I../../sleipnir_amd/csrc chain.hip -o chain_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ldlt_kernels.h"

using namespace slpx;

// The rolled alternative that lost (kept here for the record): the row registers are shifted by
// one per column, so the pivot column is always a[0]; bodies of WB - 1 steps.
template <int WB>
__device__ __forceinline__ void sn_finish_rolled(double* __restrict__ U, double* __restrict__ invd, uint32_t base0,
                                                 uint32_t w, uint32_t nr, uint32_t col0, uint32_t lane) {
  const uint32_t t = lane;
  const bool live = t < nr;
  double a[WB];
#pragma unroll
  for (int c = 0; c < WB; ++c) {
    const uint32_t offc = base0 + c * nr - (c * (c - 1)) / 2;
    a[c] = (live && static_cast<uint32_t>(c) < w && static_cast<uint32_t>(c) <= t) ? U[offc + (t - c)] : 0.0;
  }
  uint32_t offc = base0;
#pragma unroll 1
  for (uint32_t c = 0; c < w; ++c) {
    const double piv = a[0];
    const double inv = chain_reciprocal(readlane_f64(piv, static_cast<int>(c)));
    const double lc = piv * inv;
    if (live && t >= c) U[offc + (t - c)] = piv;
    if (lane == c) invd[col0 + c] = inv;
#pragma unroll
    for (int j = 1; j < WB; ++j) a[j - 1] = __builtin_fma(-lc, readlane_f64(piv, static_cast<int>(c) + j), a[j]);
    a[WB - 1] = 0.0;
    offc += nr - c;
  }
}
__device__ __forceinline__ void sn_finish_rolled_any(double* U, double* invd, uint32_t base0, uint32_t w, uint32_t nr,
                                                     uint32_t col0, uint32_t lane) {
  if (w <= 4) sn_finish_rolled<4>(U, invd, base0, w, nr, col0, lane);
  else sn_finish_rolled<8>(U, invd, base0, w, nr, col0, lane);
}

template <int VARIANT>
__global__ void k_chain(long long* clk, uint32_t w, uint32_t nr, int reps, int threads_active) {
  __shared__ double U[4096];
  __shared__ double invd[64];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) U[i] = 1.0 + 0.001 * (i % 97);
  // diagonally dominant block
  uint32_t off = 0;
  for (uint32_t c = 0; c < w; ++c) {
    if (threadIdx.x == 0) U[off] = 10.0 + c;
    off += nr - c;
  }
  __syncthreads();
  const uint32_t ws = __builtin_amdgcn_readfirstlane(w), nrs = __builtin_amdgcn_readfirstlane(nr);
  long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    if ((threadIdx.x >> 6) < threads_active) {
      if (VARIANT == 0) sn_finish_rolled_any(U, invd, 0u, ws, nrs, 0u, threadIdx.x & 63);
      else sn_finish_wave(lds_cast(U), lds_cast(invd), 0u, ws, nrs, 0u, threadIdx.x & 63);
    }
    __syncthreads();
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) clk[0] = (t1 - t0) / reps;
  if (threadIdx.x == 0) clk[1] = static_cast<long long>(U[5] * 1000);
}

int main() {
  long long* clk;
  (void)hipMalloc(&clk, 16 * sizeof(long long));
  const int ws[] = {2, 3, 4, 5, 6, 8};
  for (int variant : {0, 1}) {
    const int threads = 1024;
    for (int w : ws) {
      long long c[2] = {0, 0};
      for (int rep = 0; rep < 2; ++rep) {
        if (variant == 0) k_chain<0><<<1, threads>>>(clk, w, w + 9, 1, 1); else k_chain<1><<<1, threads>>>(clk, w, w + 9, 1, 1);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(&c[0], clk, sizeof(long long), hipMemcpyDeviceToHost);
        if (variant == 0) k_chain<0><<<1, threads>>>(clk, w, w + 9, 16, 1); else k_chain<1><<<1, threads>>>(clk, w, w + 9, 16, 1);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(&c[1], clk, sizeof(long long), hipMemcpyDeviceToHost);
      }
      std::printf("%s workgroup %4d threads, w = %2d, nr = %2d: first call %6lld clocks, average of 16 calls %6lld (incl. one barrier)\n", variant ? "exact-width code" : "rolled, shifted registers", threads, w,
                  w + 9, c[0], c[1]);
    }
  }
  return 0;
}
