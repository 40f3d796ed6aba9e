// mf_pivots / mf_schur (csrc/ldlt_mf_kernels.h) in isolation: one dense front per wave — assembly
// from two children's update blocks, w pivots in registers, Schur complement by
// v_mfma_f64_16x16x4_f64 — checked against a plain host elimination and timed in core clocks.
//   hipcc -O3 -std=c++23 --offload-arch=gfx950 -I../../sleipnir_amd/csrc front.hip -o front_bin
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "ldlt_mf_kernels.h"

using namespace slpx;

constexpr int kWaves = 8;
constexpr uint32_t kUStride = 384, kSIn = 1201, kSOut = 1024, kSTotal = kSIn + kWaves * kSOut, kGStride = 8192;

// per wave: its own front (same shape), its own children blocks
__global__ __launch_bounds__(512) void k_front(long long* clk, const double* U0, const double* S0, const uint16_t* G0,
                                               double* Uout, double* Sout, double* invd_out, uint32_t w, uint32_t nr,
                                               uint32_t nch, uint32_t n_dest, uint32_t s_off, int reps, int waves, uint32_t shift, int mode = 3) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* U = reinterpret_cast<double*>(smem + shift);  // (a run-time offset: the callees must not see compile-time LDS addresses)
  double* S = U + kWaves * kUStride;  // [zero | two children's blocks, shared | one output block per wave]
  double* invd = S + kSTotal;
  uint16_t* G = reinterpret_cast<uint16_t*>(invd + kWaves * 16);
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (uint32_t i = threadIdx.x; i < kGStride; i += blockDim.x) G[i] = G0[i];
  long long total = 0;
  for (int rep = 0; rep < reps; ++rep) {
    for (uint32_t i = threadIdx.x; i < kWaves * kUStride; i += blockDim.x) U[i] = U0[i % kUStride];
    for (uint32_t i = threadIdx.x; i < kSTotal; i += blockDim.x) S[i] = i < kSIn ? S0[i] : 0.0;
    __syncthreads();
    const long long t0 = clock64();
    if (wave < static_cast<uint32_t>(waves) && w != 0) {
      if (mode & 1) mf_pivots(lds_cast(U + wave * kUStride), lds_cast(invd + wave * 16), lds_cast(static_cast<const double*>(S)),
                lds_cast(static_cast<const uint16_t*>(G)), 0u, w, nr, 0u, 0u, nch, n_dest, lane);
      if (mode & 2) mf_schur(lds_cast(static_cast<const double*>(U + wave * kUStride)), lds_cast(static_cast<const double*>(invd + wave * 16)),
               lds_cast(S), lds_cast(static_cast<const uint16_t*>(G)), 0u, w, nr, 0u, 0u, nch, n_dest, s_off + wave * kSOut,
               nullptr, nullptr, lane);
    }
    __syncthreads();
    total += clock64() - t0;
  }
  if (threadIdx.x == 0) clk[0] = total / reps;
  for (uint32_t i = threadIdx.x; i < kUStride; i += blockDim.x) Uout[i] = U[i];
  for (uint32_t i = threadIdx.x; i < kSIn + kSOut; i += blockDim.x) Sout[i] = S[i];
  if (threadIdx.x < 16) invd_out[threadIdx.x] = invd[threadIdx.x];
}

static uint32_t col_off(uint32_t c, uint32_t nr) { return c * nr - (c * (c - 1)) / 2; }

int main() {
  long long* clk;
  double *dU, *dS, *dUo, *dSo, *dInv;
  uint16_t* dG;
  (void)hipMalloc(&clk, 64);
  (void)hipMalloc(&dU, kUStride * 8);
  (void)hipMalloc(&dS, (kSIn + kSOut) * 8);
  (void)hipMalloc(&dUo, kUStride * 8);
  (void)hipMalloc(&dSo, (kSIn + kSOut) * 8);
  (void)hipMalloc(&dInv, 16 * 8);
  (void)hipMalloc(&dG, kGStride * 2);
  const size_t lds = (kWaves * kUStride + kSTotal + kWaves * 16) * 8 + kGStride * 2;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_front), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  struct Case { uint32_t w, r, nch; };
  const Case cases[] = {{1, 6, 0}, {1, 6, 2}, {2, 8, 2}, {4, 9, 2}, {5, 9, 2}, {8, 10, 2}, {4, 20, 2}, {8, 36, 2}};
  {
    long long c0 = 0;
    k_front<<<1, 512, lds>>>(clk, dU, dS, dG, dUo, dSo, dInv, 0u, 8u, 0u, 64u, kSIn, 16, 1, 0u);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&c0, clk, 8, hipMemcpyDeviceToHost);
    std::printf("empty body (two clock readings + one barrier of 8 waves): %lld clocks\n", c0);
  }
  for (const Case& cs : cases) {
    const uint32_t w = cs.w, r = cs.r, nr = w + r + 1, nch = cs.nch;
    const uint32_t n_tr = col_off(w, nr), n_s = r * (r + 1) / 2 + r, n_dest = n_tr + n_s;
    // children's packed blocks: child k at 1 + k * 600 (index 0 = the arena's zero), this front's S after them
    const uint32_t s_off = kSIn;
    if (n_s > kSOut || nch * n_dest > kGStride || n_tr > kUStride) {
      std::printf("case w=%u r=%u does not fit the harness\n", w, r);
      continue;
    }
    std::vector<double> U(kUStride, 0.0), S(kSIn + kSOut, 0.0);
    std::vector<uint16_t> G(kGStride, 0);
    uint32_t seed = 12345u + 977u * w + 31u * r;
    auto rnd = [&] {
      seed = seed * 1664525u + 1013904223u;
      return (static_cast<double>(seed >> 8) / 16777216.0) - 0.5;
    };
    for (uint32_t c = 0; c < w; ++c)
      for (uint32_t t = c; t < nr; ++t) U[col_off(c, nr) + (t - c)] = (t == c ? 6.0 + c : 0.0) + rnd();
    for (uint32_t k = 0; k < 2; ++k)
      for (uint32_t i = 0; i < 600; ++i) S[1 + k * 600 + i] = 0.1 * rnd();
    // every destination takes source (d * 7 + 13 k) mod 600 of child k — asymmetric on purpose
    for (uint32_t k = 0; k < nch; ++k)
      for (uint32_t d = 0; d < n_dest; ++d) G[k * n_dest + d] = static_cast<uint16_t>(1 + k * 600 + (d * 7 + 13 * k) % 600);
    // host reference
    std::vector<double> F(nr * nr, 0.0);  // full front, lower part: F[a * nr + b], a >= b
    for (uint32_t c = 0; c < w; ++c)
      for (uint32_t t = c; t < nr; ++t) {
        double v = U[col_off(c, nr) + (t - c)];
        for (uint32_t k = 0; k < nch; ++k) v += S[G[k * n_dest + col_off(c, nr) + (t - c)]];
        F[t * nr + c] = v;
      }
    for (uint32_t a = 0; a <= r; ++a)
      for (uint32_t b = 0; b < r && b <= a; ++b) {
        double v = 0.0;
        for (uint32_t k = 0; k < nch; ++k) v += S[G[k * n_dest + n_tr + a * (a + 1) / 2 + b]];
        F[(w + a) * nr + (w + b)] = v;
      }
    std::vector<double> inv_ref(w);
    for (uint32_t c = 0; c < w; ++c) {
      const double inv = 1.0 / F[c * nr + c];
      inv_ref[c] = inv;
      for (uint32_t t = c + 1; t < nr; ++t) {
        const double l = F[t * nr + c] * inv;
        for (uint32_t j = c + 1; j <= t && j < nr - 1; ++j) F[t * nr + j] -= l * F[j * nr + c];
      }
    }
    (void)hipMemcpy(dU, U.data(), kUStride * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dS, S.data(), (kSIn + kSOut) * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dG, G.data(), kGStride * 2, hipMemcpyHostToDevice);
    long long c1 = 0, c16 = 0, cold = 0;
    for (int pass = 0; pass < 2; ++pass) {
      k_front<<<1, 512, lds>>>(clk, dU, dS, dG, dUo, dSo, dInv, w, nr, nch, n_dest, s_off, 1, 1, 0u);
      (void)hipDeviceSynchronize();
      if (pass == 0) (void)hipMemcpy(&cold, clk, 8, hipMemcpyDeviceToHost);
      k_front<<<1, 512, lds>>>(clk, dU, dS, dG, dUo, dSo, dInv, w, nr, nch, n_dest, s_off, 16, 1, 0u);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(&c1, clk, 8, hipMemcpyDeviceToHost);
      k_front<<<1, 512, lds>>>(clk, dU, dS, dG, dUo, dSo, dInv, w, nr, nch, n_dest, s_off, 16, kWaves, 0u);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(&c16, clk, 8, hipMemcpyDeviceToHost);
    }
    long long cp = 0, csch = 0;
    k_front<<<1, 512, lds>>>(clk, dU, dS, dG, dUo, dSo, dInv, w, nr, nch, n_dest, s_off, 16, 1, 0u, 1);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&cp, clk, 8, hipMemcpyDeviceToHost);
    k_front<<<1, 512, lds>>>(clk, dU, dS, dG, dUo, dSo, dInv, w, nr, nch, n_dest, s_off, 16, 1, 0u, 2);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&csch, clk, 8, hipMemcpyDeviceToHost);
    k_front<<<1, 512, lds>>>(clk, dU, dS, dG, dUo, dSo, dInv, w, nr, nch, n_dest, s_off, 1, 1, 0u);
    (void)hipDeviceSynchronize();
    hipError_t err = hipGetLastError();
    std::vector<double> Uo(kUStride), So(kSIn + kSOut), io(16);
    (void)hipMemcpy(Uo.data(), dUo, kUStride * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(So.data(), dSo, (kSIn + kSOut) * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(io.data(), dInv, 16 * 8, hipMemcpyDeviceToHost);
    double eu = 0, es = 0, ei = 0;
    for (uint32_t c = 0; c < w; ++c) {
      ei = std::fmax(ei, std::fabs(io[c] - inv_ref[c]) / std::fabs(inv_ref[c]));
      for (uint32_t t = c; t < nr; ++t) eu = std::fmax(eu, std::fabs(Uo[col_off(c, nr) + (t - c)] - F[t * nr + c]));
    }
    for (uint32_t a = 0; a <= r; ++a)
      for (uint32_t b = 0; b < r && b <= a; ++b)
        es = std::fmax(es, std::fabs(So[s_off + a * (a + 1) / 2 + b] - F[(w + a) * nr + (w + b)]));
    std::printf("front w = %u, r = %2u (nr = %2u), %u children: first call %6lld clocks; warm, one wave %5lld; eight waves at once %5lld "
                "(each incl. one barrier); pivots alone %5lld, update block alone %5lld   max error U %.1e  S %.1e  1/d %.1e %s\n",
                w, r, nr, nch, cold, c1, c16, cp, csch, eu, es, ei, err == hipSuccess ? "" : hipGetErrorString(err));
  }
  return 0;
}
