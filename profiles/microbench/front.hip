// mf_front / mf_solve_front (csrc/ldlt_mf_kernels.h) in isolation: one dense front per wave —
// assembly from the children's update blocks through the front's tables, w pivots in registers,
// the update block, and the backward solve of the same front — checked against a plain host
// elimination and timed in core clocks (clock64).
//   make -C profiles/microbench front_bin
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "ldlt_mf_kernels.h"

using namespace slpx;

constexpr int kWaves = 8;
// LDS image (bytes): [U 8 x 400 doubles | arena 2 + 1300 + 8 x 300 | invd 8 x 8 | x 8 x 80] then the tables
constexpr uint32_t kU = 400, kArena = 2 + 1300 + kWaves * 300, kX = 80, kTabWords = 12288;
constexpr uint32_t oArena = 8 * kWaves * kU, oInvd = oArena + 8 * kArena, oX = oInvd + 8 * kWaves * 8, oTab = oX + 8 * kWaves * kX;

struct Shape {
  uint32_t w, nr, nch, n_s;
};

// per wave: its own front (same shape): U block, output block, invd, x; the children's blocks are shared
__global__ __launch_bounds__(512) void k_front(long long* clk, const double* img0, const uint16_t* tab0, double* img_out, Shape sh,
                                               int reps, int waves, int mode, uint32_t shift) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* img = reinterpret_cast<double*>(smem + shift);
  uint16_t* tab = reinterpret_cast<uint16_t*>(smem + shift + oTab);
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (uint32_t i = threadIdx.x; i < kTabWords; i += blockDim.x) tab[i] = tab0[i];
  long long total = 0;
  const uint32_t words = sh.nr * (1 + sh.nch) * sh.w + sh.n_s * (3 + sh.nch) + (sh.nr - sh.w - 1);  // one wave's tables
  for (int rep = 0; rep < reps; ++rep) {
    for (uint32_t i = threadIdx.x; i < oTab / 8; i += blockDim.x) img[i] = img0[i];
    __syncthreads();
    const long long t0 = clock64();
    if (wave < static_cast<uint32_t>(waves) && sh.w != 0) {
      const uint32_t tb = oTab + 2u * wave * words;
      if (mode & 1) mf_front<false>(tb, sh.w, sh.nr, sh.nch, sh.n_s, 0u, oInvd + 64u * wave, nullptr, nullptr, lane);
      if (mode & 2)
        mf_solve_front(tb + 2u * (sh.nr * (1 + sh.nch) * sh.w + sh.n_s * (3 + sh.nch)), 8u * kU * wave, sh.w, sh.nr,
                       oInvd + 64u * wave, oX + 8u * kX * wave, lane);
    }
    __syncthreads();
    total += clock64() - t0;
  }
  if (threadIdx.x == 0) clk[0] = total / reps;
  for (uint32_t i = threadIdx.x; i < oTab / 8; i += blockDim.x) img_out[i] = img[i];
}

static uint32_t col_off(uint32_t c, uint32_t nr) { return c * nr - (c * (c - 1)) / 2; }

int main() {
  long long* clk;
  double *dImg, *dOut;
  uint16_t* dTab;
  (void)hipMalloc(&clk, 64);
  (void)hipMalloc(&dImg, oTab);
  (void)hipMalloc(&dOut, oTab);
  (void)hipMalloc(&dTab, kTabWords * 2);
  const size_t lds = oTab + kTabWords * 2;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_front), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  auto run = [&](Shape sh, int reps, int waves, int mode) {
    long long c = 0;
    k_front<<<1, 512, lds>>>(clk, dImg, dTab, dOut, sh, reps, waves, mode, 0u);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    return c;
  };
  std::printf("empty body (two clock readings + one barrier of 8 waves): %lld clocks\n", run(Shape{0, 8, 0, 27}, 16, 1, 3));
  struct Case { uint32_t w, r, nch; };
  const Case cases[] = {{1, 6, 0}, {1, 6, 2}, {2, 8, 2}, {3, 8, 2}, {4, 9, 2}, {5, 9, 2}, {6, 9, 2}, {7, 9, 2}, {8, 10, 2}, {4, 9, 3}, {5, 9, 3}, {5, 9, 5}, {6, 9, 3}, {4, 20, 2}, {8, 22, 2}};
  for (const Case& cs : cases) {
    const uint32_t w = cs.w, r = cs.r, nr = w + r + 1, nch = cs.nch;
    const uint32_t n_tr = col_off(w, nr), n_s = r * (r + 1) / 2 + r;
    const uint32_t words = nr * (1 + nch) * w + n_s * (3 + nch) + r;
    if (n_s > 300 || kWaves * words > kTabWords || n_tr > kU || r + w > kX) {
      std::printf("case w=%u r=%u does not fit the harness\n", w, r);
      continue;
    }
    std::vector<double> img(oTab / 8, 0.0);
    std::vector<uint16_t> tab(kTabWords, 0);
    uint32_t seed = 0;
    auto rnd = [&] {
      seed = seed * 1664525u + 1013904223u;
      return (static_cast<double>(seed >> 8) / 16777216.0) - 0.5;
    };
    // every wave's U block holds the same trapezoid; the children's blocks sit at arena[2 .. 1302)
    for (int wv = 0; wv < kWaves; ++wv) {
      seed = 12345u + 977u * w + 31u * r;
      for (uint32_t c = 0; c < w; ++c)
        for (uint32_t t = c; t < nr; ++t) img[kU * wv + col_off(c, nr) + (t - c)] = (t == c ? 6.0 + c : 0.0) + rnd();
      for (uint32_t a = 0; a < r; ++a) img[oX / 8 + kX * wv + w + a] = 0.25 + 0.01 * a;  // x of the rows of R
    }
    for (uint32_t i = 0; i < 1300; ++i) img[oArena / 8 + 2 + i] = 0.1 * rnd();
    // tables, one set per wave (their own U block and output block)
    for (int wv = 0; wv < kWaves; ++wv) {
      uint16_t* T = tab.data() + wv * words;
      const uint32_t u0 = 8 * kU * wv, out0 = oArena + 8 * (2 + 1300 + 300 * wv);
      for (uint32_t row = 0; row < nr; ++row)
        for (uint32_t k = 0; k <= nch; ++k)
          for (uint32_t c = 0; c < w; ++c) {
            if (k == 0) *T++ = row >= c ? static_cast<uint16_t>(u0 + 8 * (col_off(c, nr) + row - c)) : static_cast<uint16_t>(oArena + 8);
            else *T++ = row >= c ? static_cast<uint16_t>(oArena + 8 * (2 + (k - 1) * 650 + ((row * w + c) * 7 + 13 * k) % 650)) : static_cast<uint16_t>(oArena);
          }
      for (uint32_t a = 0; a <= r; ++a)
        for (uint32_t b = 0; b < r && b <= a; ++b) {
          const uint32_t e = a * (a + 1) / 2 + b;
          *T++ = static_cast<uint16_t>(out0 + 8 * e);
          *T++ = static_cast<uint16_t>(u0 + 8 * (w + a));
          *T++ = static_cast<uint16_t>(u0 + 8 * (w + b));
          for (uint32_t k = 0; k < nch; ++k) *T++ = static_cast<uint16_t>(oArena + 8 * (2 + k * 650 + (e * 11 + 5 * k) % 650));
        }
      for (uint32_t a = 0; a < r; ++a) *T++ = static_cast<uint16_t>(oX + 8 * (kX * wv + w + a));
    }
    // host reference on wave 0's data
    auto at = [&](uint16_t off) -> double { return img[off / 8]; };
    const uint16_t* T0 = tab.data();
    std::vector<double> F(nr * nr, 0.0), inv_ref(w), x_ref(w);
    for (uint32_t row = 0; row < nr; ++row)
      for (uint32_t c = 0; c < w && c <= row; ++c) {
        double v = 0.0;
        for (uint32_t k = 0; k <= nch; ++k) v += at(T0[(row * (1 + nch) + k) * w + c]);
        F[row * nr + c] = v;
      }
    const uint16_t* Tu = T0 + nr * (1 + nch) * w;
    for (uint32_t e = 0; e < n_s; ++e) {
      uint32_t a = 0;
      while ((a + 1) * (a + 2) / 2 <= e) ++a;
      const uint32_t b = e - a * (a + 1) / 2;
      double v = 0.0;
      for (uint32_t k = 0; k < nch; ++k) v += at(Tu[e * (3 + nch) + 3 + k]);
      F[(w + a) * nr + (w + b)] = v;
    }
    for (uint32_t c = 0; c < w; ++c) {
      inv_ref[c] = 1.0 / F[c * nr + c];
      for (uint32_t t = c + 1; t < nr; ++t) {
        const double l = F[t * nr + c] * inv_ref[c];
        for (uint32_t j = c + 1; j <= t && j < nr - 1; ++j) F[t * nr + j] -= l * F[j * nr + c];
      }
    }
    for (int c = static_cast<int>(w) - 1; c >= 0; --c) {
      double dot = 0.0;
      for (uint32_t a = 0; a < r; ++a) dot += F[(w + a) * nr + c] * (0.25 + 0.01 * a);
      for (uint32_t k = c + 1; k < w; ++k) dot += F[k * nr + c] * x_ref[k];
      x_ref[c] = (F[(nr - 1) * nr + c] - dot) * inv_ref[c];
    }
    (void)hipMemcpy(dImg, img.data(), oTab, hipMemcpyHostToDevice);
    (void)hipMemcpy(dTab, tab.data(), kTabWords * 2, hipMemcpyHostToDevice);
    const Shape sh{w, nr, nch, n_s};
    const long long cold = run(sh, 1, 1, 1);
    run(sh, 4, 1, 3);
    const long long c1 = run(sh, 16, 1, 1), c8 = run(sh, 16, kWaves, 1), s1 = run(sh, 16, 1, 2), s8 = run(sh, 16, kWaves, 2);
    run(sh, 1, kWaves, 3);
    hipError_t err = hipGetLastError();
    std::vector<double> o(oTab / 8);
    (void)hipMemcpy(o.data(), dOut, oTab, hipMemcpyDeviceToHost);
    double eu = 0, es = 0, ei = 0, ex = 0;
    for (int wv : {0, kWaves - 1}) {
      for (uint32_t c = 0; c < w; ++c) {
        ei = std::fmax(ei, std::fabs(o[oInvd / 8 + 8 * wv + c] - inv_ref[c]) / std::fabs(inv_ref[c]));
        ex = std::fmax(ex, std::fabs(o[oX / 8 + kX * wv + c] - x_ref[c]) / std::fmax(1.0, std::fabs(x_ref[c])));
        for (uint32_t t = c; t < nr; ++t) eu = std::fmax(eu, std::fabs(o[kU * wv + col_off(c, nr) + (t - c)] - F[t * nr + c]));
      }
      for (uint32_t a = 0; a <= r; ++a)
        for (uint32_t b = 0; b < r && b <= a; ++b)
          es = std::fmax(es, std::fabs(o[oArena / 8 + 2 + 1300 + 300 * wv + a * (a + 1) / 2 + b] - F[(w + a) * nr + (w + b)]));
    }
    std::printf("front w = %u, r = %2u (nr = %2u), %u children: factor first call %6lld clocks; warm, one wave %5lld; eight waves at once %5lld; "
                "backward solve one wave %5lld, eight %5lld (each incl. one barrier)   max error U %.1e  S %.1e  1/d %.1e  x %.1e %s\n",
                w, r, nr, nch, cold, c1, c8, s1, s8, eu, es, ei, ex, err == hipSuccess ? "" : hipGetErrorString(err));
  }
  return 0;
}
