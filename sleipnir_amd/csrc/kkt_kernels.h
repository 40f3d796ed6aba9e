// KKT assembly (interior_point.hpp:426-448): per-entry functions and the grid-stride bodies
// built on them.  Shared by the stand-alone kernels of kernels.hip and by the factorization
// kernel, whose tasks can evaluate the entries they need themselves (device.hpp: KktFuse).
// COHERENT: the results cross workgroups inside the launch.
#pragma once

#include <hip/hip_runtime.h>

#include "coherent.h"
#include "device.hpp"

namespace slpx {

// The entry functions below are chains of dependent trips to memory (pointer pair -> indices ->
// values); what a lone lane can do about that is to have every index of a row in flight at
// once and then every value: the loops run kKktUnroll entries per trip, predicated, and sum in
// the entries' own order (so the result does not depend on the unrolling).
constexpr int kKktUnroll = 8;

// Rounding is pinned in everything below (no contraction into fused multiply-adds left to the
// compiler's mood in each inlining context): the factorization's tasks evaluate the same terms
// one per lane (kkt_term_product, kkt_terms_sum) and must get the same bits as the stand-alone
// kernels — a batch of problems and a single problem take different routes to the same system
// (tests: batch items are independent; the evaluated system equals the assembled one).
// a * (-Σ c_i + μ/s + z) (interior_point.hpp:441) and (a Σ) b (:426-431), Σ = z/s.
__device__ __forceinline__ double kkt_ait_term(double a, double s, double z, double ci, double m) {
#pragma clang fp contract(off)
  const double sinv = 1.0 / s;
  const double sigma = sinv * z;
  const double t = -sigma * ci;
  return a * (__builtin_fma(m, sinv, t) + z);
}
__device__ __forceinline__ double kkt_prod_term(double a, double s, double z, double b) {
#pragma clang fp contract(off)
  const double sigma = (1.0 / s) * z;
  const double as = a * sigma;
  return as * b;
}
__device__ __forceinline__ double kkt_mul(double a, double b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ double kkt_add(double a, double b) {
#pragma clang fp contract(off)
  return a + b;
}

// One entry of the lhs that is not a plain copy of a V value: the sum of the Hessian / Jacobian
// values that land on it plus the A_i^T Sigma A_i products (interior_point.hpp:426-431).
__device__ __forceinline__ double kkt_lhs_general(const KktDev& K, const double* __restrict__ V,
                                                  const double* __restrict__ s, const double* __restrict__ z,
                                                  int k) {
  const int db = K.dptr[k], de = K.dptr[k + 1], pb = K.pptr[k], pe = K.pptr[k + 1];
  double direct = 0.0;
  for (int d = db; d < de; d += kKktUnroll) {
    int idx[kKktUnroll];
#pragma unroll
    for (int u = 0; u < kKktUnroll; ++u) idx[u] = d + u < de ? K.dsrc[d + u] : 0;
    double v[kKktUnroll];
#pragma unroll
    for (int u = 0; u < kKktUnroll; ++u) v[u] = d + u < de ? V[idx[u]] : 0.0;
#pragma unroll
    for (int u = 0; u < kKktUnroll; ++u)
      if (d + u < de) direct += v[u];
  }
  double prod = 0.0;
  constexpr int kU = kKktUnroll / 2;
  for (int p = pb; p < pe; p += kU) {
    int r[kU], ia[kU], ib[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const bool on = p + u < pe;
      r[u] = on ? K.pr[p + u] : 0;
      ia[u] = on ? K.pa[p + u] : 0;
      ib[u] = on ? K.pb[p + u] : 0;
    }
    double sr[kU], zr[kU], va[kU], vb[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const bool on = p + u < pe;
      sr[u] = on ? s[r[u]] : 1.0;
      zr[u] = on ? z[r[u]] : 0.0;
      va[u] = on ? V[ia[u]] : 0.0;
      vb[u] = on ? V[ib[u]] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (p + u < pe) prod = kkt_add(prod, kkt_prod_term(va[u], sr[u], zr[u], vb[u]));
    }
  }
  return direct + prod;
}

// One entry of the rhs (interior_point.hpp:437-448)
__device__ __forceinline__ double kkt_rhs_entry(const KktDev& K, const double* __restrict__ V,
                                                const double* __restrict__ s, const double* __restrict__ y,
                                                const double* __restrict__ z, const double m, int j) {
  if (j >= K.n) return -V[K.off_ce + j - K.n];
  const double* ci = V + K.off_ci;
  const double* Ae = V + K.off_Ae;
  const double* Ai = V + K.off_Ai;
  const int gs = K.g_src[j];
  const int eb = K.ae_colptr[j], ee = K.ae_colptr[j + 1], ib = K.ai_colptr[j], ie = K.ai_colptr[j + 1];
  const double g = gs >= 0 ? V[gs] : 0.0;
  double aey = 0.0;
  for (int p = eb; p < ee; p += kKktUnroll) {
    int r[kKktUnroll];
    double a[kKktUnroll];
#pragma unroll
    for (int u = 0; u < kKktUnroll; ++u) {
      const bool on = p + u < ee;
      r[u] = on ? K.ae_rowidx[p + u] : 0;
      a[u] = on ? Ae[p + u] : 0.0;
    }
    double yr[kKktUnroll];
#pragma unroll
    for (int u = 0; u < kKktUnroll; ++u) yr[u] = p + u < ee ? y[r[u]] : 0.0;
#pragma unroll
    for (int u = 0; u < kKktUnroll; ++u)
      if (p + u < ee) aey = kkt_add(aey, kkt_mul(a[u], yr[u]));
  }
  double ait = 0.0;
  constexpr int kU = kKktUnroll / 2;
  for (int p = ib; p < ie; p += kU) {
    int r[kU];
    double a[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const bool on = p + u < ie;
      r[u] = on ? K.ai_rowidx[p + u] : 0;
      a[u] = on ? Ai[p + u] : 0.0;
    }
    double sr[kU], zr[kU], cr[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const bool on = p + u < ie;
      sr[u] = on ? s[r[u]] : 1.0;
      zr[u] = on ? z[r[u]] : 0.0;
      cr[u] = on ? ci[r[u]] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (p + u < ie) ait = kkt_add(ait, kkt_ait_term(a[u], sr[u], zr[u], cr[u], m));
    }
  }
  return -g + aey + ait;
}

// Entries given as lists of terms in LDS (device.hpp: KktTerm; built by
// DeviceNlp::build_inline_kkt from the same maps the functions above walk), evaluated by a whole
// workgroup: ONE TERM PER LANE — a lone lane issues an instruction every 5-9 clocks, so walking a
// ten-term row alone costs more than the trip to memory — then every sum adds up its own terms
// from LDS in their order (kkt_terms_sum; same accumulators and order of summation as
// kkt_lhs_general / kkt_rhs_entry).
struct KktTermLoads {
  int kind;
  double va, x1, zr, vc;
};
__device__ __forceinline__ KktTermLoads kkt_term_fetch(const KktTerm t, bool on, const double* __restrict__ V,
                                                       const double* __restrict__ s, const double* __restrict__ y,
                                                       const double* __restrict__ z) {
  KktTermLoads r;
  r.kind = on ? (t.b >> 28) : -1;
  const int row = t.b & 0x0fffffff;
  r.va = on ? V[t.a] : 0.0;
  const double* p1 = r.kind == 2 ? y : s;
  r.x1 = r.kind >= 2 ? p1[row] : 1.0;
  r.zr = r.kind >= 3 ? z[row] : 0.0;
  r.vc = r.kind >= 3 ? V[t.c] : 0.0;
  return r;
}
__device__ __forceinline__ double kkt_term_product(const KktTermLoads& r, const double m) {
  if (r.kind < 2) return r.va;
  if (r.kind == 2) return kkt_mul(r.va, r.x1);
  return r.kind == 3 ? kkt_ait_term(r.va, r.x1, r.zr, r.vc, m) : kkt_prod_term(r.va, r.x1, r.zr, r.vc);
}
__device__ __forceinline__ double kkt_terms_sum(const KktTerm* __restrict__ terms, const double* __restrict__ prod,
                                                uint32_t first, uint32_t count, bool is_rhs) {
  double direct = 0.0, pr = 0.0, g = 0.0, aey = 0.0, ait = 0.0;
  for (uint32_t q = first; q < first + count; ++q) {
    const int kind = terms[q].b >> 28;
    const double v = prod[q];
    if (kind == 0) direct += v;
    else if (kind == 1) g = v;
    else if (kind == 2) aey += v;
    else if (kind == 3) ait += v;
    else pr += v;
  }
  return is_rhs ? -g + aey + ait : direct + pr;
}

// The same sum for the step kernel of the fronts, whose image says how many terms of each kind an entry has
// (kMfTerm*: DeviceNlp::build_mf re-encodes the entry's word; the terms of an entry are laid down kind by kind —
// [g] [A_e^T y ...] [A_i^T ...] for a right-hand-side row, [direct ...] [products ...] for an entry of the matrix,
// DeviceNlp::build_inline_kkt), so the kinds need not be read back and nothing branches on them: one trip to LDS
// per four terms of both groups, one select and one add per term.  The loop above compiles to a chain of divergent
// branches, ~100 instructions per term, and a lane's entry has up to a dozen terms (a right-hand-side row sums a
// column of A_e^T y): 2 us of the step's critical path at cart-pole N=1000 by the kernel's own clocks.  Same
// accumulators, same order, same bits: acc + 0.0 is acc for every acc but -0.0, which a sum that starts at +0.0
// never is.
constexpr uint32_t kMfTermFirstBits = 14, kMfTermABits = 8, kMfTermBBits = 7;  // + 1 bit: the row has a gradient term
__host__ __device__ inline bool mf_term_code_fits(uint32_t first, uint32_t n_a, uint32_t n_b) {
  return first < (1u << kMfTermFirstBits) && n_a < (1u << kMfTermABits) && n_b < (1u << kMfTermBBits);
}
__host__ __device__ inline uint32_t mf_term_code(uint32_t first, uint32_t has_g, uint32_t n_a, uint32_t n_b) {
  return first | (n_a << kMfTermFirstBits) | (n_b << (kMfTermFirstBits + kMfTermABits)) |
         (has_g << (kMfTermFirstBits + kMfTermABits + kMfTermBBits));
}
__device__ __forceinline__ double kkt_terms_sum_grouped(const double* __restrict__ prod, uint32_t code, bool is_rhs) {
  const uint32_t first = code & ((1u << kMfTermFirstBits) - 1u);
  const uint32_t n_a = (code >> kMfTermFirstBits) & ((1u << kMfTermABits) - 1u);
  const uint32_t n_b = (code >> (kMfTermFirstBits + kMfTermABits)) & ((1u << kMfTermBBits) - 1u);
  const uint32_t has_g = code >> (kMfTermFirstBits + kMfTermABits + kMfTermBBits);
  const double g = prod[first];  // (read whether it is one or not: a select, not a branch)
  const uint32_t p_a = first + has_g, p_b = p_a + n_a;
  const uint32_t n = n_a > n_b ? n_a : n_b;
  double a = 0.0, b = 0.0;
  for (uint32_t i0 = 0; i0 < n; i0 += 4) {
    double va[4], vb[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      va[u] = prod[p_a + (i0 + u < n_a ? i0 + u : 0u)];
      vb[u] = prod[p_b + (i0 + u < n_b ? i0 + u : 0u)];
    }
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u) {
      a += i0 + u < n_a ? va[u] : 0.0;
      b += i0 + u < n_b ? vb[u] : 0.0;
    }
  }
  // (kkt_terms_sum: -g + aey + ait with g = 0.0 where the row has no gradient term; direct + pr)
  return is_rhs ? -(has_g ? g : 0.0) + a + b : a + b;
}

// (`vblock` of `vgrid`: the block's position among the blocks doing this job — the fused
// kkt_build_kernel gives each job a slice of one launch)
template <bool COHERENT = false>
__device__ __forceinline__ void kkt_assemble_body(const KktDev& K, const double* __restrict__ V,
                                                  const double* __restrict__ s,
                                                  const double* __restrict__ z, double* __restrict__ lhs,
                                                  int vblock, int vgrid) {
  auto general = [&](int k) { return kkt_lhs_general(K, V, s, z, k); };
  // Four entries per thread with their gathers in flight together (the kernel is a chain
  // of dependent loads; HBM bandwidth needs the memory-level parallelism), and a one-index
  // fast path for the entries that are plain copies of a V value (all of A_e, most of H).
  const int stride = vgrid * blockDim.x;
  int k = vblock * blockDim.x + threadIdx.x;
  for (; k + 3 * stride < K.nnz_lhs; k += 4 * stride) {
    const int f0 = K.fast_src[k], f1 = K.fast_src[k + stride], f2 = K.fast_src[k + 2 * stride],
              f3 = K.fast_src[k + 3 * stride];
    double v0 = f0 >= 0 ? V[f0] : 0.0, v1 = f1 >= 0 ? V[f1] : 0.0, v2 = f2 >= 0 ? V[f2] : 0.0,
           v3 = f3 >= 0 ? V[f3] : 0.0;
    if (f0 == -2) v0 = general(k);
    if (f1 == -2) v1 = general(k + stride);
    if (f2 == -2) v2 = general(k + 2 * stride);
    if (f3 == -2) v3 = general(k + 3 * stride);
    coherent_store(&lhs[k], v0, COHERENT);
    coherent_store(&lhs[k + stride], v1, COHERENT);
    coherent_store(&lhs[k + 2 * stride], v2, COHERENT);
    coherent_store(&lhs[k + 3 * stride], v3, COHERENT);
  }
  for (; k < K.nnz_lhs; k += stride) {
    const int f = K.fast_src[k];
    coherent_store(&lhs[k], f >= 0 ? V[f] : (f == -2 ? general(k) : 0.0), COHERENT);
  }
}


template <bool COHERENT = false>
__device__ __forceinline__ void kkt_rhs_body(const KktDev& K, const double* __restrict__ V,
                                             const double* __restrict__ s, const double* __restrict__ y,
                                             const double* __restrict__ z, const double m,
                                             double* __restrict__ rhs, int vblock, int vgrid) {
  for (int j = vblock * blockDim.x + threadIdx.x; j < K.dim; j += vgrid * blockDim.x)
    coherent_store(&rhs[j], kkt_rhs_entry(K, V, s, y, z, m, j), COHERENT);
}


// p_s = (c_i - s) + A_i p_x, p_z = mu/s - z - Sigma p_s (interior_point.hpp:479-480); rounding
// pinned like the assembly's: the stand-alone kernel and the backward solve's tasks
// (device.hpp: BacksubFuse) must give the same bits.
__device__ __forceinline__ double backsub_dot(double acc, double a, double p) { return __builtin_fma(a, p, acc); }
__device__ __forceinline__ void backsub_row(double ci, double s, double z, double m, double aipx, double* ps,
                                            double* pz) {
#pragma clang fp contract(off)
  const double sinv = 1.0 / s;
  const double p_s = (ci - s) + aipx;
  *ps = p_s;
  const double sz = sinv * z;
  const double t = sz * p_s;
  *pz = (m * sinv - z) - t;
}

template <bool COHERENT = false>
__device__ __forceinline__ void step_backsub_body(const KktDev& K, const double* __restrict__ V,
                                                  const double* __restrict__ p, const double* __restrict__ s,
                                                  const double* __restrict__ z, double m, double* __restrict__ ps,
                                                  double* __restrict__ pz, int first, int stride) {
  const double* ci = V + K.off_ci;
  for (int r = first; r < K.m_i; r += stride) {
    double aipx = 0.0;
    for (int q = K.ai_rowptr[r]; q < K.ai_rowptr[r + 1]; ++q)
      aipx = backsub_dot(aipx, V[K.ai_src[q]], coherent_load(&p[K.ai_col[q]], COHERENT));
    backsub_row(ci[r], s[r], z[r], m, aipx, &ps[r], &pz[r]);
  }
}

}  // namespace slpx
