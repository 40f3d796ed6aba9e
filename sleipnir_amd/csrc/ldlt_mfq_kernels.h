// Multifrontal factorization for a BATCH in the interleaved layout: fronts, and FOUR LANES PER PROBLEM.
//
// ldlt_il_kernels.h (lanes = problems, left-looking pair lists) pays, per multiply-add of the
// factorization, a pair record and three operand rows from LDS: 512 x N=1000 sits at a fifth of the
// HBM roofline, a 64-problem share of config 4 at 20 us per round of latency.  The single problem's
// medicine (ldlt_mf_kernels.h) is dense FRONTS found through precomputed tables — here with sixteen
// problems side by side:
//
//   * a workgroup = one task x one group of 16 problems (one 128-byte row of every interleaved
//     array), a WAVE = one front of the level for all sixteen: lane = 4 * problem + slot.  The four
//     lanes of a problem hold the front's rows t = slot, slot + 4, ... (up to kMfqRows each) with
//     their w pivot columns IN REGISTERS; the pivot row reaches the other three lanes by a DPP quad
//     broadcast — no LDS traffic inside the elimination, as with v_readlane for one problem;
//   * the update block S(a, b) row-wise: for every row b of the structure below the pivots the
//     finished U(b, .) comes back from LDS once (the four lanes of a problem read the same address),
//     and every lane updates the rows a >= b it holds: w multiply-adds per entry against registers;
//   * the SAME plan and tables as the single problem (LdltFront, mf_tab: 16-bit byte offsets into
//     [U | arena | 1/d]): a table word o means byte 16 o + 8 p of the task's LDS for problem p of the
//     group — every value access of sixteen problems is one 128-byte row.
//
// The factorization leaves L, D, z in the interleaved arrays exactly where the pair-list kernel leaves
// them (Lx_il, D_il, zv_il; update blocks between tasks in mfc_il, one slot per entry of a root front's
// block), so the triangular solves are ldlt_fwd_il_kernel / ldlt_bwd_il_kernel unchanged (the backward
// one deals CHAINS, not columns, to its waves: a supernodal plan has a chain's columns in one level).
//
// MEASURED AND LEFT OFF (SLPX_IL_FRONTS=1 turns it on; profiles/r04_il_fronts_probe.txt): correct — the batch parity
// tests pass on it — and 1.5 x (64 x N=500) to 3 x (512 x N=1000) SLOWER than the pair-list kernel: a front is ~5000
// clocks of one wave, a task's levels hold 16, 8, 3, 2, 1 fronts, and sixteen problems' values leave room for one
// workgroup per CU — two busy waves per CU on average against the pair-list kernel's dozen.
//
// Replaces Eigen::SimplicialLDLT::factorize as used by util/sparse_regularized_ldlt.hpp:74,105
// (multistart.hpp:45-74: many instances of one model).
#pragma once

#include <hip/hip_runtime.h>

#include "ldlt_il_kernels.h"
#include "ldlt_mf_kernels.h"

namespace slpx {

constexpr int kMfqRows = 5;         // rows of a front per lane: fronts of up to 4 * kMfqRows rows
constexpr int kMfqThreads = 512;    // eight waves: a level has up to ~16 fronts
constexpr uint32_t kMfqMaxFrontRows = 4 * kMfqRows;

using LdsF64q = __attribute__((address_space(3))) double;
// value of table word `o` (byte offset in the single-problem layout) for this lane's problem
__device__ __forceinline__ double mfq_ld(uint32_t o, uint32_t pofs) {
  return *reinterpret_cast<const LdsF64q*>(static_cast<uintptr_t>(16u * o + pofs));
}
__device__ __forceinline__ void mfq_st(uint32_t o, uint32_t pofs, double v) {
  *reinterpret_cast<LdsF64q*>(static_cast<uintptr_t>(16u * o + pofs)) = v;
}
// the value lane 4 q + S of every quad q holds, in all four lanes of the quad
template <int S>
__device__ __forceinline__ double mfq_quad_bcast(double v) {
  return dpp_move<S | (S << 2) | (S << 4) | (S << 6)>(v);
}
// One front, one wave, sixteen problems.  `tabq`: LDS byte address of the front's tables (shared by the
// problems); `u0`: byte offset (single-problem layout) of the front's first entry; `invd_o`: of 1/d of its
// first column; `pofs` = 8 * problem.  Everything but `lane` and `pofs` wave-uniform.
template <int W>
__device__ __forceinline__ void mfq_front_w(uint32_t tabq, uint32_t u0, uint32_t nr, uint32_t nch, uint32_t n_s, bool root,
                                            uint32_t invd_o, const uint32_t* __restrict__ ext, double* __restrict__ contrib_p,
                                            uint32_t slot, uint32_t pofs, bool send) {
  const uint32_t stride = 2u * W * (1u + nch);
  const uint32_t upd = tabq + nr * stride;
  const uint32_t ustride = 2u * (3u + nch);
  const uint32_t R = (nr + 3u) >> 2;  // rows per lane in use (wave-uniform)
  // ---- the pivot columns: own entries + the children's values (rows beyond the front shadow its last row) ----
  double a[kMfqRows][W];
#pragma unroll
  for (int i = 0; i < kMfqRows; ++i) {
    if (static_cast<uint32_t>(i) >= R) break;
    const uint32_t t = 4u * i + slot;
    const uint32_t pr = tabq + __umul24(t < nr ? t : nr - 1u, stride);
    uint32_t ua[W];
#pragma unroll
    for (int c = 0; c < W; ++c) ua[c] = lds_ld16(pr + 2u * c);
#pragma unroll
    for (int c = 0; c < W; ++c) a[i][c] = mfq_ld(ua[c], pofs);
    for (uint32_t k = 1; k <= nch; ++k) {
      uint32_t xa[W];
#pragma unroll
      for (int c = 0; c < W; ++c) xa[c] = lds_ld16(pr + 2u * (W * k + c));
#pragma unroll
      for (int c = 0; c < W; ++c) a[i][c] += mfq_ld(xa[c], pofs);
    }
  }
  // ---- w pivots in registers.  Column c of the rows c .. w - 1 (the diagonal and the multipliers' partners
  // U(j, c)) reaches the four lanes of the problem by quad broadcasts — what v_readlane is for one problem ----
  double inv[W];
#pragma unroll
  for (int c = 0; c < W; ++c) {
    double pcol[W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
      if (j < c) {
        pcol[j] = 0.0;
        continue;
      }
      // (j, c compile-time: register row j >> 2, DPP control of slot j & 3)
      switch (j & 3) {
        case 0: pcol[j] = mfq_quad_bcast<0>(a[j >> 2][c]); break;
        case 1: pcol[j] = mfq_quad_bcast<1>(a[j >> 2][c]); break;
        case 2: pcol[j] = mfq_quad_bcast<2>(a[j >> 2][c]); break;
        default: pcol[j] = mfq_quad_bcast<3>(a[j >> 2][c]); break;
      }
    }
    inv[c] = chain_reciprocal(pcol[c]);
#pragma unroll
    for (int i = 0; i < kMfqRows; ++i) {
      if (static_cast<uint32_t>(i) >= R) break;
      const double lc = a[i][c] * inv[c];
#pragma unroll
      for (int j = c + 1; j < W; ++j) a[i][j] = __builtin_fma(-lc, pcol[j], a[i][j]);
    }
  }
  // ---- the finished columns back to U (rows above the diagonal: the scratch double), 1/d ----
#pragma unroll
  for (int i = 0; i < kMfqRows; ++i) {
    if (static_cast<uint32_t>(i) >= R) break;
    const uint32_t t = 4u * i + slot;
    const uint32_t pr = tabq + __umul24(t < nr ? t : nr - 1u, stride);
#pragma unroll
    for (int c = 0; c < W; ++c) mfq_st(lds_ld16(pr + 2u * c), pofs, a[i][c]);
  }
  if (slot == 0) {
#pragma unroll
    for (int c = 0; c < W; ++c) mfq_st(invd_o + 8u * c, pofs, inv[c]);
  }
  if (n_s == 0) return;
  // ---- update block, row-wise: S(a, b) = children - sum_c L(a, c) U(b, c) for the rows a >= b this lane holds ----
#pragma unroll
  for (int i = 0; i < kMfqRows; ++i)
#pragma unroll
    for (int c = 0; c < W; ++c) a[i][c] *= inv[c];  // L = U / d
  const uint32_t r = nr - W - 1u;
  for (uint32_t b = 0; b < r; ++b) {
    const uint32_t tb = W + b;
    double ub[W];
#pragma unroll
    for (int c = 0; c < W; ++c) ub[c] = mfq_ld(u0 + 8u * tb + mf_coff(c, nr), pofs);
    const uint32_t i0 = tb >> 2;
#pragma unroll
    for (int i = 0; i < kMfqRows; ++i) {
      if (static_cast<uint32_t>(i) < i0 || static_cast<uint32_t>(i) >= R) continue;  // (wave-uniform)
      const uint32_t t = 4u * i + slot;
      const bool valid = t >= tb && t < nr;
      const uint32_t ai = valid ? t - W : b;  // row of the block (r: the right-hand-side row)
      const uint32_t e = (ai * (ai + 1u)) / 2u + b;
      const uint32_t ur = upd + __umul24(e, ustride);
      const uint32_t o = lds_ld16(ur);
      double v = 0.0;
      for (uint32_t k = 0; k < nch; ++k) v += mfq_ld(lds_ld16(ur + 6u + 2u * k), pofs);
#pragma unroll
      for (int c = 0; c < W; ++c) v = __builtin_fma(-a[i][c], ub[c], v);
      if (valid) {
        // (the receiving entry subtracts; one slot per entry of a root front's block)
        if (!root) mfq_st(o, pofs, v);
        else if (send) contrib_p[static_cast<size_t>(ext[o]) * kIlW] = -v;  // (not for a problem outside this attempt)
      }
    }
  }
}

__device__ __forceinline__ void mfq_front(uint32_t tabq, uint32_t u0, uint32_t w, uint32_t nr, uint32_t nch, uint32_t n_s,
                                          uint32_t root, uint32_t invd_o, const uint32_t* __restrict__ ext,
                                          double* __restrict__ contrib_p, uint32_t slot, uint32_t pofs, bool send) {
  tabq = __builtin_amdgcn_readfirstlane(tabq);
  u0 = __builtin_amdgcn_readfirstlane(u0);
  w = __builtin_amdgcn_readfirstlane(w);
  nr = __builtin_amdgcn_readfirstlane(nr);
  nch = __builtin_amdgcn_readfirstlane(nch);
  n_s = __builtin_amdgcn_readfirstlane(n_s);
  root = __builtin_amdgcn_readfirstlane(root);
  invd_o = __builtin_amdgcn_readfirstlane(invd_o);
  switch (w) {
#define SLPX_MFQ_CASE(W) case W: mfq_front_w<W>(tabq, u0, nr, nch, n_s, (root & 1u) != 0, invd_o, ext, contrib_p, slot, pofs, send); break;
    SLPX_MFQ_CASE(1) SLPX_MFQ_CASE(2) SLPX_MFQ_CASE(3) SLPX_MFQ_CASE(4) SLPX_MFQ_CASE(5) SLPX_MFQ_CASE(6) SLPX_MFQ_CASE(7)
    SLPX_MFQ_CASE(8)
#undef SLPX_MFQ_CASE
    default: break;
  }
}

// LDS of a task (bytes): [U | arena | 1/d] x 16 problems (128 bytes per double of the single-problem layout),
// then the task's image — the tables and lists of mf_carve from o_tab up to the counters — as it is in memory.
struct MfqDev {
  const double* lhs_il = nullptr;  // [group][nnz_lhs][16]
  const double* rhs_il = nullptr;  // [group][n][16]
  const double* reg = nullptr;     // [b]{delta, gamma}; delta = NaN: not part of this attempt
  double* Lx_il = nullptr;         // [group][nnzL][16]
  double* D_il = nullptr;
  double* zv_il = nullptr;
  double* contrib_il = nullptr;    // [group][mf_n_contrib][16]: update blocks between tasks
  LdltStats* stats_part = nullptr; // [task][b]
  long long nnz_lhs = 0, nnzL = 0, n_contrib = 0;
  int n = 0, batch = 0;
};

__global__ __launch_bounds__(kMfqThreads) void ldlt_mfq_factor_kernel(LdltDev L, MfDev Mf, uint32_t task_base, MfqDev Q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const uint32_t lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t task_index = task_base + blockIdx.x;
  const LdltTask t = L.tasks[task_index];
  const LdltMfTask m = Mf.tasks[task_index];
  const MfCarve cv = mf_carve(t, m);
  const int g = blockIdx.y;  // group of 16 problems
  // the load / store phases: sixteen consecutive lanes = the sixteen problems of a row
  const int pl = tid & (kIlW - 1), row0 = tid >> kIlWShift;
  constexpr int kRowsAtOnce = kMfqThreads / kIlW;
  const int b = g * kIlW + pl;
  const bool in_batch = b < Q.batch;
  const double delta = in_batch ? Q.reg[2 * b] : 0.0, gamma = in_batch ? Q.reg[2 * b + 1] : 0.0;
  const bool active = in_batch && delta == delta;
  // (a group none of whose problems takes part in this attempt has nothing to do; every lane looks at all sixteen
  // itself: a workgroup vote would bring static LDS, and the tables address the dynamic block from byte 0)
  {
    bool any = false;
    for (int k = 0; k < kIlW; ++k) {
      const int bk = g * kIlW + k;
      const double dk = bk < Q.batch ? Q.reg[2 * bk] : __builtin_nan("");
      any = any || dk == dk;
    }
    if (!any) return;
  }

  const uint32_t n_val = t.n_ent + m.arena + t.n_col;  // doubles of the single-problem value region in use
  const uint32_t o_img = 128u * n_val;                // the image starts behind the sixteen-fold value region
  const uint32_t img_shift = o_img - cv.o_tab;
  double* Uq = reinterpret_cast<double*>(smem_raw);
  auto lds_at = [&](uint32_t carve_off) { return smem_raw + carve_off + img_shift; };
  const uint32_t* lvl = reinterpret_cast<const uint32_t*>(lds_at(cv.o_lvl));
  const uint32_t* ext = reinterpret_cast<const uint32_t*>(lds_at(cv.o_ext));
  const int32_t* src = reinterpret_cast<const int32_t*>(lds_at(cv.o_src));
  const uint8_t* flags = reinterpret_cast<const uint8_t*>(lds_at(cv.o_flags));
  const uint16_t* cent = reinterpret_cast<const uint16_t*>(lds_at(cv.o_cent));
  const uint32_t* cptr = reinterpret_cast<const uint32_t*>(lds_at(cv.o_cptr));
  const uint32_t* cidx = reinterpret_cast<const uint32_t*>(lds_at(cv.o_cidx));
  {
    const uint4* src16 = Mf.image + static_cast<size_t>(task_index) * Mf.image_stride16;
    uint4* dst16 = reinterpret_cast<uint4*>(smem_raw + o_img);
    const uint32_t n16 = Mf.image_desc[task_index].y;
    uint32_t i = tid;
    for (; i + 3 * kMfqThreads < n16; i += 4 * kMfqThreads) {
      const uint4 v0 = src16[i], v1 = src16[i + kMfqThreads], v2 = src16[i + 2 * kMfqThreads], v3 = src16[i + 3 * kMfqThreads];
      dst16[i] = v0;
      dst16[i + kMfqThreads] = v1;
      dst16[i + 2 * kMfqThreads] = v2;
      dst16[i + 3 * kMfqThreads] = v3;
    }
    for (; i < n16; i += kMfqThreads) dst16[i] = src16[i];
  }
  __syncthreads();

  const double* lhs = Q.lhs_il + static_cast<size_t>(g) * Q.nnz_lhs * kIlW + pl;
  const double* rhs = Q.rhs_il + static_cast<size_t>(g) * Q.n * kIlW + pl;
  double* contrib = Q.contrib_il + static_cast<size_t>(g) * Q.n_contrib * kIlW + pl;
  // ---- matrix values + regularization (four loads in flight); entries with update slots below ----
  auto matrix_value = [&](uint32_t e) {
    const int32_t s0 = src[e];
    const uint8_t fl = flags[e];
    const double* from = (fl & 4) ? rhs : lhs;
    const double v = from[static_cast<size_t>(s0 >= 0 ? s0 : 0) * kIlW];
    double acc = s0 >= 0 ? v : 0.0;
    if (fl & 1) acc += (fl & 2) ? -gamma : delta;
    return acc;
  };
  {
    uint32_t e = row0;
    for (; e + 3 * kRowsAtOnce < t.n_ent; e += 4 * kRowsAtOnce) {
      double v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = matrix_value(e + kRowsAtOnce * j);
#pragma unroll
      for (int j = 0; j < 4; ++j) Uq[(e + kRowsAtOnce * j) * kIlW + pl] = v[j];
    }
    for (; e < t.n_ent; e += kRowsAtOnce) Uq[e * kIlW + pl] = matrix_value(e);
  }
  // the arena's constants: 0.0 (an absent child value) and the scratch double
  if (row0 < 2) Uq[(t.n_ent + row0) * kIlW + pl] = 0.0;
  __syncthreads();
  for (uint32_t j = row0; j < m.n_cent; j += kRowsAtOnce) {
    const uint32_t i = cent[j];
    const uint32_t cb = cptr[j], ce = cptr[j + 1];
    double acc = Uq[i * kIlW + pl];
    for (uint32_t c = cb; c < ce; c += 4) {  // (four in flight, subtracted in list order)
      const double v0 = contrib[static_cast<size_t>(cidx[c]) * kIlW];
      const double v1 = contrib[static_cast<size_t>(cidx[c + 1 < ce ? c + 1 : c]) * kIlW];
      const double v2 = contrib[static_cast<size_t>(cidx[c + 2 < ce ? c + 2 : c]) * kIlW];
      const double v3 = contrib[static_cast<size_t>(cidx[c + 3 < ce ? c + 3 : c]) * kIlW];
      acc -= v0;
      if (c + 1 < ce) acc -= v1;
      if (c + 2 < ce) acc -= v2;
      if (c + 3 < ce) acc -= v3;
    }
    Uq[i * kIlW + pl] = acc;
  }
  __syncthreads();

  // ---- levels: a wave per front, four lanes per problem ----
  {
    const LdltFront* gfr = Mf.fronts + m.front_off;
    const uint32_t slot = lane & 3u, pofs = 8u * (lane >> 2);
    const int bq = g * kIlW + static_cast<int>(lane >> 2);
    const double delta_q = bq < Q.batch ? Q.reg[2 * bq] : 0.0;
    const bool send = bq < Q.batch && delta_q == delta_q;
    double* contrib_q = Q.contrib_il + static_cast<size_t>(g) * Q.n_contrib * kIlW + (lane >> 2);
    const uint32_t tabq = o_img;  // (cv.o_tab is the image's first byte)
    uint32_t beg = __builtin_amdgcn_readfirstlane(lvl[0]), end = __builtin_amdgcn_readfirstlane(t.n_lvl ? lvl[1] : 0);
    for (uint32_t l = 0; l < t.n_lvl; ++l) {
      const uint32_t next_end = __builtin_amdgcn_readfirstlane(lvl[l + 2 <= t.n_lvl ? l + 2 : t.n_lvl]);
      for (uint32_t q = beg + wave; q < end; q += kMfqThreads / 64) {
        const u32x4 d = s_load_desc(gfr + q);
        const uint32_t w = d[2] & 0xffu, nr = (d[2] >> 8) & 0xffu, nch = (d[2] >> 16) & 0xffu, root = (d[2] >> 24) & 3u;
        mfq_front(tabq + 2u * d[0], 8u * (d[1] & 0xffffu), w, nr, nch, d[3] & 0xffffu, root, cv.o_invd + 8u * (d[1] >> 16),
                  ext + (d[3] >> 16), contrib_q, slot, pofs, send);
      }
      __syncthreads();
      beg = end;
      end = next_end;
    }
  }

  // ---- results: D, L = U / d, z in the interleaved arrays; the task's inertia counters per problem ----
  double* Lx = Q.Lx_il + static_cast<size_t>(g) * Q.nnzL * kIlW + pl;
  double* D = Q.D_il + static_cast<size_t>(g) * Q.n * kIlW + pl;
  double* zv = Q.zv_il + static_cast<size_t>(g) * Q.n * kIlW + pl;
  const double* invd = Uq + static_cast<size_t>(cv.o_invd / 8u) * kIlW + pl;
  const uint32_t* g_out = L.ent_out + t.ent_off;
  const uint16_t* g_col = L.ent_col + t.ent_off;
  int n_pos = 0, n_neg = 0, n_zero = 0, n_bad = 0;
  double min_abs = __longlong_as_double(0x7ff0000000000000ll);
  for (uint32_t i = row0; i < t.n_ent; i += kRowsAtOnce) {
    const double u = Uq[i * kIlW + pl];
    const uint8_t fl = flags[i];
    const uint32_t o = g_out[i];
    if (fl & 1) {
      if (active) D[static_cast<size_t>(o) * kIlW] = u;
      const double eps = 2.220446049250313e-16;  // inertia.hpp:40-50
      if (u > eps) ++n_pos;
      else if (u < -eps) ++n_neg;
      else ++n_zero;
      if (u == 0.0 || !isfinite(u)) ++n_bad;
      else min_abs = fmin(min_abs, fabs(u));
    } else if (active) {
      const double v = u * invd[static_cast<size_t>(g_col[i]) * kIlW];
      if (fl & 4) zv[static_cast<size_t>(o) * kIlW] = v;
      else Lx[static_cast<size_t>(o) * kIlW] = v;
    }
  }
  __syncthreads();
  {
    int* ci = reinterpret_cast<int*>(smem_raw);                                   // [4][rows][16]
    double* cm = reinterpret_cast<double*>(smem_raw) + 2 * kRowsAtOnce * kIlW;   // [rows][16], after the 4 int planes
    ci[(0 * kRowsAtOnce + row0) * kIlW + pl] = n_pos;
    ci[(1 * kRowsAtOnce + row0) * kIlW + pl] = n_neg;
    ci[(2 * kRowsAtOnce + row0) * kIlW + pl] = n_zero;
    ci[(3 * kRowsAtOnce + row0) * kIlW + pl] = n_bad;
    cm[row0 * kIlW + pl] = min_abs;
    __syncthreads();
    if (row0 == 0) {
      for (int r = 1; r < kRowsAtOnce; ++r) {
        n_pos += ci[(0 * kRowsAtOnce + r) * kIlW + pl];
        n_neg += ci[(1 * kRowsAtOnce + r) * kIlW + pl];
        n_zero += ci[(2 * kRowsAtOnce + r) * kIlW + pl];
        n_bad += ci[(3 * kRowsAtOnce + r) * kIlW + pl];
        min_abs = fmin(min_abs, cm[r * kIlW + pl]);
      }
      if (active) {
        LdltStats st;
        st.n_pos = n_pos;
        st.n_neg = n_neg;
        st.n_zero = n_zero;
        st.n_bad = n_bad;
        st.min_abs_bits = static_cast<unsigned long long>(__double_as_longlong(min_abs));
        Q.stats_part[static_cast<size_t>(task_index) * Q.batch + b] = st;
      }
    }
  }
}

}  // namespace slpx
