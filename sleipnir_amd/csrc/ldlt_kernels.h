// Task-parallel sparse LDLᵀ on gfx950: numeric factorization + inertia, forward and
// backward substitution.  One workgroup = one task (an elimination-subtree that fits
// in LDS, see ldlt_symbolic.hpp); one launch = one round of independent tasks.
//
// Every kernel first STAGES its task into LDS with bulk coalesced loads (matrix
// values, update-pair lists, level pointers; for the solves the L values it will
// touch), so the dependent chain of a level is LDS latency only.  A single
// direct-transcription KKT system is bound by (etree height) × (LDS round trip),
// not by HBM bandwidth or flops (SURVEY.md §7 hard parts 1-3): nnz(L) ≈ 74 k and
// ≈0.7 MFLOP at N=1000.  Batched, the bound is the HBM traffic 12·nnz(lhs) + 16·nnz(L)
// per factorization and 32·nnz(L) + 16·n per solve.
//
// Replaces Eigen::SimplicialLDLT::factorize / vectorD / solve as used by
// util/sparse_regularized_ldlt.hpp:74-83,105-109,159-161 and Inertia (inertia.hpp:40-50).
#pragma once

#include <hip/hip_runtime.h>

#include "device.hpp"

namespace slpx {

__device__ __forceinline__ uint32_t up8l(uint32_t b) { return (b + 7u) & ~7u; }

// Sum over the 8 lanes of an aligned lane group.
__device__ __forceinline__ double group8_sum(double v) {
  v += __shfl_xor(v, 4, 8);
  v += __shfl_xor(v, 2, 8);
  v += __shfl_xor(v, 1, 8);
  return v;
}

// ---------------------------------------------------------------------------
// Factorization.  Entry (i,j) of column j:  U(i,j) = A(i,j) [+δ | −γ on the
// diagonal] − Σ contributions of child tasks − Σ_k U(i,k)·U(j,k)/d_k, the last sum
// over an explicit pair list (left-looking, entry-parallel).  d_j = U(j,j),
// L(i,j) = U(i,j)/d_j.  Eight lanes cooperate on one entry's pair list.
//
// LDS (bytes, 8-aligned sections):
//   U[n_ent] f64 | invd[n_col] f64 | pairs[np] 8 B | pptr[n_ent+n_ext+1] u32 |
//   col[n_ent] u16 | flags[n_ent] u8 | lvl[n_lvl+1] u32 | counters 32 B
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ldlt_factor_kernel(
    LdltDev L, uint32_t task_base, const double* __restrict__ lhs, int lhs_stride,
    const double* __restrict__ reg, const uint8_t* __restrict__ active, double* __restrict__ Lx,
    long long lx_stride, double* __restrict__ D, int n, double* __restrict__ contrib,
    int contrib_stride, LdltStats* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const LdltTask t = L.tasks[task_base + blockIdx.x];
  const int b = blockIdx.y;
  if (!active[b]) return;
  const double delta = reg[2 * b], gamma = reg[2 * b + 1];
  const int tid = threadIdx.x;
  lhs += static_cast<size_t>(b) * lhs_stride;
  Lx += static_cast<size_t>(b) * lx_stride;
  D += static_cast<size_t>(b) * n;
  contrib += static_cast<size_t>(b) * contrib_stride;

  const uint32_t* g_pptr = L.ent_pair_ptr + t.pair_ptr_off;
  const uint32_t n_pp = t.n_ent + t.n_ext + 1;
  const uint32_t np = g_pptr[n_pp - 1];

  double* U = reinterpret_cast<double*>(smem_raw);
  double* invd = U + t.n_ent;
  unsigned char* cur = reinterpret_cast<unsigned char*>(invd + t.n_col);
  uint2* pairs = reinterpret_cast<uint2*>(cur);
  cur += 8 * np;
  uint32_t* pptr = reinterpret_cast<uint32_t*>(cur);
  cur += up8l(4 * n_pp);
  uint16_t* col = reinterpret_cast<uint16_t*>(cur);
  cur += up8l(2 * t.n_ent);
  uint8_t* flags = cur;
  cur += up8l(t.n_ent);
  uint32_t* lvl = reinterpret_cast<uint32_t*>(cur);
  cur += up8l(4 * (t.n_lvl + 1));
  int* s_cnt = reinterpret_cast<int*>(cur);
  unsigned long long* s_minp = reinterpret_cast<unsigned long long*>(cur + 16);

  // ---- stage ----
  {
    const uint2* g_pairs = reinterpret_cast<const uint2*>(L.pairs) + t.pair_off;
    for (uint32_t i = tid; i < np; i += 256) pairs[i] = g_pairs[i];
    for (uint32_t i = tid; i < n_pp; i += 256) pptr[i] = g_pptr[i];
    const uint32_t* g_lvl = L.lvl_ptr + t.lvl_off;
    for (uint32_t i = tid; i < t.n_lvl + 1; i += 256) lvl[i] = g_lvl[i];
    const uint32_t* cptr = L.ent_contrib_ptr + t.contrib_ptr_off;
    const uint32_t* cidx = L.contrib_idx + t.contrib_off;
    for (uint32_t i = tid; i < t.n_ent; i += 256) {
      const uint32_t e = t.ent_off + i;
      const int32_t src = L.ent_src[e];
      const uint8_t fl = L.ent_flags[e];
      double acc = src >= 0 ? lhs[src] : 0.0;
      if (fl & 1) acc += (fl & 2) ? -gamma : delta;
      for (uint32_t c = cptr[i]; c < cptr[i + 1]; ++c) acc -= contrib[cidx[c]];
      U[i] = acc;
      col[i] = L.ent_col[e];
      flags[i] = fl;
    }
    if (tid < 4) s_cnt[tid] = 0;
    if (tid == 0) *s_minp = 0x7ff0000000000000ull;  // +inf
  }
  __syncthreads();

  // ---- level loop: all in LDS ----
  const int lane8 = tid & 7, grp = tid >> 3;
  for (uint32_t l = 0; l < t.n_lvl; ++l) {
    const uint32_t beg = lvl[l], end = lvl[l + 1];
    for (uint32_t i = beg + grp; i < end; i += 32) {
      const uint32_t pb = pptr[i], pe = pptr[i + 1];
      double partial = 0.0;
      for (uint32_t q = pb + lane8; q < pe; q += 8) {
        const uint2 pr = pairs[q];
        partial += (U[pr.x & 0xffffu] * invd[pr.y & 0xffffu]) * U[pr.x >> 16];
      }
      partial = group8_sum(partial);
      if (lane8 == 0) {
        const double u = U[i] - partial;
        U[i] = u;
        if (flags[i] & 1) invd[col[i]] = 1.0 / u;
      }
    }
    __syncthreads();
  }

  // ---- update blocks for ancestor tasks (later rounds) ----
  for (uint32_t x = grp; x < t.n_ext; x += 32) {
    const uint32_t pb = pptr[t.n_ent + x], pe = pptr[t.n_ent + x + 1];
    double partial = 0.0;
    for (uint32_t q = pb + lane8; q < pe; q += 8) {
      const uint2 pr = pairs[q];
      partial += (U[pr.x & 0xffffu] * invd[pr.y & 0xffffu]) * U[pr.x >> 16];
    }
    partial = group8_sum(partial);
    if (lane8 == 0) contrib[L.ext_dst[t.ext_off + x]] = partial;
  }

  // ---- results + inertia (inertia.hpp:40-50: |d| <= eps counts as zero) ----
  for (uint32_t i = tid; i < t.n_ent; i += 256) {
    const uint32_t e = t.ent_off + i;
    const double u = U[i];
    if (flags[i] & 1) {
      D[L.ent_out[e]] = u;
      const double eps = 2.220446049250313e-16;
      if (u > eps) atomicAdd(&s_cnt[0], 1);
      else if (u < -eps) atomicAdd(&s_cnt[1], 1);
      else atomicAdd(&s_cnt[2], 1);
      if (u == 0.0 || !isfinite(u)) atomicAdd(&s_cnt[3], 1);
      else atomicMin(s_minp, static_cast<unsigned long long>(__double_as_longlong(fabs(u))));
    } else {
      Lx[L.ent_out[e]] = u * invd[col[i]];
    }
  }
  __syncthreads();
  if (tid < 4 && s_cnt[tid] != 0) atomicAdd(reinterpret_cast<int*>(&stats[b]) + tid, s_cnt[tid]);
  if (tid == 0) atomicMin(&stats[b].min_abs_bits, *s_minp);
}

// ---------------------------------------------------------------------------
// Forward substitution L y = P b followed by z = D⁻¹ y.
// LDS: y[n_col+1] f64 | vals[n_items] f64 | refs[n_items] u32 | ptr[n_col+1] u32 | lvl[n_lvl+1] u32
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ldlt_fwd_kernel(
    LdltDev L, uint32_t task_base, const double* __restrict__ rhs, int n,
    const double* __restrict__ Lx, long long lx_stride, const double* __restrict__ D,
    double* __restrict__ scontrib, int scontrib_stride, double* __restrict__ zv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const LdltTask t = L.tasks[task_base + blockIdx.x];
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  rhs += static_cast<size_t>(b) * n;
  Lx += static_cast<size_t>(b) * lx_stride;
  D += static_cast<size_t>(b) * n;
  zv += static_cast<size_t>(b) * n;
  scontrib += static_cast<size_t>(b) * scontrib_stride;

  const uint32_t* g_ptr = L.fwd_ptr + t.colptr_off;
  const uint32_t n_items = g_ptr[t.n_col];
  double* y = reinterpret_cast<double*>(smem_raw);
  double* vals = y + t.n_col + 1;
  unsigned char* cur = reinterpret_cast<unsigned char*>(vals + n_items);
  uint32_t* refs = reinterpret_cast<uint32_t*>(cur);
  cur += up8l(4 * n_items);
  uint32_t* ptr = reinterpret_cast<uint32_t*>(cur);
  cur += up8l(4 * (t.n_col + 1));
  uint32_t* lvl = reinterpret_cast<uint32_t*>(cur);

  {
    const LdltSolveItem* items = L.fwd_items + t.fwd_item_off;
    for (uint32_t q = tid; q < n_items; q += 256) {
      const LdltSolveItem it = items[q];
      vals[q] = Lx[it.lpos];
      refs[q] = it.ref;
    }
    for (uint32_t i = tid; i < t.n_col + 1; i += 256) ptr[i] = g_ptr[i];
    const uint32_t* g_lvl = L.col_lvl_ptr + t.lvl_off;
    for (uint32_t i = tid; i < t.n_lvl + 1; i += 256) lvl[i] = g_lvl[i];
    const uint32_t* fcptr = L.fwd_contrib_ptr + t.colptr_off;
    const uint32_t* scidx = L.scontrib_idx + t.scontrib_off;
    for (uint32_t i = tid; i < t.n_col; i += 256) {
      const uint32_t pj = L.col_perm[t.col_off + i];
      double acc = rhs[L.perm[pj]];
      for (uint32_t c = fcptr[i]; c < fcptr[i + 1]; ++c) acc -= scontrib[scidx[c]];
      y[i] = acc;
    }
  }
  __syncthreads();
  for (uint32_t l = 0; l < t.n_lvl; ++l) {
    const uint32_t beg = lvl[l], end = lvl[l + 1];
    for (uint32_t i = beg + tid; i < end; i += 256) {
      double acc = y[i];
      for (uint32_t q = ptr[i]; q < ptr[i + 1]; ++q) acc -= vals[q] * y[refs[q]];
      y[i] = acc;
    }
    __syncthreads();
  }
  // partial sums for rows owned by ancestor tasks
  const uint32_t* sptr = L.sext_ptr + t.sext_ptr_off;
  const LdltSolveItem* sitems = L.sext_items + t.sext_item_off;
  for (uint32_t x = tid; x < t.n_sext; x += 256) {
    double acc = 0.0;
    for (uint32_t q = sptr[x]; q < sptr[x + 1]; ++q) acc += Lx[sitems[q].lpos] * y[sitems[q].ref];
    scontrib[L.sext_dst[t.sext_off + x]] = acc;
  }
  for (uint32_t i = tid; i < t.n_col; i += 256) {
    const uint32_t pj = L.col_perm[t.col_off + i];
    zv[pj] = y[i] / D[pj];
  }
}

// ---------------------------------------------------------------------------
// Backward substitution Lᵀ x = z; result un-permuted.  Rows owned by ancestor tasks
// are final (earlier launch): their products are folded into the staged values and
// point at the constant-one slot x[n_col], so the level loop is uniform.
// LDS: x[n_col+1] f64 | vals[n_items] f64 | refs[n_items] u32 | ptr[n_col+1] u32 | lvl[n_lvl+1] u32
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ldlt_bwd_kernel(
    LdltDev L, uint32_t task_base, int n, const double* __restrict__ Lx, long long lx_stride,
    const double* __restrict__ zv, double* __restrict__ xg, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const LdltTask t = L.tasks[task_base + blockIdx.x];
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  Lx += static_cast<size_t>(b) * lx_stride;
  zv += static_cast<size_t>(b) * n;
  xg += static_cast<size_t>(b) * n;
  out += static_cast<size_t>(b) * n;

  const uint32_t* g_ptr = L.bwd_ptr + t.colptr_off;
  const uint32_t n_items = g_ptr[t.n_col];
  double* x = reinterpret_cast<double*>(smem_raw);
  double* vals = x + t.n_col + 1;
  unsigned char* cur = reinterpret_cast<unsigned char*>(vals + n_items);
  uint32_t* refs = reinterpret_cast<uint32_t*>(cur);
  cur += up8l(4 * n_items);
  uint32_t* ptr = reinterpret_cast<uint32_t*>(cur);
  cur += up8l(4 * (t.n_col + 1));
  uint32_t* lvl = reinterpret_cast<uint32_t*>(cur);

  {
    const LdltSolveItem* items = L.bwd_items + t.bwd_item_off;
    for (uint32_t q = tid; q < n_items; q += 256) {
      const LdltSolveItem it = items[q];
      const double lv = Lx[it.lpos];
      if (it.ref & 0x80000000u) {
        vals[q] = lv * xg[it.ref & 0x7fffffffu];
        refs[q] = t.n_col;
      } else {
        vals[q] = lv;
        refs[q] = it.ref;
      }
    }
    for (uint32_t i = tid; i < t.n_col + 1; i += 256) ptr[i] = g_ptr[i];
    const uint32_t* g_lvl = L.col_lvl_ptr + t.lvl_off;
    for (uint32_t i = tid; i < t.n_lvl + 1; i += 256) lvl[i] = g_lvl[i];
    for (uint32_t i = tid; i < t.n_col; i += 256) x[i] = zv[L.col_perm[t.col_off + i]];
    if (tid == 0) x[t.n_col] = 1.0;
  }
  __syncthreads();
  for (int l = static_cast<int>(t.n_lvl) - 1; l >= 0; --l) {
    const uint32_t beg = lvl[l], end = lvl[l + 1];
    for (uint32_t i = beg + tid; i < end; i += 256) {
      double acc = x[i];
      for (uint32_t q = ptr[i]; q < ptr[i + 1]; ++q) acc -= vals[q] * x[refs[q]];
      x[i] = acc;
    }
    __syncthreads();
  }
  for (uint32_t i = tid; i < t.n_col; i += 256) {
    const uint32_t pj = L.col_perm[t.col_off + i];
    xg[pj] = x[i];
    out[L.perm[pj]] = x[i];
  }
}

__global__ void ldlt_stats_reset_kernel(LdltStats* stats, int batch) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < batch) {
    stats[b].n_pos = stats[b].n_neg = stats[b].n_zero = stats[b].n_bad = 0;
    stats[b].min_abs_bits = 0x7ff0000000000000ull;
  }
}

}  // namespace slpx
