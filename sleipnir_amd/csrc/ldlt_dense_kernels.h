// The dense branch of the regularized LDLᵀ (util/dense_regularized_ldlt.hpp:59-136): the reference factors the KKT
// system as a dense matrix when it is dense (interior_point.hpp:340-352: nnz >= 25 % — single shooting over hundreds
// of steps, small problems) — and so does this library where a column of L does not fit the LDS of a task
// (ldlt_symbolic.cpp: "a single column exceeds the LDS task budget" used to be a refusal).  One workgroup per
// problem, the matrix in memory (L2 at these sizes), column k and its scaled copy in LDS for the rank-1 update of
// step k, no pivoting: the delta / gamma policy loop around it (NewtonSystem::compute_impl,
// sparse_regularized_ldlt.hpp:64-152 = dense_regularized_ldlt.hpp:59-136) asks for the inertia, as everywhere.
// A slow answer instead of none: ~2 dim barriers and dim^3 / 3 multiply-adds on one CU.
#pragma once

#include <hip/hip_runtime.h>

#include "ldlt_kernels.h"

namespace slpx {

constexpr int kDenseThreads = 1024;

// lhs (lower CSC, `nnz` values per problem) + (delta on the first n_dec diagonal entries, -gamma on the others)
// -> A = L (unit lower, below the diagonal) and D (the diagonal), column-major dim x dim per problem;
// D, Lx (the strictly lower part column by column: LdltPlan::Lp of the dense plan) and the inertia counters.
// LDS: 2 dim doubles + 32 bytes.
__global__ __launch_bounds__(kDenseThreads) void ldlt_dense_factor_kernel(
    int dim, int n_dec, const int32_t* __restrict__ colptr, const int32_t* __restrict__ rowidx, int nnz,
    const double* __restrict__ lhs, const double* __restrict__ reg, double* __restrict__ A_all, double* __restrict__ D_all,
    double* __restrict__ Lx_all, long long lx_stride, LdltStats* __restrict__ stats_cur, LdltStats* __restrict__ stats_next) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dense_smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const double delta = reg[2 * b], gamma = reg[2 * b + 1];
  if (isnan(delta)) return;  // (a problem of the batch that is not part of this attempt)
  double* A = A_all + static_cast<size_t>(b) * dim * dim;
  const double* v = lhs + static_cast<size_t>(b) * nnz;
  double* s_u = reinterpret_cast<double*>(dense_smem);
  double* s_l = s_u + dim;
  int* s_cnt = reinterpret_cast<int*>(s_l + dim);
  unsigned long long* s_min = reinterpret_cast<unsigned long long*>(s_cnt + 4);
  const size_t total = static_cast<size_t>(dim) * dim;
  for (size_t i = tid; i < total; i += kDenseThreads) A[i] = 0.0;
  if (tid < 4) s_cnt[tid] = 0;
  if (tid == 0) *s_min = 0x7ff0000000000000ull;
  __syncthreads();
  for (int c = tid; c < dim; c += kDenseThreads) {
    double* col = A + static_cast<size_t>(c) * dim;
    for (int p = colptr[c]; p < colptr[c + 1]; ++p) col[rowidx[p]] += v[p];
    col[c] += c < n_dec ? delta : -gamma;
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  constexpr int kWaves = kDenseThreads / 64;
  for (int k = 0; k < dim; ++k) {
    double* colk = A + static_cast<size_t>(k) * dim;
    const double d = colk[k];
    const double inv = 1.0 / d;
    for (int i = k + 1 + tid; i < dim; i += kDenseThreads) {
      const double u = colk[i];
      s_u[i] = u;
      const double l = u * inv;
      s_l[i] = l;
      colk[i] = l;
    }
    __syncthreads();
    for (int j = k + 1 + wave; j < dim; j += kWaves) {
      double* colj = A + static_cast<size_t>(j) * dim;
      const double uj = s_u[j];
      for (int i = j + lane; i < dim; i += 64) colj[i] = __builtin_fma(-s_l[i], uj, colj[i]);
    }
    __syncthreads();
  }
  // D, the inertia (inertia.hpp:40-50), L in the plan's order
  double* D = D_all + static_cast<size_t>(b) * dim;
  for (int k = tid; k < dim; k += kDenseThreads) {
    const double u = A[static_cast<size_t>(k) * dim + k];
    D[k] = u;
    const double eps = 2.220446049250313e-16;
    if (u > eps) atomicAdd(&s_cnt[0], 1);
    else if (u < -eps) atomicAdd(&s_cnt[1], 1);
    else atomicAdd(&s_cnt[2], 1);
    if (u == 0.0 || !isfinite(u)) atomicAdd(&s_cnt[3], 1);
    else atomicMin(s_min, static_cast<unsigned long long>(__double_as_longlong(fabs(u))));
  }
  if (Lx_all != nullptr) {
    double* Lx = Lx_all + static_cast<size_t>(b) * lx_stride;
    for (int j = wave; j < dim; j += kWaves) {
      const size_t base = static_cast<size_t>(j) * (dim - 1) - (static_cast<size_t>(j) * (j - 1)) / 2;  // sum_{c<j} (dim - 1 - c)
      const double* colj = A + static_cast<size_t>(j) * dim;
      for (int i = j + 1 + lane; i < dim; i += 64) Lx[base + (i - j - 1)] = colj[i];
    }
  }
  __syncthreads();
  if (tid == 0) {
    stats_cur[b] = LdltStats{s_cnt[0], s_cnt[1], s_cnt[2], s_cnt[3], *s_min};
    if (stats_next != nullptr) stats_next[b] = LdltStats{0, 0, 0, 0, 0x7ff0000000000000ull};
  }
}

// The dense branch AS THE REFERENCE TAKES IT (interior_point.hpp:340-352: the KKT system is factored dense when its lower
// triangle fills a quarter of it or more — the small problems of its unit tests, single shooting): Eigen::LDLT
// (util/dense_regularized_ldlt.hpp:167), i.e. symmetric pivoting on the largest |diagonal| of the part not yet
// factored, left-looking — the diagonal the pivot is chosen from is NOT yet updated, as in Eigen's unblocked kernel —
// with Eigen's treatment of zero pivots (a zero pivot is an error only if something non-zero follows it).  The
// arithmetic is the textbook's, term by term in column order and without fused multiply-adds, so that D — and with it
// every decision of the regularization policy — is what the reference's CPU path computes on the same matrix.
// One workgroup per problem; `trans` (dim ints per problem): the transpositions.
// LDS: dim doubles + 64 bytes + the reduction scratch.
__global__ __launch_bounds__(kDenseThreads) void ldlt_dense_pivoted_factor_kernel(
    int dim, int n_dec, const int32_t* __restrict__ colptr, const int32_t* __restrict__ rowidx, int nnz,
    const double* __restrict__ lhs, const double* __restrict__ reg, double* __restrict__ A_all, int32_t* __restrict__ trans_all,
    double* __restrict__ D_all, LdltStats* __restrict__ stats_cur, LdltStats* __restrict__ stats_next) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) unsigned char dense_smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const double delta = reg[2 * b], gamma = reg[2 * b + 1];
  if (isnan(delta)) return;
  double* A = A_all + static_cast<size_t>(b) * dim * dim;
  int32_t* trans = trans_all + static_cast<size_t>(b) * dim;
  const double* v = lhs + static_cast<size_t>(b) * nnz;
  double* temp = reinterpret_cast<double*>(dense_smem);
  double* s_val = temp + dim;                       // [kDenseThreads / 64] wave maxima
  int* s_idx = reinterpret_cast<int*>(s_val + kDenseThreads / 64);
  int* s_flag = s_idx + kDenseThreads / 64;          // [0] pivot index, [1] ret, [2] found_zero_pivot, [3] done
  int* s_cnt = s_flag + 4;                           // inertia counters
  unsigned long long* s_min = reinterpret_cast<unsigned long long*>(s_cnt + 4);
  auto at = [&](int r, int c) -> double& { return A[static_cast<size_t>(c) * dim + r]; };
  const size_t total = static_cast<size_t>(dim) * dim;
  for (size_t i = tid; i < total; i += kDenseThreads) A[i] = 0.0;
  if (tid < 4) s_cnt[tid] = 0;
  if (tid == 0) {
    *s_min = 0x7ff0000000000000ull;
    s_flag[1] = 1;
    s_flag[2] = 0;
    s_flag[3] = 0;
  }
  __syncthreads();
  for (int c = tid; c < dim; c += kDenseThreads) {
    double* col = A + static_cast<size_t>(c) * dim;
    for (int p = colptr[c]; p < colptr[c + 1]; ++p) col[rowidx[p]] += v[p];
    col[c] += c < n_dec ? delta : -gamma;
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  for (int k = 0; k < dim; ++k) {
    // the largest |diagonal| of the rest, the first of equals (Eigen's maxCoeff)
    {
      double best = -1.0;
      int bi = dim;
      for (int j = k + tid; j < dim; j += kDenseThreads) {
        const double a = fabs(at(j, j));
        if (a > best) {
          best = a;
          bi = j;
        }
      }
      for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (ob > best || (ob == best && oi < bi)) {
          best = ob;
          bi = oi;
        }
      }
      if (lane == 0) {
        s_val[wave] = best;
        s_idx[wave] = bi;
      }
      __syncthreads();
      if (tid == 0) {
        double bb = s_val[0];
        int ii = s_idx[0];
        for (int w = 1; w < kDenseThreads / 64; ++w)
          if (s_val[w] > bb || (s_val[w] == bb && s_idx[w] < ii)) {
            bb = s_val[w];
            ii = s_idx[w];
          }
        s_flag[0] = ii;
        trans[k] = ii;
      }
      __syncthreads();
    }
    const int big = s_flag[0];
    if (big != k) {  // the symmetric transposition k <-> big on the lower triangle
      for (int c = tid; c < k; c += kDenseThreads) {
        const double t = at(k, c);
        at(k, c) = at(big, c);
        at(big, c) = t;
      }
      for (int r = big + 1 + tid; r < dim; r += kDenseThreads) {
        const double t = at(r, k);
        at(r, k) = at(r, big);
        at(r, big) = t;
      }
      for (int i = k + 1 + tid; i < big; i += kDenseThreads) {
        const double t = at(i, k);
        at(i, k) = at(big, i);
        at(big, i) = t;
      }
      if (tid == 0) {
        const double t = at(k, k);
        at(k, k) = at(big, big);
        at(big, big) = t;
      }
      __syncthreads();
    }
    const int rs = dim - k - 1;
    if (k > 0) {
      for (int c = tid; c < k; c += kDenseThreads) temp[c] = at(c, c) * at(k, c);
      __syncthreads();
      // row k (thread rs) and the rows below it, a thread each, their sums in column order
      for (int r = tid; r <= rs; r += kDenseThreads) {
        const int row = r == rs ? k : k + 1 + r;
        double sum = 0.0;
        for (int c = 0; c < k; ++c) sum += at(row, c) * temp[c];
        at(row, k) -= sum;
      }
      __syncthreads();
    }
    const double akk = at(k, k);
    const bool valid = fabs(akk) > 0.0;
    if (k == 0 && !valid) {  // (Eigen: the matrix is zero, or it is not factorizable this way)
      int nz = 0;
      for (size_t i = tid; i < total; i += kDenseThreads) {
        const int r = static_cast<int>(i % dim), c = static_cast<int>(i / dim);
        if (r > c && A[i] != 0.0) nz = 1;
      }
      if (nz) atomicAnd(&s_flag[1], 0);
      for (int j = tid; j < dim; j += kDenseThreads) trans[j] = j;
      __syncthreads();
      break;
    }
    if (rs > 0 && valid) {
      for (int r = tid; r < rs; r += kDenseThreads) at(k + 1 + r, k) /= akk;
    } else if (rs > 0) {
      int nz = 0;
      for (int r = tid; r < rs; r += kDenseThreads)
        if (at(k + 1 + r, k) != 0.0) nz = 1;
      if (nz) atomicAnd(&s_flag[1], 0);
    }
    if (tid == 0) {
      if (s_flag[2] && valid) s_flag[1] = 0;
      else if (!valid) s_flag[2] = 1;
    }
    __syncthreads();
  }
  double* D = D_all + static_cast<size_t>(b) * dim;
  for (int k = tid; k < dim; k += kDenseThreads) {
    const double u = at(k, k);
    D[k] = u;
    const double eps = 2.220446049250313e-16;
    if (u > eps) atomicAdd(&s_cnt[0], 1);
    else if (u < -eps) atomicAdd(&s_cnt[1], 1);
    else atomicAdd(&s_cnt[2], 1);
    if (u != 0.0 && isfinite(u)) atomicMin(s_min, static_cast<unsigned long long>(__double_as_longlong(fabs(u))));
    if (!isfinite(u)) atomicAnd(&s_flag[1], 0);
  }
  __syncthreads();
  if (tid == 0) {
    // n_bad: Eigen's info() != Success (the policy loop's "the decomposition failed", dense_regularized_ldlt.hpp:107-112)
    stats_cur[b] = LdltStats{s_cnt[0], s_cnt[1], s_cnt[2], s_flag[1] ? 0 : 1, *s_min};
    if (stats_next != nullptr) stats_next[b] = LdltStats{0, 0, 0, 0, 0x7ff0000000000000ull};
  }
}

// x = L^-T D^-1 L^-1 rhs of the factors ldlt_dense_factor_kernel left in A (trans == nullptr), or
// P^T L^-T D^-1 L^-1 P rhs of ldlt_dense_pivoted_factor_kernel's (Eigen::LDLT::solve: a pivot below the smallest
// normal number gives 0).  LDS: dim doubles.
__global__ __launch_bounds__(kDenseThreads) void ldlt_dense_solve_kernel(int dim, const double* __restrict__ A_all,
                                                                         const double* __restrict__ rhs, double* __restrict__ out,
                                                                         const int32_t* __restrict__ trans_all) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dense_smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const double* A = A_all + static_cast<size_t>(b) * dim * dim;
  double* x = reinterpret_cast<double*>(dense_smem);
  for (int i = tid; i < dim; i += kDenseThreads) x[i] = rhs[static_cast<size_t>(b) * dim + i];
  __syncthreads();
  if (trans_all != nullptr) {
    if (tid == 0) {
      const int32_t* trans = trans_all + static_cast<size_t>(b) * dim;
      for (int k = 0; k < dim; ++k) {
        const double t = x[k];
        x[k] = x[trans[k]];
        x[trans[k]] = t;
      }
    }
    __syncthreads();
  }
  for (int k = 0; k < dim; ++k) {  // L y = b, column by column
    const double xk = x[k];
    const double* colk = A + static_cast<size_t>(k) * dim;
    for (int i = k + 1 + tid; i < dim; i += kDenseThreads) x[i] = __builtin_fma(-colk[i], xk, x[i]);
    __syncthreads();
  }
  for (int i = tid; i < dim; i += kDenseThreads) {
    const double d = A[static_cast<size_t>(i) * dim + i];
    x[i] = (trans_all == nullptr || fabs(d) > 2.2250738585072014e-308) ? x[i] / d : 0.0;
  }
  __syncthreads();
  for (int k = dim - 1; k >= 0; --k) {  // L^T x = z: x_i -= L(k, i) x_k for i < k (row k of L)
    const double xk = x[k];
    for (int i = tid; i < k; i += kDenseThreads) x[i] = __builtin_fma(-A[static_cast<size_t>(i) * dim + k], xk, x[i]);
    __syncthreads();
  }
  if (trans_all != nullptr) {
    if (tid == 0) {
      const int32_t* trans = trans_all + static_cast<size_t>(b) * dim;
      for (int k = dim - 1; k >= 0; --k) {
        const double t = x[k];
        x[k] = x[trans[k]];
        x[trans[k]] = t;
      }
    }
    __syncthreads();
  }
  for (int i = tid; i < dim; i += kDenseThreads) out[static_cast<size_t>(b) * dim + i] = x[i];
}

}  // namespace slpx
