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

// x = L^-T D^-1 L^-1 rhs of the factors ldlt_dense_factor_kernel left in A.  LDS: dim doubles.
__global__ __launch_bounds__(kDenseThreads) void ldlt_dense_solve_kernel(int dim, const double* __restrict__ A_all,
                                                                         const double* __restrict__ rhs, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dense_smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const double* A = A_all + static_cast<size_t>(b) * dim * dim;
  double* x = reinterpret_cast<double*>(dense_smem);
  for (int i = tid; i < dim; i += kDenseThreads) x[i] = rhs[static_cast<size_t>(b) * dim + i];
  __syncthreads();
  for (int k = 0; k < dim; ++k) {  // L y = b, column by column
    const double xk = x[k];
    const double* colk = A + static_cast<size_t>(k) * dim;
    for (int i = k + 1 + tid; i < dim; i += kDenseThreads) x[i] = __builtin_fma(-colk[i], xk, x[i]);
    __syncthreads();
  }
  for (int i = tid; i < dim; i += kDenseThreads) x[i] = x[i] / A[static_cast<size_t>(i) * dim + i];
  __syncthreads();
  for (int k = dim - 1; k >= 0; --k) {  // L^T x = z: x_i -= L(k, i) x_k for i < k (row k of L)
    const double xk = x[k];
    for (int i = tid; i < k; i += kDenseThreads) x[i] = __builtin_fma(-A[static_cast<size_t>(i) * dim + k], xk, x[i]);
    __syncthreads();
  }
  for (int i = tid; i < dim; i += kDenseThreads) out[static_cast<size_t>(b) * dim + i] = x[i];
}

}  // namespace slpx
