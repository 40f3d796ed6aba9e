// Device side of restoration.hpp: the reduced Newton-KKT system of the restoration problem on the outer problem's
// pattern, and the row-local rest of its interior-point iteration.  Index conventions: `e` in [0, M), M = 2 m_e + 2 m_i,
// walks the extra variables [p_e | n_e | p_i | n_i] — and the inequality rows "variable >= 0" that belong to them, with
// their slacks sx[e] and duals zx[e]; block 0 of the inequality rows (c_i(x) - p_i + n_i >= 0) has its slack and dual
// in the outer system's s, z.
#include "restoration.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>

#include "coherent.h"
#include "ipm_reduce.h"

namespace slpx {

struct FrDevice::Args {
  int n, m_e, m_i;
  int off_Hf, off_Hc;
  double rho;
  KktDev K;
  // the outer system's buffers
  const double* V;
  double* Vt;        // the trial V (values at the trial x)
  double* in;        // [x | y | z_0] as the tape reads it
  double* trial_in;  // the trial x (and whatever else of the tape's input)
  double *s0, *y, *z0, *p, *ps0, *pz0;
  // fixed for the phase
  const double *xr, *w, *g_outer, *s_outer, *scales;
  // the rest of the restoration iterate and of its direction
  double *pn, *sx, *zx, *dpn, *psx, *pzx;
  double *soc_ce, *soc_c0, *soc_x;
  // the look-ahead iterate (expand with ahead): s_0, y, z_0 beside the trial input, the rest of it
  double *s0_t, *y_t, *z0_t, *pn_t, *sx_t, *zx_t;
};

namespace {

constexpr double kFrRho = 1e3;  // feasibility_restoration.hpp:391

// One equality row j of the restoration problem, c_e(x)_j - p_e_j + n_e_j = 0, with its two bound rows p_e_j >= 0,
// n_e_j >= 0: Sigma of the bound rows, the right-hand sides of the p_e and n_e rows of the Newton-KKT system
// (interior_point.hpp:441-448 on the restoration callbacks), the constraint value (or its second-order-correction
// accumulator) and the (c_i' - s') terms of the two bound rows.
struct FrEq {
  double S1, S2, rpe, rne, ce, c1, c2, s1, z1, s2, z2, i1, i2;
};
__device__ __forceinline__ FrEq fr_eq_row(const FrDevice::Args& A, int j, double mu, bool soc) {
  FrEq r;
  const int me = A.m_e;
  r.s1 = A.sx[j];
  r.z1 = A.zx[j];
  r.s2 = A.sx[me + j];
  r.z2 = A.zx[me + j];
  const double pe = A.pn[j], ne = A.pn[me + j], yj = A.y[j];
  r.i1 = 1.0 / r.s1;
  r.i2 = 1.0 / r.s2;
  r.S1 = r.i1 * r.z1;
  r.S2 = r.i2 * r.z2;
  double t1, t2;
  if (soc) {  // :611-616: mu S^-1 e - Sigma (c_i - s)^soc
    r.c1 = A.soc_x[j];
    r.c2 = A.soc_x[me + j];
    r.ce = A.soc_ce[j];
    t1 = mu * r.i1 - r.S1 * r.c1;
    t2 = mu * r.i2 - r.S2 * r.c2;
  } else {  // :444-447: -Sigma c_i + mu S^-1 e + z
    r.c1 = pe - r.s1;
    r.c2 = ne - r.s2;
    r.ce = (A.V[A.K.off_ce + j] - pe) + ne;
    t1 = (-(r.S1 * pe) + mu * r.i1) + r.z1;
    t2 = (-(r.S2 * ne) + mu * r.i2) + r.z2;
  }
  // -g' + A_e'^T y + A_i'^T t: the cost's gradient is rho, the column of p_e has -1 in row j of A_e' and 1 in its bound row
  r.rpe = (-A.rho - yj) + t1;
  r.rne = (-A.rho + yj) + t2;
  return r;
}

// One inequality row r of the outer problem inside the restoration problem, c_i(x)_r - p_i_r + n_i_r >= 0 (block 0),
// with the bound rows of p_i_r and n_i_r (blocks 3, 4).
// (u3 = -rho + t3, u4 = -rho + t4: the right-hand sides of the p_i and n_i rows are r_pi = u3 - t0, r_ni = u4 + t0.
// Sigma_0 = z_0 / s_0 of a row whose slack a run of short steps has driven to 1e-10 is 1e21, and t0 with it: every
// formula below is arranged so that no two terms of that size are subtracted from each other — eliminating the rows
// in closed form is then BETTER conditioned than factoring them)
struct FrIn {
  double sig, S3, S4, a3, a4, det, u3, u4, t0, c0, c3, c4, s0, z0, s3, z3, s4, z4, i0, i3, i4;
};
__device__ __forceinline__ FrIn fr_in_row(const FrDevice::Args& A, int r, double delta, double mu, bool soc) {
  FrIn q;
  const int e3 = 2 * A.m_e + r, e4 = 2 * A.m_e + A.m_i + r;
  q.s0 = A.s0[r];
  q.z0 = A.z0[r];
  q.s3 = A.sx[e3];
  q.z3 = A.zx[e3];
  q.s4 = A.sx[e4];
  q.z4 = A.zx[e4];
  const double pi = A.pn[e3], ni = A.pn[e4];
  q.i0 = 1.0 / q.s0;
  q.i3 = 1.0 / q.s3;
  q.i4 = 1.0 / q.s4;
  q.sig = q.i0 * q.z0;
  q.S3 = q.i3 * q.z3;
  q.S4 = q.i4 * q.z4;
  double t3, t4;
  if (soc) {
    q.c0 = A.soc_c0[r];
    q.c3 = A.soc_x[e3];
    q.c4 = A.soc_x[e4];
    q.t0 = mu * q.i0 - q.sig * q.c0;
    t3 = mu * q.i3 - q.S3 * q.c3;
    t4 = mu * q.i4 - q.S4 * q.c4;
  } else {
    const double ci = (A.V[A.K.off_ci + r] - pi) + ni;
    q.c0 = ci - q.s0;
    q.c3 = pi - q.s3;
    q.c4 = ni - q.s4;
    q.t0 = (-(q.sig * ci) + mu * q.i0) + q.z0;
    t3 = (-(q.S3 * pi) + mu * q.i3) + q.z3;
    t4 = (-(q.S4 * ni) + mu * q.i4) + q.z4;
  }
  // the column of p_i has -1 in row r of block 0 and 1 in its bound row; n_i: 1 and 1
  q.u3 = -A.rho + t3;
  q.u4 = -A.rho + t4;
  q.a3 = q.S3 + delta;
  q.a4 = q.S4 + delta;
  q.det = q.sig * (q.a3 + q.a4) + q.a3 * q.a4;
  return q;
}
__device__ __forceinline__ double fr_sigma_eff(const FrIn& q) { return q.sig * q.a3 * q.a4 / q.det; }
// what row r contributes to the x rows of the reduced right-hand side, per unit of A_i(r, :)
// (t0 + Sigma_0 (a4 r_pi - a3 r_ni) / det with the two t0 terms combined: 1 - Sigma_0 (a3 + a4) / det = a3 a4 / det)
__device__ __forceinline__ double fr_rhs_mult(const FrIn& q) {
  return (q.a3 * q.a4 * q.t0 + q.sig * (q.a4 * q.u3 - q.a3 * q.u4)) / q.det;
}

// ---------------------------------------------------------------------------------------------------------------
// The reduced system into the outer system's lhs / rhs arrays (restoration.hpp's header comment)
// ---------------------------------------------------------------------------------------------------------------
// (gridDim.y == 2: the systems of TWO regularizations side by side — blockIdx.y = 1 takes delta_b, lhs_b, rhs_b)
__global__ __launch_bounds__(256) void fr_build_kernel(FrDevice::Args A, const int32_t* __restrict__ diag_of, double delta, double mu,
                                                       int soc, int rhs_only, double* __restrict__ lhs, double* __restrict__ rhs,
                                                       double delta_b, double* __restrict__ lhs_b, double* __restrict__ rhs_b) {
  if (blockIdx.y == 1) {
    delta = delta_b;
    lhs = lhs_b;
    rhs = rhs_b;
  }
  const KktDev& K = A.K;
  const double* V = A.V;
  const int stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  auto not_Hf = [&](int src) { return src < A.off_Hf || src >= A.off_Hc; };  // the restoration cost has its own Hessian
  if (!rhs_only)
    for (int k = t0; k < K.nnz_lhs; k += stride) {
      const int f = K.fast_src[k];
      double v = 0.0;
      if (f >= 0) {
        v = not_Hf(f) ? V[f] : 0.0;
      } else if (f == -2) {
        double direct = 0.0, prod = 0.0;
        for (int d = K.dptr[k]; d < K.dptr[k + 1]; ++d) {
          const int src = K.dsrc[d];
          if (not_Hf(src)) direct += V[src];
        }
        for (int q = K.pptr[k]; q < K.pptr[k + 1]; ++q) {
          const FrIn row = fr_in_row(A, K.pr[q], delta, mu, false);
          prod += (V[K.pa[q]] * fr_sigma_eff(row)) * V[K.pb[q]];
        }
        v = direct + prod;
      }
      const int d = diag_of[k];
      if (d >= 0) {
        if (d < A.n) {
          v += A.w[d];  // zeta D_R
        } else {
          const FrEq e = fr_eq_row(A, d - A.n, mu, false);
          v = -(1.0 / (e.S1 + delta) + 1.0 / (e.S2 + delta));
        }
      }
      lhs[k] = v;
    }
  for (int j = t0; j < K.dim; j += stride) {
    if (j >= A.n) {
      const FrEq e = fr_eq_row(A, j - A.n, mu, soc != 0);
      rhs[j] = (-e.ce + e.rpe / (e.S1 + delta)) - e.rne / (e.S2 + delta);
      continue;
    }
    const double* Ae = V + K.off_Ae;
    const double* Ai = V + K.off_Ai;
    double aey = 0.0, ait = 0.0;
    for (int q = K.ae_colptr[j]; q < K.ae_colptr[j + 1]; ++q) aey += Ae[q] * A.y[K.ae_rowidx[q]];
    for (int q = K.ai_colptr[j]; q < K.ai_colptr[j + 1]; ++q)
      ait += Ai[q] * fr_rhs_mult(fr_in_row(A, K.ai_rowidx[q], delta, mu, soc != 0));
    const double g = A.w[j] * (A.in[j] - A.xr[j]);  // zeta D_R (x - x_R)
    rhs[j] = (-g + aey) + ait;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// p = (dx, w) -> dp_e, dn_e, dp_i, dn_i, p_s, p_z of the five inequality blocks (interior_point.hpp:470-481 on the
// restoration problem), the step sizes (fraction_to_the_boundary_rule.hpp:19-43) and the directional derivative
// (:508-509) over ALL of them, the smallest eliminated pivot, and the first trial x.  One workgroup.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kFrExpandThreads = 256, kFrExpandMaxBlocks = 16;
__global__ __launch_bounds__(kFrExpandThreads) void fr_expand_kernel(FrDevice::Args A, double delta, double mu, double tau, int soc, int ahead,
                                                                     double* __restrict__ alpha_dev, FrDirOut* __restrict__ out,
                                                                     double* __restrict__ partial, unsigned int* __restrict__ sync,
                                                                     unsigned int generation) {
  // A handful of workgroups, a row per lane (one workgroup walking 4000 rows of a dozen loads and six divisions each
  // took 27 us): each reduces its share, the last one through folds the shares in workgroup order and raises the
  // launch's generation number; everybody waits for that — the grid is small enough to be resident at once — and
  // then moves ITS OWN rows on by the step sizes (nothing but the four scalars crosses workgroups).
  __shared__ double scratch[(kFrExpandThreads / 64 + 1) * 4];
  __shared__ int last;
  const KktDev& K = A.K;
  const int me = A.m_e, mi = A.m_i, n = A.n;
  const int stride = gridDim.x * kFrExpandThreads, t0 = blockIdx.x * kFrExpandThreads + threadIdx.x;
  double acc[4] = {1.0, 1.0, 0.0, 1e300};  // alpha_max, alpha_z, D_phi, smallest eliminated pivot
  auto ftb = [&](double s, double ps, double z, double pz) {
    if (ps < 0.0) acc[0] = fmin(acc[0], -tau / ps * s);
    if (pz < 0.0) acc[1] = fmin(acc[1], -tau / pz * z);
  };
  for (int j = t0; j < me; j += stride) {
    const FrEq e = fr_eq_row(A, j, mu, soc != 0);
    const double w = A.p[n + j];
    const double d1 = e.S1 + delta, d2 = e.S2 + delta;
    const double dpe = (e.rpe + w) / d1, dne = (e.rne - w) / d2;
    const double ps1 = e.c1 + dpe, ps2 = e.c2 + dne;
    const double pz1 = (mu * e.i1 - e.z1) - e.S1 * ps1, pz2 = (mu * e.i2 - e.z2) - e.S2 * ps2;
    A.dpn[j] = dpe;
    A.dpn[me + j] = dne;
    A.psx[j] = ps1;
    A.psx[me + j] = ps2;
    A.pzx[j] = pz1;
    A.pzx[me + j] = pz2;
    ftb(e.s1, ps1, e.z1, pz1);
    ftb(e.s2, ps2, e.z2, pz2);
    acc[2] += A.rho * dpe + A.rho * dne;
    acc[2] -= mu * (e.i1 * ps1) + mu * (e.i2 * ps2);
    acc[3] = fmin(acc[3], fmin(d1, d2));
  }
  for (int r = t0; r < mi; r += stride) {
    double aidx = 0.0;
    for (int q = K.ai_rowptr[r]; q < K.ai_rowptr[r + 1]; ++q) aidx += A.V[K.ai_src[q]] * A.p[K.ai_col[q]];
    const FrIn f = fr_in_row(A, r, delta, mu, soc != 0);
    const double a = f.sig + f.a3;
    // [a -sig; -sig b] [dp_i; dn_i] = [r_pi + sig q; r_ni - sig q], q = A_i(r, :) dx, with the sig^2 q terms of the
    // solution cancelled by hand; and p_s of block 0, c0 + q - dp_i + dn_i, likewise (q - (dp_i - dn_i) is q a3 a4 / det - ...)
    const double rsum = f.u3 + f.u4;  // r_pi + r_ni
    const double dpi = (f.a4 * ((f.u3 - f.t0) + f.sig * aidx) + f.sig * rsum) / f.det;
    const double dni = (f.a3 * ((f.u4 + f.t0) - f.sig * aidx) + f.sig * rsum) / f.det;
    const double ps0 = f.c0 + (f.a3 * f.a4 * aidx + (f.a3 + f.a4) * f.t0 - f.a4 * f.u3 + f.a3 * f.u4) / f.det;
    const double ps3 = f.c3 + dpi, ps4 = f.c4 + dni;
    const double pz0 = (mu * f.i0 - f.z0) - f.sig * ps0, pz3 = (mu * f.i3 - f.z3) - f.S3 * ps3,
                 pz4 = (mu * f.i4 - f.z4) - f.S4 * ps4;
    const int e3 = 2 * me + r, e4 = 2 * me + mi + r;
    A.dpn[e3] = dpi;
    A.dpn[e4] = dni;
    A.ps0[r] = ps0;
    A.pz0[r] = pz0;
    A.psx[e3] = ps3;
    A.pzx[e3] = pz3;
    A.psx[e4] = ps4;
    A.pzx[e4] = pz4;
    ftb(f.s0, ps0, f.z0, pz0);
    ftb(f.s3, ps3, f.z3, pz3);
    ftb(f.s4, ps4, f.z4, pz4);
    acc[2] += A.rho * dpi + A.rho * dni;
    acc[2] -= mu * (f.i0 * ps0) + mu * (f.i3 * ps3) + mu * (f.i4 * ps4);
    acc[3] = fmin(acc[3], fmin(a, f.det / a));  // (the second pivot, b - sig^2 / a)
  }
  for (int j = t0; j < n; j += stride) acc[2] += (A.w[j] * (A.in[j] - A.xr[j])) * A.p[j];
  const int ops[4] = {IPM_MIN, IPM_MIN, IPM_SUM, IPM_MIN};
  block_reduce<4, kFrExpandThreads>(acc, ops, scratch);
  if (threadIdx.x < 4) coherent_store(&partial[blockIdx.x * 4 + threadIdx.x], scratch[(kFrExpandThreads / 64) * 4 + threadIdx.x], true);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int old = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = old + 1 == gridDim.x;
    if (last) __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (last) {
    if (threadIdx.x < 64) {
      // lane = (workgroup, quantity): sixteen workgroups' shares in ONE trip to memory, then combined over the
      // workgroups by a butterfly in fixed order (one lane per quantity walking them was sixteen dependent trips)
      static_assert(kFrExpandMaxBlocks == 16, "64 lanes = 16 workgroups x 4 quantities");
      const int q = threadIdx.x & 3, b = threadIdx.x >> 2;
      const int op = q == 2 ? IPM_SUM : IPM_MIN;
      double v = b < static_cast<int>(gridDim.x) ? coherent_load(&partial[b * 4 + q], true) : (op == IPM_SUM ? 0.0 : 1e300);
      for (int off = 4; off < 64; off <<= 1) v = ipm_combine(op, v, __shfl_xor(v, off));
      if (threadIdx.x < 4) {
        coherent_store(&alpha_dev[q], v, true);
        reinterpret_cast<double*>(out)[q] = v;  // alpha_max, alpha_z, D_phi, eliminated_min_pivot
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(sync + 16, generation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) {
    unsigned int spins = 0;
    while (__hip_atomic_load(sync + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != generation) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) break;  // (never expected: the launch ends with whatever step sizes it finds)
    }
  }
  __syncthreads();
  const double alpha = coherent_load(&alpha_dev[0], true), alpha_z = coherent_load(&alpha_dev[1], true);
  for (int j = t0; j < n; j += stride) A.trial_in[j] = A.in[j] + alpha * A.p[j];
  if (ahead) {  // interior_point.hpp:775-801 for the full step, into the look-ahead buffers: every lane the rows it expanded
    constexpr double kappa = 1e10;
    auto clamp_z = [&](double zn, double sn) {
      const double lo = 1.0 / kappa * mu / sn, hi = kappa * mu / sn;
      return zn < lo ? lo : (zn > hi ? hi : zn);
    };
    auto bound_row = [&](int e) {
      A.pn_t[e] = A.pn[e] + alpha * A.dpn[e];
      const double sn = A.sx[e] + alpha * A.psx[e];
      A.sx_t[e] = sn;
      A.zx_t[e] = clamp_z(A.zx[e] + alpha_z * A.pzx[e], sn);
    };
    for (int j = t0; j < me; j += stride) {
      const double v = A.y[j] + alpha_z * (-A.p[n + j]);
      A.y_t[j] = v;
      A.trial_in[n + j] = v;
      bound_row(j);
      bound_row(me + j);
    }
    for (int r = t0; r < mi; r += stride) {
      const double sn = A.s0[r] + alpha * A.ps0[r];
      const double zn = clamp_z(A.z0[r] + alpha_z * A.pz0[r], sn);
      A.s0_t[r] = sn;
      A.z0_t[r] = zn;
      A.trial_in[n + me + r] = zn;
      bound_row(2 * me + r);
      bound_row(2 * me + mi + r);
    }
  }
}

// Filter entry (filter.hpp:30-60) of the trial point X + alpha dX of the restoration problem; c_e, c_i of the outer
// problem at the trial x are in Vt (a value sweep).  alpha < 0: the device's alpha_max.
__global__ __launch_bounds__(kIpmThreads) void fr_trial_metrics_kernel(FrDevice::Args A, double alpha, const double* __restrict__ alpha_dev,
                                                                       IpmTrialOut* __restrict__ out,
                                                                       unsigned long long* __restrict__ seq_dev,
                                                                       volatile unsigned long long* seq_host) {
  __shared__ double scratch[17 * 4];
  const KktDev& K = A.K;
  const int tid = threadIdx.x, me = A.m_e, mi = A.m_i, n = A.n;
  if (alpha < 0.0) alpha = alpha_dev[0];
  double acc[4] = {0.0, 0.0, 0.0, 1.0};  // cost, violation, log sum, finite
  for (int j = tid; j < n; j += kIpmThreads) {
    const double d = A.trial_in[j] - A.xr[j];
    acc[0] += 0.5 * (A.w[j] * (d * d));
  }
  auto bound_row = [&](int e, double v) {  // the row "variable e >= 0" at its trial value v
    const double st = A.sx[e] + alpha * A.psx[e];
    acc[1] += fabs(v - st);
    acc[2] += log(st);
  };
  for (int j = tid; j < me; j += kIpmThreads) {
    const double pe = A.pn[j] + alpha * A.dpn[j], ne = A.pn[me + j] + alpha * A.dpn[me + j];
    const double c = A.Vt[K.off_ce + j];
    acc[0] += A.rho * pe + A.rho * ne;
    acc[1] += fabs((c - pe) + ne);
    bound_row(j, pe);
    bound_row(me + j, ne);
    if (!isfinite(c)) acc[3] = 0.0;
  }
  for (int r = tid; r < mi; r += kIpmThreads) {
    const int e3 = 2 * me + r, e4 = 2 * me + mi + r;
    const double pi = A.pn[e3] + alpha * A.dpn[e3], ni = A.pn[e4] + alpha * A.dpn[e4];
    const double c = A.Vt[K.off_ci + r];
    const double st = A.s0[r] + alpha * A.ps0[r];
    acc[0] += A.rho * pi + A.rho * ni;
    acc[1] += fabs(((c - pi) + ni) - st);
    acc[2] += log(st);
    bound_row(e3, pi);
    bound_row(e4, ni);
    if (!isfinite(c)) acc[3] = 0.0;
  }
  const int ops[4] = {IPM_SUM, IPM_SUM, IPM_SUM, IPM_MIN};
  block_reduce<4, kIpmThreads>(acc, ops, scratch);
  if (tid == 0) {
    out->f = acc[0];
    out->viol = acc[1];
    out->logsum = acc[2];
    out->finite = (acc[3] != 0.0 && isfinite(acc[0])) ? 1.0 : 0.0;
    ipm_publish(seq_dev, seq_host);
  }
}

__global__ __launch_bounds__(256) void fr_trial_point_kernel(int n, const double* __restrict__ x, const double* __restrict__ p, double alpha,
                                                             double* __restrict__ trial_x) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) trial_x[j] = x[j] + alpha * p[j];
}

// interior_point.hpp:775-801 on the whole restoration iterate
__global__ __launch_bounds__(256) void fr_commit_kernel(FrDevice::Args A, double alpha, double alpha_z, double mu) {
  const int stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = A.n, me = A.m_e, mi = A.m_i, M = 2 * me + 2 * mi;
  constexpr double kappa = 1e10;
  auto clamp_z = [&](double zn, double sn) {
    const double lo = 1.0 / kappa * mu / sn, hi = kappa * mu / sn;
    return zn < lo ? lo : (zn > hi ? hi : zn);
  };
  for (int j = t0; j < n; j += stride) A.in[j] = A.in[j] + alpha * A.p[j];
  for (int r = t0; r < me; r += stride) {
    const double v = A.y[r] + alpha_z * (-A.p[n + r]);
    A.y[r] = v;
    A.in[n + r] = v;
  }
  for (int r = t0; r < mi; r += stride) {
    const double sn = A.s0[r] + alpha * A.ps0[r];
    const double zn = clamp_z(A.z0[r] + alpha_z * A.pz0[r], sn);
    A.s0[r] = sn;
    A.z0[r] = zn;
    A.in[n + me + r] = zn;
  }
  for (int e = t0; e < M; e += stride) {
    A.pn[e] = A.pn[e] + alpha * A.dpn[e];
    const double sn = A.sx[e] + alpha * A.psx[e];
    A.sx[e] = sn;
    A.zx[e] = clamp_z(A.zx[e] + alpha_z * A.pzx[e], sn);
  }
}

// interior_point.hpp:590-600 on the restoration problem's rows; the trial values of the outer c_e, c_i are in Vt,
// the trial point is X + alpha dX with the direction in place
__global__ __launch_bounds__(256) void fr_soc_accumulate_kernel(FrDevice::Args A, double alpha, int first) {
  const KktDev& K = A.K;
  const int stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  const int me = A.m_e, mi = A.m_i;
  auto bound_row = [&](int e, double v_now, double v_trial) {
    const double prev = first ? v_now - A.sx[e] : A.soc_x[e];
    A.soc_x[e] = alpha * prev + (v_trial - (A.sx[e] + alpha * A.psx[e]));
  };
  for (int j = t0; j < me; j += stride) {
    const double pe = A.pn[j], ne = A.pn[me + j];
    const double pet = pe + alpha * A.dpn[j], net = ne + alpha * A.dpn[me + j];
    const double prev = first ? (A.V[K.off_ce + j] - pe) + ne : A.soc_ce[j];
    A.soc_ce[j] = alpha * prev + ((A.Vt[K.off_ce + j] - pet) + net);
    bound_row(j, pe, pet);
    bound_row(me + j, ne, net);
  }
  for (int r = t0; r < mi; r += stride) {
    const int e3 = 2 * me + r, e4 = 2 * me + mi + r;
    const double pi = A.pn[e3], ni = A.pn[e4];
    const double pit = pi + alpha * A.dpn[e3], nit = ni + alpha * A.dpn[e4];
    const double prev = first ? ((A.V[K.off_ci + r] - pi) + ni) - A.s0[r] : A.soc_c0[r];
    A.soc_c0[r] = alpha * prev + (((A.Vt[K.off_ci + r] - pit) + nit) - (A.s0[r] + alpha * A.ps0[r]));
    bound_row(e3, pi, pit);
    bound_row(e4, ni, nit);
  }
}

__global__ __launch_bounds__(256) void fr_copy_direction_kernel(int dim, int mi, int M, const double* __restrict__ p, const double* __restrict__ ps0,
                                                                const double* __restrict__ pz0, const double* __restrict__ dpn,
                                                                const double* __restrict__ psx, const double* __restrict__ pzx,
                                                                double* __restrict__ p_to, double* __restrict__ ps0_to, double* __restrict__ pz0_to,
                                                                double* __restrict__ dpn_to, double* __restrict__ psx_to, double* __restrict__ pzx_to) {
  const int stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  for (int j = t0; j < dim; j += stride) p_to[j] = p[j];
  for (int r = t0; r < mi; r += stride) {
    ps0_to[r] = ps0[r];
    pz0_to[r] = pz0[r];
  }
  for (int e = t0; e < M; e += stride) {
    dpn_to[e] = dpn[e];
    psx_to[e] = psx[e];
    pzx_to[e] = pzx[e];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Error norms (kkt_error.hpp:92-146, un-scaled :216-251), filter entry, local-infeasibility and divergence quantities
// of the restoration problem at the freshly swept iterate — the 23 quantities of ipm_err (ipm_kernels.h) with the
// rows and columns of p, n added in closed form — and the outer problem's filter quantities at (x, s_0).
// Workgroups of 256 lanes reduce their slices, the last one through folds the partials in workgroup order.
// ---------------------------------------------------------------------------------------------------------------
namespace fr_err {
enum {
  DUAL_U, SZ_MAX_U, CE_U, CIS_U, Y1_U, Z1_U, DUAL, SZ_MIN, SZ_MAX, CE, CIS, Y1, Z1, VIOL, LOGSUM,
  AETCE, CESQ, AITCP, CPSQ, XINF, SINF, FINITE, CIPOS,
  COST, VIOL_O, LOGSUM_O, DPHI_O, EMIN0, NQ
};
}
#define SLPX_FR_ERR_OPS                                                                               \
  {IPM_MAX, IPM_MAX, IPM_MAX, IPM_MAX, IPM_SUM, IPM_SUM, IPM_MAX, IPM_MIN, IPM_MAX, IPM_MAX, IPM_MAX, \
   IPM_SUM, IPM_SUM, IPM_SUM, IPM_SUM, IPM_SUM, IPM_SUM, IPM_SUM, IPM_SUM, IPM_MAX, IPM_MAX, IPM_MIN, \
   IPM_MIN, IPM_SUM, IPM_SUM, IPM_SUM, IPM_SUM, IPM_MIN}
constexpr int kFrErrThreads = 256;

// (n_err_blocks < gridDim.x: the tape's separable sums — which make f — ride as the workgroups behind them, one each)
struct FrSumsRide {
  int n_err_blocks = 0;
  const NlpStructure::SumReduce* red = nullptr;
  const double* tape_scales = nullptr;
  double* Vw = nullptr;
};
__global__ __launch_bounds__(kFrErrThreads) void fr_errors_kernel(FrDevice::Args A, int nV, int check_all_V, double mu_outer,
                                                                  double* __restrict__ partial, unsigned int* __restrict__ done,
                                                                  FrErrOut* __restrict__ out, unsigned long long* __restrict__ seq_dev,
                                                                  volatile unsigned long long* seq_host, FrSumsRide R) {
  using namespace fr_err;
  static_assert(kFrErrThreads == 256 && NQ <= 32, "the fold deals (slice, quantity) pairs to 8 x 32 lanes");
  __shared__ double scratch[(kFrErrThreads / 64 + 1) * NQ];
  __shared__ double fold[9 * NQ];
  __shared__ int last;
  const KktDev& K = A.K;
  const double* V = A.V;
  const int me = A.m_e, mi = A.m_i, n = A.n;
  const int ops[NQ] = SLPX_FR_ERR_OPS;
  const int n_err_blocks = R.n_err_blocks;
  const bool sum_block = static_cast<int>(blockIdx.x) >= n_err_blocks;
  if (sum_block) {
    const NlpStructure::SumReduce r = R.red[blockIdx.x - n_err_blocks];
    const int tid = threadIdx.x;
    double a = 0.0;
    if (tid < 64)
      for (int k = tid; k < r.count; k += 64) a += V[r.src_off + k];
    if (tid < 64) scratch[tid] = a;
    __syncthreads();
    for (int w = 32; w > 0; w >>= 1) {
      if (tid < w) scratch[tid] += scratch[tid + w];
      __syncthreads();
    }
    if (tid == 0) coherent_store(&R.Vw[r.dst], (r.scale_idx >= 0 ? R.tape_scales[r.scale_idx] : 1.0) * scratch[0], true);
  }
  double acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = ops[q] == IPM_MIN ? 1.0 : 0.0;
  acc[SZ_MIN] = 1e300;
  acc[EMIN0] = 1e300;
  // (a workgroup of the sums has no rows: its lanes start past every range)
  const int stride = n_err_blocks * kFrErrThreads, t0 = sum_block ? (1 << 30) : blockIdx.x * kFrErrThreads + threadIdx.x;
  const double* d_ce = A.scales + 1;
  const double* d_ci = A.scales + 1 + me;
  const double* ce = V + K.off_ce;
  const double* ci = V + K.off_ci;
  const double* Ae = V + K.off_Ae;
  const double* Ai = V + K.off_Ai;
  auto dual_entry = [&](double d, double du) {
    acc[DUAL] = fmax(acc[DUAL], fabs(d));
    acc[DUAL_U] = fmax(acc[DUAL_U], fabs(du));
  };
  auto var_entry = [&](double v) {
    acc[XINF] = fmax(acc[XINF], fabs(v));
    if (!isfinite(v)) acc[FINITE] = 0.0;
  };
  // an inequality row with slack sr, dual zr, value c and row scale dr (1 for the bound rows)
  auto ineq_row = [&](double c, double sr, double zr, double dr) {
    const double inv = 1.0 / dr, su = inv * sr, zu = dr * zr;
    acc[CIS] = fmax(acc[CIS], fabs(c - sr));
    acc[CIS_U] = fmax(acc[CIS_U], fabs(inv * c - su));
    acc[SZ_MIN] = fmin(acc[SZ_MIN], sr * zr);
    acc[SZ_MAX] = fmax(acc[SZ_MAX], sr * zr);
    acc[SZ_MAX_U] = fmax(acc[SZ_MAX_U], fabs(su * zu));
    acc[Z1] += fabs(zr);
    acc[Z1_U] += fabs(zu);
    const double cp = fmin(c, 0.0);
    acc[CPSQ] += cp * cp;
    acc[VIOL] += fabs(c - sr);
    acc[LOGSUM] += log(sr);
    acc[SINF] = fmax(acc[SINF], fabs(sr));
    if (!isfinite(sr) || !isfinite(c)) acc[FINITE] = 0.0;
    if (!(c > 0.0)) acc[CIPOS] = 0.0;
  };
  // columns of x: EIGHT LANES per column, an entry of the column each (walked by one lane a column of A_e is 5-9
  // dependent trips to memory; this is two — ipm_error_accumulate does the same), summed by DPP in lane order
  const int lane8 = t0 & 7;
  for (int j = t0 >> 3; j < n; j += stride >> 3) {
    double a1 = 0.0, a1u = 0.0, aetce = 0.0;
    for (int q = K.ae_colptr[j] + lane8; q < K.ae_colptr[j + 1]; q += 8) {
      const int r = K.ae_rowidx[q];
      const double a = Ae[q], dr = d_ce[r];
      const double cer = (ce[r] - A.pn[r]) + A.pn[me + r];
      a1 += a * A.y[r];
      a1u += ((1.0 / dr) * a) * (dr * A.y[r]);
      aetce += a * cer;
    }
    double a2 = 0.0, a2u = 0.0, aitcp = 0.0;
    for (int q = K.ai_colptr[j] + lane8; q < K.ai_colptr[j + 1]; q += 8) {
      const int r = K.ai_rowidx[q];
      const double a = Ai[q], dr = d_ci[r];
      const double cir = (ci[r] - A.pn[2 * me + r]) + A.pn[2 * me + mi + r];
      a2 += a * A.z0[r];
      a2u += ((1.0 / dr) * a) * (dr * A.z0[r]);
      aitcp += a * fmin(cir, 0.0);
    }
    const double xj = A.in[j], dx = xj - A.xr[j], wj = A.w[j], go = A.g_outer[j];
    a1 = ipm_group8_sum(a1);
    a1u = ipm_group8_sum(a1u);
    aetce = ipm_group8_sum(aetce);
    a2 = ipm_group8_sum(a2);
    a2u = ipm_group8_sum(a2u);
    aitcp = ipm_group8_sum(aitcp);
    if (lane8 == 0) {
      const double g = wj * dx;
      dual_entry((g - a1) - a2, (g - a1u) - a2u);
      acc[AETCE] += aetce * aetce;
      acc[AITCP] += aitcp * aitcp;
      var_entry(xj);
      acc[COST] += 0.5 * (wj * (dx * dx));
      acc[DPHI_O] += go * dx;
    }
  }
  // equality rows, with the columns and bound rows of p_e, n_e
  for (int j = t0; j < me; j += stride) {
    const double pe = A.pn[j], ne = A.pn[me + j], yr = A.y[j], dr = d_ce[j];
    const double c = (ce[j] - pe) + ne;
    acc[CE] = fmax(acc[CE], fabs(c));
    acc[CE_U] = fmax(acc[CE_U], fabs((1.0 / dr) * c));
    acc[Y1] += fabs(yr);
    acc[Y1_U] += fabs(dr * yr);
    acc[CESQ] += c * c;
    acc[VIOL] += fabs(c);
    if (!isfinite(c)) acc[FINITE] = 0.0;
    acc[VIOL_O] += fabs(ce[j]);
    // columns: g' - A_e'^T y - A_i'^T z' = rho + y - z_1 (p_e), rho - y - z_2 (n_e)
    const double z1 = A.zx[j], z2 = A.zx[me + j];
    const double yu = (1.0 / dr) * (dr * yr);
    dual_entry((A.rho + yr) - z1, (A.rho + yu) - z1);
    dual_entry((A.rho - yr) - z2, (A.rho - yu) - z2);
    acc[AETCE] += 2.0 * (c * c);  // (-c)^2 + c^2
    const double cp1 = fmin(pe, 0.0), cp2 = fmin(ne, 0.0);
    acc[AITCP] += cp1 * cp1 + cp2 * cp2;
    var_entry(pe);
    var_entry(ne);
    acc[COST] += A.rho * pe + A.rho * ne;
    ineq_row(pe, A.sx[j], z1, 1.0);
    ineq_row(ne, A.sx[me + j], z2, 1.0);
    acc[EMIN0] = fmin(acc[EMIN0], fmin((1.0 / A.sx[j]) * z1, (1.0 / A.sx[me + j]) * z2));
  }
  // inequality rows of block 0, with the columns and bound rows of p_i, n_i
  for (int r = t0; r < mi; r += stride) {
    const int e3 = 2 * me + r, e4 = 2 * me + mi + r;
    const double pi = A.pn[e3], ni = A.pn[e4], sr = A.s0[r], zr = A.z0[r], dr = d_ci[r];
    const double c = (ci[r] - pi) + ni;
    ineq_row(c, sr, zr, dr);
    const double z3 = A.zx[e3], z4 = A.zx[e4];
    const double zu = (1.0 / dr) * (dr * zr);
    dual_entry((A.rho + zr) - z3, (A.rho + zu) - z3);
    dual_entry((A.rho - zr) - z4, (A.rho - zu) - z4);
    const double cp0 = fmin(c, 0.0), cp3 = fmin(pi, 0.0), cp4 = fmin(ni, 0.0);
    acc[AITCP] += (cp3 - cp0) * (cp3 - cp0) + (cp0 + cp4) * (cp0 + cp4);
    var_entry(pi);
    var_entry(ni);
    acc[COST] += A.rho * pi + A.rho * ni;
    ineq_row(pi, A.sx[e3], z3, 1.0);
    ineq_row(ni, A.sx[e4], z4, 1.0);
    {  // the pivots of the (p_i, n_i) block eliminated in that order: a = Sigma_0 + Sigma_3, then det / a (fr_expand_kernel)
      const double sig = (1.0 / sr) * zr, a3 = (1.0 / A.sx[e3]) * z3, a4 = (1.0 / A.sx[e4]) * z4;
      const double a = sig + a3;
      acc[EMIN0] = fmin(acc[EMIN0], fmin(a, (sig * (a3 + a4) + a3 * a4) / a));
    }
    // the outer problem's filter quantities at (x, s_0)
    acc[VIOL_O] += fabs(ci[r] - sr);
    acc[LOGSUM_O] += log(sr);
    const double so = A.s_outer[r];
    acc[DPHI_O] -= mu_outer * ((1.0 / so) * (sr - so));
  }
  if (check_all_V)
    for (int k = t0; k < nV; k += stride)
      if (!isfinite(V[k])) acc[FINITE] = 0.0;
  if (!sum_block) {
    block_reduce<NQ, kFrErrThreads>(acc, ops, scratch);
    // (block_reduce leaves the workgroup's totals behind its per-wave rows)
    if (threadIdx.x < NQ) coherent_store(&partial[blockIdx.x * NQ + threadIdx.x], scratch[(kFrErrThreads / 64) * NQ + threadIdx.x], true);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int old = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = old + 1 == gridDim.x;
    if (last) __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!last) return;
  // thread = (slice of the workgroups, quantity): eight slices walk the partials side by side, four loads in flight
  // each, then the slices are combined in slice order (a fixed order: the same bits run after run).  One lane per
  // quantity walking them one after the other was a dependent trip to memory per workgroup: 12 of the kernel's 19 us.
  double* tot = fold;
  double* part = fold + NQ;  // [8][NQ]
  {
    const int q = threadIdx.x & 31, sl = threadIdx.x >> 5;
    if (q < NQ) {
      int op = ops[0];
#pragma unroll
      for (int k = 1; k < NQ; ++k)
        if (q == k) op = ops[k];
      double v = op == IPM_SUM ? 0.0 : (op == IPM_MAX ? -1e300 : 1e300);
      const int nb = n_err_blocks;
      int b = sl;
      for (; b + 24 < nb; b += 32) {
        const double v0 = coherent_load(&partial[b * NQ + q], true), v1 = coherent_load(&partial[(b + 8) * NQ + q], true),
                     v2 = coherent_load(&partial[(b + 16) * NQ + q], true), v3 = coherent_load(&partial[(b + 24) * NQ + q], true);
        v = ipm_combine(op, ipm_combine(op, ipm_combine(op, ipm_combine(op, v, v0), v1), v2), v3);
      }
      for (; b < nb; b += 8) v = ipm_combine(op, v, coherent_load(&partial[b * NQ + q], true));
      part[sl * NQ + q] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x < NQ) {
    const int q = threadIdx.x;
    int op = ops[0];
#pragma unroll
    for (int k = 1; k < NQ; ++k)
      if (q == k) op = ops[k];
    double v = part[q];
    for (int k = 1; k < 8; ++k) v = ipm_combine(op, v, part[k * NQ + q]);
    tot[q] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    IpmErrOut e;
    e.dual_inf_u = tot[DUAL_U];
    e.sz_max_u = tot[SZ_MAX_U];
    e.ce_inf_u = tot[CE_U];
    e.cis_inf_u = tot[CIS_U];
    e.y1_u = tot[Y1_U];
    e.z1_u = tot[Z1_U];
    e.dual_inf = tot[DUAL];
    e.sz_min = tot[SZ_MIN];
    e.sz_max = tot[SZ_MAX];
    e.ce_inf = tot[CE];
    e.cis_inf = tot[CIS];
    e.y1 = tot[Y1];
    e.z1 = tot[Z1];
    e.f = tot[COST];
    e.viol = tot[VIOL];
    e.logsum = tot[LOGSUM];
    e.aetce_sq = tot[AETCE];
    e.ce_sq = tot[CESQ];
    e.aitcp_sq = tot[AITCP];
    e.cp_sq = tot[CPSQ];
    e.x_inf = tot[XINF];
    e.s_inf = tot[SINF];
    const double f_outer = coherent_load(&V[K.off_f], true);  // (a riding sum's, or the sweep's own)
    e.finite = (tot[FINITE] != 0.0 && isfinite(tot[COST])) ? 1.0 : 0.0;
    e.ci_all_pos = tot[CIPOS];
    out->e = e;
    out->f_outer = f_outer;
    out->viol_outer = tot[VIOL_O];
    out->logsum_outer = tot[LOGSUM_O];
    out->dphi_outer = tot[DPHI_O];
    out->eliminated_min_pivot = tot[EMIN0];
    ipm_publish(seq_dev, seq_host);
  }
}

int grid_for(int work, int block, int cap = 2048) { return std::max(1, std::min((work + block - 1) / block, cap)); }

}  // namespace

FrDevice::FrDevice(DeviceNlp& dev) : m_dev(dev) {
  const NlpStructure& s = dev.structure();
  const KktPlan& k = dev.kkt();
  if (dev.batch() != 1) throw std::runtime_error("slpx: feasibility restoration handles one problem");
  m_n = s.n;
  m_me = s.m_e;
  m_mi = s.m_i;
  m_M = 2 * m_me + 2 * m_mi;
  std::vector<int32_t> diag_of(static_cast<size_t>(k.lhs.nnz()), -1);
  for (int c = 0; c < k.dim; ++c)
    for (int p = k.lhs.colptr[c]; p < k.lhs.colptr[c + 1]; ++p)
      if (k.lhs.rowidx[p] == c) diag_of[p] = c;
  m_diag_of.upload(diag_of);
  const size_t n1 = static_cast<size_t>(std::max(1, m_n)), e1 = static_cast<size_t>(std::max(1, m_me)), i1 = static_cast<size_t>(std::max(1, m_mi)),
               M1 = static_cast<size_t>(std::max(1, m_M));
  m_xr.alloc(n1);
  m_w.alloc(n1);
  m_g_outer.alloc(n1);
  m_s_outer.alloc(i1);
  m_scales.alloc(1 + e1 + i1);
  for (DevBuf<double>* b : {&m_pn, &m_sx, &m_zx, &m_dpn, &m_psx, &m_pzx, &m_soc_x, &m_keep_dpn, &m_keep_psx, &m_keep_pzx, &m_pn_t, &m_sx_t, &m_zx_t}) {
    b->alloc(M1);
    b->zero();
  }
  m_soc_ce.alloc(e1);
  m_soc_c0.alloc(i1);
  m_keep_p.alloc(static_cast<size_t>(std::max(1, k.dim)));
  m_keep_ps0.alloc(i1);
  m_keep_pz0.alloc(i1);
  m_alpha.alloc(4);
  m_alpha.zero();
  m_partial.alloc(static_cast<size_t>(64) * fr_err::NQ);
  m_done.upload(std::vector<unsigned int>(1, 0u));
  m_expand_partial.alloc(static_cast<size_t>(kFrExpandMaxBlocks) * 4);
  m_expand_sync.upload(std::vector<unsigned int>(32, 0u));
  m_seq_dev.alloc(1);
  m_seq_dev.zero();
  unsigned long long* seq = nullptr;
  SLPX_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&seq), sizeof(unsigned long long)));
  *seq = 0;
  m_h_seq = seq;
  SLPX_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&m_host), sizeof(FrHost)));
  std::memset(m_host, 0, sizeof(FrHost));
  SLPX_HIP_CHECK(hipDeviceSynchronize());
}

FrDevice::~FrDevice() {
  if (m_h_seq) (void)hipHostFree(const_cast<unsigned long long*>(m_h_seq));
  if (m_host) (void)hipHostFree(m_host);
}

FrDevice::Args FrDevice::args(bool ahead) const {
  Args a;
  const NlpStructure& s = m_dev.structure();
  a.n = m_n;
  a.m_e = m_me;
  a.m_i = m_mi;
  a.off_Hf = s.off_Hf;
  a.off_Hc = s.off_Hc;
  a.rho = kFrRho;
  a.K = m_dev.kdev();
  a.V = m_dev.d_V();
  a.Vt = m_dev.d_V_trial();
  a.in = m_dev.d_x();
  a.trial_in = m_dev.d_trial_in();
  a.s0 = m_dev.d_s();
  a.y = m_dev.d_y();
  a.z0 = m_dev.d_z();
  a.p = m_dev.d_p();
  a.ps0 = m_dev.d_ps();
  a.pz0 = m_dev.d_pz();
  a.xr = m_xr.p;
  a.w = m_w.p;
  a.g_outer = m_g_outer.p;
  a.s_outer = m_s_outer.p;
  a.scales = m_scales.p;
  a.pn = m_pn.p;
  a.sx = m_sx.p;
  a.zx = m_zx.p;
  a.dpn = m_dpn.p;
  a.psx = m_psx.p;
  a.pzx = m_pzx.p;
  a.soc_ce = m_soc_ce.p;
  a.soc_c0 = m_soc_c0.p;
  a.soc_x = m_soc_x.p;
  a.s0_t = m_dev.d_s_ahead();
  a.y_t = m_dev.d_y_ahead();
  a.z0_t = m_dev.d_z_ahead();
  a.pn_t = m_pn_t.p;
  a.sx_t = m_sx_t.p;
  a.zx_t = m_zx_t.p;
  if (ahead) {  // the kernels' view of the look-ahead iterate as THE iterate (errors)
    a.V = m_dev.d_V_trial();
    a.in = m_dev.d_trial_in();
    a.s0 = a.s0_t;
    a.y = a.y_t;
    a.z0 = a.z0_t;
    a.pn = a.pn_t;
    a.sx = a.sx_t;
    a.zx = a.zx_t;
  }
  return a;
}

void FrDevice::begin(const double* x_r, const double* w, const double* g_outer, const double* s_outer, double mu_outer, const double* pn,
                     const double* sx, const double* zx, const std::vector<double>& err_scales) {
  hipStream_t st = m_dev.stream();
  auto up = [&](DevBuf<double>& b, const double* src, int count) {
    if (count > 0) SLPX_HIP_CHECK(hipMemcpyAsync(b.p, src, static_cast<size_t>(count) * sizeof(double), hipMemcpyHostToDevice, st));
  };
  up(m_xr, x_r, m_n);
  up(m_w, w, m_n);
  up(m_g_outer, g_outer, m_n);
  up(m_s_outer, s_outer, m_mi);
  up(m_pn, pn, m_M);
  up(m_sx, sx, m_M);
  up(m_zx, zx, m_M);
  if (static_cast<int>(err_scales.size()) != 1 + m_me + m_mi) throw std::runtime_error("FrDevice::begin: wrong length of the scaling vector");
  up(m_scales, err_scales.data(), 1 + m_me + m_mi);
  m_mu_outer = mu_outer;
  SLPX_HIP_CHECK(hipStreamSynchronize(st));  // (the host vectors may go away)
}

void FrDevice::build(double delta, double mu, bool soc, bool rhs_only, bool second) {
  const KktDev K = m_dev.kdev();
  const int work = std::max(rhs_only ? 0 : K.nnz_lhs, K.dim);
  if (second && m_lhs2.n == 0) {
    m_lhs2.alloc(static_cast<size_t>(std::max(1, K.nnz_lhs)));
    m_rhs2.alloc(static_cast<size_t>(std::max(1, K.dim)));
  }
  hipLaunchKernelGGL(fr_build_kernel, dim3(grid_for(work, 256)), dim3(256), 0, m_dev.stream(), args(), m_diag_of.p, delta, mu, soc ? 1 : 0,
                     rhs_only ? 1 : 0, second ? m_lhs2.p : m_dev.lhs_raw(), second ? m_rhs2.p : m_dev.rhs_raw(), 0.0,
                     static_cast<double*>(nullptr), static_cast<double*>(nullptr));
  SLPX_HIP_CHECK(hipGetLastError());
  if (!second) m_dev.system_written_by_caller(!rhs_only, true);
}

void FrDevice::build_pair(double delta, double delta_second, double mu) {
  const KktDev K = m_dev.kdev();
  const int work = std::max(K.nnz_lhs, K.dim);
  if (m_lhs2.n == 0) {
    m_lhs2.alloc(static_cast<size_t>(std::max(1, K.nnz_lhs)));
    m_rhs2.alloc(static_cast<size_t>(std::max(1, K.dim)));
  }
  hipLaunchKernelGGL(fr_build_kernel, dim3(grid_for(work, 256), 2), dim3(256), 0, m_dev.stream(), args(), m_diag_of.p, delta, mu, 0, 0,
                     m_dev.lhs_raw(), m_dev.rhs_raw(), delta_second, m_lhs2.p, m_rhs2.p);
  SLPX_HIP_CHECK(hipGetLastError());
  m_dev.system_written_by_caller(true, true);
}

void FrDevice::expand(double delta, double mu, double tau, bool soc, bool ahead) {
  const int work = std::max({m_n, m_me, m_mi, 1});
  const int blocks = grid_for(work, kFrExpandThreads, kFrExpandMaxBlocks);
  hipLaunchKernelGGL(fr_expand_kernel, dim3(blocks), dim3(kFrExpandThreads), 0, m_dev.stream(), args(), delta, mu, tau, soc ? 1 : 0,
                     ahead ? 1 : 0, m_alpha.p, &m_host->dir, m_expand_partial.p, m_expand_sync.p, ++m_expand_generation);
  SLPX_HIP_CHECK(hipGetLastError());
}

void FrDevice::accept_lookahead() {
  m_dev.ipm_accept_lookahead();  // x | y | z_0, s_0, y, z_0 and V change roles with their look-ahead twins
  m_pn.swap(m_pn_t);
  m_sx.swap(m_sx_t);
  m_zx.swap(m_zx_t);
}

void FrDevice::trial_point(double alpha) {
  hipLaunchKernelGGL(fr_trial_point_kernel, dim3(grid_for(m_n, 256)), dim3(256), 0, m_dev.stream(), m_n, m_dev.d_x(), m_dev.d_p(), alpha,
                     m_dev.d_trial_in());
  SLPX_HIP_CHECK(hipGetLastError());
}

void FrDevice::trial_metrics(double alpha, double /*mu*/) {
  hipLaunchKernelGGL(fr_trial_metrics_kernel, dim3(1), dim3(kIpmThreads), 0, m_dev.stream(), args(), alpha, m_alpha.p, &m_host->trial,
                     m_seq_dev.p, m_h_seq);
  ++m_seq_expected;
  SLPX_HIP_CHECK(hipGetLastError());
}

void FrDevice::commit(double alpha, double alpha_z, double mu) {
  const int work = std::max({m_n, m_me, m_mi, m_M, 1});
  hipLaunchKernelGGL(fr_commit_kernel, dim3(grid_for(work, 256)), dim3(256), 0, m_dev.stream(), args(), alpha, alpha_z, mu);
  SLPX_HIP_CHECK(hipGetLastError());
  m_dev.state_changed_by_caller();
}

void FrDevice::errors(bool check_all_V, double /*mu*/, bool ahead, bool sums_ride) {
  const int work = std::max({8 * m_n, m_me, m_mi, 1});  // (eight lanes per column of x)
  const int blocks = grid_for(work, kFrErrThreads, 64);
  const Args a = args(ahead);
  FrSumsRide ride;
  ride.n_err_blocks = blocks;
  ride.red = m_dev.reduces_dev();
  ride.tape_scales = m_dev.tape_scales_dev();
  ride.Vw = const_cast<double*>(a.V);
  const int n_sums = sums_ride ? m_dev.n_reduces() : 0;
  hipLaunchKernelGGL(fr_errors_kernel, dim3(blocks + n_sums), dim3(kFrErrThreads), 0, m_dev.stream(), a, m_dev.structure().nV,
                     check_all_V ? 1 : 0, m_mu_outer, m_partial.p, m_done.p, ahead ? &m_host->err_ahead : &m_host->err, m_seq_dev.p, m_h_seq,
                     ride);
  ++m_seq_expected;
  SLPX_HIP_CHECK(hipGetLastError());
}

void FrDevice::soc_accumulate(double alpha, bool first) {
  const int work = std::max({m_me, m_mi, 1});
  hipLaunchKernelGGL(fr_soc_accumulate_kernel, dim3(grid_for(work, 256)), dim3(256), 0, m_dev.stream(), args(), alpha, first ? 1 : 0);
  SLPX_HIP_CHECK(hipGetLastError());
}

void FrDevice::save_direction() {
  const KktDev K = m_dev.kdev();
  hipLaunchKernelGGL(fr_copy_direction_kernel, dim3(grid_for(std::max(K.dim, m_M), 256)), dim3(256), 0, m_dev.stream(), K.dim, m_mi, m_M,
                     m_dev.d_p(), m_dev.d_ps(), m_dev.d_pz(), m_dpn.p, m_psx.p, m_pzx.p, m_keep_p.p, m_keep_ps0.p, m_keep_pz0.p, m_keep_dpn.p,
                     m_keep_psx.p, m_keep_pzx.p);
  SLPX_HIP_CHECK(hipGetLastError());
}

void FrDevice::restore_direction() {
  const KktDev K = m_dev.kdev();
  hipLaunchKernelGGL(fr_copy_direction_kernel, dim3(grid_for(std::max(K.dim, m_M), 256)), dim3(256), 0, m_dev.stream(), K.dim, m_mi, m_M,
                     m_keep_p.p, m_keep_ps0.p, m_keep_pz0.p, m_keep_dpn.p, m_keep_psx.p, m_keep_pzx.p, m_dev.d_p(), m_dev.d_ps(), m_dev.d_pz(),
                     m_dpn.p, m_psx.p, m_pzx.p);
  SLPX_HIP_CHECK(hipGetLastError());
}

void FrDevice::wait_published() {
  spin_on_published([&] { return *m_h_seq >= m_seq_expected; }, m_dev.raw_stream(), "slpx: a restoration chain finished without publishing");
}

void FrDevice::download_state(double* pn, double* sx, double* zx) {
  hipStream_t st = m_dev.stream();
  if (m_M > 0) {
    SLPX_HIP_CHECK(hipMemcpyAsync(pn, m_pn.p, static_cast<size_t>(m_M) * sizeof(double), hipMemcpyDeviceToHost, st));
    SLPX_HIP_CHECK(hipMemcpyAsync(sx, m_sx.p, static_cast<size_t>(m_M) * sizeof(double), hipMemcpyDeviceToHost, st));
    SLPX_HIP_CHECK(hipMemcpyAsync(zx, m_zx.p, static_cast<size_t>(m_M) * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  SLPX_HIP_CHECK(hipStreamSynchronize(st));
}

void FrDevice::download_direction(double* dpn, double* psx, double* pzx) {
  hipStream_t st = m_dev.stream();
  if (m_M > 0) {
    SLPX_HIP_CHECK(hipMemcpyAsync(dpn, m_dpn.p, static_cast<size_t>(m_M) * sizeof(double), hipMemcpyDeviceToHost, st));
    SLPX_HIP_CHECK(hipMemcpyAsync(psx, m_psx.p, static_cast<size_t>(m_M) * sizeof(double), hipMemcpyDeviceToHost, st));
    SLPX_HIP_CHECK(hipMemcpyAsync(pzx, m_pzx.p, static_cast<size_t>(m_M) * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  SLPX_HIP_CHECK(hipStreamSynchronize(st));
}

}  // namespace slpx
