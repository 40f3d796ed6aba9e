// Device side of the interior-point ITERATION around the Newton step:
// fraction-to-the-boundary rule, trial iterate, trial merit quantities, second-order
// correction bookkeeping, iterate update, and the error / infeasibility / divergence
// reductions — so that a whole `Problem::solve()` keeps x, s, y, z, the step and every
// O(n) vector in HBM and only a few dozen scalars per iteration cross PCIe (written
// straight into pinned host memory by the kernels).
//
// Reference formulas: interior_point.hpp:488-509 (step sizes, directional derivative),
// :512-563 (trial iterate), :566-668 (second-order corrections), :775-801 (iterate update,
// z reset), util/fraction_to_the_boundary_rule.hpp:19-43, util/filter.hpp:30-60 (entry =
// f − μ Σ ln s, ‖c_e‖₁ + ‖c_i − s‖₁), util/kkt_error.hpp:92-146 and :216-251 (un-scaling),
// util/is_locally_infeasible.hpp:17-60, interior_point.hpp:399-407 (divergence).
//
// The reductions are latency-, not bandwidth-bound (n is a few thousand to a few ten
// thousand): wave totals by DPP butterflies + scalar lane reads (no LDS traffic), a fixed
// combination order (reproducible run to run); the big one (23 quantities) is spread over
// workgroups of 256 lanes with a tiny second launch folding the per-workgroup partials.
#pragma once

#include <hip/hip_runtime.h>

#include "coherent.h"
#include "device.hpp"
#include "ipm_decide.h"
#include "ipm_reduce.h"

namespace slpx {

// Step sizes and directional derivative for the direction (p, ps, pz), then the first trial
// point x + alpha_max p_x.  `out` (pinned host) and `alpha_dev` (device copy, read by
// ipm_trial_metrics_kernel of the same speculative chain) may be the only consumers.
__global__ __launch_bounds__(kIpmThreads) void ipm_direction_kernel(
    KktDev K, const double* __restrict__ V, const double* __restrict__ x, const double* __restrict__ s,
    const double* __restrict__ z, const double* __restrict__ p, const double* __restrict__ ps,
    const double* __restrict__ pz, const double* __restrict__ mu_dev, double tau, double* __restrict__ trial_x,
    double* __restrict__ alpha_dev, IpmDirOut* __restrict__ out) {
  __shared__ double scratch[17 * 3];
  const int tid = threadIdx.x;
  const double mu = mu_dev[0];
  double acc[3] = {1.0, 1.0, 0.0};  // alpha_max, alpha_z, D_phi
  for (int r = tid; r < K.m_i; r += kIpmThreads) {
    const double sr = s[r], psr = ps[r], zr = z[r], pzr = pz[r];
    if (psr < 0.0) acc[0] = fmin(acc[0], -tau / psr * sr);
    if (pzr < 0.0) acc[1] = fmin(acc[1], -tau / pzr * zr);
    acc[2] -= mu * ((1.0 / sr) * psr);
  }
  for (int j = tid; j < K.n; j += kIpmThreads) {
    const int gs = K.g_src[j];
    if (gs >= 0) acc[2] += V[gs] * p[j];
  }
  const int ops[3] = {IPM_MIN, IPM_MIN, IPM_SUM};
  block_reduce<3, kIpmThreads>(acc, ops, scratch);
  const double alpha = acc[0];
  for (int j = tid; j < K.n; j += kIpmThreads) trial_x[j] = x[j] + alpha * p[j];
  if (tid == 0) {
    alpha_dev[0] = acc[0];
    alpha_dev[1] = acc[1];
    out->alpha_max = acc[0];
    out->alpha_z = acc[1];
    out->D_phi = acc[2];
  }
}

// ipm_direction_kernel + the whole look-ahead iterate: what ipm_commit_kernel would make of the step
// (alpha_max, alpha_z) — x + alpha p_x, s + alpha p_s, y + alpha_z p_y, z + alpha_z p_z with the z reset —
// written to a SECOND set of buffers (`in_t` = [x | y | z] as the tape reads it, s_t, y_t, z_t) instead of
// over the current one.  The full tape and the error reductions run on those next (DeviceNlp::
// ipm_lookahead); if the filter takes the point the buffers swap roles and nothing is recomputed.
// (struct IpmTwin, struct IpmLookaheadArgs: device.hpp)
template <int THREADS>
__device__ __forceinline__ void ipm_lookahead_body(IpmLookaheadArgs A, double* scratch) {
  const int tid = threadIdx.x;
  auto ld = [](const double* q) { return *q; };
  auto ld_stats = [](const LdltStats* q) { return q[0]; };
  const int n = A.n, m_e = A.m_e, m_i = A.m_i;
  const double *p = A.p, *ps = A.ps, *pz = A.pz;
  // The factorization this direction comes from has the wrong inertia (or failed): the policy loop will
  // redo the attempt and look at nothing of this chain — alpha_dev[2] tells its error launch to pass.
  // (one lane-uniform 16-byte load, issued with the others below)
  const LdltStats st = ld_stats(A.stats);
  bool wrong = st.n_bad != 0 || st.n_pos != n || st.n_neg != m_e || st.n_zero != 0;
  if (A.tw.mode != 0) {
    // A twin attempt (ldlt_mf_twin_kernel): the policy's choice between the two, from the same counters the host
    // reads (NewtonSystem::compute_impl — keep the two in step).  The second attempt stands for the policy's next
    // one only if the first failed the way that leads to it: beside the unregularized attempt (mode 2) any failure
    // does (:82-102, also a pivot below 1e-4); in the loop too many negative pivots (mode 1: delta x 10, :127-130) or too
    // many positive ones (mode 3: gamma x 10, :131-135), nothing else.
    const LdltStats st2 = ld_stats(A.tw.stats);
    if (A.tw.mode == 2 && !wrong && __longlong_as_double(static_cast<long long>(st.min_abs_bits)) < 1e-4) wrong = true;
    const bool inertia_only = st.n_bad == 0 && st.n_zero == 0;
    const bool leads_to_second = A.tw.mode == 2 || (A.tw.mode == 1 && inertia_only && st.n_neg > m_e) ||
                                 (A.tw.mode == 3 && inertia_only && st.n_neg <= m_e && st.n_pos > n);
    const bool second_good = st2.n_bad == 0 && st2.n_pos == n && st2.n_neg == m_e && st2.n_zero == 0;
    if (wrong && leads_to_second && second_good) {
      wrong = false;
      p = A.tw.p;
      ps = A.tw.ps;
      pz = A.tw.pz;
    }
  }
  if (tid == 0) A.alpha_dev[2] = wrong ? 1.0 : 0.0;
  if (wrong) return;
  const double mu = A.mu[0], tau = A.tau;
  const double *V = A.V, *in = A.in, *s = A.s, *y = A.y, *z = A.z;
  double *in_t = A.in_t, *s_t = A.s_t, *y_t = A.y_t, *z_t = A.z_t;
  double acc[3] = {1.0, 1.0, 0.0};  // alpha_max, alpha_z, D_phi
  // Systems of up to kPre x 1024 rows (every BASELINE horizon): EVERYTHING the kernel reads is requested before the
  // reduction — the iterate and the direction for the second pass too — so that the launch is two trips to memory
  // (g's sources, then the values) instead of four; a trip after a kernel boundary is ~1 us.
  constexpr int kPre = 5;
  if (THREADS == kIpmThreads && n <= kPre * THREADS && m_e <= kPre * THREADS && m_i <= kPre * THREADS) {
    double xs[kPre], px[kPre], ys[kPre], py[kPre], ss[kPre], pss[kPre], zs[kPre], pzs[kPre], gv[kPre];
    int gs[kPre];
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
      const int j = tid + k * THREADS;
      gs[k] = j < n ? A.g_src[j] : -1;
      xs[k] = j < n ? in[j] : 0.0;
      px[k] = j < n ? ld(p + j) : 0.0;
      ys[k] = j < m_e ? y[j] : 0.0;
      py[k] = j < m_e ? ld(p + n + j) : 0.0;
      ss[k] = j < m_i ? s[j] : 1.0;
      pss[k] = j < m_i ? ld(ps + j) : 0.0;
      zs[k] = j < m_i ? z[j] : 1.0;
      pzs[k] = j < m_i ? ld(pz + j) : 0.0;
    }
#pragma unroll
    for (int k = 0; k < kPre; ++k) gv[k] = gs[k] >= 0 ? V[gs[k]] : 0.0;
    // (the same order of accumulation as the loops below: rows r = tid, tid + 1024, ...; then the columns)
#pragma unroll
    for (int k = 0; k < kPre; ++k)
      if (tid + k * THREADS < m_i) {
        if (pss[k] < 0.0) acc[0] = fmin(acc[0], -tau / pss[k] * ss[k]);
        if (pzs[k] < 0.0) acc[1] = fmin(acc[1], -tau / pzs[k] * zs[k]);
        acc[2] -= mu * ((1.0 / ss[k]) * pss[k]);
      }
#pragma unroll
    for (int k = 0; k < kPre; ++k)
      if (gs[k] >= 0) acc[2] += gv[k] * px[k];
    const int ops[3] = {IPM_MIN, IPM_MIN, IPM_SUM};
    block_reduce<3, THREADS>(acc, ops, scratch);
    const double alpha = acc[0], alpha_z = acc[1];
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
      const int j = tid + k * THREADS;
      if (j < n) in_t[j] = xs[k] + alpha * px[k];
      if (j < m_e) {
        const double v = ys[k] + alpha_z * (-py[k]);
        y_t[j] = v;
        in_t[n + j] = v;
      }
      if (j < m_i) {
        const double sn = ss[k] + alpha * pss[k];
        double zn = zs[k] + alpha_z * pzs[k];
        constexpr double kappa = 1e10;
        const double lo = 1.0 / kappa * mu / sn, hi = kappa * mu / sn;
        zn = zn < lo ? lo : (zn > hi ? hi : zn);
        s_t[j] = sn;
        z_t[j] = zn;
        in_t[n + m_e + j] = zn;
      }
    }
  } else {
    for (int r = tid; r < m_i; r += THREADS) {
      const double sr = s[r], psr = ld(ps + r), zr = z[r], pzr = ld(pz + r);
      if (psr < 0.0) acc[0] = fmin(acc[0], -tau / psr * sr);
      if (pzr < 0.0) acc[1] = fmin(acc[1], -tau / pzr * zr);
      acc[2] -= mu * ((1.0 / sr) * psr);
    }
    for (int j = tid; j < n; j += THREADS) {
      const int gs = A.g_src[j];
      if (gs >= 0) acc[2] += V[gs] * ld(p + j);
    }
    const int ops[3] = {IPM_MIN, IPM_MIN, IPM_SUM};
    block_reduce<3, THREADS>(acc, ops, scratch);
    const double alpha = acc[0], alpha_z = acc[1];
    for (int j = tid; j < n; j += THREADS) in_t[j] = in[j] + alpha * ld(p + j);
    for (int r = tid; r < m_e; r += THREADS) {
      const double v = y[r] + alpha_z * (-ld(p + n + r));
      y_t[r] = v;
      in_t[n + r] = v;
    }
    for (int r = tid; r < m_i; r += THREADS) {
      const double sn = s[r] + alpha * ld(ps + r);
      double zn = z[r] + alpha_z * ld(pz + r);
      constexpr double kappa = 1e10;
      const double lo = 1.0 / kappa * mu / sn, hi = kappa * mu / sn;
      zn = zn < lo ? lo : (zn > hi ? hi : zn);
      s_t[r] = sn;
      z_t[r] = zn;
      in_t[n + m_e + r] = zn;
    }
  }
  if (tid == 0) {
    A.alpha_dev[0] = acc[0];
    A.alpha_dev[1] = acc[1];
    A.alpha_dev[3] = acc[2];  // (D_phi: the decision made on the device reads it, ipm_error_fold)
    A.out->alpha_max = acc[0];
    A.out->alpha_z = acc[1];
    A.out->D_phi = acc[2];
  }
}

__global__ __launch_bounds__(kIpmThreads) void ipm_lookahead_kernel(IpmLookaheadArgs A) {
  __shared__ double scratch[17 * 3];
  ipm_lookahead_body<kIpmThreads>(A, scratch);
}

// trial_x = x + alpha p_x (backtracking)
__global__ __launch_bounds__(256) void ipm_trial_point_kernel(int n, const double* __restrict__ x,
                                                              const double* __restrict__ p, double alpha,
                                                              double* __restrict__ trial_x) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
    trial_x[j] = x[j] + alpha * p[j];
}

// Filter quantities of the trial point whose f, c_e, c_i a forward sweep left in Vt.
// alpha < 0: take the step size from alpha_dev[0].  s_from_ci: trial_s = trial c_i
// (feasible-IPM option, interior_point.hpp:520-526).
// The separable sums the forward sweep left as partial terms (the cost: one term per stage
// group) are finished here, in the order tape_reduce_body uses (same bits), instead of by a
// launch of their own between the sweep and this kernel.
__global__ __launch_bounds__(kIpmThreads) void ipm_trial_metrics_kernel(
    KktDev K, double* __restrict__ Vt, const double* __restrict__ s, const double* __restrict__ ps,
    double alpha, const double* __restrict__ alpha_dev, int s_from_ci, IpmTrialOut* __restrict__ out,
    unsigned long long* __restrict__ seq_dev, volatile unsigned long long* seq_host,
    const NlpStructure::SumReduce* __restrict__ red, int n_red, const double* __restrict__ scales) {
  __shared__ double scratch[17 * 3];
  __shared__ double part[64];
  const int tid = threadIdx.x;
  for (int q = 0; q < n_red; ++q) {
    const NlpStructure::SumReduce r = red[q];
    if (tid < 64) {
      double acc = 0.0;
      for (int k = tid; k < r.count; k += 64) acc += Vt[r.src_off + k];
      part[tid] = acc;
    }
    __syncthreads();
    for (int w = 32; w > 0; w >>= 1) {
      if (tid < w) part[tid] += part[tid + w];
      __syncthreads();
    }
    if (tid == 0) Vt[r.dst] = (r.scale_idx >= 0 ? scales[r.scale_idx] : 1.0) * part[0];
    __syncthreads();
  }
  if (alpha < 0.0) alpha = alpha_dev[0];
  double acc[3] = {0.0, 0.0, 1.0};  // violation, log sum, finite
  for (int r = tid; r < K.m_e; r += kIpmThreads) {
    const double c = Vt[K.off_ce + r];
    acc[0] += fabs(c);
    if (!isfinite(c)) acc[2] = 0.0;
  }
  for (int r = tid; r < K.m_i; r += kIpmThreads) {
    const double c = Vt[K.off_ci + r];
    const double st = s_from_ci ? c : s[r] + alpha * ps[r];
    acc[0] += fabs(c - st);
    acc[1] += log(st);
    if (!isfinite(c)) acc[2] = 0.0;
  }
  const int ops[3] = {IPM_SUM, IPM_SUM, IPM_MIN};
  block_reduce<3, kIpmThreads>(acc, ops, scratch);
  if (tid == 0) {
    const double f = Vt[K.off_f];
    out->f = f;
    out->viol = acc[0];
    out->logsum = acc[1];
    out->finite = (acc[2] != 0.0 && isfinite(f)) ? 1.0 : 0.0;
    ipm_publish(seq_dev, seq_host);
  }
}

// Accepts the step: x += alpha p_x, s += alpha p_s (or s = trial c_i), y += alpha_z p_y with
// p_y = −p[n:], z += alpha_z p_z, then the z reset of interior_point.hpp:797-801.  x, y, z
// also live in the tape's input vector `in` = [x | y | z].
__global__ __launch_bounds__(256) void ipm_commit_kernel(KktDev K, const double* __restrict__ Vt,
                                                         const double* __restrict__ p,
                                                         const double* __restrict__ ps,
                                                         const double* __restrict__ pz, double alpha,
                                                         double alpha_z, const double* __restrict__ mu_dev,
                                                         int s_from_ci, double* __restrict__ in,
                                                         double* __restrict__ s, double* __restrict__ y,
                                                         double* __restrict__ z) {
  const double mu = mu_dev[0];
  const int stride = gridDim.x * blockDim.x;
  const int t0 = blockIdx.x * blockDim.x + threadIdx.x;
  for (int j = t0; j < K.n; j += stride) in[j] = in[j] + alpha * p[j];
  for (int r = t0; r < K.m_e; r += stride) {
    const double v = y[r] + alpha_z * (-p[K.n + r]);
    y[r] = v;
    in[K.n + r] = v;
  }
  for (int r = t0; r < K.m_i; r += stride) {
    const double sn = s_from_ci ? Vt[K.off_ci + r] : s[r] + alpha * ps[r];
    double zn = z[r] + alpha_z * pz[r];
    constexpr double kappa = 1e10;
    const double lo = 1.0 / kappa * mu / sn, hi = kappa * mu / sn;
    zn = zn < lo ? lo : (zn > hi ? hi : zn);
    s[r] = sn;
    z[r] = zn;
    in[K.n + K.m_e + r] = zn;
  }
}

// (p, p_s, p_z) -> their keep buffers or back, in ONE launch (three device-to-device copies through the
// runtime cost ~5 us each on the stream)
__global__ __launch_bounds__(256) void ipm_copy_direction_kernel(int dim, int m_i, const double* __restrict__ p,
                                                                 const double* __restrict__ ps, const double* __restrict__ pz,
                                                                 double* __restrict__ p_to, double* __restrict__ ps_to,
                                                                 double* __restrict__ pz_to) {
  const int stride = gridDim.x * blockDim.x;
  const int t0 = blockIdx.x * blockDim.x + threadIdx.x;
  for (int j = t0; j < dim; j += stride) p_to[j] = p[j];
  for (int r = t0; r < m_i; r += stride) {
    ps_to[r] = ps[r];
    pz_to[r] = pz[r];
  }
}

// Second-order correction accumulators (interior_point.hpp:590-600):
//   c_e_soc = alpha c_e_soc + trial c_e,  (c_i − s)_soc = alpha (c_i − s)_soc + trial c_i − trial s
// first != 0: start from the current c_e, c_i − s (in V).
__global__ __launch_bounds__(256) void ipm_soc_accumulate_kernel(KktDev K, const double* __restrict__ V,
                                                                 const double* __restrict__ Vt,
                                                                 const double* __restrict__ s,
                                                                 const double* __restrict__ ps, double alpha,
                                                                 int first, int s_from_ci,
                                                                 double* __restrict__ soc_ce,
                                                                 double* __restrict__ soc_cims) {
  const int stride = gridDim.x * blockDim.x;
  const int t0 = blockIdx.x * blockDim.x + threadIdx.x;
  for (int r = t0; r < K.m_e; r += stride) {
    const double prev = first ? V[K.off_ce + r] : soc_ce[r];
    soc_ce[r] = alpha * prev + Vt[K.off_ce + r];
  }
  for (int r = t0; r < K.m_i; r += stride) {
    const double prev = first ? V[K.off_ci + r] - s[r] : soc_cims[r];
    const double ct = Vt[K.off_ci + r];
    const double st = s_from_ci ? ct : s[r] + alpha * ps[r];
    soc_cims[r] = alpha * prev + ct - st;
  }
}

// Right-hand side of a second-order correction (interior_point.hpp:611-616):
//   [−g + A_eᵀ y + A_iᵀ (μ/s − Σ (c_i − s)_soc);  −c_e_soc]
__global__ __launch_bounds__(256) void ipm_soc_rhs_kernel(KktDev K, const double* __restrict__ V,
                                                          const double* __restrict__ s,
                                                          const double* __restrict__ y,
                                                          const double* __restrict__ z,
                                                          const double* __restrict__ mu_dev,
                                                          const double* __restrict__ soc_ce,
                                                          const double* __restrict__ soc_cims,
                                                          double* __restrict__ rhs) {
  const double m = mu_dev[0];
  const double* Ae = V + K.off_Ae;
  const double* Ai = V + K.off_Ai;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < K.dim; j += gridDim.x * blockDim.x) {
    if (j >= K.n) {
      rhs[j] = -soc_ce[j - K.n];
      continue;
    }
    const int gs = K.g_src[j];
    double acc = -(gs >= 0 ? V[gs] : 0.0);
    double aey = 0.0;
    for (int q = K.ae_colptr[j]; q < K.ae_colptr[j + 1]; ++q) aey += Ae[q] * y[K.ae_rowidx[q]];
    acc += aey;
    double ait = 0.0;
    for (int q = K.ai_colptr[j]; q < K.ai_colptr[j + 1]; ++q) {
      const int r = K.ai_rowidx[q];
      const double sinv = 1.0 / s[r];
      ait += Ai[q] * (m * sinv - (sinv * z[r]) * soc_cims[r]);
    }
    rhs[j] = acc + ait;
  }
}

// p_s, p_z of a second-order correction (interior_point.hpp:625-630): the back-substitution
// with (c_i − s)_soc in place of c_i − s.
__global__ __launch_bounds__(256) void ipm_soc_backsub_kernel(KktDev K, const double* __restrict__ V,
                                                              const double* __restrict__ p,
                                                              const double* __restrict__ s,
                                                              const double* __restrict__ z,
                                                              const double* __restrict__ mu_dev,
                                                              const double* __restrict__ soc_cims,
                                                              double* __restrict__ ps, double* __restrict__ pz) {
  const double m = mu_dev[0];
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < K.m_i; r += gridDim.x * blockDim.x) {
    double aipx = 0.0;
    for (int q = K.ai_rowptr[r]; q < K.ai_rowptr[r + 1]; ++q) aipx += V[K.ai_src[q]] * p[K.ai_col[q]];
    const double sinv = 1.0 / s[r];
    const double p_s = soc_cims[r] + aipx;
    ps[r] = p_s;
    pz[r] = m * sinv - z[r] - (sinv * z[r]) * p_s;
  }
}

// Everything the host needs to decide what happens next, from the freshly swept V at the
// current iterate (see IpmErrOut), in two launches: workgroups of 256 lanes reduce their
// slice of the columns / rows into `partial[block][kIpmErrQ]`, one small workgroup folds
// the partials in block order.  scales = [d_f | d_ce | d_ci].
constexpr int kIpmErrQ = 23;
constexpr int kIpmErrThreads = 256;
namespace ipm_err {
enum {
  DUAL_U, SZ_MAX_U, CE_U, CIS_U, Y1_U, Z1_U, DUAL, SZ_MIN, SZ_MAX, CE, CIS, Y1, Z1, VIOL, LOGSUM,
  AETCE, CESQ, AITCP, CPSQ, XINF, SINF, FINITE, CIPOS
};
#define SLPX_IPM_ERR_OPS                                                                              \
  {IPM_MAX, IPM_MAX, IPM_MAX, IPM_MAX, IPM_SUM, IPM_SUM, IPM_MAX, IPM_MIN, IPM_MAX, IPM_MAX, IPM_MAX, \
   IPM_SUM, IPM_SUM, IPM_SUM, IPM_SUM, IPM_SUM, IPM_SUM, IPM_SUM, IPM_SUM, IPM_MAX, IPM_MAX, IPM_MIN, \
   IPM_MIN}
}  // namespace ipm_err

static_assert(sizeof(IpmErrOut) == 24 * sizeof(double), "ipm_error_fold writes IpmErrOut as 24 doubles");

struct IpmErrFinish {
  int n_err_blocks = 0;   // 0: two launches (partials, then ipm_error_final_kernel)
  int n_total_blocks = 0; // error workgroups + one per separable sum
  const NlpStructure::SumReduce* red = nullptr;
  const double* tape_scales = nullptr;
  double* Vw = nullptr;
  unsigned int* done = nullptr;  // workgroups through (the last one clears it)
  IpmErrOut* out = nullptr;
  unsigned long long* seq_dev = nullptr;
  volatile unsigned long long* seq_host = nullptr;
  // (look-ahead chain) non-zero: the factorization before this chain has the wrong inertia — the attempt will
  // be redone and nothing of this launch is looked at: workgroup 0 publishes, everybody leaves
  const double* skip = nullptr;
  // the common iteration decided HERE (ipm_decide.h), when the host asked for it: `ctl` the iteration's state on the
  // device,
  // `dir` = [alpha_max, alpha_z, chain void, D_phi] of the look-ahead launch, `gate` the word the NEXT step's kernel
  // — already enqueued behind this launch — reads: 1.0 run, 0.0 pass
  IpmCtl* ctl = nullptr;
  const double* dir = nullptr;
  double* gate = nullptr;
  double* go_host = nullptr;  // the verdict for the host, beside the scalars of this launch (IpmHost::go)
  // This launch RIDES in the step it decides about (ldlt_mf_kernels.h: MfRide): the verdict goes to that launch's
  // tasks as +-ride_ticket in *ride_verdict, and to the host as +-ride_ticket in *go_host — which is then also the
  // host's word that this launch's scalars are in (no sequence number: the step's own publication owns that).
  double* ride_verdict = nullptr;
  double ride_ticket = 0.0;
  unsigned long long* check_host = nullptr;  // IpmHost::check beside go_host
};

// folds the per-workgroup partials in workgroup order and hands the result to the host;
// `in_launch`: the partials and f were written by other workgroups of this launch
__device__ __forceinline__ void ipm_error_fold(const KktDev& K, const double* __restrict__ V,
                                               const double* __restrict__ partial, int n_blocks, bool in_launch,
                                               IpmErrOut* __restrict__ out, unsigned long long* __restrict__ seq_dev,
                                               volatile unsigned long long* seq_host, double* tot,
                                               const IpmErrFinish* decide = nullptr) {
  using namespace ipm_err;
  constexpr int NQ = kIpmErrQ;
  const int ops[NQ] = SLPX_IPM_ERR_OPS;
  // thread = (slice of the workgroups, quantity): eight slices walk the partials side by side, four loads in
  // flight each, then the slices are combined in slice order (fixed order: the same bits run after run)
  const int q = threadIdx.x & 31, sl = threadIdx.x >> 5;
  double* part = tot + NQ;  // [8][NQ]
  // (a deciding fold: what the decisions read of the iteration's state on the device — its scalars, the first two
  // filter entries of every lane — is asked for NOW, by the wave that will decide, and arrives while the partials are
  // folded: one trip to memory instead of three dependent ones behind the fold)
  const bool deciding = decide != nullptr && decide->ctl != nullptr && threadIdx.x < 64;
  IpmCtl head;  // (the scalars only: the table stays where it is)
  double cur_f = 0.0, cur_logsum = 0.0, cur_viol = 0.0, f_min_cv = 0.0, f_max_cv = 0.0, alpha_max = 0.0, D_phi = 0.0;
  double ent_pre[4] = {0.0, 0.0, 0.0, 0.0};
  int n_ent = 0, last_rej = 0;
  if (deciding) {
    const IpmCtl* C = decide->ctl;
    head.mu = C->mu;
    head.mu_min = C->mu_min;
    head.tolerance = C->tolerance;
    head.m_e = C->m_e;
    head.m_i = C->m_i;
    head.identity_scaling = C->identity_scaling;
    cur_f = C->cur_f;
    cur_logsum = C->cur_logsum;
    cur_viol = C->cur_viol;
    f_min_cv = C->filter.min_constraint_violation;
    f_max_cv = C->filter.max_constraint_violation;
    n_ent = C->filter.n;
    last_rej = C->filter.last_rejection_due_to_filter;
    alpha_max = decide->dir[0];
    D_phi = decide->dir[3];
    const int k = threadIdx.x;
    ent_pre[0] = C->filter.ent[2 * k];  // (within the table whatever n is: kFilterCapacity >= 128)
    ent_pre[1] = C->filter.ent[2 * k + 1];
    ent_pre[2] = C->filter.ent[2 * (k + 64)];
    ent_pre[3] = C->filter.ent[2 * (k + 64) + 1];
  }
  if (q < NQ && sl < 8) {
    int op = ops[0];
#pragma unroll
    for (int k = 1; k < NQ; ++k)
      if (q == k) op = ops[k];
    double v = op == IPM_SUM ? 0.0 : (op == IPM_MAX ? -1e300 : 1e300);
    int b = sl;
    for (; b + 24 < n_blocks; b += 32) {
      const double v0 = coherent_load(&partial[b * NQ + q], in_launch), v1 = coherent_load(&partial[(b + 8) * NQ + q], in_launch),
                   v2 = coherent_load(&partial[(b + 16) * NQ + q], in_launch), v3 = coherent_load(&partial[(b + 24) * NQ + q], in_launch);
      v = ipm_combine(op, ipm_combine(op, ipm_combine(op, ipm_combine(op, v, v0), v1), v2), v3);
    }
    for (; b < n_blocks; b += 8) v = ipm_combine(op, v, coherent_load(&partial[b * NQ + q], in_launch));
    part[sl * NQ + q] = v;
  }
  __syncthreads();
  if (threadIdx.x < NQ) {
    const int qq = threadIdx.x;
    int op = ops[0];
#pragma unroll
    for (int k = 1; k < NQ; ++k)
      if (qq == k) op = ops[k];
    double v = part[qq];
    for (int k = 1; k < 8; ++k) v = ipm_combine(op, v, part[k * NQ + qq]);
    tot[qq] = v;
  }
  __syncthreads();
  // the 24 doubles of IpmErrOut by 24 lanes of the first wave (one burst towards the host instead of 24 stores of
  // one lane), then the wave's system-scope fence and the sequence number
  if (threadIdx.x < 64) {
    const int k = threadIdx.x;
    double v = 0.0;
    if (k < 24) {
      const double f = coherent_load(&V[K.off_f], in_launch);
      v = k < 13 ? tot[k] : (k == 13 ? f : tot[k - 1]);  // (IpmErrOut: f sits between z1 and viol)
      if (k == SZ_MIN) v = K.m_i ? v : 0.0;
      if (k == FINITE + 1) v = (v != 0.0 && isfinite(f)) ? 1.0 : 0.0;
      reinterpret_cast<double*>(out)[k] = v;
    }
    if (decide != nullptr && decide->ctl != nullptr) {
#pragma clang fp contract(off)  // (the filter entries as the host forms them: no fused multiply-adds)
      // the common iteration's decisions (ipm_decide.h), by the wave that publishes: the 24 scalars through LDS (`part`
      // is free), the rules by every lane alike, the filter's table an entry per lane
      double* fin = tot + NQ;
      if (k < 24) fin[k] = v;
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      IpmErrOut e;
      for (int q = 0; q < 24; ++q) reinterpret_cast<double*>(&e)[q] = fin[q];
      IpmCtl* C = decide->ctl;
      bool go = ipm_next_iteration_is_plain(head, e, alpha_max);
      // (the wave is uniform in `go`: every lane holds the same scalars)
      if (go) {
        const FilterEntry current{cur_f - head.mu * cur_logsum, cur_viol};
        const FilterEntry trial{e.f - head.mu * e.logsum, e.viol};
        // the rules' three powers by three lanes at once (ipm_decide.h: FilterPowers)
        const double base = k == 0 ? -D_phi : (k == 1 ? current.constraint_violation : alpha_max);
        const double expo = k == 0 ? 2.3 : (k == 1 ? 1.1 : 1.5);
        const double pw_lane = k < 3 ? pow(base, expo) : 0.0;
        const FilterPowers pw{__shfl(pw_lane, 0), __shfl(pw_lane, 1), __shfl(pw_lane, 2)};
        FilterEntry add;
        bool insert = false;
        double* ent = C->filter.ent;
        go = filter_rules(f_min_cv, f_max_cv, &last_rej, current, trial, D_phi, alpha_max, pw, &add, &insert) != 0;
        if (go) {
          bool dominated = false, removes = false;
          if (k < n_ent) {
            const FilterEntry en{ent_pre[0], ent_pre[1]};
            dominated = filter_dominated_by(trial, en);
            removes = filter_dominated_by(en, add);
          }
          if (k + 64 < n_ent) {
            const FilterEntry en{ent_pre[2], ent_pre[3]};
            dominated = dominated || filter_dominated_by(trial, en);
            removes = removes || filter_dominated_by(en, add);
          }
          for (int q = k + 128; q < n_ent; q += 64) {
            const FilterEntry en{ent[2 * q], ent[2 * q + 1]};
            dominated = dominated || filter_dominated_by(trial, en);
            removes = removes || filter_dominated_by(en, add);
          }
          if (__ballot(dominated) != 0ull) go = false;  // (the host takes it from here, its own try_add sets the flag)
          else if (insert) {
            const bool any_removed = __ballot(removes) != 0ull;
            if (!any_removed && n_ent >= kFilterCapacity) go = false;
            else if (k == 0) {
              int w = n_ent;
              if (any_removed) {  // (rare: the new entry dominates older ones — in order, by one lane)
                w = 0;
                for (int q = 0; q < n_ent; ++q)
                  if (!filter_dominated_by(FilterEntry{ent[2 * q], ent[2 * q + 1]}, add)) {
                    ent[2 * w] = ent[2 * q];
                    ent[2 * w + 1] = ent[2 * q + 1];
                    ++w;
                  }
              }
              ent[2 * w] = add.cost;
              ent[2 * w + 1] = add.constraint_violation;
              C->filter.n = w + 1;
            }
          }
        }
      }
      if (k == 0) {
        if (go) {
          C->cur_f = e.f;
          C->cur_logsum = e.logsum;
          C->cur_viol = e.viol;
        }
        C->go = go ? 1 : 0;
        *decide->gate = go ? 1.0 : 0.0;
        if (decide->ride_verdict != nullptr) {
          __hip_atomic_store(decide->ride_verdict, go ? decide->ride_ticket : -decide->ride_ticket, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        } else if (decide->go_host != nullptr) {
          *decide->go_host = go ? 1.0 : 0.0;
        }
      }
      if (decide->ride_verdict != nullptr) {
        // the hand-over to the host (IpmHost::check): the exclusive-or of the 24 words this wave stored and of the verdict
        unsigned long long bits = k < 24 ? static_cast<unsigned long long>(__double_as_longlong(v)) : 0ull;
        for (int off = 1; off < 64; off <<= 1) bits ^= __shfl_xor(bits, off);
        if (k == 0) {
          __threadfence_system();  // (the 24 scalars above are the wave's: one fence)
          // (a trip to device memory between the scalars and the word that announces them, as ipm_publish has)
          const double settled = __hip_atomic_load(decide->ride_verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          *decide->check_host = bits ^ static_cast<unsigned long long>(__double_as_longlong(settled));
          *decide->go_host = settled;
        }
        return;
      }
    }
    if (k == 0) ipm_publish(seq_dev, seq_host);
  }
}

// end of a workgroup of the one-launch error computation: count it in; the last one folds
// (`tot`: 9 kIpmErrQ + 1 doubles of LDS; every thread of the workgroup calls)
__device__ __forceinline__ void ipm_error_finish(const KktDev& K, const double* __restrict__ V,
                                                 const double* __restrict__ partial, const IpmErrFinish& fin, double* tot) {
  int* last = reinterpret_cast<int*>(tot + 9 * kIpmErrQ);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this workgroup's coherent stores are in
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int old = __hip_atomic_fetch_add(fin.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *last = old + 1 == static_cast<unsigned int>(fin.n_total_blocks);
    if (*last) __hip_atomic_store(fin.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!*last) return;
  ipm_error_fold(K, V, partial, fin.n_err_blocks, true, fin.out, fin.seq_dev, fin.seq_host, tot, &fin);
}

// this lane's share of the 23 quantities: rows t0, t0 + stride, ..., columns t0 / 8, (t0 + stride) / 8, ...
// (stride a multiple of 8; ipm_err: which is which)
__device__ __forceinline__ void ipm_error_accumulate(const KktDev& K, const double* __restrict__ V, int nV,
                                                     const double* __restrict__ x, const double* __restrict__ s,
                                                     const double* __restrict__ y, const double* __restrict__ z,
                                                     const double* __restrict__ scales, int check_all_V, int t0, int stride,
                                                     double (&acc)[kIpmErrQ]) {
  using namespace ipm_err;
  constexpr int NQ = kIpmErrQ;
  const int ops[NQ] = SLPX_IPM_ERR_OPS;
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = ops[q] == IPM_MIN ? 1.0 : 0.0;
  acc[SZ_MIN] = 1e300;
  const double inv_f = 1.0 / scales[0];
  const double* d_ce = scales + 1;
  const double* d_ci = scales + 1 + K.m_e;
  const double* ce = V + K.off_ce;
  const double* ci = V + K.off_ci;
  const double* Ae = V + K.off_Ae;
  const double* Ai = V + K.off_Ai;
  // columns: EIGHT LANES per column, an entry of the column each (a column of A_e has 5-9 entries: walked
  // by one lane they were as many dependent trips to memory; this is two), summed by DPP in lane order
  const int lane8 = t0 & 7;
  for (int j = t0 >> 3; j < K.n; j += stride >> 3) {
    double a1 = 0.0, a1u = 0.0, aetce = 0.0;
    for (int q = K.ae_colptr[j] + lane8; q < K.ae_colptr[j + 1]; q += 8) {
      const int r = K.ae_rowidx[q];
      const double a = Ae[q], dr = d_ce[r];
      a1 += a * y[r];
      a1u += ((1.0 / dr) * a) * (dr * y[r] * inv_f);
      aetce += a * ce[r];
    }
    double a2 = 0.0, a2u = 0.0, aitcp = 0.0;
    for (int q = K.ai_colptr[j] + lane8; q < K.ai_colptr[j + 1]; q += 8) {
      const int r = K.ai_rowidx[q];
      const double a = Ai[q], dr = d_ci[r];
      a2 += a * z[r];
      a2u += ((1.0 / dr) * a) * (dr * z[r] * inv_f);
      aitcp += a * fmin(ci[r], 0.0);
    }
    const int gs = K.g_src[j];
    const double g = gs >= 0 ? V[gs] : 0.0;
    const double xj = x[j];
    a1 = ipm_group8_sum(a1);
    a1u = ipm_group8_sum(a1u);
    aetce = ipm_group8_sum(aetce);
    a2 = ipm_group8_sum(a2);
    a2u = ipm_group8_sum(a2u);
    aitcp = ipm_group8_sum(aitcp);
    if (lane8 == 0) {
      const double dual = (g - a1) - a2, dual_u = (inv_f * g - a1u) - a2u;
      acc[DUAL] = fmax(acc[DUAL], fabs(dual));
      acc[DUAL_U] = fmax(acc[DUAL_U], fabs(dual_u));
      acc[AETCE] += aetce * aetce;
      acc[AITCP] += aitcp * aitcp;
      acc[XINF] = fmax(acc[XINF], fabs(xj));
      if (!isfinite(xj)) acc[FINITE] = 0.0;
    }
  }
  for (int r = t0; r < K.m_e; r += stride) {
    const double c = ce[r], dr = d_ce[r], yr = y[r];
    acc[CE] = fmax(acc[CE], fabs(c));
    acc[CE_U] = fmax(acc[CE_U], fabs((1.0 / dr) * c));
    acc[Y1] += fabs(yr);
    acc[Y1_U] += fabs(dr * yr * inv_f);
    acc[CESQ] += c * c;
    acc[VIOL] += fabs(c);
    if (!isfinite(c)) acc[FINITE] = 0.0;
  }
  for (int r = t0; r < K.m_i; r += stride) {
    const double c = ci[r], dr = d_ci[r], sr = s[r], zr = z[r];
    const double inv = 1.0 / dr, su = inv * sr, zu = dr * zr * inv_f;
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
  }
  if (check_all_V)
    for (int k = t0; k < nV; k += stride)
      if (!isfinite(V[k])) acc[FINITE] = 0.0;
}

// LDS of a workgroup of the error launch, in doubles: the transposed partials (or a sum's tree), then the fold's
constexpr int kIpmErrLdsDoubles = kIpmErrQ * (kIpmErrThreads + 1) + 9 * kIpmErrQ + 2;

// Workgroup `block` of [error workgroups | one per separable sum] of the one-launch error computation
// (fin.n_err_blocks != 0: whichever workgroup finishes last folds the partials and publishes, ipm_error_finish) or
// of the partials alone (0: ipm_error_final_kernel follows; `n_blocks_two_launch` of them).  EVERY thread of the
// calling workgroup calls — it may be larger than kIpmErrThreads (the step kernel's, when the launch rides in it:
// ldlt_mf_kernels.h): the first kIpmErrThreads lanes work, the others keep the barriers company.
__device__ __forceinline__ void ipm_error_block(const KktDev& K, const double* __restrict__ V, int nV, const double* __restrict__ x,
                                                const double* __restrict__ s, const double* __restrict__ y,
                                                const double* __restrict__ z, const double* __restrict__ scales, int check_all_V,
                                                double* __restrict__ partial, const IpmErrFinish& fin, int block,
                                                int n_blocks_two_launch, double* lds) {
  using namespace ipm_err;
  constexpr int NQ = kIpmErrQ;
  constexpr int kPad = kIpmErrThreads + 1;
  double* tr = lds;                 // [NQ][kPad]
  double* tot = lds + NQ * kPad;    // ipm_error_finish's
  const bool active = threadIdx.x < kIpmErrThreads;
  if (fin.n_err_blocks != 0 && block >= fin.n_err_blocks) {
    const NlpStructure::SumReduce r = fin.red[block - fin.n_err_blocks];
    const int tid = threadIdx.x;
    double* scratch = tr;
    double acc = 0.0;
    if (tid < 64)
      for (int k = tid; k < r.count; k += 64) acc += V[r.src_off + k];
    if (tid < 64) scratch[tid] = acc;
    __syncthreads();
    for (int w = 32; w > 0; w >>= 1) {
      if (tid < w) scratch[tid] += scratch[tid + w];
      __syncthreads();
    }
    if (tid == 0)
      coherent_store(&fin.Vw[r.dst], (r.scale_idx >= 0 ? fin.tape_scales[r.scale_idx] : 1.0) * scratch[0], true);
    ipm_error_finish(K, V, partial, fin, tot);
    return;
  }
  const int n_blocks = fin.n_err_blocks != 0 ? fin.n_err_blocks : n_blocks_two_launch;
  double acc[NQ];
  const int ops[NQ] = SLPX_IPM_ERR_OPS;
  if (active)
    ipm_error_accumulate(K, V, nV, x, s, y, z, scales, check_all_V, block * kIpmErrThreads + static_cast<int>(threadIdx.x),
                         n_blocks * kIpmErrThreads, acc);
  // the workgroup's partial of every quantity: the lanes' values through LDS, transposed — thread (q, sub) folds
  // 32 of the 256 values of quantity q in a fixed order, the eight subs by DPP; 23 wave reductions by butterfly
  // were ~700 instructions per wave
  {
    if (active) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) tr[q * kPad + threadIdx.x] = acc[q];
    }
    __syncthreads();
    const int q = threadIdx.x >> 3, sub = threadIdx.x & 7;
    if (q < NQ) {
      int op = ops[0];
#pragma unroll
      for (int k = 1; k < NQ; ++k)
        if (q == k) op = ops[k];
      // (all 32 values requested at once, and the three kinds of fold side by side — the kind is picked once at the
      // end: as a loop of 31 load-select-combine steps this was 3.3 us of the launch's 14)
      double xv[kIpmErrThreads / 8];
#pragma unroll
      for (int k = 0; k < kIpmErrThreads / 8; ++k) xv[k] = tr[q * kPad + sub + 8 * k];
      double vs = xv[0], vmax = xv[0], vmin = xv[0];
#pragma unroll
      for (int k = 1; k < kIpmErrThreads / 8; ++k) {
        vs += xv[k];
        vmax = fmax(vmax, xv[k]);
        vmin = fmin(vmin, xv[k]);
      }
      double v = op == IPM_SUM ? vs : (op == IPM_MAX ? vmax : vmin);
      v = ipm_combine(op, v, ipm_dpp<0xB1>(v));
      v = ipm_combine(op, v, ipm_dpp<0x4E>(v));
      v = ipm_combine(op, v, ipm_dpp<0x141>(v));
      if (sub == 0) coherent_store(&partial[block * NQ + q], v, fin.n_err_blocks != 0);
    }
  }
  if (fin.n_err_blocks != 0) ipm_error_finish(K, V, partial, fin, tot);
}

__global__ __launch_bounds__(kIpmErrThreads) void ipm_error_partial_kernel(
    KktDev K, const double* __restrict__ V, int nV, const double* __restrict__ x,
    const double* __restrict__ s, const double* __restrict__ y, const double* __restrict__ z,
    const double* __restrict__ scales, int check_all_V, double* __restrict__ partial, IpmErrFinish fin) {
  __shared__ double lds[kIpmErrLdsDoubles];
  if (fin.skip != nullptr && fin.skip[0] != 0.0) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      if (fin.ctl != nullptr) {  // (a void chain decides nothing: whatever was enqueued behind it passes too)
        fin.ctl->go = 0;
        *fin.gate = 0.0;
        if (fin.go_host != nullptr) *fin.go_host = 0.0;
      }
      ipm_publish(fin.seq_dev, fin.seq_host);
    }
    return;
  }
  ipm_error_block(K, V, nV, x, s, y, z, scales, check_all_V, partial, fin, static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x), lds);
}

__global__ __launch_bounds__(256) void ipm_error_final_kernel(KktDev K, const double* __restrict__ V,
                                                             const double* __restrict__ partial, int n_blocks,
                                                             IpmErrOut* __restrict__ out,
                                                             unsigned long long* __restrict__ seq_dev,
                                                             volatile unsigned long long* seq_host) {
  __shared__ double tot[9 * kIpmErrQ];
  ipm_error_fold(K, V, partial, n_blocks, false, out, seq_dev, seq_host, tot);
}

}  // namespace slpx
