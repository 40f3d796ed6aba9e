// gfx950 (MI355X) kernels for the interior-point Newton step, and the DeviceNlp
// host object that owns their buffers.  See DESIGN.md §3 for the roofline of each.
//
//   tape_sweep     (tape_kernels.h) AD refresh: forward values + local partials, then
//                  the per-row adjoint gather, all inside LDS (one workgroup per task)
//                  replaces update_values/append_triplets/setFromTriplets
//                  (expression_graph.hpp:86-153, jacobian.hpp:134-156, hessian.hpp:132-157)
//   kkt_assemble   lhs values by gather   (interior_point.hpp:426-440)
//   kkt_rhs        rhs by column gathers  (interior_point.hpp:444-448)
//   ldlt_factor    (ldlt_kernels.h) one round of the task-parallel left-looking LDLᵀ
//                  + inertia (sparse_regularized_ldlt.hpp:74-83,105-109, inertia.hpp:40-50)
//   ldlt_fwd/bwd   (ldlt_kernels.h) triangular solves (sparse_regularized_ldlt.hpp:159-161)
//   step_backsub   pˢ, pᶻ                 (interior_point.hpp:479-480)
#include <hip/hip_runtime.h>

#include <chrono>

#include <algorithm>
#include <bit>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

#include "device.hpp"
#include "kkt_kernels.h"
#include "ldlt_kernels.h"
#include "ldlt_mf_kernels.h"
#include "ldlt_dense_kernels.h"
#include "setup_threads.hpp"
#include "ldlt_il_kernels.h"
#include "tape_jit.hpp"
#include "ipm_decide.h"
#include "ipm_kernels.h"
#include "tape_kernels.h"
#include "tape_ops.h"
#include "setup_timing.hpp"

namespace slpx {

// ============================================================================
// kkt_assemble / kkt_rhs / step_backsub: pure gathers over static index maps.
// HBM-bound; algorithmic bytes: assemble 12(h+a+i) + 16 m_i + 8 k, rhs
// 16 n + 24 m_e + 24 m_i + 12(a+i).  One thread per output entry,
// consecutive threads write consecutive entries; the sources of consecutive lhs
// entries are (nearly) consecutive in V because both are CSC-ordered.
// ============================================================================
__global__ __launch_bounds__(256) void kkt_assemble_kernel(KktDev K, const double* __restrict__ V,
                                                           int v_stride,
                                                           const double* __restrict__ s,
                                                           const double* __restrict__ z,
                                                           double* __restrict__ lhs) {
  const int b = blockIdx.y;
  kkt_assemble_body(K, V + static_cast<size_t>(b) * v_stride, s + static_cast<size_t>(b) * K.m_i,
                    z + static_cast<size_t>(b) * K.m_i, lhs + static_cast<size_t>(b) * K.nnz_lhs, blockIdx.x,
                    gridDim.x);
}

// Batch variant: one thread serves the SAME entry of kBatchPerThread consecutive problems,
// so every index of the static maps (fast_src, dptr/dsrc, pptr/pa/pb/pr) is fetched once per
// kBatchPerThread problems instead of once per problem — the maps are a fifth of the bytes
// a thread touches — and the problems' gathers are in flight together.
constexpr int kBatchPerThread = 4;

template <int P>
__global__ __launch_bounds__(256) void kkt_assemble_batch_kernel(KktDev K, const double* __restrict__ V,
                                                                 int v_stride,
                                                                 const double* __restrict__ s,
                                                                 const double* __restrict__ z,
                                                                 double* __restrict__ lhs, int batch) {
  const int b0 = blockIdx.y * P;
  const int nb = min(P, batch - b0);
  V += static_cast<size_t>(b0) * v_stride;
  s += static_cast<size_t>(b0) * K.m_i;
  z += static_cast<size_t>(b0) * K.m_i;
  lhs += static_cast<size_t>(b0) * K.nnz_lhs;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K.nnz_lhs; k += gridDim.x * blockDim.x) {
    const int f = K.fast_src[k];
    double v[P];
#pragma unroll
    for (int q = 0; q < P; ++q) v[q] = 0.0;
    if (f >= 0) {
#pragma unroll
      for (int q = 0; q < P; ++q)
        if (q < nb) v[q] = V[static_cast<size_t>(q) * v_stride + f];
    } else if (f == -2) {
      double prod[P];
#pragma unroll
      for (int q = 0; q < P; ++q) prod[q] = 0.0;
      for (int d = K.dptr[k]; d < K.dptr[k + 1]; ++d) {
        const int src = K.dsrc[d];
#pragma unroll
        for (int q = 0; q < P; ++q)
          if (q < nb) v[q] += V[static_cast<size_t>(q) * v_stride + src];
      }
      for (int p = K.pptr[k]; p < K.pptr[k + 1]; ++p) {
        const int r = K.pr[p], a = K.pa[p], bb = K.pb[p];
#pragma unroll
        for (int q = 0; q < P; ++q)
          if (q < nb) {
            const double* Vq = V + static_cast<size_t>(q) * v_stride;
            prod[q] += kkt_prod_term(Vq[a], s[q * K.m_i + r], z[q * K.m_i + r], Vq[bb]);
          }
      }
#pragma unroll
      for (int q = 0; q < P; ++q) v[q] += prod[q];
    }
#pragma unroll
    for (int q = 0; q < P; ++q)
      if (q < nb) lhs[static_cast<size_t>(q) * K.nnz_lhs + k] = v[q];
  }
}

// The batch-interleaved factorization's inputs written WHERE IT READS THEM (ldlt_il_kernels.h:
// [b / 16][entry][16]) — the batch-major lhs / rhs and the two transposing launches between the
// assembly and the factorization (il_gather_kernel: 0.084 ms of a 1.15 ms step at 512 x N=1000) are
// only made when somebody asks for them (DeviceNlp::d_lhs / d_rhs).
// lhs: a thread = one entry of FOUR consecutive problems, four adjacent lanes = the 16 problems of
// the entry's row: every 128-byte row is written whole by one quad, and a wave's loads of one
// problem's V are sixteen consecutive entries' sources.
__device__ __forceinline__ void kkt_assemble_il_body(const KktDev& K, const double* __restrict__ V, int v_stride,
                                                     const double* __restrict__ s, const double* __restrict__ z,
                                                     double* __restrict__ lhs_il, int batch, int vblock, int vgrid) {
  constexpr int P = 4;
  const int g = blockIdx.y, sub = threadIdx.x & 3;
  const int b0 = g * kIlW + sub * P;
  const int nb = max(0, min(P, batch - b0));
  const size_t bb0 = static_cast<size_t>(nb > 0 ? b0 : 0);  // (no problem of this quad exists: nothing is dereferenced)
  V += bb0 * v_stride;
  s += bb0 * K.m_i;
  z += bb0 * K.m_i;
  double* out = lhs_il + static_cast<size_t>(g) * K.nnz_lhs * kIlW + sub * P;
  for (int k = vblock * 64 + (threadIdx.x >> 2); k < K.nnz_lhs; k += vgrid * 64) {
    const int f = K.fast_src[k];
    double v[P];
#pragma unroll
    for (int q = 0; q < P; ++q) v[q] = 0.0;
    if (f >= 0) {
#pragma unroll
      for (int q = 0; q < P; ++q)
        if (q < nb) v[q] = V[static_cast<size_t>(q) * v_stride + f];
    } else if (f == -2) {
      double prod[P];
#pragma unroll
      for (int q = 0; q < P; ++q) prod[q] = 0.0;
      for (int d = K.dptr[k]; d < K.dptr[k + 1]; ++d) {
        const int src = K.dsrc[d];
#pragma unroll
        for (int q = 0; q < P; ++q)
          if (q < nb) v[q] += V[static_cast<size_t>(q) * v_stride + src];
      }
      for (int p = K.pptr[k]; p < K.pptr[k + 1]; ++p) {
        const int r = K.pr[p], a = K.pa[p], c = K.pb[p];
#pragma unroll
        for (int q = 0; q < P; ++q)
          if (q < nb) {
            const double* Vq = V + static_cast<size_t>(q) * v_stride;
            prod[q] += kkt_prod_term(Vq[a], s[q * K.m_i + r], z[q * K.m_i + r], Vq[c]);
          }
      }
#pragma unroll
      for (int q = 0; q < P; ++q) v[q] += prod[q];
    }
    double2* o = reinterpret_cast<double2*>(out + static_cast<size_t>(k) * kIlW);
    o[0] = double2{v[0], v[1]};
    o[1] = double2{v[2], v[3]};
  }
}

__global__ __launch_bounds__(256) void kkt_assemble_il_kernel(KktDev K, const double* __restrict__ V, int v_stride,
                                                              const double* __restrict__ s, const double* __restrict__ z,
                                                              double* __restrict__ lhs_il, int batch) {
  kkt_assemble_il_body(K, V, v_stride, s, z, lhs_il, batch, blockIdx.x, gridDim.x);
}

// rhs: 64 rows x 16 problems per workgroup through LDS — a row is computed by one lane per problem
// group of four (consecutive lanes = consecutive rows: the row's sources in V are nearly consecutive),
// written as whole 128-byte rows.
__device__ __forceinline__ void kkt_rhs_il_body(const KktDev& K, const double* __restrict__ V, int v_stride,
                                                const double* __restrict__ s, const double* __restrict__ y,
                                                const double* __restrict__ z, const double* __restrict__ mu,
                                                double* __restrict__ rhs_il, int batch, int vblock) {
  __shared__ double tile[kIlW][65];
  const int g = blockIdx.y, i0 = vblock * 64;
  const int il = threadIdx.x & 63, bq = threadIdx.x >> 6;
  const int j = i0 + il;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int pl = bq * 4 + q, b = g * kIlW + pl;
    double val = 0.0;
    if (b < batch && j < K.dim) {
      const size_t sb = static_cast<size_t>(b);
      val = kkt_rhs_entry(K, V + sb * v_stride, s + sb * K.m_i, y + sb * K.m_e, z + sb * K.m_i, mu[b], j);
    }
    tile[pl][il] = val;
  }
  __syncthreads();
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int idx = threadIdx.x + 256 * jj, pl = idx & (kIlW - 1), r = idx >> kIlWShift;
    if (i0 + r < K.dim) rhs_il[(static_cast<size_t>(g) * K.dim + i0 + r) * kIlW + pl] = tile[pl][r];
  }
}
__global__ __launch_bounds__(256) void kkt_rhs_il_kernel(KktDev K, const double* __restrict__ V, int v_stride,
                                                         const double* __restrict__ s, const double* __restrict__ y,
                                                         const double* __restrict__ z, const double* __restrict__ mu,
                                                         double* __restrict__ rhs_il, int batch) {
  kkt_rhs_il_body(K, V, v_stride, s, y, z, mu, rhs_il, batch, blockIdx.x);
}
// lhs and rhs in ONE launch (a Newton step builds both: neither reads the other's output) — workgroups
// [0, na) assemble, the rest take 64 rows of the right-hand side each.  A small batch's two launches are
// latency each (64 x N=500: 15 + 16 us); side by side they cost the longer one.
__global__ __launch_bounds__(256) void kkt_build_il_kernel(KktDev K, const double* __restrict__ V, int v_stride,
                                                           const double* __restrict__ s, const double* __restrict__ y,
                                                           const double* __restrict__ z, const double* __restrict__ mu,
                                                           double* __restrict__ lhs_il, double* __restrict__ rhs_il, int batch,
                                                           int na) {
  if (static_cast<int>(blockIdx.x) < na) kkt_assemble_il_body(K, V, v_stride, s, z, lhs_il, batch, blockIdx.x, na);
  else kkt_rhs_il_body(K, V, v_stride, s, y, z, mu, rhs_il, batch, static_cast<int>(blockIdx.x) - na);
}

// Least-squares multiplier estimate (util/lagrange_multiplier_estimate.hpp:56-133) on the
// KKT pattern: the top-left block is I + A_i^T S^-2 A_i (H sources skipped, identity added
// by kkt_add_identity_kernel), the A_e block is unchanged.
__global__ __launch_bounds__(256) void kkt_assemble_lsq_kernel(KktDev K, const double* __restrict__ V,
                                                               int v_stride,
                                                               const double* __restrict__ s,
                                                               double* __restrict__ lhs) {
  const int b = blockIdx.y;
  V += static_cast<size_t>(b) * v_stride;
  s += static_cast<size_t>(b) * K.m_i;
  lhs += static_cast<size_t>(b) * K.nnz_lhs;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K.nnz_lhs; k += gridDim.x * blockDim.x) {
    double direct = 0.0;
    for (int d = K.dptr[k]; d < K.dptr[k + 1]; ++d) {
      const int src = K.dsrc[d];
      if (src >= K.off_Ae && src < K.off_Ai) direct += V[src];
    }
    double prod = 0.0;
    for (int p = K.pptr[k]; p < K.pptr[k + 1]; ++p) {
      const double sinv = 1.0 / s[K.pr[p]];
      prod += (V[K.pa[p]] * (sinv * sinv)) * V[K.pb[p]];
    }
    lhs[k] = direct + prod;
  }
}

__global__ __launch_bounds__(256) void kkt_add_identity_kernel(KktDev K, const int32_t* __restrict__ diag_pos,
                                                               double* __restrict__ lhs) {
  lhs += static_cast<size_t>(blockIdx.y) * K.nnz_lhs;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < K.n; j += gridDim.x * blockDim.x)
    lhs[diag_pos[j]] += 1.0;
}

__global__ __launch_bounds__(256) void kkt_rhs_kernel(KktDev K, const double* __restrict__ V,
                                                      int v_stride, const double* __restrict__ s,
                                                      const double* __restrict__ y,
                                                      const double* __restrict__ z,
                                                      const double* __restrict__ mu,
                                                      double* __restrict__ rhs) {
  const int b = blockIdx.y;
  kkt_rhs_body(K, V + static_cast<size_t>(b) * v_stride, s + static_cast<size_t>(b) * K.m_i,
               y + static_cast<size_t>(b) * K.m_e, z + static_cast<size_t>(b) * K.m_i, mu[b],
               rhs + static_cast<size_t>(b) * K.dim, blockIdx.x, gridDim.x);
}

// lhs, rhs and the separable-sum reductions of the tape in ONE launch: all three only read
// what the sweep left in V, none reads another's output (the sums feed f, which nothing in
// the Newton step consumes).  Block ranges: [0, na) assembly, [na, na + nr) right-hand
// side, the rest one reduction each.
__global__ __launch_bounds__(256) void kkt_build_kernel(KktDev K, double* __restrict__ V, int v_stride,
                                                        const double* __restrict__ s,
                                                        const double* __restrict__ y,
                                                        const double* __restrict__ z,
                                                        const double* __restrict__ mu,
                                                        double* __restrict__ lhs, double* __restrict__ rhs,
                                                        int na, int nr,
                                                        const NlpStructure::SumReduce* __restrict__ red,
                                                        const double* __restrict__ scales) {
  __shared__ double part[64];
  const int b = blockIdx.y;
  V += static_cast<size_t>(b) * v_stride;
  s += static_cast<size_t>(b) * K.m_i;
  z += static_cast<size_t>(b) * K.m_i;
  const int blk = blockIdx.x;
  if (blk < na) {
    kkt_assemble_body(K, V, s, z, lhs + static_cast<size_t>(b) * K.nnz_lhs, blk, na);
  } else if (blk < na + nr) {
    kkt_rhs_body(K, V, s, y + static_cast<size_t>(b) * K.m_e, z, mu[b], rhs + static_cast<size_t>(b) * K.dim,
                 blk - na, nr);
  } else {
    // tape_reduce_body wants a 64-lane workgroup: the first wave does it, the others idle
    // at its barriers
    const NlpStructure::SumReduce r = red[blk - na - nr];
    const int tid = threadIdx.x;
    double acc = 0.0;
    if (tid < 64)
      for (int k = tid; k < r.count; k += 64) acc += V[r.src_off + k];
    if (tid < 64) part[tid] = acc;
    __syncthreads();
    for (int w = 32; w > 0; w >>= 1) {
      if (tid < w) part[tid] += part[tid + w];
      __syncthreads();
    }
    if (tid == 0) V[r.dst] = (r.scale_idx >= 0 ? scales[r.scale_idx] : 1.0) * part[0];
  }
}

__global__ __launch_bounds__(256) void step_backsub_kernel(KktDev K, const double* __restrict__ V,
                                                           int v_stride,
                                                           const double* __restrict__ p,
                                                           const double* __restrict__ s,
                                                           const double* __restrict__ z,
                                                           const double* __restrict__ mu,
                                                           double* __restrict__ ps,
                                                           double* __restrict__ pz,
                                                           const LdltStats* __restrict__ stats_src,
                                                           LdltStats* __restrict__ stats_host,
                                                           unsigned long long* __restrict__ seq_dev,
                                                           volatile unsigned long long* seq_host) {
  const int b = blockIdx.y;
  // last kernel of a step: hand the inertia counters of the factorization to the host
  // (pinned memory) instead of spending a copy node on them; with one problem also a
  // sequence number the host spins on — it learns of the result a few microseconds before
  // the stream would report the kernel complete, and what it does next (another attempt or
  // the next step) is ordered behind this kernel by the stream anyway
  if (stats_host != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
    stats_host[b] = stats_src[b];
    if (seq_host != nullptr) {
      __threadfence_system();
      const unsigned long long v = *seq_dev + 1;
      *seq_dev = v;
      *seq_host = v;
    }
  }
  V += static_cast<size_t>(b) * v_stride;
  p += static_cast<size_t>(b) * K.dim;
  s += static_cast<size_t>(b) * K.m_i;
  z += static_cast<size_t>(b) * K.m_i;
  ps += static_cast<size_t>(b) * K.m_i;
  pz += static_cast<size_t>(b) * K.m_i;
  step_backsub_body(K, V, p, s, z, mu[b], ps, pz, static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x),
                    static_cast<int>(gridDim.x * blockDim.x));
}

// Iterative refinement of a solve (used where the regularization would otherwise show in the
// answer: the least-squares multiplier estimate): r = b - K p for the UNREGULARIZED symmetric K
// given by its lower CSC values in `lhs`.  One thread per row i: the column i of the lower
// triangle gives K(r, i), r >= i, and a row index of the same triangle (rowptr / rowent / rowcol)
// gives K(i, c), c < i — a fixed order of summation, so the same bits on every run (scattering
// the mirrored half with floating-point atomics made the multiplier estimate, and with it the
// whole trajectory of a solve, differ from run to run in the last bits).
__global__ __launch_bounds__(256) void sym_residual_kernel(int dim, const int32_t* __restrict__ colptr,
                                                           const int32_t* __restrict__ rowidx,
                                                           const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ rowent,
                                                           const int32_t* __restrict__ rowcol,
                                                           const double* __restrict__ lhs,
                                                           const double* __restrict__ p,
                                                           double* __restrict__ res) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int q = rowptr[i]; q < rowptr[i + 1]; ++q) acc += lhs[rowent[q]] * p[rowcol[q]];  // K(i, c), c < i
    for (int q = colptr[i]; q < colptr[i + 1]; ++q) acc += lhs[q] * p[rowidx[q]];          // K(r, i), r >= i
    res[i] -= acc;
  }
}
__global__ __launch_bounds__(256) void axpy_kernel(int count, const double* __restrict__ x, double* __restrict__ y) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) y[i] += x[i];
}

// Separable sums on their own (tape_reduce_body, tape_kernels.h) — used when the reductions
// cannot ride along with the interpreted tail of the tape (launch_tape).
__global__ __launch_bounds__(64) void tape_reduce_kernel(const NlpStructure::SumReduce* __restrict__ red,
                                                         const double* __restrict__ scales,
                                                         double* __restrict__ V, int v_stride) {
  __shared__ double part[64];
  V += static_cast<size_t>(blockIdx.y) * v_stride;
  tape_reduce_body(red[blockIdx.x], scales, V, part, threadIdx.x);
}

// Can two kernels of this process run at once?  Chained steps (DeviceNlp::sweep_full_for_step) need the
// sweep and the step kernel resident together; a profiler collecting hardware counters (rocprofv3 --pmc)
// or AMD_SERIALIZE_KERNEL runs one kernel at a time, and a step kernel waiting for a sweep that cannot
// start would sit out its spin bound (seen: 247 ms per step under --pmc).  So, once per system: a kernel
// on the sweep's stream waits — a few milliseconds at most — for a word a kernel on the main stream sets.
__global__ void chain_probe_wait_kernel(unsigned int* w) {
  unsigned int seen = 0;
  for (int k = 0; k < 3000 && !seen; ++k) {
    seen = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_sleep(32);
  }
  w[1] = seen ? 1u : 2u;
}
__global__ void chain_probe_set_kernel(unsigned int* w) { __hip_atomic_store(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ============================================================================
// DeviceNlp
// ============================================================================

void TapeDevice::upload(const TapeProgram& p, int batch, uint32_t n_unscaled_inputs, int chain_mode) {
  tasks.upload(p.tasks);
  // families of structurally identical tasks and the big singles get a body in the
  // generated lane-per-task kernel; everything else is interpreted
  TapeJitOptions jit_opt;
  jit_opt.n_unscaled_inputs = n_unscaled_inputs;  // x: set_scaling leaves their factor at 1
  jit_opt.chain_mode = chain_mode;
  const TapeJitResult jit = build_tape_templates(p, jit_opt);
  jit_seconds = jit.compile_seconds;
  tmpl_fn = jit.fn;
  tmpl_params = jit.params;
  {
    std::vector<uint64_t> words((tmpl_params.size() + 7) / 8, 0);
    std::memcpy(words.data(), tmpl_params.data(), tmpl_params.size());
    tmpl_params_dev.upload(words);
  }
  tmpl_mod = jit.mod;
  tmpl_threads = jit.block_threads;
  n_bodies = static_cast<uint32_t>(jit.groups.size());
  n_templated_tasks = 0;
  tmpl_blocks[0] = tmpl_blocks[1] = 0;
  if (n_bodies) {
    std::vector<uint32_t> inst, table[2];
    for (const TapeTemplateGroup& g : jit.groups) {
      // the family's leaf bindings and output destinations, transposed: row r of instance i at
      // [r * n_inst + i] (rows: leaves | value dst | value scale | jacobian dst | jacobian scale),
      // so that the lanes of a wave — consecutive instances — read consecutive words
      const uint32_t n_inst = static_cast<uint32_t>(g.tasks.size()), inst_off = static_cast<uint32_t>(inst.size());
      const TapeTask& rep = p.tasks[g.tasks.front()];
      const uint32_t rows = rep.n_leaf + 2 * rep.n_vout + 2 * rep.n_jout;
      inst.resize(inst.size() + static_cast<size_t>(rows) * n_inst);
      for (uint32_t i = 0; i < n_inst; ++i) {
        const TapeTask& t = p.tasks[g.tasks[i]];
        uint32_t* col = inst.data() + inst_off + i;
        uint32_t r = 0;
        for (uint32_t q = 0; q < rep.n_leaf; ++q) col[static_cast<size_t>(r++) * n_inst] = p.leaf_src[t.leaf_off + q];
        for (uint32_t q = 0; q < rep.n_vout; ++q) col[static_cast<size_t>(r++) * n_inst] = p.vout_dst[t.vout_off + q];
        for (uint32_t q = 0; q < rep.n_vout; ++q)
          col[static_cast<size_t>(r++) * n_inst] = static_cast<uint32_t>(p.vout_scale[t.vout_off + q]);
        for (uint32_t q = 0; q < rep.n_jout; ++q) col[static_cast<size_t>(r++) * n_inst] = p.jout_dst[t.jout_off + q];
        for (uint32_t q = 0; q < rep.n_jout; ++q)
          col[static_cast<size_t>(r++) * n_inst] = static_cast<uint32_t>(p.jout_scale[t.jout_off + q]);
      }
      n_templated_tasks += n_inst;
      // Adjoint rows are split into wave-uniform groups while the launch would otherwise
      // leave most SIMDs idle; with enough instances x batch items every lane runs all
      // groups (-1) and the forward part is not recomputed per group.
      const uint32_t waves = (n_inst + tmpl_threads - 1) / tmpl_threads;  // workgroups of one row group
      const bool split = ((n_inst + 63) / 64) * static_cast<uint32_t>(batch) < 512 && g.n_groups > 1;
      const int mode[2] = {0, split ? static_cast<int>(g.n_groups) : -1};
      for (int m = 0; m < 2; ++m) {
        table[m].push_back(tmpl_blocks[m]);
        table[m].push_back(n_inst);
        table[m].push_back(inst_off);
        table[m].push_back(static_cast<uint32_t>(mode[m]));
        tmpl_blocks[m] += waves * static_cast<uint32_t>(std::max(1, mode[m]));
      }
    }
    tmpl_inst.upload(inst);
    tmpl_table[0].upload(table[0]);
    tmpl_table[1].upload(table[1]);
  }
  auto interpreted = [&](const std::vector<uint32_t>& list) {
    std::vector<uint32_t> out;
    for (uint32_t ti : list)
      if (!jit.task_is_templated[ti]) out.push_back(ti);
    return out;
  };
  std::vector<uint32_t> small_rest = interpreted(p.small_tasks), large_rest = interpreted(p.large_tasks);
  uint32_t ride_lds = p.small_lds_bytes;
  tmpl_wide_for_chain = n_bodies && tmpl_threads == 256 && large_rest.empty() && chain_mode != 0;
  if (n_bodies && tmpl_threads == 256) {
    // (tape_jit.cpp: the generated kernel's workgroups are 256 threads: every interpreted task rides in its launch)
    small_rest.insert(small_rest.end(), large_rest.begin(), large_rest.end());
    large_rest.clear();
    ride_lds = std::max(p.small_lds_bytes, p.large_lds_bytes);
  }
  small_list.upload(small_rest);
  large_list.upload(large_rest);
  global_list.upload(p.global_tasks);
  if (std::getenv("SLPX_TAPE_JIT_VERBOSE") && jit.fn)
    std::fprintf(stderr, "slpx tape kernel: %zu bodies, %s code object (%zu bytes of model numbers), %.3f s\n",
                 jit.groups.size(), jit.specialized ? "specialized" : "generic", jit.params.size(), jit.compile_seconds);
  if (std::getenv("SLPX_TAPE_JIT_VERBOSE"))
    for (uint32_t ti : p.global_tasks)
      std::fprintf(stderr, "slpx tape GLOBAL task: leaf %u node %u slot %u vout %u jout %u levels %u+%u\n", p.tasks[ti].n_leaf,
                   p.tasks[ti].n_node, p.tasks[ti].n_slot, p.tasks[ti].n_vout, p.tasks[ti].n_jout, p.tasks[ti].n_lvl, p.tasks[ti].n_slvl);
  leaf_src.upload(p.leaf_src);
  // never empty: the generated bodies read consts[0] in lanes whose leaf is not a constant
  consts.upload(p.consts.empty() ? std::vector<double>{0.0} : p.consts);
  node_rec.upload(p.node_rec);
  lvl_ptr.upload(p.lvl_ptr);
  slot_edge_ptr.upload(p.slot_edge_ptr);
  slvl_ptr.upload(p.slvl_ptr);
  edges.upload(p.edges);
  vout_src.upload(p.vout_src);
  vout_dst.upload(p.vout_dst);
  vout_scale.upload(p.vout_scale);
  jout_slot.upload(p.jout_slot);
  jout_dst.upload(p.jout_dst);
  jout_scale.upload(p.jout_scale);
  node_rec16.upload(p.node_rec16);
  slot_edge_ptr16.upload(p.slot_edge_ptr16);
  edges16.upload(p.edges16);
  n_small = static_cast<uint32_t>(small_rest.size());
  n_large = static_cast<uint32_t>(large_rest.size());
  n_global = static_cast<uint32_t>(p.global_tasks.size());
  small_lds = ride_lds;
  large_lds = p.large_lds_bytes;
  scratch_doubles = p.global_scratch_doubles;
  basic_ops = p.basic_ops;
}

TapeDev TapeDevice::view() const {
  return TapeDev{tasks.p,         leaf_src.p,   consts.p,   node_rec.p,   lvl_ptr.p,
                 slot_edge_ptr.p, slvl_ptr.p,   edges.p,    vout_src.p,   vout_dst.p,
                 vout_scale.p,    jout_slot.p,  jout_dst.p, jout_scale.p, node_rec16.p,
                 slot_edge_ptr16.p, edges16.p};
}

DeviceNlp::DeviceNlp(const NlpStructure& s, const KktPlan& k, const LdltPlan& l, int batch,
                     int device)
    : m_s_ref(s), m_k_ref(k), m_l_ref(l), m_batch(batch), m_device(device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
    throw std::runtime_error("slpx: no HIP device available (the product path has no CPU fallback)");
  SLPX_HIP_CHECK(hipSetDevice(device));
  SetupLap lap;
  // every upload() of this constructor goes into the system's arena: a few allocations, one copy each (commit below)
  struct ArenaGuard {
    ~ArenaGuard() { DeviceArena::current() = nullptr; }
  } arena_guard;
  DeviceArena::current() = &m_arena;

  // Chained steps (sweep_full_for_step): for one problem whose multifrontal step kernel leaves the sweep
  // room on the chip (at most CUs - 64 workgroups: cart-pole N=5000's 265 do not, and there chaining costs
  // 25 %, profiles/r03_chain_ab.txt).  SLPX_CHAIN_TAPE=0: off.
  {
    int cus = 0;
    SLPX_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m_device));
    if (batch == 1 && l.mf && static_cast<int>(l.tasks.size() + s.reduces.size()) + 64 <= cus) m_chain_mode = 1;
    if (const char* env = std::getenv("SLPX_CHAIN_TAPE"))
      if (env[0] == '0') m_chain_mode = 0;
  }
  m_full.upload(s.full, batch, static_cast<uint32_t>(s.n), m_chain_mode);
  m_values.upload(s.values, batch, static_cast<uint32_t>(s.n));
  m_reduces.upload(s.reduces);
  lap("  upload: tapes (+ code objects)");
  // allow > 64 KB dynamic LDS
  for (const void* fn : {reinterpret_cast<const void*>(&tape_sweep_lds_kernel<256, true>),
                         reinterpret_cast<const void*>(&tape_sweep_lds_kernel<256, false>),
                         reinterpret_cast<const void*>(&tape_sweep_lds_kernel<64, true>),
                         reinterpret_cast<const void*>(&tape_sweep_lds_kernel<64, false>)})
    SLPX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  SLPX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ldlt_factor_kernel<256>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  SLPX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ldlt_factor_kernel<1024>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  SLPX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ldlt_fwd_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  SLPX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ldlt_bwd_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

  m_dptr.upload(k.dptr);
  m_dsrc.upload(k.dsrc);
  m_pptr.upload(k.pptr);
  m_pa.upload(k.pa);
  m_pb.upload(k.pb);
  m_pr.upload(k.pr);
  m_gsrc.upload(k.g_src);
  m_ae_colptr.upload(s.Ae.colptr);
  m_ae_rowidx.upload(s.Ae.rowidx);
  m_ai_colptr.upload(s.Ai.colptr);
  m_ai_rowidx.upload(s.Ai.rowidx);
  m_ai_rowptr.upload(k.ai_rowptr);
  m_ai_col.upload(k.ai_col);
  m_ai_src.upload(k.ai_src);
  m_fast_src.upload(k.fast_src);
  {
    std::vector<int32_t> diag_pos(std::max(1, k.n), 0);
    for (int c = 0; c < k.n; ++c)
      for (int p = k.lhs.colptr[c]; p < k.lhs.colptr[c + 1]; ++p)
        if (k.lhs.rowidx[p] == c) diag_pos[c] = p;
    m_diag_pos.upload(diag_pos);
  }
  m_kdev = KktDev{k.n,           k.m_e,        k.m_i,        k.dim,         k.lhs.nnz(),  m_dptr.p,
                  m_dsrc.p,      m_pptr.p,     m_pa.p,       m_pb.p,        m_pr.p,       m_gsrc.p,
                  m_ae_colptr.p, m_ae_rowidx.p, m_ai_colptr.p, m_ai_rowidx.p, m_ai_rowptr.p,
                  m_ai_col.p,    m_ai_src.p,   m_fast_src.p, s.off_f,      s.off_ce,     s.off_ci,     s.off_g,
                  s.off_Ae,      s.off_Ai};

  lap("  upload: kernel attributes, KKT plan");
  m_ltasks.upload(l.tasks);
  m_ent_src.upload(l.ent_src);
  m_ent_flags.upload(l.ent_flags);
  m_ent_col.upload(l.ent_col);
  m_ent_out.upload(l.ent_out);
  m_ent_pair_ptr.upload(l.ent_pair_ptr);
  m_ent_contrib_ptr.upload(l.ent_contrib_ptr);
  m_contrib_idx.upload(l.contrib_idx);
  m_ext_dst.upload(l.ext_dst);
  m_llvl_ptr.upload(l.lvl_ptr);
  m_pairs.upload(l.pairs);
  m_col_perm.upload(l.col_perm);
  m_col_lvl_ptr.upload(l.col_lvl_ptr);
  m_fwd_ptr.upload(l.fwd_ptr);
  m_fwd_contrib_ptr.upload(l.fwd_contrib_ptr);
  m_scontrib_idx.upload(l.scontrib_idx);
  m_fwd_items.upload(l.fwd_items);
  m_sext_ptr.upload(l.sext_ptr);
  m_sext_dst.upload(l.sext_dst);
  m_sext_items.upload(l.sext_items);
  m_bwd_ptr.upload(l.bwd_ptr);
  m_bwd_items.upload(l.bwd_items);
  m_perm.upload(l.perm);
  m_ldev = LdltDev{m_ltasks.p,       m_ent_src.p,  m_ent_flags.p,       m_ent_col.p,
                   m_ent_out.p,      m_ent_pair_ptr.p, m_ent_contrib_ptr.p, m_contrib_idx.p,
                   m_ext_dst.p,      m_llvl_ptr.p, m_pairs.p,           m_col_perm.p,
                   m_col_lvl_ptr.p,  m_fwd_ptr.p,  m_fwd_contrib_ptr.p, m_scontrib_idx.p,
                   m_fwd_items.p,    m_sext_ptr.p, m_sext_dst.p,        m_sext_items.p,
                   m_bwd_ptr.p,      m_bwd_items.p, m_perm.p,            nullptr,
                   l.n_rounds};
  m_round_ptr.upload(l.round_ptr);
  m_ldev.round_ptr = m_round_ptr.p;
  m_ldev.clock_task = 0xffffffffu;
  m_sn_desc.upload(l.sn_desc);
  m_sn_lvl_ptr.upload(l.sn_lvl_ptr);
  m_col_sn.upload(l.col_sn);
  m_ldev.sn_desc = m_sn_desc.p;
  m_ldev.sn_lvl_ptr = m_sn_lvl_ptr.p;
  m_ldev.col_sn = m_col_sn.p;
  {
    std::vector<uint32_t> lp(l.lvl_ptr.size()), cp(l.col_lvl_ptr.size());
    for (size_t i = 0; i < lp.size(); ++i) {
      if (l.lvl_ptr[i] > 0xffffu || l.col_lvl_ptr[i] > 0xffffu || l.sn_lvl_ptr[i] > 0xffffu)
        throw std::runtime_error("slpx: an LDLT task exceeds the 16-bit packing of its level table");
      lp[i] = l.lvl_ptr[i] | (l.sn_lvl_ptr[i] << 16);
      cp[i] = l.col_lvl_ptr[i] | (l.sn_lvl_ptr[i] << 16);
    }
    m_lvl_pack.upload(lp);
    m_col_lvl_pack.upload(cp);
    std::vector<uint2> rng(l.col_perm.size(), uint2{0, 0});
    for (const LdltTask& t : l.tasks)
      for (uint32_t i = 0; i < t.n_col; ++i) {
        const uint32_t cs = l.col_sn[t.col_off + i];
        rng[t.col_off + i] = uint2{l.bwd_ptr[t.colptr_off + i] + ((cs >> 8) - (cs & 0xffu) - 1u),
                                   l.bwd_ptr[t.colptr_off + i + 1]};
      }
    m_bwd_range.upload(rng);
  }
  m_ldev.lvl_pack = m_lvl_pack.p;
  m_ldev.col_lvl_pack = m_col_lvl_pack.p;
  m_ldev.bwd_range = m_bwd_range.p;
  {
    std::vector<unsigned int> zero(2 * static_cast<size_t>(batch) * std::max(1, l.n_rounds), 0u);
    m_fround_cnt.upload(zero);
    m_bround_cnt.upload(zero);
  }
  // Rounds in one launch pay off while every task of the launch is resident at once (a
  // single problem: 137 tasks on 256 CUs).  With a batch the later-round workgroups would
  // spin on CUs the earlier rounds of other problems are waiting for (measured at batch
  // 512: factorization 0.93 -> 4.2 ms), so batches keep one launch per round.
  lap("  upload: LDLT plan");
  m_dense = l.dense;
  m_il = !m_dense && interleaved_for(batch);
  m_single_launch = !m_il && static_cast<size_t>(batch) * l.tasks.size() <= 1024;
  if (const char* env = std::getenv("SLPX_SINGLE_LAUNCH")) m_single_launch = env[0] != '0';
  // launch fusion (device.hpp: KktFuse / BacksubFuse): one problem, single-launch factorization
  m_fuse_launches = m_single_launch && batch == 1 && l.factor_lds_bytes >= 64 * sizeof(double);
  if (const char* env = std::getenv("SLPX_FUSE_LAUNCHES")) m_fuse_launches = m_fuse_launches && env[0] != '0';
  m_fuse_kkt = m_fuse_backsub = m_fuse_launches;
  if (const char* env = std::getenv("SLPX_FUSE_KKT")) m_fuse_kkt = m_fuse_kkt && env[0] != '0';
  if (const char* env = std::getenv("SLPX_FUSE_BACKSUB")) m_fuse_backsub = m_fuse_backsub && env[0] != '0';
  if (const char* env = std::getenv("SLPX_FUSE_KKT_STORE")) m_fuse_kkt_store = env[0] != '0';
  if (m_fuse_kkt) build_inline_kkt(s, k, l);
  lap("    images: inline KKT terms");
  if (m_fuse_backsub) build_inline_backsub(k, l);
  lap("    images: back-substitution rows");
  // (one problem's factorization and solve in one launch: the multifrontal step; the pair-list kernels — batches,
  // plans without fronts — take a launch per round and phase)
  if (m_fuse_launches && m_fuse_backsub && m_fuse_kkt && l.mf) build_mf(l);
  lap("    images: multifrontal task images");
  m_fuse_solve = m_mf;
  m_chain_on = m_mf && m_chain_mode != 0;
  lap("  upload: inline KKT / back-substitution / multifrontal images");

  const size_t B = static_cast<size_t>(batch);
  m_in.alloc(B * s.n_inputs());
  m_in.zero();
  m_V.alloc(B * s.nV);
  m_s.alloc(B * std::max(1, s.m_i));
  m_y.alloc(B * std::max(1, s.m_e));
  m_z.alloc(B * std::max(1, s.m_i));
  m_mu.alloc(B);
  m_lhs.alloc(B * k.lhs.nnz());
  m_rhs.alloc(B * k.dim);
  m_p.alloc(B * k.dim);
  m_ps.alloc(B * std::max(1, s.m_i));
  m_pz.alloc(B * std::max(1, s.m_i));
  m_D.alloc(B * l.n);
  m_Lx.alloc(B * std::max<int64_t>(1, l.nnzL));
  m_contrib.alloc(B * std::max<uint32_t>(1, l.n_contrib));
  {
    // slot hand-over (ldlt_kernels.h: slot_take) needs every slot to have exactly one reader
    std::vector<uint8_t> readers(std::max<uint32_t>(1, l.n_contrib), 0);
    bool single_reader = true;
    for (const LdltTask& t : l.tasks)  // (each task's slice is padded to a multiple of four)
      for (uint32_t c = 0; c < t.n_contrib_idx; ++c)
        single_reader = single_reader && ++readers[l.contrib_idx[t.contrib_off + c]] == 1;
    m_slot_handoff = m_single_launch && single_reader;
    if (m_slot_handoff) {
      std::vector<double> empty(m_contrib.n);
      const unsigned long long bits = kSlotEmpty;
      double e;
      std::memcpy(&e, &bits, sizeof(e));
      std::fill(empty.begin(), empty.end(), e);
      SLPX_HIP_CHECK(hipMemcpy(m_contrib.p, empty.data(), empty.size() * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  if (m_dense) {
    // dim x dim per problem, the pattern of lhs for the scatter; LDS: column k and its scaled copy
    m_dense_A.alloc(B * static_cast<size_t>(l.n) * l.n);
    m_dense_colptr.upload(k.lhs.colptr);
    m_dense_rowidx.upload(k.lhs.rowidx);
    m_dense_lds = 16u * static_cast<uint32_t>(l.n) + 320u;
    m_dense_pivoted = l.dense_pivoted;
    if (m_dense_pivoted) {
      m_dense_trans.alloc(B * static_cast<size_t>(l.n));
      m_dense_trans.zero();
      SLPX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ldlt_dense_pivoted_factor_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    if (m_dense_lds > 160u * 1024u) throw std::runtime_error("slpx: the dense factorization holds two columns in LDS: at most 10 000 rows");
    SLPX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ldlt_dense_factor_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SLPX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ldlt_dense_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  m_scontrib.alloc(B * std::max<uint32_t>(1, l.n_scontrib));
  m_zv.alloc(B * l.n);
  m_xg.alloc(B * l.n);
  // Every value buffer starts at zero, not at whatever the allocator hands back: the right-hand
  // side rides in the factorization as an extra row, and compute() may come before any rhs was
  // set (RegularizedLDLT::compute(lhs), regularized_ldlt.hpp:72) — recycled memory of an earlier
  // system can hold the hand-over sentinel (a NaN pattern: the armed slots below), a NaN keeps
  // its payload through arithmetic, and an update block that IS the sentinel is never taken
  // (seen on the pair-list kernels: the fourth solver of a process spinning to its time-out).
  for (DevBuf<double>* buf : {&m_V, &m_s, &m_y, &m_z, &m_mu, &m_lhs, &m_rhs, &m_p, &m_ps, &m_pz, &m_D, &m_Lx, &m_scontrib, &m_zv})
    buf->zero();
  SLPX_HIP_CHECK(hipDeviceSynchronize());  // (the memsets ran on the null stream, the kernels will not)
  // the backward solve's hand-over through the data (ldlt_kernels.h: slot_read): two buffers of
  // x, both armed
  m_xg_by_data = m_single_launch;
  if (m_xg_by_data) {
    const std::vector<double> armed(static_cast<size_t>(B) * l.n, std::bit_cast<double>(kSlotEmpty));
    m_xg.upload(armed);
    m_xg2.upload(armed);
  }
  {
    // two inertia-counter buffers (see factor()), both cleared once here
    std::vector<LdltStats> zero(2 * static_cast<size_t>(B), LdltStats{0, 0, 0, 0, 0x7ff0000000000000ull});
    m_stats.upload(zero);
  }
  SLPX_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&m_h_reg), 2 * static_cast<size_t>(std::max<int>(B, 6)) * sizeof(double)));  // (B = 1: a twin attempt's second pair)
  if (B > 8) m_reg_dev.alloc(2 * B);
  if (m_il) {
    // batch-interleaved LDLT (ldlt_il_kernels.h): [chunk of 64 problems][index][lane]
    const size_t C = (B + 63) / 64, W = 64;
    m_lhs_il.alloc(C * k.lhs.nnz() * W);
    m_rhs_il.alloc(C * l.n * W);
    m_Lx_il.alloc(C * std::max<int64_t>(1, l.nnzL) * W);
    m_D_il.alloc(C * l.n * W);
    m_contrib_il.alloc(C * std::max<uint32_t>(1, l.n_contrib) * W);
    m_scontrib_il.alloc(C * std::max<uint32_t>(1, l.n_scontrib) * W);
    m_zv_il.alloc(C * l.n * W);
    m_xg_il.alloc(C * l.n * W);
    m_stats_part.alloc(l.tasks.size() * B);
    for (auto* buf : {&m_Lx_il, &m_D_il, &m_zv_il, &m_xg_il, &m_contrib_il, &m_scontrib_il}) buf->zero();
    // per task: the plan slices the factor kernel keeps in LDS, packed back to back
    // [n_cref, n_words | pairs (2 words each) | pair ptr | src | out |
    //  flags + column | ext dst | level ptr | per-slot update-block refs]
    uint32_t fbytes = 0, col = 0, solve_bytes = 0;
    std::vector<uint32_t> meta, meta_off;
    for (const LdltTask& t : l.tasks) {
      const uint32_t n_cref = l.ent_contrib_ptr[t.contrib_ptr_off + t.n_ent];
      meta_off.push_back(static_cast<uint32_t>(meta.size()));
      const size_t head = meta.size();
      meta.push_back(n_cref);
      meta.push_back(0);
      for (uint32_t q = 0; q < t.n_pairs; ++q) {
        const LdltPair& pr = l.pairs[t.pair_off + q];
        meta.push_back(static_cast<uint32_t>(pr.a) | (static_cast<uint32_t>(pr.b) << 16));
        meta.push_back(pr.k);
      }
      for (uint32_t i = 0; i < t.n_ent + t.n_ext + 1; ++i) meta.push_back(l.ent_pair_ptr[t.pair_ptr_off + i]);
      for (uint32_t i = 0; i < t.n_ent; ++i) meta.push_back(static_cast<uint32_t>(l.ent_src[t.ent_off + i]));
      for (uint32_t i = 0; i < t.n_ent; ++i) meta.push_back(l.ent_out[t.ent_off + i]);
      for (uint32_t i = 0; i < t.n_ent; ++i)
        meta.push_back(static_cast<uint32_t>(l.ent_flags[t.ent_off + i]) | (static_cast<uint32_t>(l.ent_col[t.ent_off + i]) << 8));
      for (uint32_t i = 0; i < t.n_ext; ++i) meta.push_back(l.ext_dst[t.ext_off + i]);
      for (uint32_t i = 0; i < t.n_lvl + 1; ++i) meta.push_back(l.lvl_ptr[t.lvl_off + i]);
      // update-block refs, flat per entry slot (entry e belongs to slot e % kIlSlots), entry order and
      // block order within an entry as in the plan
      {
        const size_t ptr_at = meta.size();
        meta.insert(meta.end(), kIlSlots + 1, 0u);  // filled below: first ref of each slot, total
        uint32_t n_refs = 0;
        for (uint32_t sidx = 0; sidx < kIlSlots; ++sidx) {
          meta[ptr_at + sidx] = n_refs;
          for (uint32_t e = sidx; e < t.n_ent; e += kIlSlots)
            for (uint32_t cix = l.ent_contrib_ptr[t.contrib_ptr_off + e]; cix < l.ent_contrib_ptr[t.contrib_ptr_off + e + 1]; ++cix) {
              meta.push_back(e);
              meta.push_back(l.contrib_idx[t.contrib_off + cix]);
              ++n_refs;
            }
        }
        meta[ptr_at + kIlSlots] = n_refs;
      }
      const uint32_t n_words = static_cast<uint32_t>(meta.size() - head - 2);
      meta[head + 1] = n_words;
      fbytes = std::max(fbytes, std::max<uint32_t>((t.n_ent + t.n_col) * kIlW * 8u, 3u * kIlSlots * kIlW * 8u) + 4u * n_words + 16u);
      col = std::max(col, t.n_col + 1);
      solve_bytes = std::max(solve_bytes, t.n_col * 64u * 8u + 4u * (3u * t.n_col + t.n_lvl + 4u) + 8u * t.n_bwd_items + 16u);
    }
    if (meta.empty()) meta.push_back(0);
    if (meta_off.empty()) meta_off.push_back(0);
    m_il_meta.upload(meta);
    m_il_meta_off.upload(meta_off);
    m_il_factor_lds = fbytes;
    m_il_solve_lds = std::max(col * 64u * 8u, solve_bytes);  // fwd: y rows; bwd: x rows + its plan slices
    if (m_il_factor_lds > 160u * 1024u)
      throw std::runtime_error("slpx: an LDLT task does not fit the interleaved kernel's LDS (LdltOptions::task_entries)");
    SLPX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ldlt_factor_il_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SLPX_HIP_CHECK(hipDeviceSynchronize());
  }
  SLPX_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&m_h_stats), static_cast<size_t>(std::max<int>(B, 2)) * sizeof(LdltStats)));
  {
    unsigned long long* seq = nullptr;
    SLPX_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&seq), sizeof(unsigned long long)));
    *seq = 0;
    m_h_seq = seq;
    m_seq_dev.alloc(1);
    m_seq_dev.zero();
  }
  const uint64_t scratch = std::max(s.full.global_scratch_doubles, s.values.global_scratch_doubles);
  m_scratch.alloc(B * std::max<uint64_t>(1, scratch));
  DeviceArena::current() = nullptr;
  m_arena.commit();
  lap("  upload: value buffers");
  set_scaling(std::vector<double>(s.n_scales(), 1.0));
  lap("  upload: scaling, static V");
}

DeviceNlp::~DeviceNlp() {
  if (m_h_reg) (void)hipHostFree(m_h_reg);
  if (m_h_stats) (void)hipHostFree(m_h_stats);
  if (m_h_seq) (void)hipHostFree(const_cast<unsigned long long*>(m_h_seq));
  if (m_ipm_host) (void)hipHostFree(m_ipm_host);
  if (m_ipm_ctl_host) (void)hipHostFree(m_ipm_ctl_host);
  if (m_ipm_ctl_dev) (void)hipFree(m_ipm_ctl_dev);
  if (m_tape_stream) {
    (void)hipStreamSynchronize(m_tape_stream);
    (void)hipStreamDestroy(m_tape_stream);
  }
  if (m_chain_ev) (void)hipEventDestroy(m_chain_ev);
  if (m_stream.ev) (void)hipEventDestroy(m_stream.ev);
}

void DeviceNlp::set_scaling(const std::vector<double>& scales) {
  const NlpStructure& s = m_s_ref;
  if (static_cast<int>(scales.size()) != s.n_scales())
    throw std::runtime_error("set_scaling: wrong length");
  m_scales.upload(scales);
  std::vector<double> in_scale(s.n_inputs(), 1.0);
  for (int j = 0; j < s.m_e; ++j) in_scale[s.n + j] = scales[1 + j];
  for (int j = 0; j < s.m_i; ++j) in_scale[s.n + s.m_e + j] = scales[1 + s.m_e + j];
  m_in_scale.upload(in_scale);
  m_V_static.assign(s.nV, 0.0);
  for (int k = 0; k < s.nV; ++k) {
    const int32_t sc = s.V_scale_idx[k];
    m_V_static[k] = sc >= 0 ? scales[sc] * s.V_static_raw[k] : s.V_static_raw[k];
  }
  std::vector<double> all(static_cast<size_t>(m_batch) * s.nV);
  for (int b = 0; b < m_batch; ++b)
    std::copy(m_V_static.begin(), m_V_static.end(), all.begin() + static_cast<size_t>(b) * s.nV);
  SLPX_HIP_CHECK(hipMemcpy(m_V.p, all.data(), all.size() * sizeof(double), hipMemcpyHostToDevice));
  if (m_ipm)
    SLPX_HIP_CHECK(hipMemcpy(m_V_trial.p, m_V_static.data(), m_V_static.size() * sizeof(double), hipMemcpyHostToDevice));
}

void DeviceNlp::upload_x(const double* x) {
  const NlpStructure& s = m_s_ref;
  SLPX_HIP_CHECK(hipMemcpy2DAsync(m_in.p, s.n_inputs() * sizeof(double), x, s.n * sizeof(double),
                                  s.n * sizeof(double), m_batch, hipMemcpyHostToDevice, m_stream));
}

void DeviceNlp::upload_duals(const double* sv, const double* y, const double* z) {
  const NlpStructure& s = m_s_ref;
  const size_t B = m_batch;
  if (s.m_i) {
    SLPX_HIP_CHECK(hipMemcpyAsync(m_s.p, sv, B * s.m_i * sizeof(double), hipMemcpyHostToDevice, m_stream));
    SLPX_HIP_CHECK(hipMemcpyAsync(m_z.p, z, B * s.m_i * sizeof(double), hipMemcpyHostToDevice, m_stream));
    SLPX_HIP_CHECK(hipMemcpy2DAsync(m_in.p + s.n + s.m_e, s.n_inputs() * sizeof(double), z,
                                    s.m_i * sizeof(double), s.m_i * sizeof(double), m_batch,
                                    hipMemcpyHostToDevice, m_stream));
  }
  if (s.m_e) {
    SLPX_HIP_CHECK(hipMemcpyAsync(m_y.p, y, B * s.m_e * sizeof(double), hipMemcpyHostToDevice, m_stream));
    SLPX_HIP_CHECK(hipMemcpy2DAsync(m_in.p + s.n, s.n_inputs() * sizeof(double), y,
                                    s.m_e * sizeof(double), s.m_e * sizeof(double), m_batch,
                                    hipMemcpyHostToDevice, m_stream));
  }
}

void DeviceNlp::upload_mu(const double* mu) {
  SLPX_HIP_CHECK(hipMemcpyAsync(m_mu.p, mu, m_batch * sizeof(double), hipMemcpyHostToDevice, m_stream));
}

void DeviceNlp::download_V(double* V) { download(m_V.p, V, static_cast<size_t>(m_batch) * m_s_ref.nV); }

void DeviceNlp::download(const double* dev, double* host, size_t count) {
  SLPX_HIP_CHECK(hipMemcpyAsync(host, dev, count * sizeof(double), hipMemcpyDeviceToHost, m_stream));
  SLPX_HIP_CHECK(hipStreamSynchronize(m_stream));
}

void DeviceNlp::launch_tape(const TapeDevice& t, bool reverse) {
  launch_tape(t, reverse, m_stream, m_stream);
}

// The task classes are independent; `other`: the stream of the few large / global tasks.
void DeviceNlp::launch_tape(const TapeDevice& t, bool reverse, hipStream_t small_stream,
                            hipStream_t other) {
  const TapeDev view = t.view();
  const int in_stride = m_s_ref.n_inputs(), v_stride = m_s_ref.nV;
  const unsigned long long sstride = m_scratch.n / m_batch;
  const double* in_p = m_in_override ? m_in_override : m_in.p;
  double* V_p = m_V_override ? m_V_override : m_V.p;
  // basic_ops: the program only uses + - * / sin cos sqrt and the piecewise ops, so
  // the kernel specialization without pow/exp/log/erf/... (fewer VGPRs, less code)
  // generated template kernels: one lane per task instance (tape_jit.hpp)
  // interpreted 64-thread tasks ride along in the generated kernel's launch when there is one
  const bool small_rides = t.n_bodies > 0 && t.n_small > 0;
  if (t.n_bodies) {
    const int mode = reverse ? 1 : 0;
    const unsigned* table = t.tmpl_table[mode].p;
    int n_bodies = static_cast<int>(t.n_bodies);
    const unsigned* inst = t.tmpl_inst.p;
    const unsigned* leaf_src = view.leaf_src;
    const double* consts = view.consts;
    const double* in = in_p;
    int in_stride_arg = in_stride;
    const double* in_scale = m_in_scale.p;
    const double* scales = m_scales.p;
    double* V = V_p;
    int v_stride_arg = v_stride;
    const unsigned* vout_dst = view.vout_dst;
    const int* vout_scale = view.vout_scale;
    const unsigned* jout_dst = view.jout_dst;
    const int* jout_scale = view.jout_scale;
    TapeDev view_arg = view;
    const unsigned* task_list = t.small_list.p;
    int n_template_blocks = static_cast<int>(t.tmpl_blocks[mode]);
    int do_reverse = reverse ? 1 : 0;
    const uint64_t* params_dev = t.tmpl_params_dev.p;
    unsigned int* chain = m_chain_args.chain;
    unsigned int wait_step = m_chain_args.wait_step, this_step = m_chain_args.this_step;
    const unsigned grid = t.tmpl_blocks[mode] + (small_rides ? t.n_small : 0u);
    unsigned int n_workgroups = m_chain_args.skip_flag ? 0u : grid * static_cast<unsigned>(m_batch);
    m_last_tape_workgroups = grid * static_cast<unsigned>(m_batch);
    void* args[] = {&table,  &n_bodies, &inst,         &leaf_src, &consts,     &in,       &in_stride_arg,
                    &in_scale, &scales, &V,            &v_stride_arg, &vout_dst, &vout_scale, &jout_dst,
                    &jout_scale, &view_arg, &task_list, &n_template_blocks, &do_reverse,
                    &params_dev, &chain, &wait_step, &this_step, &n_workgroups};
    SLPX_HIP_CHECK(hipModuleLaunchKernel(t.tmpl_fn, grid, m_batch, 1, t.tmpl_threads, 1, 1, small_rides ? t.small_lds : 0u,
                                         small_stream, args, nullptr));
  }
  auto small_fn = t.basic_ops ? tape_sweep_lds_kernel<64, false> : tape_sweep_lds_kernel<64, true>;
  auto large_fn = t.basic_ops ? tape_sweep_lds_kernel<256, false> : tape_sweep_lds_kernel<256, true>;
  if (t.n_small && !small_rides)
    hipLaunchKernelGGL(small_fn, dim3(t.n_small, m_batch), dim3(64), t.small_lds, small_stream, view,
                       t.small_list.p, in_p, in_stride, m_in_scale.p, m_scales.p, V_p, v_stride,
                       reverse ? 1 : 0);
  if (t.n_large)
    hipLaunchKernelGGL(large_fn, dim3(t.n_large, m_batch), dim3(256), t.large_lds, other, view,
                       t.large_list.p, in_p, in_stride, m_in_scale.p, m_scales.p, V_p, v_stride,
                       reverse ? 1 : 0);
  if (t.n_global)
    hipLaunchKernelGGL(tape_sweep_global_kernel, dim3(t.n_global, m_batch), dim3(1024), 0, other,
                       view, t.global_list.p, in_p, in_stride, m_in_scale.p, m_scales.p, V_p,
                       v_stride, m_scratch.p, sstride, reverse ? 1 : 0);
  if (m_reduces.n && m_tape_reduce)
    hipLaunchKernelGGL(tape_reduce_kernel, dim3(static_cast<uint32_t>(m_reduces.n), m_batch), dim3(64), 0,
                       small_stream, m_reduces.p, m_scales.p, V_p, v_stride);
  SLPX_HIP_CHECK(hipGetLastError());
}

void DeviceNlp::sweep_full(bool with_reduce) {
  m_tape_reduce = with_reduce;
  launch_tape(m_full, true);
  m_tape_reduce = true;
}
void DeviceNlp::sweep_full_for_step() {
  const TapeDevice& t = m_full;
  // one generated kernel is the whole sweep (nothing interpreted beside it), the step is the one-launch
  // multifrontal kernel
  // (and its workgroups are single waves: g-fold's sweep, 256-thread workgroups interpreting its packs of
  // rows, shares the chip badly with the step kernel — 13.3 k steps/s chained against 13.6 k)
  const bool one_kernel = t.n_bodies > 0 && t.n_large == 0 && t.n_global == 0 && (t.tmpl_threads == 64 || t.tmpl_wide_for_chain);
  if (!m_chain_on || !one_kernel || !m_mf || m_batch != 1 || xg_other() == nullptr || !m_fuse_solve) {
    sweep_full(/*with_reduce=*/false);
    return;
  }
  if (m_tape_stream == nullptr) {
    SLPX_HIP_CHECK(hipStreamCreateWithFlags(&m_tape_stream, hipStreamNonBlocking));
    SLPX_HIP_CHECK(hipEventCreateWithFlags(&m_chain_ev, hipEventDisableTiming));
    SLPX_HIP_CHECK(hipEventCreateWithFlags(&m_stream.ev, hipEventDisableTiming));
    m_stream.tape = m_tape_stream;
    {
      // (tests: the running totals start so far below their bound, so that a few thousand chained steps cross it —
      // SLPX_DEBUG_CHAIN_TOTALS_HEADROOM workgroups of headroom — and the restart of the counts is exercised)
      std::vector<unsigned int> words(128, 0u);
      if (const char* env = std::getenv("SLPX_DEBUG_CHAIN_TOTALS_HEADROOM")) {
        const unsigned int headroom = static_cast<unsigned int>(std::max(1L, std::atol(env)));
        m_chain_sweep_wgs = m_chain_step_wgs = (1u << 29) - std::min(headroom, 1u << 28);
        words[16] = m_chain_sweep_wgs;
        words[48] = m_chain_step_wgs;
      }
      m_chain.upload(words);
    }
    // (words 96, 97: the concurrency probe, see chain_probe_wait_kernel)
    hipLaunchKernelGGL(chain_probe_wait_kernel, dim3(1), dim3(1), 0, m_tape_stream, m_chain.p + 96);
    hipLaunchKernelGGL(chain_probe_set_kernel, dim3(1), dim3(1), 0, m_stream.raw(), m_chain.p + 96);
    SLPX_HIP_CHECK(hipStreamSynchronize(m_tape_stream));
    SLPX_HIP_CHECK(hipStreamSynchronize(m_stream.raw()));
    unsigned int verdict = 0;
    SLPX_HIP_CHECK(hipMemcpy(&verdict, m_chain.p + 97, sizeof(verdict), hipMemcpyDeviceToHost));
    if (verdict != 1u) {
      m_chain_on = false;  // kernels run one at a time here: the sweep stays in the main stream
      if (std::getenv("SLPX_LDLT_VERBOSE")) std::fprintf(stderr, "slpx: kernels of two streams do not run at once here: steps are not chained\n");
      sweep_full(/*with_reduce=*/false);
      return;
    }
  }
  m_tape_reduce = false;
  if (m_stream.tape_pending) {
    // The last chained sweep was never consumed (its step threw before the step kernel was launched, or
    // the caller swept twice): no step kernel will signal the word a chained sweep would wait for.  Order
    // the streams by an event and start over.
    (void)static_cast<hipStream_t>(m_stream);  // waits for that sweep, marks the stream touched
    m_last_step_chained = false;
  }
  if (m_chain_seq >= (1u << 30) - 1u || m_chain_sweep_wgs >= (1u << 29) || m_chain_step_wgs >= (1u << 29)) {
    // step numbers stay in [1, 2^30) (the kernels compare them as signed differences, use 0 for "no wait"
    // and bit 31 for failure): every ~14 hours of chained steps, one unchained step and a fresh count
    SLPX_HIP_CHECK(hipStreamSynchronize(m_tape_stream));
    SLPX_HIP_CHECK(hipStreamSynchronize(m_stream.raw()));
    SLPX_HIP_CHECK(hipMemset(m_chain.p, 0, 64 * sizeof(unsigned int)));
    m_chain_seq = 0;
    m_chain_sweep_wgs = m_chain_step_wgs = 0;
    m_last_step_chained = false;
    m_stream.touched = true;
  }
  if (m_stream.touched) {
    // Something else went to the main stream since the last step (an upload of the state, a download of
    // the result, another kernel): this step's sweep goes there too, behind all of it — the pattern of a
    // caller that looks at every step costs nothing extra.  Only steps that FOLLOW a step directly chain.
    m_stream.touched = false;
    launch_tape(m_full, true, m_stream.raw(), m_stream.raw());
    m_tape_reduce = true;
    return;
  }
  // the step kernel before this sweep: a chained one tells the sweep itself when its last workgroup is
  // through; any other (the first step of a run, a re-attempt of the policy loop) through an event, once
  unsigned int wait_step = m_chain_step_wgs;  // (every workgroup of the chained step kernels so far)
  if (!m_last_step_chained) {
    SLPX_HIP_CHECK(hipEventRecord(m_chain_ev, m_stream.raw()));
    SLPX_HIP_CHECK(hipStreamWaitEvent(m_tape_stream, m_chain_ev, 0));
    wait_step = 0;
  }
  if (m_debug_break_chain && wait_step != 0u) {
    m_debug_break_chain = false;
    wait_step += 1u << 20;  // (slpx_debug_chain) step kernels nobody launched
  }
  ++m_chain_seq;
  m_chain_args = ChainArgs{m_chain.p, wait_step, m_chain_seq};
  launch_tape(m_full, true, m_tape_stream, m_tape_stream);
  m_chain_sweep_wgs += m_last_tape_workgroups;
  m_tape_reduce = true;
  m_chain_args = ChainArgs{};
  m_stream.tape_pending = true;  // until the step kernel that waits for this sweep is launched
}
// A chained step reported kLdltChainFailure: one of the two kernels gave up waiting for the other (never
// expected — a shared / preempted / serialized GPU).  Drain both streams, clear the words, keep this
// system's steps unchained from now on and leave V as a fresh sweep of the current state writes it.
void DeviceNlp::recover_from_chain_failure() {
  if (m_tape_stream != nullptr) SLPX_HIP_CHECK(hipStreamSynchronize(m_tape_stream));
  SLPX_HIP_CHECK(hipStreamSynchronize(m_stream.raw()));
  if (m_chain.p != nullptr) SLPX_HIP_CHECK(hipMemset(m_chain.p, 0, 64 * sizeof(unsigned int)));
  m_chain_on = false;
  m_chain_sweep_wgs = m_chain_step_wgs = 0;
  m_last_step_chained = false;
  m_stream.tape_pending = false;
  m_stream.touched = true;
  ++m_chain_failures;
  if (std::getenv("SLPX_LDLT_VERBOSE")) std::fprintf(stderr, "slpx: a chained step lost its hand-over: the step is redone, steps are unchained from here on\n");
  sweep_full(/*with_reduce=*/false);
}
void DeviceNlp::sweep_values() { launch_tape(m_values, false); }
void DeviceNlp::sweep_values_trial() {
  m_in_override = m_trial_in.p;
  m_V_override = m_V_trial.p;
  m_tape_reduce = false;  // ipm_trial_metrics_kernel, the only reader of these values, finishes the sums
  launch_tape(m_values, false);
  m_tape_reduce = true;
  m_in_override = nullptr;
  m_V_override = nullptr;
}

static inline int grid_for(int work, int block, int cap = 2048) {
  return std::max(1, std::min((work + block - 1) / block, cap));
}

// The static side of KktFuse: for every entry of the factorization plan what it is made of.
void DeviceNlp::build_inline_kkt(const NlpStructure& s, const KktPlan& k, const LdltPlan& l) {
  std::vector<int32_t> vsrc(l.ent_src.size(), -1);
  std::vector<KktTerm> terms;
  std::vector<uint2> task_terms(l.tasks.size(), uint2{0, 0});
  auto term = [](int kind, int a, int row, int c) { return KktTerm{a, (kind << 28) | row, c}; };
  bool ok = s.nV < (1 << 30) && k.m_i < (1 << 28) && k.m_e < (1 << 28);
  uint32_t widest = 0;
  for (size_t ti = 0; ti < l.tasks.size() && ok; ++ti) {
    const LdltTask& t = l.tasks[ti];
    const size_t block = terms.size();
    for (uint32_t i = 0; i < t.n_ent; ++i) {
      const size_t e = t.ent_off + i;
      const int32_t s0 = l.ent_src[e];
      if (s0 < 0) continue;
      const size_t first = terms.size();
      if (l.ent_flags[e] & 4) {  // rhs entry s0 (kkt_kernels.h: kkt_rhs_entry)
        const int j = s0;
        if (j >= k.n) {
          vsrc[e] = (s.off_ce + j - k.n) | (1 << 30);
          continue;
        }
        if (k.g_src[j] >= 0) terms.push_back(term(1, k.g_src[j], 0, 0));
        for (int p = s.Ae.colptr[j]; p < s.Ae.colptr[j + 1]; ++p)
          terms.push_back(term(2, s.off_Ae + p, s.Ae.rowidx[p], 0));
        for (int p = s.Ai.colptr[j]; p < s.Ai.colptr[j + 1]; ++p)
          terms.push_back(term(3, s.off_Ai + p, s.Ai.rowidx[p], s.off_ci + s.Ai.rowidx[p]));
      } else {  // lhs entry s0 (kkt_kernels.h: kkt_assemble_body)
        const int f = k.fast_src[s0];
        if (f >= 0) {
          vsrc[e] = f;
          continue;
        }
        if (f != -2) continue;
        for (int q = k.dptr[s0]; q < k.dptr[s0 + 1]; ++q) terms.push_back(term(0, k.dsrc[q], 0, 0));
        for (int q = k.pptr[s0]; q < k.pptr[s0 + 1]; ++q) terms.push_back(term(4, k.pa[q], k.pr[q], k.pb[q]));
      }
      const size_t rel = first - block, cnt = terms.size() - first;
      if (rel >= (1u << 20) || cnt >= (1u << 11)) {
        ok = false;
        break;
      }
      // (a sum of nothing — an x row with no gradient, no A_e, no A_i entry — is a zero)
      vsrc[e] = cnt == 0 ? -1 : -static_cast<int32_t>(2 + (rel | (cnt << 20)));
    }
    while ((terms.size() - block) % 4 != 0) terms.push_back(KktTerm{0, 0, 0});  // 4 terms = 3 x 16 bytes
    const uint32_t off16 = static_cast<uint32_t>(block * sizeof(KktTerm) / 16);
    const uint32_t len16 = static_cast<uint32_t>((terms.size() - block) * sizeof(KktTerm) / 16);
    task_terms[ti] = uint2{off16, len16};
    widest = std::max(widest, len16);
  }
  // the staged terms and, behind them, one product per term (4 terms = 3 x 16 bytes)
  m_factor_lds_inline = l.factor_lds_bytes + 16u + 16u * widest + 8u * (widest * 4u / 3u);
  if (!ok || m_factor_lds_inline > 160u * 1024u) {
    m_fuse_kkt = false;  // (the stand-alone assembly kernels then)
    return;
  }
  if (terms.empty()) terms.push_back(KktTerm{0, 0, 0});
  m_ent_vsrc.upload(vsrc);
  m_kkt_terms.upload(terms);
  m_task_terms.upload(task_terms);
  m_h_vsrc = std::move(vsrc);
  m_h_terms = std::move(terms);
  m_h_task_terms = std::move(task_terms);
}

// The static side of BacksubFuse: every row of A_i goes to the task that owns the deepest of
// its columns.
void DeviceNlp::build_inline_backsub(const KktPlan& k, const LdltPlan& l) {
  static_assert(sizeof(BsRow) == 8 && sizeof(BsTerm) == 8, "rows and terms are staged as 16-byte pairs");
  const int dim = l.n;
  std::vector<int32_t> inv_perm(dim, -1), task_of(dim, -1), local_of(dim, -1);
  for (int pj = 0; pj < dim; ++pj) inv_perm[l.perm[pj]] = pj;
  for (size_t ti = 0; ti < l.tasks.size(); ++ti)
    for (uint32_t i = 0; i < l.tasks[ti].n_col; ++i) {
      const uint32_t pj = l.col_perm[l.tasks[ti].col_off + i];
      task_of[pj] = static_cast<int32_t>(ti);
      local_of[pj] = static_cast<int32_t>(i);
    }
  std::vector<std::vector<int>> rows_of(l.tasks.size());
  bool ok = true;
  for (int r = 0; r < k.m_i && ok; ++r) {
    int deepest = -1;
    for (int q = k.ai_rowptr[r]; q < k.ai_rowptr[r + 1]; ++q) {
      const int pj = inv_perm[k.ai_col[q]];
      if (deepest < 0 || pj < deepest) deepest = pj;
    }
    if (deepest < 0) deepest = 0;  // an empty row: anybody
    const int owner = task_of[deepest];
    // what the scheme rests on: every other column is the owner's or an ancestor's (a later round's)
    for (int q = k.ai_rowptr[r]; q < k.ai_rowptr[r + 1]; ++q) {
      const int other = task_of[inv_perm[k.ai_col[q]]];
      if (other != owner && l.tasks[other].round <= l.tasks[owner].round) ok = false;
    }
    rows_of[owner].push_back(r);
  }
  std::vector<BsRow> plan;  // BsRow and BsTerm are both two 32-bit words: one array
  std::vector<uint4> task_plan(l.tasks.size(), uint4{0, 0, 0, 0});
  uint32_t widest = 0;
  for (size_t ti = 0; ti < l.tasks.size() && ok; ++ti) {
    const size_t block = plan.size();
    const std::vector<int>& rows = rows_of[ti];
    const size_t n_rows = rows.size(), rows_padded = (n_rows + 1) / 2 * 2;
    plan.resize(block + rows_padded, BsRow{0, 0});
    uint32_t term = 0;
    for (size_t j = 0; j < n_rows; ++j) {
      const int r = rows[j];
      const uint32_t cnt = static_cast<uint32_t>(k.ai_rowptr[r + 1] - k.ai_rowptr[r]);
      if (term >= (1u << 20) || cnt >= (1u << 12)) ok = false;
      plan[block + j] = BsRow{r, term | (cnt << 20)};
      for (int q = k.ai_rowptr[r]; q < k.ai_rowptr[r + 1]; ++q) {
        const int pj = inv_perm[k.ai_col[q]];
        const uint32_t ref = task_of[pj] == static_cast<int32_t>(ti) ? static_cast<uint32_t>(local_of[pj])
                                                                     : (0x80000000u | static_cast<uint32_t>(pj));
        plan.push_back(BsRow{k.ai_src[q], ref});  // (a BsTerm)
        ++term;
      }
    }
    if ((plan.size() - block) % 2 != 0) plan.push_back(BsRow{0, 0});
    const uint32_t len16 = static_cast<uint32_t>((plan.size() - block) / 2);
    task_plan[ti] = uint4{static_cast<uint32_t>(block / 2), len16, static_cast<uint32_t>(n_rows),
                          static_cast<uint32_t>(rows_padded / 2)};
    widest = std::max(widest, len16);
  }
  m_solve_lds_inline = l.solve_lds_bytes + 16u + 16u * widest;
  if (!ok || m_solve_lds_inline > 160u * 1024u) {
    m_fuse_backsub = false;  // (the stand-alone back-substitution kernel then)
    return;
  }
  if (plan.empty()) plan.push_back(BsRow{0, 0});
  m_bs_plan.upload(plan);
  m_bs_task_plan.upload(task_plan);
  m_h_bs_plan = std::move(plan);
  m_h_bs_task_plan = std::move(task_plan);
}

constexpr int kFactorThreadsSingle = 1024;
// The multifrontal step (ldlt_mf_kernels.h): plan upload, LDS footprint, co-residency of every task's workgroup.
void DeviceNlp::build_mf(const LdltPlan& l) {
  if (const char* env = std::getenv("SLPX_LDLT_MF"))
    if (env[0] == '0') return;
  if (m_h_task_terms.size() != l.tasks.size() || m_h_bs_task_plan.size() != l.tasks.size()) return;
  // Every task's static LDS content as ONE image in LDS order (mf_carve: from the tables to the KKT
  // terms, then — behind the terms' products, which are not staged — its back-substitution rows):
  // thirteen arrays staged one after the other were thirteen dependent waits on memory (4.5 us);
  // one copy loop with every load in flight is a single trip.
  uint32_t lds = 0;
  std::vector<uint4> image, desc(l.tasks.size());
  // sizes first (the slots are of one size: the largest image), then every task's image written straight into its
  // slot, the tasks in chunks on the setup threads
  std::vector<uint32_t> blob_bytes(l.tasks.size());
  size_t stride16 = 1;
  for (size_t ti = 0; ti < l.tasks.size(); ++ti) {
    const MfCarve cv = mf_carve(l.tasks[ti], l.mf_tasks[ti]);
    const uint32_t terms16 = m_h_task_terms[ti].y, bs16 = m_h_bs_task_plan[ti].y;
    const uint32_t n_terms = terms16 * 4u / 3u;
    const uint32_t end_terms = cv.o_terms + 16u * terms16;
    lds = std::max(lds, mf_align16(end_terms + 8u * n_terms) + 16u * bs16);
    blob_bytes[ti] = end_terms - cv.o_tab + 16u * bs16;
    stride16 = std::max<size_t>(stride16, blob_bytes[ti] / 16u);
    desc[ti] = uint4{0u, (end_terms - cv.o_tab) / 16u, bs16, terms16};
  }
  if (stride16 > kMfImageGroups) return;  // (a task image beyond what the staging loop requests)
  // fixed-size slots (the kernel requests a task's image before it knows its length) + one round of padding
  image.assign(stride16 * l.tasks.size() + kMfImageGroups, uint4{0, 0, 0, 0});
  std::atomic<bool> out_of_bounds{false}, terms_do_not_fit{false};
  parallel_chunks(l.tasks.size(), 8, [&](size_t t_begin, size_t t_end, unsigned) {
    std::vector<uint8_t> fl;
    std::vector<int32_t> vs;
    for (size_t ti = t_begin; ti < t_end; ++ti) {
      const LdltTask& t = l.tasks[ti];
      const LdltMfTask& m = l.mf_tasks[ti];
      const MfCarve cv = mf_carve(t, m);
      const uint32_t terms16 = m_h_task_terms[ti].y, bs16 = m_h_bs_task_plan[ti].y;
      const uint32_t end_terms = cv.o_terms + 16u * terms16;
      unsigned char* blob = reinterpret_cast<unsigned char*>(image.data() + ti * stride16);
      const size_t blob_size = blob_bytes[ti];
      auto put = [&](uint32_t at, const void* src, size_t bytes) {
        if (at < cv.o_tab || at - cv.o_tab + bytes > blob_size) {
          out_of_bounds = true;
          return;
        }
        if (bytes) std::memcpy(blob + (at - cv.o_tab), src, bytes);
      };
      put(cv.o_tab, l.mf_tab.data() + m.tab_off, 2u * m.n_tab);
      put(cv.o_lvl, l.mf_lvl_ptr.data() + t.lvl_off, 4u * (t.n_lvl + 1));
      put(cv.o_ext, l.mf_ext.data() + m.ext_off, 4u * m.n_ext);
      {
        // where an entry's value comes from; an entry that is a sum of terms says how many of each kind
        // (kkt_terms_sum_grouped: the terms lie kind by kind)
        vs.assign(m_h_vsrc.begin() + t.ent_off, m_h_vsrc.begin() + t.ent_off + t.n_ent);
        const KktTerm* tt = m_h_terms.data() + static_cast<size_t>(m_h_task_terms[ti].x) * 4u / 3u;
        for (uint32_t i = 0; i < t.n_ent; ++i) {
          if (vs[i] >= -1) continue;
          const uint32_t code = static_cast<uint32_t>(-(vs[i] + 2));
          const uint32_t first = code & 0xfffffu, cnt = code >> 20;
          uint32_t has_g = 0, n_a = 0, n_b = 0;
          bool sorted = true;
          int last = -1;
          for (uint32_t q = first; q < first + cnt; ++q) {
            const int kind = tt[q].b >> 28;
            sorted = sorted && kind >= last;
            last = kind;
            if (kind == 1) ++has_g;
            else if (kind == 0 || kind == 2) ++n_a;
            else ++n_b;
          }
          if (!sorted || has_g > 1 || !mf_term_code_fits(first, n_a, n_b)) {
            terms_do_not_fit = true;
            break;
          }
          vs[i] = -static_cast<int32_t>(2u + mf_term_code(first, has_g, n_a, n_b));
        }
        put(cv.o_src, vs.data(), 4u * t.n_ent);
      }
      {
        // (bit 5: the entry takes update slots)
        fl.assign(l.ent_flags.begin() + t.ent_off, l.ent_flags.begin() + t.ent_off + t.n_ent);
        for (uint32_t j = 0; j < m.n_cent; ++j) fl[l.mf_cent[m.cent_off + j]] |= 0x20;
        put(cv.o_flags, fl.data(), t.n_ent);
      }
      put(cv.o_cent, l.mf_cent.data() + m.cent_off, 2u * m.n_cent);
      put(cv.o_cptr, l.mf_contrib_ptr.data() + m.contrib_ptr_off, 4u * (m.n_cent + 1));
      put(cv.o_cidx, l.mf_contrib_idx.data() + m.contrib_off, 4u * m.n_contrib_idx);
      put(cv.o_cp, l.col_perm.data() + t.col_off, 4u * t.n_col);
      put(cv.o_anc, l.mf_anc.data() + m.anc_off, 4u * m.n_anc);
      put(cv.o_fr, l.mf_fronts.data() + m.front_off, sizeof(LdltFront) * m.n_front);
      {
        // the inertia counters start at zero, the smallest |d| at +inf
        const unsigned long long inf = 0x7ff0000000000000ull;
        put(cv.o_cnt + 16u, &inf, 8);
      }
      put(cv.o_terms, reinterpret_cast<const unsigned char*>(m_h_terms.data()) + 16u * static_cast<size_t>(m_h_task_terms[ti].x),
          16u * terms16);
      put(end_terms, reinterpret_cast<const unsigned char*>(m_h_bs_plan.data()) + 16u * static_cast<size_t>(m_h_bs_task_plan[ti].x),
          16u * bs16);
    }
  });
  if (out_of_bounds) throw std::runtime_error("slpx: task image layout out of bounds");
  if (terms_do_not_fit) return;  // (an entry of hundreds of terms: the pair-list kernels' encoding holds 2047)
  lds = mf_align16(lds) + 16u;
  int per_cu = 0, cus = 0;
  hipFuncAttributes attr{};
  bool fits = false;
  if (lds <= 160u * 1024u) {
    SLPX_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m_device));
    auto resident = [&](auto kernel, int threads) {
      SLPX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      SLPX_HIP_CHECK(hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kernel)));
      SLPX_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds));
      // (the tables hold LDS byte addresses from 0: no static LDS in front of the dynamic block)
      return attr.sharedSizeBytes == 0 && l.tasks.size() + m_reduces.n <= static_cast<size_t>(per_cu) * cus;
    };
    m_mf_mfma = l.mf_n_mfma > 0;
    m_mf_threads = 1024;
    fits = m_mf_mfma ? resident(&ldlt_mf_step_kernel<1024, true, true>, 1024) && resident(&ldlt_mf_step_kernel<1024, true, false>, 1024)
                     : resident(&ldlt_mf_step_kernel<1024, false, true>, 1024) && resident(&ldlt_mf_step_kernel<1024, false, false>, 1024);
    if (const char* env = std::getenv("SLPX_MF_THREADS")) fits = fits && std::atoi(env) == 1024;
    if (!fits) {
      m_mf_threads = 512;
      fits = m_mf_mfma ? resident(&ldlt_mf_step_kernel<512, true, true>, 512) && resident(&ldlt_mf_step_kernel<512, true, false>, 512)
                       : resident(&ldlt_mf_step_kernel<512, false, true>, 512) && resident(&ldlt_mf_step_kernel<512, false, false>, 512);
    }
  }
  if (std::getenv("SLPX_LDLT_VERBOSE"))
    std::fprintf(stderr, "ldlt multifrontal step: %zu tasks, LDS %u bytes (static %zu), images %zu bytes, %d workgroup(s) of %d threads per CU x %d CUs%s\n",
                 l.tasks.size(), lds, attr.sharedSizeBytes, 16 * image.size(), per_cu, m_mf_threads, cus, fits ? "" : ": NOT resident at once");
  if (!fits) return;
  m_mf_tasks.upload(l.mf_tasks);
  m_mf_fronts.upload(l.mf_fronts);
  m_mf_image.upload(image);
  m_mf_image_stride16 = static_cast<uint32_t>(stride16);
  m_mf_image_desc.upload(desc);
  m_mf_contrib.upload(std::vector<double>(std::max<uint32_t>(1, l.mf_n_contrib), std::bit_cast<double>(kSlotEmpty)));
  if (m_exit_cnt.n == 0) m_exit_cnt.upload(std::vector<unsigned int>(1, 0u));
  m_mf_lds = lds;
  m_mf = true;
  // new right-hand sides through the fronts (one launch: every task resident, as the step kernel's)
  {
    const char* env = std::getenv("SLPX_MF_SOLVE");
    m_mf_solve = env == nullptr || env[0] != '0';
    if (m_mf_solve) {
      const void* fn = m_mf_threads == 1024 ? reinterpret_cast<const void*>(&ldlt_mf_solve_kernel<1024>)
                                            : reinterpret_cast<const void*>(&ldlt_mf_solve_kernel<512>);
      SLPX_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      int per_cu_solve = 0;
      if (m_mf_threads == 1024) SLPX_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_solve, &ldlt_mf_solve_kernel<1024>, 1024, lds));
      else SLPX_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_solve, &ldlt_mf_solve_kernel<512>, 512, lds));
      m_mf_solve = l.tasks.size() <= static_cast<size_t>(per_cu_solve) * cus;
    }
  }
}

// lhs / rhs of the CURRENT state into memory, if the last step did without them
void DeviceNlp::materialize_kkt() {
  if (m_kkt_pending) {  // the factorization that was to evaluate the system never came
    const bool with_reduce = m_kkt_pending == 2;
    m_kkt_pending = 0;
    build_kkt(with_reduce);
    return;
  }
  if (m_lhs_stale) assemble();
  if (m_rhs_stale) build_rhs();
}

// batch-major lhs / rhs for whoever asks (slpx_system_get, a caller that writes its own system):
// out of the interleaved arrays the assembly kernels filled; what the caller does with the pointer
// is not known, so the next factorization transposes again
void DeviceNlp::materialize_batch_major() {
  if (!m_il) return;
  const int C = (m_batch + 63) / 64;
  if (m_lhs_in_il) {
    const int nnz = m_kdev.nnz_lhs;
    hipLaunchKernelGGL(il_scatter_kernel, dim3((nnz + 63) / 64, C), dim3(256), 0, m_stream, m_lhs_il.p, nnz, m_lhs.p,
                       static_cast<long long>(nnz), m_batch);
    m_lhs_in_il = false;
  }
  if (m_rhs_in_il) {
    const int n = m_kdev.dim;
    hipLaunchKernelGGL(il_scatter_kernel, dim3((n + 63) / 64, C), dim3(256), 0, m_stream, m_rhs_il.p, n, m_rhs.p,
                       static_cast<long long>(n), m_batch);
    m_rhs_in_il = false;
  }
  SLPX_HIP_CHECK(hipGetLastError());
}

void DeviceNlp::assemble() {
  m_lhs_stale = false;
  m_lhs_in_il = false;
  if (m_il) {
    hipLaunchKernelGGL(kkt_assemble_il_kernel, dim3(grid_for(m_kdev.nnz_lhs, 64), il_groups(m_batch)), dim3(256), 0, m_stream,
                       m_kdev, m_V.p, m_s_ref.nV, m_s.p, m_z.p, m_lhs_il.p, m_batch);
    m_lhs_in_il = true;
  } else if (m_batch >= kBatchPerThread) {
    // measured at 512 x N=1000: 1 problem per thread 0.084 ms, 2: 0.083, 4: 0.079, 8: 0.090
    hipLaunchKernelGGL(kkt_assemble_batch_kernel<kBatchPerThread>,
                       dim3(grid_for(m_kdev.nnz_lhs, 256), (m_batch + kBatchPerThread - 1) / kBatchPerThread),
                       dim3(256), 0, m_stream, m_kdev, m_V.p, m_s_ref.nV, m_s.p, m_z.p, m_lhs.p, m_batch);
  } else {
    // four entries per thread (see the kernel)
    hipLaunchKernelGGL(kkt_assemble_kernel, dim3(grid_for((m_kdev.nnz_lhs + 3) / 4, 256), m_batch),
                       dim3(256), 0, m_stream, m_kdev, m_V.p, m_s_ref.nV, m_s.p, m_z.p, m_lhs.p);
  }
  SLPX_HIP_CHECK(hipGetLastError());
}

// lhs + rhs (+ the separable-sum reductions a sweep_full(false) left out) in one launch
void DeviceNlp::build_kkt(bool with_reduce) {
  if (m_batch >= kBatchPerThread) {  // throughput regime: the batch kernels, separately
    if (m_il) {
      // (interleaved lhs and rhs: one launch, kkt_build_il_kernel)
      const int na = grid_for(m_kdev.nnz_lhs, 64), nr = (m_kdev.dim + 63) / 64;
      hipLaunchKernelGGL(kkt_build_il_kernel, dim3(na + nr, il_groups(m_batch)), dim3(256), 0, m_stream, m_kdev, m_V.p,
                         m_s_ref.nV, m_s.p, m_y.p, m_z.p, m_mu.p, m_lhs_il.p, m_rhs_il.p, m_batch, na);
      m_lhs_stale = m_rhs_stale = false;
      m_lhs_in_il = m_rhs_in_il = true;
    } else {
      assemble();
      build_rhs();
    }
    if (with_reduce && m_reduces.n)
      hipLaunchKernelGGL(tape_reduce_kernel, dim3(static_cast<uint32_t>(m_reduces.n), m_batch), dim3(64), 0,
                         m_stream, m_reduces.p, m_scales.p, m_V.p, m_s_ref.nV);
    SLPX_HIP_CHECK(hipGetLastError());
    return;
  }
  if (m_defer_kkt) {  // evaluated inside the factorization's launch (enqueue_factor)
    m_kkt_pending = with_reduce ? 2 : 1;
    m_lhs_stale = m_rhs_stale = true;
    return;
  }
  m_lhs_stale = m_rhs_stale = false;
  m_lhs_in_il = m_rhs_in_il = false;
  const int na = grid_for((m_kdev.nnz_lhs + 3) / 4, 256), nr = grid_for(m_kdev.dim, 256);
  const int nred = with_reduce ? static_cast<int>(m_reduces.n) : 0;
  hipLaunchKernelGGL(kkt_build_kernel, dim3(na + nr + nred, m_batch), dim3(256), 0, m_stream, m_kdev, m_V.p,
                     m_s_ref.nV, m_s.p, m_y.p, m_z.p, m_mu.p, m_lhs.p, m_rhs.p, na, nr, m_reduces.p, m_scales.p);
  SLPX_HIP_CHECK(hipGetLastError());
}

// build_kkt() when a factorization attempt is the next thing enqueued (a Newton step): the
// assembly then rides in that launch (device.hpp: KktFuse)
void DeviceNlp::build_kkt_for_step(bool with_reduce) {
  m_defer_kkt = m_fuse_kkt;
  build_kkt(with_reduce);
  m_defer_kkt = false;
}

void DeviceNlp::assemble_lsq() {
  m_lhs_stale = false;
  m_lhs_in_il = false;
  hipLaunchKernelGGL(kkt_assemble_lsq_kernel, dim3(grid_for(m_kdev.nnz_lhs, 256), m_batch), dim3(256),
                     0, m_stream, m_kdev, m_V.p, m_s_ref.nV, m_s.p, m_lhs.p);
  hipLaunchKernelGGL(kkt_add_identity_kernel, dim3(grid_for(m_kdev.n, 256), m_batch), dim3(256), 0,
                     m_stream, m_kdev, m_diag_pos.p, m_lhs.p);
  SLPX_HIP_CHECK(hipGetLastError());
}

int DeviceNlp::debug_tmpl_clocks(unsigned long long* out, int blocks) {
  SLPX_HIP_CHECK(hipStreamSynchronize(m_stream));
  hipDeviceptr_t p = nullptr;
  size_t bytes = 0;
  if (!m_full.tmpl_mod || hipModuleGetGlobal(&p, &bytes, m_full.tmpl_mod, "slpx_tmpl_clocks") != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  const size_t want = std::min<size_t>(bytes, static_cast<size_t>(blocks) * 8 * sizeof(unsigned long long));
  SLPX_HIP_CHECK(hipMemcpy(out, p, want, hipMemcpyDeviceToHost));
  return static_cast<int>(m_full.tmpl_blocks[1]);
}

void DeviceNlp::debug_tape_clocks(unsigned long long* out16) {
  SLPX_HIP_CHECK(hipStreamSynchronize(m_stream));
  SLPX_HIP_CHECK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_tape_clocks), 16 * sizeof(unsigned long long)));
}

void DeviceNlp::debug_ldlt_clocks(unsigned int next_round, unsigned long long* out24) {
  SLPX_HIP_CHECK(hipStreamSynchronize(m_stream));
#ifdef SLPX_MF_CLOCKS
  if ((next_round >> 16) == 0xffffu) {  // (the instrumented build: the slots of task (next_round & 0xffff) in the last launch)
    SLPX_HIP_CHECK(hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_mf_clocks), 24 * sizeof(unsigned long long),
                                       24 * sizeof(unsigned long long) * (next_round & 0xffffu)));
    return;
  }
#endif
  SLPX_HIP_CHECK(hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_ldlt_clocks), 24 * sizeof(unsigned long long)));
  // (bits 8 and up: which task of the round, 0 = its first)
  const unsigned int round = next_round & 0xffu, within = next_round >> 8;
  m_ldev.clock_task = 0xffffffffu;
  if (round < static_cast<unsigned int>(m_l_ref.n_rounds) && m_l_ref.round_ptr[round] + within < m_l_ref.round_ptr[round + 1])
    m_ldev.clock_task = m_l_ref.round_ptr[round] + within;
}

void DeviceNlp::refresh_params(const Graph& g) {
  SLPX_HIP_CHECK(hipStreamSynchronize(m_stream));
  auto refresh = [&](const TapeProgram& prog, TapeDevice& dev) {
    if (prog.params.empty()) return;
    std::vector<double> c = prog.consts;
    for (const auto& [node, slot] : prog.params) c[slot] = g.val[node];
    dev.consts.upload(c);
  };
  refresh(m_s_ref.full, m_full);
  refresh(m_s_ref.values, m_values);
}

void DeviceNlp::build_rhs() {
  m_rhs_stale = false;
  m_rhs_in_il = false;
  if (m_il) {
    hipLaunchKernelGGL(kkt_rhs_il_kernel, dim3((m_kdev.dim + 63) / 64, il_groups(m_batch)), dim3(256), 0, m_stream, m_kdev,
                       m_V.p, m_s_ref.nV, m_s.p, m_y.p, m_z.p, m_mu.p, m_rhs_il.p, m_batch);
    m_rhs_in_il = true;
    SLPX_HIP_CHECK(hipGetLastError());
    return;
  }
  // (a batch variant like kkt_assemble_batch_kernel was measured SLOWER here: 0.102 ms vs
  // 0.070 ms at 512 x N=1000 — the per-column loops are short and the extra registers cost
  // occupancy)
  hipLaunchKernelGGL(kkt_rhs_kernel, dim3(grid_for(m_kdev.dim, 256), m_batch), dim3(256), 0, m_stream,
                     m_kdev, m_V.p, m_s_ref.nV, m_s.p, m_y.p, m_z.p, m_mu.p, m_rhs.p);
  SLPX_HIP_CHECK(hipGetLastError());
}

// Control traffic of one factorization attempt is kept off the critical path: (δ, γ) of
// every problem travel in ONE async copy from pinned memory (δ = NaN marks a problem the
// policy loop is done with), and the inertia counters are double-buffered — the launch of
// attempt k clears the buffer attempt k+1 will accumulate into — so no reset kernel runs.
// Control traffic of one factorization attempt is kept off the critical path: the kernels
// read (δ, γ) straight from pinned host memory (δ = NaN marks a problem the policy loop is
// done with), and the inertia counters are double-buffered — the launch of attempt k
// clears the buffer attempt k+1 will accumulate into — so no copy and no reset kernel run.
// The host only rewrites m_h_reg after read_stats() has synchronized.
void DeviceNlp::write_reg(const std::vector<double>& delta, const std::vector<double>& gamma,
                          const std::vector<uint8_t>& active) {
  for (int b = 0; b < m_batch; ++b) {
    m_h_reg[2 * b] = active[b] ? delta[b] : std::numeric_limits<double>::quiet_NaN();
    m_h_reg[2 * b + 1] = gamma[b];
  }
}


// the system evaluated inside the factorization's launch, if a build_kkt_for_step() asked for it
KktFuse DeviceNlp::kkt_fuse_for(int kkt_mode) const {
  KktFuse f;
  if (kkt_mode) {
    f.inline_kkt = 1;
    f.n_blocks = kkt_mode == 2 ? static_cast<int>(m_reduces.n) : 0;
    f.V = m_V.p;
    f.s = m_s.p;
    f.y = m_y.p;
    f.z = m_z.p;
    f.mu = m_mu.p;
    f.ent_vsrc = m_ent_vsrc.p;
    f.terms = reinterpret_cast<const uint4*>(m_kkt_terms.p);
    f.task_terms = m_task_terms.p;
    f.Vw = m_V.p;
    if (m_fuse_kkt_store) {
      f.store_lhs = m_lhs.p;
      f.store_rhs = m_rhs.p;
    }
    f.red = m_reduces.p;
    f.scales = m_scales.p;
  }
  return f;
}
KktFuse DeviceNlp::take_kkt_fuse() {
  const KktFuse f = kkt_fuse_for(m_kkt_pending);
  if (m_kkt_pending) {
    if (m_fuse_kkt_store) m_lhs_stale = m_rhs_stale = false;
    m_kkt_pending = 0;
  }
  return f;
}

BacksubFuse DeviceNlp::backsub_fuse(const LdltStats* publish) {
  BacksubFuse f;
  f.on = 1;
  f.V = m_V.p;
  f.s = m_s.p;
  f.z = m_z.p;
  f.mu = m_mu.p;
  f.ps = m_ps.p;
  f.pz = m_pz.p;
  f.off_ci = m_kdev.off_ci;
  f.plan = reinterpret_cast<const uint4*>(m_bs_plan.p);
  f.task_plan = m_bs_task_plan.p;
  f.stats_src = publish;
  f.stats_host = m_h_stats;
  f.seq_dev = m_seq_dev.p;
  f.seq_host = m_h_seq;
  return f;
}

void DeviceNlp::enqueue_factor(int parity, hipStream_t stream) {
  const LdltPlan& l = m_l_ref;
  if (!m_kkt_pending) materialize_kkt();  // e.g. a second attempt after one that evaluated the system in place
  LdltStats* cur = m_stats.p + static_cast<size_t>(parity) * m_batch;
  LdltStats* next = m_stats.p + static_cast<size_t>(parity ^ 1) * m_batch;
  const long long lxs = static_cast<long long>(std::max<int64_t>(1, l.nnzL));
  const int cs = static_cast<int>(std::max<uint32_t>(1, l.n_contrib));
  // a handful of workgroups read (δ, γ) straight from pinned host memory; the tens of
  // thousands of a big batch would each pay a PCIe round trip, so those get a device copy
  const double* reg = m_h_reg;
  if (m_batch > 8) {
    // (only when the values changed: consecutive steps of a batch mostly make the same first attempt, and the copy is a
    // launch of its own in front of every factorization — 6 us of a 64-problem step's 280, 21 us of a 512-problem step's)
    const size_t count = 2 * static_cast<size_t>(m_batch);
    if (m_reg_shadow.size() != count || std::memcmp(m_reg_shadow.data(), m_h_reg, count * sizeof(double)) != 0) {
      m_reg_shadow.assign(m_h_reg, m_h_reg + count);
      SLPX_HIP_CHECK(hipMemcpyAsync(m_reg_dev.p, m_h_reg, count * sizeof(double), hipMemcpyHostToDevice, stream));
    }
    reg = m_reg_dev.p;
  }
  if (m_dense) {
    if (m_dense_pivoted)
      hipLaunchKernelGGL(ldlt_dense_pivoted_factor_kernel, dim3(m_batch), dim3(kDenseThreads), m_dense_lds, stream, l.n, l.n_dec,
                         m_dense_colptr.p, m_dense_rowidx.p, m_kdev.nnz_lhs, m_lhs.p, reg, m_dense_A.p, m_dense_trans.p, m_D.p, cur, next);
    else
    hipLaunchKernelGGL(ldlt_dense_factor_kernel, dim3(m_batch), dim3(kDenseThreads), m_dense_lds, stream, l.n, l.n_dec,
                       m_dense_colptr.p, m_dense_rowidx.p, m_kdev.nnz_lhs, m_lhs.p, reg, m_dense_A.p, m_D.p, m_Lx.p, lxs, cur, next);
    SLPX_HIP_CHECK(hipGetLastError());
    return;
  }
  if (m_il) {
    const int C = (m_batch + 63) / 64;
    const int nnz = m_kdev.nnz_lhs;
    // (the assembly kernels wrote the interleaved arrays themselves unless somebody else made lhs / rhs)
    if (!m_lhs_in_il)
      hipLaunchKernelGGL(il_gather_kernel, dim3((nnz + 63) / 64, C), dim3(256), 0, stream, m_lhs.p,
                         static_cast<long long>(nnz), nnz, m_lhs_il.p, m_batch);
    if (!m_rhs_in_il)
      hipLaunchKernelGGL(il_gather_kernel, dim3((l.n + 63) / 64, C), dim3(256), 0, stream, m_rhs.p,
                         static_cast<long long>(l.n), l.n, m_rhs_il.p, m_batch);
    for (int r = 0; r < l.n_rounds; ++r) {
      const uint32_t nt = l.round_ptr[r + 1] - l.round_ptr[r];
      hipLaunchKernelGGL(ldlt_factor_il_kernel, dim3(nt, kIlRowsPerChunk * C), dim3(kIlFactorThreads), m_il_factor_lds, stream, m_ldev,
                         l.round_ptr[r], m_lhs_il.p, nnz, m_rhs_il.p, l.n, reg, m_Lx_il.p, lxs, m_D_il.p,
                         m_contrib_il.p, cs, m_zv_il.p, m_stats_part.p, m_batch, m_il_meta.p, m_il_meta_off.p);
    }
    hipLaunchKernelGGL(ldlt_stats_il_kernel, dim3(m_batch), dim3(64), 0, stream, m_stats_part.p,
                       static_cast<int>(l.tasks.size()), reg, cur, m_batch);
    m_il_outputs_stale = true;
  } else if (m_single_launch) {
    KktFuse f = take_kkt_fuse();
    // every round in one launch; tasks wait on device-side round counters
    hipLaunchKernelGGL(ldlt_factor_kernel<kFactorThreadsSingle>,
                       dim3(static_cast<uint32_t>(l.tasks.size()) + static_cast<uint32_t>(f.n_blocks), m_batch),
                       dim3(kFactorThreadsSingle), f.inline_kkt ? m_factor_lds_inline : l.factor_lds_bytes, stream,
                       m_ldev, 0u, m_lhs.p, m_kdev.nnz_lhs, reg, m_Lx.p, lxs, m_D.p, l.n, m_contrib.p, cs, cur, next,
                       m_rhs.p, m_zv.p, m_fround_cnt.p, m_slot_handoff ? 1 : 0, f);
  } else {
    for (int r = 0; r < l.n_rounds; ++r) {
      const uint32_t nt = l.round_ptr[r + 1] - l.round_ptr[r];
      hipLaunchKernelGGL(ldlt_factor_kernel<256>, dim3(nt, m_batch), dim3(256), l.factor_lds_bytes, stream,
                         m_ldev, l.round_ptr[r], m_lhs.p, m_kdev.nnz_lhs, reg, m_Lx.p, lxs, m_D.p,
                         l.n, m_contrib.p, cs, cur, r == 0 ? next : nullptr, m_rhs.p, m_zv.p,
                         static_cast<unsigned int*>(nullptr), 0, KktFuse{});
    }
  }
  SLPX_HIP_CHECK(hipGetLastError());
}

void DeviceNlp::factor(const std::vector<double>& delta, const std::vector<double>& gamma,
                       const std::vector<uint8_t>& active) {
  write_reg(delta, gamma, active);
  m_twin_mode = 0;
  m_stats_cur ^= 1;
  m_stats_in_host = false;
  enqueue_factor(m_stats_cur, m_stream);
}

void DeviceNlp::factor_solve_publish(const std::vector<double>& delta, const std::vector<double>& gamma,
                                     const std::vector<uint8_t>& active) {
  const bool mf_now = m_mf && xg_other() != nullptr;
  if (!m_fuse_solve || !mf_now) {
    factor(delta, gamma, active);
    solve_backsub_publish();
    return;
  }
  write_reg(delta, gamma, active);
  m_twin_mode = 0;
  m_stats_cur ^= 1;
  enqueue_factor_solve(m_stats_cur);
  m_stats_seq = ++m_seq_expected;
  m_stats_in_host = true;
}

// ---- twin attempt (ldlt_mf_twin_kernel) ----
bool DeviceNlp::twin_available() {
  if (m_twin_state != 0) return m_twin_state > 0 && m_mf && xg_other() != nullptr;
  m_twin_state = -1;
  const char* env = std::getenv("SLPX_TWIN");
  if (env != nullptr && env[0] == '0') return false;
  if (!m_mf || m_mf_mfma || m_batch != 1 || xg_other() == nullptr || !m_fuse_solve) return false;
  const LdltPlan& l = m_l_ref;
  // both attempts' workgroups resident at once (their tasks wait for each other inside the launch)
  int cus = 0, per_cu = 0;
  SLPX_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m_device));
  auto resident = [&](auto kernel, int threads) {
    SLPX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SLPX_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, m_mf_lds));
    return 2 * l.tasks.size() + m_reduces.n <= static_cast<size_t>(per_cu) * cus;
  };
  // (a 1024-thread workgroup of the step kernels is alone on its CU — 98 VGPRs x 16 waves; cart-pole N=1000 has 137
  // tasks, twice that is more than the 256 CUs.  Workgroups of 512 threads fit two to a CU and run the same image
  // 0.5 us slower at that horizon, to the same bits: a launch of two attempts takes those where the wide ones do not fit)
  m_twin_threads = m_mf_threads;
  bool fits = m_mf_threads == 1024 ? resident(&ldlt_mf_twin_kernel<1024, false>, 1024) : resident(&ldlt_mf_twin_kernel<512, false>, 512);
  if (!fits && m_mf_threads == 1024) {
    m_twin_threads = 512;
    fits = resident(&ldlt_mf_twin_kernel<512, false>, 512);
  }
  if (std::getenv("SLPX_LDLT_VERBOSE"))
    std::fprintf(stderr, "ldlt twin attempt: 2 x %zu tasks, %d workgroup(s) of %d threads per CU x %d CUs%s\n", l.tasks.size(), per_cu,
                 m_twin_threads, cus, fits ? "" : ": NOT resident at once");
  if (!fits) return false;
  for (auto [tw, first] : {std::pair{&m_Lx_tw, &m_Lx}, {&m_D_tw, &m_D}, {&m_zv_tw, &m_zv}, {&m_p_tw, &m_p}, {&m_ps_tw, &m_ps}, {&m_pz_tw, &m_pz}}) {
    tw->alloc(std::max<size_t>(1, first->n));
    tw->zero();
  }
  m_mf_contrib_tw.upload(std::vector<double>(std::max<size_t>(1, m_mf_contrib.n), std::bit_cast<double>(kSlotEmpty)));
  const std::vector<double> armed(static_cast<size_t>(l.n), std::bit_cast<double>(kSlotEmpty));
  m_xg_tw.upload(armed);
  m_xg2_tw.upload(armed);
  m_stats_tw.upload(std::vector<LdltStats>(2, LdltStats{0, 0, 0, 0, 0x7ff0000000000000ull}));
  SLPX_HIP_CHECK(hipDeviceSynchronize());  // (the memsets ran on the null stream)
  m_twin_state = 1;
  return true;
}

// One launch of the multifrontal step kernel (twin_mode != 0: two attempts, ldlt_mf_twin_kernel) with the buffer
// roles, parities and chain numbers of this moment; book_mf_step() is what the launch changes on the host.
void DeviceNlp::launch_mf_step(int twin_mode, const double* reg, const KktFuse& f, bool chained, const double* lhs2, const double* rhs2) {
  const LdltPlan& l = m_l_ref;
  const int parity = m_stats_cur ^ 1;
  LdltStats* cur = m_stats.p + static_cast<size_t>(parity);
  LdltStats* next = m_stats.p + static_cast<size_t>(parity ^ 1);
  BacksubFuse bf = backsub_fuse(cur);
  MfDev md;
  md.tasks = m_mf_tasks.p;
  md.fronts = m_mf_fronts.p;
  md.image = m_mf_image.p;
  md.image_stride16 = m_mf_image_stride16;
  md.image_desc = m_mf_image_desc.p;
  md.n_tasks = static_cast<unsigned int>(l.tasks.size());
  md.exit_cnt = m_exit_cnt.p;
  // a chained step (sweep_full_for_step): the kernel variant that waits for its sweep itself and whose last
  // workgroup tells the next step's sweep; the main stream stays "untouched" by a step's own launches
  md.chain = chained ? m_chain.p : nullptr;
  md.wait_step = chained ? m_chain_sweep_wgs : 0u;  // (every workgroup of the chained sweeps so far, this step's included)
  md.this_step = m_chain_seq;
  md.delta = reg[0];  // (by value: MfDev)
  md.gamma = reg[1];
  if (m_gate_next_step) {  // (one launch: the step enqueued before the iteration in front of it was decided)
    md.gate = m_ipm_gate.p;
    m_gate_next_step = false;
  }
  const double* const reg_by_value = nullptr;
  if (twin_mode != 0) {
    const int tw_parity = m_stats_tw_cur ^ 1;
    MfTwin tw;
    tw.first_end = md.n_tasks + static_cast<unsigned int>(f.n_blocks);
    tw.delta = reg[2];
    tw.gamma = reg[3];
    tw.Lx = m_Lx_tw.p;
    tw.D = m_D_tw.p;
    tw.contrib = m_mf_contrib_tw.p;
    tw.zv = m_zv_tw.p;
    tw.xg = m_xg_tw_parity ? m_xg2_tw.p : m_xg_tw.p;
    tw.xg_next = m_xg_tw_parity ? m_xg_tw.p : m_xg2_tw.p;
    tw.out = m_p_tw.p;
    tw.ps = m_ps_tw.p;
    tw.pz = m_pz_tw.p;
    tw.stats = m_stats_tw.p + static_cast<size_t>(tw_parity);
    tw.stats_next = m_stats_tw.p + static_cast<size_t>(tw_parity ^ 1);
    tw.lhs = lhs2;
    tw.rhs = rhs2;
    MfRide ride;
    size_t lds = m_mf_lds;
    if (m_ride_next) {
      // the error launch of the iteration before this step (ipm_ride_errors_in_next_step): its workgroups in front of
      // the tasks, on the CURRENT buffers — the look-ahead iterate has been made current —, the tape's sums with them
      m_ride_next = false;
      const int work = std::max({m_kdev.n, m_kdev.m_e, m_kdev.m_i, 1});
      const int blocks = grid_for(8 * work, kIpmErrThreads, 64);
      if (m_ipm_partial.n < static_cast<size_t>(blocks) * kIpmErrQ) m_ipm_partial.alloc(static_cast<size_t>(256) * kIpmErrQ);
      if (m_ipm_err_done.n == 0) m_ipm_err_done.upload(std::vector<unsigned int>(1, 0u));
      const int n_sums = static_cast<int>(m_reduces.n);
      m_ride_ticket += 1.0;
      md.ride_verdict = m_ipm_ride_verdict.p;
      md.ride_ticket = m_ride_ticket;
      md.gate = nullptr;  // (nothing in front of this launch is undecided: its own riding workgroups decide)
      ride.n_blocks = static_cast<uint32_t>(blocks + n_sums);
      ride.K = m_kdev;
      ride.V = m_V.p;
      ride.x = m_in.p;
      ride.s = m_s.p;
      ride.y = m_y.p;
      ride.z = m_z.p;
      ride.scales = m_ipm_scales.p;
      ride.nV = m_s_ref.nV;
      ride.partial = m_ipm_partial.p;
      IpmErrFinish& fin = ride.fin;
      fin.n_err_blocks = blocks;
      fin.n_total_blocks = blocks + n_sums;
      fin.red = m_reduces.p;
      fin.tape_scales = m_scales.p;
      fin.Vw = m_V.p;
      fin.done = m_ipm_err_done.p;
      fin.out = &m_ipm_host[m_ride_slot].err_ahead;
      fin.ctl = static_cast<IpmCtl*>(m_ipm_ctl_dev);
      fin.dir = m_ipm_alpha.p;
      fin.gate = m_ipm_gate.p;
      fin.go_host = &m_ipm_host[m_ride_slot].go;
      fin.check_host = &m_ipm_host[m_ride_slot].check;
      fin.ride_verdict = m_ipm_ride_verdict.p;
      fin.ride_ticket = m_ride_ticket;
      lds = std::max<size_t>(lds, sizeof(double) * kIpmErrLdsDoubles);
    }
    const dim3 grid(2u * md.n_tasks + static_cast<uint32_t>(f.n_blocks) + ride.n_blocks);
    md.n_workgroups = grid.x;
    auto launch = [&](auto kernel, int threads) {
      hipLaunchKernelGGL(kernel, grid, dim3(threads), lds, m_stream.raw(), m_ldev, md, m_lhs.p, m_rhs.p, reg_by_value, m_Lx.p, m_D.p, l.n,
                         m_mf_contrib.p, cur, next, m_zv.p, f, xg_now(), xg_other(), m_p.p, bf, tw, ride);
    };
    if (ride.n_blocks != 0) {
      if (m_twin_threads == 1024) launch(&ldlt_mf_twin_kernel<1024, true>, 1024);
      else launch(&ldlt_mf_twin_kernel<512, true>, 512);
    } else {
      if (m_twin_threads == 1024) launch(&ldlt_mf_twin_kernel<1024, false>, 1024);
      else launch(&ldlt_mf_twin_kernel<512, false>, 512);
    }
  } else {
    const dim3 grid(static_cast<uint32_t>(l.tasks.size()) + static_cast<uint32_t>(f.n_blocks));
    md.n_workgroups = grid.x;
    auto launch = [&](auto kernel, int threads) {
      hipLaunchKernelGGL(kernel, grid, dim3(threads), m_mf_lds, m_stream.raw(), m_ldev, md, m_lhs.p, m_rhs.p, reg_by_value, m_Lx.p, m_D.p,
                         l.n, m_mf_contrib.p, cur, next, m_zv.p, f, xg_now(), xg_other(), m_p.p, bf);
    };
    if (m_mf_threads == 1024) {
      if (chained) m_mf_mfma ? launch(&ldlt_mf_step_kernel<1024, true, true>, 1024) : launch(&ldlt_mf_step_kernel<1024, false, true>, 1024);
      else m_mf_mfma ? launch(&ldlt_mf_step_kernel<1024, true, false>, 1024) : launch(&ldlt_mf_step_kernel<1024, false, false>, 1024);
    } else {
      if (chained) m_mf_mfma ? launch(&ldlt_mf_step_kernel<512, true, true>, 512) : launch(&ldlt_mf_step_kernel<512, false, true>, 512);
      else m_mf_mfma ? launch(&ldlt_mf_step_kernel<512, true, false>, 512) : launch(&ldlt_mf_step_kernel<512, false, false>, 512);
    }
    if (chained) m_chain_step_wgs += md.n_workgroups;  // (what the next chained sweep waits for in chain[48])
  }
  SLPX_HIP_CHECK(hipGetLastError());
}

void DeviceNlp::book_mf_step(int twin_mode, bool chained) {
  m_stats_cur ^= 1;
  if (twin_mode != 0) {
    m_stats_tw_cur ^= 1;
    m_xg_tw_parity ^= 1;
  }
  xg_flip();
  m_stream.tape_pending = false;
  m_last_step_chained = chained;
  m_twin_mode = twin_mode;
}

bool DeviceNlp::factor_solve_publish_twin(double delta0, double gamma0, double delta1, double gamma1, int mode) {
  if (!twin_available() || m_stream.tape_pending) return false;
  m_h_reg[0] = delta0;
  m_h_reg[1] = gamma0;
  m_h_reg[2] = delta1;
  m_h_reg[3] = gamma1;
  if (!m_kkt_pending) materialize_kkt();
  const KktFuse f = take_kkt_fuse();
  launch_mf_step(mode, m_h_reg, f, false);
  book_mf_step(mode, false);
  m_stats_seq = ++m_seq_expected;
  m_stats_in_host = true;
  return true;
}

bool DeviceNlp::factor_solve_publish_twin_written(double delta0, double gamma0, double delta1, double gamma1, int mode, const double* lhs2,
                                                  const double* rhs2) {
  if (!twin_available() || m_stream.tape_pending || m_kkt_pending || m_lhs_stale || m_rhs_stale) return false;
  m_h_reg[0] = delta0;
  m_h_reg[1] = gamma0;
  m_h_reg[2] = delta1;
  m_h_reg[3] = gamma1;
  launch_mf_step(mode, m_h_reg, KktFuse{}, false, lhs2, rhs2);
  book_mf_step(mode, false);
  m_stats_seq = ++m_seq_expected;
  m_stats_in_host = true;
  return true;
}

void DeviceNlp::adopt_twin() {
  // every launch takes these pointers when it is made (ipm_accept_lookahead): later solves with this factorization,
  // the committed direction and the counters are the second attempt's
  m_Lx.swap(m_Lx_tw);
  m_D.swap(m_D_tw);
  m_zv.swap(m_zv_tw);
  m_p.swap(m_p_tw);
  m_ps.swap(m_ps_tw);
  m_pz.swap(m_pz_tw);
  m_stats.swap(m_stats_tw);
  std::swap(m_stats_cur, m_stats_tw_cur);
  m_h_stats[0] = m_h_stats[1];
  m_twin_mode = 0;
}

void DeviceNlp::enqueue_factor_solve(int parity) {
  if (!(m_mf && xg_other() != nullptr)) throw std::logic_error("slpx: a one-launch step without the multifrontal plan");
  if (!m_kkt_pending) materialize_kkt();
  KktFuse f = take_kkt_fuse();
  const bool chained = m_stream.tape_pending;
  m_stats_cur = parity ^ 1;  // (the callers flipped it already; launch_mf_step / book_mf_step do it themselves)
  launch_mf_step(0, m_h_reg, f, chained);
  book_mf_step(0, chained);
}

void DeviceNlp::read_stats(std::vector<LdltStats>& out) {
  out.resize(m_batch);
  if (!m_stats_in_host)
    SLPX_HIP_CHECK(hipMemcpyAsync(m_h_stats, m_stats.p + static_cast<size_t>(m_stats_cur) * m_batch,
                                  m_batch * sizeof(LdltStats), hipMemcpyDeviceToHost, m_stream));
  // Busy-poll instead of a blocking wait: the interrupt-driven wake-up of
  // hipStreamSynchronize costs tens of microseconds, a tenth of a whole Newton step.
  if (m_stats_in_host && m_batch == 1) {
    // the publishing kernel's sequence number (see step_backsub_kernel); the stream is
    // consulted now and then so that a failed launch cannot hang the host
    spin_on_published([&] { return *m_h_seq >= m_stats_seq; }, m_stream.raw(), "slpx: step finished without publishing its counters");
  } else {
    hipError_t st;
    while ((st = hipStreamQuery(m_stream.raw())) == hipErrorNotReady) {
    }
    SLPX_HIP_CHECK(st);
  }
  std::copy(m_h_stats, m_h_stats + m_batch, out.begin());
  // (an experiment's knob: the host sits on the verdict for so many nanoseconds — how much of its reaction a step
  // hides: profiles/host_slack.sh)
  static const long delay_ns = [] {
    const char* env = std::getenv("SLPX_DEBUG_STATS_DELAY_NS");
    return env ? std::atol(env) : 0L;
  }();
  if (delay_ns > 0) {
    const auto until = std::chrono::steady_clock::now() + std::chrono::nanoseconds(delay_ns);
    while (std::chrono::steady_clock::now() < until) {
    }
  }
}

// Full solve for a right-hand side that arrived AFTER the factorization (second-order
// corrections, multiplier estimate, slpx_ldlt_solve): forward, then backward.
void DeviceNlp::solve() {
  if (m_rhs_stale) build_rhs();
  const LdltPlan& l = m_l_ref;
  if (m_dense) {
    solve_after_factor();
    return;
  }
  if (m_mf && m_mf_solve && m_batch == 1 && xg_other() != nullptr) {
    MfDev md;
    md.tasks = m_mf_tasks.p;
    md.fronts = m_mf_fronts.p;
    md.image = m_mf_image.p;
    md.image_stride16 = m_mf_image_stride16;
    md.image_desc = m_mf_image_desc.p;
    md.n_tasks = static_cast<unsigned int>(l.tasks.size());
    const dim3 grid(static_cast<uint32_t>(l.tasks.size()));
    if (m_mf_threads == 1024)
      hipLaunchKernelGGL(ldlt_mf_solve_kernel<1024>, grid, dim3(1024), m_mf_lds, m_stream, m_ldev, md, m_Lx.p, m_D.p, m_rhs.p,
                         m_mf_contrib.p, xg_now(), xg_other(), m_p.p);
    else
      hipLaunchKernelGGL(ldlt_mf_solve_kernel<512>, grid, dim3(512), m_mf_lds, m_stream, m_ldev, md, m_Lx.p, m_D.p, m_rhs.p,
                         m_mf_contrib.p, xg_now(), xg_other(), m_p.p);
    xg_flip();
    SLPX_HIP_CHECK(hipGetLastError());
    return;
  }
  const long long lxs = static_cast<long long>(std::max<int64_t>(1, l.nnzL));
  const int scs = static_cast<int>(std::max<uint32_t>(1, l.n_scontrib));
  if (m_il) {
    const int C = (m_batch + 63) / 64;
    if (!m_rhs_in_il)
      hipLaunchKernelGGL(il_gather_kernel, dim3((l.n + 63) / 64, C), dim3(256), 0, m_stream, m_rhs.p,
                         static_cast<long long>(l.n), l.n, m_rhs_il.p, m_batch);
    for (int r = 0; r < l.n_rounds; ++r) {
      const uint32_t nt = l.round_ptr[r + 1] - l.round_ptr[r];
      hipLaunchKernelGGL(ldlt_fwd_il_kernel, dim3(nt, C), dim3(kIlLanes), m_il_solve_lds, m_stream, m_ldev,
                         l.round_ptr[r], m_rhs_il.p, l.n, m_Lx_il.p, lxs, m_D_il.p, m_scontrib_il.p, scs,
                         m_zv_il.p);
    }
    solve_after_factor();
    return;
  }
  if (m_single_launch && m_batch == 1) {
    // one problem: every round in ONE launch (the tasks order themselves through the round counters)
    hipLaunchKernelGGL(ldlt_fwd_kernel, dim3(static_cast<uint32_t>(l.tasks.size()), 1), dim3(256), l.solve_lds_bytes, m_stream,
                       m_ldev, 0u, m_rhs.p, l.n, m_Lx.p, lxs, m_D.p, m_scontrib.p, scs, m_zv.p, m_fround_cnt.p);
  } else {
    for (int r = 0; r < l.n_rounds; ++r) {
      const uint32_t nt = l.round_ptr[r + 1] - l.round_ptr[r];
      hipLaunchKernelGGL(ldlt_fwd_kernel, dim3(nt, m_batch), dim3(256), l.solve_lds_bytes, m_stream,
                         m_ldev, l.round_ptr[r], m_rhs.p, l.n, m_Lx.p, lxs, m_D.p, m_scontrib.p, scs,
                         m_zv.p, static_cast<unsigned int*>(nullptr));
    }
  }
  solve_after_factor();
}

// The factorization carried the rhs along as an extra row and left z = D⁻¹L⁻¹Pb behind
// (ldlt_symbolic.cpp), so only the backward substitution remains.
void DeviceNlp::solve_after_factor() { solve_after_factor_impl(nullptr); }

// backward solve, back-substitution and the hand-over of the current attempt's counters to the
// host: one launch where the back-substitution can ride along (device.hpp: BacksubFuse)
void DeviceNlp::solve_backsub_publish() {
  const LdltStats* src = m_stats.p + static_cast<size_t>(m_stats_cur) * m_batch;
  if (m_fuse_backsub) {
    solve_after_factor_impl(src);
    if (m_batch == 1) m_stats_seq = ++m_seq_expected;
  } else {
    solve_after_factor_impl(nullptr);
    backsub_and_publish(src);
  }
  m_stats_in_host = true;
}

void DeviceNlp::solve_after_factor_impl(const LdltStats* publish) {
  const LdltPlan& l = m_l_ref;
  const long long lxs = static_cast<long long>(std::max<int64_t>(1, l.nnzL));
  if (m_dense) {  // (nothing rides in a dense factorization: forward and backward substitution from the rhs in memory)
    if (m_rhs_stale) build_rhs();
    hipLaunchKernelGGL(ldlt_dense_solve_kernel, dim3(m_batch), dim3(kDenseThreads), 8u * static_cast<uint32_t>(l.n) + 16u, m_stream, l.n,
                       m_dense_A.p, m_rhs.p, m_p.p, m_dense_pivoted ? m_dense_trans.p : nullptr);
    SLPX_HIP_CHECK(hipGetLastError());
    return;
  }
  if (m_il) {
    const int C = (m_batch + 63) / 64;
    for (int r = l.n_rounds - 1; r >= 0; --r) {
      const uint32_t nt = l.round_ptr[r + 1] - l.round_ptr[r];
      hipLaunchKernelGGL(ldlt_bwd_il_kernel, dim3(nt, C), dim3(kIlLanes * kIlBwdWaves), m_il_solve_lds, m_stream, m_ldev,
                         l.round_ptr[r], l.n, m_Lx_il.p, lxs, m_zv_il.p, m_xg_il.p, m_p.p, m_batch);
    }
    SLPX_HIP_CHECK(hipGetLastError());
    return;
  }
  if (m_single_launch) {
    const uint32_t nt = static_cast<uint32_t>(l.tasks.size());
    const BacksubFuse f = publish != nullptr ? backsub_fuse(publish) : BacksubFuse{};
    hipLaunchKernelGGL(ldlt_bwd_kernel, dim3(nt, m_batch), dim3(256), f.on ? m_solve_lds_inline : l.solve_lds_bytes,
                       m_stream, m_ldev, nt - 1, l.n, m_Lx.p, lxs, m_zv.p, xg_now(), xg_other(), m_p.p, m_bround_cnt.p, f);
    xg_flip();
  } else {
    for (int r = l.n_rounds - 1; r >= 0; --r) {
      const uint32_t nt = l.round_ptr[r + 1] - l.round_ptr[r];
      hipLaunchKernelGGL(ldlt_bwd_kernel, dim3(nt, m_batch), dim3(256), l.solve_lds_bytes, m_stream,
                         m_ldev, l.round_ptr[r], l.n, m_Lx.p, lxs, m_zv.p, m_xg.p, static_cast<double*>(nullptr), m_p.p,
                         static_cast<unsigned int*>(nullptr), BacksubFuse{});
    }
  }
  SLPX_HIP_CHECK(hipGetLastError());
}

// Batch-interleaved mode keeps L and D as [chunk][index][lane]; the batch-major copies that
// slpx_system_get hands out are made on demand.
void DeviceNlp::materialize_factor() {
  if (!m_il || !m_il_outputs_stale) return;
  const LdltPlan& l = m_l_ref;
  const int C = (m_batch + 63) / 64;
  const int nl = static_cast<int>(std::max<int64_t>(1, l.nnzL));
  hipLaunchKernelGGL(il_scatter_kernel, dim3((nl + 63) / 64, C), dim3(256), 0, m_stream, m_Lx_il.p, nl, m_Lx.p,
                     static_cast<long long>(nl), m_batch);
  hipLaunchKernelGGL(il_scatter_kernel, dim3((l.n + 63) / 64, C), dim3(256), 0, m_stream, m_D_il.p, l.n, m_D.p,
                     static_cast<long long>(l.n), m_batch);
  SLPX_HIP_CHECK(hipGetLastError());
  m_il_outputs_stale = false;
}

// Lane-per-problem pays from one 64-problem chunk per task on (r02: from ~200 problems — the backward
// solve was one wave per task and chunk, a chain of columns; with the columns of a level dealt to four
// waves, 64 x N=500: 251 k steps/s against 224-238 k for the per-task kernels,
// profiles/r03_b64_probe.txt; 512 x N=1000: 508 k vs 171 k).  SLPX_LDLT_IL=0 turns it off,
// SLPX_IL_MIN_BATCH moves the threshold.
bool DeviceNlp::interleaved_for(int batch) {
  if (const char* env = std::getenv("SLPX_LDLT_IL"))
    if (env[0] == '0') return false;
  int min_batch = 64;
  if (const char* env = std::getenv("SLPX_IL_MIN_BATCH")) min_batch = std::atoi(env);
  return batch >= min_batch;
}

// p <- p + K_reg^-1 (b - K p), `iters` times: b = the right-hand side now in m_rhs, p = the
// solution of the last solve(), K = the unregularized matrix in m_lhs, K_reg = what was factored.
// Converges like (regularization / smallest eigenvalue)^iters.  Single problem.
void DeviceNlp::refine_solution(int iters) {
  materialize_kkt();
  if (m_batch != 1) throw std::runtime_error("slpx: refine_solution handles one problem");
  const int dim = m_kdev.dim;
  if (m_lhs_colptr.n == 0) {
    m_lhs_colptr.upload(m_k_ref.lhs.colptr);
    m_lhs_rowidx.upload(m_k_ref.lhs.rowidx);
    {
      // the strictly lower triangle by rows, entries of a row in column order
      const CscPattern& lp = m_k_ref.lhs;
      std::vector<int32_t> rowptr(dim + 1, 0), rowent, rowcol;
      for (int c = 0; c < dim; ++c)
        for (int q = lp.colptr[c]; q < lp.colptr[c + 1]; ++q)
          if (lp.rowidx[q] != c) ++rowptr[lp.rowidx[q] + 1];
      for (int i = 0; i < dim; ++i) rowptr[i + 1] += rowptr[i];
      rowent.resize(std::max(1, rowptr[dim]));
      rowcol.resize(std::max(1, rowptr[dim]));
      std::vector<int32_t> fill(rowptr.begin(), rowptr.end() - 1);
      for (int c = 0; c < dim; ++c)
        for (int q = lp.colptr[c]; q < lp.colptr[c + 1]; ++q)
          if (lp.rowidx[q] != c) {
            const int at = fill[lp.rowidx[q]]++;
            rowent[at] = q;
            rowcol[at] = c;
          }
      m_lhs_rowptr.upload(rowptr);
      m_lhs_rowent.upload(rowent);
      m_lhs_rowcol.upload(rowcol);
    }
    m_rhs0.alloc(dim);
    m_p_acc.alloc(dim);
  }
  SLPX_HIP_CHECK(hipMemcpyAsync(m_rhs0.p, m_rhs.p, dim * sizeof(double), hipMemcpyDeviceToDevice, m_stream));
  SLPX_HIP_CHECK(hipMemcpyAsync(m_p_acc.p, m_p.p, dim * sizeof(double), hipMemcpyDeviceToDevice, m_stream));
  for (int it = 0; it < iters; ++it) {
    SLPX_HIP_CHECK(hipMemcpyAsync(m_rhs.p, m_rhs0.p, dim * sizeof(double), hipMemcpyDeviceToDevice, m_stream));
    hipLaunchKernelGGL(sym_residual_kernel, dim3(grid_for(dim, 256)), dim3(256), 0, m_stream, dim, m_lhs_colptr.p,
                       m_lhs_rowidx.p, m_lhs_rowptr.p, m_lhs_rowent.p, m_lhs_rowcol.p, m_lhs.p, m_p_acc.p, m_rhs.p);
    solve();  // m_rhs -> m_p
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(dim, 256)), dim3(256), 0, m_stream, dim, m_p.p, m_p_acc.p);
  }
  SLPX_HIP_CHECK(hipMemcpyAsync(m_p.p, m_p_acc.p, dim * sizeof(double), hipMemcpyDeviceToDevice, m_stream));
  SLPX_HIP_CHECK(hipMemcpyAsync(m_rhs.p, m_rhs0.p, dim * sizeof(double), hipMemcpyDeviceToDevice, m_stream));
  SLPX_HIP_CHECK(hipGetLastError());
}

void DeviceNlp::backsub() { backsub_and_publish(nullptr); }

// back-substitution that also hands the counters of the factorization just enqueued to the
// host (read_stats() then needs no copy)
void DeviceNlp::backsub_publish() {
  backsub_and_publish(m_stats.p + static_cast<size_t>(m_stats_cur) * m_batch);
  m_stats_in_host = true;
}

// stats_src != nullptr: also copy those counters to the pinned host buffer
void DeviceNlp::backsub_and_publish(const LdltStats* stats_src) {
  if (m_kdev.m_i == 0 && stats_src == nullptr) return;
  hipLaunchKernelGGL(step_backsub_kernel, dim3(grid_for(std::max(1, m_kdev.m_i), 256), m_batch),
                     dim3(256), 0, m_stream, m_kdev, m_V.p, m_s_ref.nV, m_p.p, m_s.p, m_z.p, m_mu.p,
                     m_ps.p, m_pz.p, stats_src, stats_src ? m_h_stats : nullptr, m_seq_dev.p,
                     (stats_src && m_batch == 1) ? m_h_seq : nullptr);
  if (stats_src && m_batch == 1) m_stats_seq = ++m_seq_expected;
  SLPX_HIP_CHECK(hipGetLastError());
}

// ============================================================================
// Interior-point iteration on the device (ipm_kernels.h)
// ============================================================================

void DeviceNlp::ipm_enable() {
  if (m_ipm) return;
  if (m_batch != 1) throw std::runtime_error("slpx: the device-resident IPM iteration handles one problem");
  const NlpStructure& s = m_s_ref;
  SLPX_HIP_CHECK(hipStreamSynchronize(m_stream));
  m_trial_in.alloc(s.n_inputs());
  SLPX_HIP_CHECK(hipMemcpy(m_trial_in.p, m_in.p, s.n_inputs() * sizeof(double), hipMemcpyDeviceToDevice));
  m_V_trial.alloc(s.nV);
  // the trial V starts as a copy of V: static entries in place, the swept ones overwritten
  SLPX_HIP_CHECK(hipMemcpy(m_V_trial.p, m_V.p, static_cast<size_t>(s.nV) * sizeof(double), hipMemcpyDeviceToDevice));
  m_soc_ce.alloc(std::max(1, s.m_e));
  m_soc_cims.alloc(std::max(1, s.m_i));
  m_p_keep.alloc(m_kdev.dim);
  m_ps_keep.alloc(std::max(1, s.m_i));
  m_pz_keep.alloc(std::max(1, s.m_i));
  m_ipm_alpha.alloc(4);  // alpha_max, alpha_z, "this attempt's look-ahead chain is void", -
  m_ipm_alpha.zero();
  m_s_ahead.alloc(std::max(1, s.m_i));
  m_y_ahead.alloc(std::max(1, s.m_e));
  m_z_ahead.alloc(std::max(1, s.m_i));
  SLPX_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&m_ipm_host), 2 * sizeof(IpmHost)));
  std::memset(m_ipm_host, 0, 2 * sizeof(IpmHost));
  SLPX_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&m_ipm_ctl_host), 3 * sizeof(IpmCtl)));
  for (int k = 0; k < 3; ++k) m_ipm_ctl_host[k] = IpmCtl{};
  SLPX_HIP_CHECK(hipMalloc(&m_ipm_ctl_dev, sizeof(IpmCtl)));
  SLPX_HIP_CHECK(hipMemset(m_ipm_ctl_dev, 0, sizeof(IpmCtl)));
  m_ipm_gate.upload(std::vector<double>(1, 1.0));
  m_ipm = true;
}

void DeviceNlp::ipm_set_error_scaling(const std::vector<double>& scales) {
  if (static_cast<int>(scales.size()) != m_s_ref.n_scales())
    throw std::runtime_error("ipm_set_error_scaling: wrong length");
  SLPX_HIP_CHECK(hipStreamSynchronize(m_stream));
  m_ipm_scales.upload(scales);
}

// After ipm_trial_metrics() / ipm_errors(): spins on the sequence number those launches
// publish (the stream is consulted now and then so a failed launch cannot hang the host).
void DeviceNlp::wait_published() {
  spin_on_published([&] { return *m_h_seq >= m_seq_expected; }, m_stream.raw(), "slpx: chain finished without publishing");
}

void DeviceNlp::wait() {
  hipError_t st;
  while ((st = hipStreamQuery(m_stream.raw())) == hipErrorNotReady) {
  }
  SLPX_HIP_CHECK(st);
}

void DeviceNlp::ipm_direction(double tau) {
  hipLaunchKernelGGL(ipm_direction_kernel, dim3(1), dim3(kIpmThreads), 0, m_stream, m_kdev, m_V.p, m_in.p, m_s.p,
                     m_z.p, m_p.p, m_ps.p, m_pz.p, m_mu.p, tau, m_trial_in.p, m_ipm_alpha.p, &m_ipm_host[m_ipm_slot].dir);
  SLPX_HIP_CHECK(hipGetLastError());
}

IpmLookaheadArgs DeviceNlp::lookahead_args(double tau, int twin_mode) {
  IpmLookaheadArgs a;
  a.n = m_kdev.n;
  a.m_e = m_kdev.m_e;
  a.m_i = m_kdev.m_i;
  a.g_src = m_kdev.g_src;
  a.V = m_V.p;
  a.in = m_in.p;
  a.s = m_s.p;
  a.y = m_y.p;
  a.z = m_z.p;
  a.p = m_p.p;
  a.ps = m_ps.p;
  a.pz = m_pz.p;
  a.mu = m_mu.p;
  a.tau = tau;
  a.in_t = m_trial_in.p;
  a.s_t = m_s_ahead.p;
  a.y_t = m_y_ahead.p;
  a.z_t = m_z_ahead.p;
  a.alpha_dev = m_ipm_alpha.p;
  a.out = &m_ipm_host[m_ipm_slot].dir;
  a.stats = m_stats.p + static_cast<size_t>(m_stats_cur) * m_batch;
  if (twin_mode != 0) {  // (a twin launch: the kernel takes the direction of the attempt the policy takes)
    a.tw.mode = twin_mode;
    a.tw.p = m_p_tw.p;
    a.tw.ps = m_ps_tw.p;
    a.tw.pz = m_pz_tw.p;
    a.tw.stats = m_stats_tw.p + static_cast<size_t>(m_stats_tw_cur);
  }
  return a;
}

void DeviceNlp::ipm_lookahead(double tau) {
  hipLaunchKernelGGL(ipm_lookahead_kernel, dim3(1), dim3(kIpmThreads), 0, m_stream, lookahead_args(tau, m_twin_mode));
  SLPX_HIP_CHECK(hipGetLastError());
}

void DeviceNlp::sweep_full_lookahead(bool with_reduce, bool skippable) {
  m_in_override = m_trial_in.p;
  m_V_override = m_V_trial.p;
  m_tape_reduce = with_reduce;  // false: the separable sums ride in ipm_errors(.., sums_ride, ahead)
  // (the generated kernel leaves at once when the look-ahead launch flagged the attempt as rejected: `chain` with
  // n_workgroups = 0 is that flag, tape_jit.cpp)
  if (skippable && m_full.n_bodies && m_ipm_alpha.p != nullptr) {
    m_chain_args = ChainArgs{reinterpret_cast<unsigned int*>(m_ipm_alpha.p + 2), 0u, 0u};
    m_chain_args.skip_flag = true;
  }
  launch_tape(m_full, true);
  m_chain_args = ChainArgs{};
  m_tape_reduce = true;
  m_in_override = nullptr;
  m_V_override = nullptr;
}

void DeviceNlp::ipm_accept_lookahead() {
  // every launch takes these pointers when it is made: the next step reads the accepted iterate
  m_in.swap(m_trial_in);
  m_s.swap(m_s_ahead);
  m_y.swap(m_y_ahead);
  m_z.swap(m_z_ahead);
  m_V.swap(m_V_trial);
  m_lhs_stale = m_rhs_stale = true;
}

void DeviceNlp::ipm_trial_point(double alpha) {
  hipLaunchKernelGGL(ipm_trial_point_kernel, dim3(grid_for(m_kdev.n, 256)), dim3(256), 0, m_stream, m_kdev.n,
                     m_in.p, m_p.p, alpha, m_trial_in.p);
  SLPX_HIP_CHECK(hipGetLastError());
}

void DeviceNlp::ipm_trial_metrics(double alpha, bool s_from_ci) {
  hipLaunchKernelGGL(ipm_trial_metrics_kernel, dim3(1), dim3(kIpmThreads), 0, m_stream, m_kdev, m_V_trial.p,
                     m_s.p, m_ps.p, alpha, m_ipm_alpha.p, s_from_ci ? 1 : 0, &m_ipm_host[m_ipm_slot].trial, m_seq_dev.p, m_h_seq,
                     m_reduces.p, static_cast<int>(m_reduces.n), m_scales.p);
  ++m_seq_expected;
  SLPX_HIP_CHECK(hipGetLastError());
}

void DeviceNlp::ipm_commit(double alpha, double alpha_z, bool s_from_ci) {
  const int work = std::max({m_kdev.n, m_kdev.m_e, m_kdev.m_i, 1});
  hipLaunchKernelGGL(ipm_commit_kernel, dim3(grid_for(work, 256)), dim3(256), 0, m_stream, m_kdev, m_V_trial.p,
                     m_p.p, m_ps.p, m_pz.p, alpha, alpha_z, m_mu.p, s_from_ci ? 1 : 0, m_in.p, m_s.p, m_y.p,
                     m_z.p);
  SLPX_HIP_CHECK(hipGetLastError());
}

// `sums_ride`: the sweep before this call left the tape's separable sums out
// (sweep_full(false)); they ride in this launch
void DeviceNlp::ipm_errors(bool check_all_V, bool sums_ride, bool ahead) {
  const int work = std::max({m_kdev.n, m_kdev.m_e, m_kdev.m_i, 1});
  // eight lanes per column (ipm_error_accumulate): up to 64 workgroups of 256
  const int blocks = grid_for(8 * work, kIpmErrThreads, 64);
  if (m_ipm_partial.n < static_cast<size_t>(blocks) * kIpmErrQ) m_ipm_partial.alloc(static_cast<size_t>(256) * kIpmErrQ);
  if (m_ipm_err_done.n == 0) m_ipm_err_done.upload(std::vector<unsigned int>(1, 0u));
  const int n_sums = sums_ride ? static_cast<int>(m_reduces.n) : 0;
  IpmErrFinish fin;
  fin.n_err_blocks = blocks;
  fin.n_total_blocks = blocks + n_sums;
  fin.red = m_reduces.p;
  fin.tape_scales = m_scales.p;
  fin.Vw = ahead ? m_V_trial.p : m_V.p;
  fin.done = m_ipm_err_done.p;
  fin.out = ahead ? &m_ipm_host[m_ipm_slot].err_ahead : &m_ipm_host[m_ipm_slot].err;
  if (ahead && m_errors_decide) {
    fin.ctl = static_cast<IpmCtl*>(m_ipm_ctl_dev);
    fin.dir = m_ipm_alpha.p;
    fin.gate = m_ipm_gate.p;
    fin.go_host = &m_ipm_host[m_ipm_slot].go;
  }
  fin.seq_dev = m_seq_dev.p;
  fin.seq_host = m_h_seq;
  fin.skip = ahead ? m_ipm_alpha.p + 2 : nullptr;
  hipLaunchKernelGGL(ipm_error_partial_kernel, dim3(blocks + n_sums), dim3(kIpmErrThreads), 0, m_stream, m_kdev, fin.Vw,
                     m_s_ref.nV, ahead ? m_trial_in.p : m_in.p, ahead ? m_s_ahead.p : m_s.p, ahead ? m_y_ahead.p : m_y.p,
                     ahead ? m_z_ahead.p : m_z.p, m_ipm_scales.p, check_all_V ? 1 : 0, m_ipm_partial.p, fin);
  ++m_seq_expected;
  SLPX_HIP_CHECK(hipGetLastError());
}

bool DeviceNlp::ipm_ride_errors_in_next_step(int slot_out) {
  if (!ipm_ride_possible()) return false;
  m_ride_next = true;
  m_ride_slot = slot_out;
  return true;
}

bool DeviceNlp::ipm_ride_possible() {
  if (!twin_available()) return false;
  if (m_ride_lds_ok < 0) {
    // the riding workgroups' LDS (ipm_error_block) beside the tasks': the launch takes the larger of the two
    const char* env = std::getenv("SLPX_IPM_RIDE");
    m_ride_lds_ok = (env == nullptr || env[0] != '0') ? 1 : 0;
    if (m_ride_lds_ok) {
      int cus = 0, per_cu = 0;
      SLPX_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m_device));
      const size_t lds = std::max<size_t>(m_mf_lds, sizeof(double) * kIpmErrLdsDoubles);
      const size_t wgs = 2 * m_l_ref.tasks.size() + 64 + m_reduces.n;
      auto resident = [&](auto kernel, int threads) {
        SLPX_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SLPX_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds));
        return wgs <= static_cast<size_t>(per_cu) * cus;
      };
      // (the tasks of both attempts AND the riding workgroups resident at once: a task that asks for the verdict must
      // never keep a riding workgroup from being dispatched)
      const bool fits = m_twin_threads == 1024 ? resident(&ldlt_mf_twin_kernel<1024, true>, 1024) : resident(&ldlt_mf_twin_kernel<512, true>, 512);
      if (!fits) m_ride_lds_ok = 0;
      if (m_ipm_ride_verdict.n == 0) m_ipm_ride_verdict.upload(std::vector<double>(1, 0.0));
      if (std::getenv("SLPX_LDLT_VERBOSE"))
        std::fprintf(stderr, "ldlt riding error launch: %zu workgroups, %d of %d threads per CU x %d CUs%s\n", wgs, per_cu, m_twin_threads, cus,
                     fits ? "" : ": NOT resident at once");
    }
  }
  return m_ride_lds_ok != 0;
}

bool DeviceNlp::ipm_ride_wait(int slot_out) {
  volatile IpmHost* h = &m_ipm_host[slot_out];
  const double ticket = m_ride_ticket;
  double verdict = 0.0;
  // (the verdict AND every word it announces: writes to host memory may arrive in any order — IpmHost::check)
  auto handed_over = [&] {
    const double go = h->go;
    if (std::fabs(go) != ticket) return false;
    unsigned long long bits = std::bit_cast<unsigned long long>(go);
    const volatile unsigned long long* w = reinterpret_cast<const volatile unsigned long long*>(&h->err_ahead);
    for (size_t k = 0; k < sizeof(IpmErrOut) / sizeof(unsigned long long); ++k) bits ^= w[k];
    if (bits != h->check) return false;
    verdict = go;
    return true;
  };
  spin_on_published(handed_over, m_stream.raw(), "slpx: a riding error launch finished without its verdict");
  return verdict > 0.0;
}

void DeviceNlp::ipm_errors_deciding(bool sums_ride) {
  m_errors_decide = true;
  ipm_errors(false, sums_ride, /*ahead=*/true);
  m_errors_decide = false;
}

// (only the used part of the filter's table travels, either way)
static size_t ipm_ctl_bytes(const IpmCtl& c) { return offsetof(IpmCtl, filter) + offsetof(FilterState, ent) + 16u * static_cast<size_t>(c.filter.n); }
void DeviceNlp::ipm_pipeline_upload(const IpmCtl& ctl) {
  m_ipm_ctl_stage = 1 + (m_ipm_ctl_stage & 1);  // (1, 2, 1, ...: the copy of the upload before this one has long run)
  IpmCtl& stage = m_ipm_ctl_host[m_ipm_ctl_stage];
  std::memcpy(&stage, &ctl, ipm_ctl_bytes(ctl));
  SLPX_HIP_CHECK(hipMemcpyAsync(m_ipm_ctl_dev, &stage, ipm_ctl_bytes(ctl), hipMemcpyHostToDevice, m_stream));
}
const IpmCtl& DeviceNlp::ipm_pipeline_fetch() {
  IpmCtl& into = m_ipm_ctl_host[0];
  // (the head says how long the table is)
  SLPX_HIP_CHECK(hipMemcpyAsync(&into, m_ipm_ctl_dev, offsetof(IpmCtl, filter) + offsetof(FilterState, ent), hipMemcpyDeviceToHost, m_stream));
  SLPX_HIP_CHECK(hipStreamSynchronize(m_stream));
  if (into.filter.n > 0) {
    SLPX_HIP_CHECK(hipMemcpyAsync(into.filter.ent, static_cast<const char*>(m_ipm_ctl_dev) + offsetof(IpmCtl, filter) + offsetof(FilterState, ent),
                                  16u * static_cast<size_t>(into.filter.n), hipMemcpyDeviceToHost, m_stream));
    SLPX_HIP_CHECK(hipStreamSynchronize(m_stream));
  }
  return into;
}

void DeviceNlp::wait_published_until(unsigned long long seq) {
  spin_on_published([&] { return *m_h_seq >= seq; }, m_stream.raw(), "slpx: chain finished without publishing");
}

void DeviceNlp::ipm_soc_accumulate(double alpha, bool first, bool s_from_ci) {
  const int work = std::max({m_kdev.m_e, m_kdev.m_i, 1});
  hipLaunchKernelGGL(ipm_soc_accumulate_kernel, dim3(grid_for(work, 256)), dim3(256), 0, m_stream, m_kdev, m_V.p,
                     m_V_trial.p, m_s.p, m_ps.p, alpha, first ? 1 : 0, s_from_ci ? 1 : 0, m_soc_ce.p,
                     m_soc_cims.p);
  SLPX_HIP_CHECK(hipGetLastError());
}

void DeviceNlp::ipm_soc_rhs() {
  m_rhs_stale = false;
  m_rhs_in_il = false;
  hipLaunchKernelGGL(ipm_soc_rhs_kernel, dim3(grid_for(m_kdev.dim, 256)), dim3(256), 0, m_stream, m_kdev, m_V.p,
                     m_s.p, m_y.p, m_z.p, m_mu.p, m_soc_ce.p, m_soc_cims.p, m_rhs.p);
  SLPX_HIP_CHECK(hipGetLastError());
}

void DeviceNlp::ipm_soc_backsub() {
  if (m_kdev.m_i == 0) return;
  hipLaunchKernelGGL(ipm_soc_backsub_kernel, dim3(grid_for(m_kdev.m_i, 256)), dim3(256), 0, m_stream, m_kdev,
                     m_V.p, m_p.p, m_s.p, m_z.p, m_mu.p, m_soc_cims.p, m_ps.p, m_pz.p);
  SLPX_HIP_CHECK(hipGetLastError());
}

void DeviceNlp::ipm_save_direction() {
  hipLaunchKernelGGL(ipm_copy_direction_kernel, dim3(grid_for(m_kdev.dim, 256)), dim3(256), 0, m_stream, m_kdev.dim, m_kdev.m_i,
                     m_p.p, m_ps.p, m_pz.p, m_p_keep.p, m_ps_keep.p, m_pz_keep.p);
  SLPX_HIP_CHECK(hipGetLastError());
}

void DeviceNlp::ipm_restore_direction() {
  hipLaunchKernelGGL(ipm_copy_direction_kernel, dim3(grid_for(m_kdev.dim, 256)), dim3(256), 0, m_stream, m_kdev.dim, m_kdev.m_i,
                     m_p_keep.p, m_ps_keep.p, m_pz_keep.p, m_p.p, m_ps.p, m_pz.p);
  SLPX_HIP_CHECK(hipGetLastError());
}

}  // namespace slpx
