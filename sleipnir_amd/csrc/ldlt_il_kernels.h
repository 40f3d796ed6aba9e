// Batch-interleaved LDLᵀ: the lanes of a wave are PROBLEMS, not entries.
//
// All problems of a batch share one symbolic plan and differ only in values, so the lanes of
// a wavefront can run the SAME task of many problems in lockstep: every index of the plan
// (entry sources, pair lists, contribution lists, solve items) is read once per wave instead
// of once per problem, a 16-lane group never diverges, and every value access is a whole
// 128-byte line: problem b lives in element b % 16 of row [b / 16][i][0..15] of the
// interleaved arrays  lhs_il, rhs_il, Lx_il, D_il, contrib_il, scontrib_il, zv_il, xg_il.
//
//  * factorization: a workgroup = one task x 16 problems x THIRTY-TWO entries of a level at a
//    time (thread = entry slot * 16 + problem).  The task's U values and 1/d of its columns sit in
//    LDS as rows of 16 — a quarter of what 64 problems per wave would need, so a dozen
//    waves share a CU and hide each other's LDS latency (measured with 64 problems per wave
//    and one entry at a time: one wave per CU, 3.9 ms per factorization of 512 x N=1000 —
//    slower than the per-task kernels);
//  * triangular solves: a wave = one task x 64 problems, one lane each (x of the task's
//    columns in LDS), streaming L once.
// The symbolic phase is run with small tasks in this mode (newton.cpp).  The per-task
// kernels of ldlt_kernels.h (one workgroup per task and problem, eight lanes per entry)
// remain the path of a single problem, where latency rules; this is the throughput path
// (DeviceNlp::interleaved_for: batches of about 200 problems and more).
//
// Arithmetic per entry is the same left-looking gather
//   U(i,j) = A(i,j) [+delta | -gamma] - sum(update blocks of child tasks) - sum_k U(i,k) U(j,k) / d_k
// (sparse_regularized_ldlt.hpp:64-152 drives it; Eigen's SimplicialLDLT computes the same
// quantities up-looking), accumulated in pair-list order with four interleaved partial sums.
#pragma once

#include <hip/hip_runtime.h>

#include "coherent.h"
#include "device.hpp"
#include "ldlt_kernels.h"  // round_wait, round_signal

namespace slpx {

// phase clocks of the first task of the selected round, first problem group (slots [8,16) of
// g_ldlt_clocks, which the per-task forward kernel does not use inside a Newton step)
#define SLPX_IL_CLOCK(k)                                                                      \
  if (task_index == L.clock_task && g == 0 && threadIdx.x == 0) \
  g_ldlt_clocks[8 + (k)] = wall_clock64()

constexpr int kIlLanes = 64;  // lanes of a wave
constexpr int kIlWShift = 4;  // (measured: 8-wide rows 0.74 ms per factorization of 512 x N=1000, 16-wide 0.67)
constexpr int kIlW = 1 << kIlWShift;  // problems per interleaved row
constexpr int kIlRowsPerChunk = 64 / kIlW;  // rows groups covering a 64-problem chunk
// a level has ~65-120 entries: with 32 of them in flight a task's level loop takes half the passes
// of 16, and the same LDS per workgroup holds twice the waves to hide its latency (measured, ms per
// factorization of 512 x N=1000: 256 threads 0.629, 512 threads 0.589, 1024 threads 0.809)
constexpr int kIlFactorThreads = 512;
constexpr int kIlSlots = kIlFactorThreads / kIlW;  // entries of a level a factorization workgroup works on at a time

// number of 16-problem rows groups a batch occupies (padded to whole waves of 64 problems)
__host__ __device__ inline int il_groups(int batch) { return kIlRowsPerChunk * ((batch + 63) / 64); }

// [b][i] (stride `stride` doubles per problem) -> [b / 16][i][16]; problems beyond the batch
// get 0.  64 x 64 tiles through LDS so that both sides move whole cache lines.
__global__ __launch_bounds__(256) void il_gather_kernel(const double* __restrict__ src, long long stride,
                                                        int count, double* __restrict__ dst, int batch) {
  __shared__ double tile[64][65];
  const int c = blockIdx.y, i0 = blockIdx.x * 64;          // c: wave-sized chunk of 64 problems
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // ty: 0..3
  for (int r = ty; r < 64; r += 4) {                       // r = problem within the chunk
    const int b = c * 64 + r, i = i0 + tx;
    tile[r][tx] = (b < batch && i < count) ? src[static_cast<size_t>(b) * stride + i] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {  // r = entry within the tile, tx = problem within the chunk
    const int i = i0 + r;
    if (i < count)
      dst[(static_cast<size_t>(c * kIlRowsPerChunk + (tx >> kIlWShift)) * count + i) * kIlW + (tx & (kIlW - 1))] = tile[tx][r];
  }
}

// [b / 16][i][16] -> [b][i]
__global__ __launch_bounds__(256) void il_scatter_kernel(const double* __restrict__ src, int count,
                                                         double* __restrict__ dst, long long stride, int batch) {
  __shared__ double tile[64][65];
  const int c = blockIdx.y, i0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int i = i0 + r;
    tile[r][tx] = i < count ? src[(static_cast<size_t>(c * kIlRowsPerChunk + (tx >> kIlWShift)) * count + i) * kIlW + (tx & (kIlW - 1))] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int b = c * 64 + r, i = i0 + tx;
    if (b < batch && i < count) dst[static_cast<size_t>(b) * stride + i] = tile[tx][r];
  }
}

__device__ __forceinline__ double il_reciprocal(double d) {
  const double r0 = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r0, 1.0);
  double r = __builtin_fma(r0, e, r0);
  e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  return (r0 != 0.0 && isfinite(r0)) ? r : r0;
}

// One workgroup = one task x 16 problems; thread = slot * 16 + problem, the kIlSlots slots work
// on that many entries of a level at a time (eight waves per workgroup: sixteen or more waves
// share a CU's LDS and hide each other's latency).
// LDS: U[n_ent][16] | invd[n_col][16] | the task's plan slices (pairs, pointers, sources, ...).
__global__ __launch_bounds__(kIlFactorThreads) void ldlt_factor_il_kernel(
    LdltDev L, uint32_t task_base, const double* __restrict__ lhs_il, int nnz_lhs,
    const double* __restrict__ rhs_il, int n, const double* __restrict__ reg, double* __restrict__ Lx_il,
    long long nnzL, double* __restrict__ D_il, double* __restrict__ contrib_il, int n_contrib,
    double* __restrict__ zv_il, LdltStats* __restrict__ stats_part, int batch,
    const uint32_t* __restrict__ il_meta, const uint32_t* __restrict__ il_meta_off) {
  extern __shared__ __attribute__((aligned(16))) double il_smem[];
  // (every round of a small batch in ONE launch, round counters per group of problems, was measured equal to
  // slower — profiles/r04_il_single_probe.txt: the ~20 us of a round are inside it — and removed)
  const uint32_t task_index = task_base + blockIdx.x;
  const int g = blockIdx.y;  // g: group of 16 problems
  const LdltTask t = L.tasks[task_index];
  const int lane = threadIdx.x;
  const int slot = lane >> kIlWShift, pl = lane & (kIlW - 1);  // slot: 0 .. kIlSlots - 1
  const int b = g * kIlW + pl;
  const bool in_batch = b < batch;
  const double delta = in_batch ? reg[2 * b] : 0.0, gamma = in_batch ? reg[2 * b + 1] : 0.0;
  const bool active = in_batch && delta == delta;  // NaN delta: not part of this attempt

  const double* lhs = lhs_il + static_cast<size_t>(g) * nnz_lhs * kIlW + pl;
  const double* rhs = rhs_il + static_cast<size_t>(g) * n * kIlW + pl;
  double* Lx = Lx_il + static_cast<size_t>(g) * nnzL * kIlW + pl;
  double* D = D_il + static_cast<size_t>(g) * n * kIlW + pl;
  double* zv = zv_il + static_cast<size_t>(g) * n * kIlW + pl;
  double* contrib = contrib_il + static_cast<size_t>(g) * n_contrib * kIlW + pl;
  double* U = il_smem + pl;
  double* invd = il_smem + static_cast<size_t>(t.n_ent) * kIlW + pl;

  // The plan slices of the task go to LDS first: the loops below are chains of dependent
  // reads, and a read from global memory in them costs a full memory round trip each time.
  // The host packed them back to back (DeviceNlp: il_meta), so this is ONE copy loop with
  // four loads per lane in flight instead of nine loops with a round trip each.
  SLPX_IL_CLOCK(0);
  const uint32_t* gm = il_meta + il_meta_off[task_index];
  const uint32_t n_words = gm[1];  // gm[0]: number of update-block refs (informational)
  uint32_t* meta = reinterpret_cast<uint32_t*>(il_smem + static_cast<size_t>(t.n_ent + t.n_col) * kIlW);
  {
    const uint32_t* src_w = gm + 2;
    uint32_t i = lane;
    for (; i + 3 * kIlFactorThreads < n_words; i += 4 * kIlFactorThreads) {
      const uint32_t w0 = src_w[i], w1 = src_w[i + kIlFactorThreads], w2 = src_w[i + 2 * kIlFactorThreads],
                     w3 = src_w[i + 3 * kIlFactorThreads];
      meta[i] = w0;
      meta[i + kIlFactorThreads] = w1;
      meta[i + 2 * kIlFactorThreads] = w2;
      meta[i + 3 * kIlFactorThreads] = w3;
    }
    for (; i < n_words; i += kIlFactorThreads) meta[i] = src_w[i];
  }
  const uint32_t n_pp = t.n_ent + t.n_ext + 1;
  const uint2* s_pairs = reinterpret_cast<const uint2*>(meta);  // 8-byte aligned: offset multiple of 128 B
  const uint32_t* s_pptr = meta + 2 * t.n_pairs;
  const int32_t* s_src = reinterpret_cast<const int32_t*>(s_pptr + n_pp);
  const uint32_t* s_out = reinterpret_cast<const uint32_t*>(s_src + t.n_ent);
  const uint32_t* s_fc = s_out + t.n_ent;  // flags | local column << 8
  const uint32_t* s_ext = s_fc + t.n_ent;
  const uint32_t* s_lvl = s_ext + t.n_ext;
  const uint32_t* s_crptr = s_lvl + t.n_lvl + 1;  // kIlSlots + 1 offsets: the update-block refs of each slot
  const uint32_t* s_cref = s_crptr + kIlSlots + 1;           // (target entry, block slot) per ref
  __syncthreads();
  SLPX_IL_CLOCK(1);
  const uint2* pairs = s_pairs;  // x = a | b << 16, y = k

  auto term = [&](const uint2 pr) {
    return (U[(pr.x & 0xffffu) * kIlW] * invd[(pr.y & 0xffffu) * kIlW]) * U[(pr.x >> 16) * kIlW];
  };
  // Four pairs per trip, the last trip padded with masked repeats of the final pair: an
  // entry has 3-4 pairs on average, and one pair per trip would put a full LDS round trip
  // (pair record -> operands) on the wave's critical path for each of them.
  auto pair_sum = [&](uint32_t pb, uint32_t pe) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (uint32_t q = pb; q < pe; q += 4) {
      const uint32_t last = pe - 1;
      const uint32_t q1 = q + 1 < pe ? q + 1 : last, q2 = q + 2 < pe ? q + 2 : last, q3 = q + 3 < pe ? q + 3 : last;
      const uint2 p0 = pairs[q], p1 = pairs[q1], p2 = pairs[q2], p3 = pairs[q3];
      const double t0 = term(p0), t1 = term(p1), t2 = term(p2), t3 = term(p3);
      s0 += t0;
      s1 += q + 1 < pe ? t1 : 0.0;
      s2 += q + 2 < pe ? t2 : 0.0;
      s3 += q + 3 < pe ? t3 : 0.0;
    }
    return (s0 + s1) + (s2 + s3);
  };

  // Pass 1 — everything that comes from global memory: U = A [+delta | -gamma] - update
  // blocks of the child tasks (earlier launches).  Nothing here depends on anything else, so
  // it is written for memory-level parallelism: slot s owns the entries e = s (mod kIlSlots),
  // four matrix loads in flight, then eight update-block loads in flight from the slot's flat
  // (entry, block) list — in a per-entry loop every block would cost a memory round trip on
  // the wave's critical path (measured: 10-106 us of a 28-119 us task).
  auto matrix_value = [&](uint32_t e) {
    const int32_t s0 = s_src[e];
    const uint32_t fl = s_fc[e] & 0xffu;
    // unconditional load (entry 0 where there is no source): a branch around it would put the
    // four loads of a trip one after the other instead of in flight together
    const double* from = (fl & 4) ? rhs : lhs;
    const double v = from[static_cast<size_t>(s0 >= 0 ? s0 : 0) * kIlW];
    double acc = s0 >= 0 ? v : 0.0;
    if (fl & 1) acc += (fl & 2) ? -gamma : delta;
    return acc;
  };
  {
    uint32_t e = slot;
    for (; e + 3 * kIlSlots < t.n_ent; e += 4 * kIlSlots) {
      double a[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = matrix_value(e + kIlSlots * j);
#pragma unroll
      for (int j = 0; j < 4; ++j) U[(e + kIlSlots * j) * kIlW] = a[j];
    }
    for (; e < t.n_ent; e += kIlSlots) U[e * kIlW] = matrix_value(e);
  }
  {
    // (a slot only touches its own entries: no cross-slot ordering needed before this)
    uint32_t r = s_crptr[slot];
    const uint32_t re = s_crptr[slot + 1];
    for (; r + 7 < re; r += 8) {
      double v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = contrib[static_cast<size_t>(s_cref[2 * (r + j) + 1]) * kIlW];
#pragma unroll
      for (int j = 0; j < 8; ++j) U[s_cref[2 * (r + j)] * kIlW] -= v[j];  // same entry: in list order
    }
    for (; r < re; ++r) U[s_cref[2 * r] * kIlW] -= contrib[static_cast<size_t>(s_cref[2 * r + 1]) * kIlW];
  }
  __syncthreads();  // (one wave: orders the LDS traffic of the four slots)
  SLPX_IL_CLOCK(2);

  // Pass 2 — the elimination, level by level, LDS only
  int n_pos = 0, n_neg = 0, n_zero = 0, n_bad = 0;
  double min_abs = __longlong_as_double(0x7ff0000000000000ll);
  for (uint32_t l = 0; l < t.n_lvl; ++l) {
    const uint32_t end = s_lvl[l + 1];
    for (uint32_t e = s_lvl[l] + slot; e < end; e += kIlSlots) {
      const uint32_t fc = s_fc[e];
      const double acc = U[e * kIlW] - pair_sum(s_pptr[e], s_pptr[e + 1]);
      U[e * kIlW] = acc;
      if (fc & 1) {
        invd[(fc >> 8) * kIlW] = il_reciprocal(acc);
        if (active) D[static_cast<size_t>(s_out[e]) * kIlW] = acc;
        const double eps = 2.220446049250313e-16;  // inertia.hpp:40-50
        if (acc > eps) ++n_pos;
        else if (acc < -eps) ++n_neg;
        else ++n_zero;
        if (acc == 0.0 || !isfinite(acc)) ++n_bad;
        else min_abs = fmin(min_abs, fabs(acc));
      }
    }
    __syncthreads();
  }
  SLPX_IL_CLOCK(3);
  // update blocks for ancestor tasks (later rounds = later launches); lanes outside the
  // attempt keep what an earlier, accepted attempt left for their problem
  for (uint32_t x = slot; x < t.n_ext; x += kIlSlots) {
    const double v = pair_sum(s_pptr[t.n_ent + x], s_pptr[t.n_ent + x + 1]);
    if (active) contrib[static_cast<size_t>(s_ext[x]) * kIlW] = v;
  }
  SLPX_IL_CLOCK(4);
  // results: L = U / d, and z = D⁻¹L⁻¹Pb from the right-hand-side row
  for (uint32_t e = slot; e < t.n_ent; e += kIlSlots) {
    const uint32_t fc = s_fc[e];
    if ((fc & 1) || !active) continue;
    const double v = U[e * kIlW] * invd[(fc >> 8) * kIlW];
    if (fc & 4) zv[static_cast<size_t>(s_out[e]) * kIlW] = v;
    else Lx[static_cast<size_t>(s_out[e]) * kIlW] = v;
  }
  SLPX_IL_CLOCK(5);
  // inertia counters of the task: fold the slots of each problem through LDS (U is dead now)
  __syncthreads();
  {
    int* ci = reinterpret_cast<int*>(il_smem);                    // [4][slot][16]
    double* cm = il_smem + 2 * kIlSlots * kIlW;                   // [slot][16], after 4 int rows
    ci[(0 * kIlSlots + slot) * kIlW + pl] = n_pos;
    ci[(1 * kIlSlots + slot) * kIlW + pl] = n_neg;
    ci[(2 * kIlSlots + slot) * kIlW + pl] = n_zero;
    ci[(3 * kIlSlots + slot) * kIlW + pl] = n_bad;
    cm[slot * kIlW + pl] = min_abs;
    __syncthreads();
    if (slot == 0)
      for (int sl = 1; sl < kIlSlots; ++sl) {
        n_pos += ci[(0 * kIlSlots + sl) * kIlW + pl];
        n_neg += ci[(1 * kIlSlots + sl) * kIlW + pl];
        n_zero += ci[(2 * kIlSlots + sl) * kIlW + pl];
        n_bad += ci[(3 * kIlSlots + sl) * kIlW + pl];
        min_abs = fmin(min_abs, cm[sl * kIlW + pl]);
      }
  }
  if (active && slot == 0) {
    LdltStats st;
    st.n_pos = n_pos;
    st.n_neg = n_neg;
    st.n_zero = n_zero;
    st.n_bad = n_bad;
    st.min_abs_bits = static_cast<unsigned long long>(__double_as_longlong(min_abs));
    stats_part[static_cast<size_t>(task_index) * batch + b] = st;
  }
}

// stats[b] = fold of the per-task partial counters (problems outside the attempt untouched);
// one 64-lane workgroup per problem, lanes stride over the tasks.
__global__ __launch_bounds__(64) void ldlt_stats_il_kernel(const LdltStats* __restrict__ stats_part,
                                                           int n_tasks, const double* __restrict__ reg,
                                                           LdltStats* __restrict__ stats, int batch) {
  const int b = blockIdx.x;
  const double delta = reg[2 * b];
  if (delta != delta) return;
  int n_pos = 0, n_neg = 0, n_zero = 0, n_bad = 0;
  unsigned long long mn = 0x7ff0000000000000ull;
  for (int tk = threadIdx.x; tk < n_tasks; tk += 64) {
    const LdltStats s = stats_part[static_cast<size_t>(tk) * batch + b];
    n_pos += s.n_pos;
    n_neg += s.n_neg;
    n_zero += s.n_zero;
    n_bad += s.n_bad;
    mn = s.min_abs_bits < mn ? s.min_abs_bits : mn;
  }
  for (int off = 32; off > 0; off >>= 1) {
    n_pos += __shfl_xor(n_pos, off);
    n_neg += __shfl_xor(n_neg, off);
    n_zero += __shfl_xor(n_zero, off);
    n_bad += __shfl_xor(n_bad, off);
    const unsigned long long o = __shfl_xor(mn, off);
    mn = o < mn ? o : mn;
  }
  if (threadIdx.x == 0) stats[b] = LdltStats{n_pos, n_neg, n_zero, n_bad, mn};
}

// Forward substitution L y = P b, z = D⁻¹ y for a right-hand side that arrived after the
// factorization.  LDS: y[n_col][64].
__global__ __launch_bounds__(kIlLanes) void ldlt_fwd_il_kernel(
    LdltDev L, uint32_t task_base, const double* __restrict__ rhs_il, int n, const double* __restrict__ Lx_il,
    long long nnzL, const double* __restrict__ D_il, double* __restrict__ scontrib_il, int n_scontrib,
    double* __restrict__ zv_il) {
  extern __shared__ __attribute__((aligned(16))) double il_smem[];
  const LdltTask t = L.tasks[task_base + blockIdx.x];
  const int c = blockIdx.y, lane = threadIdx.x;  // c: chunk of 64 problems = four rows groups
  const size_t g = static_cast<size_t>(c) * kIlRowsPerChunk + (lane >> kIlWShift);
  const int pl = lane & (kIlW - 1);
  const double* rhs = rhs_il + g * n * kIlW + pl;
  const double* Lx = Lx_il + g * nnzL * kIlW + pl;
  const double* D = D_il + g * n * kIlW + pl;
  double* zv = zv_il + g * n * kIlW + pl;
  double* scontrib = scontrib_il + g * n_scontrib * kIlW + pl;
  double* y = il_smem + lane;
  const uint32_t* colperm = L.col_perm + t.col_off;
  const uint32_t* ptr = L.fwd_ptr + t.colptr_off;
  const uint32_t* fcptr = L.fwd_contrib_ptr + t.colptr_off;
  const uint32_t* scidx = L.scontrib_idx + t.scontrib_off;
  const LdltSolveItem* items = L.fwd_items + t.fwd_item_off;
  // rows in level order: a row's items only reference earlier local columns
  for (uint32_t i = 0; i < t.n_col; ++i) {
    double acc = rhs[static_cast<size_t>(L.perm[colperm[i]]) * kIlW];
    for (uint32_t k = fcptr[i]; k < fcptr[i + 1]; ++k) acc -= scontrib[static_cast<size_t>(scidx[k]) * kIlW];
    for (uint32_t q = ptr[i]; q < ptr[i + 1]; ++q)
      acc -= Lx[static_cast<size_t>(items[q].lpos) * kIlW] * y[items[q].ref * kIlLanes];
    y[i * kIlLanes] = acc;
  }
  const uint32_t* sptr = L.sext_ptr + t.sext_ptr_off;
  const LdltSolveItem* sitems = L.sext_items + t.sext_item_off;
  for (uint32_t x = 0; x < t.n_sext; ++x) {
    double acc = 0.0;
    for (uint32_t q = sptr[x]; q < sptr[x + 1]; ++q)
      acc += Lx[static_cast<size_t>(sitems[q].lpos) * kIlW] * y[sitems[q].ref * kIlLanes];
    scontrib[static_cast<size_t>(L.sext_dst[t.sext_off + x]) * kIlW] = acc;
  }
  for (uint32_t i = 0; i < t.n_col; ++i) {
    const uint32_t pj = colperm[i];
    zv[static_cast<size_t>(pj) * kIlW] = y[i * kIlLanes] / D[static_cast<size_t>(pj) * kIlW];
  }
}

// Backward substitution Lᵀ x = z; x of the task's columns goes to xg_il (for descendants,
// later launches) and un-permuted to the batch-major solution.
// A workgroup = one task x 64 problems x kIlBwdWaves waves: a lane is a problem, and the COLUMNS OF A
// LEVEL — independent of each other: a column's items reference later levels of the task or rows
// of ancestor tasks — are dealt to the waves, a barrier per level.  (r02: one wave per task walked the
// columns one after the other — one wave per CU in the upper rounds, a chain of ~100 columns with a
// trip to memory for each column's L values; 512 x N=1000: 0.25 ms, 64 x N=500: 0.13 ms.)
// LDS: x[n_col][64] | bwd_ptr[n_col + 1] | col_perm[n_col] | column levels[n_lvl + 1] | bwd_items[n_bwd_items].
// (measured on one box, ms per backward solve at 64 x N=500 / 512 x N=1000: 2 waves 0.101 / 0.229, 4 waves
// 0.090 / 0.218, 8 waves 0.088 / 0.226; the next column's L values requested one column — and one barrier —
// ahead, as the one-wave kernel did: 0.109 / 0.244, the registers cost more than the trip)
constexpr int kIlBwdWaves = 4;
__global__ __launch_bounds__(kIlLanes* kIlBwdWaves) void ldlt_bwd_il_kernel(
    LdltDev L, uint32_t task_base, int n, const double* __restrict__ Lx_il, long long nnzL,
    const double* __restrict__ zv_il, double* __restrict__ xg_il, double* __restrict__ out, int batch) {
  extern __shared__ __attribute__((aligned(16))) double il_smem[];
  const uint32_t task_index = task_base + blockIdx.x;
  const int c = blockIdx.y;
  const LdltTask t = L.tasks[task_index];
  const int tid = threadIdx.x, lane = tid & (kIlLanes - 1), wave = tid >> 6;
  const int b = c * kIlLanes + lane;
  const size_t g = static_cast<size_t>(c) * kIlRowsPerChunk + (lane >> kIlWShift);
  const int pl = lane & (kIlW - 1);
  const double* Lx = Lx_il + g * nnzL * kIlW + pl;
  const double* zv = zv_il + g * n * kIlW + pl;
  double* xg = xg_il + g * n * kIlW + pl;
  double* x = il_smem + lane;
  uint32_t* s_ptr = reinterpret_cast<uint32_t*>(il_smem + static_cast<size_t>(t.n_col) * kIlLanes);
  uint32_t* s_colperm = s_ptr + t.n_col + 1;
  uint32_t* s_lvl = s_colperm + t.n_col;
  uint32_t* s_colsn = s_lvl + t.n_lvl + 1;  // position in its chain | width << 8 (LdltPlan::col_sn)
  // (8-byte aligned: the x rows are a multiple of 512 bytes; the four word arrays together 3 n_col + n_lvl + 2 words)
  LdltSolveItem* s_items = reinterpret_cast<LdltSolveItem*>(s_colsn + t.n_col + ((3 * t.n_col + t.n_lvl + 2) & 1u));
  {
    const uint32_t* g_ptr = L.bwd_ptr + t.colptr_off;
    const uint32_t* g_colperm = L.col_perm + t.col_off;
    const uint32_t* g_lvl = L.col_lvl_ptr + t.lvl_off;
    const LdltSolveItem* g_items = L.bwd_items + t.bwd_item_off;
    constexpr int kThreads = kIlLanes * kIlBwdWaves;
    for (uint32_t k = tid; k <= t.n_col; k += kThreads) s_ptr[k] = g_ptr[k];
    for (uint32_t k = tid; k < t.n_col; k += kThreads) s_colperm[k] = g_colperm[k];
    for (uint32_t k = tid; k <= t.n_lvl; k += kThreads) s_lvl[k] = g_lvl[k];
    for (uint32_t k = tid; k < t.n_col; k += kThreads) s_colsn[k] = L.col_sn[t.col_off + k];
    for (uint32_t k = tid; k < t.n_bwd_items; k += kThreads) s_items[k] = g_items[k];
  }
  __syncthreads();
  // a column's items reference later levels' columns or rows of ancestor tasks (bit 31: global permuted
  // row, final since an earlier launch)
  auto operand = [&](const LdltSolveItem it) {
    return (it.ref & 0x80000000u) ? xg[static_cast<size_t>(it.ref & 0x7fffffffu) * kIlW] : x[it.ref * kIlLanes];
  };
  // eight items of a column from q on (masked beyond qe): the L values, and the records for the operands
  auto request = [&](uint32_t q, uint32_t qe, double (&lv)[8], LdltSolveItem (&its)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool live = q + j < qe;
      its[j] = s_items[live ? q + j : (qe > 0 ? qe - 1 : 0)];
      lv[j] = live ? Lx[static_cast<size_t>(its[j].lpos) * kIlW] : 0.0;
    }
  };
  for (int l = static_cast<int>(t.n_lvl) - 1; l >= 0; --l) {
    const int lo = static_cast<int>(s_lvl[l]);
    // the columns of a level are dealt to the waves — by CHAIN: the columns of a supernode (consecutive,
    // one level) depend on each other top-down, so the wave that owns the chain's first column walks all of
    // them, last first; its own stores to x are in order for it (lone columns: a chain of one)
    for (int i = static_cast<int>(s_lvl[l + 1]) - 1; i >= lo; --i) {
      if (((i - static_cast<int>(s_colsn[i] & 0xffu)) % kIlBwdWaves) != wave) continue;
      double acc = zv[static_cast<size_t>(s_colperm[i]) * kIlW];
      uint32_t q = s_ptr[i];
      const uint32_t qe = s_ptr[i + 1];
      // eight at a time, two partial sums each (the order the sums have always had)
      for (; q < qe; q += 8) {
        double lv[8];
        LdltSolveItem its[8];
        request(q, qe, lv, its);
        double xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = q + j < qe ? operand(its[j]) : 0.0;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          s0 += lv[j] * xv[j];
          s1 += lv[j + 1] * xv[j + 1];
        }
        acc -= s0 + s1;
      }
      x[i * kIlLanes] = acc;
    }
    __syncthreads();
  }
  for (uint32_t i = wave; i < t.n_col; i += kIlBwdWaves) {
    const uint32_t pj = s_colperm[i];
    const double v = x[i * kIlLanes];
    xg[static_cast<size_t>(pj) * kIlW] = v;
    if (b < batch) out[static_cast<size_t>(b) * n + L.perm[pj]] = v;
  }
}

}  // namespace slpx
