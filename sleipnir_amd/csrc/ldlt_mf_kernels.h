// Multifrontal numeric phase of the task-parallel LDLᵀ (ldlt_kernels.h): the per-level work of a
// task as DENSE FRONTS, one wave per front, everything a lane touches found through precomputed
// 16-bit LDS byte offsets (ldlt_symbolic.hpp: LdltFront) — and the Newton step built on it:
// KKT evaluation, factorization, backward solve and back-substitution in ONE launch
// (ldlt_mf_step_kernel).
//
// Why fronts.  A lone wave issues a dependent instruction every 6-9 clocks whatever it is
// (profiles/microbench), so a level costs what it EXECUTES, not what it waits for.  The left-looking
// pair lists executed, per level, a pointer trip, a pair trip, three value loads and two multiplies
// per pair (~10 pairs per entry, LDS-throughput bound in the upper rounds) + an 8-lane DPP reduction
// + a barrier + the chain pass + a second barrier.  A front J (w pivot columns, structure R below
// them, r = |R|, rows = [columns | R | right-hand-side row], nr = w + r + 1):
//   1. its first w columns (the LdltSn trapezoid in the task's entry array U) = A + the entries of
//      the child fronts' update blocks that fall into them: at most `nch` values per entry, their
//      addresses read from the front's pivot table;
//   2. w pivots in registers, a lane per row (v_readlane broadcasts, as sn_finish_wave);
//   3. its update block S_J(a, b) = Σ_children S_K(..) − Σ_c L(R_a, c) U(R_b, c), a in R + rhs, b in R,
//      b <= a — a lane per entry, addresses from the front's update table — stored for the parent
//      front, or, when the parent column lies in another task, sent straight to that task's
//      update slots;
// one barrier per level.  The backward solve walks the same fronts top-down on the U and 1/d the
// factorization left in LDS: eight lanes per pivot column over the rows of R, a DPP reduction,
// the chain's own couplings by v_readlane.
//
// Fronts wide enough for the matrix cores (w >= 4 pivots under more than 16 rows: the g-fold
// problem's separator chains) take step 3 as v_mfma_f64_16x16x4_f64 tiles (mf_update_mfma).
//
// Replaces Eigen::SimplicialLDLT::factorize / solve as used by
// util/sparse_regularized_ldlt.hpp:74,105,159-161 and Inertia (inertia.hpp:40-50).
#pragma once

#include <hip/hip_runtime.h>

#include "ldlt_kernels.h"
#include "ipm_kernels.h"

namespace slpx {

using LdsU16 = __attribute__((address_space(3))) uint16_t;
using f64x4 = __attribute__((ext_vector_type(4))) double;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

struct MfDev {
  const LdltMfTask* tasks = nullptr;
  const LdltFront* fronts = nullptr;
  const uint4* image = nullptr;       // per task: its static LDS content in LDS order (DeviceNlp::build_mf), one fixed-size slot each
  uint32_t image_stride16 = 0;        // slot size in 16-byte groups
  const uint4* image_desc = nullptr;  // per task {first group, groups up to the end of the KKT terms, groups of back-substitution rows, terms groups}
  unsigned int n_tasks = 0;
  unsigned int* exit_cnt = nullptr;  // workgroups through their exit phase (the last one publishes)
  // chained steps (DeviceNlp::sweep_full_for_step): the AD sweep of this step runs on another stream; the
  // kernel stages its plan, then waits until chain[16], the count of sweep workgroups through, reaches wait_step
  // (0: the sweep came before in this stream), and its workgroups count themselves out in chain[48] for the next
  // step's sweep
  unsigned int* chain = nullptr;
  unsigned int wait_step = 0, this_step = 0, n_workgroups = 0;
  // the attempt's regularization BY VALUE (the kernel's `reg` argument is then null): read from the pinned host
  // array the batch kernels share, delta and gamma were a trip over PCIe at the top of every workgroup, waited for
  // with the task's descriptors (scalar loads return together)
  double delta = 0.0, gamma = 0.0;
  // A step enqueued BEFORE the iteration in front of it was decided (ipm.cpp: the pipelined common iteration): the launch
  // that decides (ipm_error_partial_kernel's fold) leaves 1.0 here for "run" and 0.0 for "pass" — every workgroup then
  // leaves at once, the first one handing the host a marker in place of the counters.  null: an ordinary launch.
  const double* gate = nullptr;
  // A step that carries the error launch of the iteration BEFORE it (ldlt_mf_twin_kernel<.., true>: the workgroups of
  // ipm_error_block ride in front of the tasks): whether the step COUNTS is decided by those workgroups while the
  // tasks factor.  Their deciding wave leaves +ride_ticket (counts) or -ride_ticket (does not) in *ride_verdict; every
  // task asks before it writes anything that outlives the launch — L, D, z, the counters, p, p_s, p_z — and writes
  // none of it if the step does not count (its hand-overs to other tasks go on as always: they are the launch's own).
  const double* ride_verdict = nullptr;
  double ride_ticket = 0.0;
};

// n_bad of the marker a passed launch publishes (nothing was factored; the host restores its launch bookkeeping)
constexpr int32_t kLdltPassed = 1 << 29;

// the sweep this step reads is complete (one lane asks; the workgroup's other waves come through the barrier):
// every workgroup of the chained sweeps so far has counted itself out — Mf.wait_step of them, the host's running
// total, below 2^29; bits 30, 31 of the word: one of them gave up waiting for the step kernel before it (V may have
// been overwritten under that kernel)
__device__ __forceinline__ void mf_wait_for_sweep(const MfDev& Mf, LdltStats* stats) {
  if (Mf.wait_step != 0u) {
    if (threadIdx.x == 0) {
      unsigned int spins = 0, seen;
      while (((seen = __hip_atomic_load(Mf.chain + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0x3fffffffu) < Mf.wait_step) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) {  // never expected: mark the factorization bad instead of hanging
          seen = 0x80000000u;
          break;
        }
      }
      // kLdltChainFailure (device.hpp) in n_bad — bit 30, above anything the counts can reach: the host
      // (NewtonSystem::compute_impl) redoes the step with the chain off
      if (seen >> 30) atomicOr(&stats[0].n_bad, kLdltChainFailure);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // V as the sweep's workgroups left it, not as this XCD's L2 remembers it
    }
    __syncthreads();
  }
}
// this workgroup has read everything it will of V, s, z: counted out, not waited for (the next step's sweep asks for
// the total)
// (the bare barrier: every wave is past its last READ of V — the values were used — and that is all the sweep behind
// this kernel needs; __syncthreads() would also wait for the acknowledgements of the results just stored, a
// microsecond at the end of every chained step)
__device__ __forceinline__ void mf_signal_done(const MfDev& Mf) {
  if (Mf.chain != nullptr) {
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(Mf.chain + 48, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// LDS by byte address (what the tables hold)
__device__ __forceinline__ double lds_ld(uint32_t addr) { return *reinterpret_cast<const LdsF64*>(static_cast<uintptr_t>(addr)); }
__device__ __forceinline__ void lds_st(uint32_t addr, double v) { *reinterpret_cast<LdsF64*>(static_cast<uintptr_t>(addr)) = v; }
__device__ __forceinline__ uint32_t lds_ld16(uint32_t addr) { return *reinterpret_cast<const LdsU16*>(static_cast<uintptr_t>(addr)); }

// A front's descriptor (16 bytes, wave-uniform) from the task's image in LDS into scalar registers: one 16-byte
// read, four v_readfirstlane — the vector registers are free again at once (held across a front's body they were
// spilled: the kernel sits near the 128 the 1024-thread workgroup allows).  The descriptors used to come through
// the scalar cache from the plan in memory, load and wait in one asm statement at the end of every level: a miss
// was a trip to L2 or beyond in front of the level's barrier — by the front microbenchmark a level of the kernel
// cost 400 clocks more than its widest front in the factorization and 300-400 more in the backward solve,
// 28 levels a step.  A wave asks for the descriptor of its front in the NEXT level at the end of this level's work.
__device__ __forceinline__ u32x4 lds_load_desc(uint32_t addr) {
  using LdsU32x4 = __attribute__((address_space(3))) u32x4;
  const u32x4 v = *reinterpret_cast<const LdsU32x4*>(static_cast<uintptr_t>(addr));
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane(v[0]);
  r[1] = __builtin_amdgcn_readfirstlane(v[1]);
  r[2] = __builtin_amdgcn_readfirstlane(v[2]);
  r[3] = __builtin_amdgcn_readfirstlane(v[3]);
  return r;
}

// column c against column 0 of a trapezoid, same row, in bytes
__device__ __forceinline__ uint32_t mf_coff(uint32_t c, uint32_t nr) { return 8u * (c * nr - (c * (c - 1u)) / 2u - c); }

// ---------------------------------------------------------------------------
// One front, one wave.  `tab`: LDS byte address of the front's tables.  Wave-uniform: everything
// but `lane`.  What bounds a front is the number of instructions its wave executes and the number
// of DEPENDENT trips to LDS, so the code is laid out in three trips: (1) every table word the lane
// will need, (2) every value that does not depend on this front's elimination — its own entries,
// the children's values for the pivot columns AND for the update block —, then the pivots in
// registers, and (3) the finished columns for the update block.  NCH = children values per entry
// as a compile-time constant (0, 1, 2) or kMfNchAny: a run-time loop.
// ---------------------------------------------------------------------------
constexpr int kMfNchAny = 3;

// Phase clocks of the step kernel (profiles/ldlt_clocks.sh builds a library with -DSLPX_MF_CLOCKS): kept in LDS and
// written out when the task is through.  A clock stored to memory where it is taken is waited for at the next barrier
// (a workgroup barrier waits for the wave's outstanding stores): ~1 us of instrument per clock point, in a kernel
// whose phases are one to five microseconds.  The ordinary build carries no clock code in this kernel.
#ifdef SLPX_MF_CLOCKS
constexpr uint32_t kMfClockBytes = 192;  // the 24 slots of g_ldlt_clocks
constexpr uint32_t kMfClockTasks = 1024;
__device__ unsigned long long g_mf_clocks[kMfClockTasks * 24];  // EVERY task's slots of the last launch (one timeline)
#define SLPX_MF_CLOCK(k) \
  if (threadIdx.x == 0) s_clk[k] = wall_clock64()
#else
constexpr uint32_t kMfClockBytes = 0;
#define SLPX_MF_CLOCK(k)
#endif
constexpr uint32_t kMfImageGroups = 6144;  // 16-byte groups of a task's image the staging loop requests: 96 KB

// ---------------------------------------------------------------------------
// The update block on the MATRIX CORES (fronts flagged by the plan: at least four pivot columns under
// more rows than two lane-per-entry passes cover — the separator chains of the g-fold problem).
// S(a, b) for a in R + rhs, b in R, b <= a, in 16 x 16 tiles of v_mfma_f64_16x16x4_f64:
//   A operand (lane i + 16 k) = −L(R_{16 I + i}, 4 kb + k) = −U / d,   B operand (lane j + 16 k) = U(R_{16 J + j}, 4 kb + k),
//   C / D: column = lane & 15, row = (lane >> 4) + 4 q, q = 0..3 — C preloaded with the children's values
// (same update table as the lane-per-entry path: entry e = a (a + 1) / 2 + b).  One row of tiles at a
// time keeps the accumulators at four tiles (32 registers).  `u0`: LDS byte address of the front's
// first entry; the finished columns and 1/d are read back from LDS, where the pivots left them.
// ---------------------------------------------------------------------------
// (out of line: its accumulators and operands must not weigh on the register allocation of the level
// loop — inlined, the 1024-thread kernel went from 101 registers and no scratch to 128 and 288 bytes of it)
template <int W>
__device__ __attribute__((noinline)) void mf_update_mfma(uint32_t u0, uint32_t nr, uint32_t nch, bool root, uint32_t upd,
                                                         uint32_t ustride, uint32_t invd_addr,
                                                         const uint32_t* __restrict__ ext, double* __restrict__ contrib,
                                                         uint32_t lane) {
  u0 = __builtin_amdgcn_readfirstlane(u0);
  nr = __builtin_amdgcn_readfirstlane(nr);
  nch = __builtin_amdgcn_readfirstlane(nch);
  upd = __builtin_amdgcn_readfirstlane(upd);
  ustride = __builtin_amdgcn_readfirstlane(ustride);
  invd_addr = __builtin_amdgcn_readfirstlane(invd_addr);
  const uint32_t r = nr - W - 1u;
  const uint32_t li = lane & 15u, lk = lane >> 4;
  const uint32_t nb = (r + 1u + 15u) / 16u;  // row blocks over R + rhs (<= 4)
  constexpr int KB = (W + 3) / 4;
  // the B operands of every column block and pivot block (U of rows 16 J + li, column 4 kb + lk)
  double pB[4][KB], invc[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    const uint32_t c = 4u * kb + lk;
    const bool cok = c < static_cast<uint32_t>(W);
    const uint32_t cc = cok ? c : W - 1u;
    const uint32_t diag = u0 + 8u * (cc * nr - (cc * (cc - 1u)) / 2u);
    invc[kb] = lds_ld(invd_addr + 8u * cc);
#pragma unroll
    for (int J = 0; J < 4; ++J) {
      const uint32_t t = W + 16u * J + li;
      const bool ok = cok && t < nr;
      const double u = lds_ld(diag + 8u * ((ok ? t : static_cast<uint32_t>(W)) - cc));
      pB[J][kb] = ok ? u : 0.0;
    }
  }
  for (uint32_t I = 0; I < nb; ++I) {
    f64x4 acc[4];
    uint32_t ent[4][4];
    // children's values -> C
#pragma unroll
    for (int J = 0; J < 4; ++J) {
      acc[J] = f64x4{0.0, 0.0, 0.0, 0.0};
      if (static_cast<uint32_t>(J) > I) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t a = 16u * I + lk + 4u * q, b = 16u * J + li;
        const bool ok = b < r && b <= a && a <= r;
        ent[J][q] = ok ? upd + __umul24((a * (a + 1u)) / 2u + b, ustride) : 0xffffffffu;
        double v = 0.0;
        for (uint32_t k = 0; k < nch; ++k) v += lds_ld(lds_ld16((ok ? ent[J][q] : upd) + 6u + 2u * k));
        acc[J][q] = ok ? v : 0.0;
      }
    }
    // rank-W update of this row of tiles
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      // A operand of row block I: the same entries as the B operand of column block I, scaled
      double pa = 0.0;
#pragma unroll
      for (int J = 0; J < 4; ++J)
        if (static_cast<uint32_t>(J) == I) pa = pB[J][kb];
      const double ma = -(pa * invc[kb]);
#pragma unroll
      for (int J = 0; J < 4; ++J)
        if (static_cast<uint32_t>(J) <= I) acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, pB[J][kb], acc[J], 0, 0, 0);
    }
    // out
#pragma unroll
    for (int J = 0; J < 4; ++J) {
      if (static_cast<uint32_t>(J) > I) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (ent[J][q] != 0xffffffffu) {
          const uint32_t o = lds_ld16(ent[J][q]);
          if (root) coherent_store(&contrib[ext[o]], -acc[J][q], true);
          else lds_st(o, acc[J][q]);
        }
    }
  }
}

template <int W, int NCH, bool MFMA>
__device__ __forceinline__ void mf_front_w(uint32_t tab, uint32_t nr, uint32_t nch, uint32_t n_s, uint32_t fl,
                                           uint32_t invd_addr, const uint32_t* __restrict__ ext,
                                           double* __restrict__ contrib, uint32_t lane) {
  const bool root = (fl & 1u) != 0;
  constexpr bool kAny = NCH == kMfNchAny;
  constexpr int kN = kAny ? 0 : NCH;  // children handled in registers; the rest (kAny) by loops
  const uint32_t row = lane < nr ? lane : nr - 1u;  // (idle lanes shadow the last row: same loads, same stores)
  const uint32_t stride = 2u * W * (1u + nch);
  const uint32_t pr = tab + __umul24(row, stride);
  const uint32_t upd = tab + nr * stride;
  const uint32_t ustride = 2u * (3u + nch);
  const uint32_t e = lane < n_s ? lane : (n_s ? n_s - 1u : 0u);
  const uint32_t ur = upd + __umul24(e, ustride);
  // ---- trip 1: table words ----
  uint32_t ua[W], sa[kN ? kN : 1][W], o = 0, pa = 0, pb = 0, ss[kN ? kN : 1];
#pragma unroll
  for (int c = 0; c < W; ++c) ua[c] = lds_ld16(pr + 2u * c);
#pragma unroll
  for (int k = 0; k < kN; ++k)
#pragma unroll
    for (int c = 0; c < W; ++c) sa[k][c] = lds_ld16(pr + 2u * (W * (k + 1) + c));
  if (n_s) {
    o = lds_ld16(ur);
    pa = lds_ld16(ur + 2u);
    pb = lds_ld16(ur + 4u);
#pragma unroll
    for (int k = 0; k < kN; ++k) ss[k] = lds_ld16(ur + 6u + 2u * k);
  }
  // ---- trip 2: values ----
  double a[W], sv[kN ? kN : 1][W], cv[kN ? kN : 1];
#pragma unroll
  for (int c = 0; c < W; ++c) a[c] = lds_ld(ua[c]);
#pragma unroll
  for (int k = 0; k < kN; ++k)
#pragma unroll
    for (int c = 0; c < W; ++c) sv[k][c] = lds_ld(sa[k][c]);
  double v = 0.0;
  if (n_s) {
#pragma unroll
    for (int k = 0; k < kN; ++k) cv[k] = lds_ld(ss[k]);
  }
#pragma unroll
  for (int k = 0; k < kN; ++k)
#pragma unroll
    for (int c = 0; c < W; ++c) a[c] += sv[k][c];
  if (n_s) {
#pragma unroll
    for (int k = 0; k < kN; ++k) v = k == 0 ? cv[0] : v + cv[k];
  }
  if (kAny) {
    for (uint32_t k = 0; k < nch; ++k) {
      uint32_t xa[W];
#pragma unroll
      for (int c = 0; c < W; ++c) xa[c] = lds_ld16(pr + 2u * (W * (k + 1u) + c));
      const uint32_t xs = n_s ? lds_ld16(ur + 6u + 2u * k) : 0u;
#pragma unroll
      for (int c = 0; c < W; ++c) a[c] += lds_ld(xa[c]);
      if (n_s) v += lds_ld(xs);
    }
  }
  // ---- pivots ----
  double inv[W];
#pragma unroll
  for (int c = 0; c < W; ++c) {
    inv[c] = chain_reciprocal(readlane_f64(a[c], c));
    const double lc = a[c] * inv[c];
#pragma unroll
    for (int j = c + 1; j < W; ++j) a[j] = __builtin_fma(-lc, readlane_f64(a[c], j), a[j]);
  }
#pragma unroll
  for (int c = 0; c < W; ++c) lds_st(ua[c], a[c]);  // (rows above the diagonal: the scratch double)
#pragma unroll
  for (int c = 0; c < W; ++c) lds_st(invd_addr + 8u * c, inv[c]);  // (every lane the same value)
  if (n_s == 0) return;
  if constexpr (MFMA && W >= 4) {
    if (fl & 2u) {  // (wave-uniform)
      const uint32_t u0 = __builtin_amdgcn_readfirstlane(ua[0]);  // lane 0 holds row 0
      mf_update_mfma<W>(u0, nr, nch, root, upd, ustride, invd_addr, ext, contrib, lane);
      return;
    }
  }
  // ---- trip 3: the update block, a lane per entry ----
  {
    double ra[W], rb[W];
#pragma unroll
    for (int c = 0; c < W; ++c) {
      const uint32_t coff = mf_coff(c, nr);
      ra[c] = lds_ld(pa + coff);
      rb[c] = lds_ld(pb + coff);
    }
#pragma unroll
    for (int c = 0; c < W; ++c) v = __builtin_fma(-(ra[c] * inv[c]), rb[c], v);
    if (root) {
      // (the receiving entry subtracts; and only ONE store per slot: its reader re-arms it)
      if (lane < n_s) coherent_store(&contrib[ext[o]], -v, true);
    } else {
      lds_st(o, v);
    }
  }
  for (uint32_t e0 = 64u; e0 < n_s; e0 += 64u) {  // blocks of more than 64 entries (r >= 11)
    const uint32_t e2 = e0 + lane < n_s ? e0 + lane : n_s - 1u;
    const uint32_t ur2 = upd + __umul24(e2, ustride);
    const uint32_t o2 = lds_ld16(ur2), pa2 = lds_ld16(ur2 + 2u), pb2 = lds_ld16(ur2 + 4u);
    double ra[W], rb[W];
#pragma unroll
    for (int c = 0; c < W; ++c) {
      const uint32_t coff = mf_coff(c, nr);
      ra[c] = lds_ld(pa2 + coff);
      rb[c] = lds_ld(pb2 + coff);
    }
    double v2 = 0.0;
    for (uint32_t k = 0; k < nch; ++k) v2 += lds_ld(lds_ld16(ur2 + 6u + 2u * k));
#pragma unroll
    for (int c = 0; c < W; ++c) v2 = __builtin_fma(-(ra[c] * inv[c]), rb[c], v2);
    if (root) {
      if (e0 + lane < n_s) coherent_store(&contrib[ext[o2]], -v2, true);
    } else {
      lds_st(o2, v2);
    }
  }
}

template <int W, bool MFMA>
__device__ __forceinline__ void mf_front_nch(uint32_t tab, uint32_t nr, uint32_t nch, uint32_t n_s, uint32_t root,
                                             uint32_t invd_addr, const uint32_t* __restrict__ ext,
                                             double* __restrict__ contrib, uint32_t lane) {
  // (code size and register pressure: the in-register children only where they are common)
  if constexpr (W <= 2) {
    switch (nch) {
      case 0: mf_front_w<W, 0, MFMA>(tab, nr, nch, n_s, root, invd_addr, ext, contrib, lane); break;
      case 1: mf_front_w<W, 1, MFMA>(tab, nr, nch, n_s, root, invd_addr, ext, contrib, lane); break;
      case 2: mf_front_w<W, 2, MFMA>(tab, nr, nch, n_s, root, invd_addr, ext, contrib, lane); break;
      default: mf_front_w<W, kMfNchAny, MFMA>(tab, nr, nch, n_s, root, invd_addr, ext, contrib, lane); break;
    }
  } else if constexpr (W <= 5) {
    if (nch == 2) mf_front_w<W, 2, MFMA>(tab, nr, nch, n_s, root, invd_addr, ext, contrib, lane);
    else mf_front_w<W, kMfNchAny, MFMA>(tab, nr, nch, n_s, root, invd_addr, ext, contrib, lane);
  } else {
    mf_front_w<W, kMfNchAny, MFMA>(tab, nr, nch, n_s, root, invd_addr, ext, contrib, lane);
  }
}

// (every argument but `lane` wave-uniform: scalar registers, scalar jumps.  MFMA: the kernel variant
// for plans with fronts on the matrix cores — the mere presence of that path, even out of line, cost
// the other fronts' code 2 us per step in registers: plans without such fronts run the variant without it)
template <bool MFMA>
__device__ __forceinline__ void mf_front(uint32_t tab, uint32_t w, uint32_t nr, uint32_t nch, uint32_t n_s,
                                         uint32_t root, uint32_t invd_addr, const uint32_t* __restrict__ ext,
                                         double* __restrict__ contrib, uint32_t lane) {
  tab = __builtin_amdgcn_readfirstlane(tab);
  w = __builtin_amdgcn_readfirstlane(w);
  nr = __builtin_amdgcn_readfirstlane(nr);
  nch = __builtin_amdgcn_readfirstlane(nch);
  n_s = __builtin_amdgcn_readfirstlane(n_s);
  root = __builtin_amdgcn_readfirstlane(root);
  invd_addr = __builtin_amdgcn_readfirstlane(invd_addr);
  switch (w) {
#define SLPX_MF_CASE(W) case W: mf_front_nch<W, MFMA>(tab, nr, nch, n_s, root, invd_addr, ext, contrib, lane); break;
    SLPX_MF_CASE(1) SLPX_MF_CASE(2) SLPX_MF_CASE(3) SLPX_MF_CASE(4) SLPX_MF_CASE(5) SLPX_MF_CASE(6) SLPX_MF_CASE(7)
    SLPX_MF_CASE(8)
#undef SLPX_MF_CASE
    default: break;
  }
  static_assert(kSnWidthMax == 8, "one case per front width");
}

// ---------------------------------------------------------------------------
// Backward solve of one front, one wave: x_c = (U(rhs, c) − Σ_{t in R} U(t, c) x_t − Σ_{c' > c} U(c', c) x_c') / d_c.
// Eight lanes per pivot column over the rows of R (their x through the front's solve table), DPP
// sum, then the chain top-down with v_readlane.  `xr`: LDS byte address of the solve table,
// `u0`: of the front's first entry, `x_addr`: of x[col0], `invd_addr`: of 1/d[col0].
// Everything that does not depend on x is requested first: one dependent trip for the rows of R.
// ---------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void mf_solve_w(uint32_t xr, uint32_t u0, uint32_t nr, uint32_t invd_addr, uint32_t x_addr,
                                           uint32_t lane) {
  const uint32_t r = nr - W - 1u;
  const uint32_t c = W > 1 ? lane >> 3 : 0u, g = lane & 7u;
  const bool mine = W == 8 || c < static_cast<uint32_t>(W);
  const uint32_t cc = mine ? c : 0u;
  const uint32_t colc = u0 + 8u * (cc * nr - (cc * (cc - 1u)) / 2u);  // diagonal of column cc
  // two rows of R per lane cover r <= 16 (a clamped index and a select instead of a guarded load)
  const uint32_t a0 = g < r ? g : 0u, a1 = g + 8u < r ? g + 8u : 0u;
  const uint32_t x0a = lds_ld16(xr + 2u * a0), x1a = lds_ld16(xr + 2u * a1);
  const double u0v = lds_ld(colc + 8u * (W - cc + a0)), u1v = lds_ld(colc + 8u * (W - cc + a1));
  // the chain's own couplings U(k, cc), k > cc, 1/d and the right-hand-side row
  double cf[W];
#pragma unroll
  for (int k = 1; k < W; ++k) {
    const bool use = static_cast<uint32_t>(k) > cc;
    const double v = lds_ld(colc + 8u * (use ? k - cc : 0u));
    cf[k] = (use && mine) ? v : 0.0;
  }
  const double inv = lds_ld(invd_addr + 8u * cc);
  const double b = lds_ld(colc + 8u * (nr - 1u - cc));
  const double x0 = lds_ld(x0a), x1 = lds_ld(x1a);
  double dot = (g < r ? u0v : 0.0) * x0;
  dot = __builtin_fma(g + 8u < r ? u1v : 0.0, x1, dot);
  for (uint32_t a = g + 16u; a < r; a += 8u) dot = __builtin_fma(lds_ld(colc + 8u * (W - cc + a)), lds_ld(lds_ld16(xr + 2u * a)), dot);
  dot = group8_sum(dot);
  double p = b - dot;  // (lane 8 c holds column c)
#pragma unroll
  for (int k = W - 1; k >= 0; --k) {
    // column k is final once the columns above it are in: scale it, then hand it to the columns below
    const double xk = readlane_f64(p, W > 1 ? 8 * k : 0) * readlane_f64(inv, W > 1 ? 8 * k : 0);
    if (k > 0) p = __builtin_fma(-cf[k], xk, p);
    lds_st(x_addr + 8u * k, xk);  // (every lane the same value)
  }
}

__device__ __forceinline__ void mf_solve_front(uint32_t xr, uint32_t u0, uint32_t w, uint32_t nr, uint32_t invd_addr,
                                               uint32_t x_addr, uint32_t lane) {
  xr = __builtin_amdgcn_readfirstlane(xr);
  u0 = __builtin_amdgcn_readfirstlane(u0);
  w = __builtin_amdgcn_readfirstlane(w);
  nr = __builtin_amdgcn_readfirstlane(nr);
  invd_addr = __builtin_amdgcn_readfirstlane(invd_addr);
  x_addr = __builtin_amdgcn_readfirstlane(x_addr);
  switch (w) {
#define SLPX_MF_CASE(W) case W: mf_solve_w<W>(xr, u0, nr, invd_addr, x_addr, lane); break;
    SLPX_MF_CASE(1) SLPX_MF_CASE(2) SLPX_MF_CASE(3) SLPX_MF_CASE(4) SLPX_MF_CASE(5) SLPX_MF_CASE(6) SLPX_MF_CASE(7)
    SLPX_MF_CASE(8)
#undef SLPX_MF_CASE
    default: break;
  }
}

// ---------------------------------------------------------------------------
// The step: one workgroup per task, every round in the launch (tasks wait for each other through the
// update slots and, in the solve, through x itself: every workgroup must be resident).
// LDS (bytes; the first four regions are what the tables address, so they start at 0):
//   U[n_ent] f64 | arena f64 | 1/d[n_col] f64 | x[n_col + n_anc + 1] f64 |
//   tables u16 | levels u32 | ext u32 | src i32 | flags u8 | entries with update slots u16 | their slot ranges u32 | slots u32 |
//   colperm u32 | anc u32 | front descriptors 16 B | counters 32 B | KKT terms + products | back-substitution rows
// ---------------------------------------------------------------------------
struct MfCarve {
  uint32_t o_arena, o_invd, o_x, o_tab, o_lvl, o_ext, o_src, o_flags, o_cent, o_cptr, o_cidx, o_cp, o_anc,
      o_fr, o_cnt, o_terms;
};
__host__ __device__ inline uint32_t mf_align16(uint32_t v) { return (v + 15u) & ~15u; }
__host__ __device__ inline MfCarve mf_carve(const LdltTask& t, const LdltMfTask& m) {
  MfCarve c;
  auto q = [](uint32_t count, uint32_t per16) { return 16u * ((count + per16 - 1u) / per16); };
  c.o_arena = 8u * t.n_ent;
  c.o_invd = c.o_arena + 8u * m.arena;
  c.o_x = c.o_invd + 8u * t.n_col;
  c.o_tab = mf_align16(c.o_x + 8u * (t.n_col + m.n_anc + 1u));
  c.o_lvl = c.o_tab + q(m.n_tab, 8);
  c.o_ext = c.o_lvl + q(t.n_lvl + 1u, 4);
  c.o_src = c.o_ext + q(m.n_ext, 4);
  c.o_flags = c.o_src + q(t.n_ent, 4);
  c.o_cent = c.o_flags + q(t.n_ent, 16);
  c.o_cptr = c.o_cent + q(m.n_cent, 8);
  c.o_cidx = c.o_cptr + q(m.n_cent + 1u, 4);
  c.o_cp = c.o_cidx + q(m.n_contrib_idx, 4);
  c.o_anc = c.o_cp + q(t.n_col, 4);
  c.o_fr = c.o_anc + q(m.n_anc, 4);
  c.o_cnt = c.o_fr + 16u * m.n_front;
  c.o_terms = c.o_cnt + 32u + kMfClockBytes;
  return c;
}

// CHAINED: the variant of a chained step (DeviceNlp::sweep_full_for_step) — it waits for its sweep after
// staging and counts its workgroups out; the other variant has none of that code.
// `block`: the workgroup's place among [ride-along sums | tasks]; `exit_total` workgroups count themselves out
// before the verdict is published; `twin_stats`: the counters of a second attempt made in the same launch
// (ldlt_mf_twin_kernel), published with this one's.
template <int THREADS, bool MFMA, bool CHAINED, bool RIDE = false>
__device__ __forceinline__ void mf_step_body(
    const LdltDev& L, const MfDev& Mf, const double* __restrict__ lhs, const double* __restrict__ rhs, const double* __restrict__ reg,
    double* __restrict__ Lx, double* __restrict__ D, int n, double* __restrict__ contrib, LdltStats* __restrict__ stats,
    LdltStats* __restrict__ stats_next, double* __restrict__ zv, const KktFuse& F, double* __restrict__ xg,
    double* __restrict__ xg_next, double* __restrict__ out, const BacksubFuse& B, uint32_t block, unsigned int exit_total,
    const LdltStats* twin_stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (Mf.gate != nullptr && Mf.gate[0] == 0.0) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && B.stats_host != nullptr) {
      B.stats_host[0] = LdltStats{0, 0, 0, kLdltPassed, 0x7ff0000000000000ull};
      if (B.seq_host != nullptr) {
        __threadfence_system();
        const unsigned long long v = *B.seq_dev + 1;
        *B.seq_dev = v;
        *B.seq_host = v;
      }
    }
    return;
  }
  if (static_cast<int>(block) < F.n_blocks) {
    if constexpr (CHAINED) mf_wait_for_sweep(Mf, stats);
    ride_along_sum(F, block, smem_raw);
    if constexpr (CHAINED) mf_signal_done(Mf);
    return;
  }
  const int tid = threadIdx.x;
  const uint32_t lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t task_index = block - static_cast<uint32_t>(F.n_blocks);
  const LdltTask t = L.tasks[task_index];
  const LdltMfTask m = Mf.tasks[task_index];
  const bool top = static_cast<int>(t.round) + 1 == L.n_rounds;
  const double delta = reg != nullptr ? reg[0] : Mf.delta, gamma = reg != nullptr ? reg[1] : Mf.gamma;
#ifdef SLPX_MF_CLOCKS
  const unsigned long long clk_entry = wall_clock64();  // (slot 0, stored behind the staging loop: the image covers the slots)
#endif
  const MfCarve cv = mf_carve(t, m);
  double* U = reinterpret_cast<double*>(smem_raw);
  double* arena = reinterpret_cast<double*>(smem_raw + cv.o_arena);
  double* invd = reinterpret_cast<double*>(smem_raw + cv.o_invd);
  double* x = reinterpret_cast<double*>(smem_raw + cv.o_x);
    const uint32_t* lvl = reinterpret_cast<const uint32_t*>(smem_raw + cv.o_lvl);
  const uint32_t* ext = reinterpret_cast<const uint32_t*>(smem_raw + cv.o_ext);
  const int32_t* src = reinterpret_cast<const int32_t*>(smem_raw + cv.o_src);
  const uint8_t* flags = reinterpret_cast<const uint8_t*>(smem_raw + cv.o_flags);
  const uint16_t* cent = reinterpret_cast<const uint16_t*>(smem_raw + cv.o_cent);
  const uint32_t* cptr = reinterpret_cast<const uint32_t*>(smem_raw + cv.o_cptr);
  const uint32_t* cidx = reinterpret_cast<const uint32_t*>(smem_raw + cv.o_cidx);
  const uint32_t* colperm = reinterpret_cast<const uint32_t*>(smem_raw + cv.o_cp);
  const uint32_t* anc = reinterpret_cast<const uint32_t*>(smem_raw + cv.o_anc);
  int* s_cnt = reinterpret_cast<int*>(smem_raw + cv.o_cnt);
  unsigned long long* s_minp = reinterpret_cast<unsigned long long*>(s_cnt + 4);
#ifdef SLPX_MF_CLOCKS
  unsigned long long* s_clk = reinterpret_cast<unsigned long long*>(smem_raw + cv.o_cnt + 32u);
#endif
  uint4* s_terms = reinterpret_cast<uint4*>(smem_raw + cv.o_terms);
  auto g16 = [&](uint32_t o) { return reinterpret_cast<uint4*>(smem_raw + o); };

  // ---- stage the plan: one image, one copy loop, every load in flight ----
  // (the image sits in a slot of fixed size, so its loads are requested before the task's own
  // descriptors — which say how much of the slot is image — have arrived: one trip to memory, not two)
  constexpr int kInFlight = kMfImageGroups / THREADS;  // (DeviceNlp::build_mf: no image is longer than kMfImageGroups)
  uint4 stage_v[kInFlight];
  const uint4* src16 = Mf.image + static_cast<size_t>(task_index) * Mf.image_stride16;
#pragma unroll
  for (int k = 0; k < kInFlight; ++k) stage_v[k] = src16[tid + k * THREADS];  // (the buffer is padded by a full round)
  const uint4 img = Mf.image_desc[task_index];
  const uint32_t n_terms16 = img.w, n_terms = n_terms16 * 4u / 3u;
  double* tprod = reinterpret_cast<double*>(s_terms + n_terms16);
  uint4* s_bs = reinterpret_cast<uint4*>(smem_raw + mf_align16(cv.o_terms + 16u * n_terms16 + 8u * n_terms));
  uint4 bs_task = B.task_plan[task_index];
  {
    uint4* dst_a = g16(cv.o_tab);
    const uint32_t n_a = img.y, n_all = img.y + img.z;
#pragma unroll
    for (int k = 0; k < kInFlight; ++k) {
      const uint32_t i = tid + k * THREADS;
      if (i < n_a) dst_a[i] = stage_v[k];
      else if (i < n_all) s_bs[i - n_a] = stage_v[k];
    }
  }
  if (tid == 0) {
    arena[0] = 0.0;
    arena[1] = 0.0;
    x[t.n_col + m.n_anc] = 1.0;
  }
  __syncthreads();
#ifdef SLPX_MF_CLOCKS
  if (threadIdx.x == 0) s_clk[0] = clk_entry;
#endif
  SLPX_MF_CLOCK(1);
  if (stats_next != nullptr && task_index == 0 && tid == 0) stats_next[0] = LdltStats{0, 0, 0, 0, 0x7ff0000000000000ull};
  if constexpr (CHAINED) mf_wait_for_sweep(Mf, stats);
  SLPX_MF_CLOCK(14);  // (slot 14: the sweep's numbers may be read)

  // ---- matrix values (ldlt_factor_body) ----
  bool regularized = false;  // (uniform)
  if (!F.inline_kkt) {
    // (a re-attempt of the policy loop on the system already in memory: the image holds what every
    // entry is MADE of, where it sits in lhs / rhs comes from the plan in memory)
    for (uint32_t i = tid; i < t.n_ent; i += THREADS) {
      const int32_t s0 = L.ent_src[t.ent_off + i];
      const double* base = (flags[i] & 4) ? rhs : lhs;
      U[i] = s0 >= 0 ? base[s0] : 0.0;
    }
  } else {
    const double mu = F.mu[0];
    const KktTerm* terms = reinterpret_cast<const KktTerm*>(s_terms);
    const uint32_t span = n_terms > t.n_ent ? n_terms : t.n_ent;
    // (three rounds of the workgroup are requested together: ONE trip to memory for a task of up to 3 THREADS entries
    // and terms — a loop of one round per pass made as many dependent trips as it had passes, two to three per task)
    constexpr int kEv = 3;
    for (uint32_t k0 = tid; k0 < span; k0 += kEv * THREADS) {
      int w[kEv];
      double v[kEv];
      KktTermLoads tl[kEv];
#pragma unroll
      for (int u = 0; u < kEv; ++u) {
        const uint32_t k = k0 + u * THREADS;
        w[u] = k < t.n_ent ? src[k] : -1;
        v[u] = w[u] >= 0 ? F.V[w[u] & 0x3fffffff] : 0.0;
        const bool on = k < n_terms;
        tl[u] = kkt_term_fetch(on ? terms[k] : KktTerm{0, 0, 0}, on, F.V, F.s, F.y, F.z);
      }
#pragma unroll
      for (int u = 0; u < kEv; ++u) {
        const uint32_t k = k0 + u * THREADS;
        if (k < t.n_ent && w[u] >= -1) U[k] = (w[u] & 0x40000000) && w[u] >= 0 ? -v[u] : v[u];
        if (k < n_terms) tprod[k] = kkt_term_product(tl[u], mu);
      }
    }
    __syncthreads();
    SLPX_MF_CLOCK(15);  // (slot 15: values and products are in LDS)
    // the sums of terms and, in the same pass, the regularization (a lane reads back what it wrote itself; flag
    // bit 5, set in the task's image only: the entry takes update slots and is regularized below, with them)
    const bool fold_reg = F.store_lhs == nullptr;  // (tests ask for the evaluated system as an assembly pass leaves it: unregularized)
    for (uint32_t i = tid; i < t.n_ent; i += THREADS) {
      const int w = src[i];
      const uint8_t fl = flags[i];
      const bool summed = w < -1, reg_here = fold_reg && (fl & 0x21) == 1;
      if (summed || reg_here) {
        double u;
        if (summed) {
          u = kkt_terms_sum_grouped(tprod, static_cast<uint32_t>(-(w + 2)), (fl & 4) != 0);
        } else {
          u = U[i];
        }
        if (reg_here) u += (fl & 2) ? -gamma : delta;
        U[i] = u;
      }
    }
    regularized = fold_reg;
    SLPX_MF_CLOCK(4);  // (slot 4: this wave's sums are done)
    if (F.store_lhs != nullptr) {
      __syncthreads();
      for (uint32_t i = tid; i < t.n_ent; i += THREADS) {
        const int32_t s0 = L.ent_src[t.ent_off + i];
        if (s0 >= 0) ((flags[i] & 4) ? F.store_rhs : F.store_lhs)[s0] = U[i];
      }
    }
  }
  __syncthreads();
  SLPX_MF_CLOCK(13);  // (slot 13: every wave's)
  // ---- regularization + update blocks of child tasks (slots: ldlt_factor_body) ----
  // (flag bit 5, set in the task's image only: the entry takes update slots and is handled below)
  if (!regularized)
    for (uint32_t i = tid; i < t.n_ent; i += THREADS) {
      const uint8_t fl = flags[i];
      if ((fl & 0x21) == 1) U[i] += (fl & 2) ? -gamma : delta;
    }
  for (uint32_t j = tid; j < m.n_cent; j += THREADS) {
    const uint32_t i = cent[j];
    const uint8_t fl = flags[i];
    const uint32_t cb = cptr[j], ce = cptr[j + 1];
    double acc = U[i];
    if (fl & 1) acc += (fl & 2) ? -gamma : delta;
    for (uint32_t c = cb; c < ce; c += 4) {
      double* p0 = &contrib[cidx[c]];
      double* p1 = &contrib[cidx[c + 1 < ce ? c + 1 : c]];
      double* p2 = &contrib[cidx[c + 2 < ce ? c + 2 : c]];
      double* p3 = &contrib[cidx[c + 3 < ce ? c + 3 : c]];
      double v0, v1, v2, v3;
      unsigned int spins = 0;
      for (;;) {
        v0 = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v1 = __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v2 = __hip_atomic_load(p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v3 = __hip_atomic_load(p3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!slot_is_empty(v0) && !slot_is_empty(v1) && !slot_is_empty(v2) && !slot_is_empty(v3)) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) {  // never expected: mark the factorization bad instead of hanging
          atomicAdd(&stats[0].n_bad, 1 << 20);
          break;
        }
      }
      const double armed = __longlong_as_double(static_cast<long long>(kSlotEmpty));
      __hip_atomic_store(p0, armed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc -= v0;
      if (c + 1 < ce) {
        __hip_atomic_store(p1, armed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc -= v1;
      }
      if (c + 2 < ce) {
        __hip_atomic_store(p2, armed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc -= v2;
      }
      if (c + 3 < ce) {
        __hip_atomic_store(p3, armed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc -= v3;
      }
    }
    U[i] = acc;
  }
  __syncthreads();
  SLPX_MF_CLOCK(2);

  // ---- levels: a wave per front ----
  // (the descriptor of the wave's front in the NEXT level is requested before this level's work)
  const uint32_t gfr = cv.o_fr;  // (LDS byte address of the task's front descriptors)
  const uint32_t last_front = m.n_front ? m.n_front - 1u : 0u;
  {
    uint32_t beg = __builtin_amdgcn_readfirstlane(lvl[0]), end = __builtin_amdgcn_readfirstlane(t.n_lvl ? lvl[1] : 0);
    u32x4 d = lds_load_desc(gfr + 16u * (beg + wave < last_front ? beg + wave : last_front));
    for (uint32_t l = 0; l < t.n_lvl; ++l) {
      const uint32_t next_end = __builtin_amdgcn_readfirstlane(lvl[l + 2 <= t.n_lvl ? l + 2 : t.n_lvl]);
      for (uint32_t q = beg + wave; q < end; q += THREADS / 64) {
        if (q != beg + wave) d = lds_load_desc(gfr + 16u * q);
        const uint32_t w = d[2] & 0xffu, nr = (d[2] >> 8) & 0xffu, nch = (d[2] >> 16) & 0xffu, root = (d[2] >> 24) & 3u;
        mf_front<MFMA>(cv.o_tab + 2u * d[0], w, nr, nch, d[3] & 0xffffu, root, cv.o_invd + 8u * (d[1] >> 16), ext + (d[3] >> 16),
                 contrib, lane);
      }
      d = lds_load_desc(gfr + 16u * (end + wave < last_front ? end + wave : last_front));
      __syncthreads();
      if (l < 8u) SLPX_MF_CLOCK(8u + l);  // (slots 8..15: the end of the first eight levels)
      beg = end;
      end = next_end;
    }
  }
  SLPX_MF_CLOCK(3);
  // (a step that carries the error launch of the iteration before it, MfDev::ride_verdict: does it count?  The riding
  // workgroups were dispatched first and are through long before a task's levels are; one lane asks, the others come
  // through the barrier and read the settled word themselves)
  bool counts = true;
  if constexpr (RIDE) {
    if (Mf.ride_verdict != nullptr) {
      if (tid == 0) {
        unsigned int spins = 0;
        while (fabs(__hip_atomic_load(Mf.ride_verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != Mf.ride_ticket) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1u << 22)) {  // never expected: mark the factorization bad instead of hanging
            atomicAdd(&stats[0].n_bad, 1 << 20);
            break;
          }
        }
      }
      __syncthreads();
      counts = __hip_atomic_load(Mf.ride_verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == Mf.ride_ticket;
    }
  }
  // results + inertia (ldlt_factor_exit; where an entry goes and whose 1/d scales it: streamed from the
  // plan in memory — off the critical path, and 6 bytes per entry that need not sit in LDS)
  auto exit_and_count = [&] {
    const uint32_t* g_out = L.ent_out + t.ent_off;
    const uint16_t* g_col = L.ent_col + t.ent_off;
    for (uint32_t i = counts ? tid : t.n_ent; i < t.n_ent; i += THREADS) {
      const double u = U[i];
      const uint8_t fl = flags[i];
      const uint32_t o = g_out[i];
      if (fl & 1) {
        D[o] = u;
        const double eps = 2.220446049250313e-16;
        if (u > eps) atomicAdd(&s_cnt[0], 1);
        else if (u < -eps) atomicAdd(&s_cnt[1], 1);
        else atomicAdd(&s_cnt[2], 1);
        if (u == 0.0 || !isfinite(u)) atomicAdd(&s_cnt[3], 1);
        else atomicMin(s_minp, static_cast<unsigned long long>(__double_as_longlong(fabs(u))));
      } else if (fl & 4) {
        zv[o] = u * invd[g_col[i]];  // z = D⁻¹L⁻¹Pb: the forward solve came for free
      } else {
        Lx[o] = u * invd[g_col[i]];
      }
    }
    __syncthreads();
    if (counts) {
      if (tid < 4 && s_cnt[tid] != 0) atomicAdd(reinterpret_cast<int*>(&stats[0]) + tid, s_cnt[tid]);
      if (tid == 0) atomicMin(&stats[0].min_abs_bits, *s_minp);
    }
    if (threadIdx.x == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this lane's counter updates are in
      const unsigned int old = __hip_atomic_fetch_add(Mf.exit_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == exit_total) {
        __hip_atomic_store(Mf.exit_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (B.stats_host != nullptr) {
          if (counts) {
            if (twin_stats != nullptr) publish_stats_copy(twin_stats, B.stats_host + 1);
            publish_stats(B, true);
          } else if (B.seq_host != nullptr) {  // (the marker of a launch that does not count: kLdltPassed)
            B.stats_host[0] = LdltStats{0, 0, 0, kLdltPassed, 0x7ff0000000000000ull};
            __threadfence_system();
            const unsigned long long v = *B.seq_dev + 1;
            *B.seq_dev = v;
            *B.seq_host = v;
          }
        }
      }
    }
  };
  if (!top) exit_and_count();  // (its store acknowledgements come in while the task waits for its ancestors)
  SLPX_MF_CLOCK(5);

  // ---- backward solve on what the factorization left in LDS ----
  // this lane's row of the back-substitution (ldlt_bwd_run): everything but p is known now
  // (fetched now, while the task waits for its ancestors; parked in LDS — where the KKT terms were —
  // during the level loop: held in registers they pushed the kernel over the 128 a 1024-thread
  // workgroup allows, and the compiler's spill of one of them sat in a block that runs with a
  // partial exec mask: most lanes got zeros back)
  constexpr int kBsPre = 4;
  constexpr uint32_t kBsStash = 5u + 2u * kBsPre;  // doubles per row: s, z, c_i, a[], p[], refs (two per double)
  const BsRow* bs_rows = reinterpret_cast<const BsRow*>(s_bs);
  const BsTerm* bs_terms = reinterpret_cast<const BsTerm*>(s_bs + bs_task.w);
  double* stash = reinterpret_cast<double*>(s_terms);
  const bool stash_fits = 8u * kBsStash * bs_task.z <= 16u * n_terms16 + 8u * n_terms;
  const bool bs_mine = static_cast<uint32_t>(tid) < bs_task.z && t.round == 0 && stash_fits;
  BsRow bs_row = BsRow{0, 0};
  double bs_a[kBsPre], bs_p[kBsPre], bs_s = 1.0, bs_z = 0.0, bs_ci = 0.0;
  uint32_t bs_ref[kBsPre];
#pragma unroll
  for (int k = 0; k < kBsPre; ++k) {
    bs_a[k] = 0.0;
    bs_p[k] = 0.0;
    bs_ref[k] = 0;
  }
  if (bs_mine) {
    bs_row = bs_rows[tid];
    const uint32_t first = bs_row.terms & 0xfffffu, cnt = bs_row.terms >> 20;
    bs_s = B.s[bs_row.r];
    bs_z = B.z[bs_row.r];
    bs_ci = B.V[B.off_ci + bs_row.r];
#pragma unroll
    for (int k = 0; k < kBsPre; ++k)
      if (static_cast<uint32_t>(k) < cnt) {
        const BsTerm bt = bs_terms[first + k];
        bs_a[k] = B.V[bt.a];
        bs_ref[k] = bt.ref;
      }
  }
  SLPX_MF_CLOCK(17);
  // x of the ancestor tasks' rows the fronts reach: handed over through the values themselves
  for (uint32_t a = tid; a < m.n_anc; a += THREADS) x[t.n_col + a] = slot_read(&xg[anc[a]]);
  if (bs_mine) {
#pragma unroll
    for (int k = 0; k < kBsPre; ++k)
      if (bs_ref[k] & 0x80000000u) bs_p[k] = slot_read(&xg[bs_ref[k] & 0x7fffffffu]);
    double* st = stash + kBsStash * static_cast<uint32_t>(tid);
    st[0] = bs_s;
    st[1] = bs_z;
    st[2] = bs_ci;
#pragma unroll
    for (int k = 0; k < kBsPre; ++k) {
      st[3 + k] = bs_a[k];
      st[3 + kBsPre + k] = bs_p[k];
      reinterpret_cast<uint32_t*>(st + 3 + 2 * kBsPre)[k] = bs_ref[k];
    }
  }
  __syncthreads();
  SLPX_MF_CLOCK(18);
  {
    uint32_t end = __builtin_amdgcn_readfirstlane(lvl[t.n_lvl]), beg = __builtin_amdgcn_readfirstlane(t.n_lvl ? lvl[t.n_lvl - 1] : 0);
    u32x4 d = lds_load_desc(gfr + 16u * (beg + wave < last_front ? beg + wave : last_front));
    for (int l = static_cast<int>(t.n_lvl) - 1; l >= 0; --l) {
      const uint32_t next_beg = __builtin_amdgcn_readfirstlane(lvl[l >= 1 ? l - 1 : 0]);
      for (uint32_t q = beg + wave; q < end; q += THREADS / 64) {
        if (q != beg + wave) d = lds_load_desc(gfr + 16u * q);
        const uint32_t w = d[2] & 0xffu, nr = (d[2] >> 8) & 0xffu, nch = (d[2] >> 16) & 0xffu, n_s = d[3] & 0xffffu;
        const uint32_t xr = cv.o_tab + 2u * (d[0] + nr * w * (1u + nch) + n_s * (3u + nch));
        mf_solve_front(xr, 8u * (d[1] & 0xffffu), w, nr, cv.o_invd + 8u * (d[1] >> 16), cv.o_x + 8u * (d[1] >> 16), lane);
      }
      d = lds_load_desc(gfr + 16u * (next_beg + wave < last_front ? next_beg + wave : last_front));
      __syncthreads();
      {  // (slots 6, 7, 21, 22, 23: the end of the first five levels of the backward solve, top level first)
        const uint32_t k = t.n_lvl - 1u - static_cast<uint32_t>(l);
        if (k < 5u) SLPX_MF_CLOCK(k < 2u ? 6u + k : 19u + k);
      }
      end = beg;
      beg = next_beg;
    }
  }
  SLPX_MF_CLOCK(19);
  // the descendants wait for x alone: hand it over before anything else goes out
  for (uint32_t i = tid; i < t.n_col; i += THREADS) coherent_store(&xg[colperm[i]], x[i], true);
  for (uint32_t i = tid; i < t.n_col; i += THREADS)
    coherent_store(&xg_next[colperm[i]], __longlong_as_double(static_cast<long long>(kSlotEmpty)), true);
  if (counts)
    for (uint32_t i = tid; i < t.n_col; i += THREADS) out[L.perm[colperm[i]]] = x[i];
  if (B.on && counts) {
    const double mu = B.mu[0];
    auto p_of = [&](uint32_t ref) { return (ref & 0x80000000u) ? coherent_load(&xg[ref & 0x7fffffffu], true) : x[ref]; };
    for (uint32_t j = tid; j < bs_task.z; j += THREADS) {
      const bool ahead = bs_mine && j == static_cast<uint32_t>(tid);
      const BsRow row = bs_rows[j];
      const uint32_t first = row.terms & 0xfffffu, cnt = row.terms >> 20;
      double aipx = 0.0, s_r = 1.0, z_r = 0.0, ci_r = 0.0;
      uint32_t k0 = 0;
      if (ahead) {
        const double* st = stash + kBsStash * static_cast<uint32_t>(tid);
        s_r = st[0];
        z_r = st[1];
        ci_r = st[2];
#pragma unroll
        for (int k = 0; k < kBsPre; ++k)
          if (static_cast<uint32_t>(k) < cnt) {
            const uint32_t ref = reinterpret_cast<const uint32_t*>(st + 3 + 2 * kBsPre)[k];
            aipx = backsub_dot(aipx, st[3 + k], (ref & 0x80000000u) ? st[3 + kBsPre + k] : x[ref]);
          }
        k0 = kBsPre;
      } else {  // (nothing was fetched ahead)
        s_r = B.s[row.r];
        z_r = B.z[row.r];
        ci_r = B.V[B.off_ci + row.r];
      }
      for (uint32_t k = k0; k < cnt; ++k) {
        const BsTerm bt = bs_terms[first + k];
        aipx = backsub_dot(aipx, B.V[bt.a], p_of(bt.ref));
      }
      double ps_r, pz_r;
      backsub_row(ci_r, s_r, z_r, mu, aipx, &ps_r, &pz_r);
      B.ps[row.r] = ps_r;
      B.pz[row.r] = pz_r;
    }
  }
  SLPX_MF_CLOCK(20);
  if (top) exit_and_count();
  if constexpr (CHAINED) mf_signal_done(Mf);
#ifdef SLPX_MF_CLOCKS
  if (threadIdx.x == 0) {
    s_clk[16] = wall_clock64();  // (slot 16: the task is through)
    if (task_index == L.clock_task)
      for (int k = 0; k < 24; ++k) g_ldlt_clocks[k] = s_clk[k];
    if (task_index < kMfClockTasks)
      for (int k = 0; k < 24; ++k) g_mf_clocks[task_index * 24u + k] = s_clk[k];
  }
#endif
}

template <int THREADS, bool MFMA, bool CHAINED>
__global__ __launch_bounds__(THREADS) void ldlt_mf_step_kernel(
    LdltDev L, MfDev Mf, const double* __restrict__ lhs, const double* __restrict__ rhs, const double* __restrict__ reg,
    double* __restrict__ Lx, double* __restrict__ D, int n, double* __restrict__ contrib, LdltStats* __restrict__ stats,
    LdltStats* __restrict__ stats_next, double* __restrict__ zv, KktFuse F, double* __restrict__ xg,
    double* __restrict__ xg_next, double* __restrict__ out, BacksubFuse B) {
  mf_step_body<THREADS, MFMA, CHAINED>(L, Mf, lhs, rhs, reg, Lx, D, n, contrib, stats, stats_next, zv, F, xg, xg_next, out, B,
                                       blockIdx.x, Mf.n_tasks, nullptr);
}

// ---------------------------------------------------------------------------
// Twin attempt: TWO regularizations of the same system factored, solved and back-substituted in one launch —
// the one the policy loop (sparse_regularized_ldlt.hpp:64-152) is at, and the one it would try next if this
// one shows too many negative pivots (delta x 10, :127-130), or, beside the unregularized first attempt, its first
// guess (:95-102).  The step kernel is bound by latency and leaves most of the chip idle at the horizons one
// problem has (15 workgroups at N = 100, 69 at N = 500): the second attempt costs nothing but power, and a step
// whose first attempt fails — every third to fourth interior-point iteration on BASELINE's cart-pole, which
// regularizes throughout — no longer pays a host round trip and a second launch (~47 us).  Which of the two the
// policy takes is decided twice from the same counters, by the host (NewtonSystem::compute_impl) and by the
// launch that consumes the direction (ipm_lookahead_kernel), so the result is the sequential policy's bit for bit.
// Workgroups [0, first_end) are the first attempt's (ride-along sums and tasks), the rest the second's tasks:
// same plan, same V, their own factor, update slots, x hand-over, direction and counters.
// ---------------------------------------------------------------------------
struct MfTwin {
  unsigned int first_end = 0;
  double delta = 0.0, gamma = 0.0;  // (by value, as MfDev's)
  double *Lx = nullptr, *D = nullptr, *contrib = nullptr, *zv = nullptr, *xg = nullptr, *xg_next = nullptr, *out = nullptr,
         *ps = nullptr, *pz = nullptr;
  LdltStats *stats = nullptr, *stats_next = nullptr;
  // a second attempt whose SYSTEM differs from the first's beyond (delta, gamma) on the diagonal (feasibility
  // restoration: the eliminated rows' share of it depends on delta, restoration.hpp): its own lhs / rhs in memory
  const double *lhs = nullptr, *rhs = nullptr;
};

// The error launch of the iteration before this step, riding in front of its tasks (MfDev::ride_verdict): the
// arguments of ipm_error_partial_kernel, `n_blocks` workgroups [error workgroups | one per separable sum].
struct MfRide {
  uint32_t n_blocks = 0;
  KktDev K;
  const double *V = nullptr, *x = nullptr, *s = nullptr, *y = nullptr, *z = nullptr, *scales = nullptr;
  int nV = 0;
  double* partial = nullptr;
  IpmErrFinish fin;
};

template <int THREADS, bool RIDE = false>
__global__ __launch_bounds__(THREADS) void ldlt_mf_twin_kernel(
    LdltDev L, MfDev Mf, const double* __restrict__ lhs, const double* __restrict__ rhs, const double* __restrict__ reg,
    double* __restrict__ Lx, double* __restrict__ D, int n, double* __restrict__ contrib, LdltStats* __restrict__ stats,
    LdltStats* __restrict__ stats_next, double* __restrict__ zv, KktFuse F, double* __restrict__ xg,
    double* __restrict__ xg_next, double* __restrict__ out, BacksubFuse B, MfTwin T, MfRide R) {
  uint32_t block = blockIdx.x;
  if constexpr (RIDE) {
    static_assert(THREADS >= kIpmErrThreads, "ipm_error_block's lanes");
    if (block < R.n_blocks) {
      extern __shared__ __attribute__((aligned(16))) unsigned char ride_smem[];
      ipm_error_block(R.K, R.V, R.nV, R.x, R.s, R.y, R.z, R.scales, 0, R.partial, R.fin, static_cast<int>(block), 0,
                      reinterpret_cast<double*>(ride_smem));
      return;
    }
    block -= R.n_blocks;
  }
  if (block >= T.first_end) {  // (uniform over the workgroup: scalar selects)
    block = block - T.first_end + static_cast<uint32_t>(F.n_blocks);
    Mf.delta = T.delta;
    Mf.gamma = T.gamma;
    Lx = T.Lx;
    D = T.D;
    contrib = T.contrib;
    zv = T.zv;
    xg = T.xg;
    xg_next = T.xg_next;
    out = T.out;
    stats = T.stats;
    stats_next = T.stats_next;
    B.ps = T.ps;
    B.pz = T.pz;
    if (T.lhs != nullptr) lhs = T.lhs;
    if (T.rhs != nullptr) rhs = T.rhs;
    F.store_lhs = nullptr;  // (the first attempt's workgroups keep the assembled system for later attempts)
  }
  mf_step_body<THREADS, false, false, RIDE>(L, Mf, lhs, rhs, reg, Lx, D, n, contrib, stats, stats_next, zv, F, xg, xg_next, out, B, block,
                                            2u * Mf.n_tasks, T.stats);
}

// ---------------------------------------------------------------------------
// A NEW right-hand side through the fronts (second-order corrections, interior_point.hpp:611-619: up to five per
// iteration; the multiplier estimate; refinement; RegularizedLDLT::solve, sparse_regularized_ldlt.hpp:159-161): the
// factor the step kernel left in memory (Lx, D) goes back into every task's front layout, the forward
// substitution runs as the right-hand-side ROW of the factorization did — per front: the row's entries of the
// pivot columns (own value + the children's), eliminated against the finished columns, then its part of the update
// block, a lane per row of R —, the backward solve is the step kernel's (mf_solve_front).  One launch, every task
// resident, hand-overs through the update slots (forward: only the slots of right-hand-side entries are written
// and taken; the others stay armed) and through x itself (backward).  The pair-list kernels took two launches and
// 24 + 17.5 us for this at cart-pole N=500 (ldlt_fwd_kernel, ldlt_bwd_kernel on L in memory).
// ---------------------------------------------------------------------------
// (laid out like mf_front_w: every table word first, then every value — a lane per pivot column for the row's own
// entries and their couplings, a lane per row of R for the update block —, then the chain with v_readlane broadcasts:
// three dependent trips to LDS and w steps, where a lane walking the chain alone made ~40 trips)
__device__ __forceinline__ void mf_fwd_front(uint32_t tab, uint32_t w, uint32_t nr, uint32_t nch, uint32_t root, uint32_t invd_addr,
                                             const uint32_t* __restrict__ ext, double* __restrict__ contrib, uint32_t lane) {
  tab = __builtin_amdgcn_readfirstlane(tab);
  w = __builtin_amdgcn_readfirstlane(w);
  nr = __builtin_amdgcn_readfirstlane(nr);
  nch = __builtin_amdgcn_readfirstlane(nch);
  root = __builtin_amdgcn_readfirstlane(root);
  invd_addr = __builtin_amdgcn_readfirstlane(invd_addr);
  constexpr int W = static_cast<int>(kSnWidthMax);
  const uint32_t stride = 2u * w * (1u + nch);     // bytes of a row of the pivot table
  const uint32_t prow = tab + (nr - 1u) * stride;  // the right-hand-side row
  const uint32_t r = nr - w - 1u;
  const uint32_t c = lane < w ? lane : w - 1u;     // (idle lanes shadow the last column: same loads, no stores)
  const uint32_t b = lane < r ? lane : (r ? r - 1u : 0u);
  const uint32_t upd = tab + nr * stride, ustride = 2u * (3u + nch);
  const uint32_t ur = upd + __umul24((r * (r + 1u)) / 2u + b, ustride);
  // ---- trip 1: table words ----
  const uint32_t own = lds_ld16(prow + 2u * c);
  uint32_t cua[W - 1];
#pragma unroll
  for (int c2 = 0; c2 < W - 1; ++c2) {
    const uint32_t cc = static_cast<uint32_t>(c2) < c ? static_cast<uint32_t>(c2) : 0u;
    cua[c2] = lds_ld16(tab + c * stride + 2u * cc);  // U(c, c2): row c of finished column c2 (c2 < c; else unused)
  }
  uint32_t o = 0, pb = 0;
  if (r) {
    o = lds_ld16(ur);
    pb = lds_ld16(ur + 4u);
  }
  // ---- trip 2: values ----
  double v = lds_ld(own);
  const double inv = lds_ld(invd_addr + 8u * c);
  double cu[W - 1];
#pragma unroll
  for (int c2 = 0; c2 < W - 1; ++c2) cu[c2] = lds_ld(cua[c2]);
  double rb[W];
#pragma unroll
  for (int cc = 0; cc < W; ++cc) rb[cc] = (r && static_cast<uint32_t>(cc) < w) ? lds_ld(pb + mf_coff(static_cast<uint32_t>(cc), nr)) : 0.0;
  double acc = 0.0;
  for (uint32_t k = 1; k <= nch; ++k) {  // the children's values: of the row's entries, of its part of the update block
    const uint32_t ca = lds_ld16(prow + 2u * (w * k + c));
    const uint32_t sa = r ? lds_ld16(ur + 4u + 2u * k) : ca;
    v += lds_ld(ca);
    if (r) acc += lds_ld(sa);
  }
  // ---- the chain: column c2 is final once the columns before it are in ----
  double li[W];
#pragma unroll
  for (int c2 = 0; c2 < W; ++c2) {
    li[c2] = 0.0;
    if (static_cast<uint32_t>(c2) < w) {
      const double l = readlane_f64(v, c2) * readlane_f64(inv, c2);
      li[c2] = l;
      if (c2 < W - 1 && lane > static_cast<uint32_t>(c2) && lane < w) v = __builtin_fma(-l, cu[c2 < W - 1 ? c2 : 0], v);
    }
  }
  if (lane < w) lds_st(own, v);
  if (lane < r) {  // its part of the update block: entry (rhs row, R_lane)
#pragma unroll
    for (int cc = 0; cc < W; ++cc) acc = __builtin_fma(-li[cc], rb[cc], acc);
    if (root & 1u) coherent_store(&contrib[ext[o]], -acc, true);
    else lds_st(o, acc);
  }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void ldlt_mf_solve_kernel(LdltDev L, MfDev Mf, const double* __restrict__ Lx,
                                                                 const double* __restrict__ D, const double* __restrict__ rhs,
                                                                 double* __restrict__ contrib, double* __restrict__ xg,
                                                                 double* __restrict__ xg_next, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const uint32_t lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t task_index = blockIdx.x;
  const LdltTask t = L.tasks[task_index];
  const LdltMfTask m = Mf.tasks[task_index];
  const MfCarve cv = mf_carve(t, m);
  double* U = reinterpret_cast<double*>(smem_raw);
  double* arena = reinterpret_cast<double*>(smem_raw + cv.o_arena);
  double* invd = reinterpret_cast<double*>(smem_raw + cv.o_invd);
  double* x = reinterpret_cast<double*>(smem_raw + cv.o_x);
  const uint32_t* lvl = reinterpret_cast<const uint32_t*>(smem_raw + cv.o_lvl);
  const uint32_t* ext = reinterpret_cast<const uint32_t*>(smem_raw + cv.o_ext);
  const uint8_t* flags = reinterpret_cast<const uint8_t*>(smem_raw + cv.o_flags);
  const uint16_t* cent = reinterpret_cast<const uint16_t*>(smem_raw + cv.o_cent);
  const uint32_t* cptr = reinterpret_cast<const uint32_t*>(smem_raw + cv.o_cptr);
  const uint32_t* cidx = reinterpret_cast<const uint32_t*>(smem_raw + cv.o_cidx);
  const uint32_t* colperm = reinterpret_cast<const uint32_t*>(smem_raw + cv.o_cp);
  const uint32_t* anc = reinterpret_cast<const uint32_t*>(smem_raw + cv.o_anc);
  // ---- the task's tables (the step kernel's image, up to the KKT terms) ----
  {
    constexpr int kInFlight = kMfImageGroups / THREADS;
    uint4 stage_v[kInFlight];
    const uint4* src16 = Mf.image + static_cast<size_t>(task_index) * Mf.image_stride16;
#pragma unroll
    for (int k = 0; k < kInFlight; ++k) stage_v[k] = src16[tid + k * THREADS];
    const uint4 img = Mf.image_desc[task_index];
    uint4* dst_a = reinterpret_cast<uint4*>(smem_raw + cv.o_tab);
    const uint32_t n_a = (cv.o_terms - cv.o_tab) / 16u < img.y ? (cv.o_terms - cv.o_tab) / 16u : img.y;
#pragma unroll
    for (int k = 0; k < kInFlight; ++k) {
      const uint32_t i = tid + k * THREADS;
      if (i < n_a) dst_a[i] = stage_v[k];
    }
  }
  if (tid == 0) {
    arena[0] = 0.0;
    arena[1] = 0.0;
    x[t.n_col + m.n_anc] = 1.0;
  }
  __syncthreads();
  // ---- the factor back into the fronts' layout: U = L d, 1/d; the new right-hand side into its row ----
  {
    const uint32_t* g_out = L.ent_out + t.ent_off;
    const uint16_t* g_col = L.ent_col + t.ent_off;
    const int32_t* g_src = L.ent_src + t.ent_off;
    for (uint32_t i = tid; i < t.n_ent; i += THREADS) {
      const uint8_t fl = flags[i];
      const uint32_t o = g_out[i];
      if (fl & 1) {
        const double d = D[o];
        U[i] = d;
        invd[g_col[i]] = chain_reciprocal(d);
      } else if (fl & 4) {
        U[i] = rhs[g_src[i]];
      } else {
        U[i] = Lx[o] * D[colperm[g_col[i]]];
      }
    }
  }
  __syncthreads();
  // right-hand-side entries that take update slots from child tasks (ldlt_mf_step_kernel: the same slots, of
  // which this launch fills and empties only those of the right-hand-side row)
  for (uint32_t j = tid; j < m.n_cent; j += THREADS) {
    const uint32_t i = cent[j];
    if (!(flags[i] & 4)) continue;
    double acc = U[i];
    for (uint32_t c = cptr[j]; c < cptr[j + 1]; ++c) {
      double* p = &contrib[cidx[c]];
      const double v = slot_read(p);
      __hip_atomic_store(p, __longlong_as_double(static_cast<long long>(kSlotEmpty)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc -= v;
    }
    U[i] = acc;
  }
  __syncthreads();
  const uint32_t gfr = cv.o_fr;  // (LDS byte address of the task's front descriptors)
  for (uint32_t l = 0; l < t.n_lvl; ++l) {
    const uint32_t beg = __builtin_amdgcn_readfirstlane(lvl[l]), end = __builtin_amdgcn_readfirstlane(lvl[l + 1]);
    for (uint32_t q = beg + wave; q < end; q += THREADS / 64) {
      const u32x4 d = lds_load_desc(gfr + 16u * q);
      const uint32_t w = d[2] & 0xffu, nr = (d[2] >> 8) & 0xffu, nch = (d[2] >> 16) & 0xffu, root = (d[2] >> 24) & 3u;
      mf_fwd_front(cv.o_tab + 2u * d[0], w, nr, nch, root, cv.o_invd + 8u * (d[1] >> 16), ext + (d[3] >> 16), contrib, lane);
    }
    __syncthreads();
  }
  // ---- backward solve (ldlt_mf_step_kernel's) ----
  for (uint32_t a = tid; a < m.n_anc; a += THREADS) x[t.n_col + a] = slot_read(&xg[anc[a]]);
  __syncthreads();
  for (int l = static_cast<int>(t.n_lvl) - 1; l >= 0; --l) {
    const uint32_t beg = __builtin_amdgcn_readfirstlane(lvl[l]), end = __builtin_amdgcn_readfirstlane(lvl[l + 1]);
    for (uint32_t q = beg + wave; q < end; q += THREADS / 64) {
      const u32x4 d = lds_load_desc(gfr + 16u * q);
      const uint32_t w = d[2] & 0xffu, nr = (d[2] >> 8) & 0xffu, nch = (d[2] >> 16) & 0xffu, n_s = d[3] & 0xffffu;
      const uint32_t xr = cv.o_tab + 2u * (d[0] + nr * w * (1u + nch) + n_s * (3u + nch));
      mf_solve_front(xr, 8u * (d[1] & 0xffffu), w, nr, cv.o_invd + 8u * (d[1] >> 16), cv.o_x + 8u * (d[1] >> 16), lane);
    }
    __syncthreads();
  }
  for (uint32_t i = tid; i < t.n_col; i += THREADS) coherent_store(&xg[colperm[i]], x[i], true);
  for (uint32_t i = tid; i < t.n_col; i += THREADS)
    coherent_store(&xg_next[colperm[i]], __longlong_as_double(static_cast<long long>(kSlotEmpty)), true);
  for (uint32_t i = tid; i < t.n_col; i += THREADS) out[L.perm[colperm[i]]] = x[i];
}

}  // namespace slpx
