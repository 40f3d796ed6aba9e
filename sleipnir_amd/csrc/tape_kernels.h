// tape_sweep: the AD refresh on gfx950.
//
// One workgroup executes one TASK (a bin-packed set of graph components, normally
// one direct-transcription stage) start to finish:
//   1. stage the task's PROGRAM (16-bit packed node records, adjoint edges, level
//      pointers, leaf bindings) into LDS with unrolled 16-byte loads — every slice
//      starts on a 16-byte boundary (tape_compiler.cpp), four loads per lane are in
//      flight, and after this the level loops never touch global memory;
//   2. forward sweep: one group of independent nodes per level; every node stores
//      its value and its two local partials;
//   3. adjoint sweep: per-(row, node) slots gathered level by level:
//      slot = Σ parent_slot · partial  (the reference's per-row append_triplets,
//      expression_graph.hpp:107-153, for all rows of the task at once);
//   4. scaled outputs go straight to their fixed positions in V.
// A 64-thread workgroup = one wavefront, so the per-level barrier is free; four
// such tasks fit the 160 KB LDS of a CU.
//
// Roofline: HBM-bound in the batched regime — algorithmic bytes per sweep are
// 8·(n + m_e + m_i) in + 8·(dynamic entries of V) out; a single
// problem is bound by graph depth × LDS latency instead.
#pragma once

#include <hip/hip_runtime.h>

#include "device.hpp"

namespace slpx {
// Phase clocks of the middle workgroup of the launch (debug aid, read with slpx_debug_tape_clocks): wall_clock64()
// ticks (100 MHz) at kernel entry, after staging, after leaves, after the forward levels,
// after the value outputs, after the adjoint levels, at exit.
__device__ unsigned long long g_tape_clocks[16];  // [0,8): 64-thread kernel, [8,16): 256-thread
#define SLPX_TAPE_CLOCK(k)                                                    \
  if (blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && threadIdx.x == 0) g_tape_clocks[(THREADS == 256 ? 8 : 0) + (k)] = wall_clock64()

}  // namespace slpx

#include "tape_interp.h"
#include "tape_ops.h"

namespace slpx {

// Separable sums (nlp.cpp): V[dst] = scale * (sum of the partials the tape tasks left in the
// hidden tail of V), in a fixed order — strided per lane, then a pairwise tree.  One
// 64-thread workgroup per sum.
__device__ __forceinline__ void tape_reduce_body(const NlpStructure::SumReduce r,
                                                 const double* __restrict__ scales, double* __restrict__ V,
                                                 double* part, int tid) {
  double acc = 0.0;
  for (int k = tid; k < r.count; k += 64) acc += V[r.src_off + k];
  part[tid] = acc;
  __syncthreads();
  for (int w = 32; w > 0; w >>= 1) {
    if (tid < w) part[tid] += part[tid + w];
    __syncthreads();
  }
  if (tid == 0) V[r.dst] = (r.scale_idx >= 0 ? scales[r.scale_idx] : 1.0) * part[0];
}

template <int THREADS, bool FULL_OPS>
__global__ __launch_bounds__(THREADS) void tape_sweep_lds_kernel(
    TapeDev T, const uint32_t* __restrict__ task_list, const double* __restrict__ in, int in_stride,
    const double* __restrict__ in_scale, const double* __restrict__ scales, double* __restrict__ V,
    int v_stride, int do_reverse) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int b = blockIdx.y;
  tape_sweep_lds_body<THREADS, FULL_OPS>(T, T.tasks[task_list[blockIdx.x]], in + static_cast<size_t>(b) * in_stride,
                                         in_scale, scales, V + static_cast<size_t>(b) * v_stride, do_reverse,
                                         smem_raw);
}

// Fallback for components too large for LDS: same program (32-bit records), working
// set in a global scratch buffer, one 1024-thread workgroup per task.
__global__ __launch_bounds__(1024) void tape_sweep_global_kernel(
    TapeDev T, const uint32_t* __restrict__ task_list, const double* __restrict__ in, int in_stride,
    const double* __restrict__ in_scale, const double* __restrict__ scales, double* __restrict__ V,
    int v_stride, double* __restrict__ scratch, unsigned long long scratch_stride, int do_reverse) {
  constexpr int THREADS = 1024;
  const TapeTask t = T.tasks[task_list[blockIdx.x]];
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  in += static_cast<size_t>(b) * in_stride;
  V += static_cast<size_t>(b) * v_stride;
  double* val = scratch + static_cast<size_t>(b) * scratch_stride + t.scratch_off;
  double* part = val + t.n_leaf + t.n_node;
  double* adj = part + 2 * t.n_node;

  for (uint32_t i = tid; i < t.n_leaf; i += THREADS) {
    const uint32_t src = T.leaf_src[t.leaf_off + i];
    val[i] = (src & kLeafConstFlag) ? T.consts[src & ~kLeafConstFlag] : in[src] * in_scale[src];
  }
  __syncthreads();
  const uint32_t* lvl = T.lvl_ptr + t.lvl_off;
  const uint32_t* rec = T.node_rec + 3 * static_cast<size_t>(t.node_off);
  for (uint32_t l = 0; l < t.n_lvl; ++l) {
    const uint32_t beg = lvl[l], end = lvl[l + 1];
    for (uint32_t i = beg + tid; i < end; i += THREADS) {
      const uint32_t r0 = rec[3 * i], a0 = rec[3 * i + 1], a1 = rec[3 * i + 2];
      double v, dl, dr;
      op_forward(static_cast<Opcode>(r0 & 0xff), val[a0], val[a1], (r0 & 0x100) != 0,
                 (r0 & 0x200) != 0, v, dl, dr);
      val[t.n_leaf + i] = v;
      part[2 * i] = dl;
      part[2 * i + 1] = dr;
    }
    __syncthreads();
  }
  for (uint32_t i = tid; i < t.n_vout; i += THREADS) {
    const uint32_t k = t.vout_off + i;
    const int32_t sc = T.vout_scale[k];
    const double v = val[T.vout_src[k]];
    V[T.vout_dst[k]] = sc >= 0 ? scales[sc] * v : v;
  }
  if (!do_reverse || t.n_slot == 0) return;
  const uint32_t* slvl = T.slvl_ptr + t.slvl_off;
  const uint32_t* eptr = T.slot_edge_ptr + t.slot_off;
  const TapeEdge* edges = T.edges + t.edge_off;
  for (uint32_t l = 0; l < t.n_slvl; ++l) {
    const uint32_t beg = slvl[l], end = slvl[l + 1];
    for (uint32_t i = beg + tid; i < end; i += THREADS) {
      const uint32_t eb = eptr[i], ee = eptr[i + 1];
      double acc = eb == ee ? 1.0 : 0.0;
      for (uint32_t e = eb; e < ee; ++e) acc += adj[edges[e].parent_slot] * part[edges[e].partial];
      adj[i] = acc;
    }
    __syncthreads();
  }
  for (uint32_t i = tid; i < t.n_jout; i += THREADS) {
    const uint32_t k = t.jout_off + i;
    const int32_t sc = T.jout_scale[k];
    const double v = adj[T.jout_slot[k]];
    V[T.jout_dst[k]] = sc >= 0 ? scales[sc] * v : v;
  }
}

}  // namespace slpx
