// The task interpreter of the tape as a device function — free of host headers: compiled
// by hipcc into libslpx.so (tape_kernels.h wraps it in the interpreting kernels) and at run
// time by hipRTC as part of the prelude of the generated kernel (tape_jit.cpp), whose extra
// workgroups interpret the tasks that have no generated body IN THE SAME LAUNCH as the
// generated ones.
#pragma once

#include "tape_device.h"

namespace slpx {

// Cooperative global -> LDS copy of `n16` 16-byte groups, four loads in flight per lane.
template <int THREADS>
__device__ __forceinline__ void stage16(uint4* __restrict__ dst, const uint4* __restrict__ src,
                                        uint32_t n16, int tid) {
  uint32_t i = tid;
  for (; i + 3 * THREADS < n16; i += 4 * THREADS) {
    const uint4 a = src[i], b = src[i + THREADS], c = src[i + 2 * THREADS], d = src[i + 3 * THREADS];
    dst[i] = a;
    dst[i + THREADS] = b;
    dst[i + 2 * THREADS] = c;
    dst[i + 3 * THREADS] = d;
  }
  for (; i < n16; i += THREADS) dst[i] = src[i];
}

__device__ __forceinline__ uint32_t q16(uint32_t count, uint32_t per16) {
  return (count + per16 - 1) / per16;
}

// LDS layout (16-byte groups first, then doubles):
//   rec[n_node] 8 B | edges[n_edge] 4 B | eptr[n_slot+1] 2 B | lvl[n_lvl+1] 4 B |
//   slvl[n_slvl+1] 4 B | leaf_src[n_leaf] 4 B | val[n_leaf+n_node] f64 | part[2 n_node] f64 |
//   adj[n_slot] f64
#ifndef SLPX_TAPE_CLOCK
#define SLPX_TAPE_CLOCK(k)
#endif

// V[dst[k]] = scale >= 0 ? scales[scale] * from[src[k]] : from[src[k]] for the `count` outputs at
// `off`: kTapeUnroll per trip, every load unconditional (scale index clamped), so that the three
// binding words, the scale factor and the LDS value of four outputs are in flight together
// instead of one dependent round trip after the other.
// trips of the leaf / output loops cost a memory round trip (two for leaves) each, whatever
// their width: eight items per lane per trip
constexpr int kTapeUnroll = 8;

template <int THREADS>
__device__ __forceinline__ void tape_write_outputs(uint32_t count, uint32_t off, const int32_t* __restrict__ scale_idx,
                                                   const uint32_t* __restrict__ src, const uint32_t* __restrict__ dst,
                                                   const double* from, const double* __restrict__ scales,
                                                   double* __restrict__ V, int tid) {
  for (uint32_t i0 = tid; i0 < count; i0 += kTapeUnroll * THREADS) {
    int32_t sc[kTapeUnroll];
    uint32_t d[kTapeUnroll];
    double v[kTapeUnroll], w[kTapeUnroll];
#pragma unroll
    for (int u = 0; u < kTapeUnroll; ++u) {
      const uint32_t i = i0 + u * THREADS;
      const uint32_t k = off + (i < count ? i : i0);
      sc[u] = scale_idx[k];
      d[u] = dst[k];
      v[u] = from[src[k]];
    }
#pragma unroll
    for (int u = 0; u < kTapeUnroll; ++u) w[u] = scales[sc[u] >= 0 ? sc[u] : 0];
#pragma unroll
    for (int u = 0; u < kTapeUnroll; ++u)
      if (i0 + u * THREADS < count) V[d[u]] = sc[u] >= 0 ? w[u] * v[u] : v[u];
  }
}

// One task, start to finish, by a workgroup of THREADS lanes; `in` / `V` already point at
// the workgroup's problem, `smem_raw` at t.lds_bytes of 16-byte aligned LDS.
template <int THREADS, bool FULL_OPS>
__device__ __forceinline__ void tape_sweep_lds_body(const TapeDev& T, const TapeTask t,
                                                    const double* __restrict__ in,
                                                    const double* __restrict__ in_scale,
                                                    const double* __restrict__ scales, double* __restrict__ V,
                                                    int do_reverse, unsigned char* smem_raw) {
  const int tid = threadIdx.x;

  SLPX_TAPE_CLOCK(0);
  uint4* base = reinterpret_cast<uint4*>(smem_raw);
  const uint32_t g_rec = q16(t.n_node, 2), g_edge = q16(t.n_edge, 4), g_eptr = q16(t.n_slot + 1, 8),
                 g_lvl = q16(t.n_lvl + 1, 4), g_slvl = q16(t.n_slvl + 1, 4), g_leaf = q16(t.n_leaf, 4);
  uint4* s_rec = base;
  uint4* s_edge = s_rec + g_rec;
  uint4* s_eptr = s_edge + g_edge;
  uint4* s_lvl = s_eptr + g_eptr;
  uint4* s_slvl = s_lvl + g_lvl;
  uint4* s_leaf = s_slvl + g_slvl;
  double* val = reinterpret_cast<double*>(s_leaf + g_leaf);
  double* part = val + t.n_leaf + t.n_node;
  double* adj = part + 2 * t.n_node;
  const uint2* rec = reinterpret_cast<const uint2*>(s_rec);
  const uint32_t* edges = reinterpret_cast<const uint32_t*>(s_edge);
  const uint16_t* eptr = reinterpret_cast<const uint16_t*>(s_eptr);
  const uint32_t* lvl = reinterpret_cast<const uint32_t*>(s_lvl);
  const uint32_t* slvl = reinterpret_cast<const uint32_t*>(s_slvl);
  const uint32_t* leaf_src = reinterpret_cast<const uint32_t*>(s_leaf);

  // ---- stage the program (all offsets are multiples of 16 bytes) ----
  stage16<THREADS>(s_rec, reinterpret_cast<const uint4*>(T.node_rec16 + 4 * static_cast<size_t>(t.node_off)),
                   g_rec, tid);
  stage16<THREADS>(s_lvl, reinterpret_cast<const uint4*>(T.lvl_ptr + t.lvl_off), g_lvl, tid);
  stage16<THREADS>(s_leaf, reinterpret_cast<const uint4*>(T.leaf_src + t.leaf_off), g_leaf, tid);
  if (do_reverse && t.n_slot) {
    stage16<THREADS>(s_edge, reinterpret_cast<const uint4*>(T.edges16 + 2 * static_cast<size_t>(t.edge_off)),
                     g_edge, tid);
    stage16<THREADS>(s_eptr, reinterpret_cast<const uint4*>(T.slot_edge_ptr16 + t.slot_off), g_eptr, tid);
    stage16<THREADS>(s_slvl, reinterpret_cast<const uint4*>(T.slvl_ptr + t.slvl_off), g_slvl, tid);
  }
  __syncthreads();
  SLPX_TAPE_CLOCK(1);
  // Branch-free and kTapeUnroll leaves per trip: `flag ? consts[..] : in[..]` compiles to a divergent
  // branch around dependent loads, one memory round trip per trip of the loop (measured: 6 us
  // for a task of 640 leaves and ONE level).  All candidate operands are loaded (index 0
  // where a lane does not use one; consts is never empty), then selected.
  for (uint32_t i0 = tid; i0 < t.n_leaf; i0 += kTapeUnroll * THREADS) {
    uint32_t src[kTapeUnroll];
    double c[kTapeUnroll], x[kTapeUnroll], w[kTapeUnroll];
#pragma unroll
    for (int u = 0; u < kTapeUnroll; ++u) {
      const uint32_t i = i0 + u * THREADS;
      src[u] = leaf_src[i < t.n_leaf ? i : i0];
    }
#pragma unroll
    for (int u = 0; u < kTapeUnroll; ++u) {
      const bool is_const = (src[u] & kLeafConstFlag) != 0u;
      const uint32_t xi = is_const ? 0u : src[u];
      c[u] = T.consts[is_const ? (src[u] & ~kLeafConstFlag) : 0u];
      x[u] = in[xi];
      w[u] = in_scale[xi];
    }
#pragma unroll
    for (int u = 0; u < kTapeUnroll; ++u) {
      const uint32_t i = i0 + u * THREADS;
      if (i < t.n_leaf) val[i] = (src[u] & kLeafConstFlag) ? c[u] : x[u] * w[u];
    }
  }
  __syncthreads();
  SLPX_TAPE_CLOCK(2);

  // ---- forward ----
  // A single wave per SIMD makes this loop issue-bound, so the common ops (+ - neg *)
  // are evaluated branch-free with selects; only lanes holding another op enter the
  // switch.  The level bounds are fetched one level ahead.
  {
    uint32_t beg = lvl[0], end = t.n_lvl ? lvl[1] : 0;
    for (uint32_t l = 0; l < t.n_lvl; ++l) {
      const uint32_t next_end = lvl[l + 2 <= t.n_lvl ? l + 2 : t.n_lvl];
      for (uint32_t i = beg + tid; i < end; i += THREADS) {
        const uint2 r = rec[i];
        const uint32_t opf = r.x & 0xffffu, a0 = r.x >> 16, a1 = r.y & 0xffffu;
        const uint32_t o = opf & 0xff;
        const double lv = val[a0], rv = val[a1];
        const bool is_mul = o == OP_MUL, is_neg = o == OP_NEG, is_sub = o == OP_SUB;
        const double sum = lv + (is_sub ? -rv : rv);
        double v = is_mul ? lv * rv : (is_neg ? -lv : sum);
        double dl = is_mul ? rv : (is_neg ? -1.0 : 1.0);
        double dr = is_mul ? lv : (is_sub ? -1.0 : (is_neg ? 0.0 : 1.0));
        if (o > OP_MUL || o < OP_ADD)
          op_forward<FULL_OPS>(static_cast<Opcode>(o), lv, rv, (opf & 0x100) != 0, (opf & 0x200) != 0, v,
                               dl, dr);
        val[t.n_leaf + i] = v;
        part[2 * i] = dl;
        part[2 * i + 1] = dr;
      }
      __syncthreads();
      beg = end;
      end = next_end;
    }
  }

  SLPX_TAPE_CLOCK(3);
  // ---- value outputs (f, c_e, c_i) ----
  tape_write_outputs<THREADS>(t.n_vout, t.vout_off, T.vout_scale, T.vout_src, T.vout_dst, val, scales, V, tid);
  if (!do_reverse || t.n_slot == 0) return;
  SLPX_TAPE_CLOCK(4);

  // ---- adjoint gather ----
  {
    uint32_t beg = slvl[0], end = t.n_slvl ? slvl[1] : 0;
    for (uint32_t l = 0; l < t.n_slvl; ++l) {
      const uint32_t next_end = slvl[l + 2 <= t.n_slvl ? l + 2 : t.n_slvl];
      for (uint32_t i = beg + tid; i < end; i += THREADS) {
        const uint32_t eb = eptr[i], ee = eptr[i + 1];
        double acc = eb == ee ? 1.0 : 0.0;  // a slot without parents is a row root
        uint32_t e = eb;
        // two edges per trip: their four LDS reads are in flight together; the products
        // are still accumulated in edge order
        for (; e + 1 < ee; e += 2) {
          const uint32_t ed0 = edges[e], ed1 = edges[e + 1];
          const double a0 = adj[ed0 & 0xffffu], p0 = part[ed0 >> 16];
          const double a1 = adj[ed1 & 0xffffu], p1 = part[ed1 >> 16];
          acc += a0 * p0;
          acc += a1 * p1;
        }
        if (e < ee) {
          const uint32_t ed = edges[e];
          acc += adj[ed & 0xffffu] * part[ed >> 16];
        }
        adj[i] = acc;
      }
      __syncthreads();
      beg = end;
      end = next_end;
    }
  }
  SLPX_TAPE_CLOCK(5);

  // ---- Jacobian / Hessian entries ----
  tape_write_outputs<THREADS>(t.n_jout, t.jout_off, T.jout_scale, T.jout_slot, T.jout_dst, adj, scales, V, tid);
  SLPX_TAPE_CLOCK(6);
}

}  // namespace slpx
