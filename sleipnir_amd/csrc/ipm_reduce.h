// Reduction helpers of the interior-point iteration kernels (ipm_kernels.h, restoration.hip): wave totals by DPP
// butterflies + scalar lane reads (no LDS traffic), workgroup totals through a few LDS words, a fixed combination
// order (reproducible run to run), and the sequence number a chain of launches hands the host at its end.
#pragma once

#include <hip/hip_runtime.h>

#include "device.hpp"

namespace slpx {

constexpr int kIpmThreads = 1024;

enum IpmOp { IPM_SUM = 0, IPM_MAX = 1, IPM_MIN = 2 };

template <int kCtrl>
__device__ __forceinline__ double ipm_dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ipm_readlane(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                          __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double ipm_combine(int op, double a, double b) {
  return op == IPM_SUM ? a + b : (op == IPM_MAX ? fmax(a, b) : fmin(a, b));
}
// Total of a full 64-lane wave, in every lane: a DPP butterfly inside each row of 16 lanes
// (no LDS traffic — with one workgroup all waves share one CU's LDS pipe), then the four
// row totals through scalar registers.
__device__ __forceinline__ double wave_reduce(double v, int op) {
  v = ipm_combine(op, v, ipm_dpp<0xB1>(v));   // quad_perm [1,0,3,2]
  v = ipm_combine(op, v, ipm_dpp<0x4E>(v));   // quad_perm [2,3,0,1]
  v = ipm_combine(op, v, ipm_dpp<0x141>(v));  // row_half_mirror
  v = ipm_combine(op, v, ipm_dpp<0x140>(v));  // row_mirror
  const double r0 = ipm_readlane(v, 0), r1 = ipm_readlane(v, 16), r2 = ipm_readlane(v, 32),
               r3 = ipm_readlane(v, 48);
  return ipm_combine(op, ipm_combine(op, r0, r1), ipm_combine(op, r2, r3));
}

// sum over the eight lanes 8 k .. 8 k + 7 of a wave, in every one of them (two quad permutes and a half-row mirror)
__device__ __forceinline__ double ipm_group8_sum(double v) {
  v += ipm_dpp<0xB1>(v);
  v += ipm_dpp<0x4E>(v);
  v += ipm_dpp<0x141>(v);
  return v;
}

// Reduces NQ per-lane quantities (op per quantity) over a workgroup of THREADS lanes (full
// waves, every lane must call); the results land in `vals` of every lane.  `scratch` =
// (THREADS / 64 + 1) x NQ doubles of LDS.
template <int NQ, int THREADS>
__device__ __forceinline__ void block_reduce(double (&vals)[NQ], const int (&ops)[NQ], double* scratch) {
  constexpr int kWaves = THREADS / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const double v = wave_reduce(vals[q], ops[q]);
    if (lane == 0) scratch[wave * NQ + q] = v;
  }
  __syncthreads();
  if (threadIdx.x < NQ) {
    const int q = threadIdx.x;
    int op = ops[0];
#pragma unroll
    for (int k = 1; k < NQ; ++k)
      if (q == k) op = ops[k];
    double v = scratch[q];
    for (int w = 1; w < kWaves; ++w) v = ipm_combine(op, v, scratch[w * NQ + q]);
    scratch[kWaves * NQ + q] = v;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NQ; ++q) vals[q] = scratch[kWaves * NQ + q];
  __syncthreads();
}

// Last act of a chain of launches whose results sit in pinned host memory: bump the
// sequence number the host spins on (DeviceNlp::wait_published) — it sees it a few
// microseconds before the stream reports the kernel complete.
__device__ __forceinline__ void ipm_publish(unsigned long long* seq_dev, volatile unsigned long long* seq_host) {
  __threadfence_system();
  const unsigned long long v = *seq_dev + 1;
  *seq_dev = v;
  *seq_host = v;
}

}  // namespace slpx
