// Values that cross workgroups INSIDE one launch (update blocks, finished x of ancestor tasks,
// the KKT system when its assembly rides in the factorization's launch) are moved with
// agent-scope relaxed atomics: they bypass the non-coherent cache levels — every XCD has its
// own L2 — so a hand-over needs no L2 write-back / invalidate (an agent-scope fence does both
// to the whole XCD L2 and slows every workgroup on it: measured 29 -> 76 us for the backward
// solve).  (`in_launch` false = the kernel boundary orders everything: plain cached accesses.)
#pragma once

#include <hip/hip_runtime.h>

#include "device.hpp"

namespace slpx {

__device__ __forceinline__ double coherent_load(const double* p, bool in_launch) {
  return in_launch ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
__device__ __forceinline__ void coherent_store(double* p, double v, bool in_launch) {
  if (in_launch) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// Producer / consumer hand-over between the workgroups of one launch (device.hpp: KktFuse,
// BacksubFuse): producers add one to cnt[0] once their (coherent) stores are acknowledged,
// consumers spin until it reaches the number of producers; every consumer adds one to cnt[1]
// when it is through and the last of them clears both words for the next launch.
__device__ __forceinline__ void done_signal(unsigned int* cnt) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a (never expected) time-out shows up as "bad pivots" instead of hanging the GPU
__device__ __forceinline__ void done_wait(const unsigned int* cnt, unsigned int target, LdltStats* stats_b) {
  if (threadIdx.x == 0) {
    unsigned int spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 24)) {
        if (stats_b != nullptr) atomicAdd(&stats_b->n_bad, 1 << 20);
        break;
      }
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void done_consumed(unsigned int* cnt, unsigned int n_consumers) {
  if (threadIdx.x == 0) {
    const unsigned int old = __hip_atomic_fetch_add(&cnt[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == n_consumers) {
      __hip_atomic_store(&cnt[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&cnt[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace slpx
