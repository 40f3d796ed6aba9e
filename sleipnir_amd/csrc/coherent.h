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

}  // namespace slpx
