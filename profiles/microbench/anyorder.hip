// Can a kernel start while its predecessor IN THE SAME STREAM still runs?  hipExtLaunchKernel(...,
// hipExtAnyOrderLaunch) clears the AQL barrier bit of the dispatch packet where the runtime honours it.
// Kernel A (blocks on some CUs) spins ~30 us; kernel B records when its first workgroup starts.
//   make -C profiles/microbench anyorder_bin && profiles/microbench/anyorder_bin
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void spin_kernel(unsigned long long* stamps, int slot, long long spin_ticks) {
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * slot] = t0;
  while (static_cast<long long>(wall_clock64() - t0) < spin_ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * slot + 1] = wall_clock64();
}

int main() {
  unsigned long long* stamps;
  (void)hipHostMalloc(&stamps, 64 * sizeof(unsigned long long));
  hipStream_t s1, s2;
  (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  const long long ticks = 3000;  // wall clock: 100 MHz -> 30 us
  auto report = [&](const char* what) {
    (void)hipDeviceSynchronize();
    const double a0 = 0.0, a1 = (stamps[1] - stamps[0]) / 100.0, b0 = (static_cast<long long>(stamps[2] - stamps[0])) / 100.0,
                 b1 = (static_cast<long long>(stamps[3] - stamps[0])) / 100.0;
    std::printf("%-58s A [%5.1f, %5.1f] us   B [%5.1f, %5.1f] us   B starts %+.1f us after A ends\n", what, a0, a1, b0, b1, b0 - a1);
  };
  for (int rep = 0; rep < 3; ++rep) {
    spin_kernel<<<64, 256, 0, s1>>>(stamps, 0, ticks);
    spin_kernel<<<64, 256, 0, s1>>>(stamps, 1, 300);
    report("same stream, plain launches");
    {
      spin_kernel<<<64, 256, 0, s1>>>(stamps, 0, ticks);
      int slot = 1;
      long long t = 300;
      void* args[] = {&stamps, &slot, &t};
      const hipError_t e = hipExtLaunchKernel(reinterpret_cast<const void*>(&spin_kernel), dim3(64), dim3(256), args, 0, s1, nullptr, nullptr,
                                              hipExtAnyOrderLaunch);
      if (e != hipSuccess) std::printf("hipExtLaunchKernel: %s\n", hipGetErrorString(e));
      report("same stream, second launch with hipExtAnyOrderLaunch");
    }
    spin_kernel<<<64, 256, 0, s1>>>(stamps, 0, ticks);
    spin_kernel<<<64, 256, 0, s2>>>(stamps, 1, 300);
    report("two streams");
  }
  return 0;
}
