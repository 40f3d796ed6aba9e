#!/usr/bin/env python3
"""What a fresh process pays to bring the HIP runtime up (seconds): python profiles/hip_init_time.py"""
import ctypes
import time

t0 = time.perf_counter()
hip = ctypes.CDLL("libamdhip64.so")
t1 = time.perf_counter()
hip.hipInit(0)
t2 = time.perf_counter()
n = ctypes.c_int(0)
hip.hipGetDeviceCount(ctypes.byref(n))
t3 = time.perf_counter()
hip.hipSetDevice(0)
t4 = time.perf_counter()
hip.hipFree(None)
t5 = time.perf_counter()
p = ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(p), 1 << 20)
t6 = time.perf_counter()
print(f"dlopen {t1 - t0:.3f}  hipInit {t2 - t1:.3f}  hipGetDeviceCount {t3 - t2:.3f}  hipSetDevice {t4 - t3:.3f}  "
      f"hipFree(0) {t5 - t4:.3f}  first hipMalloc {t6 - t5:.3f}  devices {n.value}")
