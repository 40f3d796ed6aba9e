"""The benchmark models (cart-pole, flywheel) for tests, bench.py and the profiling scripts —
TEST / BENCH FIXTURE, not part of the product.

tests/support/models/bench_models.{hpp,cpp} write the reference's scalability-benchmark models as
user programs of the ``slp::`` surface and hand them to libslpx.so through its public C-ABI; this
module builds that fixture library (tests/support/libslpx_models.so, in-tree, so that it travels
to the GPU box) and returns ``sleipnir_amd.Problem`` handles.  The product library itself holds no
model (``nm -D libslpx.so | grep cart_pole`` is empty; tests/test_boundary.py checks it).
"""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import sleipnir_amd

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
SRC_DIR = HERE / "models"
LIB_PATH = HERE / "libslpx_models.so"

# horizons whose generated tape kernels build() ships in sleipnir_amd/jit_cache/
PREBUILT_MODELS = (("cart_pole", 1000), ("cart_pole", 500), ("cart_pole", 5000), ("cart_pole", 100),
                   ("cart_pole", 50), ("cart_pole", 300))  # (300: bench.py's whole solves; its restoration system has a body of its own)


def _sources():
    return [SRC_DIR / "bench_models.cpp", SRC_DIR / "bench_models.hpp"]


def build() -> Path:
    sleipnir_amd.build()
    lib_dir = sleipnir_amd.LIB_PATH.parent
    cmd = ["/opt/rocm/bin/hipcc", "-O1", "-std=c++23", "-fPIC", "-shared", "--offload-arch=gfx950", "-x", "hip",
           str(SRC_DIR / "bench_models.cpp"), "-o", str(LIB_PATH), "-I" + str(ROOT / "include"),
           "-L" + str(lib_dir), "-lslpx", "-Wl,-rpath," + str(lib_dir)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libslpx_models.so failed:\n" + res.stdout[-3000:] + res.stderr[-3000:])
    return LIB_PATH


def _stale() -> bool:
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    # (the header-only slp:: surface compiles library structs into its users: rebuild after libslpx.so)
    return t < max(s.stat().st_mtime for s in _sources()) or t < sleipnir_amd.LIB_PATH.stat().st_mtime


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    sleipnir_amd.lib()  # the same expression arena
    if _stale():
        build()
    L = ctypes.CDLL(str(LIB_PATH))
    for name in ("bench_models_cart_pole", "bench_models_flywheel"):
        fn = getattr(L, name)
        fn.restype = ctypes.c_void_p
        fn.argtypes = [ctypes.c_int32, ctypes.c_double]
    L.bench_models_last_error.restype = ctypes.c_char_p
    _lib = L
    return L


def _wrap(handle) -> "sleipnir_amd.Problem":
    if not handle:
        raise sleipnir_amd.SlpxError(lib().bench_models_last_error().decode())
    return sleipnir_amd.Problem(handle)


def cart_pole(N: int, dt: float) -> "sleipnir_amd.Problem":
    """benchmarks/scalability/cart_pole/sleipnir.cpp:76-129."""
    return _wrap(lib().bench_models_cart_pole(N, dt))


def flywheel(N: int, dt: float) -> "sleipnir_amd.Problem":
    """benchmarks/scalability/flywheel/sleipnir.cpp:12-42."""
    return _wrap(lib().bench_models_flywheel(N, dt))


def prebuild_kernels(which=PREBUILT_MODELS) -> int:
    """Code objects of the generated tape kernels of the BASELINE models into
    sleipnir_amd/jit_cache/ (built artefacts like libslpx.so: they travel with the tree, not with
    the history).  hipRTC cross-compiles for gfx950 without a device; ~1 s per model, skipped for
    code objects that are already there."""
    total = 0
    for kind, N in which:
        sleipnir_amd.lib().slpx_graph_reset()
        p = cart_pole(N, 5.0 / N) if kind == "cart_pole" else flywheel(N, 5.0 / N)
        total += p.prebuild_kernels()
        p.close()
    sleipnir_amd.lib().slpx_graph_reset()
    return total
