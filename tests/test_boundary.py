"""The drop-in boundary (CPU tier, no compute): libslpx.so loads, exports every symbol
include/slpx.h declares, carries a gfx950 code object, and the product neither links nor
mentions the oracle; without a HIP device it fails loudly instead of falling back."""
import re
import subprocess
from pathlib import Path

import pytest
from tests.support import models

ROOT = Path(__file__).resolve().parents[1]
HEADER = ROOT / "include" / "slpx.h"


def declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(slpx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(slpx):
    lib = slpx.lib()
    names = declared_symbols()
    assert len(names) >= 45, names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.slpx_abi_version() == slpx.ABI_VERSION == 6  # (the header the Python binding's struct layouts were written against)


def test_header_cites_the_reference_interfaces():
    text = HEADER.read_text()
    for needle in ("problem.hpp", "regularized_ldlt.hpp", "interior_point.hpp", "jacobian.hpp",
                   "hessian.hpp", "variable.hpp"):
        assert needle in text, needle


def test_library_carries_gfx950_code_object():
    data = (ROOT / "sleipnir_amd" / "libslpx.so").read_bytes()
    assert b"gfx950" in data
    for kernel in (b"tape_sweep_lds_kernel", b"kkt_assemble_kernel", b"kkt_rhs_kernel",
                   b"ldlt_factor_kernel", b"ldlt_fwd_kernel", b"ldlt_bwd_kernel", b"step_backsub_kernel"):
        assert kernel in data, kernel


def test_product_does_not_reference_oracle():
    pkg = ROOT / "sleipnir_amd"
    for path in list(pkg.rglob("*.cpp")) + list(pkg.rglob("*.hpp")) + list(pkg.rglob("*.h")) + \
            list(pkg.rglob("*.hip")) + list(pkg.rglob("*.py")) + list(pkg.rglob("Makefile")):
        text = path.read_text()
        assert "oracle/" not in text and "liboracle" not in text and "orc_" not in text, path
    out = subprocess.run(["ldd", str(pkg / "libslpx.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out
    assert "torch" not in out  # C-ABI: no torch types, no torch linkage


def test_product_library_holds_no_benchmark_model(slpx):
    """the reference's benchmark programs are fixtures (tests/support/models/),
    not product source — libslpx.so exports nothing named after them and its sources do not hold them."""
    pkg = Path(slpx.__file__).resolve().parent
    out = subprocess.run(["nm", "-D", "--defined-only", str(pkg / "libslpx.so")], capture_output=True, text=True).stdout
    assert out and not re.search(r"cart_pole|flywheel", out)
    for src in (pkg / "csrc").rglob("*"):
        if src.is_file() and src.suffix in (".cpp", ".hpp", ".h", ".hip"):
            assert not re.search(r"cart_pole_dynamics|build_cart_pole|build_flywheel", src.read_text()), src


def test_no_cpu_fallback(slpx, fresh):
    """Without a HIP device the compiled Newton system cannot be created and solve()
    reports a library error; nothing is computed on the CPU instead."""
    if slpx.lib().slpx_device_count() > 0:
        pytest.skip("HIP device present")
    p = models.flywheel(5, 1.0)
    with pytest.raises(slpx.SlpxError):
        slpx.System(p)
    with pytest.raises(slpx.SlpxError):
        p.solve()
    with pytest.raises(slpx.SlpxError):
        p.system()  # slpx_problem_system compiles for the device as well
    p.close()
    # the linear-solver seam on its own (slpx_ldlt_create): 2x2 lower-triangular pattern
    with pytest.raises(slpx.SlpxError):
        slpx.System.linear_solver(1, 1, [0, 2, 3], [0, 1, 1])
