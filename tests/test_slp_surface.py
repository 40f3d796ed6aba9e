"""Drop-in boundary, seam 1: the reference's benchmark program compiles against the
slp:: surface with its own include lines and spellings and links libslpx.so.

VERDICT r01: "`slp::Problem` is a plain class, so the reference benchmark does not compile against
it; no test compiles a C++ user program".  tests/support/user_program/cart_pole_user.cpp builds the
model of benchmarks/scalability/cart_pole/sleipnir.cpp:16-129 + rk4.hpp with every API spelling
that program uses, Eigen constants as slp::DenseMatrix."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
SRC = ROOT / "tests" / "support" / "user_program" / "cart_pole_user.cpp"
BIN = ROOT / "build" / "cart_pole_user"


def _fresh(out, src, slpx):
    """A user program built AFTER its source and after the library it links: the header-only slp::
    surface compiles library structs into the program, so a binary older than libslpx.so may not
    match it (seen: heap corruption after LdltOptions gained a member)."""
    return out.exists() and out.stat().st_mtime > max(src.stat().st_mtime, slpx.LIB_PATH.stat().st_mtime)


def build_user_program(slpx):
    BIN.parent.mkdir(parents=True, exist_ok=True)
    lib_dir = slpx.LIB_PATH.parent
    cmd = ["/opt/rocm/bin/hipcc", "-O1", "-std=c++23", "--offload-arch=gfx950", "-x", "hip", str(SRC), "-o", str(BIN),
           "-I" + str(ROOT / "include"), "-L" + str(lib_dir), "-lslpx", "-Wl,-rpath," + str(lib_dir)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]


def test_reference_benchmark_program_compiles_and_builds_the_model(slpx):
    build_user_program(slpx)
    N = 8
    res = subprocess.run([str(BIN), str(N)], capture_output=True, text=True, timeout=300)
    first = res.stdout.splitlines()[0]
    # sizes of SURVEY.md §8: n = 5N + 4, m_e = 4N + 8, m_i = 4N + 2; QUADRATIC cost, NONLINEAR
    # equalities, LINEAR inequalities (cart_pole_problem_test.cpp:87-89)
    assert first == f"n={5 * N + 4} m_e={4 * N + 8} m_i={4 * N + 2} cost=3 eq=4 ineq=2", res.stdout + res.stderr
    if slpx.lib().slpx_device_count() == 0:
        assert res.returncode == 3 and "no HIP device" in res.stdout  # no CPU fallback: it says so and stops


@pytest.mark.gpu
def test_reference_benchmark_program_solves_on_the_gpu(slpx):
    build_user_program(slpx) if not _fresh(BIN, SRC, slpx) else None
    res = subprocess.run([str(BIN), "100"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "status=0" in res.stdout, res.stdout + res.stderr


OCP_SRC = ROOT / "tests" / "support" / "user_program" / "flywheel_ocp_user.cpp"
OCP_BIN = ROOT / "build" / "flywheel_ocp_user"


def build_ocp_program(slpx):
    OCP_BIN.parent.mkdir(parents=True, exist_ok=True)
    lib_dir = slpx.LIB_PATH.parent
    cmd = ["/opt/rocm/bin/hipcc", "-O1", "-std=c++23", "--offload-arch=gfx950", "-x", "hip", str(OCP_SRC), "-o",
           str(OCP_BIN), "-I" + str(ROOT / "include"), "-L" + str(lib_dir), "-lslpx", "-Wl,-rpath," + str(lib_dir)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]


def test_ocp_helper_builds_the_models_of_the_reference_test(slpx):
    """slp::OCP (ocp.hpp:49-400) under the reference's include paths: the flywheel OCP of
    test/src/optimization/flywheel_ocp_test.cpp with every transcription method and both kinds
    of dynamics has a QUADRATIC cost, LINEAR equalities (none at all for single shooting) and
    LINEAR inequalities (:83-85)."""
    build_ocp_program(slpx)
    for method in (0, 1, 2):
        for kind in (0, 1):
            if method == 1 and kind == 1:
                continue  # direct collocation needs an explicit ODE (ocp.hpp:323)
            res = subprocess.run([str(OCP_BIN), str(method), str(kind), "50", "model-only"], capture_output=True,
                                 text=True, timeout=300)
            assert res.returncode == 0, res.stdout + res.stderr
            assert res.stdout.splitlines()[0] == "cost=3 eq=2 ineq=2", (method, kind, res.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("method,kind,steps", [(0, 0, 1000), (0, 1, 1000), (1, 0, 1000), (2, 0, 50), (2, 1, 50),
                                               (2, 0, 300), (2, 1, 1000)])
def test_ocp_helper_flywheel_known_answer_on_the_gpu(slpx, method, kind, steps):
    """flywheel_ocp_test.cpp:87-140: bang-then-hold input, states within 1e-2 of the discrete
    model's, final state r = 10 within 2e-6 — direct transcription and collocation at the
    reference's own size (1000 steps of 5 ms).  Single shooting makes the Hessian dense in the
    inputs; the reference then takes its DENSE LDLT branch (interior_point.hpp:340-349) — and so does
    the product where a column of L no longer fits a task of its sparse plan (r05: LdltPlan::dense,
    ldlt_dense_kernels.h; through r04 that was a refusal and the method was exercised at 50 steps only):
    300 steps, and the reference's own 1000."""
    build_ocp_program(slpx) if not _fresh(OCP_BIN, OCP_SRC, slpx) else None
    res = subprocess.run([str(OCP_BIN), str(method), str(kind), str(steps)], capture_output=True, text=True,
                         timeout=900)
    assert res.returncode == 0 and "status=0" in res.stdout, res.stdout + res.stderr


@pytest.mark.gpu
def test_ocp_helper_shared_timestep_is_a_hub_the_ordering_sets_aside(slpx):
    """TimestepMethod::VARIABLE_SINGLE: one decision variable in every dynamics row.  Level-set
    nested dissection cannot separate anything while it is in the graph (ldlt_symbolic.cpp: hubs
    are eliminated last); the larger timestep wins (the cost is a tracking error)."""
    build_ocp_program(slpx) if not _fresh(OCP_BIN, OCP_SRC, slpx) else None
    res = subprocess.run([str(OCP_BIN), "0", "0", "200", "shared-dt"], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "status=0" in res.stdout, res.stdout + res.stderr


DD_SRC = ROOT / "tests" / "support" / "user_program" / "differential_drive_ocp_user.cpp"
DD_BIN = ROOT / "build" / "differential_drive_ocp_user"


def build_dd_program(slpx):
    DD_BIN.parent.mkdir(parents=True, exist_ok=True)
    lib_dir = slpx.LIB_PATH.parent
    cmd = ["/opt/rocm/bin/hipcc", "-O1", "-std=c++23", "--offload-arch=gfx950", "-x", "hip", str(DD_SRC), "-o",
           str(DD_BIN), "-I" + str(ROOT / "include"), "-L" + str(lib_dir), "-lslpx", "-Wl,-rpath," + str(lib_dir)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]


def test_differential_drive_ocp_model_of_the_reference_test(slpx):
    """differential_drive_ocp_test.cpp:61-63: LINEAR cost (the sum of the timesteps), NONLINEAR
    equality constraints, LINEAR inequalities."""
    build_dd_program(slpx)
    res = subprocess.run([str(DD_BIN), "model-only"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.splitlines()[0] == "cost=2 eq=4 ineq=2", res.stdout + res.stderr


@pytest.mark.gpu
def test_differential_drive_minimum_time_ocp_on_the_gpu(slpx):
    """differential_drive_ocp_test.cpp:65-107: SUCCESS, initial and final state to 1e-8 — a
    nonlinear minimum-time problem whose single timestep variable sits in every dynamics row."""
    build_dd_program(slpx) if not _fresh(DD_BIN, DD_SRC, slpx) else None
    res = subprocess.run([str(DD_BIN)], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "status=0" in res.stdout, res.stdout + res.stderr


CPO_SRC = ROOT / "tests" / "support" / "user_program" / "cart_pole_ocp_user.cpp"
CPO_BIN = ROOT / "build" / "cart_pole_ocp_user"


def build_cart_pole_ocp_program(slpx):
    CPO_BIN.parent.mkdir(parents=True, exist_ok=True)
    lib_dir = slpx.LIB_PATH.parent
    cmd = ["/opt/rocm/bin/hipcc", "-O1", "-std=c++23", "--offload-arch=gfx950", "-x", "hip", str(CPO_SRC), "-o",
           str(CPO_BIN), "-I" + str(ROOT / "include"), "-L" + str(lib_dir), "-lslpx", "-Wl,-rpath," + str(lib_dir)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]


def test_cart_pole_ocp_model_of_the_reference_test(slpx):
    """cart_pole_ocp_test.cpp:87-89: QUADRATIC cost, NONLINEAR equalities, LINEAR inequalities
    (direct collocation, shared timestep variable, position bounds through for_each_step)."""
    build_cart_pole_ocp_program(slpx)
    res = subprocess.run([str(CPO_BIN), "model-only"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.splitlines()[0] == "cost=3 eq=4 ineq=2", res.stdout + res.stderr


@pytest.mark.gpu
def test_cart_pole_collocation_ocp_on_the_gpu(slpx):
    """cart_pole_ocp_test.cpp:91-131: SUCCESS, initial and final state to 1e-8."""
    build_cart_pole_ocp_program(slpx) if not _fresh(CPO_BIN, CPO_SRC, slpx) else None
    res = subprocess.run([str(CPO_BIN)], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "status=0" in res.stdout, res.stdout + res.stderr


# ---- the other problem tests of the reference (test/src/optimization/*_problem_test.cpp) ----
USER_DIR = ROOT / "tests" / "support" / "user_program"


def build_named_program(slpx, name):
    src, out = USER_DIR / f"{name}.cpp", ROOT / "build" / name
    if _fresh(out, src, slpx):
        return out
    out.parent.mkdir(parents=True, exist_ok=True)
    lib_dir = slpx.LIB_PATH.parent
    cmd = ["/opt/rocm/bin/hipcc", "-O1", "-std=c++23", "--offload-arch=gfx950", "-x", "hip", str(src), "-o", str(out),
           "-I" + str(ROOT / "include"), "-L" + str(lib_dir), "-lslpx", "-Wl,-rpath," + str(lib_dir), "-pthread"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    return out


def test_host_side_tests_of_the_reference_surface(slpx):
    """trivial_problem_test.cpp, decision_variable_test.cpp and constraints_test.cpp: 279 checks that
    need no device (an empty or cost-free problem is SUCCESS before anything is compiled,
    problem.hpp:304-313)."""
    exe = build_named_program(slpx, "small_surface_user")
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "failed=0" in res.stdout, res.stdout + res.stderr


# (program, arguments for the model only, expected types): cost / equality / inequality ExpressionType
PROBLEM_PROGRAMS = [
    ("arm_on_elevator_user", ["800", "model-only"], "cost=3 eq=2 ineq=4"),  # arm_on_elevator_problem_test.cpp:109-111
    ("double_integrator_user", ["model-only"], "cost=3 eq=2 ineq=2"),        # double_integrator_problem_test.cpp:80-82
    ("differential_drive_user", ["model-only"], "cost=3 eq=4 ineq=2"),       # differential_drive_problem_test.cpp:83-85
]


@pytest.mark.parametrize("name,args,types", PROBLEM_PROGRAMS, ids=[p[0] for p in PROBLEM_PROGRAMS])
def test_problem_models_of_the_reference_tests(slpx, name, args, types):
    exe = build_named_program(slpx, name)
    res = subprocess.run([str(exe), *args], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.splitlines()[0] == types, res.stdout + res.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("name,args", [("arm_on_elevator_user", ["800"]), ("double_integrator_user", []),
                                       ("differential_drive_user", [])], ids=lambda v: v if isinstance(v, str) else "")
def test_problem_tests_of_the_reference_on_the_gpu(slpx, name, args):
    """SUCCESS and the known answers of the reference's own tests: arm on elevator N = 800
    (arm_on_elevator_problem_test.cpp:113), the double integrator's bang-coast-bang profile
    (double_integrator_problem_test.cpp:84-131), the differential drive's states against the RK4
    model to 1e-8 (differential_drive_problem_test.cpp:87-119)."""
    exe = build_named_program(slpx, name)
    res = subprocess.run([str(exe), *args], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "status=0" in res.stdout, res.stdout + res.stderr


def test_matrix_surface_of_the_reference_unit_tests(slpx):
    """variable_matrix_test.cpp / slice_test.cpp with their own spellings: slp::Slice and `_`,
    strided views written through, compound assignment on views, iterators of views, the static
    constructors, cwise_reduce<T>, the free block() and solve() up to 5x5 — 143 checks, no device."""
    exe = build_named_program(slpx, "variable_matrix_user")
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "failed=0" in res.stdout, res.stdout + res.stderr


def test_multistart_program_builds_and_refuses_without_a_device(slpx):
    """slp::multistart (multistart.hpp:17-79) under the reference's include path; without a HIP
    device the solves say so (the exception crosses the std::async boundary), nothing falls back."""
    exe = build_named_program(slpx, "multistart_user")
    if slpx.lib().slpx_device_count() == 0:
        res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
        assert res.returncode == 3 and "no HIP device" in res.stdout, res.stdout + res.stderr


@pytest.mark.gpu
def test_multistart_of_the_reference_test_on_the_gpu(slpx):
    """multistart_test.cpp:16-53: Mishra's bird function from two starts on two threads; the better
    optimum (-3.13024680, -1.58214218) to 1e-8."""
    exe = build_named_program(slpx, "multistart_user")
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "status=0" in res.stdout and "failed_checks=0" in res.stdout, res.stdout + res.stderr


def test_derivative_classes_symbolic_side(slpx):
    """slp::Gradient / Jacobian / Hessian (gradient.hpp, jacobian.hpp, hessian.hpp) under the
    reference's include paths: get() is the gradient tree — known answers of the reference's unit
    tests, no device.  value() refuses without one."""
    exe = build_named_program(slpx, "derivatives_user")
    res = subprocess.run([str(exe), "symbolic"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "failed=0" in res.stdout, res.stdout + res.stderr
    if slpx.lib().slpx_device_count() == 0:
        res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
        assert res.returncode == 3 and "no HIP device" in res.stdout, res.stdout + res.stderr


@pytest.mark.gpu
def test_derivative_classes_values_from_the_device(slpx):
    """value(): the compiled tape on the GPU — jacobian_test.cpp (y = x, products, re-evaluation at
    new values), hessian_test.cpp (quadratic, sum of squares, product of sines; lower triangle
    only), gradient_test.cpp."""
    exe = build_named_program(slpx, "derivatives_user")
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "failed=0" in res.stdout, res.stdout + res.stderr


@pytest.mark.gpu
def test_spy_files_of_the_reference_test_on_the_gpu(slpx, tmp_path):
    """problem_spy_test.cpp:54-148: solve(options, spy = true) leaves H.spy, A_e.spy, A_i.spy — one
    record per iteration, coordinates with the signs of the entries — read back like the test does."""
    exe = build_named_program(slpx, "problem_spy_user")
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, cwd=tmp_path)
    assert res.returncode == 0 and "failed_checks=0" in res.stdout, res.stdout + res.stderr
    assert {"H.spy", "A_e.spy", "A_i.spy"} <= {f.name for f in tmp_path.iterdir()}
