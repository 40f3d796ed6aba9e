"""The reference's own autodiff unit tests, re-expressed on tests/support/model.py and
run against (a) the ORACLE — this is what pins the oracle to the reference's golden
values — and (b) the PRODUCT pipeline (graph -> NLP structure -> compiled tape,
interpreted on the host in the CPU tier, and executed by the HIP kernels through the
C-ABI in the `-m gpu` tier: fixture parameter `product_gpu`).

Sources (values and expressions taken from there, `==` kept where the reference
uses exact equality):
  test/src/autodiff/gradient_test.cpp:19-838
  test/src/autodiff/jacobian_test.cpp:13-247
  test/src/autodiff/hessian_test.cpp:22-509
"""
import math

import numpy as np
import pytest

from tests.support import model

EXACT = 0.0
# The product stores local partials (1/r, -l/r^2, ...) and multiplies by the adjoint,
# where the reference divides the adjoint; results may differ in the last ulp.
ULP = 4e-16


@pytest.fixture(params=["oracle", "product", pytest.param("product_gpu", marks=pytest.mark.gpu)])
def m(request, fresh):
    if request.param == "oracle":
        be = model.OracleBackend()
    elif request.param == "product":
        be = model.ProductBackend("hostcheck")  # compiled plans interpreted on the host
    else:
        be = model.ProductBackend("gpu")  # the HIP kernels through the C-ABI
    be.reset()
    mm = model.Model(be)
    mm.tol = EXACT if request.param == "oracle" else ULP
    mm.is_oracle = request.param == "oracle"
    return mm


def close(a, b, tol):
    return abs(a - b) <= tol * max(1.0, abs(b))


def check_grad(m, f, x, expected, tol=None):
    tol = m.tol if tol is None else max(tol, m.tol)
    assert close(m.gradient(f, x)[0], expected, tol), (m.gradient(f, x)[0], expected)
    assert close(m.gradient_symbolic(f, x)[0], expected, tol), (m.gradient_symbolic(f, x)[0], expected)


# ---- gradient_test.cpp ---------------------------------------------------------

def test_gradient_trivial_case(m):  # :19-36
    a, b = m.variable(10), m.variable(20)
    c = a
    assert m.gradient(a, a)[0] == 1.0
    assert m.gradient(a, b)[0] == 0.0
    assert m.gradient(c, a)[0] == 1.0
    assert m.gradient(c, b)[0] == 0.0


def test_gradient_unary_plus_minus(m):  # :38-66
    a = m.variable(10)
    c = +a
    assert c.value() == a.value()
    assert m.gradient(c, a)[0] == 1.0
    c = -a
    assert c.value() == -a.value()
    assert m.gradient(c, a)[0] == -1.0


def test_gradient_identical_variables(m):  # :68-85
    a = m.variable(10)
    x = a
    c = a * a + x
    assert c.value() == a.value() * a.value() + x.value()
    assert m.gradient(c, a)[0] == 2 * a.value() + m.gradient(x, a)[0]
    assert m.gradient(c, x)[0] == 2 * a.value() * m.gradient(a, x)[0] + 1


def test_gradient_elementary(m):  # :87-122
    a, b = m.variable(1), m.variable(2)
    assert m.gradient(-2 * a, a)[0] == -2.0
    assert close(m.gradient(a / 3.0, a)[0], 1.0 / 3.0, m.tol)
    a.set_value(100)
    b.set_value(200)
    assert list(m.gradient(a + b, [a, b])) == [1.0, 1.0]
    assert list(m.gradient(a - b, [a, b])) == [1.0, -1.0]
    assert list(m.gradient(-a + b, [a, b])) == [-1.0, 1.0]
    assert m.gradient(a + 1, a)[0] == 1.0


def test_gradient_trigonometry(m):  # :191-249
    x = m.variable(0.5)
    v = 0.5
    assert m.sin(x).value() == math.sin(v)
    check_grad(m, m.sin(x), x, math.cos(v))
    assert m.cos(x).value() == math.cos(v)
    check_grad(m, m.cos(x), x, -math.sin(v))
    assert m.tan(x).value() == math.tan(v)
    check_grad(m, m.tan(x), x, 1.0 / (math.cos(v) * math.cos(v)))
    assert m.asin(x).value() == math.asin(v)
    check_grad(m, m.asin(x), x, 1.0 / math.sqrt(1 - v * v))
    assert m.acos(x).value() == math.acos(v)
    check_grad(m, m.acos(x), x, -1.0 / math.sqrt(1 - v * v))
    assert m.atan(x).value() == math.atan(v)
    check_grad(m, m.atan(x), x, 1.0 / (1 + v * v))


def test_gradient_hyperbolic(m):  # :251-284
    x = m.variable(1)
    check_grad(m, m.sinh(x), x, math.cosh(1.0))
    check_grad(m, m.cosh(x), x, math.sinh(1.0))
    check_grad(m, m.tanh(x), x, 1.0 / (math.cosh(1.0) * math.cosh(1.0)))
    assert m.tanh(x).value() == math.tanh(1.0)


def test_gradient_exponential(m):  # :286-319
    x = m.variable(1)
    assert m.log(x).value() == math.log(1.0)
    check_grad(m, m.log(x), x, 1.0)
    check_grad(m, m.log10(x), x, 1.0 / (math.log(10.0) * 1.0))
    assert m.exp(x).value() == math.exp(1.0)
    check_grad(m, m.exp(x), x, math.exp(1.0))


def test_gradient_power(m):  # :321-427
    x, a = m.variable(1), m.variable(2)
    y = 2 * a
    check_grad(m, m.sqrt(x), x, 0.5 / math.sqrt(1.0))
    check_grad(m, m.sqrt(a), a, 0.5 / math.sqrt(2.0))
    cb = lambda v: math.copysign(abs(v) ** (1.0 / 3.0), v)  # noqa: E731
    check_grad(m, m.cbrt(x), x, 1.0 / (3.0 * 1.0 * 1.0))
    c2 = m.cbrt(a).value()
    check_grad(m, m.cbrt(a), a, 1.0 / (3.0 * c2 * c2), 1e-15)
    assert m.pow(x, 2.0).value() == 1.0
    check_grad(m, m.pow(x, 2.0), x, 2.0)
    assert m.pow(2.0, x).value() == 2.0
    check_grad(m, m.pow(2.0, x), x, math.log(2.0) * 2.0)
    check_grad(m, m.pow(x, x), x, (math.log(1.0) + 1) * 1.0)
    assert y.value() == 4.0
    check_grad(m, y, a, 2.0)
    check_grad(m, m.pow(x, y), x, 4.0 / 1.0 * 1.0)
    check_grad(m, m.pow(x, y), a, 1.0 * (4.0 / 1.0 * 0.0 + math.log(1.0) * 2.0))
    check_grad(m, m.pow(x, y), y, math.log(1.0) * 1.0)


def test_gradient_abs(m):  # :429-467
    x = m.variable(0)
    for v, g in ((1, 1.0), (-1, -1.0), (0, 0.0)):
        x.set_value(v)
        assert m.abs(x).value() == abs(v)
        check_grad(m, m.abs(x), x, g)
    f = m.abs(x * x - 4)
    x.set_value(3)
    check_grad(m, f, x, 6.0)
    x.set_value(1)
    check_grad(m, f, x, -2.0)


def test_gradient_atan2(m):  # :469-576
    x, y = m.variable(1), m.variable(0.9)
    assert m.atan2(2.0, x).value() == math.atan2(2.0, 1.0)
    check_grad(m, m.atan2(2.0, x), x, -2.0 / (4.0 + 1.0), 1e-15)
    x.set_value(-2)
    assert m.atan2(0.0, x).value() == math.atan2(0.0, -2.0)
    check_grad(m, m.atan2(0.0, x), x, 0.0)
    x.set_value(1)
    check_grad(m, m.atan2(x, 2.0), x, 2.0 / (4.0 + 1.0), 1e-15)
    x.set_value(-2)
    check_grad(m, m.atan2(x, 0.0), x, 0.0)
    x.set_value(1.1)
    xv, yv = 1.1, 0.9
    assert m.atan2(y, x).value() == math.atan2(yv, xv)
    check_grad(m, m.atan2(y, x), y, xv / (xv * xv + yv * yv), 1e-15)
    check_grad(m, m.atan2(y, x), x, -yv / (xv * xv + yv * yv), 1e-15)
    f = 3 * m.atan2(m.sin(y), 2 * x + 1)
    assert close(f.value(), 3 * math.atan2(math.sin(yv), 2 * xv + 1), 1e-15)
    den = (2 * xv + 1) * (2 * xv + 1) + math.sin(yv) * math.sin(yv)
    check_grad(m, f, y, 3 * (2 * xv + 1) * math.cos(yv) / den, 1e-15)
    check_grad(m, f, x, 3 * -2 * math.sin(yv) / den, 1e-15)


def test_gradient_hypot(m):  # :578-676
    x, y = m.variable(1.8), m.variable(1.5)
    assert m.hypot(x, 2.0).value() == math.hypot(1.8, 2.0)
    check_grad(m, m.hypot(x, 2.0), x, 1.8 / math.hypot(1.8, 2.0))
    x.set_value(-1)
    assert m.hypot(x, 0.0).value() == 1.0
    check_grad(m, m.hypot(x, 0.0), x, -1.0)
    check_grad(m, m.hypot(2.0, y), y, 1.5 / math.hypot(2.0, 1.5))
    y.set_value(-2)
    check_grad(m, m.hypot(0.0, y), y, -1.0)
    x.set_value(1.3)
    y.set_value(2.3)
    h = math.hypot(1.3, 2.3)
    assert m.hypot(x, y).value() == h
    check_grad(m, m.hypot(x, y), x, 1.3 / h)
    check_grad(m, m.hypot(x, y), y, 2.3 / h)
    h2 = math.hypot(2 * 1.3, 3 * 2.3)
    check_grad(m, m.hypot(2 * x, 3 * y), x, 4 * 1.3 / h2, 1e-15)
    check_grad(m, m.hypot(2 * x, 3 * y), y, 9 * 2.3 / h2, 1e-15)
    z = m.variable(3.3)
    h3 = model.py_hypot3(1.3, 2.3, 3.3)
    f = m.hypot(x, y, z)
    assert close(f.value(), h3, 1e-15)
    for var, v in ((x, 1.3), (y, 2.3), (z, 3.3)):
        check_grad(m, f, var, v / h3, 1e-15)


def test_gradient_max_min(m):  # :678-738
    x = m.variable(2)
    x2, x3 = x * x, x * x * x
    g3 = m.gradient(x3, x)[0]
    g2 = m.gradient(x2, x)[0]
    assert m.max(x2, x3).value() == x3.value()
    check_grad(m, m.max(x2, x3), x, g3)
    check_grad(m, m.max(x3, x2), x, g3)
    assert m.max(x, x).value() == x.value()
    check_grad(m, m.max(x, x), x, 1.0)
    assert m.min(x2, x3).value() == x2.value()
    check_grad(m, m.min(x2, x3), x, g2)
    check_grad(m, m.min(x3, x2), x, g2)
    check_grad(m, m.min(x, x), x, 1.0)


def test_gradient_miscellaneous_and_sign(m):  # :740-836
    x = m.variable(3)
    check_grad(m, x, x, 1.0)
    x.set_value(0.5)
    assert m.erf(x).value() == math.erf(0.5)
    check_grad(m, m.erf(x), x, 2.0 / math.sqrt(math.pi) * math.exp(-0.25), 1e-15)
    for v in (1, -1, 0):
        x.set_value(v)
        assert m.sign(x).value() == model.py_sign(v)
        check_grad(m, m.sign(x), x, 0.0)


def test_gradient_variable_reuse(m):  # :770-793
    a, b = m.variable(10), m.variable(20)
    x = a * b
    check_grad(m, x, a, 20.0)
    b.set_value(10)
    check_grad(m, x, a, 10.0)


def test_gradient_non_scalar(m):  # :838-860
    x = [m.variable(v) for v in (1, 2, 3)]
    y = x[0] + 3 * x[1] - 5 * x[2]
    assert list(m.gradient(y, x)) == [1.0, 3.0, -5.0]
    assert list(m.gradient_symbolic(y, x)) == [1.0, 3.0, -5.0]


# ---- jacobian_test.cpp -----------------------------------------------------------

def test_jacobian_identity_and_scaling(m):  # :13-61
    x = [m.variable(i + 1) for i in range(3)]
    assert np.array_equal(m.jacobian(x, x), np.eye(3))
    y = [3 * xi for xi in x]
    assert np.array_equal(m.jacobian(y, x), 3 * np.eye(3))


def test_jacobian_products(m):  # :63-96
    x = [m.variable(i + 1) for i in range(3)]
    y = [x[0] * x[1], x[1] * x[2], x[0] * x[2]]
    assert np.array_equal(m.jacobian(y, x), np.array([[2, 1, 0], [0, 3, 2], [3, 0, 1]], dtype=float))


def test_jacobian_nested_products(m):  # :98-176
    x = [m.variable(3)]
    y = [5 * x[0], 7 * x[0], 11 * x[0]]
    assert [v.value() for v in y] == [15.0, 21.0, 33.0]
    z = [y[0] * y[1], y[1] * y[2], y[0] * y[2]]
    assert [v.value() for v in z] == [315.0, 693.0, 495.0]
    assert np.array_equal(m.jacobian(y, x), np.array([[5.0], [7.0], [11.0]]))
    if m.is_oracle:
        # wrt = intermediate expressions: general in the reference/oracle; the product
        # differentiates w.r.t. decision variables only (problem.hpp:535-560 is all it serves)
        assert np.array_equal(m.jacobian(z, y), np.array([[21, 15, 0], [0, 33, 21], [33, 0, 15]], dtype=float))
    assert np.array_equal(m.jacobian(z, x), np.array([[210.0], [462.0], [330.0]]))


def test_jacobian_non_square_and_reuse(m):  # :178-247
    x = [m.variable(i + 1) for i in range(3)]
    y = [x[0] + 3 * x[1] - 5 * x[2]]
    J = m.jacobian(y, x)
    assert J.shape == (1, 3) and np.array_equal(J, np.array([[1.0, 3.0, -5.0]]))
    x = [m.variable(1), m.variable(2)]
    y = [x[0] * x[1]]
    assert np.array_equal(m.jacobian(y, x), np.array([[2.0, 1.0]]))
    x[0].set_value(2)
    x[1].set_value(1)
    assert np.array_equal(m.jacobian(y, x), np.array([[1.0, 2.0]]))


# ---- hessian_test.cpp ---------------------------------------------------------------

@pytest.mark.parametrize("power,g,h", [(1, 1.0, 0.0), (2, 6.0, 2.0), (3, 27.0, 18.0), (4, 108.0, 108.0)])
def test_hessian_monomials(m, power, g, h):  # :22-105
    x = [m.variable(3)]
    y = x[0]
    for _ in range(power - 1):
        y = y * x[0]
    assert m.gradient(y, x[0])[0] == g
    assert m.hessian(y, x)[0, 0] == h


def test_hessian_sum_and_sum_of_products(m):  # :107-159
    x = [m.variable(i + 1) for i in range(5)]
    y = m.constant(0)
    for xi in x:
        y = y + xi
    assert y.value() == 15.0
    assert np.array_equal(m.gradient(y, x), np.ones(5))
    assert np.array_equal(m.hessian(y, x), np.zeros((5, 5)))
    y = m.constant(0)
    for xi in x:  # x.T() * x  (variable_matrix.hpp:505-521)
        y = y + xi * xi
    assert y.value() == 55.0
    assert np.array_equal(m.gradient(y, x), 2.0 * np.arange(1, 6))
    assert np.array_equal(m.hessian(y, x), 2.0 * np.eye(5))


def test_hessian_product_of_sines(m):  # :161-205
    x = [m.variable(i + 1) for i in range(5)]
    y = m.constant(1)
    for xi in x:
        y = y * m.sin(xi)
    yv = math.sin(1) * math.sin(2) * math.sin(3) * math.sin(4) * math.sin(5)
    assert abs(y.value() - yv) <= 1e-15
    g = m.gradient(y, x)
    for i in range(5):
        assert abs(g[i] - y.value() / math.tan(i + 1)) <= 1e-15
    H = m.hessian(y, x)
    for i in range(5):
        for j in range(5):
            e = -y.value() if i == j else y.value() / (math.tan(i + 1) * math.tan(j + 1))
            assert abs(H[i, j] - e) <= 1e-15


def test_hessian_sum_of_squared_residuals(m):  # :207-246
    x = [m.variable(1) for _ in range(5)]
    y = m.constant(0)
    for i in range(4):
        y = y + m.pow(x[i] - x[i + 1], 2)
    assert y.value() == 0.0
    assert np.array_equal(m.gradient(y, x), np.zeros(5))
    expected = np.array([[2, -2, 0, 0, 0], [-2, 4, -2, 0, 0], [0, -2, 4, -2, 0], [0, 0, -2, 4, -2],
                         [0, 0, 0, -2, 2]], dtype=float)
    assert np.array_equal(m.hessian(y, x), expected)


def test_hessian_sum_of_squares(m):  # :248-274
    r = [m.constant(v) for v in (25, 10, 5, 0)]
    x = [m.variable(0) for _ in range(4)]
    J = m.constant(0)
    for i in range(4):
        J = J + (r[i] - x[i]) * (r[i] - x[i])
    assert np.array_equal(m.hessian(J, x), 2.0 * np.eye(4))


def test_hessian_nested_powers(m):  # :276-295
    x = m.variable(3)
    y = m.pow(m.pow(x, 2), 2)
    assert abs(m.jacobian([y], [x])[0, 0] - 4 * 27) <= 1e-12
    assert abs(m.hessian(y, [x])[0, 0] - 12 * 9) <= 1e-12


def test_hessian_max_min(m):  # :297-332
    inp = [m.variable(0), m.variable(0)]
    ey = np.array([[0.0, 0.0], [0.0, 2.0]])
    ex = np.array([[2.0, 0.0], [0.0, 0.0]])
    xs, ys = inp[0] * inp[0], inp[1] * inp[1]
    for f, pts in ((m.max(xs, ys), (1, 2, 3, 2)), (m.min(xs, ys), (2, 1, 2, 3))):
        inp[0].set_value(pts[0])
        inp[1].set_value(pts[1])
        assert np.array_equal(m.hessian(f, inp), ey)
        inp[0].set_value(pts[2])
        inp[1].set_value(pts[3])
        assert np.array_equal(m.hessian(f, inp), ex)
        inp[0].set_value(2)
        inp[1].set_value(2)
        assert np.array_equal(m.hessian(f, inp), ex)


def test_hessian_pow(m):  # :334-363
    inp = [m.variable(0), m.variable(0)]
    f = m.pow(inp[0] + 1, inp[1])
    inp[0].set_value(2)
    inp[1].set_value(3)
    mixed = 9 * (1 + 3 * math.log(3))
    assert np.allclose(m.hessian(f, inp), [[18, mixed], [mixed, 27 * math.log(3) ** 2]], atol=1e-12, rtol=0)
    inp[0].set_value(3)
    inp[1].set_value(2)
    mixed = 4 * (1 + 2 * math.log(4))
    assert np.allclose(m.hessian(f, inp), [[2, mixed], [mixed, 16 * math.log(4) ** 2]], atol=1e-12, rtol=0)


def test_hessian_rosenbrock_grid(m):  # :365-404 (coarser grid: 11 x 11 of the reference's 50 x 50)
    x, y = m.variable(0), m.variable(0)
    f = m.pow(1 - x, 2) + 100 * m.pow(y - m.pow(x, 2), 2)
    for x0 in np.arange(-2.5, 2.5, 0.5):
        for y0 in np.arange(-2.5, 2.5, 0.5):
            x.set_value(x0)
            y.set_value(y0)
            H = m.hessian(f, [x, y])
            assert abs(H[0, 0] - (1200 * x0 * x0 - 400 * y0 + 2)) <= 1e-11
            assert close(H[0, 1], -400 * x0, m.tol) and close(H[1, 0], -400 * x0, m.tol)
            assert H[1, 1] == 200.0


def test_hessian_edge_pushing_examples(m):  # :406-475
    x = [m.variable(3), m.variable(4)]
    y = (x[0] * m.sin(x[1])) * x[0]
    J = m.jacobian([y], x)
    assert close(J[0, 0], 6 * math.sin(4), m.tol) and close(J[0, 1], 9 * math.cos(4), m.tol)
    H = m.hessian(y, x)
    expected = np.array([[2 * math.sin(4), 6 * math.cos(4)], [6 * math.cos(4), -9 * math.sin(4)]])
    assert np.max(np.abs(H - expected)) <= m.tol * 10
    p1 = m.constant(2.0)
    x = [m.variable(2), m.variable(3)]
    y = p1 * m.log(x[0] * x[1])
    H = m.hessian(y, x)
    assert np.max(np.abs(H - np.array([[-2.0 / 4.0, 0.0], [0.0, -2.0 / 9.0]]))) <= m.tol


def test_hessian_variable_reuse(m):  # :477-509
    x = [m.variable(1)]
    y = x[0] * x[0] * x[0]
    assert m.hessian(y, x)[0, 0] == 6.0
    x[0].set_value(2)
    assert m.hessian(y, x)[0, 0] == 12.0
