// TEST FIXTURE — a USER program of the slp:: surface, not part of the product.
//
// slp::Gradient / Jacobian / Hessian the way the reference's unit tests use them
// (test/src/autodiff/gradient_test.cpp, jacobian_test.cpp:13-247, hessian_test.cpp:22-509):
// get() (symbolic) and value() (the compiled tape on the GPU) against the same known answers.
// "symbolic" as the only argument checks get() alone — no device needed.
#include <cmath>
#include <cstdio>
#include <exception>

#include <sleipnir/autodiff/gradient.hpp>
#include <sleipnir/autodiff/hessian.hpp>
#include <sleipnir/autodiff/jacobian.hpp>
#include <sleipnir/autodiff/variable.hpp>
#include <sleipnir/autodiff/variable_matrix.hpp>

namespace {
int failed = 0, checked = 0;
#define CHECK(...)                                                           \
  do {                                                                       \
    ++checked;                                                               \
    if (!(__VA_ARGS__)) {                                                    \
      ++failed;                                                              \
      std::printf("line %d: CHECK(%s) failed\n", __LINE__, #__VA_ARGS__);    \
    }                                                                        \
  } while (0)
using T = double;
using M = slp::DenseMatrix;
bool near(const M& a, const M& b, double tol) {
  if (a.rows() != b.rows() || a.cols() != b.cols()) return false;
  for (int r = 0; r < a.rows(); ++r)
    for (int c = 0; c < a.cols(); ++c)
      if (!(std::abs(a[r, c] - b[r, c]) <= tol)) return false;
  return true;
}
}  // namespace

int main(int argc, char**) {
  const bool device = argc < 2;
  try {
    {  // hessian_test.cpp: Linear, Quadratic
      slp::VariableMatrix<T> x{1};
      x[0].set_value(T(3));
      slp::Variable y = x[0] * x[0];
      CHECK(slp::Gradient(y, x[0]).get().value(0, 0) == T(6));
      auto H = slp::Hessian(y, x);
      CHECK(H.get().value(0, 0) == T(2));
      if (device) {
        CHECK(slp::Gradient(y, x[0]).value().coeff(0) == T(6));
        CHECK(H.value().coeff(0, 0) == T(2));
      }
    }
    {  // jacobian_test.cpp: y = x, y = 3x, products
      slp::VariableMatrix<T> x{3};
      for (int i = 0; i < 3; ++i) x[i].set_value(T(i + 1));
      auto J1 = slp::Jacobian(x, x);
      CHECK((J1.get().value() == M{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}));
      slp::VariableMatrix<T> y{3};
      y[0] = x[0] * x[1];
      y[1] = x[1] * x[2];
      y[2] = x[0] * x[2];
      auto J = slp::Jacobian(y, x);
      const M expected{{2, 1, 0}, {0, 3, 2}, {3, 0, 1}};
      CHECK((J.get().value() == expected));
      if (device) {
        CHECK((J1.value().toDense() == M{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}));
        CHECK((J.value().toDense() == expected));
        x[0].set_value(T(5));  // the same evaluator at new values
        CHECK((J.value().toDense() == M{{2, 5, 0}, {0, 3, 2}, {3, 0, 5}}));
        CHECK((J.get().value() == J.value().toDense()));
      }
    }
    {  // hessian_test.cpp: sum of squares, product of sines, lower triangle only
      slp::VariableMatrix<T> v{5};
      for (int i = 0; i < 5; ++i) v[i].set_value(T(i + 1));
      slp::Variable f = T(0);
      for (int i = 0; i < 5; ++i) f += v[i] * v[i];
      CHECK((slp::Gradient(f, v).get().value() == M{{2}, {4}, {6}, {8}, {10}}));
      slp::Variable g = sin(v[0]) * sin(v[1]) + v[0] * v[2];
      const T s0 = std::sin(1.0), s1 = std::sin(2.0), c0 = std::cos(1.0), c1 = std::cos(2.0);
      const M Hg{{-s0 * s1, c0 * c1, 1}, {c0 * c1, -s0 * s1, 0}, {1, 0, 0}};
      CHECK(near(slp::Hessian(g, v.segment(0, 3)).get().value(), Hg, 1e-15));
      if (device) {
        CHECK((slp::Gradient(f, v).value().toDense() == M{{2}, {4}, {6}, {8}, {10}}));
        M two_I{5, 5};
        for (int i = 0; i < 5; ++i) two_I[i, i] = 2;
        CHECK((slp::Hessian(f, v).value().toDense() == two_I));
        auto H = slp::Hessian(g, v.segment(0, 3));
        CHECK(near(H.value().toDense(), Hg, 1e-15));
        auto HL = slp::Hessian<T, slp::Lower>(g, v.segment(0, 3));
        CHECK(near(HL.value().toDense(), M{{-s0 * s1, 0, 0}, {c0 * c1, -s0 * s1, 0}, {1, 0, 0}}, 1e-15));
        CHECK(HL.value().coeff(0, 1) == T(0) && H.value().coeff(0, 1) == H.value().coeff(1, 0));
      }
    }
    std::printf("checks=%d failed=%d\n", checked, failed);
    return failed == 0 ? 0 : 1;
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 3;
  }
}
