// TEST FIXTURE — a USER program of the slp:: surface, not part of the product.
//
// The matrix side of the model-building surface the way the reference's own unit tests use it
// (test/src/autodiff/variable_matrix_test.cpp:42-665, slice_test.cpp): assignment and aliasing,
// block(), Python-style slices with steps and negative indices (slp::Slice, `_`), slices of
// slices written through, compound assignment on matrices and views, iterators (forward,
// reverse, of a view), value(), cwise_transform, the static constructors, cwise_reduce, the
// free block() and solve().  One substitution: Eigen matrices are slp::DenseMatrix.
// No device is involved.  Prints the number of failed checks.
#include <cmath>
#include <cstdio>
#include <functional>
#include <iterator>
#include <ranges>

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
using namespace slp::slicing;

void assignment_and_aliasing() {
  slp::VariableMatrix<T> mat;
  CHECK(mat.rows() == 0 && mat.cols() == 0);
  mat = slp::VariableMatrix<T>{2, 2};
  CHECK(mat.rows() == 2 && mat.cols() == 2);
  CHECK((mat[0, 0] == T(0)) && (mat[1, 1] == T(0)));
  mat[0, 0] = T(1);
  mat[0, 1] = T(2);
  mat[1, 0] = T(3);
  mat[1, 1] = T(4);
  CHECK((mat.value() == M{{1, 2}, {3, 4}}));

  slp::VariableMatrix<T> A{{T(1), T(2)}, {T(3), T(4)}};
  slp::VariableMatrix<T> B{{T(5), T(6)}, {T(7), T(8)}};
  A = B;  // handles are shared after assignment
  B[0, 0].set_value(T(2));
  CHECK((A.value() == M{{2, 6}, {7, 8}}));
}

void block_member() {
  slp::VariableMatrix<T> A{{T(1), T(2), T(3)}, {T(4), T(5), T(6)}, {T(7), T(8), T(9)}};
  CHECK((A.block(1, 1, 2, 2).value() == M{{5, 6}, {8, 9}}));
  CHECK((A.block(1, 1, 2, 2).block(1, 1, 1, 1).value() == M{{9}}));
  A.block(1, 1, 2, 2).block(1, 1, 1, 1) = T(10);
  CHECK(A.value(2, 2) == T(10));
}

void slicing() {
  slp::VariableMatrix<T> mat{{T(1), T(2), T(3), T(4)}, {T(5), T(6), T(7), T(8)}, {T(9), T(10), T(11), T(12)},
                             {T(13), T(14), T(15), T(16)}};
  for (int i = 0; i < 16; ++i) CHECK(bool{mat[i] == T(i + 1)});
  {
    auto s = mat[slp::Slice{1, _}, slp::Slice{2, _}];
    CHECK(s.rows() == 3 && s.cols() == 2);
    const T flat[6] = {7, 8, 11, 12, 15, 16};
    for (int i = 0; i < 6; ++i) CHECK(bool{s[i] == flat[i]});
    CHECK(bool{s[2, 1] == T(16)});
  }
  {
    auto s = mat[slp::Slice{-1, _}, slp::Slice{-2, _}];
    CHECK(s.rows() == 1 && s.cols() == 2 && bool{s[0] == T(15)} && bool{s[0, 1] == T(16)});
  }
  CHECK((mat[_, slp::Slice{_, _, 2}].value() == M{{1, 3}, {5, 7}, {9, 11}, {13, 15}}));
  CHECK((mat[slp::Slice{_, _, -1}, slp::Slice{_, _, -2}].value() == M{{16, 14}, {12, 10}, {8, 6}, {4, 2}}));
  CHECK((mat[slp::Slice{1, _}, -1].value() == M{{8}, {12}, {16}}));
  CHECK((mat[slp::Slice{1, _}, -2].value() == M{{7}, {11}, {15}}));
  mat[slp::Slice{_, _, 2}, slp::Slice{_, _, 2}] = M{{17, 18}, {19, 20}};
  CHECK((mat.value() == M{{17, 2, 18, 4}, {5, 6, 7, 8}, {19, 10, 20, 12}, {13, 14, 15, 16}}));
  const slp::VariableMatrix<T>& cmat = mat;  // a slice of a const matrix is a copy
  CHECK((cmat[slp::Slice{0, 2}, slp::Slice{0, 2}].value() == M{{17, 2}, {5, 6}}));
}

void subslicing() {
  {
    slp::VariableMatrix<T> mat{5, 5};
    auto s = mat[slp::Slice{_, _, 2}, slp::Slice{_, _, 1}][slp::Slice{1, 3}, slp::Slice{1, 4}];
    CHECK(s.rows() == 2 && s.cols() == 3);
    s = M{{1, 2, 3}, {4, 5, 6}};
    CHECK((mat.value() == M{{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}, {0, 1, 2, 3, 0}, {0, 0, 0, 0, 0}, {0, 4, 5, 6, 0}}));
  }
  {
    slp::VariableMatrix<T> mat{5, 5};
    auto s = mat[slp::Slice{_, _, -2}, slp::Slice{_, _, -1}][slp::Slice{1, 3}, slp::Slice{1, 4}];
    s = M{{1, 2, 3}, {4, 5, 6}};
    CHECK((mat.value() == M{{0, 6, 5, 4, 0}, {0, 0, 0, 0, 0}, {0, 3, 2, 1, 0}, {0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}}));
  }
  {
    slp::VariableMatrix<T> mat{5, 5};
    auto s = mat[slp::Slice{_, _, 1}, slp::Slice{_, _, 2}][slp::Slice{1, 4}, slp::Slice{1, 3}];
    CHECK(s.rows() == 3 && s.cols() == 2);
    s = M{{1, 2}, {3, 4}, {5, 6}};
    CHECK((mat.value() == M{{0, 0, 0, 0, 0}, {0, 0, 1, 0, 2}, {0, 0, 3, 0, 4}, {0, 0, 5, 0, 6}, {0, 0, 0, 0, 0}}));
  }
  {
    slp::VariableMatrix<T> mat{5, 5};
    auto s = mat[slp::Slice{_, _, -1}, slp::Slice{_, _, -2}][slp::Slice{1, 4}, slp::Slice{1, 3}];
    s = M{{1, 2}, {3, 4}, {5, 6}};
    CHECK((mat.value() == M{{0, 0, 0, 0, 0}, {6, 0, 5, 0, 0}, {4, 0, 3, 0, 0}, {2, 0, 1, 0, 0}, {0, 0, 0, 0, 0}}));
  }
}

void slices_like_python() {  // slice_test.cpp: adjust() is slice.indices()
  auto check = [](slp::Slice s, int length, int start, int stop, int step, int count) {
    const int n = s.adjust(length);
    CHECK(n == count && s.start == start && s.stop == stop && s.step == step);
  };
  check(slp::Slice{}, 3, 0, 0, 1, 0);
  check(slp::Slice{_}, 3, 0, 3, 1, 3);
  check(slp::Slice{1}, 3, 1, 2, 1, 1);
  check(slp::Slice{-1}, 3, 2, 3, 1, 1);
  check(slp::Slice{1, 3}, 3, 1, 3, 1, 2);
  check(slp::Slice{-2, -1}, 3, 1, 2, 1, 1);
  check(slp::Slice{_, 2}, 3, 0, 2, 1, 2);
  check(slp::Slice{1, _}, 3, 1, 3, 1, 2);
  check(slp::Slice{_, _, 2}, 5, 0, 5, 2, 3);
  check(slp::Slice{_, _, -1}, 3, 2, -1, -1, 3);
  check(slp::Slice{_, _, -2}, 5, 4, -1, -2, 3);
  check(slp::Slice{3, 1, -1}, 5, 3, 1, -1, 2);
  check(slp::Slice{1, 3, -1}, 5, 1, 3, -1, 0);
  check(slp::Slice{10, 20}, 5, 5, 5, 1, 0);
  check(slp::Slice{-10, 2}, 5, 0, 2, 1, 2);
}

void compound_assignment() {
  slp::VariableMatrix<T> A1{{T(1)}};
  slp::VariableMatrix<T> A2{{T(1), T(2)}, {T(3), T(4)}};
  const slp::VariableMatrix<T> B{{T(1), T(2)}, {T(3), T(4)}};
  const T b{2};
  A2 += B;
  CHECK((A2.value() == M{{2, 4}, {6, 8}}));
  A2 -= B;
  CHECK((A2.value() == M{{1, 2}, {3, 4}}));
  A2 *= B;
  CHECK((A2.value() == M{{7, 10}, {15, 22}}));
  A2.set_value(M{{1, 2}, {3, 4}});
  A2.block(0, 0, 2, 2) += B;
  CHECK((A2.value() == M{{2, 4}, {6, 8}}));
  A2.block(0, 0, 2, 2) -= B;
  CHECK((A2.value() == M{{1, 2}, {3, 4}}));
  A2.block(0, 0, 2, 2) *= B;
  CHECK((A2.value() == M{{7, 10}, {15, 22}}));
  A2.set_value(M{{1, 2}, {3, 4}});
  A1 += b;
  CHECK((A1.value() == M{{3}}));
  A1 -= b;
  CHECK((A1.value() == M{{1}}));
  A2 *= b;
  CHECK((A2.value() == M{{2, 4}, {6, 8}}));
  A2 /= b;
  CHECK((A2.value() == M{{1, 2}, {3, 4}}));
  A2.block(0, 0, 1, 1) += b;
  CHECK((A2.value() == M{{3, 2}, {3, 4}}));
  A2.block(0, 0, 1, 1) -= b;
  CHECK((A2.value() == M{{1, 2}, {3, 4}}));
  A2.block(0, 0, 2, 2) *= b;
  CHECK((A2.value() == M{{2, 4}, {6, 8}}));
  A2.block(0, 0, 2, 2) /= b;
  CHECK((A2.value() == M{{1, 2}, {3, 4}}));
}

void iterators_and_values() {
  slp::VariableMatrix<T> A{{T(1), T(2), T(3)}, {T(4), T(5), T(6)}, {T(7), T(8), T(9)}};
  auto sub_A = A.block(2, 1, 1, 2);
  CHECK(std::distance(A.begin(), A.end()) == 9 && std::distance(A.cbegin(), A.cend()) == 9);
  CHECK(std::distance(A.rbegin(), A.rend()) == 9 && std::distance(A.crbegin(), A.crend()) == 9);
  CHECK(std::distance(sub_A.begin(), sub_A.end()) == 2 && std::distance(sub_A.cbegin(), sub_A.cend()) == 2);
  CHECK(std::distance(sub_A.rbegin(), sub_A.rend()) == 2 && std::distance(sub_A.crbegin(), sub_A.crend()) == 2);
  int i = 1;
  for (auto& elem : A) CHECK(elem.value() == T(i++));
  i = 9;
  for (auto& elem : A | std::views::reverse) CHECK(elem.value() == T(i--));
  i = 8;
  for (auto& elem : sub_A) CHECK(elem.value() == T(i++));
  i = 9;
  for (auto& elem : sub_A | std::views::reverse) CHECK(elem.value() == T(i--));

  CHECK(A.value(3) == T(4) && A.T().value(3) == T(2));
  CHECK(A.block(1, 1, 2, 2).value(2) == T(8) && A.T().block(1, 1, 2, 2).value(2) == T(6));
  CHECK((A[slp::Slice{1, 3}, slp::Slice{1, 3}].value() == M{{5, 6}, {8, 9}}));
  CHECK(A[slp::Slice{1, 3}, slp::Slice{1, 3}].value(2) == T(8));
  CHECK(A[slp::Slice{1, 3}, slp::Slice{1, 3}].T().value(2) == T(6));
  CHECK((A.block(1, 1, 2, 2).block(0, 1, 2, 1).value() == M{{6}, {9}}));
  CHECK((A[slp::Slice{1, 3}, slp::Slice{1, 3}][_, slp::Slice{1, _}].value() == M{{6}, {9}}));
  CHECK(A[slp::Slice{1, 3}, slp::Slice{1, 3}][_, slp::Slice{1, _}].value(1) == T(9));
}

void element_functions_and_statics() {
  slp::VariableMatrix<T> A{{T(-2), T(-3), T(-4)}, {T(-5), T(-6), T(-7)}};
  auto abs_A = A.cwise_transform(slp::abs<T>);
  CHECK((abs_A.value() == M{{2, 3, 4}, {5, 6, 7}}));
  auto sub = A.block(0, 0, 2, 2).cwise_transform([](const slp::Variable<T>& x) { return x * T(2); });
  CHECK((sub.value() == M{{-4, -6}, {-10, -12}}));

  CHECK((slp::VariableMatrix<T>::identity(3).value() == M{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}));
  for (auto& e : slp::VariableMatrix<T>::zero(2, 3)) CHECK(e.value() == T(0));
  for (auto& e : slp::VariableMatrix<T>::one(2, 3)) CHECK(e.value() == T(1));
  auto C = slp::VariableMatrix<T>::constant(2, 3, T(2));
  CHECK(C.rows() == 2 && C.cols() == 3);
  for (auto& e : C) CHECK(e.value() == T(2));

  slp::VariableMatrix<T> P{{T(2), T(3), T(4)}, {T(5), T(6), T(7)}};
  slp::VariableMatrix<T> Q{{T(8), T(9), T(10)}, {T(11), T(12), T(13)}};
  slp::VariableMatrix result = slp::cwise_reduce<T>(P, Q, std::multiplies<>{});
  CHECK((result.value() == M{{16, 27, 40}, {55, 72, 91}}));
}

bool within(const M& a, const M& b, double tol) {
  if (a.rows() != b.rows() || a.cols() != b.cols()) return false;
  for (int r = 0; r < a.rows(); ++r)
    for (int c = 0; c < a.cols(); ++c)
      if (!(std::abs(a[r, c] - b[r, c]) <= tol)) return false;
  return true;
}
M product(const M& a, const M& b) {
  M p{a.rows(), b.cols()};
  for (int r = 0; r < a.rows(); ++r)
    for (int c = 0; c < b.cols(); ++c)
      for (int k = 0; k < a.cols(); ++k) p[r, c] += a[r, k] * b[k, c];
  return p;
}

void matrix_exponential() {  // variable_matrix_test.cpp:534-595
  auto A1 = slp::VariableMatrix<T>{{T(4)}};
  CHECK(within(A1.exp().value(), M{{std::exp(4.0)}}, 1e-13));
  CHECK(within(A1[_, _].exp().value(), M{{std::exp(4.0)}}, 1e-13));
  const M eye{{1, 0}, {0, 1}};
  for (auto A : {slp::VariableMatrix<T>{{T(0), T(1)}, {T(0), T(-0.5)}}, slp::VariableMatrix<T>{{T(0), T(1)}, {T(0), T(10)}},
                 slp::VariableMatrix<T>{{T(1), T(10)}, {T(0), T(0)}}, slp::VariableMatrix<T>{{T(2), T(3)}, {T(4), T(5)}}}) {
    CHECK(within(product(A.exp().value(), (-A).exp().value()), eye, 1e-12));
    CHECK(within(product(A[_, _].exp().value(), (-A[_, _]).exp().value()), eye, 1e-12));
  }
  // exp of the subdiagonal 1..6 is Pascal's triangle
  auto pascal = slp::VariableMatrix<T>::zero(7, 7);
  for (int col = 0; col < 6; ++col) pascal[col + 1, col] = T(col + 1);
  M expected{7, 7};
  for (int r = 0; r < 7; ++r) expected[r, 0] = 1;
  for (int col = 1; col < 7; ++col)
    for (int row = col; row < 7; ++row) expected[row, col] = expected[row - 1, col - 1] + expected[row - 1, col];
  CHECK(within(pascal.exp().value(), expected, 1e-13));
  CHECK(within(pascal[_, _].exp().value(), expected, 1e-13));
}

void block_free_function() {
  slp::VariableMatrix<T> A{{T(1), T(2), T(3)}, {T(4), T(5), T(6)}};
  slp::VariableMatrix<T> B{{T(7)}, {T(8)}};
  slp::VariableMatrix mat1 = slp::block({{A, B}});
  CHECK(mat1.rows() == 2 && mat1.cols() == 4 && (mat1.value() == M{{1, 2, 3, 7}, {4, 5, 6, 8}}));
  slp::VariableMatrix<T> C{{T(9), T(10), T(11), T(12)}};
  slp::VariableMatrix mat2 = slp::block({{A, B}, {C}});
  CHECK(mat2.rows() == 3 && mat2.cols() == 4 && (mat2.value() == M{{1, 2, 3, 7}, {4, 5, 6, 8}, {9, 10, 11, 12}}));
}

void check_solve(const slp::VariableMatrix<T>& A, const slp::VariableMatrix<T>& B) {
  auto X = solve(A, B);
  CHECK(X.rows() == A.cols() && X.cols() == B.cols());
  const M a = A.value(), x = X.value(), b = B.value();
  double norm2 = 0.0;
  for (int r = 0; r < a.rows(); ++r)
    for (int c = 0; c < b.cols(); ++c) {
      double acc = -b[r, c];
      for (int k = 0; k < a.cols(); ++k) acc += a[r, k] * x[k, c];
      norm2 += acc * acc;
    }
  CHECK(std::sqrt(norm2) < 1e-12);
}

void solve_free_function() {
  check_solve({{T(2)}}, {{T(5)}});
  check_solve({{T(1), T(2)}, {T(3), T(4)}}, {{T(5)}, {T(6)}});
  check_solve({{T(1), T(2), T(3)}, {T(-4), T(-5), T(6)}, {T(7), T(8), T(9)}}, {{T(10)}, {T(11)}, {T(12)}});
  check_solve({{T(1), T(2), T(3), T(-4)}, {T(-5), T(6), T(7), T(8)}, {T(9), T(10), T(11), T(12)}, {T(13), T(14), T(15), T(16)}},
              {{T(17)}, {T(18)}, {T(19)}, {T(20)}});
  check_solve({{T(1), T(2), T(3), T(-4), T(5)},
               {T(-5), T(6), T(7), T(8), T(9)},
               {T(9), T(10), T(11), T(12), T(13)},
               {T(13), T(14), T(15), T(16), T(17)},
               {T(17), T(18), T(19), T(20), T(21)}},
              {{T(21)}, {T(22)}, {T(23)}, {T(24)}, {T(25)}});
}
}  // namespace

int main() {
  assignment_and_aliasing();
  block_member();
  slicing();
  subslicing();
  slices_like_python();
  compound_assignment();
  iterators_and_values();
  element_functions_and_statics();
  matrix_exponential();
  block_free_function();
  solve_free_function();
  std::printf("checks=%d failed=%d\n", checked, failed);
  return failed == 0 ? 0 : 1;
}
