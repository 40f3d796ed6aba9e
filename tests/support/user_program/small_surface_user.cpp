// TEST FIXTURE — a USER program of the slp:: surface, not part of the product.
//
// The host-side tests of the reference that need no solver iteration, with their own spellings:
//   * test/src/optimization/trivial_problem_test.cpp:12-65 (empty problem; no cost, unconstrained)
//   * test/src/optimization/decision_variable_test.cpp:11-182 (init / assign / blocks / segments /
//     symmetric matrices)
//   * test/src/optimization/constraints_test.cpp:14-278 (boolean value of every combination of
//     scalar, Variable, VariableMatrix, VariableBlock and constant matrix under ==, <, <=, >, >=;
//     concatenation)
// The one substitution: Eigen matrices are slp::DenseMatrix.  Prints the number of failed checks.
#include <array>
#include <cstdio>
#include <tuple>

#include <sleipnir/autodiff/variable.hpp>
#include <sleipnir/autodiff/variable_matrix.hpp>
#include <sleipnir/optimization/problem.hpp>

namespace {
int failed = 0, checked = 0;
#define CHECK(cond)                                                   \
  do {                                                                \
    ++checked;                                                        \
    if (!(cond)) {                                                    \
      ++failed;                                                       \
      std::printf("line %d: CHECK(%s) failed\n", __LINE__, #cond);    \
    }                                                                 \
  } while (0)
#define CHECK_FALSE(cond) CHECK(!(cond))

using T = double;
using MatrixXT = slp::DenseMatrix;

void trivial_problems() {
  {
    slp::Problem<T> problem;
    CHECK(problem.cost_function_type() == slp::ExpressionType::NONE);
    CHECK(problem.equality_constraint_type() == slp::ExpressionType::NONE);
    CHECK(problem.inequality_constraint_type() == slp::ExpressionType::NONE);
    CHECK(problem.solve({.diagnostics = true}) == slp::ExitStatus::SUCCESS);
  }
  {
    slp::Problem<T> problem;
    auto X = problem.decision_variable(2, 3);
    CHECK(problem.cost_function_type() == slp::ExpressionType::NONE);
    CHECK(problem.equality_constraint_type() == slp::ExpressionType::NONE);
    CHECK(problem.inequality_constraint_type() == slp::ExpressionType::NONE);
    CHECK(problem.solve({.diagnostics = true}) == slp::ExitStatus::SUCCESS);
    for (int row = 0; row < X.rows(); ++row)
      for (int col = 0; col < X.cols(); ++col) CHECK(X.value(row, col) == T(0));
  }
  {
    slp::Problem<T> problem;
    auto X = problem.decision_variable(2, 3);
    X.set_value(MatrixXT{{1.0, 1.0, 1.0}, {1.0, 1.0, 1.0}});
    CHECK(problem.solve({.diagnostics = true}) == slp::ExitStatus::SUCCESS);
    for (int row = 0; row < X.rows(); ++row)
      for (int col = 0; col < X.cols(); ++col) CHECK(X.value(row, col) == T(1));
  }
}

void decision_variables() {
  slp::Problem<T> problem;

  // scalar zero init, assignment
  auto x = problem.decision_variable();
  CHECK(x.value() == T(0));
  x.set_value(T(1));
  CHECK(x.value() == T(1));
  x.set_value(T(2));
  CHECK(x.value() == T(2));

  // vector zero init, assignment
  auto y = problem.decision_variable(2);
  CHECK(y.value(0) == T(0));
  CHECK(y.value(1) == T(0));
  y[0].set_value(T(1));
  y[1].set_value(T(2));
  CHECK(y.value(0) == T(1));
  CHECK(y.value(1) == T(2));
  y[0].set_value(T(3));
  y[1].set_value(T(4));
  CHECK(y.value(0) == T(3));
  CHECK(y.value(1) == T(4));

  // matrix zero init
  auto z = problem.decision_variable(3, 2);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 2; ++c) CHECK(z.value(r, c) == T(0));

  // matrix assignment; element comparison
  z.set_value(MatrixXT{{T(1), T(2)}, {T(3), T(4)}, {T(5), T(6)}});
  CHECK(z.value(0, 0) == T(1));
  CHECK(z.value(0, 1) == T(2));
  CHECK(z.value(1, 0) == T(3));
  CHECK(z.value(1, 1) == T(4));
  CHECK(z.value(2, 0) == T(5));
  CHECK(z.value(2, 1) == T(6));

  // matrix assignment; matrix comparison
  {
    MatrixXT expected{{T(7), T(8)}, {T(9), T(10)}, {T(11), T(12)}};
    z.set_value(expected);
    CHECK(z.value() == expected);
  }

  // block assignment
  {
    MatrixXT expected_block{{T(1)}, {T(1)}};
    z.block(0, 0, 2, 1).set_value(expected_block);
    MatrixXT expected_result{{T(1), T(8)}, {T(1), T(10)}, {T(11), T(12)}};
    CHECK(z.value() == expected_result);
  }

  // segment assignment (of a block's column, and of the matrix: a column segment either way)
  {
    MatrixXT expected_block{{T(1)}, {T(1)}};
    z.block(0, 0, 3, 1).segment(0, 2).set_value(expected_block);
    z.segment(0, 2).set_value(expected_block);
    MatrixXT expected_result{{T(1), T(8)}, {T(1), T(10)}, {T(11), T(12)}};
    CHECK(z.value() == expected_result);
  }

  // symmetric matrix: the upper triangle IS the lower one
  auto A = problem.symmetric_decision_variable(2);
  CHECK(A.value(0, 0) == T(0));
  CHECK(A.value(0, 1) == T(0));
  CHECK(A.value(1, 0) == T(0));
  CHECK(A.value(1, 1) == T(0));
  A[0, 0].set_value(T(1));
  A[1, 0].set_value(T(2));
  A[1, 1].set_value(T(3));
  CHECK(A.value(0, 0) == T(1));
  CHECK(A.value(0, 1) == T(2));
  CHECK(A.value(1, 0) == T(2));
  CHECK(A.value(1, 1) == T(3));
}

void constraint_booleans() {
  using slp::Variable;
  using slp::VariableMatrix;
  constexpr std::array args{std::tuple{T(1), T(1)}, std::tuple{T(1), T(2)}, std::tuple{T(2), T(1)}};

  for (const auto& [lhs, rhs] : args) {
    // equality: every pairing of constraints_test.cpp:24-91
    CHECK(bool{T{lhs} == Variable<T>{rhs}} == (lhs == rhs));
    CHECK(bool{T{lhs} == VariableMatrix<T>{{rhs}}} == (lhs == rhs));
    CHECK(bool{Variable<T>{lhs} == T{rhs}} == (lhs == rhs));
    CHECK(bool{Variable<T>{lhs} == Variable<T>{rhs}} == (lhs == rhs));
    CHECK(bool{Variable<T>{lhs} == VariableMatrix<T>{{rhs}}} == (lhs == rhs));
    CHECK(bool{VariableMatrix<T>{{lhs}} == T{rhs}} == (lhs == rhs));
    CHECK(bool{VariableMatrix<T>{{lhs}} == Variable<T>{rhs}} == (lhs == rhs));
    CHECK(bool{VariableMatrix<T>{{lhs}} == VariableMatrix<T>{{rhs}}} == (lhs == rhs));
    CHECK(bool{MatrixXT{{lhs}} == Variable<T>{rhs}} == (lhs == rhs));
    CHECK(bool{MatrixXT{{lhs}} == VariableMatrix<T>{{rhs}}} == (lhs == rhs));
    CHECK(bool{MatrixXT{{lhs}} == VariableMatrix<T>{{rhs}}.block(0, 0, 1, 1)} == (lhs == rhs));
    CHECK(bool{Variable<T>{lhs} == MatrixXT{{rhs}}} == (lhs == rhs));
    CHECK(bool{VariableMatrix<T>{{lhs}} == MatrixXT{{rhs}}} == (lhs == rhs));
    CHECK(bool{VariableMatrix<T>{{lhs}}.block(0, 0, 1, 1) == MatrixXT{{rhs}}} == (lhs == rhs));

    // inequalities (:96-242): < is <=, > is >=
#define CHECK_FOUR(L, R)                     \
  CHECK(bool{(L) < (R)} == (lhs <= rhs));    \
  CHECK(bool{(L) <= (R)} == (lhs <= rhs));   \
  CHECK(bool{(L) > (R)} == (lhs >= rhs));    \
  CHECK(bool{(L) >= (R)} == (lhs >= rhs))
    CHECK_FOUR(T{lhs}, Variable<T>{rhs});
    CHECK_FOUR(T{lhs}, VariableMatrix<T>{{rhs}});
    CHECK_FOUR(Variable<T>{lhs}, T{rhs});
    CHECK_FOUR(Variable<T>{lhs}, Variable<T>{rhs});
    CHECK_FOUR(Variable<T>{lhs}, VariableMatrix<T>{{rhs}});
    CHECK_FOUR(VariableMatrix<T>{{lhs}}, T{rhs});
    CHECK_FOUR(VariableMatrix<T>{{lhs}}, Variable<T>{rhs});
    CHECK_FOUR(VariableMatrix<T>{{lhs}}, VariableMatrix<T>{{rhs}});
    CHECK_FOUR(MatrixXT{{lhs}}, Variable<T>{rhs});
    CHECK_FOUR(MatrixXT{{lhs}}, VariableMatrix<T>{{rhs}});
    CHECK_FOUR(MatrixXT{{lhs}}, VariableMatrix<T>{{rhs}}.block(0, 0, 1, 1));
    CHECK_FOUR(Variable<T>{lhs}, MatrixXT{{rhs}});
    CHECK_FOUR(VariableMatrix<T>{{lhs}}, MatrixXT{{rhs}});
    CHECK_FOUR(VariableMatrix<T>{{lhs}}.block(0, 0, 1, 1), MatrixXT{{rhs}});
#undef CHECK_FOUR
  }
}

void constraint_concatenation() {
  using slp::EqualityConstraints;
  using slp::InequalityConstraints;
  using slp::Variable;
  {
    EqualityConstraints eq1 = Variable<T>{1} == Variable<T>{1};
    EqualityConstraints eq2 = Variable<T>{1} == Variable<T>{2};
    EqualityConstraints eqs{eq1, eq2};
    CHECK(eq1.constraints.size() == 1);
    CHECK(eq2.constraints.size() == 1);
    CHECK(eqs.constraints.size() == 2);
    CHECK(eqs.constraints[0].value() == eq1.constraints[0].value());
    CHECK(eqs.constraints[1].value() == eq2.constraints[0].value());
    CHECK(bool{eq1});
    CHECK_FALSE(bool{eq2});
    CHECK_FALSE(bool{eqs});
  }
  {
    InequalityConstraints ineq1 = Variable<T>{2} < Variable<T>{1};
    InequalityConstraints ineq2 = Variable<T>{1} < Variable<T>{2};
    InequalityConstraints ineqs{ineq1, ineq2};
    CHECK(ineq1.constraints.size() == 1);
    CHECK(ineq2.constraints.size() == 1);
    CHECK(ineqs.constraints.size() == 2);
    CHECK(ineqs.constraints[0].value() == ineq1.constraints[0].value());
    CHECK(ineqs.constraints[1].value() == ineq2.constraints[0].value());
    CHECK_FALSE(bool{ineq1});
    CHECK(bool{ineq2});
    CHECK_FALSE(bool{ineqs});
  }
}
}  // namespace

int main() {
  trivial_problems();
  decision_variables();
  constraint_booleans();
  constraint_concatenation();
  std::printf("checks=%d failed=%d\n", checked, failed);
  return failed == 0 ? 0 : 1;
}
