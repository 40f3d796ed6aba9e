// TEST FIXTURE — a USER program of the slp:: surface, not part of the product.
//
// problem.solve(options, /*spy=*/true) (test/src/optimization/problem_spy_test.cpp:54-148):
// the sparsity files H.spy, A_e.spy, A_i.spy hold one record per iteration with the signs of
// the entries; titles, labels, shapes and coordinates as the reference's test reads them back.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <fstream>
#include <string>

#include <sleipnir/autodiff/variable.hpp>
#include <sleipnir/optimization/problem.hpp>

namespace {
int failed = 0;
#define CHECK(...)                                                           \
  do {                                                                       \
    if (!(__VA_ARGS__)) {                                                    \
      ++failed;                                                              \
      std::printf("line %d: CHECK(%s) failed\n", __LINE__, #__VA_ARGS__);    \
    }                                                                        \
  } while (0)

int32_t read_i32(std::ifstream& f) {
  unsigned char b[4] = {0, 0, 0, 0};
  f.read(reinterpret_cast<char*>(b), 4);
  return static_cast<int32_t>(b[0] | (b[1] << 8) | (b[2] << 16) | (static_cast<uint32_t>(b[3]) << 24));
}
std::string read_str(std::ifstream& f) {
  std::string s(static_cast<size_t>(read_i32(f)), '\0');
  f.read(s.data(), static_cast<std::streamsize>(s.size()));
  return s;
}
struct Coord {
  int32_t row, col;
  char sign;
  bool operator==(const Coord&) const = default;
};
Coord read_coord(std::ifstream& f) {
  const int32_t r = read_i32(f), c = read_i32(f);
  char s = 0;
  f.read(&s, 1);
  return {r, c, s};
}
}  // namespace

int main() {
  using T = double;
  try {
    slp::Problem<T> problem;
    auto x = problem.decision_variable();
    auto y = problem.decision_variable();
    x.set_value(T(20));
    y.set_value(T(20));
    problem.minimize(pow(x, T(4)) + pow(y, T(4)));
    problem.subject_to(x >= T(1));
    problem.subject_to(x <= T(10));
    problem.subject_to(y == T(2));
    int iterations = 0;
    problem.add_callback([&](const slp::IterationInfo<T>&) { ++iterations; });

    CHECK(problem.solve({}, true) == slp::ExitStatus::SUCCESS);
    CHECK(std::abs(x.value() - 1.0) < 1e-8 && std::abs(y.value() - 2.0) < 1e-8);
    CHECK(iterations > 0);
    {
      std::ifstream spy{"H.spy", std::ios::binary};
      CHECK(read_str(spy) == "Hessian" && read_str(spy) == "Decision variables" && read_str(spy) == "Decision variables");
      CHECK(read_i32(spy) == 2 && read_i32(spy) == 2);
      for (int i = 0; i < iterations; ++i) {
        CHECK(read_i32(spy) == 2);
        CHECK(read_coord(spy) == Coord{0, 0, '+'});
        CHECK(read_coord(spy) == Coord{1, 1, '+'});
        CHECK(!spy.eof());
      }
    }
    {
      std::ifstream spy{"A_e.spy", std::ios::binary};
      CHECK(read_str(spy) == "Equality constraint Jacobian" && read_str(spy) == "Constraints" && read_str(spy) == "Decision variables");
      CHECK(read_i32(spy) == 1 && read_i32(spy) == 2);
      for (int i = 0; i < iterations; ++i) {
        CHECK(read_i32(spy) == 1);
        CHECK(read_coord(spy) == Coord{0, 1, '+'});
        CHECK(!spy.eof());
      }
    }
    {
      std::ifstream spy{"A_i.spy", std::ios::binary};
      CHECK(read_str(spy) == "Inequality constraint Jacobian" && read_str(spy) == "Constraints" && read_str(spy) == "Decision variables");
      CHECK(read_i32(spy) == 2 && read_i32(spy) == 2);
      for (int i = 0; i < iterations; ++i) {
        CHECK(read_i32(spy) == 2);
        CHECK(read_coord(spy) == Coord{0, 0, '+'});
        CHECK(read_coord(spy) == Coord{1, 0, '-'});
        CHECK(!spy.eof());
      }
    }
    std::printf("iterations=%d failed_checks=%d\n", iterations, failed);
    return failed == 0 ? 0 : 1;
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 3;
  }
}
