// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked, imported or called by the
// product path (sleipnir_amd/).  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may use anything under oracle/.
//
// CPU restatement of the reference's reverse-mode autodiff layer, following
//   include/sleipnir/autodiff/expression.hpp        (node types, value/grad
//                                                    formulas, pruning + typing
//                                                    rules :155-348 and per op)
//   include/sleipnir/autodiff/expression_graph.hpp  (:29-78 topological_sort,
//                                                    :86-96 update_values,
//                                                    :107-153 append_triplets)
//   include/sleipnir/autodiff/variable_matrix.hpp   (:1757-1805 gradient_tree)
//   include/sleipnir/autodiff/jacobian.hpp          (:54-105 ctor, :134-156 value)
//   include/sleipnir/autodiff/hessian.hpp           (:49-103 ctor, :132-157 value)
//   include/sleipnir/autodiff/gradient.hpp          (:53-57)
// It is a pointer-linked graph walked one row at a time, exactly like the
// reference (this is deliberately NOT how the product does it).
//
// Parity pin status: pinned against the closed-form values the reference's own
// unit tests hold (test/src/autodiff/{gradient,jacobian,hessian,expression}_test.cpp),
// re-expressed in tests/test_oracle_autodiff.py.
#pragma once

#include <cmath>
#include <cstdint>
#include <deque>
#include <utility>
#include <vector>

namespace orc {

enum class Op : uint8_t {
  CONST = 0,
  VAR,
  ADD,
  SUB,
  NEG,
  MUL,
  DIV,
  POW,
  ABS,
  SIGN,
  SQRT,
  CBRT,
  EXP,
  LOG,
  LOG10,
  SIN,
  COS,
  TAN,
  ASIN,
  ACOS,
  ATAN,
  ATAN2,
  SINH,
  COSH,
  TANH,
  ERF,
  HYPOT,
  MAX,
  MIN,
  ISNONNEG,
  ISPOS,
  NUM_OPS
};

// expression_type.hpp:15-26
enum Type : uint8_t { NONE = 0, CONSTANT, LINEAR, QUADRATIC, NONLINEAR };

// expression.hpp:89-118
struct Expr {
  double val = 0.0;
  double adjoint = 0.0;
  Expr* adjoint_expr = nullptr;
  Expr* args[2] = {nullptr, nullptr};
  int32_t scratch = -1;
  Op op = Op::CONST;
  Type type = CONSTANT;

  bool is_constant(double c) const { return type == CONSTANT && val == c; }
};

// Arena standing in for the reference's thread-local pool (src/util/pool.cpp:5-8).
// Nodes live until reset(); the oracle does not reproduce refcounting.
struct Arena {
  std::deque<Expr> nodes;
  Expr* make(Op op, Type type, Expr* l, Expr* r, double val = 0.0) {
    nodes.emplace_back();
    Expr* e = &nodes.back();
    e->op = op;
    e->type = type;
    e->args[0] = l;
    e->args[1] = r;
    e->val = val;
    return e;
  }
  void reset() { nodes.clear(); }
};

inline Arena& arena() {
  static thread_local Arena a;
  return a;
}

inline Expr* constant(double v) {
  return arena().make(Op::CONST, CONSTANT, nullptr, nullptr, v);
}
inline Expr* decision_variable(double v = 0.0) {
  return arena().make(Op::VAR, LINEAR, nullptr, nullptr, v);
}

// ---------------------------------------------------------------------------
// Numeric op semantics: value / grad_l / grad_r (expression.hpp, one struct per op)
// ---------------------------------------------------------------------------

inline double op_value(Op op, double l, double r) {
  switch (op) {
    case Op::ADD: return l + r;
    case Op::SUB: return l - r;
    case Op::NEG: return -l;
    case Op::MUL: return l * r;
    case Op::DIV: return l / r;
    case Op::POW: return std::pow(l, r);
    case Op::ABS: return std::abs(l);
    case Op::SIGN: return l < 0.0 ? -1.0 : (l == 0.0 ? 0.0 : 1.0);
    case Op::SQRT: return std::sqrt(l);
    case Op::CBRT: return std::cbrt(l);
    case Op::EXP: return std::exp(l);
    case Op::LOG: return std::log(l);
    case Op::LOG10: return std::log10(l);
    case Op::SIN: return std::sin(l);
    case Op::COS: return std::cos(l);
    case Op::TAN: return std::tan(l);
    case Op::ASIN: return std::asin(l);
    case Op::ACOS: return std::acos(l);
    case Op::ATAN: return std::atan(l);
    case Op::ATAN2: return std::atan2(l, r);
    case Op::SINH: return std::sinh(l);
    case Op::COSH: return std::cosh(l);
    case Op::TANH: return std::tanh(l);
    case Op::ERF: return std::erf(l);
    case Op::HYPOT: return std::hypot(l, r);
    case Op::MAX: return std::max(l, r);
    case Op::MIN: return std::min(l, r);
    case Op::ISNONNEG: return l >= 0.0 ? 1.0 : 0.0;
    case Op::ISPOS: return l > 0.0 ? 1.0 : 0.0;
    default: return 0.0;
  }
}

constexpr double kLn10 = 2.302585092994045684017991454684364208;
constexpr double kTwoInvSqrtPi = 2.0 * 0.564189583547756286948079451560772586;

// grad_l already multiplied by the parent adjoint `a` (expression.hpp e.g. :670-676)
inline double op_grad_l(Op op, double a, double l, double r) {
  switch (op) {
    case Op::ADD: return a;
    case Op::SUB: return a;
    case Op::NEG: return -a;
    case Op::MUL: return a * r;
    case Op::DIV: return a / r;
    case Op::POW: return a * std::pow(l, r - 1.0) * r;
    case Op::ABS: return l < 0.0 ? -a : (l > 0.0 ? a : 0.0);
    case Op::SQRT: return a / (2.0 * std::sqrt(l));
    case Op::CBRT: {
      double c = std::cbrt(l);
      return a / (3.0 * c * c);
    }
    case Op::EXP: return a * std::exp(l);
    case Op::LOG: return a / l;
    case Op::LOG10: return a / (kLn10 * l);
    case Op::SIN: return a * std::cos(l);
    case Op::COS: return a * -std::sin(l);
    case Op::TAN: {
      double c = std::cos(l);
      return a / (c * c);
    }
    case Op::ASIN: return a / std::sqrt(1.0 - l * l);
    case Op::ACOS: return -a / std::sqrt(1.0 - l * l);
    case Op::ATAN: return a / (1.0 + l * l);
    case Op::ATAN2: return a * r / (l * l + r * r);
    case Op::SINH: return a * std::cosh(l);
    case Op::COSH: return a * std::sinh(l);
    case Op::TANH: {
      double c = std::cosh(l);
      return a / (c * c);
    }
    case Op::ERF: return a * kTwoInvSqrtPi * std::exp(-l * l);
    case Op::HYPOT: return a * l / std::hypot(l, r);
    case Op::MAX: return l >= r ? a : 0.0;
    case Op::MIN: return l <= r ? a : 0.0;
    default: return 0.0;  // SIGN, ISNONNEG, ISPOS, leaves
  }
}

inline double op_grad_r(Op op, double a, double l, double r) {
  switch (op) {
    case Op::ADD: return a;
    case Op::SUB: return -a;
    case Op::MUL: return a * l;
    case Op::DIV: return a * -l / (r * r);
    case Op::POW: return a * std::pow(l, r) * std::log(l);
    case Op::ATAN2: return a * -l / (l * l + r * r);
    case Op::HYPOT: return a * r / std::hypot(l, r);
    case Op::MAX: return l >= r ? 0.0 : a;
    case Op::MIN: return l <= r ? 0.0 : a;
    default: return 0.0;
  }
}

// ---------------------------------------------------------------------------
// Graph-building operators with the reference's pruning / folding / typing
// ---------------------------------------------------------------------------

inline Type tmax(Type a, Type b) { return a > b ? a : b; }

Expr* neg(Expr* lhs);

// expression.hpp:155-201
inline Expr* mul(Expr* lhs, Expr* rhs) {
  if (lhs->is_constant(0.0)) return lhs;
  if (rhs->is_constant(0.0)) return rhs;
  if (lhs->is_constant(1.0)) return rhs;
  if (rhs->is_constant(1.0)) return lhs;
  if (lhs->type == CONSTANT && rhs->type == CONSTANT) {
    return constant(lhs->val * rhs->val);
  }
  Type t;
  if (lhs->type == CONSTANT) {
    t = rhs->type == LINEAR ? LINEAR : (rhs->type == QUADRATIC ? QUADRATIC : NONLINEAR);
  } else if (rhs->type == CONSTANT) {
    t = lhs->type == LINEAR ? LINEAR : (lhs->type == QUADRATIC ? QUADRATIC : NONLINEAR);
  } else if (lhs->type == LINEAR && rhs->type == LINEAR) {
    t = QUADRATIC;
  } else {
    t = NONLINEAR;
  }
  return arena().make(Op::MUL, t, lhs, rhs);
}

// expression.hpp:207-237
inline Expr* div(Expr* lhs, Expr* rhs) {
  if (lhs->is_constant(0.0)) return lhs;
  if (rhs->is_constant(1.0)) return lhs;
  if (lhs->type == CONSTANT && rhs->type == CONSTANT) {
    return constant(lhs->val / rhs->val);
  }
  Type t = NONLINEAR;
  if (rhs->type == CONSTANT) {
    t = lhs->type == LINEAR ? LINEAR : (lhs->type == QUADRATIC ? QUADRATIC : NONLINEAR);
  }
  return arena().make(Op::DIV, t, lhs, rhs);
}

// expression.hpp:243-273 (nullptr operands are legal: adjoint accumulation)
inline Expr* add(Expr* lhs, Expr* rhs) {
  if (lhs == nullptr || lhs->is_constant(0.0)) return rhs;
  if (rhs == nullptr || rhs->is_constant(0.0)) return lhs;
  if (lhs->type == CONSTANT && rhs->type == CONSTANT) {
    return constant(lhs->val + rhs->val);
  }
  Type t = tmax(lhs->type, rhs->type);
  if (t != LINEAR && t != QUADRATIC) t = NONLINEAR;
  return arena().make(Op::ADD, t, lhs, rhs);
}

// expression.hpp:288-322
inline Expr* sub(Expr* lhs, Expr* rhs) {
  if (lhs->is_constant(0.0)) {
    if (rhs->is_constant(0.0)) return rhs;
    return neg(rhs);
  }
  if (rhs->is_constant(0.0)) return lhs;
  if (lhs->type == CONSTANT && rhs->type == CONSTANT) {
    return constant(lhs->val - rhs->val);
  }
  Type t = tmax(lhs->type, rhs->type);
  if (t != LINEAR && t != QUADRATIC) t = NONLINEAR;
  return arena().make(Op::SUB, t, lhs, rhs);
}

// expression.hpp:327-348
inline Expr* neg(Expr* lhs) {
  if (lhs->is_constant(0.0)) return lhs;
  if (lhs->type == CONSTANT) return constant(-lhs->val);
  Type t = lhs->type == LINEAR ? LINEAR : (lhs->type == QUADRATIC ? QUADRATIC : NONLINEAR);
  return arena().make(Op::NEG, t, lhs, nullptr);
}

inline Expr* unary_nl(Op op, Expr* x) { return arena().make(op, NONLINEAR, x, nullptr); }
inline Expr* binary_nl(Op op, Expr* l, Expr* r) { return arena().make(op, NONLINEAR, l, r); }

// expression.hpp:811-826
inline Expr* abs(Expr* x) {
  if (x->is_constant(0.0)) return x;
  if (x->type == CONSTANT) return constant(std::abs(x->val));
  return unary_nl(Op::ABS, x);
}
// :866-880
inline Expr* acos(Expr* x) {
  if (x->is_constant(0.0)) return constant(M_PI / 2.0);
  if (x->type == CONSTANT) return constant(std::acos(x->val));
  return unary_nl(Op::ACOS, x);
}
// :920-935
inline Expr* asin(Expr* x) {
  if (x->is_constant(0.0)) return x;
  if (x->type == CONSTANT) return constant(std::asin(x->val));
  return unary_nl(Op::ASIN, x);
}
// :974-989
inline Expr* atan(Expr* x) {
  if (x->is_constant(0.0)) return x;
  if (x->type == CONSTANT) return constant(std::atan(x->val));
  return unary_nl(Op::ATAN, x);
}
// :1041-1051
inline Expr* atan2(Expr* y, Expr* x) {
  if (y->type == CONSTANT && x->type == CONSTANT) return constant(std::atan2(y->val, x->val));
  return binary_nl(Op::ATAN2, y, x);
}
// :553-569
inline Expr* cbrt(Expr* x) {
  if (x->type == CONSTANT) {
    if (x->val == 0.0) return x;
    if (x->val == -1.0 || x->val == 1.0) return x;
    return constant(std::cbrt(x->val));
  }
  return unary_nl(Op::CBRT, x);
}
// :1091-1105
inline Expr* cos(Expr* x) {
  if (x->is_constant(0.0)) return constant(1.0);
  if (x->type == CONSTANT) return constant(std::cos(x->val));
  return unary_nl(Op::COS, x);
}
// :1145-1159
inline Expr* cosh(Expr* x) {
  if (x->is_constant(0.0)) return constant(1.0);
  if (x->type == CONSTANT) return constant(std::cosh(x->val));
  return unary_nl(Op::COSH, x);
}
// :1200-1215
inline Expr* erf(Expr* x) {
  if (x->is_constant(0.0)) return x;
  if (x->type == CONSTANT) return constant(std::erf(x->val));
  return unary_nl(Op::ERF, x);
}
// :1255-1269
inline Expr* exp(Expr* x) {
  if (x->is_constant(0.0)) return constant(1.0);
  if (x->type == CONSTANT) return constant(std::exp(x->val));
  return unary_nl(Op::EXP, x);
}
// :1327-1344
inline Expr* hypot(Expr* x, Expr* y) {
  if (x->is_constant(0.0)) return abs(y);
  if (y->is_constant(0.0)) return abs(x);
  if (x->type == CONSTANT && y->type == CONSTANT) return constant(std::hypot(x->val, y->val));
  return binary_nl(Op::HYPOT, x, y);
}
// :1363-1369, :1396-1402
inline Expr* is_nonnegative(Expr* x) {
  if (x->type == CONSTANT) return constant(x->val >= 0.0 ? 1.0 : 0.0);
  return unary_nl(Op::ISNONNEG, x);
}
inline Expr* is_positive(Expr* x) {
  if (x->type == CONSTANT) return constant(x->val > 0.0 ? 1.0 : 0.0);
  return unary_nl(Op::ISPOS, x);
}
// :1445-1462 (quirk: log(const 0) returns the 0 constant)
inline Expr* log(Expr* x) {
  if (x->is_constant(0.0)) return x;
  if (x->type == CONSTANT) return constant(std::log(x->val));
  return unary_nl(Op::LOG, x);
}
// :1499-1516
inline Expr* log10(Expr* x) {
  if (x->is_constant(0.0)) return x;
  if (x->type == CONSTANT) return constant(std::log10(x->val));
  return unary_nl(Op::LOG10, x);
}
// :1575-1585, :1645-1655
inline Expr* max(Expr* a, Expr* b) {
  if (a->type == CONSTANT && b->type == CONSTANT) return constant(std::max(a->val, b->val));
  return binary_nl(Op::MAX, a, b);
}
inline Expr* min(Expr* a, Expr* b) {
  if (a->type == CONSTANT && b->type == CONSTANT) return constant(std::min(a->val, b->val));
  return binary_nl(Op::MIN, a, b);
}
// :1716-1749
inline Expr* pow(Expr* base, Expr* power) {
  if (base->is_constant(0.0)) return base;
  if (base->is_constant(1.0)) return base;
  if (power->is_constant(0.0)) return constant(1.0);
  if (power->is_constant(1.0)) return base;
  if (base->type == CONSTANT && power->type == CONSTANT) {
    return constant(std::pow(base->val, power->val));
  }
  if (power->is_constant(2.0)) {
    return arena().make(Op::MUL, base->type == LINEAR ? QUADRATIC : NONLINEAR, base, base);
  }
  return binary_nl(Op::POW, base, power);
}
// :1783-1798
inline Expr* sign(Expr* x) {
  if (x->type == CONSTANT) {
    if (x->val < 0.0) return constant(-1.0);
    if (x->val == 0.0) return x;
    return constant(1.0);
  }
  return unary_nl(Op::SIGN, x);
}
// :1838-1853
inline Expr* sin(Expr* x) {
  if (x->is_constant(0.0)) return x;
  if (x->type == CONSTANT) return constant(std::sin(x->val));
  return unary_nl(Op::SIN, x);
}
// :1893-1908
inline Expr* sinh(Expr* x) {
  if (x->is_constant(0.0)) return x;
  if (x->type == CONSTANT) return constant(std::sinh(x->val));
  return unary_nl(Op::SINH, x);
}
// :1948-1964
inline Expr* sqrt(Expr* x) {
  if (x->type == CONSTANT) {
    if (x->val == 0.0) return x;
    if (x->val == 1.0) return x;
    return constant(std::sqrt(x->val));
  }
  return unary_nl(Op::SQRT, x);
}
// :2007-2022
inline Expr* tan(Expr* x) {
  if (x->is_constant(0.0)) return x;
  if (x->type == CONSTANT) return constant(std::tan(x->val));
  return unary_nl(Op::TAN, x);
}
// :2065-2080
inline Expr* tanh(Expr* x) {
  if (x->is_constant(0.0)) return x;
  if (x->type == CONSTANT) return constant(std::tanh(x->val));
  return unary_nl(Op::TANH, x);
}

// ---------------------------------------------------------------------------
// Symbolic gradients grad_expr_l / grad_expr_r (expression.hpp, per op)
// `a` is the node's adjoint_expr.
// ---------------------------------------------------------------------------

inline Expr* op_grad_expr_l(Op op, Expr* a, Expr* l, Expr* r) {
  switch (op) {
    case Op::ADD: return a;
    case Op::SUB: return a;
    case Op::NEG: return neg(a);
    case Op::MUL: return mul(a, r);
    case Op::DIV: return div(a, r);
    case Op::POW: return mul(mul(a, pow(l, sub(r, constant(1.0)))), r);
    case Op::ABS: return mul(a, sign(l));
    case Op::SQRT: return div(a, mul(constant(2.0), sqrt(l)));
    case Op::CBRT: {
      Expr* c = cbrt(l);
      return div(a, mul(mul(constant(3.0), c), c));
    }
    case Op::EXP: return mul(a, exp(l));
    case Op::LOG: return div(a, l);
    case Op::LOG10: return div(a, mul(constant(kLn10), l));
    case Op::SIN: return mul(a, cos(l));
    case Op::COS: return mul(a, neg(sin(l)));
    case Op::TAN: {
      Expr* c = cos(l);
      return div(a, mul(c, c));
    }
    case Op::ASIN: return div(a, sqrt(sub(constant(1.0), mul(l, l))));
    case Op::ACOS: return div(neg(a), sqrt(sub(constant(1.0), mul(l, l))));
    case Op::ATAN: return div(a, add(constant(1.0), mul(l, l)));
    case Op::ATAN2: return div(mul(a, r), add(mul(l, l), mul(r, r)));
    case Op::SINH: return mul(a, cosh(l));
    case Op::COSH: return mul(a, sinh(l));
    case Op::TANH: {
      Expr* c = cosh(l);
      return div(a, mul(c, c));
    }
    case Op::ERF: return mul(mul(a, constant(kTwoInvSqrtPi)), exp(neg(mul(l, l))));
    case Op::HYPOT: return div(mul(a, l), hypot(l, r));
    case Op::MAX: return mul(a, is_nonnegative(sub(l, r)));
    case Op::MIN: return mul(a, is_nonnegative(sub(r, l)));
    default: return constant(0.0);
  }
}

inline Expr* op_grad_expr_r(Op op, Expr* a, Expr* l, Expr* r) {
  switch (op) {
    case Op::ADD: return a;
    case Op::SUB: return neg(a);
    case Op::MUL: return mul(a, l);
    case Op::DIV: return div(mul(a, neg(l)), mul(r, r));
    case Op::POW: return mul(mul(a, pow(l, r)), log(l));
    case Op::ATAN2: return div(mul(a, neg(l)), add(mul(l, l), mul(r, r)));
    case Op::HYPOT: return div(mul(a, r), hypot(l, r));
    case Op::MAX: return mul(a, is_positive(sub(r, l)));
    case Op::MIN: return mul(a, is_positive(sub(l, r)));
    default: return constant(0.0);
  }
}

// ---------------------------------------------------------------------------
// expression_graph.hpp
// ---------------------------------------------------------------------------

using Graph = std::vector<Expr*>;

// :29-78
inline Graph topological_sort(Expr* root) {
  Graph list;
  if (root == nullptr || root->type == CONSTANT) return list;
  std::vector<Expr*> stack;
  stack.push_back(root);
  while (!stack.empty()) {
    Expr* node = stack.back();
    stack.pop_back();
    for (Expr* arg : node->args) {
      if (arg != nullptr && ++arg->scratch == 0) stack.push_back(arg);
    }
  }
  stack.push_back(root);
  while (!stack.empty()) {
    Expr* node = stack.back();
    stack.pop_back();
    list.push_back(node);
    for (Expr* arg : node->args) {
      if (arg != nullptr && --arg->scratch == -1) stack.push_back(arg);
    }
  }
  return list;
}

// :86-96
inline void update_values(const Graph& list) {
  for (auto it = list.rbegin(); it != list.rend(); ++it) {
    Expr* node = *it;
    Expr* lhs = node->args[0];
    Expr* rhs = node->args[1];
    if (lhs != nullptr) {
      node->val = op_value(node->op, lhs->val, rhs ? rhs->val : 0.0);
    }
  }
}

struct Triplet {
  int row, col;
  double value;
};

using OutputList = std::vector<std::pair<int, Expr*>>;

// :107-153
inline void append_triplets(const Graph& top_list, const OutputList& output_list,
                            std::vector<Triplet>& triplets, int row) {
  if (top_list.empty()) return;
  top_list[0]->adjoint = 1.0;
  for (size_t i = 1; i < top_list.size(); ++i) top_list[i]->adjoint = 0.0;
  for (Expr* node : top_list) {
    Expr* lhs = node->args[0];
    Expr* rhs = node->args[1];
    if (lhs != nullptr) {
      if (rhs != nullptr) {
        lhs->adjoint += op_grad_l(node->op, node->adjoint, lhs->val, rhs->val);
        rhs->adjoint += op_grad_r(node->op, node->adjoint, lhs->val, rhs->val);
      } else {
        lhs->adjoint += op_grad_l(node->op, node->adjoint, lhs->val, 0.0);
      }
    }
  }
  for (const auto& [col, node] : output_list) {
    triplets.push_back({row, col, node->adjoint});
  }
}

// variable_matrix.hpp:1757-1805.  Returns one Expr* per wrt entry (nullptr when
// wrt[i] is not reachable from the root).
inline std::vector<Expr*> gradient_tree(const Graph& top_list, const std::vector<Expr*>& wrt) {
  std::vector<Expr*> grad(wrt.size(), nullptr);
  if (top_list.empty()) return grad;
  top_list[0]->adjoint_expr = constant(1.0);
  for (Expr* node : top_list) {
    Expr* lhs = node->args[0];
    Expr* rhs = node->args[1];
    if (lhs != nullptr) {
      if (rhs != nullptr) {
        lhs->adjoint_expr = add(lhs->adjoint_expr, op_grad_expr_l(node->op, node->adjoint_expr, lhs, rhs));
        rhs->adjoint_expr = add(rhs->adjoint_expr, op_grad_expr_r(node->op, node->adjoint_expr, lhs, rhs));
      } else {
        lhs->adjoint_expr = add(lhs->adjoint_expr, op_grad_expr_l(node->op, node->adjoint_expr, lhs, rhs));
      }
    }
  }
  for (size_t i = 0; i < wrt.size(); ++i) {
    grad[i] = wrt[i]->adjoint_expr;
    wrt[i]->adjoint_expr = nullptr;  // std::move in the reference
  }
  for (Expr* node : top_list) node->adjoint_expr = nullptr;
  return grad;
}

}  // namespace orc
