// Host-side expression graph of the MI355X build: a flat struct-of-arrays arena
// (one int32 id per node) instead of the reference's pointer-linked, ref-counted
// nodes (include/sleipnir/autodiff/expression.hpp:89-118).  The arena IS the
// source for the device tape: ids are dense, children always have smaller ids
// than their parents, so any id-sorted subset is already topologically ordered.
//
// Build-time semantics restated from the reference so graphs come out identical:
//   * constant folding / pruning / LINEAR-QUADRATIC-NONLINEAR typing:
//     expression.hpp:155-348 (+ - * / neg) and the per-function rules at
//     :553-569 cbrt, :811-826 abs, :866-880 acos, :920-935 asin, :974-989 atan,
//     :1041-1051 atan2, :1091-1105 cos, :1145-1159 cosh, :1200-1215 erf,
//     :1255-1269 exp, :1327-1344 hypot, :1445-1462 log, :1499-1516 log10,
//     :1575-1585 max, :1645-1655 min, :1716-1749 pow, :1783-1798 sign,
//     :1838-1853 sin, :1893-1908 sinh, :1948-1964 sqrt, :2007-2022 tan,
//     :2065-2080 tanh
//   * symbolic reverse pass (detail::gradient_tree, variable_matrix.hpp:1757-1805)
//     using each op's grad_expr_l / grad_expr_r
//   * per-root parent->child ordering (detail::topological_sort,
//     expression_graph.hpp:29-78), kept because it fixes the order in which a
//     node's adjoint contributions are summed.
#pragma once

#include <cmath>
#include <cstdint>
#include <vector>

#include "tape_ops.h"

namespace slpx {

using NodeId = int32_t;
constexpr NodeId kNull = -1;

// expression_type.hpp:15-26
enum ExprType : uint8_t { T_NONE = 0, T_CONSTANT, T_LINEAR, T_QUADRATIC, T_NONLINEAR };

struct Graph {
  std::vector<uint8_t> op;    // Opcode (tape_ops.h)
  std::vector<uint8_t> type;  // ExprType
  std::vector<NodeId> a0, a1;
  std::vector<double> val;    // constants: the value; variables: current value; else cache
  std::vector<int32_t> scratch;

  NodeId make(Opcode o, ExprType t, NodeId l, NodeId r, double v = 0.0) {
    op.push_back(static_cast<uint8_t>(o));
    type.push_back(t);
    a0.push_back(l);
    a1.push_back(r);
    val.push_back(v);
    scratch.push_back(-1);
    return static_cast<NodeId>(op.size()) - 1;
  }
  size_t size() const { return op.size(); }
  void reserve(size_t n) {
    op.reserve(n);
    type.reserve(n);
    a0.reserve(n);
    a1.reserve(n);
    val.reserve(n);
    scratch.reserve(n);
  }
  void clear() {
    op.clear();
    type.clear();
    a0.clear();
    a1.clear();
    val.clear();
    scratch.clear();
  }

  bool is_const(NodeId n) const { return type[n] == T_CONSTANT; }
  bool is_const(NodeId n, double c) const { return type[n] == T_CONSTANT && val[n] == c; }

  NodeId constant(double v) { return make(OP_CONST, T_CONSTANT, kNull, kNull, v); }
  NodeId variable(double v = 0.0) { return make(OP_VAR, T_LINEAR, kNull, kNull, v); }

  // ---- arithmetic (expression.hpp:155-348) --------------------------------
  static ExprType keep_poly(uint8_t t) {
    return t == T_LINEAR ? T_LINEAR : (t == T_QUADRATIC ? T_QUADRATIC : T_NONLINEAR);
  }

  NodeId mul(NodeId l, NodeId r) {
    if (is_const(l, 0.0)) return l;
    if (is_const(r, 0.0)) return r;
    if (is_const(l, 1.0)) return r;
    if (is_const(r, 1.0)) return l;
    if (is_const(l) && is_const(r)) return constant(val[l] * val[r]);
    ExprType t = T_NONLINEAR;
    if (is_const(l)) t = keep_poly(type[r]);
    else if (is_const(r)) t = keep_poly(type[l]);
    else if (type[l] == T_LINEAR && type[r] == T_LINEAR) t = T_QUADRATIC;
    return make(OP_MUL, t, l, r);
  }
  NodeId div(NodeId l, NodeId r) {
    if (is_const(l, 0.0)) return l;
    if (is_const(r, 1.0)) return l;
    if (is_const(l) && is_const(r)) return constant(val[l] / val[r]);
    return make(OP_DIV, is_const(r) ? keep_poly(type[l]) : T_NONLINEAR, l, r);
  }
  NodeId add(NodeId l, NodeId r) {  // null operands allowed (adjoint accumulation)
    if (l == kNull || is_const(l, 0.0)) return r;
    if (r == kNull || is_const(r, 0.0)) return l;
    if (is_const(l) && is_const(r)) return constant(val[l] + val[r]);
    return make(OP_ADD, keep_poly(type[l] > type[r] ? type[l] : type[r]), l, r);
  }
  NodeId sub(NodeId l, NodeId r) {
    if (is_const(l, 0.0)) return is_const(r, 0.0) ? r : neg(r);
    if (is_const(r, 0.0)) return l;
    if (is_const(l) && is_const(r)) return constant(val[l] - val[r]);
    return make(OP_SUB, keep_poly(type[l] > type[r] ? type[l] : type[r]), l, r);
  }
  NodeId neg(NodeId l) {
    if (is_const(l, 0.0)) return l;
    if (is_const(l)) return constant(-val[l]);
    return make(OP_NEG, keep_poly(type[l]), l, kNull);
  }

  // ---- functions ---------------------------------------------------------
  // Generic unary: `zero_rule` says what f(const 0) returns: 0 -> the operand
  // itself, 1 -> constant 1, 2 -> no special case.
  NodeId unary(Opcode o, NodeId x) {
    switch (o) {
      case OP_NEG: return neg(x);
      case OP_ABS: case OP_ASIN: case OP_ATAN: case OP_ERF: case OP_LOG: case OP_LOG10:
      case OP_SIN: case OP_SINH: case OP_TAN: case OP_TANH:
        if (is_const(x, 0.0)) return x;
        break;
      case OP_COS: case OP_COSH: case OP_EXP:
        if (is_const(x, 0.0)) return constant(1.0);
        break;
      case OP_ACOS:
        if (is_const(x, 0.0)) return constant(M_PI / 2.0);
        break;
      case OP_SQRT:
        if (is_const(x) && (val[x] == 0.0 || val[x] == 1.0)) return x;
        break;
      case OP_CBRT:
        if (is_const(x) && (val[x] == 0.0 || val[x] == 1.0 || val[x] == -1.0)) return x;
        break;
      case OP_SIGN:
        if (is_const(x)) {
          if (val[x] < 0.0) return constant(-1.0);
          if (val[x] == 0.0) return x;
          return constant(1.0);
        }
        break;
      default: break;
    }
    if (is_const(x)) return constant(op_value(o, val[x], 0.0));
    return make(o, T_NONLINEAR, x, kNull);
  }
  NodeId binary(Opcode o, NodeId l, NodeId r) {
    switch (o) {
      case OP_ADD: return add(l, r);
      case OP_SUB: return sub(l, r);
      case OP_MUL: return mul(l, r);
      case OP_DIV: return div(l, r);
      case OP_POW: return pow(l, r);
      case OP_HYPOT:
        if (is_const(l, 0.0)) return unary(OP_ABS, r);
        if (is_const(r, 0.0)) return unary(OP_ABS, l);
        break;
      default: break;
    }
    if (is_const(l) && is_const(r)) return constant(op_value(o, val[l], val[r]));
    return make(o, T_NONLINEAR, l, r);
  }
  NodeId pow(NodeId base, NodeId power) {
    if (is_const(base, 0.0) || is_const(base, 1.0)) return base;
    if (is_const(power, 0.0)) return constant(1.0);
    if (is_const(power, 1.0)) return base;
    if (is_const(base) && is_const(power)) return constant(std::pow(val[base], val[power]));
    if (is_const(power, 2.0))
      return make(OP_MUL, type[base] == T_LINEAR ? T_QUADRATIC : T_NONLINEAR, base, base);
    return make(OP_POW, T_NONLINEAR, base, power);
  }

  // ---- symbolic partials (grad_expr_l / grad_expr_r of every op) -----------
  NodeId grad_expr(int side, NodeId node, NodeId a) {
    const Opcode o = static_cast<Opcode>(op[node]);
    const NodeId l = a0[node], r = a1[node];
    if (side == 0) {
      switch (o) {
        case OP_ADD: case OP_SUB: return a;
        case OP_NEG: return neg(a);
        case OP_MUL: return mul(a, r);
        case OP_DIV: return div(a, r);
        case OP_POW: return mul(mul(a, pow(l, sub(r, constant(1.0)))), r);
        case OP_ABS: return mul(a, unary(OP_SIGN, l));
        case OP_SQRT: return div(a, mul(constant(2.0), unary(OP_SQRT, l)));
        case OP_CBRT: {
          NodeId c = unary(OP_CBRT, l);
          return div(a, mul(mul(constant(3.0), c), c));
        }
        case OP_EXP: return mul(a, unary(OP_EXP, l));
        case OP_LOG: return div(a, l);
        case OP_LOG10: return div(a, mul(constant(kLn10), l));
        case OP_SIN: return mul(a, unary(OP_COS, l));
        case OP_COS: return mul(a, neg(unary(OP_SIN, l)));
        case OP_TAN: {
          NodeId c = unary(OP_COS, l);
          return div(a, mul(c, c));
        }
        case OP_ASIN: return div(a, unary(OP_SQRT, sub(constant(1.0), mul(l, l))));
        case OP_ACOS: return div(neg(a), unary(OP_SQRT, sub(constant(1.0), mul(l, l))));
        case OP_ATAN: return div(a, add(constant(1.0), mul(l, l)));
        case OP_ATAN2: return div(mul(a, r), add(mul(l, l), mul(r, r)));
        case OP_SINH: return mul(a, unary(OP_COSH, l));
        case OP_COSH: return mul(a, unary(OP_SINH, l));
        case OP_TANH: {
          NodeId c = unary(OP_COSH, l);
          return div(a, mul(c, c));
        }
        case OP_ERF: return mul(mul(a, constant(kTwoInvSqrtPi)), unary(OP_EXP, neg(mul(l, l))));
        case OP_HYPOT: return div(mul(a, l), binary(OP_HYPOT, l, r));
        case OP_MAX: return mul(a, unary(OP_ISNONNEG, sub(l, r)));
        case OP_MIN: return mul(a, unary(OP_ISNONNEG, sub(r, l)));
        default: return constant(0.0);
      }
    }
    switch (o) {
      case OP_ADD: return a;
      case OP_SUB: return neg(a);
      case OP_MUL: return mul(a, l);
      case OP_DIV: return div(mul(a, neg(l)), mul(r, r));
      case OP_POW: return mul(mul(a, pow(l, r)), unary(OP_LOG, l));
      case OP_ATAN2: return div(mul(a, neg(l)), add(mul(l, l), mul(r, r)));
      case OP_HYPOT: return div(mul(a, r), binary(OP_HYPOT, l, r));
      case OP_MAX: return mul(a, unary(OP_ISPOS, sub(r, l)));
      case OP_MIN: return mul(a, unary(OP_ISPOS, sub(l, r)));
      default: return constant(0.0);
    }
  }

  // ---- per-root ordering (expression_graph.hpp:29-78) ----------------------
  std::vector<NodeId> topological_sort(NodeId root) {
    std::vector<NodeId> list;
    if (root == kNull || type[root] == T_CONSTANT) return list;
    std::vector<NodeId> stack{root};
    while (!stack.empty()) {
      NodeId n = stack.back();
      stack.pop_back();
      for (NodeId arg : {a0[n], a1[n]})
        if (arg != kNull && ++scratch[arg] == 0) stack.push_back(arg);
    }
    stack.push_back(root);
    while (!stack.empty()) {
      NodeId n = stack.back();
      stack.pop_back();
      list.push_back(n);
      for (NodeId arg : {a0[n], a1[n]})
        if (arg != kNull && --scratch[arg] == -1) stack.push_back(arg);
    }
    return list;
  }

  // ---- symbolic gradient (variable_matrix.hpp:1757-1805) -------------------
  // Returns one node id (or kNull) per wrt entry.
  std::vector<NodeId> gradient_tree(const std::vector<NodeId>& top_list,
                                    const std::vector<NodeId>& wrt) {
    std::vector<NodeId> grad(wrt.size(), kNull);
    if (top_list.empty()) return grad;
    if (m_adj.size() < size()) m_adj.resize(size(), kNull);
    auto adj = [&](NodeId n) -> NodeId& {
      if (static_cast<size_t>(n) >= m_adj.size()) m_adj.resize(size(), kNull);
      return m_adj[n];
    };
    adj(top_list[0]) = constant(1.0);
    for (NodeId n : top_list) {
      const NodeId l = a0[n], r = a1[n];
      if (l == kNull) continue;
      NodeId gl = grad_expr(0, n, adj(n));
      adj(l) = add(adj(l), gl);
      if (r != kNull) {
        NodeId gr = grad_expr(1, n, adj(n));
        adj(r) = add(adj(r), gr);
      }
    }
    for (size_t i = 0; i < wrt.size(); ++i) {
      grad[i] = adj(wrt[i]);
      adj(wrt[i]) = kNull;
    }
    for (NodeId n : top_list) adj(n) = kNull;
    return grad;
  }

  // ---- host evaluation (Variable::value(), setup-time constants) -----------
  // expression_graph.hpp:86-96
  void update_values(const std::vector<NodeId>& list) {
    for (auto it = list.rbegin(); it != list.rend(); ++it) {
      NodeId n = *it;
      if (a0[n] != kNull)
        val[n] = op_value(static_cast<Opcode>(op[n]), val[a0[n]], a1[n] != kNull ? val[a1[n]] : 0.0);
    }
  }
  double value(NodeId root) {
    auto list = topological_sort(root);
    update_values(list);
    return val[root];
  }

 private:
  std::vector<NodeId> m_adj;
};

// One arena per host thread, like the reference's thread_local pool
// (src/util/pool.cpp:5-8).
// Defined once in graph.cpp (NOT inline: every shared object that uses the DSL
// must see libslpx's arena, also when libslpx is dlopen()ed RTLD_LOCAL).
Graph& graph();

}  // namespace slpx
