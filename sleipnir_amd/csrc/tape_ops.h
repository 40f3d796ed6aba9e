// Opcode table of the device tape and the per-op arithmetic shared by the HIP
// kernels and the host-side graph (setup-time constant folding, Variable::value()).
//
// One opcode per reference node type (include/sleipnir/autodiff/expression.hpp:
// BinaryMinus :444, BinaryPlus :481, Cbrt :517, Constant :576, DecisionVariable
// :594, Div :616, Mult :656, UnaryMinus :696, Abs :773, Acos :833, Asin :887,
// Atan :942, Atan2 :996, Cos :1058, Cosh :1112, Erf :1166, Exp :1222, Hypot
// :1280, IsNonnegative :1351, IsPositive :1384, Log :1417, Log10 :1469, Max
// :1525, Min :1594, Pow :1667, Sign :1756, Sin :1805, Sinh :1860, Sqrt :1915,
// Tan :1971, Tanh :2029).
//
// The reference's grad_l/grad_r return adjoint*partial; the tape stores the two
// local partials d(node)/d(child) once per node in the forward sweep and every
// row's adjoint sweep reuses them (one FMA per edge instead of re-evaluating
// cos/sin/pow per row as the reference does, expression_graph.hpp:138-142).
#pragma once

// This header is also compiled at run time by hipRTC as the prelude of the generated
// per-template tape kernels (tape_jit.cpp), where the HIP runtime and math declarations are
// built in and no standard headers exist.
#if !defined(__HIPCC_RTC__)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#else
typedef unsigned char uint8_t;
#endif

#define SLPX_HD __host__ __device__ __forceinline__

namespace slpx {

enum Opcode : uint8_t {
  OP_CONST = 0,
  OP_VAR,
  OP_ADD,
  OP_SUB,
  OP_NEG,
  OP_MUL,
  OP_DIV,
  OP_POW,
  OP_ABS,
  OP_SIGN,
  OP_SQRT,
  OP_CBRT,
  OP_EXP,
  OP_LOG,
  OP_LOG10,
  OP_SIN,
  OP_COS,
  OP_TAN,
  OP_ASIN,
  OP_ACOS,
  OP_ATAN,
  OP_ATAN2,
  OP_SINH,
  OP_COSH,
  OP_TANH,
  OP_ERF,
  OP_HYPOT,
  OP_MAX,
  OP_MIN,
  OP_ISNONNEG,
  OP_ISPOS,
  OP_COUNT
};

constexpr double kLn10 = 2.302585092994045684017991454684364208;
constexpr double kTwoInvSqrtPi = 1.128379167095512573896158903121545172;

inline bool op_is_binary(Opcode o) {
  switch (o) {
    case OP_ADD: case OP_SUB: case OP_MUL: case OP_DIV: case OP_POW: case OP_ATAN2:
    case OP_HYPOT: case OP_MAX: case OP_MIN:
      return true;
    default:
      return false;
  }
}

// value(l, r) of every op (expression.hpp `value()` overrides)
SLPX_HD double op_value(Opcode o, double l, double r) {
  switch (o) {
    case OP_ADD: return l + r;
    case OP_SUB: return l - r;
    case OP_NEG: return -l;
    case OP_MUL: return l * r;
    case OP_DIV: return l / r;
    case OP_POW: return pow(l, r);
    case OP_ABS: return fabs(l);
    case OP_SIGN: return l < 0.0 ? -1.0 : (l == 0.0 ? 0.0 : 1.0);
    case OP_SQRT: return sqrt(l);
    case OP_CBRT: return cbrt(l);
    case OP_EXP: return exp(l);
    case OP_LOG: return log(l);
    case OP_LOG10: return log10(l);
    case OP_SIN: return sin(l);
    case OP_COS: return cos(l);
    case OP_TAN: return tan(l);
    case OP_ASIN: return asin(l);
    case OP_ACOS: return acos(l);
    case OP_ATAN: return atan(l);
    case OP_ATAN2: return atan2(l, r);
    case OP_SINH: return sinh(l);
    case OP_COSH: return cosh(l);
    case OP_TANH: return tanh(l);
    case OP_ERF: return erf(l);
    case OP_HYPOT: return hypot(l, r);
    case OP_MAX: return l >= r ? l : r;
    case OP_MIN: return l <= r ? l : r;
    case OP_ISNONNEG: return l >= 0.0 ? 1.0 : 0.0;
    case OP_ISPOS: return l > 0.0 ? 1.0 : 0.0;
    default: return 0.0;
  }
}

// Ops of the "basic" kernel specialization: arithmetic, sin/cos/sqrt and the
// piecewise ops.  Programs that use nothing else (every BASELINE workload) run a
// kernel that does not carry pow/exp/log/erf/... code and registers.
inline bool op_is_basic(Opcode o) {
  switch (o) {
    case OP_CONST: case OP_VAR: case OP_ADD: case OP_SUB: case OP_NEG: case OP_MUL: case OP_DIV:
    case OP_ABS: case OP_SIGN: case OP_SQRT: case OP_SIN: case OP_COS: case OP_MAX: case OP_MIN:
    case OP_ISNONNEG: case OP_ISPOS:
      return true;
    default:
      return false;
  }
}

// Forward evaluation of one tape node: value and the local partials
// dl = d(node)/d(lhs), dr = d(node)/d(rhs) (only computed when requested).
// The partials are the reference's grad_l / grad_r with the adjoint factored out.
// FULL = false compiles only the basic op set (see op_is_basic).
template <bool FULL = true>
SLPX_HD void op_forward(Opcode o, double l, double r, bool want_dl, bool want_dr, double& v,
                        double& dl, double& dr) {
  dl = 0.0;
  dr = 0.0;
  if constexpr (!FULL) {
    switch (o) {
      case OP_ADD: v = l + r; dl = 1.0; dr = 1.0; return;
      case OP_SUB: v = l - r; dl = 1.0; dr = -1.0; return;
      case OP_NEG: v = -l; dl = -1.0; return;
      case OP_MUL: v = l * r; dl = r; dr = l; return;
      case OP_DIV: {
        const double inv = 1.0 / r;
        v = l / r;
        dl = inv;
        dr = -v * inv;
        return;
      }
      case OP_ABS: v = fabs(l); dl = l < 0.0 ? -1.0 : (l > 0.0 ? 1.0 : 0.0); return;
      case OP_SIGN: v = l < 0.0 ? -1.0 : (l == 0.0 ? 0.0 : 1.0); return;
      case OP_SQRT: v = sqrt(l); dl = 1.0 / (2.0 * v); return;
      case OP_SIN: sincos(l, &v, &dl); return;
      case OP_COS: { double s; sincos(l, &s, &v); dl = -s; return; }
      case OP_MAX: v = l >= r ? l : r; dl = l >= r ? 1.0 : 0.0; dr = l >= r ? 0.0 : 1.0; return;
      case OP_MIN: v = l <= r ? l : r; dl = l <= r ? 1.0 : 0.0; dr = l <= r ? 0.0 : 1.0; return;
      case OP_ISNONNEG: v = l >= 0.0 ? 1.0 : 0.0; return;
      case OP_ISPOS: v = l > 0.0 ? 1.0 : 0.0; return;
      default: v = 0.0; return;
    }
  }
  switch (o) {
    case OP_ADD: v = l + r; dl = 1.0; dr = 1.0; break;
    case OP_SUB: v = l - r; dl = 1.0; dr = -1.0; break;
    case OP_NEG: v = -l; dl = -1.0; break;
    case OP_MUL: v = l * r; dl = r; dr = l; break;
    case OP_DIV: {
      const double inv = 1.0 / r;
      v = l / r;
      dl = inv;
      dr = -v * inv;
      break;
    }
    case OP_POW:
      v = pow(l, r);
      if (want_dl) dl = pow(l, r - 1.0) * r;
      if (want_dr) dr = v * log(l);
      break;
    case OP_ABS: v = fabs(l); dl = l < 0.0 ? -1.0 : (l > 0.0 ? 1.0 : 0.0); break;
    case OP_SIGN: v = l < 0.0 ? -1.0 : (l == 0.0 ? 0.0 : 1.0); break;
    case OP_SQRT: v = sqrt(l); if (want_dl) dl = 1.0 / (2.0 * v); break;
    case OP_CBRT: v = cbrt(l); if (want_dl) dl = 1.0 / (3.0 * v * v); break;
    case OP_EXP: v = exp(l); dl = v; break;
    case OP_LOG: v = log(l); if (want_dl) dl = 1.0 / l; break;
    case OP_LOG10: v = log10(l); if (want_dl) dl = 1.0 / (kLn10 * l); break;
    case OP_SIN:
      if (want_dl) { sincos(l, &v, &dl); } else { v = sin(l); }
      break;
    case OP_COS:
      if (want_dl) { double s; sincos(l, &s, &v); dl = -s; } else { v = cos(l); }
      break;
    case OP_TAN:
      v = tan(l);
      if (want_dl) { double c = cos(l); dl = 1.0 / (c * c); }
      break;
    case OP_ASIN: v = asin(l); if (want_dl) dl = 1.0 / sqrt(1.0 - l * l); break;
    case OP_ACOS: v = acos(l); if (want_dl) dl = -1.0 / sqrt(1.0 - l * l); break;
    case OP_ATAN: v = atan(l); if (want_dl) dl = 1.0 / (1.0 + l * l); break;
    case OP_ATAN2: {
      v = atan2(l, r);
      double den = l * l + r * r;
      if (want_dl) dl = r / den;
      if (want_dr) dr = -l / den;
      break;
    }
    case OP_SINH: v = sinh(l); if (want_dl) dl = cosh(l); break;
    case OP_COSH: v = cosh(l); if (want_dl) dl = sinh(l); break;
    case OP_TANH:
      v = tanh(l);
      if (want_dl) { double c = cosh(l); dl = 1.0 / (c * c); }
      break;
    case OP_ERF: v = erf(l); if (want_dl) dl = kTwoInvSqrtPi * exp(-l * l); break;
    case OP_HYPOT:
      v = hypot(l, r);
      if (want_dl) dl = l / v;
      if (want_dr) dr = r / v;
      break;
    case OP_MAX: v = l >= r ? l : r; dl = l >= r ? 1.0 : 0.0; dr = l >= r ? 0.0 : 1.0; break;
    case OP_MIN: v = l <= r ? l : r; dl = l <= r ? 1.0 : 0.0; dr = l <= r ? 0.0 : 1.0; break;
    case OP_ISNONNEG: v = l >= 0.0 ? 1.0 : 0.0; break;
    case OP_ISPOS: v = l > 0.0 ? 1.0 : 0.0; break;
    default: v = 0.0; break;
  }
}

}  // namespace slpx
