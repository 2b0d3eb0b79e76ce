// grb_semiring.hpp — how kernels see a semiring.
//
// Kernels are written once against this interface.  `StaticSR<T,ADD,MUL>` carries the operator
// codes as template constants (the switch in apply_binop folds away: these are the fast paths for
// the semirings BASELINE.json names), `DynSR<T>` carries them as wave-uniform runtime values and
// covers every other built-in semiring of the same type.
//
// All kernels work in "mxv orientation": t(i) = (+)_j  mult(M(i,j), u(j)).  For GrB_vxm the
// multiply is u (x) A, so the driver swaps the operator for its mirror (FIRST<->SECOND, ...) or,
// for the few operators that have no mirror, sets `flip`.
#pragma once
#include "grb_ops.hpp"

namespace grb {

template <class T, int ADD, int MUL> struct StaticSR {
  static constexpr bool is_static = true;
  static constexpr bool pair_only = MUL == B_PAIR;      // the product is the constant 1 whatever the operands hold
  static constexpr int add_code = ADD;
  T identity, terminal; bool has_terminal;
  GRB_HD T mult(T a, T u) const { return apply_binop<T>(MUL, a, u); }
  GRB_HD T add(T x, T y) const { return apply_binop<T>(ADD, x, y); }
  GRB_HD int add_op() const { return ADD; }
  GRB_HD int mul_op() const { return MUL; }
  GRB_HD bool uses_a() const { return binop_uses_x(MUL); }
  GRB_HD bool uses_u() const { return binop_uses_y(MUL); }
};

template <class T> struct DynSR {
  static constexpr bool is_static = false;
  static constexpr bool pair_only = false;
  static constexpr int add_code = -1;                   // known at run time only
  T identity, terminal; bool has_terminal;
  int addop, mulop; bool flip;
  GRB_HD T mult(T a, T u) const { return flip ? apply_binop<T, false>(mulop, u, a) : apply_binop<T, false>(mulop, a, u); }
  GRB_HD T add(T x, T y) const { return apply_binop<T, false>(addop, x, y); }
  GRB_HD int add_op() const { return addop; }
  GRB_HD int mul_op() const { return mulop; }
  GRB_HD bool uses_a() const { return flip ? binop_uses_y(mulop) : binop_uses_x(mulop); }
  GRB_HD bool uses_u() const { return flip ? binop_uses_x(mulop) : binop_uses_y(mulop); }
};

// Host-side description of the semiring for one call (all values already in the Z type).
struct SemiringDesc {
  int zcode;            // type the semiring computes in
  int addop, mulop;
  bool flip;            // multiply is mult(u, a) and the operator has no mirror opcode
  uint8_t identity[16], terminal[16]; bool has_terminal;
};

// mirror of a binary operator under argument swap; returns false when none exists
inline bool mirror_binop(int op, int* out) {
  switch (op) {
    case B_FIRST: *out = B_SECOND; return true;   case B_SECOND: *out = B_FIRST; return true;
    case B_MINUS: *out = B_RMINUS; return true;   case B_RMINUS: *out = B_MINUS; return true;
    case B_DIV: *out = B_RDIV; return true;       case B_RDIV: *out = B_DIV; return true;
    case B_ISGT: *out = B_ISLT; return true;      case B_ISLT: *out = B_ISGT; return true;
    case B_ISGE: *out = B_ISLE; return true;      case B_ISLE: *out = B_ISGE; return true;
    case B_GT: *out = B_LT; return true;          case B_LT: *out = B_GT; return true;
    case B_GE: *out = B_LE; return true;          case B_LE: *out = B_GE; return true;
    case B_PAIR: case B_ANY: case B_MIN: case B_MAX: case B_PLUS: case B_TIMES: case B_ISEQ: case B_ISNE:
    case B_LOR: case B_LAND: case B_LXOR: case B_EQ: case B_NE: case B_LXNOR: case B_BOR: case B_BAND:
    case B_BXOR: case B_BXNOR: case B_HYPOT:
      *out = op; return true;
    default: return false;
  }
}

// Call f(sr) with the best semiring object for `d`.  The static list is the set of semirings the
// north-star workloads use (PLUS_TIMES, MIN_PLUS, PLUS_PAIR, LOR_LAND) plus those of the reference's
// callers (PLUS_SECOND / PLUS_FIRST: gap/prmark.py:22, gap/bcmark.py:31; ANY_PAIR: demo/Intro-Prez).
template <class T, class F> inline void with_semiring(const SemiringDesc& d, F&& f) {
  T id, term; memcpy(&id, d.identity, sizeof(T)); memcpy(&term, d.terminal, sizeof(T));
#define GRB_TRY_STATIC(A, M) \
  if (!d.flip && d.addop == A && d.mulop == M) { StaticSR<T, A, M> sr{id, term, d.has_terminal}; f(sr); return; }
  if constexpr (is_bool<T>::value) {
    GRB_TRY_STATIC(B_LOR, B_LAND) GRB_TRY_STATIC(B_ANY, B_PAIR) GRB_TRY_STATIC(B_LOR, B_PAIR)
  } else if constexpr (std::is_same<T, int32_t>::value || std::is_same<T, int64_t>::value ||
                       std::is_same<T, float>::value || std::is_same<T, double>::value) {
    GRB_TRY_STATIC(B_PLUS, B_TIMES) GRB_TRY_STATIC(B_MIN, B_PLUS) GRB_TRY_STATIC(B_PLUS, B_PAIR)
    GRB_TRY_STATIC(B_PLUS, B_SECOND) GRB_TRY_STATIC(B_PLUS, B_FIRST)
  }
#undef GRB_TRY_STATIC
  DynSR<T> sr{id, term, d.has_terminal, d.addop, d.mulop, d.flip};
  f(sr);
}

}  // namespace grb
