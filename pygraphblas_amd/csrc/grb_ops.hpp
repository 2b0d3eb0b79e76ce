// grb_ops.hpp — scalar operator semantics shared by host code and HIP kernels.
//
// Every built-in GraphBLAS binary/unary operator is one opcode; `apply_binop<T>(op, a, b)`
// is a single switch that constant-folds when `op` is a template constant (the "static"
// semiring fast paths) and stays a wave-uniform scalar branch when it is a runtime value
// (the generic path that covers every other built-in semiring).
//
// Semantics follow the GraphBLAS C API 1.3 + SuiteSparse v5.1 extensions that
// pygraphblas reflects over (reference: pygraphblas/binaryop.py:104-118 lists the operator
// names, pygraphblas/semiring.py:87-121 the semiring names, SURVEY.md App. A items 5,7).
#pragma once
#include <stdint.h>
#include <math.h>
#include <limits>
#include <type_traits>

#if defined(__HIPCC__)
#define GRB_HD __host__ __device__ __forceinline__
#else
#define GRB_HD inline
#endif

namespace grb {

enum TypeCode : int {
  T_BOOL = 0, T_INT8, T_UINT8, T_INT16, T_UINT16, T_INT32, T_UINT32, T_INT64, T_UINT64,
  T_FP32, T_FP64, T_FC32, T_FC64, T_UDT, T_NTYPES
};

GRB_HD int type_size(int code) {
  switch (code) {
    case T_BOOL: case T_INT8: case T_UINT8: return 1;
    case T_INT16: case T_UINT16: return 2;
    case T_INT32: case T_UINT32: case T_FP32: return 4;
    case T_INT64: case T_UINT64: case T_FP64: case T_FC32: return 8;
    case T_FC64: return 16;
    default: return 0;
  }
}

// Binary opcodes. z = f(x, y).
enum BinOpCode : int {
  B_FIRST = 0, B_SECOND, B_PAIR, B_ANY, B_MIN, B_MAX, B_PLUS, B_MINUS, B_RMINUS, B_TIMES,
  B_DIV, B_RDIV, B_POW, B_ISEQ, B_ISNE, B_ISGT, B_ISLT, B_ISGE, B_ISLE, B_LOR, B_LAND,
  B_LXOR,
  // comparison ops with BOOL result (ztype differs from xtype)
  B_EQ, B_NE, B_GT, B_LT, B_GE, B_LE,
  B_LXNOR,                       // bool only
  B_BOR, B_BAND, B_BXOR, B_BXNOR,  // integer bitwise
  B_ATAN2, B_HYPOT, B_FMOD, B_REMAINDER, B_COPYSIGN, B_LDEXP,  // float only
  B_BGET, B_BSET, B_BCLR, B_BSHIFT,                              // integer only
  B_FIRSTI, B_FIRSTI1, B_FIRSTJ, B_FIRSTJ1, B_SECONDI, B_SECONDI1, B_SECONDJ, B_SECONDJ1,
  B_CMPLX, B_USER, B_NOPS
};

enum UnOpCode : int {
  U_IDENTITY = 0, U_AINV, U_MINV, U_LNOT, U_ONE, U_ABS, U_BNOT, U_SQRT, U_LOG, U_EXP, U_LOG2,
  U_SIN, U_COS, U_TAN, U_ACOS, U_ASIN, U_ATAN, U_SINH, U_COSH, U_TANH, U_ACOSH, U_ASINH,
  U_ATANH, U_SIGNUM, U_CEIL, U_FLOOR, U_ROUND, U_TRUNC, U_EXP2, U_EXPM1, U_LOG10, U_LOG1P,
  U_LGAMMA, U_TGAMMA, U_ERF, U_ERFC, U_FREXPX, U_FREXPE, U_ISINF, U_ISNAN, U_ISFINITE,
  U_POSITIONI, U_POSITIONI1, U_POSITIONJ, U_POSITIONJ1, U_USER, U_NOPS
};

template <class T> struct is_bool : std::false_type {};
// BOOL is stored as one byte holding 0/1; kernels see it through this wrapper type so the
// boolean renaming of arithmetic ops (PLUS==LOR, TIMES==LAND, ...) is a compile-time choice.
struct bool8 {
  uint8_t v;
  bool8() = default;
  GRB_HD bool8(bool b) : v(b ? 1 : 0) {}
  GRB_HD operator bool() const { return v != 0; }
};
template <> struct is_bool<bool8> : std::true_type {};

template <class T> struct type_code_of;
template <> struct type_code_of<bool8>    { static constexpr int value = T_BOOL; };
template <> struct type_code_of<int8_t>   { static constexpr int value = T_INT8; };
template <> struct type_code_of<uint8_t>  { static constexpr int value = T_UINT8; };
template <> struct type_code_of<int16_t>  { static constexpr int value = T_INT16; };
template <> struct type_code_of<uint16_t> { static constexpr int value = T_UINT16; };
template <> struct type_code_of<int32_t>  { static constexpr int value = T_INT32; };
template <> struct type_code_of<uint32_t> { static constexpr int value = T_UINT32; };
template <> struct type_code_of<int64_t>  { static constexpr int value = T_INT64; };
template <> struct type_code_of<uint64_t> { static constexpr int value = T_UINT64; };
template <> struct type_code_of<float>    { static constexpr int value = T_FP32; };
template <> struct type_code_of<double>   { static constexpr int value = T_FP64; };

// ---- typecasting ------------------------------------------------
// C casts between integers; float->integer saturates with NaN->0; anything->BOOL is x != 0.
template <class Z, class X> GRB_HD Z cast_to(X x) {
  if constexpr (std::is_same<Z, X>::value) {
    return x;
  } else if constexpr (is_bool<Z>::value) {
    if constexpr (is_bool<X>::value) return x; else return bool8(x != (X)0);
  } else if constexpr (is_bool<X>::value) {
    return (Z)(x.v ? 1 : 0);
  } else if constexpr (std::is_floating_point<X>::value && std::is_integral<Z>::value) {
    if (x != x) return (Z)0;
    const double d = (double)x;
    const double lo = (double)std::numeric_limits<Z>::min();
    const double hi = (double)std::numeric_limits<Z>::max();
    if (d <= lo) return std::numeric_limits<Z>::min();
    if (d >= hi) return std::numeric_limits<Z>::max();
    return (Z)x;
  } else {
    return (Z)x;
  }
}

// ---- integer helpers with SuiteSparse's defined edge cases [upstream semantics] -----------
template <class T> GRB_HD T int_div(T x, T y) {
  if constexpr (std::is_signed<T>::value) {
    if (y == (T)-1) return (T)(0 - (typename std::make_unsigned<T>::type)x);  // avoids INT_MIN/-1 trap
    if (y == 0) return x == 0 ? (T)0 : (x < 0 ? std::numeric_limits<T>::min() : std::numeric_limits<T>::max());
    return (T)(x / y);
  } else {
    if (y == 0) return x == 0 ? (T)0 : std::numeric_limits<T>::max();
    return (T)(x / y);
  }
}

template <class T> GRB_HD T int_pow(T x, T y) {
  // SuiteSparse computes integer pow through double pow() and a saturating cast back.
  return cast_to<T, double>(pow((double)x, (double)y));
}

template <class T> GRB_HD T wrap_add(T a, T b) {
  if constexpr (std::is_integral<T>::value) {
    typedef typename std::make_unsigned<T>::type U;
    return (T)((U)a + (U)b);
  } else return a + b;
}
template <class T> GRB_HD T wrap_sub(T a, T b) {
  if constexpr (std::is_integral<T>::value) {
    typedef typename std::make_unsigned<T>::type U;
    return (T)((U)a - (U)b);
  } else return a - b;
}
template <class T> GRB_HD T wrap_mul(T a, T b) {
  if constexpr (std::is_integral<T>::value) {
    typedef typename std::make_unsigned<T>::type U;
    // promote through 64-bit so int16*int16 does not hit signed-overflow UB via int promotion
    return (T)(U)((uint64_t)(U)a * (uint64_t)(U)b);
  } else return a * b;
}

// ---- the binary operator switch, same-type ops (x, y, z all T) ----------------------------
// Comparison ops (B_EQ..B_LE) are returned as T-valued 0/1 here; callers that need the BOOL
// ztype cast the result.  i/j are the row/col coordinates for the positional ops.
// FULL = false: the multipliers and monoids of the semiring kernels (FIRST..LXOR) only.  MATH = false: everything but the
// operators that call into the math library (POW, ATAN2, HYPOT, ...): the O(n) kernels instantiate both and pick per call —
// one kernel carrying the whole switch ran a 4 M-element ABS in 133 us instead of 9.
template <class T, bool FULL = true, bool MATH = FULL> GRB_HD T apply_binop(int op, T x, T y) {
  if constexpr (is_bool<T>::value) {
    const bool a = x, b = y;
    switch (op) {
      case B_FIRST: case B_DIV: return x;
      case B_SECOND: case B_RDIV: case B_ANY: return y;
      case B_PAIR: return bool8(true);
      case B_MIN: case B_TIMES: case B_LAND: return bool8(a && b);
      case B_MAX: case B_PLUS: case B_LOR: return bool8(a || b);
      case B_MINUS: case B_RMINUS: case B_ISNE: case B_NE: case B_LXOR: return bool8(a != b);
      case B_ISEQ: case B_EQ: case B_LXNOR: return bool8(a == b);
      case B_ISGT: case B_GT: return bool8(a && !b);
      case B_ISLT: case B_LT: return bool8(!a && b);
      case B_ISGE: case B_GE: case B_POW: return bool8(a || !b);
      case B_ISLE: case B_LE: return bool8(!a || b);
      default: return bool8(false);
    }
  } else {
    switch (op) {
      case B_FIRST: return x;
      case B_SECOND: case B_ANY: return y;
      case B_PAIR: return (T)1;
      case B_MIN:
        if constexpr (std::is_floating_point<T>::value) return fmin(x, y); else return x < y ? x : y;
      case B_MAX:
        if constexpr (std::is_floating_point<T>::value) return fmax(x, y); else return x > y ? x : y;
      case B_PLUS: return wrap_add(x, y);
      case B_MINUS: return wrap_sub(x, y);
      case B_RMINUS: return wrap_sub(y, x);
      case B_TIMES: return wrap_mul(x, y);
      case B_DIV:
        if constexpr (std::is_floating_point<T>::value) return x / y; else return int_div(x, y);
      case B_RDIV:
        if constexpr (std::is_floating_point<T>::value) return y / x; else return int_div(y, x);
      case B_POW:
        if constexpr (!MATH) return (T)0;
        else if constexpr (std::is_floating_point<T>::value) return (T)pow((double)x, (double)y);
        else return int_pow(x, y);
      case B_ISEQ: case B_EQ: return (T)(x == y);
      case B_ISNE: case B_NE: return (T)(x != y);
      case B_ISGT: case B_GT: return (T)(x > y);
      case B_ISLT: case B_LT: return (T)(x < y);
      case B_ISGE: case B_GE: return (T)(x >= y);
      case B_ISLE: case B_LE: return (T)(x <= y);
      case B_LOR: return (T)((x != 0) || (y != 0));
      case B_LAND: return (T)((x != 0) && (y != 0));
      case B_LXOR: return (T)((x != 0) != (y != 0));
      default: break;
    }
    // semiring kernels (FULL = false): FIRST..LXOR plus the integer bitwise operators; everything else is refused by
    // make_semiring_desc before a kernel is chosen (semiring_op_supported below)
    if constexpr (std::is_integral<T>::value) {
      typedef typename std::make_unsigned<T>::type U;
      constexpr int bits = (int)sizeof(T) * 8;
      switch (op) {
        case B_BOR: return (T)((U)x | (U)y);
        case B_BAND: return (T)((U)x & (U)y);
        case B_BXOR: return (T)((U)x ^ (U)y);
        case B_BXNOR: return (T)~((U)x ^ (U)y);
        default: break;
      }
      if constexpr (!FULL) return (T)0;
      else switch (op) {
        case B_BGET: { int64_t k = (int64_t)y; return (k >= 1 && k <= bits) ? (T)(((U)x >> (k - 1)) & 1) : (T)0; }
        case B_BSET: { int64_t k = (int64_t)y; return (k >= 1 && k <= bits) ? (T)((U)x | ((U)1 << (k - 1))) : x; }
        case B_BCLR: { int64_t k = (int64_t)y; return (k >= 1 && k <= bits) ? (T)((U)x & ~((U)1 << (k - 1))) : x; }
        default: return (T)0;
      }
    } else if constexpr (!MATH) return (T)0;
    else {
      switch (op) {
        case B_ATAN2: return (T)atan2((double)x, (double)y);
        case B_HYPOT: return (T)hypot((double)x, (double)y);
        case B_FMOD: return (T)fmod((double)x, (double)y);
        case B_REMAINDER: return (T)remainder((double)x, (double)y);
        case B_COPYSIGN: return (T)copysign((double)x, (double)y);
        case B_LDEXP: return (T)ldexp((double)x, (int)y);
        default: return (T)0;
      }
    }
  }
}

// can the semiring kernels (apply_binop<T, false>) evaluate this operator on values of type `code`?
GRB_HD bool semiring_op_supported(int op, int code) {
  if (op <= B_LXOR) return op != B_POW || code == T_BOOL;
  if (op >= B_BOR && op <= B_BXNOR) return code >= T_INT8 && code <= T_UINT64;
  return code == T_BOOL && (op <= B_LXNOR);
}
GRB_HD bool binop_is_compare(int op) { return op >= B_EQ && op <= B_LE; }
GRB_HD bool binop_needs_math(int op) { return op == B_POW || (op >= B_ATAN2 && op <= B_LDEXP); }
GRB_HD bool binop_is_positional(int op) { return op >= B_FIRSTI && op <= B_SECONDJ1; }
// which inputs does the multiplier actually read?  (SURVEY.md App. C "kernel consequences")
GRB_HD bool binop_uses_x(int op) { return !(op == B_SECOND || op == B_PAIR || op == B_ANY || binop_is_positional(op)); }
GRB_HD bool binop_uses_y(int op) { return !(op == B_FIRST || op == B_PAIR || binop_is_positional(op)); }

template <class T, bool MATH = true> GRB_HD T apply_unop(int op, T x) {
  if constexpr (is_bool<T>::value) {
    switch (op) {
      case U_LNOT: return bool8(!(bool)x);
      case U_ONE: return bool8(true);
      default: return x;  // IDENTITY, AINV, MINV, ABS are identity on BOOL
    }
  } else {
    switch (op) {
      case U_IDENTITY: return x;
      case U_AINV: return wrap_sub((T)0, x);
      case U_MINV:
        if constexpr (std::is_floating_point<T>::value) return (T)1 / x; else return int_div((T)1, x);
      case U_LNOT: return (T)(x == 0);
      case U_ONE: return (T)1;
      case U_ABS:
        if constexpr (std::is_floating_point<T>::value) return (T)fabs((double)x);
        else if constexpr (std::is_signed<T>::value) return x < 0 ? wrap_sub((T)0, x) : x;
        else return x;
      case U_BNOT:
        if constexpr (std::is_integral<T>::value) return (T)~x; else return x;
      default: break;
    }
    if constexpr (!MATH) return x;
    else if constexpr (std::is_floating_point<T>::value) {
      const double d = (double)x;
      switch (op) {
        case U_SQRT: return (T)sqrt(d);  case U_LOG: return (T)log(d);   case U_EXP: return (T)exp(d);
        case U_LOG2: return (T)log2(d);  case U_SIN: return (T)sin(d);   case U_COS: return (T)cos(d);
        case U_TAN: return (T)tan(d);    case U_ACOS: return (T)acos(d); case U_ASIN: return (T)asin(d);
        case U_ATAN: return (T)atan(d);  case U_SINH: return (T)sinh(d); case U_COSH: return (T)cosh(d);
        case U_TANH: return (T)tanh(d);  case U_ACOSH: return (T)acosh(d); case U_ASINH: return (T)asinh(d);
        case U_ATANH: return (T)atanh(d);
        case U_SIGNUM: return (T)(d != d ? d : (d > 0) - (d < 0));
        case U_CEIL: return (T)ceil(d);  case U_FLOOR: return (T)floor(d); case U_ROUND: return (T)round(d);
        case U_TRUNC: return (T)trunc(d); case U_EXP2: return (T)exp2(d); case U_EXPM1: return (T)expm1(d);
        case U_LOG10: return (T)log10(d); case U_LOG1P: return (T)log1p(d); case U_LGAMMA: return (T)lgamma(d);
        case U_TGAMMA: return (T)tgamma(d); case U_ERF: return (T)erf(d); case U_ERFC: return (T)erfc(d);
        case U_FREXPX: { int e; return (T)frexp(d, &e); }
        case U_FREXPE: { int e; (void)frexp(d, &e); return (T)e; }
        case U_ISINF: return (T)(isinf(d) ? 1 : 0);
        case U_ISNAN: return (T)(d != d ? 1 : 0);
        case U_ISFINITE: return (T)(isfinite(d) ? 1 : 0);
        default: return x;
      }
    }
    return x;
  }
}

// ---- monoid identities / terminals ---------------------------------------------------------
template <class T> GRB_HD T monoid_identity(int op) {
  if constexpr (is_bool<T>::value) {
    switch (op) {
      case B_LAND: case B_MIN: case B_TIMES: case B_EQ: case B_LXNOR: case B_ISEQ: return bool8(true);
      default: return bool8(false);  // LOR, LXOR, PLUS, MAX, ANY
    }
  } else {
    switch (op) {
      case B_MIN:
        if constexpr (std::is_floating_point<T>::value) return (T)INFINITY; else return std::numeric_limits<T>::max();
      case B_MAX:
        if constexpr (std::is_floating_point<T>::value) return (T)-INFINITY; else return std::numeric_limits<T>::min();
      case B_TIMES: return (T)1;
      case B_BAND: case B_BXNOR:
        if constexpr (std::is_integral<T>::value) return (T)~(T)0; else return (T)0;
      default: return (T)0;  // PLUS, ANY, LOR, LXOR, BOR, BXOR
    }
  }
}

// returns true and sets *t when the monoid has a terminal ("annihilator") value
template <class T> GRB_HD bool monoid_terminal(int op, T* t) {
  if constexpr (is_bool<T>::value) {
    switch (op) {
      case B_LOR: case B_MAX: case B_PLUS: *t = bool8(true); return true;
      case B_LAND: case B_MIN: case B_TIMES: *t = bool8(false); return true;
      case B_ANY: *t = bool8(false); return true;
      default: return false;
    }
  } else {
    switch (op) {
      case B_MIN:
        if constexpr (std::is_floating_point<T>::value) *t = (T)-INFINITY; else *t = std::numeric_limits<T>::min();
        return true;
      case B_MAX:
        if constexpr (std::is_floating_point<T>::value) *t = (T)INFINITY; else *t = std::numeric_limits<T>::max();
        return true;
      case B_TIMES:
        if constexpr (std::is_integral<T>::value) { *t = (T)0; return true; } else return false;
      case B_ANY: *t = (T)0; return true;
      case B_BOR:
        if constexpr (std::is_integral<T>::value) { *t = (T)~(T)0; return true; } else return false;
      case B_BAND:
        if constexpr (std::is_integral<T>::value) { *t = (T)0; return true; } else return false;
      default: return false;
    }
  }
}

// Run `f.template operator()<T>()` for the C type behind a real type code.
template <class F> inline bool dispatch_type(int code, F&& f) {
  switch (code) {
    case T_BOOL: f.template operator()<bool8>(); return true;
    case T_INT8: f.template operator()<int8_t>(); return true;
    case T_UINT8: f.template operator()<uint8_t>(); return true;
    case T_INT16: f.template operator()<int16_t>(); return true;
    case T_UINT16: f.template operator()<uint16_t>(); return true;
    case T_INT32: f.template operator()<int32_t>(); return true;
    case T_UINT32: f.template operator()<uint32_t>(); return true;
    case T_INT64: f.template operator()<int64_t>(); return true;
    case T_UINT64: f.template operator()<uint64_t>(); return true;
    case T_FP32: f.template operator()<float>(); return true;
    case T_FP64: f.template operator()<double>(); return true;
    default: return false;
  }
}

}  // namespace grb
