// grb_exact.hpp — order-independent floating-point PLUS: the accumulators of the masked product's deterministic mode (round 5).
//
// A sum of doubles is the same bits in every run if it is formed in integers.  Every output row i gets a unit 2^u(i) such that no sum of the row's products can
// leave a signed 128-bit count of units: u(i) = E + H - 126 with 2^E above the row's largest possible |product| (k_row_unit_exp: the multiply applied to
// the largest |A(i,k)| and the largest |B|, rounded as the product itself is) and 2^H above the number of products one entry can receive (the entries of A(i,:)).
// A product p = m 2^e (m: its 53-bit significand) adds m shifted by e - u(i) <= 73 - H bits: two 64-bit integer atomics, the second only when the high
// word changes (a negative term, a carry).  k_row_unit_exp admits a row only if no product of it can have a bit below 2^u(i) (its operands are spread over
// less than ~2^(74-H)), so whatever order the atomics land in, the accumulator ends as the EXACT sum of the products (each rounded once, by the multiply, as
// always), and the result is that integer rounded ONCE to the value type (nearest, ties to even): what math.fsum of the products returns, which the tests
// compare bit by bit.  Rows that are not admitted (operands spread too far, an Inf or NaN operand) are left to k_spgemm_masked_ordered.
#pragma once
#include "grb_ops.hpp"
#include <string.h>

namespace grb {

constexpr int32_t FX_NO_EXP = INT32_MIN;      // row marker: the bound of the row's products is not finite

GRB_HD void fx_from_double(const double p, const int unit_exp, unsigned long long& lo, unsigned long long& hi) {
  unsigned long long b; memcpy(&b, &p, 8);
  const int be = (int)((b >> 52) & 0x7FFu);
  unsigned long long m = b & 0xFFFFFFFFFFFFFull; int e = -1074;
  if (be) { m |= 1ull << 52; e = be - 1075; }
  int sh = e - unit_exp; if (sh > 73) sh = 73;        // (never: the row's bound)
  unsigned long long l = 0, h = 0;
  if (sh >= 64) h = m << (sh - 64);
  else if (sh > 0) { l = m << sh; h = m >> (64 - sh); }
  else if (sh == 0) l = m;
  else if (sh > -64) l = m >> (-sh);
  if (b >> 63) { l = ~l + 1ull; h = ~h + (l == 0 ? 1ull : 0ull); }
  lo = l; hi = h;
}
#if defined(__HIPCC__)
// (lo, hi) += (xl, xh) as one 128-bit two's-complement add made of two independent integer atomics: exact in any interleaving (LDS or global memory)
__device__ __forceinline__ void fx_add_words(unsigned long long* lo, unsigned long long* hi, const unsigned long long xl, unsigned long long xh) {
  if (xl) { const unsigned long long old = atomicAdd(lo, xl); if (old + xl < xl) xh++; }
  if (xh) atomicAdd(hi, xh);
}
__device__ __forceinline__ void fx_add(unsigned long long* lo, unsigned long long* hi, const double p, const int unit_exp) {
  unsigned long long xl, xh; fx_from_double(p, unit_exp, xl, xh);
  fx_add_words(lo, hi, xl, xh);
}
// the same on an LDS accumulator as ONE out-of-line function (the kernels' hit path is inlined at some forty places: ~100 instructions each doubled their
// code, and four of them share the CUs' instruction caches); the LDS addresses travel as 32-bit offsets so the atomics stay ds_add_*
typedef __attribute__((address_space(3))) unsigned long long* fx_lds_ptr;
__device__ __attribute__((noinline)) inline void fx_add_lds(const uint32_t lo_addr, const uint32_t hi_addr, const double p, const int unit_exp) {
  unsigned long long xl, xh; fx_from_double(p, unit_exp, xl, xh);
  fx_lds_ptr lo = (fx_lds_ptr)(uintptr_t)lo_addr; fx_lds_ptr hi = (fx_lds_ptr)(uintptr_t)hi_addr;
  if (xl) { const unsigned long long old = __hip_atomic_fetch_add(lo, xl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); if (old + xl < xl) xh++; }
  if (xh) __hip_atomic_fetch_add(hi, xh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t fx_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }
#endif
// the integer rounded once to P significant bits (53: double, 24: float), nearest / ties to even; returned as a double (exactly representable)
template <int P> GRB_HD double fx_to_fp(unsigned long long lo, unsigned long long hi, const int unit_exp) {
  const bool neg = (hi >> 63) != 0;
  if (neg) { lo = ~lo + 1ull; hi = ~hi + (lo == 0 ? 1ull : 0ull); }
  if (!(lo | hi)) return 0.0;
  const int msb = hi ? 127 - __builtin_clzll(hi) : 63 - __builtin_clzll(lo);
  //  a result that is SUBNORMAL in the target type keeps fewer than P bits: its last place is 2^-1074 (double) / 2^-149 (float) — rounding to P
  // bits first and letting ldexp / the cast to float round again would round twice.  One rounding, at the place the target type really ends.
  // (A sum whose terms are all -0.0 comes out as +0.0: the integer accumulator has one zero.  IEEE addition would give -0.0; the two compare equal.)
  int s = msb + 1 - P;
  { const int smin = (P == 53 ? -1074 : -149) - unit_exp; if (smin > s) s = smin; }
  if (s > 127) return neg ? -0.0 : 0.0;                      // below half of the smallest subnormal
  unsigned long long mant = lo; int ex = unit_exp;
  if (s > 0) {
    unsigned long long q = s >= 64 ? hi >> (s - 64) : ((lo >> s) | (hi << (64 - s)));
    const int hb = s - 1;
    bool half, sticky;
    if (hb >= 64) { half = ((hi >> (hb - 64)) & 1ull) != 0; sticky = lo != 0 || (hb > 64 && (hi & ((1ull << (hb - 64)) - 1ull)) != 0); }
    else { half = ((lo >> hb) & 1ull) != 0; sticky = hb > 0 && (lo & ((1ull << hb) - 1ull)) != 0; }
    if (half && (sticky || (q & 1ull))) q++;
    mant = q; ex += s;
  }
  const double r = ldexp((double)mant, ex);
  return neg ? -r : r;
}
template <class T> struct fx_bits { static constexpr int P = sizeof(T) == 8 ? 53 : 24; };

}  // namespace grb
