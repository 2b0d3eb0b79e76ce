// grb_atomics.hpp — "combine v into *addr with a monoid", for accumulators in LDS or in HBM.
// Native atomics where the hardware has them (add / min / max on 32- and 64-bit integers, add on
// f32 / f64, byte stores for BOOL or/and); a compare-and-swap loop on the containing 32/64-bit word
// otherwise.  Accumulators of 1- and 2-byte types are held in 32-bit words (`acc_word`).
#pragma once
#include "grb_ops.hpp"
#include <hip/hip_runtime.h>

namespace grb {

template <class T> struct acc_word { typedef typename std::conditional<sizeof(T) == 8, unsigned long long, unsigned int>::type type; };

template <class T> __host__ __device__ __forceinline__ typename acc_word<T>::type to_word(T v) {
  typename acc_word<T>::type w = 0; __builtin_memcpy(&w, &v, sizeof(T)); return w;
}
template <class T> __host__ __device__ __forceinline__ T from_word(typename acc_word<T>::type w) { T v; __builtin_memcpy(&v, &w, sizeof(T)); return v; }

template <class T> __device__ __forceinline__ void word_combine(int op, typename acc_word<T>::type* addr, T v) {
  typedef typename acc_word<T>::type W;
  if constexpr (is_bool<T>::value) {
    if (op == B_LOR || op == B_PLUS || op == B_MAX) { if (v) *addr = 1; return; }
    if (op == B_LAND || op == B_TIMES || op == B_MIN) { if (!v) *addr = 0; return; }
    if (op == B_ANY) { *addr = to_word<T>(v); return; }
  } else if constexpr (sizeof(T) >= 4 && std::is_integral<T>::value) {
    if (op == B_PLUS) { atomicAdd(addr, (W)v); return; }
    if (op == B_ANY) { *addr = (W)v; return; }
    if constexpr (std::is_signed<T>::value) {
      typedef typename std::conditional<sizeof(T) == 8, long long, int>::type S;
      if (op == B_MIN) { atomicMin((S*)addr, (S)v); return; }
      if (op == B_MAX) { atomicMax((S*)addr, (S)v); return; }
    } else {
      if (op == B_MIN) { atomicMin(addr, (W)v); return; }
      if (op == B_MAX) { atomicMax(addr, (W)v); return; }
    }
  } else if constexpr (std::is_floating_point<T>::value) {
    if (op == B_PLUS) { atomicAdd((T*)addr, v); return; }
    if (op == B_ANY) { *addr = to_word<T>(v); return; }
  }
  W old = *addr, assumed;
  do {
    assumed = old;
    const T nv = apply_binop<T, false>(op, from_word<T>(assumed), v);
    const W nw = to_word<T>(nv);
    if (nw == assumed) break;
    old = atomicCAS(addr, assumed, nw);
  } while (old != assumed);
}

}  // namespace grb
