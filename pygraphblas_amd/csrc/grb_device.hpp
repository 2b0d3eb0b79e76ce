// grb_device.hpp — device-side helpers shared by the HIP kernels (wave64 only: gfx950).
#pragma once
#include "grb_internal.hpp"
#include <string.h>

namespace grb {

// ---- 64-lane cross-lane primitives ----------------------------------------------------------------
template <class T> __device__ __forceinline__ T shfl_xor_t(T v, int m) {
  if constexpr (sizeof(T) == 8) {
    union { T t; int i[2]; } u; u.t = v;
    u.i[0] = __shfl_xor(u.i[0], m, 64); u.i[1] = __shfl_xor(u.i[1], m, 64); return u.t;
  } else if constexpr (sizeof(T) == 4) {
    union { T t; int i; } u; u.t = v; u.i = __shfl_xor(u.i, m, 64); return u.t;
  } else {
    union { T t; uint16_t s; uint8_t b; } u; u.s = 0; u.t = v;
    int x = __shfl_xor((int)u.s, m, 64); u.s = (uint16_t)x; return u.t;
  }
}
template <class T> __device__ __forceinline__ T shfl_down_t(T v, int d) {
  if constexpr (sizeof(T) == 8) {
    union { T t; int i[2]; } u; u.t = v;
    u.i[0] = __shfl_down(u.i[0], d, 64); u.i[1] = __shfl_down(u.i[1], d, 64); return u.t;
  } else if constexpr (sizeof(T) == 4) {
    union { T t; int i; } u; u.t = v; u.i = __shfl_down(u.i, d, 64); return u.t;
  } else {
    union { T t; uint16_t s; } u; u.s = 0; u.t = v;
    int x = __shfl_down((int)u.s, d, 64); u.s = (uint16_t)x; return u.t;
  }
}

template <class T> __device__ __forceinline__ T shfl_up_t(T v, int d) {
  if constexpr (sizeof(T) == 8) {
    union { T t; int i[2]; } u; u.t = v;
    u.i[0] = __shfl_up(u.i[0], d, 64); u.i[1] = __shfl_up(u.i[1], d, 64); return u.t;
  } else if constexpr (sizeof(T) == 4) {
    union { T t; int i; } u; u.t = v; u.i = __shfl_up(u.i, d, 64); return u.t;
  } else {
    union { T t; uint16_t s; } u; u.s = 0; u.t = v;
    int x = __shfl_up((int)u.s, d, 64); u.s = (uint16_t)x; return u.t;
  }
}

template <class T> __device__ __forceinline__ T shfl_t(T v, int src) {
  if constexpr (sizeof(T) == 8) {
    union { T t; int i[2]; } u; u.t = v;
    u.i[0] = __shfl(u.i[0], src, 64); u.i[1] = __shfl(u.i[1], src, 64); return u.t;
  } else if constexpr (sizeof(T) == 4) {
    union { T t; int i; } u; u.t = v; u.i = __shfl(u.i, src, 64); return u.t;
  } else {
    union { T t; uint16_t s; } u; u.s = 0; u.t = v;
    int x = __shfl((int)u.s, src, 64); u.s = (uint16_t)x; return u.t;
  }
}

__device__ __forceinline__ unsigned long long wave_reduce_add_u64(unsigned long long v) {
  return __builtin_amdgcn_wave_reduce_add_u64(v, 0);
}

// All-lanes monoid reduction.  Integer PLUS/MIN/MAX/LOR use the gfx950 scalar wave-reduce
// builtins (there is no floating-point variant in this toolchain: SURVEY.md §0), everything
// else a fixed xor-butterfly, so the combination order is identical on every run.
template <class T, bool FULL = true> __device__ __forceinline__ T wave_reduce_op(int op, T v) {
  if constexpr (std::is_integral<T>::value && sizeof(T) >= 4) {
    if (op == B_PLUS) {
      if constexpr (sizeof(T) == 8) return (T)__builtin_amdgcn_wave_reduce_add_u64((uint64_t)v, 0);
      else return (T)__builtin_amdgcn_wave_reduce_add_u32((uint32_t)v, 0);
    }
    if (op == B_MIN) {
      if constexpr (std::is_same<T, int64_t>::value) return __builtin_amdgcn_wave_reduce_min_i64(v, 0);
      else if constexpr (std::is_same<T, uint64_t>::value) return __builtin_amdgcn_wave_reduce_min_u64(v, 0);
      else if constexpr (std::is_same<T, int32_t>::value) return __builtin_amdgcn_wave_reduce_min_i32(v, 0);
      else return __builtin_amdgcn_wave_reduce_min_u32(v, 0);
    }
    if (op == B_MAX) {
      if constexpr (std::is_same<T, int64_t>::value) return __builtin_amdgcn_wave_reduce_max_i64(v, 0);
      else if constexpr (std::is_same<T, uint64_t>::value) return __builtin_amdgcn_wave_reduce_max_u64(v, 0);
      else if constexpr (std::is_same<T, int32_t>::value) return __builtin_amdgcn_wave_reduce_max_i32(v, 0);
      else return __builtin_amdgcn_wave_reduce_max_u32(v, 0);
    }
  }
  if constexpr (is_bool<T>::value) {
    if (op == B_LOR || op == B_PLUS || op == B_MAX || op == B_ANY)
      return bool8(__builtin_amdgcn_wave_reduce_or_b32((uint32_t)v.v, 0) != 0);
    if (op == B_LAND || op == B_TIMES || op == B_MIN)
      return bool8(__builtin_amdgcn_wave_reduce_and_b32((uint32_t)v.v, 0) != 0);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = apply_binop<T, FULL>(op, v, shfl_xor_t<T>(v, m));
  return v;
}

// ---- DPP wave reduction (no LDS crossbar): quad swaps, row rotates, then the two row broadcasts of gfx9.
// The combination order is fixed, so floating-point results are reproducible.  Lanes that have nothing to add must
// pass the monoid identity.  Returns the total in every lane (read back from lane 63).
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_mov(uint32_t old, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xf, false);
}
template <class T, int CTRL, int ROW_MASK> __device__ __forceinline__ T dpp_move_t(T v) {
  // lanes that are not written (row_mask) or have no valid source keep their own value as `old`
  if constexpr (sizeof(T) == 8) {
    union { T t; uint32_t u[2]; } a, r; a.t = v;
    r.u[0] = dpp_mov<CTRL, ROW_MASK>(a.u[0], a.u[0]); r.u[1] = dpp_mov<CTRL, ROW_MASK>(a.u[1], a.u[1]); return r.t;
  } else {
    union { T t; uint32_t u; } a, r; a.u = 0; a.t = v; r.u = dpp_mov<CTRL, ROW_MASK>(a.u, a.u); return r.t;
  }
}
template <class T, bool FULL = true> __device__ __forceinline__ T wave_reduce_dpp(int op, T v, T identity) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "DPP reduction handles 4- and 8-byte types");
  v = apply_binop<T, FULL>(op, dpp_move_t<T, 0xB1, 0xf>(v), v);     // quad_perm [1,0,3,2]
  v = apply_binop<T, FULL>(op, dpp_move_t<T, 0x4E, 0xf>(v), v);     // quad_perm [2,3,0,1]
  v = apply_binop<T, FULL>(op, dpp_move_t<T, 0x124, 0xf>(v), v);    // row_ror:4
  v = apply_binop<T, FULL>(op, dpp_move_t<T, 0x128, 0xf>(v), v);    // row_ror:8   -> every lane of a 16-lane row holds the row total
  {                                                                 // row_bcast:15 into rows 1 and 3 (others must not change)
    const T m = dpp_move_t<T, 0x142, 0xa>(v);
    const int row = (threadIdx.x >> 4) & 3;
    if (row == 1 || row == 3) v = apply_binop<T, FULL>(op, m, v);
  }
  {                                                                 // row_bcast:31 into rows 2 and 3
    const T m = dpp_move_t<T, 0x143, 0xc>(v);
    const int row = (threadIdx.x >> 4) & 3;
    if (row >= 2) v = apply_binop<T, FULL>(op, m, v);
  }
  (void)identity;
  if constexpr (sizeof(T) == 8) {
    union { T t; uint32_t u[2]; } a, r; a.t = v;
    r.u[0] = (uint32_t)__builtin_amdgcn_readlane((int)a.u[0], 63); r.u[1] = (uint32_t)__builtin_amdgcn_readlane((int)a.u[1], 63); return r.t;
  } else {
    union { T t; uint32_t u; } a, r; a.u = 0; a.t = v; r.u = (uint32_t)__builtin_amdgcn_readlane((int)a.u, 63); return r.t;
  }
}

// sub-wave (power-of-two group of G lanes) reduction; result valid in the group's lane 0
template <class T, int G, bool FULL = true> __device__ __forceinline__ T group_reduce_op(int op, T v) {
#pragma unroll
  for (int d = G / 2; d >= 1; d >>= 1) v = apply_binop<T, FULL>(op, v, shfl_down_t<T>(v, d));
  return v;
}

template <class T> __device__ __forceinline__ bool val_eq(T a, T b) {
  if constexpr (is_bool<T>::value) return a.v == b.v; else return a == b;
}
#define memcmp_eq(a, b) ::grb::val_eq(a, b)

// ---- a mask entry's truth from the mask vector itself (value cast to BOOL: x != 0; -0.0 is false, NaN true) -------------------------
// `mcode` is wave-uniform: one scalar branch.  Lets a kernel read the mask directly instead of "allow" bytes a k_allow pass prepared.
__device__ __forceinline__ bool mask_truth_at(const void* mval, int mcode, uint64_t p, bool structural) {
  if (structural) return true;
  switch (type_size(mcode)) {
    case 1: return ((const uint8_t*)mval)[p] != 0;
    case 2: return ((const uint16_t*)mval)[p] != 0;
    case 4: return mcode == T_FP32 ? ((const float*)mval)[p] != 0.0f : ((const uint32_t*)mval)[p] != 0;
    default: return mcode == T_FP64 ? ((const double*)mval)[p] != 0.0 : ((const uint64_t*)mval)[p] != 0;
  }
}

// ---- a device word the host can read back (count results, flags) ------------------------------------------
// 16 KiB of page-locked host memory per thread: the landing place of the small device-to-host readbacks (counts, reduced scalars, the
// result summary of a BOOL product: 8.1 KiB)
inline void* pinned_scratch() {
  static thread_local void* p = nullptr;
  if (!p) GRB_HIP(hipHostMalloc(&p, 16384, hipHostMallocDefault));
  return p;
}
// Round 6 (second half): the counters reach the host WITHOUT a copy, a memset and a stream synchronisation.  Every workgroup of a counting kernel adds into the
// device slot and then draws a ticket; the workgroup that draws the last one reads the totals, zeroes slot and ticket for the next kernel and stores
// (v0, v1, tag) into three page-locked, GPU-visible HOST words; the host spins on the tag.  Before: hipMemcpyAsync (a blit kernel) + hipMemsetAsync (another)
// + hipStreamSynchronize per readback — three of them in every sweep of the shortest-path loop's `iseq`, one per level of everything that asks `nvals`.
struct ScalarPub { unsigned long long* slot; unsigned long long* host; unsigned long long tag; };      // slot: [0], [1] counters, [2] ticket (all zero between kernels)
struct ScalarSlot {
  struct State { DevBuf dev; unsigned long long* host = nullptr; unsigned long long* host_dev = nullptr; unsigned long long seq = 0; };
  static State& st() {
    static thread_local State s;
    if (!s.dev.p) {
      s.dev.alloc(32); GRB_HIP(hipMemsetAsync(s.dev.p, 0, 32, stream()));
      void* h = nullptr; GRB_HIP(hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent)); memset(h, 0, 64); s.host = (unsigned long long*)h;      // (lives as long as the thread's other pinned scratch)
      void* dp = nullptr; if (hipHostGetDevicePointer(&dp, h, 0) != hipSuccess) { (void)hipGetLastError(); dp = h; }
      s.host_dev = (unsigned long long*)dp;
    }
    return s;
  }
  unsigned long long tag = 0;
  ScalarPub pub() { State& s = st(); tag = ++s.seq; return ScalarPub{(unsigned long long*)s.dev.p, s.host_dev, tag}; }      // one kernel launch per pub()
  void zero() {}
  void read(uint64_t out[2]) {
    State& s = st();
    volatile unsigned long long* h = s.host;
    for (int round = 0; round < 2; round++) {
      for (uint64_t spin = 0; spin < (1ull << 24); spin++) { if (h[2] == tag) { out[0] = h[0]; out[1] = h[1]; return; } __builtin_ia32_pause(); }
      GRB_HIP(hipStreamSynchronize(stream()));      // (far beyond any counting kernel's time: let a failed launch report itself, then look once more)
    }
    fail(GrB_PANIC, "a counting kernel did not publish its result");
  }
  uint64_t read_u64() { uint64_t v[2]; read(v); return v[0]; }
};
#if defined(__HIPCC__)
// thread 0 of EVERY workgroup, after the workgroup's own atomics into p.slot (same thread: program order + the fence)
__device__ __forceinline__ void scalar_publish(const ScalarPub& p) {
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned long long t = atomicAdd(&p.slot[2], 1ull);
    if (t == (unsigned long long)gridDim.x - 1ull) {
      __threadfence();
      const unsigned long long v0 = atomicExch(&p.slot[0], 0ull), v1 = atomicExch(&p.slot[1], 0ull);
      atomicExch(&p.slot[2], 0ull);
      volatile unsigned long long* h = p.host;
      h[0] = v0; h[1] = v1; __threadfence_system(); h[2] = p.tag;
    }
  }
}
#endif

// ---- library-backed primitives (grb_prims.hip: rocPRIM scan / radix sort) ------------------------------------
void exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint64_t n);            // out[i] = sum in[0..i)
void inclusive_scan_max_u32(const uint32_t* in, uint32_t* out, uint64_t n);        // out[i] = max in[0..i]   (in == out allowed)
void exclusive_scan_u64(const uint64_t* in, uint64_t* out, uint64_t n);                // out[i] = sum in[0..i)
void exclusive_scan_max_u64(const uint64_t* in, uint64_t* out, uint64_t n);        // out[i] = max(0, in[0..i))
void sort_pairs_u32(const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, uint64_t n, int end_bit);
void sort_keys_u32(const uint32_t* kin, uint32_t* kout, uint64_t n, int end_bit);
void sort_pairs_u64(const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout, uint64_t n, int end_bit);

// ---- kernels in grb_vecops.hip ---------------------------------------------------------------------------------
void dev_copy2(void* d0, const void* s0, uint64_t n0, void* d1, const void* s1, uint64_t n1);      // two device-to-device copies (byte counts), one launch
uint64_t vec_iseq_mismatches(int code, uint64_t n, const void* uval, const uint8_t* upres, const void* vval, const uint8_t* vpres);      // positions where the patterns or the present values of two bitmap vectors of one type differ (one kernel, one polled read-back)
// min / max of the present finite values (in the type itself), how many present values are NaN or infinite, how many are present;
// false for types other than INT32 / INT64 / FP32 / FP64
bool value_range(int code, uint64_t n, const void* val, const uint8_t* pres, void* vmin, void* vmax, uint64_t* nonfinite, uint64_t* count);
void big_to_absent(int code, uint64_t n, const void* val, uint8_t* pres, const void* thresh, bool keep_below);   // pres[i] = 0 where val[i] is not strictly below / above thresh
void vec_epilogue(int code, uint64_t n, void* wval, uint8_t* wpres, const void* tval, const uint8_t* tpres,
                  const uint8_t* allow, int accum, bool replace);
void reduce_values(int code, uint64_t n, const void* val, const uint8_t* pres, int op, const void* identity, void* result_host);
void reduce_values_f32_f64(uint64_t n, const void* val_f32, const uint8_t* pres, int op, const void* identity_f64, void* result_host);
void vec_ewise(int code, uint64_t n, const void* uval, const uint8_t* upres, const void* vval, const uint8_t* vpres, int op,
               bool is_union, void* tval, uint8_t* tpres);
void vec_apply(int code, uint64_t n, const void* uval, const uint8_t* upres, int mode, int op, const void* scalar, void* tval, uint8_t* tpres);
// w<mask, replace> = accum(w, u op v) in one pass, everything in the type `code`; mpres == nullptr: no mask; accum < 0: none; w may alias u, v or the mask
void vec_ewise_fused(int code, uint64_t n, const void* uval, const uint8_t* upres, const void* vval, const uint8_t* vpres, int op, bool is_union,
                     int mcode, const void* mval, const uint8_t* mpres, bool mstruct, bool mcomp, int accum, bool replace, void* wval, uint8_t* wpres);
void vec_assign_scalar(int code, uint64_t n, void* wval, uint8_t* wpres, const uint8_t* allow, const uint8_t* region, const void* scalar, int accum, bool replace);
// the same over every index with the mask vector read in place (no "allow" pass): `v.assign_scalar(level, mask=q)` of a BFS level is one kernel
bool vec_assign_scalar_masked(int code, uint64_t n, void* wval, uint8_t* wpres, int mcode, const void* mval, const uint8_t* mpres, bool mstruct, bool mcomp, const void* scalar, int accum, bool replace,
                              uint8_t* code_out = nullptr);      // true: the code bytes (GrB_Vector_opaque::dcode) of every position were written into code_out on the way
void vec_code_bytes(uint64_t n, const uint8_t* val, const uint8_t* pres, uint8_t* code);      // ... from scratch, for a one-byte-typed vector
// allow bytes of a mask AND its values cast to BOOL in one pass (the mask is also the operand: `v.vxm(A, mask=v, desc=RC)` with a BOOL semiring)
void build_allow_and_bool(uint64_t n, int mcode, const void* mval, const uint8_t* mpres, bool structural, bool complement, uint8_t* allow, uint8_t* as_bool);
void select_value_flags(int code, uint64_t n, const void* val, const uint8_t* pres, int sel, const void* thunk, uint8_t* keep);

}  // namespace grb
