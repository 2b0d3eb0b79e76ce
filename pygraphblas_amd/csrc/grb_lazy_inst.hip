// grb_lazy_inst.hip — k_vec_chain (compiled once per value type, -DGRB_INST_TYPE=..., like the SpMV kernels): a short queue of element-wise vector operations (grb_lazy.cpp) as ONE streaming pass over
// the bitmap layout (val T[n] | present u8[n]), optionally with the monoid reduction of the last result on the way.
//
// HBM-bound like every O(n) companion: per position the stored operands are read once (the presence bytes of a
// full operand are not read at all), every step combines registers, the results that some vector must hold afterwards are
// written once — zeros where a result has no entry, so that a later product may gather from the values without a fill pass.
// `t -= r; t = abs(t); t.reduce_float()` (gap/prmark.py:24-26) moves 3 x 4 B per vertex instead of 8 x 4 + 6 x 1 in four kernels.
// The operator codes are wave-uniform kernel arguments: one scalar branch per step, as in the runtime-opcode semirings.
#include <atomic>
#include <type_traits>
#include "grb_api.hpp"
#include "grb_device.hpp"
#include "grb_lazy.hpp"

namespace grb {

template <class T> struct ChainK {
  uint64_t n; int nsteps, next;
  const T* ev[CHAIN_MAX_IN]; const uint8_t* ep[CHAIN_MAX_IN];      // always valid addresses (unused slots repeat slot 0): the loads are unconditional
  uint32_t full_mask, used_mask;                                    // bit k: operand k has every position present / slot k is in use
  T* ov[CHAIN_MAX_OUT]; uint8_t* op[CHAIN_MAX_OUT];
  int kind[CHAIN_MAX_STEPS], opc[CHAIN_MAX_STEPS], mode[CHAIN_MAX_STEPS], uni[CHAIN_MAX_STEPS], sa[CHAIN_MAX_STEPS], sb[CHAIN_MAX_STEPS], out[CHAIN_MAX_STEPS];
  T scalar[CHAIN_MAX_STEPS];
  int red_op;
};

template <class T, int NIN> __device__ __forceinline__ T pickn(const T (&e)[NIN], int k) {
  T r = e[0];
#pragma unroll
  for (int j = 1; j < NIN; j++) r = (k == j) ? e[j] : r;
  return r;
}

// The operator of a step is wave-uniform, but a switch per POSITION costs a chain of scalar compares and branches each time (measured:
// 20 us for a 2-step chain over 2^22 positions whatever the grid — 16 wave-iterations per SIMD x 16 switches x ~100 cycles).  So the
// switch is taken once per step and the case applies its operator — a compile-time constant there — to all VEC positions.
#define GRB_CHAIN_BINOPS(X) X(B_FIRST) X(B_SECOND) X(B_PAIR) X(B_ANY) X(B_MIN) X(B_MAX) X(B_PLUS) X(B_MINUS) X(B_RMINUS) X(B_TIMES) X(B_DIV) X(B_RDIV) \
                            X(B_ISEQ) X(B_ISNE) X(B_ISGT) X(B_ISLT) X(B_ISGE) X(B_ISLE) X(B_LOR) X(B_LAND) X(B_LXOR)
#define GRB_CHAIN_UNOPS(X) X(U_IDENTITY) X(U_AINV) X(U_MINV) X(U_LNOT) X(U_ONE) X(U_ABS) X(U_BNOT)
template <class T, int N> __device__ __forceinline__ void chain_binop_vec(int op, const T (&x)[N], const T (&y)[N], T (&z)[N]) {
  switch (op) {
#define GRB_X(K) case K: { _Pragma("unroll") for (int h = 0; h < N; h++) z[h] = apply_binop<T, false, false>(K, x[h], y[h]); } break;
    GRB_CHAIN_BINOPS(GRB_X)
#undef GRB_X
    default: { _Pragma("unroll") for (int h = 0; h < N; h++) z[h] = x[h]; } break;
  }
}
template <class T, int N> __device__ __forceinline__ void chain_unop_vec(int op, const T (&x)[N], T (&z)[N]) {
  switch (op) {
#define GRB_X(K) case K: { _Pragma("unroll") for (int h = 0; h < N; h++) z[h] = apply_unop<T, false>(K, x[h]); } break;
    GRB_CHAIN_UNOPS(GRB_X)
#undef GRB_X
    default: { _Pragma("unroll") for (int h = 0; h < N; h++) z[h] = x[h]; } break;
  }
}
template <class R, int N> __device__ __forceinline__ R chain_reduce_vec(int op, R acc, const R (&v)[N], const bool (&ok)[N]) {
  switch (op) {
#define GRB_X(K) case K: { _Pragma("unroll") for (int h = 0; h < N; h++) if (ok[h]) acc = apply_binop<R, false, false>(K, acc, v[h]); } break;
    GRB_CHAIN_BINOPS(GRB_X)
#undef GRB_X
    default: break;
  }
  return acc;
}

// RED: 0 no reduction, 1 reduce the last result in T, 2 reduce FP32 values in FP64.  R = the reduction type.  NIN: stored operands read.
// VEC consecutive positions per lane and step of the grid-stride loop (4: one 16-byte load per operand and lane — two for 8-byte
// types — and one 4-byte load of presence bytes; 1: the fallback for buffers that are not 16-byte aligned).  All loads of a step are
// issued before anything is computed and every load is unconditional (a branch around a load makes the compiler wait for it on the
// spot: the first version of this kernel had ONE load in flight per lane and ran 48 us where the 13 us k_vec_ewise does the same work);
// results leave as vector stores.
template <class T, int VEC> struct alignas(sizeof(T) * VEC >= 16 ? 16 : sizeof(T) * VEC) ChainPack { T v[VEC]; };
template <int VEC> struct alignas(VEC) ChainBytes { uint8_t v[VEC]; };

// SPEC (round 4): the queue is an interpreter — per pack of VEC positions it walks the step descriptors through scalar compares and branches, ~100
// instructions per position, and the two passes of a PageRank iteration at 2^25 vertices ran at 2.5 and 4.9 TB/s bound by instruction issue, not by
// the HBM.  The two shapes of gap/prmark.py:21-26 are compiled as they stand (the same loads, stores, reduction tree and grid as the interpreter's):
//   1  reduce(+, abs(x - y)), x and y stored and full (`t -= r; t = abs(t); t.reduce_float()`), stored or not
//   2  z = x / y on the intersection of two stored operands (`w = t / d`)
template <class T, int RED, class R, int NIN, int VEC, int SPEC = 0>
__global__ __launch_bounds__(256) void k_vec_chain(const ChainK<T> a, const R rid, R* __restrict__ partial) {
  R racc = rid;
  const uint64_t stride = gridDim.x * 256ull * VEC;
  // the loads of a lane's NEXT pack are issued before this one is worked on (round 4: with one pack in flight per lane the pass that only reduces —
  // `t -= r; abs; reduce`, 268 MB — ran at 2.5 TB/s); a lane whose next pack is the vector's tail, or nothing, loads the current one again
  ChainPack<T, VEC> ve[NIN], vn[NIN]; ChainBytes<VEC> vp[NIN], vpn[NIN];
  auto load_pack = [&](uint64_t b, ChainPack<T, VEC>* pe, ChainBytes<VEC>* pp) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NIN; k++) pe[k] = *(const ChainPack<T, VEC>*)(a.ev[k] + b);
    if constexpr (SPEC != 1) {                                           // (shape 1: both operands full, nobody looks at presence bytes)
#pragma unroll
      for (int k = 0; k < NIN; k++) pp[k] = *(const ChainBytes<VEC>*)(a.ep[k] + b);
    }
  };
  {
    const uint64_t b0 = (blockIdx.x * 256ull + threadIdx.x) * VEC;
    if (b0 + VEC <= a.n) load_pack(b0, ve, vp);
  }
  for (uint64_t base = (blockIdx.x * 256ull + threadIdx.x) * VEC; base < a.n; base += stride) {
    const int nv = a.n - base >= (uint64_t)VEC ? VEC : (int)(a.n - base);       // < VEC only for the lane that holds the vector's tail
    T e[VEC][NIN]; bool p[VEC][NIN];
    if (nv == VEC) {
      const uint64_t nb = base + stride;
      load_pack(nb + VEC <= a.n ? nb : base, vn, vpn);
#pragma unroll
      for (int k = 0; k < NIN; k++) {
        const bool used = (a.used_mask >> k) & 1u, full = (a.full_mask >> k) & 1u;
#pragma unroll
        for (int h = 0; h < VEC; h++) { e[h][k] = ve[k].v[h]; if constexpr (SPEC == 1) p[h][k] = true; else p[h][k] = used && (full || vp[k].v[h] != 0); }
      }
#pragma unroll
      for (int k = 0; k < NIN; k++) { ve[k] = vn[k]; vp[k] = vpn[k]; }
    } else {
#pragma unroll
      for (int k = 0; k < NIN; k++) {
        const bool used = (a.used_mask >> k) & 1u, full = (a.full_mask >> k) & 1u;
#pragma unroll
        for (int h = 0; h < VEC; h++) { const uint64_t i = h < nv ? base + h : base; e[h][k] = a.ev[k][i]; p[h][k] = used && (full || a.ep[k][i] != 0); }
      }
    }
    T ov[CHAIN_MAX_OUT][VEC]; bool opv[CHAIN_MAX_OUT][VEC];
#pragma unroll
    for (int oo = 0; oo < CHAIN_MAX_OUT; oo++)
#pragma unroll
      for (int h = 0; h < VEC; h++) { ov[oo][h] = T(); opv[oo][h] = false; }
    T acc[VEC]; bool ap[VEC];
#pragma unroll
    for (int h = 0; h < VEC; h++) { acc[h] = T(); ap[h] = false; }
    if constexpr (SPEC == 1) {
#pragma unroll
      for (int h = 0; h < VEC; h++) { acc[h] = apply_unop<T, false>(U_ABS, apply_binop<T, false, false>(B_MINUS, e[h][0], e[h][1])); ap[h] = true; ov[0][h] = acc[h]; opv[0][h] = true; }
    } else if constexpr (SPEC == 2) {
#pragma unroll
      for (int h = 0; h < VEC; h++) { const bool both = p[h][0] && p[h][1]; acc[h] = both ? apply_binop<T, false, false>(B_DIV, e[h][0], e[h][1]) : T(); ap[h] = both; ov[0][h] = acc[h]; opv[0][h] = both; }
    } else {
#pragma unroll 1
    for (int s = 0; s < a.nsteps; s++) {                                 // (not unrolled: the step descriptors are scalar loads, the operator switches exist VEC times)
      // one operator evaluation per step: eWise f(x, y) on stored operands / the previous result; apply with a bound scalar is the
      // same call with the scalar in one seat; only a unary operator takes the other switch.  (FULL = false: the compact switches —
      // FIRST .. LXOR and IDENTITY .. BNOT; grb_lazy.cpp queues nothing else.)
      const int ia = a.sa[s], ib = a.sb[s], kd = a.kind[s], m = a.mode[s], opc = a.opc[s], o = a.out[s];
      const bool uni = a.uni[s] != 0; const T sc = a.scalar[s];
      T x[VEC], y[VEC], z[VEC]; bool xp[VEC], yp[VEC];
#pragma unroll
      for (int h = 0; h < VEC; h++) {
        x[h] = ia == CHAIN_PREV ? acc[h] : pickn<T, NIN>(e[h], ia); xp[h] = ia == CHAIN_PREV ? ap[h] : pickn<bool, NIN>(p[h], ia);
        y[h] = sc; yp[h] = true;
        if (kd == 0) { y[h] = ib == CHAIN_PREV ? acc[h] : pickn<T, NIN>(e[h], ib); yp[h] = ib == CHAIN_PREV ? ap[h] : pickn<bool, NIN>(p[h], ib); }
      }
      if (kd == 1 && m == 0) {
        chain_unop_vec<T, VEC>(opc, x, z);
#pragma unroll
        for (int h = 0; h < VEC; h++) { ap[h] = xp[h]; acc[h] = xp[h] ? z[h] : T(); }
      } else {
        if (kd == 1 && m == 1) {                                            // z = f(s, x): the scalar takes the first seat
#pragma unroll
          for (int h = 0; h < VEC; h++) { const T tmp = x[h]; x[h] = y[h]; y[h] = tmp; const bool tp = xp[h]; xp[h] = yp[h]; yp[h] = tp; }
        }
        chain_binop_vec<T, VEC>(opc, x, y, z);
#pragma unroll
        for (int h = 0; h < VEC; h++) {
          const bool both = xp[h] && yp[h];
          const bool zp = (kd == 0 && uni) ? (xp[h] || yp[h]) : both;
          const T v = both ? z[h] : (xp[h] ? x[h] : y[h]);
          ap[h] = zp; acc[h] = zp ? v : T();
        }
      }
#pragma unroll
      for (int oo = 0; oo < CHAIN_MAX_OUT; oo++) if (o == oo) {
#pragma unroll
        for (int h = 0; h < VEC; h++) { ov[oo][h] = acc[h]; opv[oo][h] = ap[h]; }
      }
    }
    }
#pragma unroll
    for (int oo = 0; oo < CHAIN_MAX_OUT; oo++) {
      if (a.ov[oo]) {
        if (nv == VEC) {
          ChainPack<T, VEC> w; ChainBytes<VEC> wp;
#pragma unroll
          for (int h = 0; h < VEC; h++) { w.v[h] = ov[oo][h]; wp.v[h] = opv[oo][h] ? 1 : 0; }
          *(ChainPack<T, VEC>*)(a.ov[oo] + base) = w;
          if (a.op[oo]) *(ChainBytes<VEC>*)(a.op[oo] + base) = wp;
        } else {
#pragma unroll
          for (int h = 0; h < VEC; h++) if (h < nv) { a.ov[oo][base + h] = ov[oo][h]; if (a.op[oo]) a.op[oo][base + h] = opv[oo][h] ? 1 : 0; }
        }
      }
    }
    if constexpr (RED != 0) {
      R rv[VEC]; bool rok[VEC];
#pragma unroll
      for (int h = 0; h < VEC; h++) { rv[h] = (R)acc[h]; rok[h] = h < nv && ap[h]; }
      racc = chain_reduce_vec<R, VEC>(a.red_op, racc, rv, rok);
    }
  }
  if constexpr (RED != 0) {
    // lanes -> wave -> workgroup -> one partial per workgroup; a one-workgroup k_chain_final folds the partials in index order (a fixed
    // tree: the result does not depend on the order the workgroups ran in).  (Folding them in the last workgroup to finish, behind
    // an agent-scope fence and a ticket, made every workgroup write back its XCD's L2 in the middle of the stores: 102 us.  Round 5: the
    // workgroups storing their partials as tagged words straight into page-locked host memory, on which the caller spins — what the BFS's
    // `reduce_bool` does with 256 words, grb_container.cpp — made a PageRank iteration at R-MAT-22 0.171 -> 0.186 ms: 2 x 1024 small
    // writes across PCIe at the end of a 13 us kernel cost more than the copy + synchronise they replace.)
    __shared__ R sh[4];
    racc = wave_reduce_op<R, false>(a.red_op, racc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = racc;
    __syncthreads();
    if (threadIdx.x == 0) {
      R r = sh[0];
      for (int w = 1; w < 4; w++) r = apply_binop<R, false, false>(a.red_op, r, sh[w]);
      partial[blockIdx.x] = r;
    }
  }
}
template <class R> __global__ __launch_bounds__(256) void k_chain_final(const R* __restrict__ partial, uint32_t g, int op, R rid, R* __restrict__ result) {
  __shared__ R sh[4];
  R r = rid;
  for (uint32_t b = threadIdx.x; b < g; b += 256) r = apply_binop<R, false, false>(op, r, partial[b]);
  r = wave_reduce_op<R, false>(op, r);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = r;
  __syncthreads();
  if (threadIdx.x == 0) { R t = sh[0]; for (int w = 1; w < 4; w++) t = apply_binop<R, false, false>(op, t, sh[w]); *result = t; }
}

template <class T, int RED, class R, int NIN> static void launch_chain(const ChainLaunch& L, void* red_result) {
  ChainK<T> a{};
  a.n = L.n; a.nsteps = L.nsteps; a.next = L.next;
  bool aligned = true;
  for (int k = 0; k < CHAIN_MAX_IN; k++) {
    const bool used = k < L.next;
    a.ev[k] = (const T*)(used ? L.ev[k] : L.ev[0]);
    a.ep[k] = (used && L.ep[k]) ? L.ep[k] : (const uint8_t*)L.ev[0];        // not looked at for a full or unused operand: any readable n bytes
    if (used) a.used_mask |= 1u << k;
    if (used && !L.ep[k]) a.full_mask |= 1u << k;
    if (((uintptr_t)a.ev[k] & 15) || ((uintptr_t)a.ep[k] & 3)) aligned = false;
  }
  for (int k = 0; k < CHAIN_MAX_OUT; k++) {
    a.ov[k] = k < L.nout ? (T*)L.ov[k] : nullptr; a.op[k] = k < L.nout ? L.op[k] : nullptr;
    if (((uintptr_t)a.ov[k] & 15) || ((uintptr_t)a.op[k] & 3)) aligned = false;
  }
  for (int s = 0; s < CHAIN_MAX_STEPS; s++) {
    const ChainStepDesc& st = L.st[s];
    a.kind[s] = st.kind; a.opc[s] = st.op; a.mode[s] = st.mode; a.uni[s] = st.is_union; a.sa[s] = st.src[0]; a.sb[s] = st.src[1]; a.out[s] = s < L.nsteps ? st.out : -1;
    memcpy(&a.scalar[s], st.scalar, sizeof(T));
  }
  a.red_op = L.red.op;
  // one round of workgroups: 4 per CU are resident whatever the variant's register count (with 2048 workgroups of a 69-register
  // variant — 7 per CU — the eighth waited for a slot and the kernel ran a second, nearly empty round: 27 us for 50 MB)
  // (as many workgroups as are resident at once for THIS variant: the occupancy API, asked once per instantiation)
  static std::atomic<int> occ4{0}, occ1{0};            // (asked once per instantiation; concurrent first callers compute the same numbers)
  if (!occ4) { int b = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_vec_chain<T, RED, R, NIN, 4>, 256, 0) != hipSuccess || b < 1) { (void)hipGetLastError(); b = 4; } occ4.store(b > 64 ? 64 : b); }
  if (!occ1) { int b = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_vec_chain<T, RED, R, NIN, 1>, 256, 0) != hipSuccess || b < 1) { (void)hipGetLastError(); b = 4; } occ1.store(b > 64 ? 64 : b); }
  static const int env_vec = getenv("GRB_MI355X_CHAIN_VEC") ? atoi(getenv("GRB_MI355X_CHAIN_VEC")) : 4;            // measurement hooks
  static const int env_bpc = getenv("GRB_MI355X_CHAIN_BPC") ? atoi(getenv("GRB_MI355X_CHAIN_BPC")) : 0;
  if (env_vec == 1) aligned = false;
  const int use4 = env_bpc > 0 ? env_bpc : occ4.load(), use1 = env_bpc > 0 ? env_bpc : occ1.load();
  const uint64_t gmax = (uint64_t)(device_cus() > 0 ? device_cus() : 256) * (uint64_t)(aligned ? use4 : use1);
  uint64_t g = (L.n + 256ull * 4 - 1) / (256ull * 4); if (g < 1) g = 1; if (g > gmax) g = gmax; if (g > 16384) g = 16384;
  R rid{}; R* result = nullptr; R* partial = nullptr;
  if constexpr (RED != 0) {
    static thread_local DevBuf work;                                    // [result 16 B | partials]
    if (!work.p) work.alloc(16 + 16384 * sizeof(double));
    memcpy(&rid, L.red.identity, sizeof(R));
    result = (R*)work.p; partial = (R*)((char*)work.p + 16);
  }
  // the compiled shapes (see SPEC at the kernel): same grid as the interpreter's
  int spec = 0;
  if constexpr (std::is_floating_point<T>::value && NIN == 2) {
    static const bool no_spec = getenv("GRB_MI355X_CHAIN_NO_SPEC") && atoi(getenv("GRB_MI355X_CHAIN_NO_SPEC")) != 0;
    const ChainStepDesc& s0 = L.st[0]; const ChainStepDesc& s1 = L.st[1];
    if (!no_spec && RED != 0 && L.nsteps == 2 && L.next == 2 && s0.kind == 0 && s0.op == B_MINUS && s0.is_union && s0.src[0] == 0 && s0.src[1] == 1 && s1.kind == 1 && s1.mode == 0 &&
        s1.op == U_ABS && s1.src[0] == CHAIN_PREV && L.red.op == B_PLUS && a.full_mask == 3u && s0.out == -1 && (L.nout == 0 || (L.nout == 1 && s1.out == 0 && L.op[0] == nullptr))) spec = 1;
    if (!no_spec && RED == 0 && L.nsteps == 1 && L.next == 2 && s0.kind == 0 && s0.op == B_DIV && !s0.is_union && s0.src[0] == 0 && s0.src[1] == 1 && L.nout == 1 && s0.out == 0 && L.op[0] != nullptr) spec = 2;
  }
  bool launched = false;
  // a floating-point chain that was seen before runs through the kernel hipRTC compiled for exactly its steps (grb_chain_jit.cpp); the two shapes
  // below stay compiled ahead of time (no first-use compilation inside somebody's PageRank loop) unless GRB_MI355X_CHAIN_JIT=2 asks for the comparison
  if constexpr (std::is_floating_point<T>::value || (std::is_integral<T>::value && (sizeof(T) == 4 || sizeof(T) == 8))) {
    constexpr int tcode = std::is_same<T, float>::value ? T_FP32 : std::is_same<T, double>::value ? T_FP64 : std::is_same<T, int32_t>::value ? T_INT32 : std::is_same<T, uint32_t>::value ? T_UINT32
                        : std::is_same<T, int64_t>::value ? T_INT64 : std::is_same<T, uint64_t>::value ? T_UINT64 : -1;
    if (aligned && tcode >= 0) launched = chain_jit_launch(L, tcode, RED, RED != 0 ? (const void*)&rid : nullptr, (void*)partial, (unsigned)g, spec != 0);
  }
  if (launched) {}
  else if constexpr (std::is_floating_point<T>::value && NIN == 2 && RED != 0) {
    if (spec == 1) { if (aligned) hipLaunchKernelGGL((k_vec_chain<T, RED, R, NIN, 4, 1>), dim3((unsigned)g), dim3(256), 0, stream(), a, rid, partial);
                     else hipLaunchKernelGGL((k_vec_chain<T, RED, R, NIN, 1, 1>), dim3((unsigned)g), dim3(256), 0, stream(), a, rid, partial); launched = true; }
  }
  if constexpr (std::is_floating_point<T>::value && NIN == 2 && RED == 0) {
    if (!launched && spec == 2) { if (aligned) hipLaunchKernelGGL((k_vec_chain<T, RED, R, NIN, 4, 2>), dim3((unsigned)g), dim3(256), 0, stream(), a, rid, partial);
                     else hipLaunchKernelGGL((k_vec_chain<T, RED, R, NIN, 1, 2>), dim3((unsigned)g), dim3(256), 0, stream(), a, rid, partial); launched = true; }
  }
  if (!launched) {
    if (aligned) hipLaunchKernelGGL((k_vec_chain<T, RED, R, NIN, 4>), dim3((unsigned)g), dim3(256), 0, stream(), a, rid, partial);
    else hipLaunchKernelGGL((k_vec_chain<T, RED, R, NIN, 1>), dim3((unsigned)g), dim3(256), 0, stream(), a, rid, partial);
  }
  if constexpr (RED != 0) {
    // the per-workgroup partials (<= 16 384 x 8 B) travel to page-locked memory and the host folds them in index order — a fixed
    // tree like the one-workgroup k_chain_final it replaces in the common case (one kernel and one launch gap less per PageRank iteration)
    static thread_local R* pin = nullptr;
    if (!pin) GRB_HIP(hipHostMalloc((void**)&pin, 16384 * sizeof(double), hipHostMallocDefault));
    if (g <= 4096) {
      GRB_HIP(hipMemcpyAsync(pin, partial, (size_t)g * sizeof(R), hipMemcpyDeviceToHost, stream()));
      GRB_HIP(hipStreamSynchronize(stream()));
      R r = rid;
      for (uint64_t b = 0; b < g; b++) r = apply_binop<R, false, false>(L.red.op, r, pin[b]);
      memcpy(red_result, &r, sizeof(R));
    } else {
      hipLaunchKernelGGL((k_chain_final<R>), dim3(1), dim3(256), 0, stream(), (const R*)partial, (uint32_t)g, L.red.op, rid, result);
      GRB_HIP(hipMemcpyAsync(pin, result, sizeof(R), hipMemcpyDeviceToHost, stream()));
      GRB_HIP(hipStreamSynchronize(stream()));
      memcpy(red_result, pin, sizeof(R));
    }
  }
  GRB_HIP(hipGetLastError());
}

// one translation unit per value type (Makefile: *_inst.hip); the 1- and 2-byte types have no chain kernel (their vectors — BFS levels,
// frontiers — are not what the deferred element-wise queue is for) and run every operation the blocking way
template <class T> void vec_chain_launch_t(const ChainLaunch& L, void* red_result) {
  if constexpr (sizeof(T) >= 4) {
    auto go = [&](auto nin) {
      constexpr int NIN = decltype(nin)::value;
      // (operators that need the math library — pow, trigonometry, ... — are never queued: grb_lazy.cpp runs them the blocking way)
      if (!L.red.on) launch_chain<T, 0, T, NIN>(L, nullptr);
      else if (L.red.widen) { if constexpr (std::is_same<T, float>::value) launch_chain<T, 2, double, NIN>(L, red_result); }
      else launch_chain<T, 1, T, NIN>(L, red_result);
    };
    if (L.next <= 2) go(std::integral_constant<int, 2>()); else go(std::integral_constant<int, 4>());
  }
}
using std::int8_t; using std::uint8_t; using std::int16_t; using std::uint16_t; using std::int32_t; using std::uint32_t; using std::int64_t; using std::uint64_t;
template void vec_chain_launch_t<GRB_INST_TYPE>(const ChainLaunch&, void*);

}  // namespace grb
