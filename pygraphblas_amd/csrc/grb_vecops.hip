// grb_vecops.hip — O(n) streaming kernels over the bitmap vector layout (val T[n] | present u8[n]).
// All of them are HBM-bound: one coalesced pass, grid-stride, 256-thread blocks capped at 2048
// blocks (guide §6 G11).  They implement typecasting, mask -> "allow" bytes, the
// C<M,replace> = accum(C,T) epilogue of GraphBLAS, monoid
// reductions and the element-wise companions of the hot path.
#include "grb_api.hpp"
#include "grb_device.hpp"

namespace grb {

static inline int grid_for(uint64_t n, int per_thread = 1) {
  uint64_t b = (n + 256ull * per_thread - 1) / (256ull * per_thread);
  if (b < 1) b = 1; if (b > 2048) b = 2048; return (int)b;
}

// ---- typecast ---------------------------------------------------------------------------------------
template <class D, class S> __global__ void k_cast(D* __restrict__ dst, const S* __restrict__ src, uint64_t n) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) dst[i] = cast_to<D, S>(src[i]);
}
void vec_cast_values(int dst_code, void* dst, int src_code, const void* src, uint64_t n) {
  if (!n) return;
  dispatch_type(src_code, [&]<class S>() {
    dispatch_type(dst_code, [&]<class D>() {
      hipLaunchKernelGGL((k_cast<D, S>), dim3(grid_for(n)), dim3(256), 0, stream(), (D*)dst, (const S*)src, n);
    });
  });
}

// values of a bitmap vector cast into another type with the absent positions filled (one pass): what lets a product whose
// result pattern does not matter treat an operand with holes as a full one (grb_mxv.cpp)
template <class D, class S> __global__ void k_cast_fill(D* __restrict__ dst, const S* __restrict__ src, const uint8_t* __restrict__ pres, uint64_t n, D fill) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) dst[i] = pres[i] ? cast_to<D, S>(src[i]) : fill;
}
void vec_cast_fill_values(int dst_code, void* dst, int src_code, const void* src, const uint8_t* pres, uint64_t n, const void* fill) {
  if (!n) return;
  dispatch_type(src_code, [&]<class S>() {
    dispatch_type(dst_code, [&]<class D>() {
      D f; memcpy(&f, fill, sizeof(D));
      hipLaunchKernelGGL((k_cast_fill<D, S>), dim3(grid_for(n)), dim3(256), 0, stream(), (D*)dst, (const S*)src, pres, n, f);
    });
  });
}

// ---- range of the stored values (MIN_PLUS / MAX_PLUS products over an operand with holes: grb_mxv.cpp "big holes") ---------------
// min, max and the number of non-finite values of the present entries, in the type itself; one kernel, per-workgroup results
// combined with ordered-integer atomics (floats: the usual sign-flip encoding), one read-back.
template <class T> struct RangeEnc;
template <> struct RangeEnc<int32_t> { typedef int32_t E; static __host__ __device__ E enc(int32_t v) { return v; } static __host__ __device__ int32_t dec(E e) { return e; } };
template <> struct RangeEnc<int64_t> { typedef long long E; static __host__ __device__ E enc(int64_t v) { return (long long)v; } static __host__ __device__ int64_t dec(E e) { return (int64_t)e; } };
template <> struct RangeEnc<float> { typedef int32_t E;
  static __host__ __device__ E enc(float v) { int32_t b; memcpy(&b, &v, 4); return b >= 0 ? b : (int32_t)(b ^ 0x7FFFFFFF); }
  static __host__ __device__ float dec(E e) { int32_t b = e >= 0 ? e : (int32_t)(e ^ 0x7FFFFFFF); float v; memcpy(&v, &b, 4); return v; } };
template <> struct RangeEnc<double> { typedef long long E;
  static __host__ __device__ E enc(double v) { long long b; memcpy(&b, &v, 8); return b >= 0 ? b : (long long)(b ^ 0x7FFFFFFFFFFFFFFFll); }
  static __host__ __device__ double dec(E e) { long long b = e >= 0 ? e : (long long)(e ^ 0x7FFFFFFFFFFFFFFFll); double v; memcpy(&v, &b, 8); return v; } };
template <class T> __global__ void k_value_range(uint64_t n, const T* __restrict__ val, const uint8_t* __restrict__ pres, long long* __restrict__ out /* [min, max, nonfinite, count] */) {
  typedef typename RangeEnc<T>::E E;
  E mn = std::numeric_limits<E>::max(), mx = std::numeric_limits<E>::min(); unsigned long long bad = 0, cnt = 0;
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) if (!pres || pres[i]) {
    const T v = val[i]; cnt++;
    if constexpr (std::is_floating_point<T>::value) { if (!(v - v == T(0))) { bad++; continue; } }      // NaN or infinity
    const E e = RangeEnc<T>::enc(v); mn = e < mn ? e : mn; mx = e > mx ? e : mx;
  }
  __shared__ long long smn[4], smx[4]; __shared__ unsigned long long sbad[4], scnt[4];
  long long lmn = (long long)mn, lmx = (long long)mx;
  for (int o = 32; o; o >>= 1) { const long long a = __shfl_xor(lmn, o, 64), b = __shfl_xor(lmx, o, 64); lmn = a < lmn ? a : lmn; lmx = b > lmx ? b : lmx; }
  bad = wave_reduce_add_u64(bad); cnt = wave_reduce_add_u64(cnt);
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = lmn; smx[threadIdx.x >> 6] = lmx; sbad[threadIdx.x >> 6] = bad; scnt[threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) { lmn = smn[w] < lmn ? smn[w] : lmn; lmx = smx[w] > lmx ? smx[w] : lmx; bad += sbad[w]; cnt += scnt[w]; }
    if (cnt) { atomicMin(&out[0], lmn); atomicMax(&out[1], lmx); atomicAdd((unsigned long long*)&out[2], bad); atomicAdd((unsigned long long*)&out[3], cnt); }
  }
}
bool value_range(int code, uint64_t n, const void* val, const uint8_t* pres, void* vmin, void* vmax, uint64_t* nonfinite, uint64_t* count) {
  if (code != T_INT32 && code != T_INT64 && code != T_FP32 && code != T_FP64) return false;
  *nonfinite = 0; *count = 0;
  if (!n) return true;
  DevBuf acc(32);
  const long long init[4] = {std::numeric_limits<long long>::max(), std::numeric_limits<long long>::min(), 0, 0};
  long long* pin = (long long*)pinned_scratch();
  memcpy(pin, init, 32);
  GRB_HIP(hipMemcpyAsync(acc.p, pin, 32, hipMemcpyHostToDevice, stream()));
  dispatch_type(code, [&]<class T>() {
    if constexpr (std::is_same<T, int32_t>::value || std::is_same<T, int64_t>::value || std::is_same<T, float>::value || std::is_same<T, double>::value)
      hipLaunchKernelGGL((k_value_range<T>), dim3(std::min(grid_for(n, 8), 512)), dim3(256), 0, stream(), n, (const T*)val, pres, acc.as<long long>());
  });
  GRB_HIP(hipMemcpyAsync(pin + 8, acc.p, 32, hipMemcpyDeviceToHost, stream()));      // (another part of the pinned block than the source of the copy above)
  GRB_HIP(hipStreamSynchronize(stream()));
  pin += 8;
  *nonfinite = (uint64_t)pin[2]; *count = (uint64_t)pin[3];
  if (*count > *nonfinite) {
    dispatch_type(code, [&]<class T>() {
      if constexpr (std::is_same<T, int32_t>::value || std::is_same<T, int64_t>::value || std::is_same<T, float>::value || std::is_same<T, double>::value) {
        typedef typename RangeEnc<T>::E E;
        const T a = RangeEnc<T>::dec((E)pin[0]), b = RangeEnc<T>::dec((E)pin[1]); memcpy(vmin, &a, sizeof(T)); memcpy(vmax, &b, sizeof(T));
      }
    });
  }
  return true;
}
// entries whose value lies on the far side of `thresh` were made of fill values only: they are not entries
template <class T> __global__ void k_big_to_absent(uint64_t n, const T* __restrict__ val, uint8_t* __restrict__ pres, T thresh, bool keep_below) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) if (pres[i]) { const T v = val[i]; if (!(keep_below ? v < thresh : v > thresh)) pres[i] = 0; }
}
void big_to_absent(int code, uint64_t n, const void* val, uint8_t* pres, const void* thresh, bool keep_below) {
  if (!n) return;
  dispatch_type(code, [&]<class T>() {
    if constexpr (std::is_same<T, int32_t>::value || std::is_same<T, int64_t>::value || std::is_same<T, float>::value || std::is_same<T, double>::value) {
      T th; memcpy(&th, thresh, sizeof(T));
      hipLaunchKernelGGL((k_big_to_absent<T>), dim3(grid_for(n)), dim3(256), 0, stream(), n, (const T*)val, pres, th, keep_below);
    }
  });
}

// ---- mask -> allow bytes ------------------------------------------------------------------------------
template <class M> __global__ void k_allow(uint64_t n, const M* __restrict__ mval, const uint8_t* __restrict__ mpres,
                                           bool structural, bool complement, uint8_t* __restrict__ allow) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
    bool m = mpres[i] != 0;
    if (m && !structural) m = (bool)cast_to<bool8, M>(mval[i]);
    allow[i] = (uint8_t)(m != complement);
  }
}
void build_allow(uint64_t n, int mcode, const void* mval, const uint8_t* mpres, bool structural, bool complement, uint8_t* allow) {
  if (!n) return;
  dispatch_type(mcode, [&]<class M>() {
    hipLaunchKernelGGL((k_allow<M>), dim3(grid_for(n)), dim3(256), 0, stream(), n, (const M*)mval, mpres, structural, complement, allow);
  });
}

static inline bool unop_needs_math(int op) { return op >= U_SQRT && op <= U_ISFINITE; }

// ---- count present ------------------------------------------------------------------------------------
// One atomic per workgroup and at most 512 workgroups: same-address atomics complete one after the other (~10-80 ns each on
// this part), so a counter bumped by every wave of a 4096-block grid cost 40-80 us for a 4 MB bitmap.
__device__ __forceinline__ void block_add_u64(unsigned long long c, unsigned long long* out) {
  __shared__ unsigned long long sh[4];
  c = wave_reduce_add_u64(c);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) { const unsigned long long t = sh[0] + sh[1] + sh[2] + sh[3]; if (t) atomicAdd(out, t); }
}
static inline int grid_capped(uint64_t n, int per_thread, int cap = 512) { const int g = grid_for(n, per_thread); return g < cap ? g : cap; }

// (round 4: 128 workgroups, four 16-byte loads in flight per lane — 512 workgroups of two loads each ended in 512 same-address atomics: 10 us for a 4 MB
//  bitmap, and the shortest-path loop's `iseq` counts two of them per sweep)
__global__ __launch_bounds__(256) void k_count(const uint8_t* __restrict__ pres, uint64_t n, const ScalarPub pub) {
  unsigned long long* const out = pub.slot;
  unsigned long long c = 0;
  // 16 bytes per lane per load
  const uint64_t n16 = n / 16, T = gridDim.x * 256ull;
  const uint4* p4 = (const uint4*)pres;
  auto ones = [](const uint4& v) { return (unsigned)(__popc(v.x & 0x01010101u) + __popc(v.y & 0x01010101u) + __popc(v.z & 0x01010101u) + __popc(v.w & 0x01010101u)); };
  uint64_t i = blockIdx.x * 256ull + threadIdx.x;
  for (; i + 3 * T < n16; i += 4 * T) { const uint4 a = p4[i], b = p4[i + T], d = p4[i + 2 * T], e = p4[i + 3 * T]; c += ones(a) + ones(b) + ones(d) + ones(e); }
  for (; i < n16; i += T) c += ones(p4[i]);
  for (uint64_t j = n16 * 16 + blockIdx.x * 256ull + threadIdx.x; j < n; j += T) c += pres[j] != 0;
  block_add_u64(c, out);
  scalar_publish(pub);
}
// two device-to-device copies in one launch (GrB_Vector_dup: values + presence bytes — `w = v.dup()` opens every sweep of the shortest-path loop;
// two hipMemcpyAsync were two launches of the runtime's own copy kernel)
__global__ __launch_bounds__(256) void k_copy2(uint4* __restrict__ d0, const uint4* __restrict__ s0, uint64_t n0 /* bytes */, uint4* __restrict__ d1, const uint4* __restrict__ s1, uint64_t n1) {
  const uint64_t q0 = n0 / 16, q1 = n1 / 16, T = gridDim.x * 256ull;
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < q0 + q1; i += T) { if (i < q0) d0[i] = s0[i]; else d1[i - q0] = s1[i - q0]; }
  if (blockIdx.x == 0) {
    for (uint64_t b = q0 * 16 + threadIdx.x; b < n0; b += 256) ((uint8_t*)d0)[b] = ((const uint8_t*)s0)[b];
    for (uint64_t b = q1 * 16 + threadIdx.x; b < n1; b += 256) ((uint8_t*)d1)[b] = ((const uint8_t*)s1)[b];
  }
}
void dev_copy2(void* d0, const void* s0, uint64_t n0, void* d1, const void* s1, uint64_t n1) {
  if (!(n0 + n1)) return;
  if ((((uintptr_t)d0 | (uintptr_t)s0 | (uintptr_t)d1 | (uintptr_t)s1) & 15) != 0) {      // (never with the pool's blocks)
    if (n0) GRB_HIP(hipMemcpyAsync(d0, s0, n0, hipMemcpyDeviceToDevice, stream()));
    if (n1) GRB_HIP(hipMemcpyAsync(d1, s1, n1, hipMemcpyDeviceToDevice, stream()));
    return;
  }
  const uint64_t q = (n0 + n1) / 16 + 1;
  uint64_t g = (q + 256ull * 4 - 1) / (256ull * 4); const uint64_t gmax = (uint64_t)(device_cus() > 0 ? device_cus() : 256) * 8ull; if (g > gmax) g = gmax; if (g < 1) g = 1;
  hipLaunchKernelGGL(k_copy2, dim3((unsigned)g), dim3(256), 0, stream(), (uint4*)d0, (const uint4*)s0, n0, (uint4*)d1, (const uint4*)s1, n1);
}
uint64_t count_present(const uint8_t* pres, uint64_t n) {
  if (!n) return 0;
  ScalarSlot slot; slot.zero();
  hipLaunchKernelGGL(k_count, dim3(grid_capped(n, 16, 128)), dim3(256), 0, stream(), pres, n, slot.pub());
  return slot.read_u64();
}

// ---- "same pattern and equal values" of two bitmap vectors in ONE pass (GrBX_Vector_iseq, round 6) ------------------------------------------------
// What the reference's `Vector.iseq` composes from nvals, nvals, eWiseMult(EQ) into a BOOL vector, nvals and a LAND reduction (pygraphblas/vector.py:188-235) —
// five kernels, two temporaries and four host round trips in the shortest-path loop's late sweeps — counted as mismatching positions by one kernel:
// the presence bytes of both, the values only where both are present (NaN differs from NaN, -0.0 equals 0.0: the EQ operator).
template <class T> __global__ __launch_bounds__(256) void k_vec_iseq(uint64_t n, const T* __restrict__ uval, const uint8_t* __restrict__ upres, const T* __restrict__ vval, const uint8_t* __restrict__ vpres,
                                                                     const bool aligned, const ScalarPub pub) {
  unsigned long long c = 0;
  // four presence bytes of each vector per lane (consecutive lanes, consecutive words: coalesced), the values of the positions both hold — four independent
  // pairs of loads in flight.  (A first version gave a lane eight consecutive positions: its value loads touched 64 lines per instruction, 25 us for 2 x 37 MB.)
  const uint64_t n4 = aligned ? n / 4 : 0, TT = gridDim.x * 256ull;
  for (uint64_t g = blockIdx.x * 256ull + threadIdx.x; g < n4; g += TT) {
    const uint32_t pu = ((const uint32_t*)upres)[g], pv = ((const uint32_t*)vpres)[g];
    bool a[4], b[4]; T x[4], y[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { a[k] = ((pu >> (8 * k)) & 0xFFu) != 0; b[k] = ((pv >> (8 * k)) & 0xFFu) != 0; }
#pragma unroll
    for (int k = 0; k < 4; k++) if (a[k] && b[k]) { x[k] = uval[g * 4 + k]; y[k] = vval[g * 4 + k]; }
#pragma unroll
    for (int k = 0; k < 4; k++) c += (a[k] != b[k]) || (a[k] && !val_eq(x[k], y[k]));
  }
  for (uint64_t i = n4 * 4 + blockIdx.x * 256ull + threadIdx.x; i < n; i += TT) { const bool a = upres[i] != 0, b = vpres[i] != 0; c += (a != b) || (a && !val_eq(uval[i], vval[i])); }
  block_add_u64(c, pub.slot);
  scalar_publish(pub);
}
uint64_t vec_iseq_mismatches(int code, uint64_t n, const void* uval, const uint8_t* upres, const void* vval, const uint8_t* vpres) {
  if (!n) return 0;
  ScalarSlot slot;
  dispatch_type(code, [&]<class T>() {
    hipLaunchKernelGGL((k_vec_iseq<T>), dim3(grid_capped(n, 16, 2048)), dim3(256), 0, stream(), n, (const T*)uval, upres, (const T*)vval, vpres,
                       (((uintptr_t)upres | (uintptr_t)vpres) & 3u) == 0, slot.pub());
  });
  return slot.read_u64();
}

// ---- a few entries into a zeroed bitmap (upload of a sparse host vector: the `q[start] = True` of a BFS, an empty output) ----------
__global__ void k_scatter_entries(uint32_t k, const uint32_t* __restrict__ idx, const uint8_t* __restrict__ vals, uint32_t ts, uint8_t* __restrict__ val, uint8_t* __restrict__ pres) {
  for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < k; e += gridDim.x * 256u) {
    const uint64_t i = idx[e];
    for (uint32_t b = 0; b < ts; b++) val[i * ts + b] = vals[(uint64_t)e * ts + b];
    pres[i] = 1;
  }
}
// the same for up to 16 entries handed over BY VALUE (kernel arguments): no staging buffers, no host synchronisation — the one-entry
// frontier `q[start] = True` of a BFS loop reaches HBM this way
struct SmallEntries { uint32_t idx[16]; uint8_t x[16][8]; };
// the entries AND the zero fill of both arrays in one launch (round 4: the upload of a vector of <= 16 entries — the empty level vector and the one-entry
// frontier in front of a BFS loop — was two hipMemsetAsync + the scatter: three launches of ~4 us each for microseconds of work).  A thread owns 16
// positions: it writes their presence bytes and their values, zeros unless one of the entries lies among them.
__global__ __launch_bounds__(256) void k_init_small(uint32_t k, const SmallEntries e, uint32_t ts, uint8_t* __restrict__ val, uint8_t* __restrict__ pres, uint64_t n) {
  const uint64_t nchunks = (n + 15) / 16;
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (uint64_t c = blockIdx.x * 256ull + threadIdx.x; c < nchunks; c += gridDim.x * 256ull) {
    const uint64_t p0 = c * 16;
    bool any = false;
    for (uint32_t j = 0; j < k; j++) any = any || ((uint64_t)e.idx[j] >= p0 && (uint64_t)e.idx[j] < p0 + 16);
    if (!any && p0 + 16 <= n) {
      *(uint4*)(pres + p0) = z;
      for (uint32_t q = 0; q < ts; q++) *(uint4*)(val + p0 * ts + 16ull * q) = z;
    } else {                                                            // the few chunks that hold an entry, and the vector's tail: byte by byte
      for (uint64_t p = p0; p < p0 + 16 && p < n; p++) {
        int hit = -1;
        for (uint32_t j = 0; j < k; j++) if ((uint64_t)e.idx[j] == p) hit = (int)j;      // (the last one wins)
        pres[p] = hit >= 0 ? 1 : 0;
        for (uint32_t b = 0; b < ts; b++) val[p * ts + b] = hit >= 0 ? e.x[hit][b] : 0;
      }
    }
  }
}
void init_entries_small(uint32_t k, const uint64_t* idx_host, const uint8_t* vals_host, size_t ts, void* val, uint8_t* pres, uint64_t n) {
  SmallEntries e; memset(&e, 0, sizeof e);
  for (uint32_t i = 0; i < k; i++) { e.idx[i] = (uint32_t)idx_host[i]; memcpy(e.x[i], vals_host + (size_t)i * ts, ts); }
  uint64_t g = ((n + 15) / 16 + 255) / 256; if (g < 1) g = 1; if (g > 4096) g = 4096;
  hipLaunchKernelGGL(k_init_small, dim3((unsigned)g), dim3(256), 0, stream(), k, e, (uint32_t)ts, (uint8_t*)val, pres, n);
}
// a list of up to 64 indices handed over by value -> an index array in HBM (the frontier list of a push step whose operand's entries are known on the host)
struct SmallList { uint32_t idx[64]; };
__global__ void k_write_small_list(uint32_t k, const SmallList l, uint32_t* __restrict__ out) { if (threadIdx.x < k) out[threadIdx.x] = l.idx[threadIdx.x]; }
void write_small_list(uint32_t k, const uint32_t* idx_host, uint32_t* out_dev) {
  SmallList l; memset(&l, 0, sizeof l);
  for (uint32_t i = 0; i < k && i < 64; i++) l.idx[i] = idx_host[i];
  hipLaunchKernelGGL(k_write_small_list, dim3(1), dim3(64), 0, stream(), k, l, out_dev);
}
void scatter_entries(uint32_t k, const uint32_t* idx_dev, const void* vals_dev, size_t ts, void* val, uint8_t* pres) {
  if (!k) return;
  hipLaunchKernelGGL(k_scatter_entries, dim3(grid_for(k)), dim3(256), 0, stream(), k, idx_dev, (const uint8_t*)vals_dev, (uint32_t)ts, (uint8_t*)val, pres);
}

// ---- sum of the row lengths of the present entries (how many edges a push from this frontier would walk) ----------------
// (round 4: 16 presence bytes per lane and load, two loads in flight, 256 workgroups — four bytes per load, eight dependent rounds per lane and 2 x 512
//  same-address atomics took 21 us for 4 M positions: the level-2 direction choice of the BFS loop)
__global__ __launch_bounds__(256) void k_frontier_edges(const uint8_t* __restrict__ pres, const uint32_t* __restrict__ rowptr, uint64_t n, const ScalarPub pub, const bool with_count) {
  unsigned long long* const out = pub.slot; unsigned long long* const out_count = with_count ? pub.slot + 1 : nullptr;
  unsigned long long c = 0, np = 0;
  // the row pointers are only read for present entries
  const uint64_t n16 = n / 16, T = gridDim.x * 256ull;
  const uint4* p16 = (const uint4*)pres;
  auto take = [&](uint64_t i, const uint4& q) __attribute__((always_inline)) {
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; k++) if (w[k]) {
#pragma unroll
      for (int j = 0; j < 4; j++) if ((w[k] >> (8 * j)) & 0xFFu) { const uint64_t r = i * 16 + (uint64_t)(k * 4 + j); c += rowptr[r + 1] - rowptr[r]; np++; }
    }
  };
  uint64_t i = blockIdx.x * 256ull + threadIdx.x;
  for (; i + T < n16; i += 2 * T) { const uint4 a = p16[i], b = p16[i + T]; take(i, a); take(i + T, b); }
  for (; i < n16; i += T) take(i, p16[i]);
  for (uint64_t r = n16 * 16 + blockIdx.x * 256ull + threadIdx.x; r < n; r += T) if (pres[r]) { c += rowptr[r + 1] - rowptr[r]; np++; }
  block_add_u64(c, out);
  if (out_count) { __syncthreads(); block_add_u64(np, out_count); }
  scalar_publish(pub);
}
uint64_t frontier_edges(const uint8_t* pres, const uint32_t* rowptr, uint64_t n) {
  if (!n) return 0;
  ScalarSlot slot; slot.zero();
  hipLaunchKernelGGL(k_frontier_edges, dim3(grid_capped(n, 16, 256)), dim3(256), 0, stream(), pres, rowptr, n, slot.pub(), false);
  return slot.read_u64();
}
// the same with the number of present entries as a second result: one kernel, one round trip to the host
uint64_t frontier_edges_and_count(const uint8_t* pres, const uint32_t* rowptr, uint64_t n, uint64_t* count) {
  *count = 0;
  if (!n) return 0;
  ScalarSlot slot; slot.zero();
  hipLaunchKernelGGL(k_frontier_edges, dim3(grid_capped(n, 16, 256)), dim3(256), 0, stream(), pres, rowptr, n, slot.pub(), true);
  uint64_t v[2]; slot.read(v); *count = v[1]; return v[0];
}

// ---- C<M,replace> = accum(C, T) for vectors, in place on (wval, wpres) ---------------------------------
template <class T, bool MATH> __global__ void k_vec_epilogue(uint64_t n, T* __restrict__ wval, uint8_t* __restrict__ wpres,
                                                  const T* __restrict__ tval, const uint8_t* __restrict__ tpres,
                                                  const uint8_t* __restrict__ allow, int accum, bool replace) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
    const bool ok = allow ? allow[i] != 0 : true;
    if (ok) {
      const bool tp = tpres[i] != 0;
      if (accum >= 0) {
        if (tp) {
          if (wpres[i]) wval[i] = apply_binop<T, true, MATH>(accum, wval[i], tval[i]);
          else { wval[i] = tval[i]; wpres[i] = 1; }
        }
      } else {
        if (tp) wval[i] = tval[i];
        wpres[i] = tp ? 1 : 0;
      }
    } else if (replace) {
      wpres[i] = 0;
    }
  }
}
void vec_epilogue(int code, uint64_t n, void* wval, uint8_t* wpres, const void* tval, const uint8_t* tpres,
                  const uint8_t* allow, int accum, bool replace) {
  if (!n) return;
  dispatch_type(code, [&]<class T>() {
    if (accum >= 0 && binop_needs_math(accum)) hipLaunchKernelGGL((k_vec_epilogue<T, true>), dim3(grid_for(n)), dim3(256), 0, stream(), n, (T*)wval, wpres, (const T*)tval, tpres, allow, accum, replace);
    else hipLaunchKernelGGL((k_vec_epilogue<T, false>), dim3(grid_for(n)), dim3(256), 0, stream(), n, (T*)wval, wpres, (const T*)tval, tpres, allow, accum, replace);
  });
}

// ---- monoid reduction of present entries to one scalar -----------------------------------------------------
// pres == nullptr means "all n entries present" (matrix value arrays).
template <class T> __global__ void k_reduce(uint64_t n, const T* __restrict__ val, const uint8_t* __restrict__ pres, int op,
                                            T identity, T* __restrict__ partial) {
  __shared__ T sh[4];
  T acc = identity;
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull)
    if (!pres || pres[i]) acc = apply_binop<T, true, false>(op, acc, val[i]);
  acc = wave_reduce_op<T>(op, acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    T r = sh[0];
    for (int w = 1; w < 4; w++) r = apply_binop<T, true, false>(op, r, sh[w]);
    partial[blockIdx.x] = r;
  }
}
// FP32 values reduced in an FP64 monoid (`reduce_float` of an FP32 vector: the convergence test of a PageRank loop): the first
// level reads the floats and widens on the fly instead of a cast pass over the vector; same fixed two-level tree
__global__ void k_reduce_f32_f64(uint64_t n, const float* __restrict__ val, const uint8_t* __restrict__ pres, int op, double identity, double* __restrict__ partial) {
  __shared__ double sh[4];
  double acc = identity;
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull)
    if (!pres || pres[i]) acc = apply_binop<double, true, false>(op, acc, (double)val[i]);
  acc = wave_reduce_op<double>(op, acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { double r = sh[0]; for (int w = 1; w < 4; w++) r = apply_binop<double, true, false>(op, r, sh[w]); partial[blockIdx.x] = r; }
}
// BOOL with LOR / LAND (the `while q.reduce_bool()` of a BFS loop): "is any present value true / false" — one kernel, 16 bytes
// per lane per step, one atomic per workgroup into the self-cleaning counter slot
__global__ void k_any_byte(uint64_t n, const uint8_t* __restrict__ val, const uint8_t* __restrict__ pres, uint8_t want, const ScalarPub pub) {
  unsigned long long* const out = pub.slot;
  unsigned long long c = 0;
  const uint64_t n16 = n / 16;
  const uint4* v4 = (const uint4*)val; const uint4* p4 = (const uint4*)pres;
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += gridDim.x * 256ull) {
    const uint4 v = v4[i]; uint4 p = pres ? p4[i] : make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
    const uint32_t m = want ? 0u : 0x01010101u;          // count present bytes whose value (0/1) equals `want`
    c += __popc(((v.x ^ m) & 0x01010101u) & ((p.x | (p.x >> 1) | (p.x >> 2) | (p.x >> 3) | (p.x >> 4) | (p.x >> 5) | (p.x >> 6) | (p.x >> 7)) & 0x01010101u));
    c += __popc(((v.y ^ m) & 0x01010101u) & ((p.y | (p.y >> 1) | (p.y >> 2) | (p.y >> 3) | (p.y >> 4) | (p.y >> 5) | (p.y >> 6) | (p.y >> 7)) & 0x01010101u));
    c += __popc(((v.z ^ m) & 0x01010101u) & ((p.z | (p.z >> 1) | (p.z >> 2) | (p.z >> 3) | (p.z >> 4) | (p.z >> 5) | (p.z >> 6) | (p.z >> 7)) & 0x01010101u));
    c += __popc(((v.w ^ m) & 0x01010101u) & ((p.w | (p.w >> 1) | (p.w >> 2) | (p.w >> 3) | (p.w >> 4) | (p.w >> 5) | (p.w >> 6) | (p.w >> 7)) & 0x01010101u));
  }
  for (uint64_t i = n16 * 16 + blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) c += (!pres || pres[i]) && ((val[i] != 0) == (want != 0));
  block_add_u64(c, out);
  scalar_publish(pub);
}
// fixed-shape two-level tree: results are run-to-run deterministic for floating point as well
// up to 64 positions: one lane folds them in index order — one launch instead of two, and for floating point the
// same association as a sequential loop (what the reference's library does with an input this small; its docstring values
// such as Vector.reduce's 0.9517456293106079, pygraphblas/vector.py:1107-1109, come out bit-for-bit)
template <class T> __global__ void k_reduce_seq(uint32_t n, const T* __restrict__ val, const uint8_t* __restrict__ pres, int op, T id, T* __restrict__ out) {
  T acc = id; bool have = false;
  for (uint32_t i = 0; i < n; i++) if (!pres || pres[i]) { const T v = val[i]; acc = have ? apply_binop<T>(op, acc, v) : v; have = true; }
  *out = acc;
}

void reduce_values(int code, uint64_t n, const void* val, const uint8_t* pres, int op, const void* identity, void* result_host) {
  if (code == T_BOOL && n && (op == B_LOR || op == B_LAND) && ((uintptr_t)val % 16 == 0) && (!pres || (uintptr_t)pres % 16 == 0)) {
    ScalarSlot slot; slot.zero();
    hipLaunchKernelGGL(k_any_byte, dim3(grid_capped(n, 16)), dim3(256), 0, stream(), n, (const uint8_t*)val, pres, (uint8_t)(op == B_LOR ? 1 : 0), slot.pub());
    const uint64_t hits = slot.read_u64();
    const uint8_t r = op == B_LOR ? (hits != 0) : (hits == 0);        // LOR: some present value is true; LAND: no present value is false
    memcpy(result_host, &r, 1);
    return;
  }
  dispatch_type(code, [&]<class T>() {
    T id; memcpy(&id, identity, sizeof(T));
    if (!n) { memcpy(result_host, &id, sizeof(T)); return; }
    const int g = grid_for(n, 4);
    DevBuf part((size_t)g * sizeof(T)), fin(sizeof(T) * 1);
    if (n <= 64) hipLaunchKernelGGL((k_reduce_seq<T>), dim3(1), dim3(1), 0, stream(), (uint32_t)n, (const T*)val, pres, op, id, fin.as<T>());
    else {
      hipLaunchKernelGGL((k_reduce<T>), dim3(g), dim3(256), 0, stream(), n, (const T*)val, pres, op, id, part.as<T>());
      hipLaunchKernelGGL((k_reduce<T>), dim3(1), dim3(256), 0, stream(), (uint64_t)g, (const T*)part.as<T>(), (const uint8_t*)nullptr, op, id, fin.as<T>());
    }
    void* pin = pinned_scratch();
    GRB_HIP(hipMemcpyAsync(pin, fin.p, sizeof(T), hipMemcpyDeviceToHost, stream()));
    GRB_HIP(hipStreamSynchronize(stream()));
    memcpy(result_host, pin, sizeof(T));
  });
}

void reduce_values_f32_f64(uint64_t n, const void* val_f32, const uint8_t* pres, int op, const void* identity_f64, void* result_host) {
  double id; memcpy(&id, identity_f64, 8);
  if (!n) { memcpy(result_host, &id, 8); return; }
  const int g = grid_for(n, 4);
  DevBuf part((size_t)g * 8), fin(8);
  hipLaunchKernelGGL(k_reduce_f32_f64, dim3(g), dim3(256), 0, stream(), n, (const float*)val_f32, pres, op, id, part.as<double>());
  hipLaunchKernelGGL((k_reduce<double>), dim3(1), dim3(256), 0, stream(), (uint64_t)g, (const double*)part.as<double>(), (const uint8_t*)nullptr, op, id, fin.as<double>());
  void* pin = pinned_scratch();
  GRB_HIP(hipMemcpyAsync(pin, fin.p, 8, hipMemcpyDeviceToHost, stream()));
  GRB_HIP(hipStreamSynchronize(stream()));
  memcpy(result_host, pin, 8);
}

// ---- element-wise union / intersection of two bitmap vectors ----------------------------------------------------
template <class T, bool MATH> __global__ void k_vec_ewise(uint64_t n, const T* __restrict__ uval, const uint8_t* __restrict__ upres,
                                               const T* __restrict__ vval, const uint8_t* __restrict__ vpres, int op, bool is_union,
                                               T* __restrict__ tval, uint8_t* __restrict__ tpres) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
    const bool a = upres[i] != 0, b = vpres[i] != 0;
    if (a && b) { tval[i] = apply_binop<T, true, MATH>(op, uval[i], vval[i]); tpres[i] = 1; }
    else if (is_union && a) { tval[i] = uval[i]; tpres[i] = 1; }
    else if (is_union && b) { tval[i] = vval[i]; tpres[i] = 1; }
    else tpres[i] = 0;
  }
}
void vec_ewise(int code, uint64_t n, const void* uval, const uint8_t* upres, const void* vval, const uint8_t* vpres, int op,
               bool is_union, void* tval, uint8_t* tpres) {
  if (!n) return;
  dispatch_type(code, [&]<class T>() {
    if (binop_needs_math(op)) hipLaunchKernelGGL((k_vec_ewise<T, true>), dim3(grid_for(n)), dim3(256), 0, stream(), n, (const T*)uval, upres, (const T*)vval, vpres, op, is_union, (T*)tval, tpres);
    else hipLaunchKernelGGL((k_vec_ewise<T, false>), dim3(grid_for(n)), dim3(256), 0, stream(), n, (const T*)uval, upres, (const T*)vval, vpres, op, is_union, (T*)tval, tpres);
  });
}

// ---- w<mask, replace> = accum(w, u op v) in ONE pass (round 6, second half) ------------------------------------------------------------------------
// The general route is three kernels and two temporaries: k_allow (mask -> allow bytes), k_vec_ewise (T = u op v), k_vec_epilogue (w from T).  When every
// operand, the operator and the accumulator work in w's own type, one kernel does it: the mask is read in place (mask_truth_at), T(i) lives in a register.
// Element-wise, everything of position i is read before anything of position i is written: w may be u, v or the mask (no __restrict__ on those).
// (The element-wise steps of the BC driver on its ns x n batches — `bcu.emult(paths, DIV, out=W, mask=S[i], desc=R)`, `W.emult(paths, TIMES, out=bcu, accum=PLUS)`,
//  `paths.assign_matrix(frontier, accum=PLUS)` — are this; 155 -> ~60 us each at R-MAT-22, ns = 4.)
template <class T, bool MATH> __global__ void k_vec_ewise_fused(uint64_t n, const T* uval, const uint8_t* upres, const T* vval, const uint8_t* vpres, int op, bool is_union,
                                                                int mcode, const void* mval, const uint8_t* mpres, bool mstruct, bool mcomp, int accum, bool replace,
                                                                T* wval, uint8_t* wpres) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
    const bool ok = mpres ? ((mpres[i] != 0 && mask_truth_at(mval, mcode, i, mstruct)) != mcomp) : true;
    if (ok) {
      const bool a = upres[i] != 0, b = vpres[i] != 0;
      const bool tp = is_union ? (a || b) : (a && b);
      T tv{};
      if (a && b) tv = apply_binop<T, true, MATH>(op, uval[i], vval[i]);
      else if (tp) tv = a ? uval[i] : vval[i];
      if (accum >= 0) {
        if (tp) {
          if (wpres[i]) wval[i] = apply_binop<T, true, MATH>(accum, wval[i], tv);
          else { wval[i] = tv; wpres[i] = 1; }
        }
      } else {
        if (tp) wval[i] = tv;
        wpres[i] = tp ? 1 : 0;
      }
    } else if (replace) {
      wpres[i] = 0;
    }
  }
}
void vec_ewise_fused(int code, uint64_t n, const void* uval, const uint8_t* upres, const void* vval, const uint8_t* vpres, int op, bool is_union,
                     int mcode, const void* mval, const uint8_t* mpres, bool mstruct, bool mcomp, int accum, bool replace, void* wval, uint8_t* wpres) {
  if (!n) return;
  dispatch_type(code, [&]<class T>() {
    if (binop_needs_math(op) || (accum >= 0 && binop_needs_math(accum)))
      hipLaunchKernelGGL((k_vec_ewise_fused<T, true>), dim3(grid_for(n)), dim3(256), 0, stream(), n, (const T*)uval, upres, (const T*)vval, vpres, op, is_union, mcode, mval, mpres, mstruct, mcomp, accum, replace, (T*)wval, wpres);
    else
      hipLaunchKernelGGL((k_vec_ewise_fused<T, false>), dim3(grid_for(n)), dim3(256), 0, stream(), n, (const T*)uval, upres, (const T*)vval, vpres, op, is_union, mcode, mval, mpres, mstruct, mcomp, accum, replace, (T*)wval, wpres);
  });
}

// ---- apply: unary op, or binary op with one bound scalar (mode 1: z=f(s,x)  mode 2: z=f(x,s)) ----------------------
template <class T, bool MATH> __global__ void k_vec_apply(uint64_t n, const T* __restrict__ uval, const uint8_t* __restrict__ upres, int mode, int op,
                                               T s, T* __restrict__ tval, uint8_t* __restrict__ tpres) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
    const bool a = upres ? upres[i] != 0 : true;
    if (a) {
      T x = uval[i];
      tval[i] = mode == 0 ? apply_unop<T, MATH>(op, x) : (mode == 1 ? apply_binop<T, true, MATH>(op, s, x) : apply_binop<T, true, MATH>(op, x, s));
    }
    if (tpres) tpres[i] = a ? 1 : 0;
  }
}
void vec_apply(int code, uint64_t n, const void* uval, const uint8_t* upres, int mode, int op, const void* scalar, void* tval, uint8_t* tpres) {
  if (!n) return;
  dispatch_type(code, [&]<class T>() {
    T s{}; if (scalar) memcpy(&s, scalar, sizeof(T));
    if (mode == 0 ? unop_needs_math(op) : binop_needs_math(op)) hipLaunchKernelGGL((k_vec_apply<T, true>), dim3(grid_for(n)), dim3(256), 0, stream(), n, (const T*)uval, upres, mode, op, s, (T*)tval, tpres);
    else hipLaunchKernelGGL((k_vec_apply<T, false>), dim3(grid_for(n)), dim3(256), 0, stream(), n, (const T*)uval, upres, mode, op, s, (T*)tval, tpres);
  });
}

// ---- w<allow>(:) = accum(w, scalar) over all indices (GrB_Vector_assign_<T> with GrB_ALL) ---------------------------------
template <class T, bool MATH> __global__ void k_vec_assign_scalar(uint64_t n, T* __restrict__ wval, uint8_t* __restrict__ wpres,
                                                       const uint8_t* __restrict__ allow, const uint8_t* __restrict__ region, T s, int accum, bool replace) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
    const bool ok = allow ? allow[i] != 0 : true;
    const bool in = region ? region[i] != 0 : true;
    if (ok && in) {
      if (accum >= 0 && wpres[i]) wval[i] = apply_binop<T, true, MATH>(accum, wval[i], s); else wval[i] = s;
      wpres[i] = 1;
    } else if (!ok && replace) wpres[i] = 0;
  }
}
void vec_assign_scalar(int code, uint64_t n, void* wval, uint8_t* wpres, const uint8_t* allow, const uint8_t* region, const void* scalar, int accum, bool replace) {
  if (!n) return;
  dispatch_type(code, [&]<class T>() {
    T s; memcpy(&s, scalar, sizeof(T));
    if (accum >= 0 && binop_needs_math(accum)) hipLaunchKernelGGL((k_vec_assign_scalar<T, true>), dim3(grid_for(n)), dim3(256), 0, stream(), n, (T*)wval, wpres, allow, region, s, accum, replace);
    else hipLaunchKernelGGL((k_vec_assign_scalar<T, false>), dim3(grid_for(n)), dim3(256), 0, stream(), n, (T*)wval, wpres, allow, region, s, accum, replace);
  });
}

template <class T, bool MATH> __global__ void k_vec_assign_scalar_masked(uint64_t n, T* __restrict__ wval, uint8_t* __restrict__ wpres, int mcode, const void* __restrict__ mval,
                                                                         const uint8_t* __restrict__ mpres, bool mstruct, bool mcomp, T s, int accum, bool replace) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
    const bool ok = (mpres[i] != 0 && mask_truth_at(mval, mcode, i, mstruct)) != mcomp;
    if (ok) {
      if (accum >= 0 && wpres[i]) wval[i] = apply_binop<T, true, MATH>(accum, wval[i], s); else wval[i] = s;
      wpres[i] = 1;
    } else if (replace) wpres[i] = 0;
  }
}
// the same with one-byte values on both sides and no accumulator (`v[q] = level`: UINT8 levels under a BOOL frontier, once per BFS level):
// 16 positions per thread — 16-byte loads and stores, the mask test and the select done on four bytes at a time — instead of a byte per
// thread and instruction (7-13 us for 4 M positions; the bytes alone are 6 streams of 4 MB)
__device__ __forceinline__ uint32_t nz_bytes(uint32_t x) { return ((x | ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu)) & 0x80808080u) >> 7; }      // 0x01 in every byte of x that is not zero
__global__ void k_assign_masked_bytes(uint64_t n, uint8_t* __restrict__ wval, uint8_t* __restrict__ wpres, const uint8_t* __restrict__ mval, const uint8_t* __restrict__ mpres,
                                      bool mstruct, bool mcomp, uint8_t s, bool replace, uint8_t* __restrict__ code) {
  const uint64_t nv = n / 16; const uint32_t s4 = (uint32_t)s * 0x01010101u;
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < nv; i += gridDim.x * 256ull) {
    const uint4 mp = ((const uint4*)mpres)[i]; uint4 mv = make_uint4(0, 0, 0, 0);
    if (!mstruct) mv = ((const uint4*)mval)[i];
    uint4 wv = ((const uint4*)wval)[i], wp = ((const uint4*)wpres)[i];
    uint32_t* pmp = (uint32_t*)&mp; uint32_t* pmv = (uint32_t*)&mv; uint32_t* pwv = (uint32_t*)&wv; uint32_t* pwp = (uint32_t*)&wp;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint32_t ok = nz_bytes(pmp[k]) & (mstruct ? 0x01010101u : nz_bytes(pmv[k]));
      if (mcomp) ok ^= 0x01010101u;
      const uint32_t m = ok * 0xFFu;                                        // 0xFF in the bytes that are written
      pwv[k] = (pwv[k] & ~m) | (s4 & m);
      pwp[k] = replace ? ok : (nz_bytes(pwp[k]) | ok);
    }
    ((uint4*)wval)[i] = wv; ((uint4*)wpres)[i] = wp;
    if (code) {                                                             // the vector's code bytes (GrB_Vector_opaque::dcode): present | (present and not zero) << 1
      uint4 cd; uint32_t* pc = (uint32_t*)&cd;
#pragma unroll
      for (int k = 0; k < 4; k++) pc[k] = pwp[k] | ((nz_bytes(pwv[k]) & pwp[k]) << 1);
      ((uint4*)code)[i] = cd;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (uint64_t i = nv * 16; i < n; i++) {
      const bool ok = (mpres[i] != 0 && (mstruct || mval[i] != 0)) != mcomp;
      if (ok) { wval[i] = s; wpres[i] = 1; } else if (replace) wpres[i] = 0;
      if (code) code[i] = (uint8_t)((wpres[i] != 0 ? 1 : 0) | ((wpres[i] != 0 && wval[i] != 0) ? 2 : 0));
    }
}
// the code bytes of a one-byte-typed vector from scratch (a vector that was not written by the kernel above)
__global__ void k_vec_code_bytes(uint64_t n, const uint8_t* __restrict__ val, const uint8_t* __restrict__ pres, uint8_t* __restrict__ code) {
  const uint64_t nv = n / 16;
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < nv; i += gridDim.x * 256ull) {
    const uint4 v = ((const uint4*)val)[i], pp = ((const uint4*)pres)[i];
    const uint32_t* pv = (const uint32_t*)&v; const uint32_t* pq = (const uint32_t*)&pp;
    uint4 cd; uint32_t* pc = (uint32_t*)&cd;
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t pr = nz_bytes(pq[k]); pc[k] = pr | ((nz_bytes(pv[k]) & pr) << 1); }
    ((uint4*)code)[i] = cd;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) for (uint64_t i = nv * 16; i < n; i++) code[i] = (uint8_t)((pres[i] != 0 ? 1 : 0) | ((pres[i] != 0 && val[i] != 0) ? 2 : 0));
}
void vec_code_bytes(uint64_t n, const uint8_t* val, const uint8_t* pres, uint8_t* code) {
  if (!n) return;
  hipLaunchKernelGGL(k_vec_code_bytes, dim3(grid_for(n / 16 + 1)), dim3(256), 0, stream(), n, val, pres, code);
}
bool vec_assign_scalar_masked(int code, uint64_t n, void* wval, uint8_t* wpres, int mcode, const void* mval, const uint8_t* mpres, bool mstruct, bool mcomp, const void* scalar, int accum, bool replace,
                              uint8_t* code_out) {
  if (!n) return false;
  if (accum < 0 && type_size(code) == 1 && type_size(mcode) == 1 && n >= 4096) {      // (the truth of a one-byte value of any type is "not zero")
    hipLaunchKernelGGL(k_assign_masked_bytes, dim3(grid_for(n / 16)), dim3(256), 0, stream(), n, (uint8_t*)wval, wpres, (const uint8_t*)mval, mpres, mstruct, mcomp, *(const uint8_t*)scalar, replace, code_out);
    return code_out != nullptr;                                                         // the code bytes of every position were written on the way
  }
  dispatch_type(code, [&]<class T>() {
    T s; memcpy(&s, scalar, sizeof(T));
    if (accum >= 0 && binop_needs_math(accum)) hipLaunchKernelGGL((k_vec_assign_scalar_masked<T, true>), dim3(grid_for(n)), dim3(256), 0, stream(), n, (T*)wval, wpres, mcode, mval, mpres, mstruct, mcomp, s, accum, replace);
    else hipLaunchKernelGGL((k_vec_assign_scalar_masked<T, false>), dim3(grid_for(n)), dim3(256), 0, stream(), n, (T*)wval, wpres, mcode, mval, mpres, mstruct, mcomp, s, accum, replace);
  });
  return false;
}

__global__ void k_allow_and_bool(uint64_t n, int mcode, const void* __restrict__ mval, const uint8_t* __restrict__ mpres, bool structural, bool complement,
                                 uint8_t* __restrict__ allow, uint8_t* __restrict__ as_bool) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
    const bool truth = mask_truth_at(mval, mcode, i, false);
    const bool m = mpres[i] != 0 && (structural || truth);
    allow[i] = (uint8_t)(m != complement);
    as_bool[i] = truth ? 1 : 0;
  }
}
void build_allow_and_bool(uint64_t n, int mcode, const void* mval, const uint8_t* mpres, bool structural, bool complement, uint8_t* allow, uint8_t* as_bool) {
  if (!n) return;
  hipLaunchKernelGGL(k_allow_and_bool, dim3(grid_for(n)), dim3(256), 0, stream(), n, mcode, mval, mpres, structural, complement, allow, as_bool);
}

// ---- value-based select on a bitmap vector / value array: keep[i] = pred(val[i]) --------------------------------------------
template <class T> __global__ void k_select_value(uint64_t n, const T* __restrict__ val, const uint8_t* __restrict__ pres, int sel, T thunk,
                                                  uint8_t* __restrict__ keep) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
    bool k = pres ? pres[i] != 0 : true;
    if (k) {
      const T x = val[i]; const T z = T();
      switch (sel) {
        case SEL_NONZERO: k = (bool)cast_to<bool8, T>(x); break;
        case SEL_EQ_ZERO: k = !(bool)cast_to<bool8, T>(x); break;
        case SEL_GT_ZERO: k = apply_binop<T>(B_ISGT, x, z); break;
        case SEL_GE_ZERO: k = apply_binop<T>(B_ISGE, x, z); break;
        case SEL_LT_ZERO: k = apply_binop<T>(B_ISLT, x, z); break;
        case SEL_LE_ZERO: k = apply_binop<T>(B_ISLE, x, z); break;
        case SEL_NE_THUNK: k = apply_binop<T>(B_ISNE, x, thunk); break;
        case SEL_EQ_THUNK: k = apply_binop<T>(B_ISEQ, x, thunk); break;
        case SEL_GT_THUNK: k = apply_binop<T>(B_ISGT, x, thunk); break;
        case SEL_GE_THUNK: k = apply_binop<T>(B_ISGE, x, thunk); break;
        case SEL_LT_THUNK: k = apply_binop<T>(B_ISLT, x, thunk); break;
        case SEL_LE_THUNK: k = apply_binop<T>(B_ISLE, x, thunk); break;
        default: break;
      }
    }
    keep[i] = k ? 1 : 0;
  }
}
void select_value_flags(int code, uint64_t n, const void* val, const uint8_t* pres, int sel, const void* thunk, uint8_t* keep) {
  if (!n) return;
  dispatch_type(code, [&]<class T>() {
    T t{}; if (thunk) memcpy(&t, thunk, sizeof(T));
    hipLaunchKernelGGL((k_select_value<T>), dim3(grid_for(n)), dim3(256), 0, stream(), n, (const T*)val, pres, sel, t, keep);
  });
}

}  // namespace grb
