// grb_spmv_tiles.hpp — the panel pipeline of kernel X (grb_spmv_xcd.hpp): kernel W's wave pipeline specialised for a
// column panel, whose sub-rows are never empty.
//
// What that buys over kernel W (grb_spmv_wavepipe.hpp):
//   * a task is a *tile* of 256 consecutive entries (W: 256 items of the entry/row-end merge, i.e. ~227 entries): every
//     lane slot carries an entry and the kernel needs no task descriptors;
//   * the sub-row an entry belongs to follows from the row-start flags that come with the column words (bit 31): it is
//     the sub-row of the tile's first entry (one word per tile) plus the number of row starts before it, which rides in
//     the flag word of the segmented wave scan.  The kernel never reads row pointers and has no per-row pass;
//   * a sub-row ends where the next entry starts one; the sums of the sub-rows that end in a tile are consecutive
//     sub-rows, so they pass through 64 staging slots per wave and leave as coalesced stores.  There is no scan buffer in
//     LDS: the LDS table grows from 16 382 to 19 454 FP64 columns per panel.
// Round 2:
//   * the plan starts a new sub-row at every chunk boundary (grb_spmv_xcd.hpp), so no sub-row spans two chunks: the
//     kernel keeps no carry records and the per-call fix-up kernel is gone (the merge kernel adds the extra partial);
//   * a wave runs ONE software pipeline over the whole sequence of chunks it is handed (static range first, then the
//     dynamic ones): the loads of the next chunk's first tiles are in flight while the current chunk is reduced, instead
//     of a drain and a refill (a full memory latency) every 4 tiles;
//   * the column words run XT_DEPTH + 2 tiles ahead of the reduction and the values / gathers XT_DEPTH tiles, and every
//     load is issued unconditionally (tiles behind the end of the work carry out-of-range offsets and make no memory
//     request), so the s_waitcnt counts are static and a wait for tile t leaves the loads of the later tiles in flight.
#pragma once
#include "grb_spmv_wavepipe.hpp"
#include <utility>

namespace grb {

#ifndef XT_DEPTH
#define XT_DEPTH 1                     // values + gathers this many tiles ahead of the reduction, column words two more
#endif
#ifndef XT_WAVES
#define XT_WAVES WP_WAVES              // waves per workgroup of the panel pipeline (one workgroup per CU)
#endif
#ifndef XT_C16
#define XT_C16 1                       // 1: 16-bit column plane + 16-bit extras for the cold columns; 0: one 32-bit column word per entry
#endif
// all of the LDS but the staging slots is the table: 19454 slots of 8 bytes.  With the 16-bit column plane a word holds a
// 15-bit code — a slot, or XT escape + the high bits of a cold column — so the table is capped at 24576 slots (4-byte and
// smaller types; 8192 escape codes x 65536 = 2^29 columns) and kernel X takes matrices of up to xt_hot<T>::MAXCOLS columns.
// The 16-bit plane is used for 8-byte types only: with 4-byte values the table would shrink from 39932 to 24576 slots, and the
// pattern-only FP32 product (PageRank's PLUS_SECOND) measured 130 us with it against 105 us with 32-bit words.
template <class T> struct xt_fmt { static constexpr bool C16 = XT_C16 != 0 && sizeof(T) >= 8; };
// 32-bit entry words (round 4): bit 31 first entry of a sub-row, bit 29 (XT_COLD) the column is not in the table, bits 28..0 the slot in the LDS
// table or the column itself.  `off = (word & 0x3FFFFFFF) * sizeof(T)` — `word << 2` for 4-byte values — is the byte offset with the row-start
// flag gone and the cold flag on top of it (bit 31 for 4-byte values).  Two instructions turn it into both addresses, no compare, no select:
//   gather from u   off ^ cold-bit   a cold entry's offset into u; a hot entry's lands beyond the descriptor's range (u has < 2^29 elements): the
//                                    load returns zero bits and makes no memory request
//   LDS table       min(off, last)   a hot entry's slot; a cold one reads the table's LAST slot, which holds zero bits and is never given to a
//                                    column (HOT = H - 1)
// and the operand's value is the OR of the two — no test at consume time either.
constexpr int XT_COLD_BIT = 29;
constexpr uint32_t XT_COLD = 1u << XT_COLD_BIT, XT_IDXMASK = XT_COLD - 1u;
template <class T> struct xt_hot {
  static constexpr int HLDS = (WP_LDS_BYTES - 16 - WP_WAVES * 64 * (int)sizeof(T)) / (int)sizeof(T);
  static constexpr int H = xt_fmt<T>::C16 ? (HLDS < 24576 ? HLDS : 24576) : HLDS;
  static constexpr int HOT = xt_fmt<T>::C16 ? H : H - 1;          // columns a table serves
  static constexpr uint64_t C16COLS = (uint64_t)(32768 - H) << 16, MAXCOLS = xt_fmt<T>::C16 && C16COLS < (uint64_t)XT_IDXMASK ? C16COLS : (uint64_t)XT_IDXMASK;      // (the plan is built from 32-bit words in either format)
};
template <class E> __device__ __forceinline__ E xt_or_bits(E a, E b) {
  if constexpr (sizeof(E) == 8) { union { E e; unsigned long long u; } x, y; x.e = a; y.e = b; x.u |= y.u; return x.e; }
  else if constexpr (sizeof(E) == 4) { union { E e; uint32_t u; } x, y; x.e = a; y.e = b; x.u |= y.u; return x.e; }
  else if constexpr (sizeof(E) == 2) { union { E e; uint16_t u; } x, y; x.e = a; y.e = b; x.u |= y.u; return x.e; }
  else { union { E e; uint8_t u; } x, y; x.e = a; y.e = b; x.u |= y.u; return x.e; }
}

// the segmented scan of the sums and the prefix count of the row starts in one pass: x = flag << 31 | count
template <class T, class SR> __device__ __forceinline__ void xt_seg_scan_count(T& v, uint32_t& x, int lane, const SR& sr) {
  if constexpr (sizeof(T) == 4 || sizeof(T) == 8) {
    const int l16 = lane & 15, row = (lane >> 4) & 3;
#define XT_SC_STEP(CTRL, MASK, COND) { const T vu = dpp_move_t<T, CTRL, MASK>(v); const uint32_t xu = dpp_mov<CTRL, MASK>(x, x); \
                                       if (COND) { if (!(x >> 31)) v = sr.add(vu, v); x = (x + (xu & 0x7FFFFFFFu)) | (xu & 0x80000000u); } }
    XT_SC_STEP(0x111, 0xf, l16 >= 1) XT_SC_STEP(0x112, 0xf, l16 >= 2) XT_SC_STEP(0x114, 0xf, l16 >= 4) XT_SC_STEP(0x118, 0xf, l16 >= 8)
    XT_SC_STEP(0x142, 0xa, row == 1 || row == 3) XT_SC_STEP(0x143, 0xc, row >= 2)
#undef XT_SC_STEP
  } else {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const T vu = shfl_up_t<T>(v, d); const uint32_t xu = (uint32_t)__shfl_up((int)x, d, 64);
      if (lane >= d) { if (!(x >> 31)) v = sr.add(vu, v); x = (x + (xu & 0x7FFFFFFFu)) | (xu & 0x80000000u); }
    }
  }
}
// Round 3: the same scan with the segment flags kept OUT of the vector registers.  Which lanes may take the value `d` lanes below them at
// a step depends only on the flag mask F (one bit per lane: "a sub-row starts in this lane's entries"): lane i accepts iff no flag in
// the d lanes ending at i — the OR of F over that window, which the scalar unit computes for all 64 lanes at once (G2 = G1 | G1 << 1, ...,
// masked at the 16-lane row boundaries DPP shifts respect).  The vector side of a step is then: DPP move, add, one v_cndmask on the
// accept mask — 3 instructions for 4-byte values where the flag-carrying form took ~10 (DPP move of value and flag word, lane-range
// compare, flag test, add, select, four operations on the flag / count word).  The count of row starts that rode in the flag word is
// taken from ballots instead (v_mbcnt).  Same steps, same association: the sums come out bit for bit as before.
template <class E> __device__ __forceinline__ E xt_select_mask(E if_clear, E if_set, unsigned long long mask) {     // per lane: bit(lane) of mask ? if_set : if_clear
  if constexpr (sizeof(E) == 8) {
    union { E e; uint32_t u[2]; } a, b, r; a.e = if_clear; b.e = if_set;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r.u[0]) : "v"(a.u[0]), "v"(b.u[0]), "s"(mask));
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r.u[1]) : "v"(a.u[1]), "v"(b.u[1]), "s"(mask));
    return r.e;
  } else {
    union { E e; uint32_t u; } a, b, r; a.u = 0; b.u = 0; a.e = if_clear; b.e = if_set;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r.u) : "v"(a.u), "v"(b.u), "s"(mask));
    return r.e;
  }
}
template <class T, int CTRL, int ROW_MASK> __device__ __forceinline__ T xt_dpp_move0(T v) {      // lanes without a source (or outside ROW_MASK) get zero bits
  if constexpr (sizeof(T) == 8) {
    union { T t; int u[2]; } a, r; a.t = v;
    r.u[0] = __builtin_amdgcn_update_dpp(0, a.u[0], CTRL, ROW_MASK, 0xf, true); r.u[1] = __builtin_amdgcn_update_dpp(0, a.u[1], CTRL, ROW_MASK, 0xf, true); return r.t;
  } else {
    union { T t; int u; } a, r; a.u = 0; a.t = v; r.u = __builtin_amdgcn_update_dpp(0, a.u, CTRL, ROW_MASK, 0xf, true); return r.t;
  }
}
template <class T, class SR> __device__ __forceinline__ void xt_seg_scan_masked(T& v, unsigned long long F, const SR& sr) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "DPP scan handles 4- and 8-byte types");
  constexpr unsigned long long LT1 = 0x0001000100010001ull, LT2 = 0x0003000300030003ull, LT4 = 0x000F000F000F000Full, LT8 = 0x00FF00FF00FF00FFull;
  constexpr unsigned long long ROWS13 = 0xFFFF0000FFFF0000ull, ROWS23 = 0xFFFFFFFF00000000ull, ROW3 = 0xFFFF000000000000ull;
  // G_d(i) = some flag among the d lanes ending at lane i, inside i's row of 16
  const unsigned long long G1 = F;
  const unsigned long long G2 = G1 | ((G1 << 1) & ~LT1);
  const unsigned long long G4 = G2 | ((G2 << 2) & ~LT2);
  const unsigned long long G8 = G4 | ((G4 << 4) & ~LT4);
  const unsigned long long G16 = G8 | ((G8 << 8) & ~LT8);                 // a flag between the start of the row and lane i
  // (the value from below arrives with zeros in the lanes that have no source — those lanes never accept — so that the compiler can fold the move into
  //  the add where the type has a DPP add: 2 instructions per step and 32-bit half instead of 4)
#define XT_MS_STEP(CTRL, MASK, ACCEPT) { const T vu = xt_dpp_move0<T, CTRL, MASK>(v); const T sum = sr.add(vu, v); v = xt_select_mask<T>(v, sum, (ACCEPT)); }
  XT_MS_STEP(0x111, 0xf, ~G1 & ~LT1) XT_MS_STEP(0x112, 0xf, ~G2 & ~LT2) XT_MS_STEP(0x114, 0xf, ~G4 & ~LT4) XT_MS_STEP(0x118, 0xf, ~G8 & ~LT8)
  XT_MS_STEP(0x142, 0xa, ~G16 & ROWS13)                                   // lane 15 / 47 into the next row
  // rows 2 and 3 take lane 31: nothing may start between lane 32 and lane i — for row 3 that includes all of row 2 (its lane 47 in G16)
  const unsigned long long H = G16 | (((G16 >> 47) & 1ull) ? ROW3 : 0ull);
  XT_MS_STEP(0x143, 0xc, ~H & ROWS23)
#undef XT_MS_STEP
}
template <class E> __device__ __forceinline__ E xt_wave_shr1(E v, E into_lane0) {     // lane i takes lane i - 1's value, lane 0 `into_lane0`
  if constexpr (sizeof(E) == 8) {
    union { E e; int i[2]; } a, o, r; a.e = v; o.e = into_lane0;
    r.i[0] = __builtin_amdgcn_update_dpp(o.i[0], a.i[0], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    r.i[1] = __builtin_amdgcn_update_dpp(o.i[1], a.i[1], 0x138, 0xf, 0xf, false); return r.e;
  } else {
    union { E e; int i; } a, o, r; a.i = 0; o.i = 0; a.e = v; o.e = into_lane0;
    r.i = __builtin_amdgcn_update_dpp(o.i, a.i, 0x138, 0xf, 0xf, false); return r.e;
  }
}
template <class E> __device__ __forceinline__ E xt_readlane(E v, int src) {          // src wave-uniform
  if constexpr (sizeof(E) == 8) { union { E e; int i[2]; } u; u.e = v; u.i[0] = __builtin_amdgcn_readlane(u.i[0], src); u.i[1] = __builtin_amdgcn_readlane(u.i[1], src); return u.e; }
  else if constexpr (sizeof(E) == 4) { union { E e; int i; } u; u.e = v; u.i = __builtin_amdgcn_readlane(u.i, src); return u.e; }
  else { union { E e; uint16_t s; } u; u.s = 0; u.e = v; const int x = __builtin_amdgcn_readlane((int)u.s, src); u.s = (uint16_t)x; return u.e; }
}

// a gather through a buffer descriptor: lanes whose offset is out of range (0xFFFFFFFF) return 0 and make no memory
// request at all.  The lanes served by the LDS table (3 in 4) and the lanes behind the end of a tile use that: a plain
// global load costs the texture path a cycle per lane even when most lanes read the same dummy address.
// Cache policy: default.  (Measured: sc0 or sc1 on these loads change nothing; nt makes them leave the L2 and the whole
// product 0.285 -> 0.392 ms — the L2 residency of the panel's lines is what the design lives on.)
typedef uint32_t xt_v2u __attribute__((ext_vector_type(2)));
#ifndef XT_GATHER_AUX
#define XT_GATHER_AUX 0
#endif
template <class E> __device__ __forceinline__ E xt_buf_load(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  if constexpr (sizeof(E) == 8) { union { xt_v2u w; E e; } x; x.w = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, XT_GATHER_AUX); return x.e; }
  else if constexpr (sizeof(E) == 4) { union { uint32_t w; E e; } x; x.w = __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, XT_GATHER_AUX); return x.e; }
  else if constexpr (sizeof(E) == 2) { union { uint16_t w; E e; } x; x.w = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(r, (int)off, 0, 0); return x.e; }
  else { union { uint8_t w; E e; } x; x.w = (uint8_t)__builtin_amdgcn_raw_buffer_load_b8(r, (int)off, 0, 0); return x.e; }
}

// the two streams of a tile through buffer descriptors; lanes behind the end of the panel read zeros, and so does a whole
// tile whose offset is out of range (`off` in bytes).  XT_STREAM_AUX = the cache-policy bits of these loads (1 sc0, 2 nt,
// 16 sc1).  Measured on R-MAT-22 FP64 (ms per mxv, stream-only variant in brackets): 0: 0.300 (0.226), sc0: 0.300 (0.227),
// nt: 0.292 (0.212), sc1: 0.318 (0.240), sc0+sc1+nt: 0.293 (0.213).
#ifndef XT_STREAM_AUX
#define XT_STREAM_AUX 2
#endif
typedef uint32_t xt_v4u __attribute__((ext_vector_type(4)));
template <class E, int N> __device__ __forceinline__ void xt_stream_load(__amdgpu_buffer_rsrc_t r, uint32_t off, E (&out)[N]) {
  static_assert((sizeof(E) * N) % 16 == 0 || sizeof(E) * N == 4 || sizeof(E) * N == 8, "tile slice per lane");
  if constexpr ((sizeof(E) * N) % 16 == 0) {
    xt_v4u tmp[sizeof(E) * N / 16];
#pragma unroll
    for (int j = 0; j < (int)(sizeof(E) * N / 16); j++) tmp[j] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(off == 0xFFFFFFFFu ? off : off + 16u * j), 0, XT_STREAM_AUX);
    __builtin_memcpy(&out[0], &tmp[0], sizeof(E) * N);
  } else if constexpr (sizeof(E) * N == 8) {
    xt_v2u tmp = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, XT_STREAM_AUX); __builtin_memcpy(&out[0], &tmp, 8);
  } else {
    uint32_t tmp = __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, XT_STREAM_AUX); __builtin_memcpy(&out[0], &tmp, 4);
  }
}

// one panel's share of the plan; the block of XP of these lives in HBM and never changes between calls
template <class T> struct XtPanel {
  const uint32_t* pcol;      // 32-bit format: column words of the panel's entries (tile t = entries [256 t, 256 t + 256)); nullptr once packed
  const T* aval;             // their values (nullptr when the plan was built for multipliers that ignore them)
  const uint32_t* trow;      // 32-bit format: [ntiles] sub-row (numbered over all panels) of every tile's first entry
  const T* xhot;             // the LDS table's contents for this call, T[nhot] (k_xp_hot_gather)
  uint32_t nnz, ntiles, tiles_per_chunk, nhot, static_pct;
  uint32_t interleave;       // 1: a wave's static chunks are strided over the stream (chunk i of wave w = w + i * waves) instead of one contiguous range —
                             //    with sub-panels (XcdPlan::S > 1) all waves of the XCD must be in the same part of the stream at the same time
  // 16-bit format: bit 15 of a word = first entry of a sub-row; the low 15 bits are the slot in the LDS table, or H + (column >> 16)
  // for a column the table does not hold — whose low 16 bits are the next halfword of `extras`, the cold entries of all tiles in
  // entry order.  tinfo[2t] = sub-row of tile t's first entry, tinfo[2t + 1] = index in `extras` of its first cold entry.
  const uint16_t* col16; const uint32_t* tinfo; const uint16_t* extras; uint64_t nextras;
};
template <class T> struct XtCall { const T* u; uint32_t ulen; uint32_t sps; T* partial; };     // what changes from call to call (sps: streams per XCD — a plan constant that rides along)

template <class T> struct XtStage {       // what one tile has in flight
  uint32_t c[WP_PER]; T v[WP_PER], g[WP_PER]; uint32_t rf; uint32_t tile;   // column words (32-bit form), values, gathered operands; rf: sub-row of its first entry
  T hl[WP_PER];                             // 32-bit format: what the LDS table holds for the entry (its zero slot for a cold one), read when the gather is issued
  uint32_t h[2], xq[3], cb, xr;             // 16-bit format: the four raw words, three dwords of extras, the tile's base in `extras`, index of the lane's first extra
};
typedef uint32_t xt_v3u __attribute__((ext_vector_type(3)));

#ifdef XT_PROFILE
// measurement build (make BUILD=build_prof LIB=../libgrb_prof.so XTFLAGS=-DXT_PROFILE): cycles of every wave of the last launch by phase —
// [0] until the tile's products exist (waits for column words, values, gathers, LDS table), [1] issuing the next tiles' loads,
// [2] scan and end flags, [3] staging and stores, [4] tiles, [5] whole kernel.  tools/xt_phase_probe.py reads them.
static __device__ unsigned long long g_xt_prof[4096 * 8];
#define XT_PF(K) { const unsigned long long pf_t = __builtin_amdgcn_s_memtime(); pf[K] += pf_t - pf_prev; pf_prev = pf_t; }
#else
#define XT_PF(K)
#endif
template <class F, int... I> __device__ __forceinline__ bool xt_unroll_steps(F&& f, std::integer_sequence<int, I...>) { return (f.template operator()<I>() && ...); }

// D = prefetch depth, W = waves per workgroup; EXP selects a timing experiment (wrong results!): 1 = no gathers of u (streams
// only), 2 = loads only (no scan, no stores: what the load side of the pipeline can deliver)
// VB = bytes per STORED matrix value (round 6): sizeof(T), or 2 for an integer matrix all of whose values fit 16 signed bits — the plan then keeps the
// panel-major value plane as int16 (XcdPlan::vbytes) and the lane's four values are one 8-byte load instead of 16 / 32 bytes: the weights of a shortest-path
// problem (1 ... 255 in INT64) cost 2 of the 12 bytes an entry's stream was.
template <class T, class SR, int D = XT_DEPTH, int W = XT_WAVES, int EXP = 0, bool C16 = xt_fmt<T>::C16, int VB = (int)sizeof(T)>
__global__ __launch_bounds__(W * 64, 1) void k_spmv_tiles(const XtCall<T> call, const XtPanel<T>* __restrict__ panels, const SR sr) {
  constexpr int H = xt_hot<T>::H;
  constexpr int NS = D + 3;
  constexpr int XT_WAVES_ = W;
  __shared__ T s_hot[H];
  __shared__ T s_stage[W][64];                          // per wave: sub-row sums on their way out
  __shared__ uint32_t s_next;                           // next dynamic chunk of this workgroup
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  T* const stage = s_stage[wv];
  const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)call.u, (short)0, (int)(call.ulen * (uint32_t)sizeof(T)), 0x00020000);
  // workgroup b works on the column panel(s) of XCD b % 8 — the XCD it is observed to run on: one stream, or (a table per sub-panel,
  // XcdPlan::own) its `sps` streams one after the other, the LDS table refilled in between
  for (uint32_t sp = 0; sp < call.sps; sp++) {
  const XtPanel<T> a = panels[(blockIdx.x & 7) * call.sps + sp];
  __syncthreads();                                      // (every wave is done with the table and the chunk counter of the stream before)
  if (threadIdx.x == 0) s_next = 0;
  const bool use_a = sr.uses_a() && a.aval != nullptr, use_u = sr.uses_u();
  if (use_u) for (uint32_t h = threadIdx.x; h < a.nhot; h += XT_WAVES_ * 64) s_hot[h] = wp_ld(a.xhot + h);      // the table's contents, gathered from u once per call
  constexpr bool ZSLOT = !xt_fmt<T>::C16;               // the plan left the table's last slot free (32-bit entry words are this type's own format, not a measurement variant)
  if constexpr (!C16 && ZSLOT) { if (threadIdx.x == 0) { T z; __builtin_memset(&z, 0, sizeof(T)); s_hot[H - 1] = z; } }      // the zero slot (nhot <= H - 1)
  // (16-bit words: the range covers whole tiles — the plan pads them with zeros — because the range check works on dwords and an
  //  odd entry count would otherwise cut the panel's last word off)
  const __amdgpu_buffer_rsrc_t c_rsrc = C16 ? __builtin_amdgcn_make_buffer_rsrc((void*)a.col16, (short)0, (int)(a.ntiles * (uint32_t)WP_ENT * 2u), 0x00020000)
                                            : __builtin_amdgcn_make_buffer_rsrc((void*)a.pcol, (short)0, (int)(a.nnz * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.extras, (short)0, (int)(C16 ? (a.nextras * 2u + 15u) & ~15ull : 0ull), 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.aval, (short)0, (int)(!a.aval ? 0u : (VB == (int)sizeof(T) ? a.nnz * (uint32_t)sizeof(T) : a.ntiles * (uint32_t)WP_ENT * (uint32_t)VB)), 0x00020000);
  // (narrow plane: the range covers whole tiles — the plan's padding is zeros — because a lane's four values are ONE 8-byte load and the range check would
  //  cut the last lane's live values off with the panel's end; the wide plane's loads are checked dword by dword)
  __syncthreads();
  // Work split (see k_spmv_wavepipe): chunk ids [0, dyn0) are static ranges of s0 chunks dealt to (workgroup, wave), ids >= dyn0
  // are handed out one at a time by the workgroup's LDS counter; workgroup j of the panel owns those congruent to j.
  const uint32_t K = a.tiles_per_chunk, nchunks = (K + a.ntiles - 1) / K;
  const uint32_t nwg = gridDim.x >> 3, jwg = blockIdx.x >> 3;
  uint32_t s0 = (uint32_t)((uint64_t)nchunks * a.static_pct / 100 / (nwg * XT_WAVES_)); if (s0 > WP_MAX_STATIC) s0 = WP_MAX_STATIC;
  const uint32_t dyn0 = s0 * nwg * XT_WAVES_;
  const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane(wv) * nwg + jwg;
  const uint32_t st_step = a.interleave ? nwg * XT_WAVES_ : 1u;
  uint32_t st_next = a.interleave ? wid : wid * s0, st_left = s0;
  auto next_chunk = [&]() __attribute__((always_inline)) -> uint32_t {
    if (st_left) { st_left--; const uint32_t c = st_next; st_next += st_step; return c; }
    uint32_t v = 0; if (lane == 0) v = atomicAdd(&s_next, 1u);
    return dyn0 + (uint32_t)__builtin_amdgcn_readfirstlane(v) * nwg + jwg;
  };
  // the wave's tiles, in the order of its chunks; WP_NONE once the work is exhausted
  uint32_t ic = next_chunk(), ij = 0;
  auto next_tile = [&]() __attribute__((always_inline)) -> uint32_t {
    for (;;) {
      if (ic >= nchunks) return WP_NONE;
      const uint32_t t = ic * K + ij;
      if (ij < K && t < a.ntiles) { ij++; return t; }
      ic = next_chunk(); ij = 0;
    }
  };

  XtStage<T> S[NS];
  static_assert(!C16 || WP_PER == 4, "the 16-bit column plane packs a lane's four words in two dwords");
  auto load_cols = [&](XtStage<T>& s) __attribute__((always_inline)) {
    const bool ok = s.tile != WP_NONE;
    if constexpr (C16) {
      xt_stream_load<uint32_t, 2>(c_rsrc, ok ? (s.tile * (uint32_t)WP_ENT + lane * WP_PER) * 2u : 0xFFFFFFFFu, s.h);
      const uint2 ti = *(const uint2*)(a.tinfo + 2u * (ok ? s.tile : 0u));
      s.rf = ti.x; s.cb = ti.y;
    } else {
      xt_stream_load<uint32_t, WP_PER>(c_rsrc, ok ? (s.tile * (uint32_t)WP_ENT + lane * WP_PER) * 4u : 0xFFFFFFFFu, s.c);
      s.rf = wp_ld(a.trow + (ok ? s.tile : 0u));
    }
  };
  // 16-bit format, one tile behind load_cols: the lane's cold entries are consecutive halfwords of `extras` (entry order =
  // lane-major), starting at the tile's base + the number of cold entries in the lanes before it: three aligned dwords cover
  // any four consecutive halfwords.  Lanes without a cold entry make no request.
  auto load_extras = [&](XtStage<T>& s) __attribute__((always_inline)) {
    if constexpr (C16) {
      const bool ok = s.tile != WP_NONE;
      const uint32_t h0 = s.h[0], h1 = s.h[1];
      uint32_t before = 0; bool any = false;
#define XT_COLD(W) { const bool cold = ok && ((W) & 0x7FFFu) >= (uint32_t)H; const unsigned long long m = __ballot(cold); any = any || cold; \
                     before = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, before)); }
      XT_COLD(h0) XT_COLD(h0 >> 16) XT_COLD(h1) XT_COLD(h1 >> 16)
#undef XT_COLD
      s.xr = (uint32_t)__builtin_amdgcn_readfirstlane(s.cb) + before;
      // (default cache policy, not the streams' nt: a tile's extras end inside a 128-byte line that the next tile's begin in;
      //  with nt that line was fetched from HBM twice — 65 MB of extras traffic per product instead of 30)
      const xt_v3u q = __builtin_amdgcn_raw_buffer_load_b96(x_rsrc, (int)(any ? (s.xr >> 1) * 4u : 0xFFFFFFFFu), 0, 0);
      s.xq[0] = q.x; s.xq[1] = q.y; s.xq[2] = q.z;
    }
  };
  auto issue_gather = [&](XtStage<T>& s) __attribute__((always_inline)) {
    const bool ok = s.tile != WP_NONE;
    const uint32_t e0 = ok ? s.tile * (uint32_t)WP_ENT : 0u, left = a.nnz - e0, cnt = !ok ? 0u : (left < (uint32_t)WP_ENT ? left : (uint32_t)WP_ENT);
    if (use_a) {
      if constexpr (VB == (int)sizeof(T)) xt_stream_load<T, WP_PER>(v_rsrc, ok ? (e0 + lane * WP_PER) * (uint32_t)sizeof(T) : 0xFFFFFFFFu, s.v);
      else {
        static_assert(VB == 2 && WP_PER == 4, "the narrow value plane holds four int16 per lane: one 8-byte load");
        uint32_t q[2];
        xt_stream_load<uint32_t, 2>(v_rsrc, ok ? (e0 + lane * WP_PER) * 2u : 0xFFFFFFFFu, q);
        if constexpr (std::is_integral<T>::value) {
          s.v[0] = (T)(int16_t)(q[0] & 0xFFFFu); s.v[1] = (T)(int16_t)(q[0] >> 16); s.v[2] = (T)(int16_t)(q[1] & 0xFFFFu); s.v[3] = (T)(int16_t)(q[1] >> 16);
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < WP_PER; u++) s.v[u] = T();
    }
    if constexpr (C16) {
      // the lane's extras as a window of four halfwords (two overlapping 64-bit views of the three dwords, shifted by the
      // parity of its first one); every cold entry takes the lowest and shifts the window on.  Entries behind the end of the
      // panel are zero words — slot 0 of the table — and fetch nothing.
      const uint32_t sh = (s.xr & 1u) << 4;
      uint32_t wl = __builtin_amdgcn_alignbit(s.xq[1], s.xq[0], sh), wh = __builtin_amdgcn_alignbit(s.xq[2], s.xq[1], sh);
#pragma unroll
      for (int u = 0; u < WP_PER; u++) {
        const uint32_t code = (s.h[u >> 1] >> (16 * (u & 1))) & 0x7FFFu;
        const bool cold = code >= (uint32_t)H;
        const uint32_t col = ((code - (uint32_t)H) << 16) | (wl & 0xFFFFu);
        s.g[u] = (use_u && EXP != 1) ? xt_buf_load<T>(u_rsrc, cold ? col * (uint32_t)sizeof(T) : 0xFFFFFFFFu) : T();
        const uint32_t adv = cold ? 16u : 0u;
        wl = __builtin_amdgcn_alignbit(wh, wl, adv); wh >>= adv;
      }
    } else {
      // (entries behind the end of the panel are zero words — the plan pads with zeros, the descriptor's range ends with the panel — i.e. slot 0
      //  of the table: they fetch nothing and are never looked at)
#pragma unroll
      for (int u = 0; u < WP_PER; u++) {
        const uint32_t w = s.c[u];
        const uint32_t off = (w & (XT_COLD | XT_IDXMASK)) * (uint32_t)sizeof(T);                  // (4-byte values: w << 2)
        if constexpr (ZSLOT) {
          s.g[u] = (use_u && EXP != 1) ? xt_buf_load<T>(u_rsrc, off ^ (XT_COLD * (uint32_t)sizeof(T))) : T();      // only the columns the table does not hold are fetched (the others are out of range: zero bits)
          const uint32_t last = (uint32_t)(H - 1) * (uint32_t)sizeof(T);
          s.hl[u] = use_u ? *(const T*)((const char*)s_hot + (off < last ? off : last)) : T();
        } else {
          const bool cold = (w & XT_COLD) != 0;
          s.g[u] = (use_u && EXP != 1) ? xt_buf_load<T>(u_rsrc, cold ? (w & XT_IDXMASK) * (uint32_t)sizeof(T) : 0xFFFFFFFFu) : T();
          s.hl[u] = T();
        }
      }
    }
  };

#ifdef XT_PROFILE
  unsigned long long pf[6] = {0, 0, 0, 0, 0, 0}; const unsigned long long pf_start = __builtin_amdgcn_s_memtime(); unsigned long long pf_prev = pf_start;
#endif
  T carry = sr.identity; bool carry_has = false;        // partial of the sub-row the current tile starts in (wave-uniform)
#pragma unroll
  for (int d = 0; d < D + 2; d++) { S[d].tile = next_tile(); load_cols(S[d]); }
#pragma unroll
  for (int d = 0; d < D + 1; d++) load_extras(S[d]);
#pragma unroll
  for (int d = 0; d < D; d++) issue_gather(S[d]);

  // one tile: slot I is reduced while the values and gathers of the tile D ahead and the column words of the tile D + 2
  // ahead are issued; the loop is unrolled over the ring so that the register sets swap roles without being copied
  auto step = [&]<int I>() __attribute__((always_inline)) -> bool {
    XtStage<T>& A = S[I % NS]; XtStage<T>& N1 = S[(I + 1) % NS]; XtStage<T>& G = S[(I + D) % NS]; XtStage<T>& M = S[(I + D + 1) % NS]; XtStage<T>& C = S[(I + D + 2) % NS];
    if (A.tile == WP_NONE) return false;
#ifdef XT_PROFILE
    pf_prev = __builtin_amdgcn_s_memtime(); pf[4]++;
#endif
    const uint32_t t = A.tile;
    const uint32_t e0 = t * (uint32_t)WP_ENT, cnt = a.nnz - e0 < (uint32_t)WP_ENT ? a.nnz - e0 : (uint32_t)WP_ENT;
    if constexpr (D == 0) issue_gather(A);
    // products (LDS table for the panel's hottest columns)
    T p[WP_PER];
#pragma unroll
    for (int u = 0; u < WP_PER; u++) {
      T uvv;
      if constexpr (C16) {
        const uint32_t cc = (A.h[u >> 1] >> (16 * (u & 1))) & 0x7FFFu;
        uvv = use_u ? (cc < (uint32_t)H ? s_hot[cc < (uint32_t)H ? cc : 0] : A.g[u]) : T();
      } else if constexpr (ZSLOT) uvv = use_u ? xt_or_bits<T>(A.g[u], A.hl[u]) : T();   // one of the two is zero bits (see XT_COLD)
      else { const uint32_t w = A.c[u]; uvv = use_u ? ((w & XT_COLD) ? A.g[u] : s_hot[w & XT_IDXMASK]) : T(); }      // (measurement variant: 32-bit words for an 8-byte type)
      p[u] = sr.mult(A.v[u], uvv);                             // entries past cnt hold junk: a forward scan never lets it reach a live position
    }
#ifdef XT_PROFILE
    { T sink = p[0]; for (int u = 1; u < WP_PER; u++) sink = sr.add(sink, p[u]); asm volatile("" :: "v"(sink)); }     // the products must exist here
    XT_PF(0)
#endif
    if constexpr (D > 0) issue_gather(G);
    load_extras(M);
    C.tile = next_tile(); load_cols(C);
    XT_PF(1)
    const uint32_t rf = (uint32_t)__builtin_amdgcn_readfirstlane(A.rf);
    if constexpr (EXP == 2) {                                  // timing experiment: consume the loads, nothing else
      T q = sr.add(sr.add(p[0], p[1]), sr.add(p[2], p[3]));
      if ((C16 ? A.h[0] : A.c[0]) == 0x12345678u && rf == 0x7FFFFFFFu) wp_st(call.partial + lane, q);
      return true;
    }
    bool rs[WP_PER];                                           // my entry u is the first of its sub-row
#pragma unroll
    for (int u = 0; u < WP_PER; u++) rs[u] = C16 ? ((A.h[u >> 1] >> (16 * (u & 1) + 15)) & 1u) != 0u : (int32_t)A.c[u] < 0;
    // what follows the tile's last entry: a row start (always, at the end of a chunk), or the end of the work?
    const bool last_end = N1.tile == WP_NONE || (C16 ? (__builtin_amdgcn_readfirstlane(N1.h[0]) & 0x8000u) != 0u : (int32_t)__builtin_amdgcn_readfirstlane(N1.c[0]) < 0);
    // ---- segmented inclusive scan of the products in entry order, and in the same wave scan the number of row starts
    // behind the tile's first entry (sub-row of an entry = rf + that count, up to and including the entry)
    // row starts behind the tile's first entry, in the lanes below mine (entry order = lane-major): ballots + v_mbcnt; the total is scalar
    uint32_t excl = 0, starts = 0;
    if constexpr (sizeof(T) == 4 || sizeof(T) == 8) {
#pragma unroll
      for (int u = 0; u < WP_PER; u++) {
        unsigned long long mk = __ballot(rs[u]); if (u == 0) mk &= ~1ull;
        excl = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, excl));
        starts += (uint32_t)__popcll(mk);
      }
    }
    uint32_t mine = 0;
#pragma unroll
    for (int u = 0; u < WP_PER; u++) mine += (rs[u] && (u > 0 || lane > 0)) ? 1u : 0u;
    uint32_t incl;
    {
      bool st0 = rs[0] || (lane == 0 && !carry_has);           // nothing carried in: entry 0 starts a segment whatever it is
      T agg = p[0]; bool anyf = st0;
#pragma unroll
      for (int u = 1; u < WP_PER; u++) { agg = rs[u] ? p[u] : sr.add(agg, p[u]); anyf = anyf || rs[u]; }
      if (lane == 0 && !anyf) agg = sr.add(carry, agg);        // the carried partial flows through lane 0
      T v = agg;
      if constexpr (sizeof(T) == 4 || sizeof(T) == 8) {
        xt_seg_scan_masked<T, SR>(v, __ballot(anyf), sr);
        incl = excl + mine;
      } else {
        uint32_t x = (anyf ? 0x80000000u : 0u) | mine;
        xt_seg_scan_count<T, SR>(v, x, lane, sr);
        incl = x & 0x7FFFFFFFu;
        starts = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      }
      // what flows into my first entry (unused when it starts a row): the scanned value of the lane below — one DPP wave shift
      // (wave_shr:1; gfx9 has it) instead of a ds_bpermute through the LDS crossbar and the wait for it; lane 0 takes the carry
      T run;
      if constexpr (sizeof(T) == 4 || sizeof(T) == 8) run = xt_wave_shr1<T>(v, carry); else { run = shfl_up_t<T>(v, 1); if (lane == 0) run = carry; }
      run = st0 ? p[0] : sr.add(run, p[0]); p[0] = run;
#pragma unroll
      for (int u = 1; u < WP_PER; u++) { run = rs[u] ? p[u] : sr.add(run, p[u]); p[u] = run; }
    }
    // ---- an entry ends its sub-row when the next entry starts one
    const int nxt0 = (int)__builtin_amdgcn_update_dpp(0, (int)rs[0], 0x130 /* wave_shl:1 */, 0xf, 0xf, false);   // first flag of the next lane (lane 63: 0, never used — pos + 1 == cnt there)
    bool end[WP_PER];
    if (cnt == (uint32_t)WP_ENT) {                             // a full tile (wave-uniform; all but a panel's last): only its last entry looks beyond it
#pragma unroll
      for (int u = 0; u + 1 < WP_PER; u++) end[u] = rs[u + 1];
      end[WP_PER - 1] = lane == 63 ? last_end : nxt0 != 0;
    } else {
#pragma unroll
      for (int u = 0; u < WP_PER; u++) {
        const uint32_t pos = (uint32_t)(lane * WP_PER + u);
        const bool nx = u + 1 < WP_PER ? rs[u + 1 < WP_PER ? u + 1 : u] : nxt0 != 0;
        end[u] = pos + 1 < cnt ? nx : (pos + 1 == cnt ? last_end : false);
      }
    }
    // ---- the sums of the sub-rows that end in this tile leave through the wave's staging slots: the ends are ranked by
    // sub-row (rf, rf+1, ... — consecutive), so 64 of them at a time become one coalesced store.  (Storing from the
    // owning lanes, 8 scattered bytes per sub-row in four sparse store instructions, cost 25-30 us per product.)
    const uint32_t nends = starts + (last_end ? 1u : 0u);             // starts: row starts behind the tile's first entry
#ifdef XT_PROFILE
    { asm volatile("" :: "v"(p[WP_PER - 1])); }
    XT_PF(2)
#endif
    for (uint32_t base = 0; base < nends; base += 64) {
      uint32_t row = incl - mine;
#pragma unroll
      for (int u = 0; u < WP_PER; u++) {
        row += (rs[u] && (u > 0 || lane > 0)) ? 1u : 0u;
        const uint32_t k = row - base;
        if (end[u] && k < 64u) stage[k] = p[u];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
      const uint32_t k = base + (uint32_t)lane;
      if (k < nends) wp_st(call.partial + rf + k, stage[lane]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();   // the slots are free again
    }
    if (cnt == (uint32_t)WP_ENT && !last_end) { carry = xt_readlane<T>(p[WP_PER - 1], 63); carry_has = true; }
    else { carry = sr.identity; carry_has = false; }
    XT_PF(3)
    return true;
  };
  while (xt_unroll_steps(step, std::make_integer_sequence<int, NS>{})) {}
#ifdef XT_PROFILE
  if (lane == 0) { pf[5] = __builtin_amdgcn_s_memtime() - pf_start; for (int k = 0; k < 6; k++) g_xt_prof[((size_t)blockIdx.x * W + wv) * 8 + k] = pf[k]; }
#endif
  }     // streams of this XCD
}

}  // namespace grb
