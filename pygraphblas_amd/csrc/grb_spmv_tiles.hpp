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
// Work split, carry records and fix-up are W's (chunks of tiles; wp_* helpers).
#pragma once
#include "grb_spmv_wavepipe.hpp"

namespace grb {

template <class T> struct xt_hot { static constexpr int H = (WP_LDS_BYTES - 16 - WP_WAVES * 64 * (int)sizeof(T)) / (int)sizeof(T); };   // all of the LDS but the staging slots is the table: 19454 (8 B) / 39932 (4 B)

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

// the two streams of a tile through buffer descriptors; lanes behind the end of the panel read zeros.  XT_STREAM_AUX = the
// cache-policy bits of these loads (1 sc0, 2 nt, 16 sc1).  Measured on R-MAT-22 FP64 (ms per mxv, stream-only variant in
// brackets): 0: 0.300 (0.226), sc0: 0.300 (0.227), nt: 0.292 (0.212), sc1: 0.318 (0.240), sc0+sc1+nt: 0.293 (0.213).
#ifndef XT_STREAM_AUX
#define XT_STREAM_AUX 2
#endif
typedef uint32_t xt_v4u __attribute__((ext_vector_type(4)));
template <class E, int N> __device__ __forceinline__ void xt_stream_load(__amdgpu_buffer_rsrc_t r, uint32_t first, E (&out)[N]) {
  static_assert((sizeof(E) * N) % 16 == 0 || sizeof(E) * N == 4 || sizeof(E) * N == 8, "tile slice per lane");
  if constexpr ((sizeof(E) * N) % 16 == 0) {
    xt_v4u tmp[sizeof(E) * N / 16];
#pragma unroll
    for (int j = 0; j < (int)(sizeof(E) * N / 16); j++) tmp[j] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(first * (uint32_t)sizeof(E) + 16u * j), 0, XT_STREAM_AUX);
    __builtin_memcpy(&out[0], &tmp[0], sizeof(E) * N);
  } else if constexpr (sizeof(E) * N == 8) {
    xt_v2u tmp = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(first * (uint32_t)sizeof(E)), 0, XT_STREAM_AUX); __builtin_memcpy(&out[0], &tmp, 8);
  } else {
    uint32_t tmp = __builtin_amdgcn_raw_buffer_load_b32(r, (int)(first * (uint32_t)sizeof(E)), 0, XT_STREAM_AUX); __builtin_memcpy(&out[0], &tmp, 4);
  }
}

template <class T> struct XtStage { uint32_t c[WP_PER]; T v[WP_PER], g[WP_PER]; uint32_t rf; };   // what one tile has in flight (rf: sub-row of its first entry)

// a.trow = first sub-row of every tile [ntiles + 1]; a.ntasks = tiles; a.rowptr / a.tent / a.ypres are not used
template <class T, class SR>
__global__ __launch_bounds__(WP_WAVES * 64, WP_WGS_PER_CU * WP_WAVES / 4) void k_spmv_tiles(const WpArgs<T> a0, const WpArgs<T>* __restrict__ panels, const SR sr) {
  const WpArgs<T> a = panels[blockIdx.x & 7];          // workgroup b works on column panel b % 8 — the XCD it is observed to run on
  constexpr int H = xt_hot<T>::H;
  __shared__ T s_hot[H];
  __shared__ T s_stage[WP_WAVES][64];                   // per wave: sub-row sums on their way out
  __shared__ uint32_t s_next;                           // next dynamic chunk of this workgroup
  if (threadIdx.x == 0) s_next = 0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  T* const stage = s_stage[wv];
  const bool use_a = sr.uses_a(), use_u = sr.uses_u();
  // u comes with the launch (a0.xorig, a0.nrows = its length), the rest of `a` is the plan's
  const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a0.xorig, (short)0, (int)(a0.nrows * (uint32_t)sizeof(T)), 0x00020000);
  if (use_u) for (uint32_t h = threadIdx.x; h < a.nhot; h += WP_WAVES * 64) s_hot[h] = wp_ld(a.x + h);      // the table's contents, gathered from u once per call
  const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.pcol, (short)0, (int)(a.nnz * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.aval, (short)0, (int)(a.nnz * (uint32_t)sizeof(T)), 0x00020000);
  __syncthreads();
#ifdef WP_PROFILE
  const unsigned long long pf_t0 = WP_CLK();
#endif
  // work split: see k_spmv_wavepipe
  const uint32_t K = a.tasks_per_chunk, nchunks = (K + a.ntasks - 1) / K;
  const uint32_t nwg = gridDim.x >> 3, jwg = blockIdx.x >> 3;
  uint32_t s0 = (uint32_t)((uint64_t)nchunks * a.static_pct / 100 / (nwg * WP_WAVES)); if (s0 > WP_MAX_STATIC) s0 = WP_MAX_STATIC;
  const uint32_t dyn0 = s0 * nwg * WP_WAVES;
  auto grab = [&]() { uint32_t v = 0; if (lane == 0) v = atomicAdd(&s_next, 1u); return v; };
  uint32_t rec = ((uint32_t)__builtin_amdgcn_readfirstlane(wv) * nwg + jwg) * s0, nrec = s0;
  if (nrec == 0) { rec = dyn0 + (uint32_t)__builtin_amdgcn_readfirstlane(grab()) * nwg + jwg; nrec = 1; }

  auto load_cols = [&](uint32_t t, uint32_t (&c)[WP_PER], uint32_t& rf) {
    const uint32_t e0 = t * (uint32_t)WP_ENT;
    xt_stream_load<uint32_t, WP_PER>(c_rsrc, e0 + lane * WP_PER, c);
    rf = wp_ld(a.trow + t);
  };
  auto issue_gather = [&](uint32_t t, const uint32_t (&c)[WP_PER], T (&v)[WP_PER], T (&g)[WP_PER]) {
    const uint32_t e0 = t * (uint32_t)WP_ENT, cnt = a.nnz - e0 < (uint32_t)WP_ENT ? a.nnz - e0 : (uint32_t)WP_ENT;
    if (use_a) xt_stream_load<T, WP_PER>(v_rsrc, e0 + lane * WP_PER, v);
    else {
#pragma unroll
      for (int u = 0; u < WP_PER; u++) v[u] = T();
    }
#pragma unroll
    for (int u = 0; u < WP_PER; u++) {
      const uint32_t cc = (uint32_t)(lane * WP_PER + u) < cnt ? (c[u] & WP_COLMASK) : 0u;   // slot in the LDS table, or H + column
      g[u] = use_u ? xt_buf_load<T>(u_rsrc, cc >= (uint32_t)H ? (cc - (uint32_t)H) * (uint32_t)sizeof(T) : 0xFFFFFFFFu) : T();   // only the columns the table does not hold are fetched
    }
  };

  while (rec < nchunks) {
    const uint32_t t0 = rec * K;
    uint32_t t1 = t0 + nrec * K; if (t1 > a.ntasks) t1 = a.ntasks;
    WpCarry<T> cr; cr.head_row = cr.tail_row = WP_NONE; cr.head_has = cr.head_done = cr.tail_has = cr.pad = 0; cr.head_val = cr.tail_val = sr.identity;
    T carry = sr.identity; bool carry_has = false, owned = true;        // partial of the row the current tile starts in (wave-uniform)
    uint32_t last_row = 0;
    // does a row start right behind the range?  (asked now, needed at its last tile)
    const uint32_t e_end = t1 * (uint32_t)WP_ENT;
    const uint32_t behind_w = wp_ld(a.pcol + (e_end < a.nnz ? e_end : a.nnz - 1));
    const uint32_t next_raw = grab();                                   // and the next chunk

    XtStage<T> S0, S1, S2;
    load_cols(t0, S0.c, S0.rf);
    if (t0 + 1 < t1) load_cols(t0 + 1, S1.c, S1.rf);
    issue_gather(t0, S0.c, S0.v, S0.g);
    // one tile: A is reduced while the values and gathers of B and the column words of C are issued; unrolled three
    // times so that the register sets swap roles without being copied
    auto step = [&](uint32_t t, XtStage<T>& A, XtStage<T>& B, XtStage<T>& C) __attribute__((always_inline)) {
      const uint32_t e0 = t * (uint32_t)WP_ENT, cnt = a.nnz - e0 < (uint32_t)WP_ENT ? a.nnz - e0 : (uint32_t)WP_ENT;
      const bool more = t + 1 < t1, more2 = t + 2 < t1;
      // products (LDS table for the panel's hottest columns)
      T p[WP_PER];
#pragma unroll
      for (int u = 0; u < WP_PER; u++) {
        const uint32_t cc = A.c[u] & WP_COLMASK;
        const T uvv = use_u ? (cc < (uint32_t)H ? s_hot[cc < (uint32_t)H ? cc : 0] : A.g[u]) : T();
        p[u] = sr.mult(A.v[u], uvv);                             // entries past cnt hold junk: a forward scan never lets it reach a live position
      }
      if (more) issue_gather(t + 1, B.c, B.v, B.g);
      if (more2) load_cols(t + 2, C.c, C.rf);
      const uint32_t rf = (uint32_t)__builtin_amdgcn_readfirstlane(A.rf);
      bool rs[WP_PER];                                           // my entry u is the first of its sub-row
#pragma unroll
      for (int u = 0; u < WP_PER; u++) rs[u] = (int32_t)A.c[u] < 0;
      if (t == t0) owned = (int32_t)__builtin_amdgcn_readfirstlane(A.c[0]) < 0;     // does the range begin with a row start?
      // what follows the tile's last entry: a row start, or the end of the panel?
      const uint32_t nextw = more ? (uint32_t)__builtin_amdgcn_readfirstlane(B.c[0]) : (uint32_t)__builtin_amdgcn_readfirstlane(behind_w);
      const bool last_end = e0 + cnt >= a.nnz || (int32_t)nextw < 0;
      // ---- segmented inclusive scan of the products in entry order, and in the same wave scan the number of row starts
      // behind the tile's first entry (sub-row of an entry = rf + that count, up to and including the entry)
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
        T v = agg; uint32_t x = (anyf ? 0x80000000u : 0u) | mine;
        xt_seg_scan_count<T, SR>(v, x, lane, sr);
        incl = x & 0x7FFFFFFFu;
        T run = shfl_up_t<T>(v, 1); if (lane == 0) run = carry;  // what flows into my first entry (unused when it starts a row)
        run = st0 ? p[0] : sr.add(run, p[0]); p[0] = run;
#pragma unroll
        for (int u = 1; u < WP_PER; u++) { run = rs[u] ? p[u] : sr.add(run, p[u]); p[u] = run; }
      }
      // ---- an entry ends its sub-row when the next entry starts one
      const int nxt0 = __shfl_down((int)rs[0], 1, 64);           // first flag of the next lane
      bool end[WP_PER];
#pragma unroll
      for (int u = 0; u < WP_PER; u++) {
        const uint32_t pos = (uint32_t)(lane * WP_PER + u);
        const bool nx = u + 1 < WP_PER ? rs[u + 1 < WP_PER ? u + 1 : u] : nxt0 != 0;
        end[u] = pos + 1 < cnt ? nx : (pos + 1 == cnt ? last_end : false);
      }
      // ---- the sums of the sub-rows that end in this tile leave through the wave's staging slots: the ends are ranked by
      // sub-row (rf, rf+1, ... — consecutive), so 64 of them at a time become one coalesced store.  (Storing from the
      // owning lanes, 8 scattered bytes per sub-row in four sparse store instructions, cost 25-30 us per product.)
      const uint32_t starts = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);      // row starts behind the tile's first entry
      const uint32_t nends = starts + (last_end ? 1u : 0u);
      bool skip_first = false;
      for (uint32_t base = 0; base < nends; base += 64) {
        uint32_t row = incl - mine;
#pragma unroll
        for (int u = 0; u < WP_PER; u++) {
          row += (rs[u] && (u > 0 || lane > 0)) ? 1u : 0u;
          const uint32_t k = row - base;
          if (end[u] && k < 64u) stage[k] = p[u];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
        if (base == 0 && !owned) {                               // the first row end of the range closes a row that began in another range
          cr.head_row = rf; cr.head_val = wp_uniform(stage[0]); cr.head_has = 1; cr.head_done = 1;
          owned = true; skip_first = true;
        }
        const uint32_t k = base + (uint32_t)lane;
        if (k < nends && !(skip_first && k == 0)) wp_st(a.y + rf + k, stage[lane]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();   // the slots are free again
      }
      last_row = rf + starts;
      if (cnt == (uint32_t)WP_ENT && !last_end) { carry = xt_readlane<T>(p[WP_PER - 1], 63); carry_has = true; }
      else { carry = sr.identity; carry_has = false; }
    };
    for (uint32_t t = t0; t < t1; t += 3) {
      step(t, S0, S1, S2);
      if (t + 1 < t1) step(t + 1, S1, S2, S0);
      if (t + 2 < t1) step(t + 2, S2, S0, S1);
    }
    // the row the range ends in, if it ends strictly inside it
    if (carry_has || !owned) {
      if (owned) { cr.tail_row = last_row; cr.tail_val = carry; cr.tail_has = carry_has; }
      else { cr.head_row = last_row; cr.head_val = carry; cr.head_has = carry_has; cr.head_done = 0; }   // the whole range lies inside one row
    }
    if (lane == 0) wp_st_carry(a.carry + rec, cr);
    if (lane > 0 && (uint32_t)lane < nrec) {       // the other chunk ids of a static range: empty records
      WpCarry<T> e; e.head_row = e.tail_row = WP_NONE; e.head_has = e.head_done = e.tail_has = e.pad = 0; e.head_val = e.tail_val = sr.identity;
      wp_st_carry(a.carry + rec + lane, e);
    }
    rec = dyn0 + (uint32_t)__builtin_amdgcn_readfirstlane(next_raw) * nwg + jwg; nrec = 1;
  }
#ifdef WP_PROFILE
  if (lane == 0) { const unsigned long long t = WP_CLK(); g_wp_prof[blockIdx.x * WP_WAVES + wv] = t - pf_t0; g_wp_prof[4096 + blockIdx.x * WP_WAVES + wv] = 1ull; }
#endif
}

// first sub-row of every tile: the largest s with rowptr[s] <= 256 t (sub-rows are never empty)
static __global__ void k_xt_tile_rows(const uint32_t* __restrict__ rowptr, uint32_t nsub, uint32_t ntiles, uint32_t* __restrict__ trow) {
  for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t <= ntiles; t += gridDim.x * 256) {
    const unsigned long long e = (unsigned long long)t * WP_ENT;
    uint32_t lo = 0, hi = nsub;                          // rowptr[nsub] = entries of the panel
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (rowptr[mid] <= e) lo = mid; else hi = mid - 1; }
    trow[t] = lo < nsub ? lo : (nsub ? nsub - 1 : 0);
  }
}

}  // namespace grb
