// grb_spmv_sell.hpp — round 6: the "lane per piece" layout of kernel X's panel streams and the kernel that walks it.
//
// Why.  The tile pipeline (grb_spmv_tiles.hpp) spends 160-280 VALU instructions per 256 entries on finding out which sub-row an
// entry belongs to (row-start flags, a segmented DPP scan, ranks, staging) and keeps so much of that state in registers that only
// ONE tile of values and gathers per wave can be in flight (profiles/r03_spmv_pipeline_analysis.txt: the pattern-only FP32 product
// is bound by instruction issue, the FP64 one by the latency its depth cannot cover).  Here the plan does that work once:
//   * every sub-row (the entries of one row in one panel) is cut into PIECES: at most SELL_CAP entries — one lane adds them one
//     after the other — or, for a sub-row of >= SELL_LONG entries, slices of up to 64 x SELL_CAP entries that a whole wave adds
//     (a lane takes a contiguous run, the wave reduces at the end).  A piece is a sub-row to the rest of the plan: it has its own
//     partial sum, and the merge kernels add the partials of a row as before (grb_spmv_xcd.hpp);
//   * inside windows of SELL_W consecutive pieces of a stream the one-lane pieces are sorted by length (descending in even windows,
//     ascending in odd ones, so that neighbours across a window boundary are alike) and dealt 64 at a time to CHUNKS: lane l of a
//     chunk owns one piece, step t of the chunk holds entry t of every lane's piece — 64 column words (and 64 values) that one load
//     instruction of the wave fetches, padded to the chunk's longest piece (R-MAT-22: 3 % of padding).  A wave slice is a chunk of
//     its own.  The windows keep a chunk's 64 partial sums within 32 KB of one another in the partial array;
//   * what the kernel must know about a step rides in the column words: bit 31 of lane l's word = "lane l has an entry at this
//     step", bit 30 of lane l's word = bit l of a 64-bit record {last step of the chunk, wave slice, first step, chunk number} —
//     one v_cmp each turns them into scalar masks.  No row pointers, no flags to scan, no per-chunk descriptors to wait for.
// The kernel is then a stream of steps: column words D1 steps ahead, gathers / LDS reads / values D2 steps ahead, ~10 VALU
// instructions per 64 entries, and a chunk's sums leave from the lanes that hold them (one store per piece, scattered inside the
// window: the L2 merges them).  Sums are formed in a fixed order => reproducible.
#pragma once

namespace grb {

#ifndef SELL_CAP_V
#define SELL_CAP_V 32
#endif
#ifndef SELL_D1
#define SELL_D1 12                    // column words this many steps ahead of the accumulation
#endif
#ifndef SELL_D2
#define SELL_D2 6                     // gathers, LDS reads, values and the partial ids this many
#endif
constexpr uint32_t SELL_CAP = SELL_CAP_V, SELL_WCAP = 64u * SELL_CAP, SELL_LONG = 256, SELL_W = 4096, SELL_ITEM = 32;
constexpr uint32_t SW_VALID = 0x80000000u, SW_META = 0x40000000u;      // (bit 29 = XT_COLD, bits 28..0 = slot or column, as in the 32-bit entry words)
constexpr uint32_t SM_LAST = 1u, SM_WAVE = 2u, SM_FIRST = 4u;           // low bits of a step's record; the chunk number sits above bit 8
static_assert(SELL_LONG > SELL_CAP && SELL_CAP < 256, "a one-lane piece is shorter than a wave slice; its length is a sort key of 8 bits");

struct SellStreams { uint32_t ns; uint32_t tbase[XPMAX + 1]; unsigned long long ne[XPMAX]; };      // tiles before stream k, entries of stream k
struct SellGroups { uint32_t ns; uint32_t g1[XPMAX], g0[XPMAX], gend[XPMAX]; uint32_t nc1[XPMAX], cbase[XPMAX + 1]; };   // sorted positions: one-lane pieces [g1, g0), wave slices [g0, gend); chunks
template <class T> struct SellPanel {
  const uint32_t* scol; const T* sval; const uint32_t* item_first; const T* xhot;      // the stream's steps (64 words each), its work items (first step of each, relative to the matrix's first step)
  uint32_t step0, nsteps, nitems, nhot, static_pct, interleave;
};

__device__ __forceinline__ uint32_t sell_stream_of_tile(const SellStreams& st, uint32_t g) { uint32_t k = 0; for (uint32_t j = 1; j < st.ns; j++) k = g >= st.tbase[j] ? j : k; return k; }

// piece s starts at the s-th flagged entry (the numbering of k_xp_subrows)
static __global__ __launch_bounds__(256) void k_sell_pstart(const uint32_t* __restrict__ pcol, const uint32_t* __restrict__ E, uint32_t ntiles, uint32_t* __restrict__ pstart) {
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6); g < ntiles; g += gridDim.x * 4) {
    const size_t q0 = (size_t)g * WP_ENT + lane * 4;
    const uint4 wd = *(const uint4*)(pcol + q0);
    const uint32_t f[4] = {wd.x >> 31, wd.y >> 31, wd.z >> 31, wd.w >> 31};
    const uint32_t mine = f[0] + f[1] + f[2] + f[3];
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if ((int)lane >= d) incl += o; }
    uint32_t s = E[g] + incl - mine;
#pragma unroll
    for (int j = 0; j < 4; j++) if (f[j]) { pstart[s] = (uint32_t)(q0 + j); s++; }
  }
}
static __global__ void k_sell_plen(const uint32_t* __restrict__ pstart, uint64_t F, const SellStreams st, uint32_t* __restrict__ plen) {
  for (uint64_t s = blockIdx.x * 256ull + threadIdx.x; s < F; s += gridDim.x * 256ull) {
    const uint32_t q = pstart[s], k = sell_stream_of_tile(st, q / WP_ENT);
    const unsigned long long send = (unsigned long long)st.tbase[k] * WP_ENT + st.ne[k];
    const unsigned long long nx = s + 1 < F ? pstart[s + 1] : ~0ull;
    plen[s] = (uint32_t)((nx < send ? nx : send) - q);
  }
}
// cut the sub-rows into pieces: more row-start flags (and the row of every new one)
static __global__ __launch_bounds__(256) void k_sell_cap(uint32_t* __restrict__ pcol, uint32_t* __restrict__ rowtmp, const uint32_t* __restrict__ E0, const uint32_t* __restrict__ pstart0,
                                                         const uint32_t* __restrict__ plen0, const SellStreams st, uint32_t ntiles) {
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6); g < ntiles; g += gridDim.x * 4) {
    const uint32_t k = sell_stream_of_tile(st, g);
    const unsigned long long send = (unsigned long long)st.tbase[k] * WP_ENT + st.ne[k];
    const size_t q0 = (size_t)g * WP_ENT + lane * 4;
    const uint4 wd = *(const uint4*)(pcol + q0);
    const uint32_t w[4] = {wd.x, wd.y, wd.z, wd.w};
    const uint32_t mine = (w[0] >> 31) + (w[1] >> 31) + (w[2] >> 31) + (w[3] >> 31);
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if ((int)lane >= d) incl += o; }
    uint32_t cnt = E0[g] + incl - mine;            // flagged entries before my first one
#pragma unroll
    for (int j = 0; j < 4; j++) {
      cnt += w[j] >> 31;
      const unsigned long long q = q0 + j;
      if (q >= send || (w[j] >> 31) || cnt == 0) continue;
      const uint32_t id = cnt - 1, ps = pstart0[id], o = (uint32_t)q - ps, L = plen0[id];
      bool cut;
      if (L < SELL_LONG) cut = o % SELL_CAP == 0;
      else {
        const uint32_t tail0 = L / SELL_WCAP * SELL_WCAP, tail = L - tail0;
        if (o < tail0) cut = o % SELL_WCAP == 0;
        else cut = tail >= SELL_LONG ? o == tail0 : (o - tail0) % SELL_CAP == 0;
      }
      if (cut) { pcol[q] = w[j] | WP_ROWSTART; rowtmp[q] = rowtmp[ps]; }
    }
  }
}
// sort key of a piece: stream | wave slice? | window | length order inside the window
static __global__ void k_sell_keys(const uint32_t* __restrict__ pstart, const uint32_t* __restrict__ plen, uint64_t F, const SellStreams st, const uint32_t* __restrict__ vfirst, uint32_t vps,
                                   unsigned long long* __restrict__ key, uint32_t* __restrict__ id) {
  for (uint64_t s = blockIdx.x * 256ull + threadIdx.x; s < F; s += gridDim.x * 256ull) {
    const uint32_t k = sell_stream_of_tile(st, pstart[s] / WP_ENT), L = plen[s];
    const uint32_t win = ((uint32_t)s - vfirst[k * vps]) / SELL_W;
    const bool wave = L >= SELL_LONG;
    const uint32_t lk = wave ? 0u : ((win & 1u) ? L : 255u - L);
    key[s] = ((unsigned long long)k << 40) | ((unsigned long long)(wave ? 1 : 0) << 39) | ((unsigned long long)win << 8) | lk;
    id[s] = (uint32_t)s;
  }
}
static __global__ void k_sell_groups(const unsigned long long* __restrict__ key, uint64_t F, uint32_t* __restrict__ gstart) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < F; i += gridDim.x * 256ull) {
    const uint32_t g = (uint32_t)(key[i] >> 39);
    if (i == 0 || (uint32_t)(key[i - 1] >> 39) != g) gstart[g] = (uint32_t)i;
  }
}
static __global__ void k_sell_inv(const uint32_t* __restrict__ sorted, uint64_t F, uint32_t* __restrict__ inv) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < F; i += gridDim.x * 256ull) inv[sorted[i]] = (uint32_t)i;
}
__device__ __forceinline__ uint32_t sell_stream_of_chunk(const SellGroups& gr, uint32_t c) { uint32_t k = 0; for (uint32_t j = 1; j < gr.ns; j++) k = c >= gr.cbase[j] ? j : k; return k; }
// steps of every chunk: its longest piece, or a wave slice's length over 64 lanes  (one wave per chunk)
static __global__ __launch_bounds__(256) void k_sell_lc(const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ plen, const SellGroups gr, uint32_t nchunks, uint32_t* __restrict__ lc) {
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6); c < nchunks; c += gridDim.x * 4) {
    const uint32_t k = sell_stream_of_chunk(gr, c), j = c - gr.cbase[k];
    uint32_t L;
    if (j < gr.nc1[k]) {
      const uint32_t s = gr.g1[k] + 64u * j + lane;
      const uint32_t l = s < gr.g0[k] ? plen[sorted[s]] : 0u;
      L = __builtin_amdgcn_wave_reduce_max_u32(l, 0);
    } else L = (plen[sorted[gr.g0[k] + (j - gr.nc1[k])]] + 63u) / 64u;
    if (lane == 0) lc[c] = L;
  }
}
static __global__ void k_sell_pick(const uint32_t* __restrict__ cstep, const SellGroups gr, uint32_t* __restrict__ out) { if (threadIdx.x <= gr.ns) out[threadIdx.x] = cstep[gr.cbase[threadIdx.x]]; }
// every entry to its (chunk, step, lane); the id of every piece to the lane that will hold its sum
template <class T>
__global__ __launch_bounds__(256) void k_sell_scatter(const uint32_t* __restrict__ pcol, const T* __restrict__ pval, const uint32_t* __restrict__ E, const uint32_t* __restrict__ pstart,
                                                      const uint32_t* __restrict__ inv, const uint32_t* __restrict__ lc, const uint32_t* __restrict__ cstep, const SellStreams st, const SellGroups gr,
                                                      uint32_t ntiles, uint32_t* __restrict__ scol, T* __restrict__ sval, uint32_t* __restrict__ perm) {
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6); g < ntiles; g += gridDim.x * 4) {
    const uint32_t k = sell_stream_of_tile(st, g);
    const unsigned long long send = (unsigned long long)st.tbase[k] * WP_ENT + st.ne[k];
    const size_t q0 = (size_t)g * WP_ENT + lane * 4;
    const uint4 wd = *(const uint4*)(pcol + q0);
    const uint32_t w[4] = {wd.x, wd.y, wd.z, wd.w};
    const uint32_t mine = (w[0] >> 31) + (w[1] >> 31) + (w[2] >> 31) + (w[3] >> 31);
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if ((int)lane >= d) incl += o; }
    uint32_t cnt = E[g] + incl - mine;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      cnt += w[j] >> 31;
      const unsigned long long q = q0 + j;
      if (q >= send || cnt == 0) continue;
      const uint32_t id = cnt - 1, o = (uint32_t)q - pstart[id], s = inv[id];
      uint32_t c, ln, step;
      if (s < gr.g0[k]) { const uint32_t r = s - gr.g1[k]; c = gr.cbase[k] + r / 64u; ln = r & 63u; step = o; }
      else { c = gr.cbase[k] + gr.nc1[k] + (s - gr.g0[k]); const uint32_t ll = lc[c]; ln = o / ll; step = o - ln * ll; }
      const size_t dst = ((size_t)cstep[c] + step) * 64u + ln;
      scol[dst] = (w[j] & (XT_COLD | XT_IDXMASK)) | SW_VALID;
      if (pval) sval[dst] = pval[q];
      if (o == 0) perm[(size_t)c * 64u + (s < gr.g0[k] ? ln : 0u)] = id;
    }
  }
}
// the records of the steps, one bit per lane in bit 30 of the column words  (one wave per chunk)
static __global__ __launch_bounds__(256) void k_sell_meta(uint32_t* __restrict__ scol, const uint32_t* __restrict__ lc, const uint32_t* __restrict__ cstep, const SellGroups gr, uint32_t nchunks) {
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t c = blockIdx.x * 4 + (threadIdx.x >> 6); c < nchunks; c += gridDim.x * 4) {
    const uint32_t k = sell_stream_of_chunk(gr, c), L = lc[c]; const bool wave = c - gr.cbase[k] >= gr.nc1[k];
    for (uint32_t t = 0; t < L; t++) {
      const unsigned long long rec = (t + 1 == L ? SM_LAST : 0u) | (wave ? SM_WAVE : 0u) | (t == 0 ? SM_FIRST : 0u) | ((unsigned long long)c << 8);
      const size_t i = ((size_t)cstep[c] + t) * 64u + lane;
      if ((rec >> lane) & 1ull) scol[i] |= SW_META;
    }
  }
}
// work items: runs of whole chunks of about SELL_ITEM steps
static __global__ void k_sell_items(const uint32_t* __restrict__ cstep, const SellGroups gr, const uint32_t* __restrict__ ibase, uint32_t nitems_total, uint32_t nsteps_total, uint32_t* __restrict__ item_first) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i <= nitems_total; i += gridDim.x * 256) {
    if (i == nitems_total) { item_first[i] = nsteps_total; continue; }
    uint32_t k = 0; for (uint32_t j = 1; j < gr.ns; j++) k = i >= ibase[j] ? j : k;
    const uint32_t target = cstep[gr.cbase[k]] + (i - ibase[k]) * SELL_ITEM;
    uint32_t lo = gr.cbase[k], hi = gr.cbase[k + 1];
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (cstep[mid] < target) lo = mid + 1; else hi = mid; }
    item_first[i] = cstep[lo];
  }
}

template <class T> struct SellStage {
  uint32_t step, c;                     // A -> B: the step and its column words
  uint32_t pid, meta; T v, g, hl; unsigned long long valid;      // B -> C (a step behind the end of the work has no valid lane)
};

// one workgroup of W waves per CU; workgroup b walks the stream(s) of XCD b % 8 (as k_spmv_tiles)
template <class T, class SR, int D1 = SELL_D1, int D2 = SELL_D2, int W = XT_WAVES>
__global__ __launch_bounds__(W * 64, 1) void k_spmv_sell(const XtCall<T> call, const SellPanel<T>* __restrict__ panels, const uint32_t* __restrict__ perm, const uint32_t nchunks, const SR sr) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "the lane-per-piece kernel handles 4- and 8-byte types");
  constexpr int H = xt_hot<T>::H;
  constexpr int NS = D1 + 1;
  __shared__ T s_hot[H];
  __shared__ uint32_t s_next;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t lane4 = (uint32_t)lane * 4u, laneT = (uint32_t)lane * (uint32_t)sizeof(T);
  const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)call.u, (short)0, (int)(call.ulen * (uint32_t)sizeof(T)), 0x00020000);
  const __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)perm, (short)0, (int)(nchunks * 256u), 0x00020000);
  for (uint32_t sp = 0; sp < call.sps; sp++) {
  const SellPanel<T> a = panels[(blockIdx.x & 7) * call.sps + sp];
  __syncthreads();
  if (threadIdx.x == 0) s_next = 0;
  const bool use_a = sr.uses_a() && a.sval != nullptr, use_u = sr.uses_u();
  if (use_u) for (uint32_t h = threadIdx.x; h < a.nhot; h += W * 64) s_hot[h] = wp_ld(a.xhot + h);
  if (threadIdx.x == 0) { T z; __builtin_memset(&z, 0, sizeof(T)); s_hot[H - 1] = z; }      // the zero slot (the plan gives it to no column: nhot <= H - 1)
  const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.scol, (short)0, (int)(a.nsteps * 256u), 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.sval, (short)0, (int)(a.sval ? a.nsteps * 64u * (uint32_t)sizeof(T) : 0u), 0x00020000);
  __syncthreads();
  // work items: a static share dealt to (workgroup, wave), the rest handed out by the workgroup's LDS counter (see k_spmv_wavepipe)
  const uint32_t nwg = gridDim.x >> 3, jwg = blockIdx.x >> 3;
  uint32_t s0 = (uint32_t)((uint64_t)a.nitems * a.static_pct / 100 / (nwg * W)); if (s0 > WP_MAX_STATIC) s0 = WP_MAX_STATIC;
  const uint32_t dyn0 = s0 * nwg * W;
  const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane(wv) * nwg + jwg;
  const uint32_t st_step = a.interleave ? nwg * W : 1u;
  uint32_t st_next = a.interleave ? wid : wid * s0, st_left = s0;
  auto next_item = [&]() __attribute__((always_inline)) -> uint32_t {
    if (st_left) { st_left--; const uint32_t c = st_next; st_next += st_step; return c; }
    uint32_t v = 0; if (lane == 0) v = atomicAdd(&s_next, 1u);
    return dyn0 + (uint32_t)__builtin_amdgcn_readfirstlane(v) * nwg + jwg;
  };
  uint32_t it_cur = 0, it_end = 0; bool done = false;
  auto next_step = [&]() __attribute__((always_inline)) -> uint32_t {
    while (it_cur >= it_end) {
      if (done) return WP_NONE;
      const uint32_t it = next_item();
      if (it >= a.nitems) { done = true; return WP_NONE; }
      // the item's bounds through the scalar cache: a vector load here would have to be waited for behind everything the pipeline has in flight
      const uint32_t* ip = a.item_first + it; uint2 b;
      asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(b) : "s"(ip) : "memory");
      it_cur = b.x - a.step0; it_end = b.y - a.step0;
    }
    return it_cur++;
  };

  SellStage<T> S[NS];
  auto stage_a = [&](SellStage<T>& s, uint32_t stp) __attribute__((always_inline)) {        // the step's 64 column words
    s.step = stp;
    s.c = __builtin_amdgcn_raw_buffer_load_b32(c_rsrc, (int)(s.step != WP_NONE ? s.step * 256u + lane4 : 0xFFFFFFFFu), 0, XT_STREAM_AUX);
  };
  auto stage_b = [&](SellStage<T>& s) __attribute__((always_inline)) {        // decode; gather / LDS read / value / partial id on their way
    const uint32_t w = s.c;                                                    // (a step behind the end of the work: zero words — nobody's entry, hot slot 0, no record)
    s.valid = __ballot((int32_t)w < 0);
    const unsigned long long rec = __ballot((w & SW_META) != 0);
    s.meta = (uint32_t)rec & 0xFFu;
    const uint32_t chunk = (uint32_t)(rec >> 8);
    const bool cold = (w & XT_COLD) != 0;
    const uint32_t off = (w & XT_IDXMASK) * (uint32_t)sizeof(T);
    // one of the two reads returns zero bits: the lanes the table serves make no memory request, a cold lane reads the table's zero slot
    s.g = use_u ? xt_buf_load<T>(u_rsrc, cold ? off : 0xFFFFFFFFu) : T();
    s.hl = use_u ? *(const T*)((const char*)s_hot + (cold ? (uint32_t)(H - 1) * (uint32_t)sizeof(T) : off)) : T();
    { T tmp[1]; xt_stream_load<T, 1>(v_rsrc, s.step != WP_NONE ? s.step * (64u * (uint32_t)sizeof(T)) + laneT : 0xFFFFFFFFu, tmp); s.v = tmp[0]; }      // (no values in the plan: a descriptor of length zero — no request, zero bits)
    s.pid = __builtin_amdgcn_raw_buffer_load_b32(p_rsrc, (int)((s.meta & SM_LAST) ? chunk * 256u + lane4 : 0xFFFFFFFFu), 0, 0);
  };
  T acc = sr.identity;
#pragma unroll
  for (int d = 0; d < D1; d++) stage_a(S[d], next_step());
#pragma unroll
  for (int d = 0; d < D2; d++) stage_b(S[d]);
  // the steps of the next round of the ring are drawn BEFORE the round (scalar code with its loops and the items' loads): the round itself is
  // straight-line code around the chunk ends, so that the compiler's s_waitcnt counts leave the later loads in flight
  uint32_t nxt[NS];
  auto step = [&]<int I>() __attribute__((always_inline)) -> bool {
    SellStage<T>& C = S[I % NS]; SellStage<T>& B = S[(I + D2) % NS]; SellStage<T>& A = S[(I + D1) % NS];
    stage_b(B);
    stage_a(A, nxt[I]);
    const T x = use_u ? xt_or_bits<T>(C.g, C.hl) : T();
    const T p = sr.mult(C.v, x);
    if (C.meta & SM_FIRST) acc = xt_select_mask<T>(sr.identity, p, C.valid);
    else acc = xt_select_mask<T>(acc, sr.add(acc, p), C.valid);
    if (C.meta & SM_LAST) {
      if (C.meta & SM_WAVE) {
        const T tot = wp_wave_total<T>(sr.add_op(), acc, sr.identity);       // fixed tree => reproducible
        if (lane == 0) wp_st(call.partial + C.pid, tot);
      } else if (C.pid != WP_NONE) wp_st(call.partial + C.pid, acc);
      acc = sr.identity;
    }
    return true;
  };
  // (no exit inside a round: a step behind the end of the work has no valid lane and no record — it loads nothing and changes nothing — and the
  //  accumulator is carried around the loop on ONE path; thirteen exits made the compiler copy it through registers that loads were in flight to)
  for (;;) {
#pragma unroll
    for (int i = 0; i < NS; i++) nxt[i] = next_step();
    xt_unroll_steps(step, std::make_integer_sequence<int, NS>{});
    if (S[NS - 1].valid == 0ull) break;
  }
  }     // streams of this XCD
}

}  // namespace grb
