// grb_spmv_wavepipe.hpp — SpMV kernel "W": persistent, wave-pipelined, with an LDS-resident table of the hottest
// operand entries.  The FP64 PLUS_TIMES north-star path when u is full and there is no mask.
//
// Why (PMC on kernel A, profiles/r01_spmv_v1_pmc_summary.txt): at R-MAT-22 every one of the 65 M gathers of u costs
// a 128-byte L2->L1 line fill for 8 useful bytes (66 M L1->L2 requests per launch), and kernel A's phases add up
// instead of overlapping (stream-only 0.22 ms + gather 0.36 ms).  Kernel W attacks both:
//   * hot table: columns are ranked by how often they occur; the H most frequent (H*sizeof(T) = 96 KiB of the CU's
//     160 KiB LDS) are staged in LDS once per workgroup.  The plan keeps the column array re-labelled by rank
//     (pcol = rank of the column) and every call first writes u in rank order (one 3n-word pass, ~4 % of the
//     kernel): a hot gather is one ds_read and never reaches L1/L2 (≈ 40 % of all gathers at R-MAT-22), and the
//     remaining gathers find frequently used entries packed into the same 128-byte lines (-25 % L2 misses).
//   * one 1024-thread workgroup per CU, each WAVE runs its own software pipeline over a contiguous range of
//     512-entry tasks (merge-style: tasks split the entry range evenly, rows are found with a per-task row index):
//     the coalesced col/val loads of task t+1 are in flight while task t gathers, multiplies into the wave's
//     private LDS slice and reduces its rows — no workgroup barrier after start-up.
//   * rows that straddle task boundaries are carried in registers along the wave's range; rows that straddle two
//     waves' ranges leave (head, tail) partials that a tiny second kernel combines in wave order, so sums are
//     formed in a fixed order and results are reproducible run to run.
#pragma once
#include "grb_api.hpp"
#include "grb_device.hpp"
#include "grb_semiring.hpp"
#include "grb_spmv.hpp"

namespace grb {

constexpr int WP_ENT = 256;                 // merge items per task = 4 per lane (measured: 512 with 16 waves spills for 8-byte types; 8 waves x 512 is 6 % slower;
                                            // a fourth stage — values two, column words three tasks ahead — spills too and is 4 % slower)
constexpr int WP_PER = WP_ENT / 64;
constexpr int WP_WAVES = 16;                // waves per workgroup (1024 threads)
constexpr int WP_WGS_PER_CU = 1;            // one workgroup per CU (measured: 2 x 768 threads with 256-item tasks spills registers and is slower)
constexpr int WP_LDS_BYTES = 160 * 1024 / WP_WGS_PER_CU;
template <class T> struct wp_hot { static constexpr int H = (WP_LDS_BYTES - WP_WAVES * WP_ENT * (int)sizeof(T) - 16) / (int)sizeof(T); };   // what the scan slices leave: 16382 (8 B) / 36860 (4 B)
#ifndef WP_NT_COLS
#define WP_NT_COLS false
#endif
#ifndef WP_NT_VALS
#define WP_NT_VALS false
#endif
constexpr uint32_t WP_NONE = 0xFFFFFFFFu;
constexpr uint32_t WP_ROWSTART = 0x80000000u, WP_COLMASK = 0x7FFFFFFFu;   // plan column words: bit 31 marks the first entry of a row
constexpr uint32_t WP_CHUNK = 4;              // tasks per chunk (the unit handed out dynamically; one carry record each)
constexpr uint32_t WP_STATIC_PCT = 40;        // share of the chunks that is split statically
inline uint32_t wp_env(const char* name, uint32_t dflt) { const char* e = getenv(name); return e && *e ? (uint32_t)atoi(e) : dflt; }   // tuning hooks
constexpr uint32_t WP_MAX_STATIC = 64;        // a static range owns at most this many chunk ids (one lane writes each empty record)

template <class T> __device__ __forceinline__ T wp_wave_total(int op, T v, T identity) {
  if constexpr (sizeof(T) >= 4) return wave_reduce_dpp<T, false>(op, v, identity); else return wave_reduce_op<T, false>(op, v);
}

#ifdef WP_PROFILE
static __device__ unsigned long long g_wp_prof[8192];   // experiment: cycles of every wave of the last launch, then its task count
#define WP_CLK() __builtin_amdgcn_s_memtime()
#endif
template <class T> struct WpCarry {           // per wave: partial of the row it entered in the middle of / left open
  uint32_t head_row, tail_row; uint8_t head_has, head_done, tail_has, pad; T head_val, tail_val;
};

template <class T> __device__ __forceinline__ void wp_st_carry(WpCarry<T>* p, const WpCarry<T>& c);
template <class T> struct WpArgs {
  const uint32_t* rowptr; const uint32_t* pcol; const T* aval; const T* x; const T* xorig; const uint32_t* hot_cols;
  const uint32_t* trow; const uint32_t* tent;      // merge-path task starts: task t begins at (row trow[t], entry tent[t]); [ntasks+1]
  T* y; uint8_t* ypres; WpCarry<T>* carry; uint32_t nrows, ntasks, nnz, tasks_per_chunk, static_pct, nhot, nwarm;
};

// Tasks are equal slices of the merge of {entries} and {row ends} (WP_ENT items each), so a task holds at most WP_ENT
// entries AND completes at most WP_ENT rows: runs of empty or tiny rows cannot unbalance the waves (R-MAT has both
// 10^5-entry rows and long runs of empty rows).  Task t owns entries [tent[t], tent[t+1]) and completes rows
// [trow[t], trow[t+1]); the entries of row trow[t+1] seen so far are carried to the next task.
//
// A lane owns WP_PER *consecutive* entries of the task (one wide load per stream).  The row sums are a segmented
// inclusive scan of the products in entry order: the plan marks the first entry of every row in bit 31 of its column
// word, the entry lanes scan their own entries, a 6-step wave scan carries sums across lanes, the scanned values go to
// LDS once (conflict-free) and every row reads the value at its last entry.  The cost of a task does not depend on how its
// entries are split into rows (measured before: per-row serial sums spent half of every wave's time in divergent,
// bank-conflicting LDS reads).
// The panel pipeline (grb_spmv_tiles.hpp) reads its argument block from memory, so the compiler cannot tell that the pointers in it are global
// and would emit FLAT loads — which count on lgkmcnt as well as vmcnt, so every wait for an LDS operation would also
// drain the loads the pipeline wants to keep in flight.  All HBM traffic of the kernel goes through these helpers.
#define WP_G __attribute__((address_space(1)))
template <int B> struct wp_word;
template <> struct wp_word<1> { typedef uint8_t type; };
template <> struct wp_word<2> { typedef uint16_t type; };
template <> struct wp_word<4> { typedef uint32_t type; };
template <> struct wp_word<8> { typedef uint64_t type; };
template <class E> __device__ __forceinline__ E wp_ld(const E* p) {
  typedef typename wp_word<sizeof(E)>::type W;
  const W w = *(const WP_G W*)(uintptr_t)p;
  E e; __builtin_memcpy(&e, &w, sizeof(E)); return e;
}
// a wave-uniform value kept in scalar registers (the carried partials live across the whole task loop)
template <class E> __device__ __forceinline__ E wp_uniform(E v) {
  if constexpr (sizeof(E) == 8) { union { E e; int i[2]; } u; u.e = v; u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]); u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]); return u.e; }
  else if constexpr (sizeof(E) == 4) { union { E e; int i; } u; u.e = v; u.i = __builtin_amdgcn_readfirstlane(u.i); return u.e; }
  else { union { E e; uint16_t s; } u; u.s = 0; u.e = v; const int x = __builtin_amdgcn_readfirstlane((int)u.s); u.s = (uint16_t)x; return u.e; }
}
template <class E> __device__ __forceinline__ void wp_st(E* p, E v) {
  typedef typename wp_word<sizeof(E)>::type W;
  W w; __builtin_memcpy(&w, &v, sizeof(E));
  *(WP_G W*)(uintptr_t)p = w;
}
typedef uint32_t wp_u32x4 __attribute__((ext_vector_type(4)));
typedef wp_u32x4 wp_u32x4_u __attribute__((aligned(4)));
template <class T> __device__ __forceinline__ void wp_st_carry(WpCarry<T>* p, const WpCarry<T>& c) {
  wp_st(&p->head_row, c.head_row); wp_st(&p->tail_row, c.tail_row);
  wp_st(&p->head_has, c.head_has); wp_st(&p->head_done, c.head_done); wp_st(&p->tail_has, c.tail_has);
  wp_st(&p->head_val, c.head_val); wp_st(&p->tail_val, c.tail_val);
}
template <class E, int N, bool NT = false> __device__ __forceinline__ void wp_load_run(const E* __restrict__ arr, uint32_t first, uint32_t len, bool fast, E (&out)[N]) {
  if (fast) {
    if constexpr (sizeof(E) >= 4 && (sizeof(E) * N) % 16 == 0) {                // 16-byte loads, element-aligned
      wp_u32x4 tmp[sizeof(E) * N / 16];
#pragma unroll
      for (int j = 0; j < (int)(sizeof(E) * N / 16); j++) {
        const WP_G wp_u32x4_u* q = &((const WP_G wp_u32x4_u*)(uintptr_t)(arr + first))[j];
        tmp[j] = NT ? __builtin_nontemporal_load(q) : *q;
      }
      __builtin_memcpy(&out[0], &tmp[0], sizeof(E) * N);
    } else {
#pragma unroll
      for (int i = 0; i < N; i++) out[i] = wp_ld(arr + first + i);
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; i++) { const uint32_t k = first + i; out[i] = wp_ld(arr + (k < len ? k : len - 1)); }
  }
}

// Inclusive segmented scan of (v, f) over the 64 lanes with DPP moves (no LDS crossbar round trips: the 6-step chain is
// on every task's critical path): 4 shifts inside the 16-lane rows, then the two row broadcasts of gfx9.  f = "a segment
// starts at or before this lane (within what has been scanned)".
template <class T, class SR> __device__ __forceinline__ void wp_seg_scan(T& v, int& f, int lane, const SR& sr) {
  if constexpr (sizeof(T) == 4 || sizeof(T) == 8) {
    const int l16 = lane & 15, row = (lane >> 4) & 3;
#define WP_SCAN_STEP(CTRL, MASK, COND) { const T vu = dpp_move_t<T, CTRL, MASK>(v); const int fu = (int)dpp_mov<CTRL, MASK>((uint32_t)f, (uint32_t)f); \
                                         if (COND) { if (!f) v = sr.add(vu, v); f |= fu; } }
    WP_SCAN_STEP(0x111, 0xf, l16 >= 1)      // row_shr:1
    WP_SCAN_STEP(0x112, 0xf, l16 >= 2)      // row_shr:2
    WP_SCAN_STEP(0x114, 0xf, l16 >= 4)      // row_shr:4
    WP_SCAN_STEP(0x118, 0xf, l16 >= 8)      // row_shr:8
    WP_SCAN_STEP(0x142, 0xa, row == 1 || row == 3)   // row_bcast:15: the total of row r reaches row r+1 (rows 1 and 3)
    WP_SCAN_STEP(0x143, 0xc, row >= 2)               // row_bcast:31: lane 31 (rows 0-1 scanned) reaches rows 2 and 3
#undef WP_SCAN_STEP
  } else {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const T vu = shfl_up_t<T>(v, d); const int fu = __shfl_up(f, d, 64);
      if (lane >= d) { if (!f) v = sr.add(vu, v); f |= fu; }
    }
  }
}

template <class T> struct WpStage { uint32_t c[WP_PER]; T v[WP_PER], g[WP_PER]; uint32_t rpa, rpb; };   // what one task has in flight

template <class T, class SR>
__global__ __launch_bounds__(WP_WAVES * 64, WP_WGS_PER_CU * WP_WAVES / 4) void k_spmv_wavepipe(const WpArgs<T> a, const SR sr) {
  constexpr int H = wp_hot<T>::H;
  __shared__ T s_hot[H];
  __shared__ __attribute__((aligned(16))) T s_scan[WP_WAVES][WP_ENT];
  __shared__ uint32_t s_next;                                         // next dynamic chunk of this workgroup
  if (threadIdx.x == 0) s_next = 0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool use_a = sr.uses_a(), use_u = sr.uses_u();
  if (use_u) for (uint32_t h = threadIdx.x; h < a.nhot; h += WP_WAVES * 64) s_hot[h] = wp_ld(a.x + h);      // the table's contents, gathered from u once per call
  __syncthreads();
  T* scan = s_scan[wv];
#ifdef WP_PROFILE
  const unsigned long long pf_t0 = WP_CLK();
#endif
  // Work split.  Waves do not run at the same speed (measured: per-wave times of one launch spread +-30 % around the
  // mean whatever the static split — the sum of ~70 tasks of randomly stalled gathers), so with equal shares most of the
  // chip idles while the slowest wave finishes.  The chunks (`tasks_per_chunk` tasks each) are split evenly between the
  // workgroups; inside a workgroup every wave first takes a contiguous static range (40 % of the chunks) and then the
  // rest one chunk at a time from a counter in LDS.  (Counters in HBM were measured and rejected: same-address atomics
  // complete one per ~80 ns per address, which made the whole kernel 1.5-2x slower.)  Every chunk id has a carry
  // record; a static range uses the record of its first chunk and leaves the others empty.
  const uint32_t K = a.tasks_per_chunk, nchunks = (K + a.ntasks - 1) / K;
  const uint32_t nwg = gridDim.x, jwg = blockIdx.x;
  // chunk ids [0, dyn0) are static ranges of s0 chunks, dealt to (workgroup, wave) so that every workgroup samples the
  // whole matrix (its parts differ in cost); ids >= dyn0 are dynamic, workgroup j owning those congruent to j
  uint32_t s0 = (uint32_t)((uint64_t)nchunks * a.static_pct / 100 / (nwg * WP_WAVES)); if (s0 > WP_MAX_STATIC) s0 = WP_MAX_STATIC;
  const uint32_t dyn0 = s0 * nwg * WP_WAVES;
  auto grab = [&]() { uint32_t v = 0; if (lane == 0) v = atomicAdd(&s_next, 1u); return v; };
  uint32_t rec = ((uint32_t)__builtin_amdgcn_readfirstlane(wv) * nwg + jwg) * s0, nrec = s0;
  if (nrec == 0) { rec = dyn0 + (uint32_t)__builtin_amdgcn_readfirstlane(grab()) * nwg + jwg; nrec = 1; }
  while (rec < nchunks) {
  const uint32_t t0 = rec * K;
  uint32_t t1 = t0 + nrec * K; if (t1 > a.ntasks) t1 = a.ntasks;
  uint32_t next_raw = 0;
  uint32_t end_r = 0;                                                    // row the range ends in
  WpCarry<T> cr; cr.head_row = cr.tail_row = WP_NONE; cr.head_has = cr.head_done = cr.tail_has = cr.pad = 0; cr.head_val = cr.tail_val = sr.identity;

  T carry = sr.identity; bool carry_has = false, owned = true;        // partial of the row the current task starts in (wave-uniform)
  // stage 1: the column indices of a task + the row pointers of the first 64 rows it completes.
  // stage 2: its values and the gathers of u that the LDS table does not serve.
  // Lanes past the end of a task read the next task's entries (in bounds; only the very last task takes the clamped path).
  auto load_cols = [&](uint32_t e0, uint32_t r0, uint32_t (&c)[WP_PER], uint32_t& rpa, uint32_t& rpb) {
    wp_load_run<uint32_t, WP_PER, WP_NT_COLS>(a.pcol, e0 + lane * WP_PER, a.nnz, a.nnz - e0 >= (uint32_t)WP_ENT, c);
    const uint32_t rq = r0 + lane;
    rpa = wp_ld(a.rowptr + (rq < a.nrows ? rq : a.nrows)); rpb = wp_ld(a.rowptr + (rq + 1 < a.nrows ? rq + 1 : a.nrows));
  };
  auto issue_gather = [&](uint32_t e0, uint32_t cnt, const uint32_t (&c)[WP_PER], T (&v)[WP_PER], T (&g)[WP_PER]) {
    if (use_a) wp_load_run<T, WP_PER, WP_NT_VALS>(a.aval, e0 + lane * WP_PER, a.nnz, a.nnz - e0 >= (uint32_t)WP_ENT, v);
    else {
#pragma unroll
      for (int u = 0; u < WP_PER; u++) v[u] = T();
    }
#pragma unroll
    for (int u = 0; u < WP_PER; u++) {
      const uint32_t cc = (uint32_t)(lane * WP_PER + u) < cnt ? (c[u] & WP_COLMASK) : 0u;   // rank of the column if < nwarm (0 = most frequent), else nwarm + column
      const T* xb = cc < a.nwarm ? a.x : a.xorig - a.nwarm;           // warm: rank-ordered copy of the top of u; cold: u itself
      g[u] = use_u ? wp_ld(xb + (cc >= (uint32_t)H ? cc : 0u)) : T();  // LDS-resident ranks read element 0 (always cached) and are replaced below
    }
  };
  // task descriptors live in registers, one task per lane, 61 tasks + 3 look-ahead at a time: the steady state
  // issues no dependent global load
  for (uint32_t tc = t0; tc < t1; tc += 61) {
    const uint32_t tl = tc + lane <= a.ntasks ? tc + lane : a.ntasks;
    const uint32_t d_r = wp_ld(a.trow + tl), d_e = wp_ld(a.tent + tl);
    const uint32_t nin = t1 - tc < 61u ? t1 - tc : 61u;
    if (tc + 61 >= t1) { next_raw = grab(); end_r = __builtin_amdgcn_readlane(d_r, (int)nin); }   // last window: ask for the next chunk now
    // three tasks are in flight per wave: task i (values + gathers issued, being reduced), task i+1 (column indices
    // loaded, gathers issued during iteration i) and task i+2 (column indices issued)
    WpStage<T> S0, S1, S2;
    {
      const uint32_t r00 = __builtin_amdgcn_readlane(d_r, 0), e00 = __builtin_amdgcn_readlane(d_e, 0), e01 = __builtin_amdgcn_readlane(d_e, 1);
      if (tc == t0) owned = wp_ld(a.rowptr + (r00 < a.nrows ? r00 : a.nrows)) == e00;   // does this wave see the start of its first row?
      load_cols(e00, r00, S0.c, S0.rpa, S0.rpb);
      if (nin > 1) load_cols(e01, __builtin_amdgcn_readlane(d_r, 1), S1.c, S1.rpa, S1.rpb);
      issue_gather(e00, e01 - e00, S0.c, S0.v, S0.g);
    }
    // one task: A is reduced while the gathers of B and the column indices of C are issued.  The loop below is unrolled
    // three times so that the three register sets swap roles without being copied.
    auto step = [&](uint32_t i, WpStage<T>& A, WpStage<T>& B, WpStage<T>& C) __attribute__((always_inline)) {
      const int iu = (int)__builtin_amdgcn_readfirstlane(i);
      const uint32_t r0 = __builtin_amdgcn_readlane(d_r, iu), e0 = __builtin_amdgcn_readlane(d_e, iu);
      const uint32_t r1 = __builtin_amdgcn_readlane(d_r, iu + 1), e1 = __builtin_amdgcn_readlane(d_e, iu + 1);
      const uint32_t cnt = e1 - e0, nr = r1 - r0;
      const bool more = i + 1 < nin, more2 = i + 2 < nin;
      // ---- entries: products (LDS table for the hottest ranks) and their segmented scan in entry order
      T p[WP_PER];
#pragma unroll
      for (int u = 0; u < WP_PER; u++) {
        const uint32_t cc = A.c[u] & WP_COLMASK;
        const T uvv = use_u ? (cc < (uint32_t)H ? s_hot[cc < (uint32_t)H ? cc : 0] : A.g[u]) : T();
        p[u] = sr.mult(A.v[u], uvv);                             // entries past cnt hold junk: a forward scan never lets it reach a live position
      }
      if (more) issue_gather(e1, __builtin_amdgcn_readlane(d_e, iu + 2) - e1, B.c, B.v, B.g);                          // stage 2 of task i+1
      if (more2) load_cols(__builtin_amdgcn_readlane(d_e, iu + 2), __builtin_amdgcn_readlane(d_r, iu + 2), C.c, C.rpa, C.rpb);   // stage 1 of task i+2
      {
        bool st[WP_PER];                                        // my entry u is the first of its row (bit 31 of the column word, set by the plan)
#pragma unroll
        for (int u = 0; u < WP_PER; u++) st[u] = (int32_t)A.c[u] < 0;
        if (lane == 0 && !carry_has) st[0] = true;              // nothing carried in: entry 0 starts a segment whatever it is
        T agg = p[0]; bool anyf = st[0];
#pragma unroll
        for (int u = 1; u < WP_PER; u++) { agg = st[u] ? p[u] : sr.add(agg, p[u]); anyf = anyf || st[u]; }
        if (lane == 0 && !anyf) agg = sr.add(carry, agg);       // the carried partial flows through lane 0
        T v = agg; int f = anyf;                                // sum since the last row start at or before my last entry
        wp_seg_scan<T, SR>(v, f, lane, sr);
        T run = shfl_up_t<T>(v, 1); if (lane == 0) run = carry;   // what flows into my first entry (unused when it starts a row)
#pragma unroll
        for (int u = 0; u < WP_PER; u++) { run = st[u] ? p[u] : sr.add(run, p[u]); p[u] = run; }
        __builtin_memcpy(&scan[lane * WP_PER], &p[0], sizeof(T) * WP_PER);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
      // ---- rows: a row's sum is the scanned value at its last entry; row r0 also owns what was carried in
      uint32_t tail_start = 0;                                  // task-local offset where the entries of row r1 begin
      for (uint32_t rbase = 0; rbase < nr; rbase += 64) {
        const uint32_t ri = rbase + lane; const bool live = ri < nr; const uint32_t r = r0 + ri;
        uint32_t rs_, re_;
        if (rbase == 0) { rs_ = A.rpa; re_ = A.rpb; } else { rs_ = live ? wp_ld(a.rowptr + r) : e0; re_ = live ? wp_ld(a.rowptr + r + 1) : e0; }
        if (rs_ < e0) rs_ = e0;                                 // only row r0 can have started in an earlier task
        if (!live) { rs_ = re_ = e0; }
        const uint32_t qs = rs_ - e0, qe = re_ - e0;
        if (rbase + 64 >= nr) tail_start = __shfl(qe, (int)(nr - 1 - rbase), 64);
        T acc = sr.identity; bool has = false;
        if (live && qe > qs) { acc = scan[qe - 1]; has = true; }
        else if (rbase == 0 && lane == 0 && carry_has) { acc = carry; has = true; }   // row r0 ended exactly where this task starts
        const bool to_fixup = rbase == 0 && lane == 0 && !owned;      // the row began in another wave's range
        if (live && !to_fixup) { if (has) wp_st(a.y + r, acc); wp_st(a.ypres + r, (uint8_t)(has ? 1 : 0)); }
        if (rbase == 0 && !owned) {
          cr.head_row = r0; cr.head_val = wp_uniform(acc); cr.head_has = (uint8_t)__builtin_amdgcn_readfirstlane((int)has); cr.head_done = 1;
        }
      }
      // entries [tail_start, cnt) belong to row r1, which ends in a later task: they become the carry
      if (nr) {
        owned = true;
        if (tail_start < cnt) { carry = wp_uniform(scan[cnt - 1]); carry_has = true; } else { carry = sr.identity; carry_has = false; }
      } else if (cnt) { carry = wp_uniform(scan[cnt - 1]); carry_has = true; }           // still inside row r0 (includes what was carried in)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();   // LDS slices are free for the next task
    };
    for (uint32_t i = 0; i < nin; i += 3) {
      step(i, S0, S1, S2);
      if (i + 1 < nin) step(i + 1, S1, S2, S0);
      if (i + 2 < nin) step(i + 2, S2, S0, S1);
    }
  }
  // the row the range ends in, if it ends strictly inside it or at its very end without its end marker
  if (end_r < a.nrows && (carry_has || !owned)) {
    if (owned) { cr.tail_row = end_r; cr.tail_val = carry; cr.tail_has = carry_has; }
    else { cr.head_row = end_r; cr.head_val = carry; cr.head_has = carry_has; cr.head_done = 0; }   // the whole range lies inside one row
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

// combine the partials of rows that span several waves' ranges, in wave order
template <class T, class SR>
__global__ void k_spmv_wavepipe_fixup(const WpCarry<T>* __restrict__ carry0, uint32_t nwaves /* records */, T* __restrict__ y0, uint8_t* __restrict__ ypres0,
                                      const WpArgs<T>* __restrict__ panels, const SR sr) {
  const WpCarry<T>* carry = panels ? panels[blockIdx.y].carry : carry0;
  if (panels) nwaves = (panels[blockIdx.y].ntasks + panels[blockIdx.y].tasks_per_chunk - 1) / panels[blockIdx.y].tasks_per_chunk;   // records = chunks of this panel
  T* y = panels ? panels[blockIdx.y].y : y0; uint8_t* ypres = panels ? panels[blockIdx.y].ypres : ypres0;
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < nwaves; w += gridDim.x * blockDim.x) {
    const WpCarry<T> c = carry[w];
    if (c.head_row == WP_NONE || !c.head_done) continue;
    const uint32_t row = c.head_row;
    uint32_t v = w;                                         // walk back to the wave that owns the start of the row
    while (v > 0) { v--; if (carry[v].tail_row == row) break; }
    T acc = carry[v].tail_val; bool has = carry[v].tail_has != 0;
    for (uint32_t m = v + 1; m <= w; m++) {
      const WpCarry<T> h = carry[m];
      if (h.head_has) { acc = has ? sr.add(acc, h.head_val) : h.head_val; has = true; }
    }
    if (has) y[row] = acc;
    ypres[row] = has ? 1 : 0;
  }
}

// tasks per chunk: about 16 chunks per wave, at least WP_CHUNK tasks each
inline uint32_t wp_chunk_tasks(uint32_t ntasks, uint32_t nwaves) {
  const uint32_t k = ntasks / ((nwaves ? nwaves : 1) * wp_env("GRB_MI355X_WP_CHUNKS", 16u));
  const uint32_t lo = wp_env("GRB_MI355X_WP_MINCHUNK", WP_CHUNK);
  return k > lo ? k : lo;
}
// ---- plan pieces -------------------------------------------------------------------------------------------------------
static __global__ void k_wp_task_starts(const uint32_t* __restrict__ rowptr, uint32_t nrows, uint32_t nnz, uint32_t ntasks,
                                         uint32_t* __restrict__ trow, uint32_t* __restrict__ tent) {
  for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t <= ntasks; t += gridDim.x * 256) {
    if (t == ntasks) { trow[t] = nrows; tent[t] = nnz; continue; }
    const unsigned long long D = (unsigned long long)t * WP_ENT;       // diagonal of the (entries x row-ends) merge
    uint32_t lo = 0, hi = nrows;                                         // number of row ends before D: largest r with rowptr[r] + r <= D
    while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if ((unsigned long long)rowptr[mid] + mid <= D) lo = mid; else hi = mid - 1; }
    trow[t] = lo; tent[t] = (uint32_t)(D - lo);
  }
}
static __global__ void k_iota_u32_wp(uint32_t* p, uint64_t n) { for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = (uint32_t)i; }
// counts = run lengths of a sorted key array (the column counts of kernel X's plan, the sampled ones of kernel W's)
static __global__ void k_xp_run_starts(const uint32_t* __restrict__ sorted, uint64_t nnz, uint32_t* __restrict__ first) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < nnz; i += gridDim.x * 256ull) if (i == 0 || sorted[i] != sorted[i - 1]) first[sorted[i]] = (uint32_t)i;
}
static __global__ void k_xp_run_lengths(const uint32_t* __restrict__ sorted, uint64_t nnz, const uint32_t* __restrict__ first, uint32_t* __restrict__ cnt) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < nnz; i += gridDim.x * 256ull) if (i + 1 == nnz || sorted[i] != sorted[i + 1]) { const uint32_t c = sorted[i]; cnt[c] = (uint32_t)(i + 1) - first[c]; }
}
// every stride-th entry's column (the sample the hot / warm ranking is taken from)
static __global__ void k_wp_sample(const uint32_t* __restrict__ col, uint64_t nnz, uint64_t stride, uint64_t m, uint32_t* __restrict__ out) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < m; i += gridDim.x * 256ull) { const uint64_t p = i * stride; out[i] = col[p < nnz ? p : nnz - 1]; }
}
static __global__ void k_wp_neg_keys(const uint32_t* __restrict__ cnt, uint32_t n, uint32_t top, uint32_t* __restrict__ key, uint32_t* __restrict__ id) {
  for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) { key[j] = top - (cnt[j] < top ? cnt[j] : top); id[j] = j; }   // ascending sort = descending count, ties by column
}
static __global__ void k_wp_rank(const uint32_t* __restrict__ sorted_id, uint32_t n, uint32_t* __restrict__ rank) {
  for (uint32_t h = blockIdx.x * 256 + threadIdx.x; h < n; h += gridDim.x * 256) rank[sorted_id[h]] = h;
}
static __global__ void k_wp_remap(const uint32_t* __restrict__ col, uint64_t nnz, const uint32_t* __restrict__ rank, uint32_t nwarm, uint32_t* __restrict__ pcol) {
  for (uint64_t p = blockIdx.x * 256ull + threadIdx.x; p < nnz; p += gridDim.x * 256ull) {
    const uint32_t c = col[p], r = rank[c];
    pcol[p] = r < nwarm ? r : nwarm + c;
  }
}

// bit 31 of the column word of the first entry of every non-empty row
static __global__ void k_wp_mark_row_starts(const uint32_t* __restrict__ rowptr, uint32_t nrows, uint32_t* __restrict__ pcol) {
  for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < nrows; r += gridDim.x * 256) { const uint32_t s = rowptr[r]; if (rowptr[r + 1] > s) pcol[s] |= WP_ROWSTART; }
}

template <class T> void build_wavepipe_plan(DevCSR& M) {
  constexpr uint32_t H = wp_hot<T>::H;
  auto grid_n = [](uint64_t n) { uint64_t b = (n + 255) / 256; if (b < 1) b = 1; if (b > 8192) b = 8192; return (unsigned)b; };
  const uint32_t n = M.ncols, ntasks = (uint32_t)((M.nnz + M.nrows + WP_ENT - 1) / WP_ENT);
  M.wp_rs.alloc(((size_t)ntasks + 1) * 8);            // trow[ntasks+1] then tent[ntasks+1]
  hipLaunchKernelGGL(k_wp_task_starts, dim3(grid_n(ntasks + 1)), dim3(256), 0, stream(), M.rowptr.as<uint32_t>(), M.nrows, (uint32_t)M.nnz, ntasks,
                     M.wp_rs.as<uint32_t>(), M.wp_rs.as<uint32_t>() + (ntasks + 1));
  DevBuf cnt((size_t)n * 4 + 4), key((size_t)n * 4 + 4), id((size_t)n * 4 + 4), key2((size_t)n * 4 + 4), id2((size_t)n * 4 + 4), rank((size_t)n * 4 + 4);
  GRB_HIP(hipMemsetAsync(cnt.p, 0, (size_t)n * 4 + 4, stream()));
  // Column frequencies from a SAMPLE of ~2^22 entries (round 4).  The ranking only decides which columns sit in the LDS table and in the
  // rank-ordered warm copy — any choice is correct, and the columns that matter are those a sample finds.  Counting every entry cost
  // 4.5 ms at R-MAT-22 (device atomics; the hottest columns serialise), most of this plan; the sample is sorted (a radix sort of 4 M
  // keys) and counted as run lengths — no atomics.  This is what makes kernel W the product a matrix runs FIRST (grb_spmv_kernels.hpp).
  { const uint64_t target = (uint64_t)wp_env("GRB_MI355X_WP_SAMPLE", 1u << 22);
    const uint64_t stride = M.nnz > target ? M.nnz / target : 1, m = (M.nnz + stride - 1) / stride;
    DevBuf samp(m * 4 + 4), sorted(m * 4 + 4), first((size_t)n * 4 + 4);
    int cb = 1; while ((1ull << cb) < (unsigned long long)n) cb++;
    hipLaunchKernelGGL(k_wp_sample, dim3(grid_n(m)), dim3(256), 0, stream(), M.col.as<uint32_t>(), M.nnz, stride, m, samp.as<uint32_t>());
    sort_keys_u32(samp.as<uint32_t>(), sorted.as<uint32_t>(), m, cb);
    hipLaunchKernelGGL(k_xp_run_starts, dim3(grid_n(m)), dim3(256), 0, stream(), sorted.as<uint32_t>(), m, first.as<uint32_t>());
    hipLaunchKernelGGL(k_xp_run_lengths, dim3(grid_n(m)), dim3(256), 0, stream(), sorted.as<uint32_t>(), m, first.as<uint32_t>(), cnt.as<uint32_t>());
    uint32_t top = 1; int kb = 1; while ((uint64_t)top < m + 1 && kb < 32) { top <<= 1; kb++; }       // counts <= m: keys of kb bits
    hipLaunchKernelGGL(k_wp_neg_keys, dim3(grid_n(n)), dim3(256), 0, stream(), cnt.as<uint32_t>(), n, top, key.as<uint32_t>(), id.as<uint32_t>());
    sort_pairs_u32(key.as<uint32_t>(), key2.as<uint32_t>(), id.as<uint32_t>(), id2.as<uint32_t>(), n, kb); }
  const uint32_t nhot = n < H ? n : H;   // (nwarm >= nhot always: H*sizeof(T) <= 96 KiB)
  // the `nwarm` most frequent columns (two XCD-L2s' worth of u) are gathered from a rank-ordered copy made per call;
  // rarer ones straight from u
  const uint32_t nwarm_cap = (uint32_t)((8u << 20) / sizeof(T));
  const uint32_t nwarm = n < nwarm_cap ? n : nwarm_cap;
  M.wp_hot.alloc((size_t)nwarm * 4 + 4);               // order[rank] = original column for rank < nwarm
  GRB_HIP(hipMemcpyAsync(M.wp_hot.p, id2.p, (size_t)nwarm * 4, hipMemcpyDeviceToDevice, stream()));
  M.wp_nwarm = nwarm;
  GRB_HIP(hipMemsetAsync(rank.p, 0xFF, (size_t)n * 4 + 4, stream()));
  hipLaunchKernelGGL(k_wp_rank, dim3(grid_n(n)), dim3(256), 0, stream(), id2.as<uint32_t>(), n, rank.as<uint32_t>());
  M.wp_pcol.alloc(M.nnz * 4 + 4);
  hipLaunchKernelGGL(k_wp_remap, dim3(grid_n(M.nnz)), dim3(256), 0, stream(), M.col.as<uint32_t>(), M.nnz, rank.as<uint32_t>(), nwarm, M.wp_pcol.as<uint32_t>());
  hipLaunchKernelGGL(k_wp_mark_row_starts, dim3(grid_n(M.nrows)), dim3(256), 0, stream(), M.rowptr.as<uint32_t>(), M.nrows, M.wp_pcol.as<uint32_t>());
  M.wp_nhot = nhot; M.wp_ntasks = ntasks; M.wp_tsize = (int)sizeof(T);
  GRB_HIP(hipStreamSynchronize(stream()));
}

template <class T> __global__ void k_wp_permute(const T* __restrict__ x, const uint32_t* __restrict__ order, uint32_t n, T* __restrict__ xp) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) xp[i] = x[order[i]];
}

template <class T> bool run_wavepipe(const SpmvCall& c, const SemiringDesc& d, int ncu) {
  DevCSR& M = *c.M;
  if (M.wp_tsize != (int)sizeof(T)) build_wavepipe_plan<T>(M);
  // u in column-rank order: frequently used entries share cache lines (and the first H of them are the LDS table)
  DevBuf xp((size_t)M.wp_nwarm * sizeof(T) + 8);
  const bool uses_u = d.flip ? binop_uses_x(d.mulop) : binop_uses_y(d.mulop);
  if (uses_u) hipLaunchKernelGGL((k_wp_permute<T>), dim3(2048), dim3(256), 0, stream(), (const T*)c.uval, M.wp_hot.as<uint32_t>(), M.wp_nwarm, xp.as<T>());
  const uint32_t tpw = wp_chunk_tasks(M.wp_ntasks, (uint32_t)ncu * WP_WGS_PER_CU * WP_WAVES), nwaves = (M.wp_ntasks + tpw - 1) / tpw;      // one record per chunk
  if (M.wp_carry.bytes < (size_t)nwaves * sizeof(WpCarry<T>)) M.wp_carry.alloc((size_t)nwaves * sizeof(WpCarry<T>));
  WpArgs<T> a{M.rowptr.as<uint32_t>(), M.wp_pcol.as<uint32_t>(), (const T*)c.aval, (const T*)xp.p, (const T*)c.uval, M.wp_hot.as<uint32_t>(), M.wp_rs.as<uint32_t>(), M.wp_rs.as<uint32_t>() + (M.wp_ntasks + 1),
              (T*)c.tval, c.tpres, M.wp_carry.as<WpCarry<T>>(), M.nrows, M.wp_ntasks, (uint32_t)M.nnz, tpw, wp_env("GRB_MI355X_WP_STATIC", WP_STATIC_PCT), M.wp_nhot, M.wp_nwarm};
  with_semiring<T>(d, [&](auto sr) {
    typedef decltype(sr) SR;
    hipLaunchKernelGGL((k_spmv_wavepipe<T, SR>), dim3(ncu * WP_WGS_PER_CU), dim3(WP_WAVES * 64), 0, stream(), a, sr);
    hipLaunchKernelGGL((k_spmv_wavepipe_fixup<T, SR>), dim3((nwaves + 255) / 256), dim3(256), 0, stream(), M.wp_carry.as<WpCarry<T>>(), nwaves, (T*)c.tval, c.tpres,
                       (const WpArgs<T>*)nullptr, sr);
    g_last_plan += std::string("k_spmv_wavepipe<") + (sr.is_static ? "static" : "dynamic") + ",hot=" + std::to_string(M.wp_nhot) + "> ";
  });
  return true;
}

}  // namespace grb
