// grb_spgemm_hash.hpp — the unmasked (or complement-masked) SpGEMM  T = A (+).(x) B : Gustavson row by row with an
// LDS hash accumulator, in two passes (the path BASELINE.json's north star names; reference call: lib.GrB_mxm at
// pygraphblas/matrix.py:2572-2583 without a mask, e.g. `A @ A`).
//
//   symbolic  nnz(T(i,:)) for every row: the columns of the products of row i go into a hash set in LDS (keys only:
//             up to 32768 slots = 128 KB); the set's size is bounded by the row's product count
//             ub(i) = sum_{k in A(i,:)} nnz(B(k,:)), by which the rows are binned (<= 128 / 1024 / 4096 / 16384 products: 256 /
//             2048 / 8192 / 32768 slots, a wave / 256 / 512 / 1024 threads per row).  Rows with more products count into a bitmap of
//             ncols bits in HBM (one per persistent workgroup; atomicOr returns which bits were new).
//   scan      row pointers of T (one host round trip: the size of T).
//   numeric   the rows are binned again, now by their exact entry count (<= 128 / 1024 / 4096 entries: 256 / 2048 / 8192
//             slots of key + accumulator, load <= 1/2); products combine into the slot of their column with the monoid's
//             native LDS atomic (CAS loop for monoids that have none), and the live slots are written to T's row.  Rows
//             with more entries accumulate into a dense array of ncols accumulators in HBM (one per persistent
//             workgroup, initialised to the identity once and restored after each row), claiming columns in a bitmap.
//   order     the entries of a row leave the tables in slot order: one segmented radix sort (rocPRIM) over (column, position)
//             pairs puts every row in column order, and the values follow their positions.
// Teams walk the entries k of A(i,:) with 16-lane groups, each streaming one row B(k,:) 16 entries at a time.
// Temporaries: 8 B per row + 8 B per entry of T (sort) — against ~40 B per *product* of the expand/sort/compress path,
// which stays as the deterministic alternative (floating-point sums here depend on the order the atomics land in; integer
// and boolean results are exact): GRB_MI355X_SPGEMM=esc, or descriptor AxB method GxB_AxB_DOT.
#pragma once
#include "grb_spgemm_kernels.hpp"

namespace grb {

void segmented_sort_pairs_u32(const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, uint64_t n, uint32_t nseg, const uint32_t* begins, const uint32_t* ends, int end_bit);

struct HashArgs {
  const uint32_t* arp; const uint32_t* acol; const uint32_t* brp; const uint32_t* bcol;
  const uint32_t* rows; uint32_t nrows_bin;            // the rows of this bin
  uint32_t* rownnz;                                     // symbolic: out
  const uint32_t* crp; uint32_t* ccol;                  // numeric: T's row pointers, columns (unsorted inside a row)
};

// rows -> bins by a per-row weight (product count or entry count); bin b holds rows with limit[b-1] < w <= limit[b], the last bin the rest
static __global__ void k_hash_bin(uint32_t nrows, const unsigned long long* __restrict__ w64, const uint32_t* __restrict__ w32, unsigned long long l0, unsigned long long l1,
                                  unsigned long long l2, uint32_t* __restrict__ counts, uint32_t* __restrict__ lists /* 4 x nrows */) {
  const int lane = threadIdx.x & 63;
  const uint64_t nround = ((uint64_t)nrows + 63) / 64 * 64;
  for (uint64_t r = blockIdx.x * 256ull + threadIdx.x; r < nround; r += (uint64_t)gridDim.x * 256ull) {
    int b = -1;
    if (r < nrows) { const unsigned long long w = w64 ? w64[r] : w32[r]; if (w) b = w <= l0 ? 0 : (w <= l1 ? 1 : (w <= l2 ? 2 : 3)); }
    for (int bb = 0; bb < 4; bb++) {
      const unsigned long long m = __ballot(b == bb);
      if (!m) continue;
      const int leader = __builtin_ctzll(m);
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(&counts[bb], (uint32_t)__popcll(m));
      base = __shfl(base, leader, 64);
      if (b == bb) lists[(size_t)bb * nrows + base + __popcll(m & ((1ull << lane) - 1))] = (uint32_t)r;
    }
  }
}

// the symbolic pass bins by product count into FIVE classes (<= l0 / l1 / l2 / l3, the rest): rows of 1025 ... 4096 products get a
// table of their own (8192 slots, 512 threads, five workgroups per CU) instead of sharing the 32 768-slot one, whose 128 KiB are
// cleared and counted per row by a workgroup that is alone on its CU (30 of the 190 ms of A@A on R-MAT-18)
static __global__ void k_hash_bin5(uint32_t nrows, const unsigned long long* __restrict__ w64, unsigned long long l0, unsigned long long l1, unsigned long long l2, unsigned long long l3,
                                   uint32_t* __restrict__ counts, uint32_t* __restrict__ lists /* 5 x nrows */) {
  const int lane = threadIdx.x & 63;
  const uint64_t nround = ((uint64_t)nrows + 63) / 64 * 64;
  for (uint64_t r = blockIdx.x * 256ull + threadIdx.x; r < nround; r += (uint64_t)gridDim.x * 256ull) {
    int b = -1;
    if (r < nrows) { const unsigned long long w = w64[r]; if (w) b = w <= l0 ? 0 : (w <= l1 ? 1 : (w <= l2 ? 2 : (w <= l3 ? 3 : 4))); }
    for (int bb = 0; bb < 5; bb++) {
      const unsigned long long m = __ballot(b == bb);
      if (!m) continue;
      const int leader = __builtin_ctzll(m);
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(&counts[bb], (uint32_t)__popcll(m));
      base = __shfl(base, leader, 64);
      if (b == bb) lists[(size_t)bb * nrows + base + __popcll(m & ((1ull << lane) - 1))] = (uint32_t)r;
    }
  }
}

// LDS hash, one team per row.  NUMERIC = false: count the distinct columns; true: accumulate and write the row.
template <class T, class SR, int SLOTS, int TEAM, int BLOCK, bool NUMERIC>
__global__ __launch_bounds__(BLOCK) void k_spgemm_hash(const HashArgs a, const T* __restrict__ aval, const T* __restrict__ bval, T* __restrict__ cval, const SR sr) {
  typedef typename acc_word<T>::type W;
  constexpr int TEAMS = BLOCK / TEAM, NG = TEAM / 16;
  __shared__ uint32_t s_key[TEAMS][SLOTS];
  __shared__ W s_acc[NUMERIC ? TEAMS : 1][NUMERIC ? SLOTS : 1];
  __shared__ uint32_t s_cnt[TEAMS];
  const int team = threadIdx.x / TEAM, t = threadIdx.x % TEAM, lane16 = t & 15, grp = t >> 4;
  uint32_t* key = s_key[team];
  const bool use_a = sr.uses_a(), use_b = sr.uses_u();
  const W idw = to_word<T>(sr.identity);
  auto team_sync = [&]() {
    if constexpr (TEAM == 64) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
  };
  const uint32_t nblk_rows = (a.nrows_bin + TEAMS - 1) / TEAMS * TEAMS;
  for (uint32_t rbase = blockIdx.x * TEAMS; rbase < nblk_rows; rbase += gridDim.x * TEAMS) {
    const uint32_t ridx = rbase + team; const bool live = ridx < a.nrows_bin;
    const uint32_t i = live ? a.rows[ridx] : 0;
    for (int s = t; s < SLOTS; s += TEAM) { key[s] = HASH_EMPTY; if constexpr (NUMERIC) s_acc[team][s] = idw; }
    if (t == 0) s_cnt[team] = 0;
    team_sync();
    const uint32_t ab = live ? a.arp[i] : 0, ae = live ? a.arp[i + 1] : 0;
    uint32_t mine = 0;
    for (uint32_t pa = ab + grp; pa < ae; pa += NG) {
      const uint32_t k = a.acol[pa]; const T av = (NUMERIC && use_a) ? aval[pa] : T();
      const uint32_t bb = a.brp[k], be = a.brp[k + 1];
      for (uint32_t pb = bb + lane16; pb < be; pb += 16) {
        const uint32_t j = a.bcol[pb];
        uint32_t h = hash_col(j, SLOTS - 1);
        for (;;) {
          const uint32_t old = atomicCAS(&key[h], HASH_EMPTY, j);
          if (old == HASH_EMPTY) { mine++; break; }
          if (old == j) break;
          h = (h + 1) & (SLOTS - 1);
        }
        if constexpr (NUMERIC) word_combine<T>(sr.add_op(), &s_acc[team][h], sr.mult(av, use_b ? bval[pb] : T()));
      }
    }
    if constexpr (!NUMERIC) {
      if (mine) atomicAdd(&s_cnt[team], mine);
      team_sync();
      if (live && t == 0) a.rownnz[i] = s_cnt[team];
    } else {
      team_sync();
      const uint32_t base = live ? a.crp[i] : 0;
      for (int s = t; s < SLOTS; s += TEAM) if (key[s] != HASH_EMPTY) {
        const uint32_t w = base + atomicAdd(&s_cnt[team], 1u);
        a.ccol[w] = key[s]; cval[w] = from_word<T>(s_acc[team][s]);
      }
    }
    team_sync();
  }
}

// rows beyond the LDS tables.  One persistent workgroup owns a bitmap of ncols bits (symbolic) or a dense accumulator of
// ncols words + the bitmap (numeric) in HBM and walks its rows one after the other.
template <class T, class SR, bool NUMERIC>
__global__ __launch_bounds__(1024) void k_spgemm_dense(const HashArgs a, const T* __restrict__ aval, const T* __restrict__ bval, T* __restrict__ cval, uint32_t ncols,
                                                       uint32_t* __restrict__ bitmaps, typename acc_word<T>::type* __restrict__ accs, const SR sr) {
  typedef typename acc_word<T>::type W;
  const uint32_t words = (ncols + 31) / 32;
  uint32_t* bits = bitmaps + (size_t)blockIdx.x * words;
  W* acc = NUMERIC ? accs + (size_t)blockIdx.x * ncols : nullptr;
  __shared__ uint32_t s_cnt;
  const int t = threadIdx.x, lane16 = t & 15, grp = t >> 4; constexpr int NG = 1024 / 16;
  const bool use_a = sr.uses_a(), use_b = sr.uses_u();
  const W idw = to_word<T>(sr.identity);
  for (uint32_t ridx = blockIdx.x; ridx < a.nrows_bin; ridx += gridDim.x) {
    const uint32_t i = a.rows[ridx];
    if (t == 0) s_cnt = 0;
    __syncthreads();
    const uint32_t ab = a.arp[i], ae = a.arp[i + 1], base = NUMERIC ? a.crp[i] : 0;
    uint32_t mine = 0;
    for (uint32_t pa = ab + grp; pa < ae; pa += NG) {
      const uint32_t k = a.acol[pa]; const T av = (NUMERIC && use_a) ? aval[pa] : T();
      const uint32_t bb = a.brp[k], be = a.brp[k + 1];
      for (uint32_t pb = bb + lane16; pb < be; pb += 16) {
        const uint32_t j = a.bcol[pb], bit = 1u << (j & 31);
        const bool fresh = !(__hip_atomic_load(&bits[j >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) && !(atomicOr(&bits[j >> 5], bit) & bit);
        if constexpr (NUMERIC) {
          if (fresh) a.ccol[base + atomicAdd(&s_cnt, 1u)] = j;            // first product of this column: it joins the row
          word_combine<T>(sr.add_op(), &acc[j], sr.mult(av, use_b ? bval[pb] : T()));
        } else if (fresh) mine++;
      }
    }
    if constexpr (!NUMERIC) { if (mine) atomicAdd(&s_cnt, mine); }
    __threadfence(); __syncthreads();
    if constexpr (!NUMERIC) {
      if (t == 0) a.rownnz[i] = s_cnt;
      // clear the bits again: walk the products once more (the row's columns are not stored anywhere yet)
      for (uint32_t pa = ab + grp; pa < ae; pa += NG) {
        const uint32_t k = a.acol[pa]; const uint32_t bb = a.brp[k], be = a.brp[k + 1];
        for (uint32_t pb = bb + lane16; pb < be; pb += 16) { const uint32_t j = a.bcol[pb]; bits[j >> 5] = 0; }
      }
    } else {
      const uint32_t n = s_cnt;
      for (uint32_t e = t; e < n; e += 1024) {                           // values out, accumulators and bits back to their rest state
        const uint32_t j = a.ccol[base + e];
        cval[base + e] = from_word<T>(__hip_atomic_load(&acc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); acc[j] = idw; bits[j >> 5] = 0;
      }
    }
    __threadfence(); __syncthreads();
  }
}
// ---- rows beyond the LDS tables, when the column range is moderate: a dense accumulator IN LDS, one block of columns at a time ----
// Round 3.  The HBM-resident dense accumulator above costs two global atomics per product (9.5e9 products of A@A on R-MAT-18:
// 0.73 s numeric + 0.27 s symbolic of a 1.1 s call), and its rows leave in claim order, so the whole result goes through a
// segmented sort.  Here a workgroup walks its row once per block of WD columns (16 384 eight-byte / 32 768 four-byte accumulators =
// 128 KiB of LDS): where the blocks of every B row begin is computed once per call (k_spa_split: the rows are sorted by column),
// the products of a block are dealt to the lanes of a wave whatever the lengths of the row parts are, they combine with LDS atomics, and the block is emitted by scanning its bitmap — in column order, straight into the
// result: no claim list, no sort, no gather for these rows.  The symbolic pass marks a bitmap of all ncols bits in LDS (<= 2^20
// columns).  Taken when ncols <= 64 blocks and <= 2^20 columns; the HBM path remains for wider results.
#ifndef SPA_WD8_V
#define SPA_WD8_V 16384
#endif
#ifndef SPA_WD4_V
#define SPA_WD4_V 28672
#endif
// LDS of the numeric kernel: WD accumulators + WD flag bytes + 16.1 KiB of walk state <= 160 KiB.  One workgroup per CU, but fewer, longer (row, block) steps
// win — A@A R-MAT-18: 4096 columns 0.252 s, 6144 0.227, 8192 0.210, 12288 0.190, 16384 0.184 (measurement builds: -DSPA_WD8_V=...; again in round 4 with the
// faster walk, two workgroups of 8192 columns per CU: numeric pass 57 -> 78 ms)
template <class T> struct spa_cfg { static constexpr uint32_t WD = sizeof(typename acc_word<T>::type) >= 8 ? (uint32_t)SPA_WD8_V : (uint32_t)SPA_WD4_V; };
constexpr uint32_t SPA_SYM_WORDS = 32768;      // 2^20 bits
constexpr uint32_t SPA_RANK_MAXBLK = 32;     // ranked rows (k_spgemm_spa_numeric): at most this many blocks of columns
constexpr uint32_t SPA_CHUNK = 992;             // entries of A(i,:) per walk of the numeric kernel (the last 32 threads carry none: their share of the walk's arrays is the room the flag bytes need)
// The products of up to 1024 entries k of A(i,:) (one per thread: `len` entries of B starting at `st`), dealt evenly to the 16
// waves of the workgroup whatever the lengths are — most are empty or a single entry, a hub's is tens of thousands: an exclusive
// scan of the lengths in LDS, the product range is cut into batches of up to SPA_R rounds of 64, and a lane finds the entry its
// product belongs to by a binary search over the scan.  load(v, pb): entry v of the chunk, position pb in B; apply(item).  (A group
// of 16 lanes per entry, as the table kernels do it, left this kernel waiting 96 % of its cycles: rows with a few dozen entries
// pointing at hub rows kept one group busy and 63 idle.)
// Round 4: a lane's products were one dependent chain each — ten LDS reads of the search, the B entry from the L2, the bitmap word, the
// atomic — and a wave walked its rounds one after the other: ~2 700 clocks per round of 64 products with nothing else in flight (A@A on
// R-MAT-18: 36 000 rounds per wave, 46 ms symbolic; the numeric pass 108 ms).  Now
//  (a) the rounds of a batch are searched, loaded and combined together — `load` fetches, `apply` combines: up to SPA_R independent
//      chains per lane, every load of the batch in one basic block (46 -> 25 ms, 108 -> 73 ms with four);
//  (b) the search runs over the entries the batch begins and ends in only — the entries of a hub's row cover whole rounds and need no
//      step at all.  Those two entries cost two LDS reads and four ballots: the sixteen wave totals sit in the lanes of a register (which
//      64-entry chunk), the chunk's 64 offsets are read one per lane (which entry);
//  (c) the batches go round the waves instead of every wave taking a sixteenth of the range: the first entries of a row of a power-law
//      graph are its hubs (long parts, no search, consecutive loads), the last ones its tail (a part per product) — thread 0's wave
//      spent 6 400 clocks per (row, block) step in its rounds and 4 500 waiting for the others (SPA_PROFILE build);
//  (d) the scans are DPP scans (six VALU steps instead of six trips through the LDS crossbar).
// Measured and dropped in round 4 (A@A R-MAT-18, numeric + symbolic pass in ms; this walk: 55.5 + 13.8):
//  * parts of up to eight entries handled by their own lane (no scan, no search), only the longer ones listed in LDS and dealt by ballots and
//    readlanes: 58-63 + 17-18 — the search's LDS chain became VALU work of the same length (SQ counters of that version: VALU 36 %, SALU 29 %,
//    LDS 37 % busy, a wave waiting 53 % of its cycles: no unit is the limit, the step's dependent chain is — four barriers, one memory latency,
//    the emission loop — and a (row, block) step has only ~5 150 products and ~1 600 results to hide it behind);
//  * reading the batch's bitmap words together before its atomics: 57 -> 61 (later rounds no longer see the bits the earlier ones set);
//  * 512 threads per workgroup (half the fixed per-wave work of a step): 63 -> 75 + 30; two workgroups of 8192 columns per CU: 57 -> 78;
//  * barriers that wait for the LDS only (s_waitcnt lgkmcnt(0); s_barrier — not for the stores on their way to the HBM): no change.
//  * the emission a LANE per column (a wave walks its 1024 columns 64 at a time; ballot + v_mbcnt ranks: the lanes of a store write consecutive
//    entries) instead of 16 columns per thread: 55 -> 59.5 (the scattered stores were not what the emission costs).
#ifndef SPA_R_V
#define SPA_R_V 8
#endif
#ifdef SPA_PROFILE
// measurement build (make BUILD=build_spa LIB=../libgrb_spa.so XTFLAGS=-DSPA_PROFILE): clocks of thread 0 of every workgroup of the last numeric launch by phase —
// [0] scan of the lengths + two barriers, [1] its rounds, [2] waiting for the other waves' rounds, [3] scan of the bitmap, [4] emission, [5] the last barrier,
// [6] (row, block) steps with products, [7] the kernel, [8] steps without products.  tools/spa_phase_probe.py reads them.
static __device__ unsigned long long g_spa_prof[1024 * 16];
#define SPA_PF(K) { if (pf) { const unsigned long long pf_t = __builtin_amdgcn_s_memtime(); pf[K] += pf_t - pf[15]; pf[15] = pf_t; } }
#else
#define SPA_PF(K)
#endif
constexpr int SPA_R = SPA_R_V;
__device__ __forceinline__ uint32_t spa_wave_incl_add(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);      // row_shr:1 (lanes without a source add 0)
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
  return v;
}
// the sixteen per-wave counts in s_w[16] (written before the last barrier): their sum, and the sum of those before `wave`; the exclusive
// offset of chunk (lane & 15) is left in `pre`
__device__ __forceinline__ void spa_wave_offsets(const uint32_t* s_w, uint32_t lane, uint32_t wave, uint32_t& woff, uint32_t& total, uint32_t& pre) {
  const uint32_t wt = s_w[lane & 15];
  uint32_t v = wt;                                                                       // inclusive scan inside every row of 16 lanes
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
  pre = v - wt;
  total = (uint32_t)__builtin_amdgcn_readlane((int)v, 15);
  woff = (uint32_t)__builtin_amdgcn_readlane((int)pre, (int)wave);
}
// `ordered` (round 5, the library's deterministic mode — GRB_MI355X_DETERMINISTIC=1 or the descriptor's GxB_AxB_GUSTAVSON): the batches combine ONE AFTER THE
// OTHER, in batch order, each behind a ticket the batch before it hands on — the products of a batch are dealt to lanes and rounds statically and a wave's LDS
// atomics execute in program order, so every accumulator receives its terms in the same order in every run and a floating-point sum is reproducible bit
// for bit (A@A on R-MAT-18: see DESIGN.md).  Without it the waves' atomics land as they come.
// A product is combined in two steps: slot_of(item) finds its accumulator (for a ranked row three LDS reads and a bit count — nothing that depends on the other
// products), commit(slot, item) is the atomic.  The ordered walk computes the slots of a batch BEFORE it waits for its turn: only the atomics are serialised.
template <bool ordered = false, class L, class S, class A> __device__ __forceinline__ uint32_t spa_flat_walk2(uint32_t st, uint32_t len, uint32_t* s_exc /* [1025] */, uint32_t* s_shift /* [1024] */, uint32_t* s_wtot /* [16] */, L&& load, S&& slot_of, A&& commit, unsigned long long* pf = nullptr, uint32_t* s_turn = nullptr) {
  const uint32_t t = threadIdx.x, lane = t & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6));
  const uint32_t inc = spa_wave_incl_add(len);
  if (lane == 63) s_wtot[wave] = inc;
  if constexpr (ordered) { if (t == 0) *s_turn = 0; }      // the batch whose turn it is to combine
  __syncthreads();
  uint32_t woff, total, cpre;                                // cpre: first product of the 64-entry chunk (lane & 15)
  spa_wave_offsets(s_wtot, lane, wave, woff, total, cpre);
  const uint32_t exc = woff + inc - len;
  s_exc[t] = exc; if (len) s_shift[t] = st - exc;          // (an empty part is never the answer of a search)
  if (t == 0) s_exc[1024] = total;
  __syncthreads();
  SPA_PF(0)
  // entry(q) = the last entry whose exclusive offset is <= q (the entries before it with the same offset are empty)
  auto entry_of = [&](uint32_t q) __attribute__((always_inline)) -> uint32_t {
    const uint32_t chunk = (uint32_t)__popc((uint32_t)__ballot(cpre <= q) & 0xFFFFu) - 1u;          // (chunk 0 begins at 0 <= q)
    const uint32_t x = s_exc[chunk * 64u + lane];
    return chunk * 64u + (uint32_t)__popcll(__ballot(x <= q)) - 1u;                                  // (the chunk's first offset is <= q)
  };
  uint32_t rpb = (total + 1023u) >> 10; rpb = rpb < 1u ? 1u : (rpb > (uint32_t)SPA_R ? (uint32_t)SPA_R : rpb);      // rounds per batch: one batch per wave while that fits
  const uint32_t bsz = 64u * rpb, nb = (total + bsz - 1u) / bsz;
  for (uint32_t b0 = 0; b0 < nb; b0 += 16u) {
    const uint32_t b = b0 + wave;
    if (b >= nb) break;                                      // (its batches are b = wave, wave + 16, ...: none left)
    const uint32_t qb = b * bsz, qe = qb + bsz < total ? qb + bsz : total;
    const uint32_t vbase = entry_of(qb), vlast = entry_of(qe - 1u);
    uint32_t q[SPA_R], vlo[SPA_R], vhi[SPA_R];
#pragma unroll
    for (int r = 0; r < SPA_R; r++) { q[r] = qb + 64u * r + lane; vlo[r] = vbase; vhi[r] = vlast + 1; }      // exc[vbase] <= q < exc[vlast + 1] for every live q
    const uint32_t span = vlast - vbase;
    const int steps = span ? 32 - __builtin_clz(span) : 0;                                                    // span + 1 candidates; wave-uniform
    for (int s2 = 0; s2 < steps; s2++) {
#pragma unroll
      for (int r = 0; r < SPA_R; r++) { const uint32_t mid = (vlo[r] + vhi[r]) >> 1; if (s_exc[mid] <= q[r]) vlo[r] = mid; else vhi[r] = mid; }
    }
    uint32_t pb[SPA_R];
#pragma unroll
    for (int r = 0; r < SPA_R; r++) { const uint32_t sh = s_shift[vlo[r]]; pb[r] = q[r] < qe ? sh + q[r] : 0u; }      // (lanes behind the batch load B's entry 0 — every load of the batch in one basic block — and drop it)
    decltype(load(0u, 0u)) item[SPA_R];
#pragma unroll
    for (int r = 0; r < SPA_R; r++) item[r] = load(vlo[r], pb[r]);
    // (the bitmap words are read round by round, not together before the batch's atomics: a hub's part sets the bits of its 32-column
    //  words in its first round and the later rounds see them — reading all of them first meant more same-word atomics: 56.9 -> 61.1 ms)
    if constexpr (!ordered) {
#pragma unroll
      for (int r = 0; r < SPA_R; r++) if (q[r] < qe) commit(slot_of(item[r]), item[r]);
    } else {
      uint32_t slot[SPA_R];
#pragma unroll
      for (int r = 0; r < SPA_R; r++) slot[r] = q[r] < qe ? slot_of(item[r]) : 0u;
      // batch b combines when the batches before it have: a ticket in LDS instead of sixteen barriers per group of batches — a wave goes on to search and load
      // its next batch while the others combine (the LDS executes a wave's operations in order: the ticket a wave hands on is behind its atomics)
      while (__hip_atomic_load(s_turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != b) __builtin_amdgcn_s_sleep(1);
#pragma unroll
      for (int r = 0; r < SPA_R; r++) if (q[r] < qe) commit(slot[r], item[r]);
      if (lane == 0) __hip_atomic_store(s_turn, b + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  SPA_PF(1)
  __syncthreads();
  SPA_PF(2)
  return total;                                             // (the same in every thread)
}
template <bool ordered = false, class L, class A> __device__ __forceinline__ uint32_t spa_flat_walk(uint32_t st, uint32_t len, uint32_t* s_exc, uint32_t* s_shift, uint32_t* s_wtot, L&& load, A&& apply, unsigned long long* pf = nullptr) {
  static_assert(!ordered, "the ordered walk takes slot_of / commit (spa_flat_walk2)");
  return spa_flat_walk2<false>(st, len, s_exc, s_shift, s_wtot, load, [](const auto&) { return 0u; }, [&](uint32_t, const auto& it) { apply(it); }, pf, nullptr);
}
// the next row of a persistent workgroup of the dense paths, handed out by a counter (round 5).  A static deal — row b, b + grid, ... — left the slowest
// workgroup of the numeric pass of A@A on R-MAT-18 at 1.48 x the mean (rows of 6 000 and of 3 000 000 products), and the launch ends with it.  The counter is
// drawn at the START of a row and looked at behind its end, so its latency is never waited for.
__device__ __forceinline__ uint32_t spa_next_row(uint32_t* s_next, uint32_t drawn) {
  if (threadIdx.x == 0) *s_next = drawn;
  __syncthreads();
  const uint32_t r = *s_next;
  __syncthreads();
  return r;
}
template <class T, class SR>
__global__ __launch_bounds__(1024) void k_spgemm_spa_symbolic(const HashArgs a, uint32_t ncols, uint32_t* __restrict__ rowctr, uint32_t* __restrict__ bitmaps = nullptr, uint32_t* __restrict__ bmslot = nullptr) {
  __shared__ uint32_t s_bits[SPA_SYM_WORDS];
  __shared__ uint32_t s_exc[1025], s_shift[1024], s_wtot[16];
  __shared__ uint32_t s_cnt, s_next;
  const uint32_t words = (ncols + 31) / 32;
  const uint32_t t = threadIdx.x;
  for (uint32_t w = t; w < words; w += 1024) s_bits[w] = 0;
  if (t == 0) s_cnt = 0;
  __syncthreads();
  uint32_t drawn = 0;
  for (uint32_t ridx = blockIdx.x; ridx < a.nrows_bin; ridx = spa_next_row(&s_next, drawn)) {
    if (t == 0) drawn = gridDim.x + atomicAdd(rowctr, 1u);
    const uint32_t i = a.rows[ridx];
    const uint32_t ab = a.arp[i], ae = a.arp[i + 1];
    for (uint32_t base = ab; base < ae; base += 1024) {
      const uint32_t pa = base + t; uint32_t st = 0, len = 0;
      if (pa < ae) { const uint32_t k = a.acol[pa]; st = a.brp[k]; len = a.brp[k + 1] - st; }
      spa_flat_walk(st, len, s_exc, s_shift, s_wtot, [&](uint32_t, uint32_t pb) { return a.bcol[pb]; },
                    [&](uint32_t j) { const uint32_t bit = 1u << (j & 31); if (!(s_bits[j >> 5] & bit)) atomicOr(&s_bits[j >> 5], bit); });
    }
    uint32_t c = 0;
    // (round 5: the row's bitmap is kept for the numeric pass — 32 KiB per row at 2^18 columns — which ranks a column among the row's entries with it)
    if (bitmaps) { uint32_t* const bm = bitmaps + (size_t)ridx * words; for (uint32_t w = t; w < words; w += 1024) { const uint32_t x = s_bits[w]; c += __popc(x); bm[w] = x; s_bits[w] = 0; } }
    else for (uint32_t w = t; w < words; w += 1024) { c += __popc(s_bits[w]); s_bits[w] = 0; }
    for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((t & 63) == 0 && c) atomicAdd(&s_cnt, c);
    __syncthreads();
    if (t == 0) { a.rownnz[i] = s_cnt; s_cnt = 0; if (bmslot) bmslot[i] = ridx; }
    __syncthreads();
  }
}
// where block c of row k of B begins: split[k * (nblk + 1) + c] = first position of B(k,:) whose column is >= c * WD (one wave per row;
// the rows are sorted by column, so every boundary is written exactly once)
static __global__ void k_spa_split(uint32_t nrows, const uint32_t* __restrict__ brp, const uint32_t* __restrict__ bcol, uint32_t wd, uint32_t nblk, uint32_t* __restrict__ split) {
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t k = (blockIdx.x * 256u + threadIdx.x) >> 6; k < nrows; k += gridDim.x * 4u) {
    const uint32_t bb = brp[k], be = brp[k + 1];
    uint32_t* sp = split + (size_t)k * (nblk + 1);
    for (uint32_t p = bb + lane; p < be; p += 64) {
      const uint32_t blk = bcol[p] / wd, first = p == bb ? 0u : bcol[p - 1] / wd + 1;
      for (uint32_t c = first; c <= blk; c++) sp[c] = p;
    }
    const uint32_t last = be > bb ? bcol[be - 1] / wd + 1 : 0u;
    for (uint32_t c = last + lane; c <= nblk; c += 64) sp[c] = be;
  }
}
template <class T, class SR, bool ORDERED = false>
__global__ __launch_bounds__(1024) void k_spgemm_spa_numeric(const HashArgs a, const T* __restrict__ aval, const T* __restrict__ bval, uint32_t* __restrict__ ocol, T* __restrict__ oval,
                                                             uint32_t ncols, const uint32_t* __restrict__ split, const SR sr, uint32_t* __restrict__ rowctr,
                                                             const uint32_t* __restrict__ bitmaps = nullptr, const uint32_t* __restrict__ bmslot = nullptr) {
  typedef typename acc_word<T>::type W;
  constexpr uint32_t WD = spa_cfg<T>::WD;
  // One region, two uses (round 5).  DIRECT: WD accumulators + one flag byte per column of the block: 1 = the accumulator holds a product (round 4;
  // a bitmap before: a read, a test and an atomic OR per product — an LDS latency in every combine, eight of them one after the other per batch.  A byte
  // is a plain store).  RANKED: the row's bitmap of all ncols bits (kept by the symbolic pass), a 16-bit exclusive bit count per bitmap word (relative
  // to the wave's chunk of words) and accumulators indexed by a column's RANK among the row's entries — a row of <= W2 entries is ONE step instead of
  // one per block of columns (see "ranked rows" below).
  constexpr uint32_t REGION = WD * (uint32_t)sizeof(W) + WD;
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[REGION];
  W* const s_acc = (W*)s_raw;
  uint32_t* const s_flag = (uint32_t*)(s_raw + (size_t)WD * sizeof(W));
  __shared__ T s_av[SPA_CHUNK];
  __shared__ uint32_t s_exc[1025], s_shift[SPA_CHUNK], s_wtot[16];
  __shared__ uint32_t s_wsum[16], s_woff[16], s_blk[SPA_RANK_MAXBLK + 2], s_next, s_turn;
  const uint32_t t = threadIdx.x, lane = t & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(t >> 6));
  const bool use_a = sr.uses_a(), use_b = sr.uses_u();
  const W idw = to_word<T>(sr.identity);
#ifdef SPA_PROFILE
  unsigned long long pfa[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long* const pf = pfa; const unsigned long long pf_start = __builtin_amdgcn_s_memtime(); pf[15] = pf_start;
#else
  unsigned long long* const pf = nullptr;
#endif
  const uint32_t nblk = (uint32_t)(((uint64_t)ncols + WD - 1) / WD);
  // ---- ranked rows (round 5) ----------------------------------------------------------------------------------------------------------
  // A (row, block) step costs ~18 000 clocks whatever it carries (four barriers, a memory latency, the emission), and 73 % of the rows beyond the
  // tables of A@A on R-MAT-18 have <= 32 768 entries spread over all 16 blocks: ~1 000 products per step.  With the row's bitmap in LDS a column's
  // place among the row's entries is  rank(j) = bits set below j  = chunk offset + 16-bit word prefix + popcount of the word's low bits — three LDS
  // reads — and accumulators indexed by rank need as many slots as the step has ENTRIES, not columns: consecutive blocks are grouped while
  // their entries fit the W2 accumulators the bitmap leaves room for (12 288 eight-byte ones at 2^18 columns), a row of <= W2 entries is one step,
  // and the emission needs no scan (the ranks ARE the output positions).  A row one of whose blocks alone holds more than W2 entries (the hub
  // rows: dense, many products per step already) takes the direct path.  The B-row boundaries are those of k_spa_split: a product is visited once.
  const bool rank_call = bitmaps != nullptr;
  const uint32_t words = (ncols + 31u) >> 5;
  // the slab of a ranked row: the whole row when its bitmap and word prefixes (6 bytes per word) fit 2/5 of the region — and 8 words per thread — else as
  // many whole blocks as do
  constexpr uint32_t WBc = WD / 32u;
  constexpr uint32_t slab_fit = (uint32_t)(((uint64_t)REGION * 2 / 5) / 6) < 8192u ? (uint32_t)(((uint64_t)REGION * 2 / 5) / 6) : 8192u;
  const uint32_t slabw = words <= slab_fit ? ((words + WBc - 1) / WBc) * WBc : (slab_fit / WBc) * WBc;      // (a multiple of a block's words; >= words when the row is one slab)
  uint32_t wpt = 1, csh = 6; while (wpt * 1024u < (words < slabw ? words : slabw)) { wpt <<= 1; csh++; }         // bitmap words of a slab per thread (<= 8), log2 of the words per wave
  const uint32_t swords = words < slabw ? words : slabw;
  const uint32_t rk_off = (swords * 6u + 15u) & ~15u;
  uint32_t* const s_bits = (uint32_t*)s_raw; uint16_t* const s_pre = (uint16_t*)(s_raw + (size_t)swords * 4);
  W* const s_racc = (W*)(s_raw + rk_off); const uint32_t W2 = rank_call ? (REGION - rk_off) / (uint32_t)sizeof(W) : 0u;
  constexpr uint32_t WB = WD / 32u; static_assert(WD % 32u == 0, "a block of columns is whole bitmap words");
  constexpr uint32_t CMAX = ((REGION - 5120u) / (8u + (uint32_t)sizeof(W))) & ~7u;      // entries of a row of few entries (the compact rank structure below): 8 B per non-empty word + an accumulator each
  static_assert(CMAX < 65536u && 8u * CMAX + 5120u + CMAX * sizeof(W) <= REGION, "ranks and slots are 16-bit; the structure fits the region");
  static_assert(SPA_SYM_WORDS / 64u * 8u <= 4096u && SPA_SYM_WORDS / 64u * 2u <= 1024u && SPA_SYM_WORDS <= 65536u, "the chunk index of the compact rank structure (4096 B of masks, 1024 B of first slots, 16-bit word numbers) covers every column range the dense path takes");
  bool racc_clean = false;
  if (!rank_call) {
    for (uint32_t e = t; e < WD; e += 1024) s_acc[e] = idw;
    for (uint32_t w = t; w < WD / 4; w += 1024) s_flag[w] = 0;
    __syncthreads();
  }
  uint32_t drawn = 0;
  for (uint32_t ridx = blockIdx.x; ridx < a.nrows_bin; ridx = spa_next_row(&s_next, drawn)) {
    if (t == 0) drawn = gridDim.x + atomicAdd(rowctr, 1u);
    const uint32_t i = a.rows[ridx];
    const uint32_t ab = a.arp[i], ae = a.arp[i + 1];
    uint32_t obase = a.crp[i];
    // one chunk of the row's entries (their A values in s_av) against block c (columns lo ...): returns the number of products
    struct Prod { uint32_t rel; T x; };
    auto walk = [&](uint32_t lo, uint32_t st, uint32_t len) __attribute__((always_inline)) -> uint32_t {
      return spa_flat_walk2<ORDERED>(st, len, s_exc, s_shift, s_wtot,
        [&](uint32_t v, uint32_t pb) { Prod p; p.rel = a.bcol[pb] - lo; p.x = sr.mult(use_a ? s_av[v] : T(), use_b ? bval[pb] : T()); return p; },
        [&](const Prod& p) -> uint32_t { ((unsigned char*)s_flag)[p.rel] = 1; return p.rel; },
        [&](const uint32_t rel, const Prod& p) { word_combine<T>(sr.add_op(), &s_acc[rel], p.x); }, pf, &s_turn);
    };
    // emit the block in column order: every thread owns WD / 1024 columns; exclusive prefix of their counts, then every thread writes its own run —
    // four accumulators in flight.  (Round 4: 32 columns on half the threads, thread 0 adding up the wave sums between two barriers and a
    // barrier behind the emission took 6 500 of a step's 19 700 clocks; the next walk's two barriers already stand between this emission's
    // resets and the next block's atomics.)
    auto emit = [&](uint32_t lo) __attribute__((always_inline)) {
      constexpr uint32_t BPT = WD / 1024u; static_assert(BPT % 4u == 0 && BPT <= 32u && BPT * 1024u == WD, "a thread emits up to 32 columns, whole flag words");
      uint32_t mybits = 0;
#pragma unroll
      for (uint32_t w = 0; w < BPT / 4u; w++) {                                  // four flag bytes (0 / 1) -> four bits: the partial products of the multiply land on distinct bits, the wanted ones on 24..27
        const uint32_t f = s_flag[t * (BPT / 4u) + w];
        mybits |= ((f * 0x01020408u) >> 24) << (4u * w);
      }
      const uint32_t mycnt = (uint32_t)__popc(mybits);
      const uint32_t inc = spa_wave_incl_add(mycnt);
      if (lane == 63) s_wsum[wave] = inc;
      __syncthreads();
      uint32_t woff, etotal, unused;
      spa_wave_offsets(s_wsum, lane, wave, woff, etotal, unused);
      SPA_PF(3)
      if (mybits) {
        uint32_t zero; asm volatile("v_mov_b32 %0, 0" : "=v"(zero));            // (made here: hoisted out of the loops the compiler spilled the zeros to scratch and reloaded them for every block)
#pragma unroll
        for (uint32_t w = 0; w < BPT / 4u; w++) s_flag[t * (BPT / 4u) + w] = zero;
      }
      uint32_t o = obase + woff + inc - mycnt;
      const uint32_t rel0 = t * BPT;
      while (mybits) {
        uint32_t bb[4]; bool has[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { has[j] = mybits != 0; bb[j] = has[j] ? (uint32_t)__builtin_ctz(mybits) : 0u; mybits &= mybits - 1u; }
        W acc4[4];
#pragma unroll
        for (int j = 0; j < 4; j++) acc4[j] = s_acc[rel0 + bb[j]];
#pragma unroll
        for (int j = 0; j < 4; j++) if (has[j]) { ocol[o] = lo + rel0 + bb[j]; oval[o] = from_word<T>(acc4[j]); s_acc[rel0 + bb[j]] = idw; o++; }
      }
      obase += etotal;
      SPA_PF(4)
    };
    // the blocks [cb0, cb1) of the row through the direct accumulators (the region holds WD accumulators + flag bytes)
    auto direct_range = [&](uint32_t cb0, uint32_t cb1) __attribute__((always_inline)) {
    if (ae - ab <= SPA_CHUNK) {
      // a row of at most SPA_CHUNK entries — nearly all of them — is one chunk: its B rows and A values stay in place over the blocks, and the
      // boundary of the block after the next is loaded while this block is worked on (consecutive blocks share a boundary).  `nxt` is only
      // looked at a whole step later; loaded into the variable the step itself reads — or under a branch, or in a loop shared with the
      // long rows' path — the compiler copied it into place, and waited for it, on the spot: a memory latency per step.
      const bool has = t < SPA_CHUNK && ab + t < ae;
      const uint32_t* const sp0 = has ? split + (size_t)a.acol[ab + t] * (nblk + 1) : split;
      uint32_t cur_st = sp0[cb0], cur_en = sp0[cb0 + 1], nxt = sp0[cb0 + 2 <= nblk ? cb0 + 2 : nblk];
      if (has && use_a) s_av[t] = aval[ab + t];                                  // (read by other threads only behind the walk's first barrier)
      for (uint32_t c = cb0; c < cb1; c++) {
        const uint32_t st = cur_st, len = has ? cur_en - cur_st : 0u;
        cur_st = cur_en; cur_en = nxt;
        nxt = sp0[c + 3 <= nblk ? c + 3 : nblk];
        const uint32_t products = walk(c * WD, st, len);
#ifdef SPA_PROFILE
        pf[products ? 6 : 8]++;
#endif
        if (products) emit(c * WD);                         // (else nothing of this row falls into this block — the whole workgroup agrees: no emission, no barrier)
      }
    } else {
      for (uint32_t c = cb0; c < cb1; c++) {
        uint32_t products = 0;
        for (uint32_t base = ab; base < ae; base += SPA_CHUNK) {
          const uint32_t pa = base + t;
          uint32_t st = 0, len = 0;
          if (t < SPA_CHUNK && pa < ae) {
            const uint32_t* sp = split + (size_t)a.acol[pa] * (nblk + 1) + c;
            st = sp[0]; len = sp[1] - st; if (use_a) s_av[t] = aval[pa];        // (read by other threads only behind the walk's first barrier)
          }
          products |= walk(c * WD, st, len);
        }
#ifdef SPA_PROFILE
        pf[products ? 6 : 8]++;
#endif
        if (products) emit(c * WD);
      }
    }
    };
    if (rank_call) {
      // Round 6: the row in SLABS of `slabw` bitmap words (<= 2^18 columns: what leaves room for accumulators beside it) — the whole row when it fits, as
      // in round 5.  A slab's bitmap, word prefixes and block boundaries are loaded like the row's were; ranks are relative to the slab, its entries
      // follow those of the slabs before it in the result.  A slab without an entry costs one load of its words and no step; a slab one of whose
      // blocks alone holds more than W2 entries goes block by block (the region re-initialised around it).  So a result of 2^20 columns takes ~4 steps
      // per row instead of 64 (A@A on R-MAT-20 with edge factor 4: 0.19 s — 7.6 % of the roofline — with the block-by-block path for every row).
      const uint32_t* const bm = bitmaps + (size_t)bmslot[i] * words;
      const uint32_t rowent = a.crp[i + 1] - obase;
      const bool one_chunk = ae - ab <= SPA_CHUNK;
      const bool has = one_chunk && t < SPA_CHUNK && ab + t < ae;
      const uint32_t* const sp0 = has ? split + (size_t)a.acol[ab + t] * (nblk + 1) : split;
      // ---- a row of FEW entries in a wide result (round 6): ONE step whatever the column range -----------------------------------------------------
      // All 2^20-column rows of A@A on R-MAT-20 / edge factor 4 touch every slab, and 63 % of them hold <= 8 832 entries (11 % of the products): four slab
      // steps of ~10 us for a handful of products each.  Their rank structure need not scale with the columns: only the NON-EMPTY bitmap words are kept —
      // c_bits / c_pre (entries before the word) / c_widx (which word), in word order — behind an index per chunk of 64 words (a 64-bit mask of its non-empty
      // words + the slot of its first): rank(j) = c_pre[slot] + popcount(c_bits[slot] below j) with slot = first[chunk] + popcount(mask below the word) —
      // four LDS reads instead of three — and the accumulators, indexed by rank, need as many slots as the ROW has entries.
      if (words > slabw && rowent <= CMAX && rowent) {
        __syncthreads();                                                         // (whatever used the region before is done with it)
        uint32_t* const c_bits = (uint32_t*)s_raw; uint16_t* const c_pre = (uint16_t*)(s_raw + 4u * CMAX); uint16_t* const c_widx = (uint16_t*)(s_raw + 6u * CMAX);
        unsigned long long* const c_mask = (unsigned long long*)(s_raw + 8u * CMAX); uint16_t* const c_first = (uint16_t*)(s_raw + 8u * CMAX + 4096u);
        W* const c_acc = (W*)(s_raw + 8u * CMAX + 5120u);
        for (uint32_t e = t; e < rowent; e += 1024) c_acc[e] = idw;
        uint32_t slot_base = 0, ent_base = 0;
        for (uint32_t base = 0; base < words; base += 8192u) {
          uint32_t xs[8], nzm = 0, nbits = 0;
#pragma unroll
          for (uint32_t q = 0; q < 8; q++) { const uint32_t w = base + t * 8u + q; xs[q] = w < words ? bm[w] : 0u; nzm |= xs[q] ? (1u << q) : 0u; nbits += (uint32_t)__popc(xs[q]); }
          const uint32_t nzw = (uint32_t)__popc(nzm);
          const uint32_t inc_w = spa_wave_incl_add(nzw), inc_b = spa_wave_incl_add(nbits);
          if (lane == 63) { s_wsum[wave] = inc_w; s_woff[wave] = inc_b; }
          __syncthreads();
          uint32_t woff_w, tot_w, pre_w, woff_b, tot_b, pre_b;
          spa_wave_offsets(s_wsum, lane, wave, woff_w, tot_w, pre_w);
          spa_wave_offsets(s_woff, lane, wave, woff_b, tot_b, pre_b);
          uint32_t slot = slot_base + woff_w + inc_w - nzw, ent = ent_base + woff_b + inc_b - nbits;
          const uint32_t chunk = (base >> 6) + (t >> 3);
          if (base + t * 8u < words) { ((unsigned char*)c_mask)[chunk * 8u + (t & 7u)] = (unsigned char)nzm; if ((t & 7u) == 0) c_first[chunk] = (uint16_t)slot; }
#pragma unroll
          for (uint32_t q = 0; q < 8; q++) if (xs[q]) { c_bits[slot] = xs[q]; c_pre[slot] = (uint16_t)ent; c_widx[slot] = (uint16_t)(base + t * 8u + q); slot++; ent += (uint32_t)__popc(xs[q]); }
          slot_base += tot_w; ent_base += tot_b;
          __syncthreads();                                                       // (the wave totals are read: the next round may overwrite them)
        }
        const uint32_t nzwords = slot_base;
        struct CProd { uint32_t col; T x; };
        auto cwalk = [&](uint32_t st, uint32_t len) __attribute__((always_inline)) -> uint32_t {
          return spa_flat_walk2<ORDERED>(st, len, s_exc, s_shift, s_wtot,
            [&](uint32_t v, uint32_t pb) { CProd p; p.col = a.bcol[pb]; p.x = sr.mult(use_a ? s_av[v] : T(), use_b ? bval[pb] : T()); return p; },
            [&](const CProd& p) -> uint32_t { const uint32_t w = p.col >> 5, ch = w >> 6; const unsigned long long m = c_mask[ch];
                                  const uint32_t sl = (uint32_t)c_first[ch] + (uint32_t)__popcll(m & ((1ull << (w & 63u)) - 1ull));
                                  return (uint32_t)c_pre[sl] + (uint32_t)__popc(c_bits[sl] & ((1u << (p.col & 31u)) - 1u)); },
            [&](const uint32_t rk, const CProd& p) { word_combine<T>(sr.add_op(), &c_acc[rk], p.x); }, nullptr, &s_turn);
        };
        if (one_chunk) { if (has && use_a) s_av[t] = aval[ab + t]; cwalk(has ? sp0[0] : 0u, has ? sp0[nblk] - sp0[0] : 0u); }
        else for (uint32_t base = ab; base < ae; base += SPA_CHUNK) {
          const uint32_t pa = base + t; uint32_t st = 0, len = 0;
          if (t < SPA_CHUNK && pa < ae) { const uint32_t* sp = split + (size_t)a.acol[pa] * (nblk + 1); st = sp[0]; len = sp[nblk] - st; if (use_a) s_av[t] = aval[pa]; }
          cwalk(st, len);
        }
        // the entries in column order: the non-empty words are in word order, a word's first entry has rank c_pre
        for (uint32_t sl = t; sl < nzwords; sl += 1024) {
          uint32_t bits = c_bits[sl], rk = c_pre[sl]; const uint32_t c0 = (uint32_t)c_widx[sl] * 32u;
          while (bits) { const uint32_t bb = (uint32_t)__builtin_ctz(bits); bits &= bits - 1u; ocol[obase + rk] = c0 + bb; oval[obase + rk] = from_word<T>(c_acc[rk]); rk++; }
        }
        racc_clean = false;
        continue;
      }
      for (uint32_t w_base = 0, cb0 = 0; w_base < words; w_base += slabw, cb0 += slabw / WB) {
      const uint32_t words_s = words - w_base < slabw ? words - w_base : slabw;
      const uint32_t nblk_s = nblk - cb0 < slabw / WB ? nblk - cb0 : slabw / WB;
      __syncthreads();                                                           // (the slab / the row before this one has read its bitmap to the end)
      uint32_t xs[8], run = 0;
#pragma unroll
      for (uint32_t q = 0; q < 8; q++) { const uint32_t w = t * wpt + q; xs[q] = (q < wpt && w < words_s) ? bm[w_base + w] : 0u; }
      uint32_t pl[8];
#pragma unroll
      for (uint32_t q = 0; q < 8; q++) { pl[q] = run; run += (uint32_t)__popc(xs[q]); }
      const uint32_t inc = spa_wave_incl_add(run), wexc = inc - run;
#pragma unroll
      for (uint32_t q = 0; q < 8; q++) { const uint32_t w = t * wpt + q; if (q < wpt && w < words_s) { s_bits[w] = xs[q]; s_pre[w] = (uint16_t)(wexc + pl[q]); } }
      if (lane == 63) s_wsum[wave] = inc;
      __syncthreads();
      uint32_t woff, rtotal, cpre;
      spa_wave_offsets(s_wsum, lane, wave, woff, rtotal, cpre);
      if (t < 16) s_woff[t] = cpre;                                              // entries of the slab before wave t's words
      __syncthreads();
      if (rtotal == 0) continue;                                                 // (the whole workgroup agrees: nothing of this row falls into the slab)
      if (t <= nblk_s) { const uint32_t w0 = t * WB; s_blk[t] = w0 < words_s ? s_woff[w0 >> csh] + s_pre[w0] : rtotal; }      // entries before block t of the slab
      __syncthreads();
      bool ranked = true;
      for (uint32_t c = 0; c < nblk_s; c++) ranked = ranked && s_blk[c + 1] - s_blk[c] <= W2;
      if (ranked) {
        if (!racc_clean) { for (uint32_t e = t; e < W2; e += 1024) s_racc[e] = idw; racc_clean = true; }      // (the walk's barriers stand between this and the first atomic)
        struct RProd { uint32_t col; T x; };
        uint32_t rbase = 0;
        auto rwalk = [&](uint32_t st, uint32_t len) __attribute__((always_inline)) -> uint32_t {
          return spa_flat_walk2<ORDERED>(st, len, s_exc, s_shift, s_wtot,
            [&](uint32_t v, uint32_t pb) { RProd p; p.col = a.bcol[pb]; p.x = sr.mult(use_a ? s_av[v] : T(), use_b ? bval[pb] : T()); return p; },
            [&](const RProd& p) -> uint32_t { const uint32_t w = (p.col >> 5) - w_base; const uint32_t bits = s_bits[w];
                                  return s_woff[w >> csh] + s_pre[w] + (uint32_t)__popc(bits & ((1u << (p.col & 31u)) - 1u)) - rbase; },
            [&](const uint32_t rk, const RProd& p) { word_combine<T>(sr.add_op(), &s_racc[rk], p.x); }, nullptr, &s_turn);
        };
        // the entries of the slab's blocks [c0, c1): the threads walk their own bitmap words, the rank of a word's first bit is known — no scan, no barrier
        auto remit = [&](uint32_t c0, uint32_t c1) __attribute__((always_inline)) {
          const uint32_t wlo = c0 * WB, whi = c1 * WB < words_s ? c1 * WB : words_s;
          for (uint32_t q = 0; q < wpt; q++) {
            const uint32_t w = t * wpt + q;
            if (w < wlo || w >= whi) continue;
            uint32_t bits = s_bits[w];
            uint32_t rk = s_woff[w >> csh] + s_pre[w];
            while (bits) {
              uint32_t bb[4]; bool hs4[4];
#pragma unroll
              for (int j = 0; j < 4; j++) { hs4[j] = bits != 0; bb[j] = hs4[j] ? (uint32_t)__builtin_ctz(bits) : 0u; bits &= bits - 1u; }
              W acc4[4];
#pragma unroll
              for (int j = 0; j < 4; j++) acc4[j] = s_racc[(hs4[j] ? rk + j : rk) - rbase];
#pragma unroll
              for (int j = 0; j < 4; j++) if (hs4[j]) { ocol[obase + rk] = (w_base + w) * 32u + bb[j]; oval[obase + rk] = from_word<T>(acc4[j]); s_racc[rk - rbase] = idw; rk++; }
            }
          }
        };
        if (has && use_a) s_av[t] = aval[ab + t];                                // (read by other threads only behind the walk's first barrier)
        uint32_t c0 = 0, c1 = 1;
        while (c1 < nblk_s && s_blk[c1 + 1] - s_blk[c0] <= W2) c1++;
        uint32_t cur_st = sp0[cb0], cur_en = sp0[cb0 + c1];
        while (c0 < nblk_s) {
          uint32_t n1 = c1 < nblk_s ? c1 + 1 : c1;                               // the group after this one, its boundary loaded a step ahead
          while (n1 < nblk_s && s_blk[n1 + 1] - s_blk[c1 < nblk_s ? c1 : c0] <= W2) n1++;
          const uint32_t nxt_en = sp0[cb0 + n1];
          rbase = s_blk[c0];
          if (s_blk[c1] - rbase) {                                               // (no entry in these blocks: no product either)
            if (one_chunk) rwalk(cur_st, has ? cur_en - cur_st : 0u);
            else for (uint32_t base = ab; base < ae; base += SPA_CHUNK) {
              const uint32_t pa = base + t; uint32_t st = 0, len = 0;
              if (t < SPA_CHUNK && pa < ae) { const uint32_t* sp = split + (size_t)a.acol[pa] * (nblk + 1) + cb0; st = sp[c0]; len = sp[c1] - st; if (use_a) s_av[t] = aval[pa]; }
              rwalk(st, len);
            }
            remit(c0, c1);
          }
          cur_st = cur_en; cur_en = nxt_en; c0 = c1; c1 = n1;
        }
        obase += rtotal;
        continue;
      }
      // a slab that goes block by block between ranked ones: the region held a bitmap
      __syncthreads();
      for (uint32_t e = t; e < WD; e += 1024) s_acc[e] = idw;
      for (uint32_t w = t; w < WD / 4; w += 1024) s_flag[w] = 0;
      racc_clean = false;
      __syncthreads();
      direct_range(cb0, cb0 + nblk_s);
      }     // slabs of the row
      continue;
    }
    direct_range(0, nblk);
  }
#ifdef SPA_PROFILE
  if (t == 0) { pf[7] = __builtin_amdgcn_s_memtime() - pf_start; for (int k = 0; k < 16; k++) g_spa_prof[(size_t)(blockIdx.x & 1023) * 16 + k] = pf[k]; }
#endif
}
// the table rows after their sort: columns and values move from the compact staging arrays (row pointers trp) into the result (orp);
// one wave per row of the three LDS-table bins
template <class T> __global__ void k_hash_place_rows(const uint32_t* __restrict__ rows, uint32_t nrows_list, const uint32_t* __restrict__ trp, const uint32_t* __restrict__ orp,
                                                     const uint32_t* __restrict__ scol, const T* __restrict__ uval, const uint32_t* __restrict__ perm, uint32_t* __restrict__ ocol, T* __restrict__ oval) {
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t r = (blockIdx.x * 256u + threadIdx.x) >> 6; r < nrows_list; r += gridDim.x * 4u) {
    const uint32_t i = rows[r]; const uint32_t tb = trp[i], n = trp[i + 1] - tb, ob = orp[i];
    for (uint32_t q = lane; q < n; q += 64) { ocol[ob + q] = scol[tb + q]; oval[ob + q] = uval[perm[tb + q]]; }
  }
}
// entry counts of the rows that go through the tables (the others are written in place by the LDS dense path); [nrows] = 0 for the scan
static __global__ void k_table_counts(uint32_t nrows, const uint32_t* __restrict__ rownnz, uint32_t big_from, uint32_t* __restrict__ out) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i <= nrows; i += gridDim.x * 256ull) out[i] = (i < nrows && rownnz[i] <= big_from) ? rownnz[i] : 0u;
}
template <class W> __global__ void k_hash_gather(const W* __restrict__ in, const uint32_t* __restrict__ perm, uint64_t n, W* __restrict__ out) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) out[i] = in[perm[i]];
}

// 64-bit sum of the rows' entry counts: the 32-bit row pointers of the result must not wrap (A@A on a power-law graph can exceed 2^32 entries)
static __global__ void k_hash_total(const uint32_t* __restrict__ rownnz, uint32_t nrows, unsigned long long* __restrict__ total) {
  unsigned long long s = 0;
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < nrows; i += gridDim.x * 256ull) s += rownnz[i];
  for (int o = 32; o; o >>= 1) s += __shfl_down(s, o);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(total, s);
}

template <class T> void run_spgemm_hash(const SpgemmCall& c, const SemiringDesc& d, DevCSR& out) {
  typedef typename acc_word<T>::type W;
  const DevCSR& A = *c.A; const DevCSR& B = *c.B;
  const uint32_t nrows = A.nrows, ncols = B.ncols;
  out.clear(); out.nrows = nrows; out.ncols = ncols;
  out.rowptr.alloc(((size_t)nrows + 1) * 4);
  auto grid_n = [](uint64_t n) { uint64_t b = (n + 255) / 256; if (b < 1) b = 1; if (b > 8192) b = 8192; return (unsigned)b; };
  if (!nrows || !A.nnz || !B.nnz) { GRB_HIP(hipMemsetAsync(out.rowptr.p, 0, ((size_t)nrows + 1) * 4, stream())); out.nnz = 0; out.col.alloc(8); out.val.alloc(8); out.valid = true; return; }
  const int ncu = device_cus() > 0 ? device_cus() : 256;
  auto nblocks = [&](uint32_t rows, int teams) { uint64_t b = ((uint64_t)rows + teams - 1) / teams; if (b > (uint64_t)ncu * 32) b = (uint64_t)ncu * 32; if (b < 1) b = 1; return (unsigned)b; };
  // ---- product counts, symbolic bins -----------------------------------------------------------------------------------------
  DevBuf ub((size_t)nrows * 8 + 8), rownnz(((size_t)nrows + 1) * 4), counts(64), lists((size_t)5 * nrows * 4 + 4);
  row_upper_bound(A, B, ub.as<unsigned long long>());
  GRB_HIP(hipMemsetAsync(rownnz.p, 0, ((size_t)nrows + 1) * 4, stream()));
  GRB_HIP(hipMemsetAsync(counts.p, 0, 64, stream()));
  // the LDS dense-accumulator path for the rows beyond the tables (k_spgemm_spa_*): a moderate column range only
  const bool no_spa = getenv("GRB_MI355X_SPGEMM_NO_SPA") != nullptr;             // measurement / test hook: the HBM accumulators of round 2
  const bool spa = !no_spa && (uint64_t)ncols <= 64ull * spa_cfg<T>::WD && (uint64_t)ncols <= 32ull * SPA_SYM_WORDS;
  // (with the LDS bitmap at hand it also counts the rows of 4097 ... 16 384 products: marking bits beats clearing and counting a 32 768-slot table per row)
  // ranked rows (k_spgemm_spa_numeric, round 5): possible when the row bitmap leaves room for accumulators beside it (<= 2/5 of the region: 2^18 columns)
  // and the column range is <= 32 blocks.  A ranked row of <= W2 entries is ONE step of ~10 us, where the 8192-slot table took 60 us per row of 1025 ... 4096
  // entries (6.6 ms of A@A on R-MAT-18, and its rows went through the staging arrays and the sort): with ranking the dense path takes every row beyond 1024
  // products / entries.  GRB_MI355X_SPA_RANK=0: round 4's bins and kernel.
  constexpr uint32_t WDc = spa_cfg<T>::WD; constexpr uint64_t REGIONc = (uint64_t)WDc * sizeof(W) + WDc;
  const bool rank_env = !(getenv("GRB_MI355X_SPA_RANK") && atoi(getenv("GRB_MI355X_SPA_RANK")) == 0);      // (read per call: a test hook)
  // (round 6: any column range the dense path takes — a row wider than that is ranked slab by slab)
  const bool rank_static = spa && rank_env && std::min<uint64_t>((REGIONc * 2 / 5) / 6, 8192) / (WDc / 32) >= 1 && std::min<uint64_t>((REGIONc * 2 / 5) / 6, 8192) / (WDc / 32) <= SPA_RANK_MAXBLK;
  // deterministic mode (SpgemmCall::ordered) for a floating-point monoid: only the one-wave table kernel (<= 128 products: a wave's LDS atomics execute in
  // program order) and the dense path with its ordered walk add in a fixed order — every other row goes to the dense path
  const bool ordered = c.ordered && spa && (d.zcode == T_FP32 || d.zcode == T_FP64);
  const unsigned long long big_from = ordered ? 128ull : rank_static ? 1024ull : 4096ull, mid_from = ordered ? 128ull : 1024ull;
  hipLaunchKernelGGL(k_hash_bin5, dim3(grid_n(nrows)), dim3(256), 0, stream(), nrows, ub.as<unsigned long long>(), 128ull, mid_from, big_from, spa ? big_from : 16384ull, counts.as<uint32_t>(), lists.as<uint32_t>());
  uint32_t hs[5];
  GRB_HIP(hipMemcpyAsync(hs, counts.p, 20, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
  const uint32_t words = (ncols + 31) / 32;
  // persistent workgroups of the dense paths: bounded by memory (bitmaps: ncols/8 bytes each; accumulators: ncols words each, <= 4 GiB in all)
  auto dense_blocks = [&](uint32_t rows, size_t per_block) { uint64_t fit = (4ull << 30) / (per_block ? per_block : 1); if (fit < 1) fit = 1; return (unsigned)std::min<uint64_t>(std::min<uint64_t>(rows, (uint64_t)ncu * 2), fit); };
  uint32_t hn[4] = {0, 0, 0, 0};
  DevBuf bitmaps, bmslot, rowctr(64); bool ranked = false;
  GRB_HIP(hipMemsetAsync(rowctr.p, 0, 64, stream()));
  // the bitmaps of every row beyond the tables: ncols / 8 bytes each — 3.8 GB for A@A on R-MAT-18, 21 GB on R-MAT-20 with edge factor 4; at most a quarter of
  // the device's memory (GRB_MI355X_SPA_BITMAP_GB), beyond that the rows go block by block as before round 5.  The figure is part of the plan string.
  uint64_t bitmap_bytes = 0;
  {
    size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); tot = 64ull << 30; }
    const char* eg = getenv("GRB_MI355X_SPA_BITMAP_GB");
    const uint64_t cap = eg && *eg ? (uint64_t)(atof(eg) * (double)(1ull << 30)) : (uint64_t)tot / 4;
    if (rank_static && hs[4] && (uint64_t)hs[4] * words * 4 <= cap) {
      bitmap_bytes = (uint64_t)hs[4] * words * 4;
      bitmaps.alloc((size_t)bitmap_bytes + 16); bmslot.alloc((size_t)nrows * 4 + 16); ranked = true;
    }
  }
  with_semiring<T>(d, [&](auto sr) {
    typedef decltype(sr) SR;
    HashArgs a{A.rowptr.as<uint32_t>(), A.col.as<uint32_t>(), B.rowptr.as<uint32_t>(), B.col.as<uint32_t>(), nullptr, 0, rownnz.as<uint32_t>(), nullptr, nullptr};
    const uint32_t* L = lists.as<uint32_t>();
    {
      DevBuf bm;
      if (hs[4] && spa) { a.rows = L + (size_t)4 * nrows; a.nrows_bin = hs[4];
                          hipLaunchKernelGGL((k_spgemm_spa_symbolic<T, SR>), dim3(std::min<unsigned>(hs[4], (unsigned)ncu)), dim3(1024), 0, stream(), a, ncols, rowctr.as<uint32_t>(),
                                             ranked ? bitmaps.as<uint32_t>() : (uint32_t*)nullptr, ranked ? bmslot.as<uint32_t>() : (uint32_t*)nullptr); }
      else if (hs[4]) { const unsigned nb = dense_blocks(hs[4], (size_t)words * 4); bm.alloc((size_t)nb * words * 4); GRB_HIP(hipMemsetAsync(bm.p, 0, (size_t)nb * words * 4, stream()));
                   a.rows = L + (size_t)4 * nrows; a.nrows_bin = hs[4];
                   hipLaunchKernelGGL((k_spgemm_dense<T, SR, false>), dim3(nb), dim3(1024), 0, stream(), a, (const T*)nullptr, (const T*)nullptr, (T*)nullptr, ncols, bm.as<uint32_t>(), (W*)nullptr, sr); }
      if (hs[3]) { a.rows = L + (size_t)3 * nrows; a.nrows_bin = hs[3]; hipLaunchKernelGGL((k_spgemm_hash<T, SR, 32768, 1024, 1024, false>), dim3(nblocks(hs[3], 1)), dim3(1024), 0, stream(), a, (const T*)nullptr, (const T*)nullptr, (T*)nullptr, sr); }
      if (hs[2]) { a.rows = L + (size_t)2 * nrows; a.nrows_bin = hs[2]; hipLaunchKernelGGL((k_spgemm_hash<T, SR, 8192, 512, 512, false>), dim3(nblocks(hs[2], 1)), dim3(512), 0, stream(), a, (const T*)nullptr, (const T*)nullptr, (T*)nullptr, sr); }
      if (hs[1]) { a.rows = L + (size_t)1 * nrows; a.nrows_bin = hs[1]; hipLaunchKernelGGL((k_spgemm_hash<T, SR, 2048, 256, 256, false>), dim3(nblocks(hs[1], 1)), dim3(256), 0, stream(), a, (const T*)nullptr, (const T*)nullptr, (T*)nullptr, sr); }
      if (hs[0]) { a.rows = L; a.nrows_bin = hs[0]; hipLaunchKernelGGL((k_spgemm_hash<T, SR, 256, 64, 256, false>), dim3(nblocks(hs[0], 4)), dim3(256), 0, stream(), a, (const T*)nullptr, (const T*)nullptr, (T*)nullptr, sr); }
      GRB_HIP(hipGetLastError());
      // ---- row pointers, the size of T, numeric bins ---------------------------------------------------------------------------
      exclusive_scan_u32(rownnz.as<uint32_t>(), out.rowptr.as<uint32_t>(), (uint64_t)nrows + 1);
      GRB_HIP(hipMemsetAsync(counts.p, 0, 64, stream()));
      hipLaunchKernelGGL(k_hash_bin, dim3(grid_n(nrows)), dim3(256), 0, stream(), nrows, (const unsigned long long*)nullptr, rownnz.as<uint32_t>(), 128ull, mid_from, big_from, counts.as<uint32_t>(), lists.as<uint32_t>());
      hipLaunchKernelGGL(k_hash_total, dim3(grid_n(nrows)), dim3(256), 0, stream(), rownnz.as<uint32_t>(), nrows, (unsigned long long*)(counts.as<uint8_t>() + 32));
      uint64_t hc[5] = {0, 0, 0, 0, 0};                                // four bin counts (u32 x 4) | - | the 64-bit total at byte 32
      GRB_HIP(hipMemcpyAsync(hc, counts.p, 40, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));   // (also: the bitmaps of the symbolic pass are idle now)
      memcpy(hn, hc, 16);
      if (hc[4] > 0xFFFFFFF0ull) fail(GrB_INSUFFICIENT_SPACE, "mxm: the result holds " + std::to_string(hc[4]) + " entries; the device layout's 32-bit offsets hold < 2^32");
      out.nnz = hc[4];
    }
    const uint64_t total = out.nnz;
    out.col.alloc(total * 4 + 8); out.val.alloc(total * sizeof(T) + 8);
    if (total) {
      const T* av = (const T*)c.aval; const T* bv = (const T*)c.bval;
      int cbits = 1; while ((1ull << cbits) < (unsigned long long)ncols) cbits++;
      a.rownnz = nullptr;
      if (spa) {
        // The rows beyond the tables go through the LDS dense accumulator straight into the result, in column order.  Only the table
        // rows (<= 4096 entries each) need the unordered staging arrays, the sort and the move: their own compact row pointers
        // (16 B of temporaries per entry of THOSE rows — the 47 GB of the R-MAT-18 A@A were 94 GB of allocations per call).
        DevBuf tcnt(((size_t)nrows + 1) * 4), trp(((size_t)nrows + 1) * 4);
        hipLaunchKernelGGL(k_table_counts, dim3(grid_n(nrows + 1)), dim3(256), 0, stream(), nrows, rownnz.as<uint32_t>(), (uint32_t)big_from, tcnt.as<uint32_t>());
        exclusive_scan_u32(tcnt.as<uint32_t>(), trp.as<uint32_t>(), (uint64_t)nrows + 1);
        uint32_t ttotal = 0;
        GRB_HIP(hipMemcpyAsync(&ttotal, trp.as<uint32_t>() + nrows, 4, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
        DevBuf split;
        if (hn[3]) {
          constexpr uint32_t WD = spa_cfg<T>::WD; const uint32_t nblk = (uint32_t)(((uint64_t)ncols + WD - 1) / WD);
          split.alloc((size_t)B.nrows * (nblk + 1) * 4 + 8);
          hipLaunchKernelGGL(k_spa_split, dim3(std::min<unsigned>((B.nrows + 3) / 4, 65535u)), dim3(256), 0, stream(), B.nrows, B.rowptr.as<uint32_t>(), B.col.as<uint32_t>(),
                             WD, nblk, split.as<uint32_t>());
          a.crp = out.rowptr.as<uint32_t>(); a.ccol = nullptr; a.rows = L + (size_t)3 * nrows; a.nrows_bin = hn[3];
          const uint32_t* const bmp = ranked ? bitmaps.as<uint32_t>() : (const uint32_t*)nullptr; const uint32_t* const bms = ranked ? bmslot.as<uint32_t>() : (const uint32_t*)nullptr;
          const dim3 ngrid(std::min<unsigned>(hn[3], (unsigned)ncu));
          bool launched = false;
          if constexpr (std::is_floating_point<T>::value) if (ordered) {
            hipLaunchKernelGGL((k_spgemm_spa_numeric<T, SR, true>), ngrid, dim3(1024), 0, stream(), a, av, bv, out.col.as<uint32_t>(), out.val.as<T>(), ncols, split.as<uint32_t>(), sr, rowctr.as<uint32_t>() + 1, bmp, bms);
            launched = true;
          }
          if (!launched) hipLaunchKernelGGL((k_spgemm_spa_numeric<T, SR, false>), ngrid, dim3(1024), 0, stream(), a, av, bv, out.col.as<uint32_t>(), out.val.as<T>(), ncols, split.as<uint32_t>(), sr, rowctr.as<uint32_t>() + 1, bmp, bms);
        }
        if (ttotal) {
          DevBuf ucol((size_t)ttotal * 4 + 8), uval((size_t)ttotal * sizeof(T) + 8), scol((size_t)ttotal * 4 + 8), perm0((size_t)ttotal * 4 + 8), perm((size_t)ttotal * 4 + 8);
          a.crp = trp.as<uint32_t>(); a.ccol = ucol.as<uint32_t>();
          if (hn[2]) { a.rows = L + (size_t)2 * nrows; a.nrows_bin = hn[2]; hipLaunchKernelGGL((k_spgemm_hash<T, SR, 8192, 1024, 1024, true>), dim3(nblocks(hn[2], 1)), dim3(1024), 0, stream(), a, av, bv, uval.as<T>(), sr); }
          if (hn[1]) { a.rows = L + (size_t)1 * nrows; a.nrows_bin = hn[1]; hipLaunchKernelGGL((k_spgemm_hash<T, SR, 2048, 256, 256, true>), dim3(nblocks(hn[1], 1)), dim3(256), 0, stream(), a, av, bv, uval.as<T>(), sr); }
          if (hn[0]) { a.rows = L; a.nrows_bin = hn[0]; hipLaunchKernelGGL((k_spgemm_hash<T, SR, 256, 64, 256, true>), dim3(nblocks(hn[0], 4)), dim3(256), 0, stream(), a, av, bv, uval.as<T>(), sr); }
          GRB_HIP(hipGetLastError());
          hipLaunchKernelGGL(k_iota32, dim3(grid_n(ttotal)), dim3(256), 0, stream(), perm0.as<uint32_t>(), (uint64_t)ttotal);
          segmented_sort_pairs_u32(ucol.as<uint32_t>(), scol.as<uint32_t>(), perm0.as<uint32_t>(), perm.as<uint32_t>(), ttotal, nrows, trp.as<uint32_t>(), trp.as<uint32_t>() + 1, cbits);
          for (int b = 0; b < 3; b++) if (hn[b])
            hipLaunchKernelGGL((k_hash_place_rows<T>), dim3(std::min<unsigned>((hn[b] + 3) / 4, 65535u)), dim3(256), 0, stream(), L + (size_t)b * nrows, hn[b], trp.as<uint32_t>(), out.rowptr.as<uint32_t>(),
                               scol.as<uint32_t>(), uval.as<T>(), perm.as<uint32_t>(), out.col.as<uint32_t>(), out.val.as<T>());
          GRB_HIP(hipGetLastError());
          GRB_HIP(hipStreamSynchronize(stream()));
        }
      } else {
      DevBuf ucol(total * 4 + 8), uval(total * sizeof(T) + 8);
      a.crp = out.rowptr.as<uint32_t>(); a.ccol = ucol.as<uint32_t>();
      DevBuf bm, accs;
      if (hn[3]) {
        const unsigned nb = dense_blocks(hn[3], (size_t)ncols * sizeof(W) + (size_t)words * 4);
        bm.alloc((size_t)nb * words * 4); accs.alloc((size_t)nb * ncols * sizeof(W));
        GRB_HIP(hipMemsetAsync(bm.p, 0, (size_t)nb * words * 4, stream()));
        hipLaunchKernelGGL((k_fill_words<W>), dim3(4096), dim3(256), 0, stream(), accs.as<W>(), (uint64_t)nb * ncols, to_word<T>(sr.identity));
        a.rows = L + (size_t)3 * nrows; a.nrows_bin = hn[3];
        hipLaunchKernelGGL((k_spgemm_dense<T, SR, true>), dim3(nb), dim3(1024), 0, stream(), a, av, bv, uval.as<T>(), ncols, bm.as<uint32_t>(), accs.as<W>(), sr);
      }
      if (hn[2]) { a.rows = L + (size_t)2 * nrows; a.nrows_bin = hn[2]; hipLaunchKernelGGL((k_spgemm_hash<T, SR, 8192, 1024, 1024, true>), dim3(nblocks(hn[2], 1)), dim3(1024), 0, stream(), a, av, bv, uval.as<T>(), sr); }
      if (hn[1]) { a.rows = L + (size_t)1 * nrows; a.nrows_bin = hn[1]; hipLaunchKernelGGL((k_spgemm_hash<T, SR, 2048, 256, 256, true>), dim3(nblocks(hn[1], 1)), dim3(256), 0, stream(), a, av, bv, uval.as<T>(), sr); }
      if (hn[0]) { a.rows = L; a.nrows_bin = hn[0]; hipLaunchKernelGGL((k_spgemm_hash<T, SR, 256, 64, 256, true>), dim3(nblocks(hn[0], 4)), dim3(256), 0, stream(), a, av, bv, uval.as<T>(), sr); }
      GRB_HIP(hipGetLastError());
      // ---- column order inside every row ----------------------------------------------------------------------------------------
      DevBuf perm0(total * 4 + 8), perm(total * 4 + 8);
      hipLaunchKernelGGL(k_iota32, dim3(grid_n(total)), dim3(256), 0, stream(), perm0.as<uint32_t>(), total);
      segmented_sort_pairs_u32(ucol.as<uint32_t>(), out.col.as<uint32_t>(), perm0.as<uint32_t>(), perm.as<uint32_t>(), total, nrows, out.rowptr.as<uint32_t>(), out.rowptr.as<uint32_t>() + 1, cbits);
      hipLaunchKernelGGL((k_hash_gather<T>), dim3(grid_n(total)), dim3(256), 0, stream(), uval.as<T>(), perm.as<uint32_t>(), total, out.val.as<T>());
      GRB_HIP(hipGetLastError());
      GRB_HIP(hipStreamSynchronize(stream()));                        // the temporaries of this scope go back to the pool
      }
    }
    g_last_plan += std::string("spgemm_hash<") + (sr.is_static ? "static" : "dynamic") + "> symbolic bins " + std::to_string(hs[0]) + "/" + std::to_string(hs[1]) + "/" + std::to_string(hs[2]) + "/" +
                   std::to_string(hs[3]) + "/" + std::to_string(hs[4]) + " numeric bins " + std::to_string(hn[0]) + "/" + std::to_string(hn[1]) + "/" + std::to_string(hn[2]) + "/" + std::to_string(hn[3]) + (ranked ? " ranked (row bitmaps " + std::to_string((bitmap_bytes + (1ull << 29)) >> 30) + " GB)" : "") + (ordered ? " ordered " : " ");
  });
  out.valid = true;
}

}  // namespace grb
