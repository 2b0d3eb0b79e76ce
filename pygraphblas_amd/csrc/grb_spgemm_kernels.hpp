// grb_spgemm_kernels.hpp — the SpGEMM kernels behind GrB_mxm  (irregular gather; no MFMA).
//
// (1) Masked Gustavson,  T<M> = A (+).(x) B  with a non-complemented mask — the triangle-counting
//     path  L.mxm(L, PLUS_PAIR, mask=L)  (reference caller: demo/TriangleCentrality.ipynb cell 17).
//     One team of threads owns output row i.  The allowed columns of M(i,:) are hashed into an
//     LDS table (open addressing, load <= 1/2); 16-lane groups walk the entries k of A(i,:) and
//     stream row B(k,:) coalesced; every product probes the table and, on a hit, combines into
//     the LDS accumulator of that mask position.  Only masked products ever touch memory beyond the
//     B stream, and nothing is written outside nnz(M) slots.  Rows are binned by mask-row length so
//     the table (64 ... 8192 slots) and the team (1 wave ... 4 waves) fit the row; rows whose mask
//     row exceeds 4096 entries use a dense position map in HBM and global atomics.
// (2) Expand / sort / compress for the unmasked (or complement-masked) product: every product is
//     emitted with key (i,j), a stable radix sort brings equal keys together in k order, and a
//     segmented in-order reduction compresses them — deterministic for floating point too.
// Algorithmic work: F = 2 * sum_{(i,k) in A} nnz(B(k,:)) flops,
//   bytes = 2*(nnz(A)*4 + (n+1)*4) + (F/2)*4 [+ (F/2)*sizeof(T) when the multiply reads B's values] + nnz(C)*(4+sizeof(T)).
#pragma once
#include "grb_api.hpp"
#include "grb_device.hpp"
#include "grb_semiring.hpp"
#include "grb_atomics.hpp"
#include "grb_exact.hpp"
#include "grb_matops.hpp"
#include "grb_spgemm_kernels_fwd.hpp"
#include <algorithm>

namespace grb {

constexpr uint32_t HASH_EMPTY = 0xFFFFFFFFu;
__device__ __forceinline__ uint32_t hash_col(uint32_t j, uint32_t mask) { return (j * 2654435761u >> 7) & mask; }

template <class T> struct SpgemmKArgs {
  const uint32_t* arp; const uint32_t* acol; const T* aval;
  const uint32_t* brp; const uint32_t* bcol; const T* bval;
  const uint32_t* mrp; const uint32_t* mcol; const void* mval; int mcode; bool mstruct;
  typename acc_word<T>::type* cacc; uint8_t* cflag;      // one slot per mask entry
  // deterministic mode, exact accumulators (grb_exact.hpp): the unit exponent of every output row, and the 128-bit integers of the rows that accumulate in HBM
  const int32_t* rowexp; unsigned long long* xlo; unsigned long long* xhi;
};

__device__ __forceinline__ bool spgemm_mask_truth(const void* mval, int mcode, uint32_t p, bool structural) {
  if (structural || !mval) return true;
  switch (type_size(mcode)) {
    case 1: return ((const uint8_t*)mval)[p] != 0;
    case 2: return ((const uint16_t*)mval)[p] != 0;
    case 4: return mcode == T_FP32 ? ((const float*)mval)[p] != 0.0f : ((const uint32_t*)mval)[p] != 0;
    default: return mcode == T_FP64 ? ((const double*)mval)[p] != 0.0 : ((const uint64_t*)mval)[p] != 0;
  }
}

// ---- (1a) LDS-hash masked Gustavson: TEAM threads per row, BLOCK / TEAM rows per block -----------------------------
// Work inside a row is wildly uneven (R-MAT: B rows of 1 ... 40 000 entries), so the entries k of A(i,:) are first
// sorted into two LDS work lists by the length of B(k,:): short rows (< 64 entries) are walked by 16-lane groups,
// four at a time per wave; long rows by a whole wave each, 64 coalesced entries per step.
struct __attribute__((packed, aligned(4))) spg_u4 { uint32_t x, y, z, w; };     // four consecutive column indices of a row, one 16-byte load at 4-byte alignment
// measurement builds (make XTFLAGS=-DSPG_EXP=n): 1 = no B-row loads (synthetic columns: what the probes and barriers cost alone),
// 2 = B-row loads without the probes (what the streams cost under this kernel's work split), 3 = every B row read from one 2 MiB window
#ifndef SPG_EXP
#define SPG_EXP 0
#endif
__device__ __forceinline__ uint32_t spg_ld(const uint32_t* __restrict__ p, uint32_t i) {
  if constexpr (SPG_EXP == 1) return (i * 2654435761u) >> 10; else return p[i];
}
__device__ __forceinline__ spg_u4 spg_ld4(const uint32_t* __restrict__ p, uint32_t i) {
  if constexpr (SPG_EXP == 1) { spg_u4 r; r.x = (i * 2654435761u) >> 10; r.y = ((i + 1) * 2654435761u) >> 10; r.z = ((i + 2) * 2654435761u) >> 10; r.w = ((i + 3) * 2654435761u) >> 10; return r; }
  else return *(const spg_u4*)(p + i);
}
#ifndef SPG_FBITS_V
#define SPG_FBITS_V 32
#endif
#ifndef SPG_FBITS_SMALL_V
#define SPG_FBITS_SMALL_V 0
#endif
#ifndef SPG_LIST_BIG_V
#define SPG_LIST_BIG_V 1024
#endif
#ifndef SPG_LONG_V
#define SPG_LONG_V 64
#endif
#ifndef SPG_SLICE_V
#define SPG_SLICE_V 2048
#endif
#ifndef SPG_HUBHUGE_V
#define SPG_HUBHUGE_V 32768
#endif
#ifndef SPG_LIST_V
#define SPG_LIST_V 512
#endif
#ifndef SPG_EXACT_EXP
#define SPG_EXACT_EXP 0
#endif
#ifndef SPG_PROBE4_V
#define SPG_PROBE4_V 1          // products that read values: four lookups, then their hits in passes (0: lookup and hit one product at a time)
#endif
#ifndef SPG_HUGE_V
#define SPG_HUGE_V 8192
#endif
constexpr int SPG_LIST = SPG_LIST_V;       // k's staged per round (per team)           (measurement builds: -DSPG_LIST_V=..., -DSPG_HUGE_V=...)
constexpr uint32_t SPG_HUGE = SPG_HUGE_V;  // B rows at least this long are shared by the whole team (R-MAT-22 triangle count: 1024 -> 51.4 ms, 2048 -> 48.4, 4096 -> 46.1, 8192 -> 45.4, 32768 -> 45.9; lists of 256 / 1024 entries: 49.1 / 52.8 against 48.4 for 512)
constexpr int SPG_QCAP = 128;       // survivor queue of a wave (flushed 64 at a time)
constexpr uint32_t SPG_FILTER_MUL = 0x9E3779u;   // 24-bit multiplier: v_mul_u32_u24 runs at full rate, v_mul_lo_u32 at a quarter
// Round 3, second half.  The measurement builds (SPG_EXP) showed what the kernel's time is: without any B-row load it still took
// 53 of 67 ms, with the loads and without the probes 38 ms, with every B row read out of one L2-resident window 63 ms.  It is the
// VALU: eight waves per SIMD keep it 80 % busy (SQ_ACTIVE_INST_VALU = 10 % of wave cycles x 8), because a probe is a
// quarter-rate integer multiply plus a linear-probing loop that the whole wave repeats until its unluckiest lane is done
// (~2.75 rounds of 5 VALU + 5 SALU instructions at load 1/4), and 96 % of the products are misses that only had to learn "no".
// Now a product first asks a BIT FILTER of the mask row (32 bits per table slot, one 24-bit multiply, one LDS read, no loop);
// the few lanes that pass (the 3.7 % hits + ~1-3 % false positives) append their column to the wave's queue in LDS (ballot +
// mbcnt), and whenever the queue holds 64 columns the wave looks all of them up in the exact table with every lane busy.
// When the multiply reads no value (PLUS_PAIR: the triangle count) the queue is carried from B row to B row; otherwise it is
// flushed at the end of every B row (the A value changes), and B rows shorter than 256 entries keep the direct lookup.
// EXACT (deterministic mode, PLUS monoid on FP32 / FP64): the accumulators are 128-bit integers in the row's unit (grb_exact.hpp) — the order the atomics
// land in no longer matters, everything else in the kernel is the same.  A hit then costs ~95 instructions (out of line: fx_add_lds) and a returning LDS
// atomic instead of a multiply and ds_add_f64, and a wave pays them whenever ANY of its lanes hits: the bins run 1.2-1.3 x their default time (R-MAT-22, FP64
// PLUS_TIMES: 54.9 / 38.5 / 34.8 / 14.7 ms against 45.3 / 30.0 / 26.0 / 11.9).  The last bin keeps its 4 096 mask entries (EXBIG below): with 2 048 the HBM-map
// kernel got five times the rows and, one 147 KB workgroup per CU, kept every other bin off the CUs three times as long.  Measurement builds: SPG_EXACT_EXP = 1
// (plain ds_add_f64 in the EXACT kernels: the same time, i.e. the integer add is not what an EXACT kernel costs), 2 (no returning atomic).
template <class T, class SR, int SLOTS, int TEAM, int BLOCK, bool EXACT = false>
__global__ __launch_bounds__(BLOCK) void k_spgemm_masked_lds(const SpgemmKArgs<T> a, const uint32_t* __restrict__ rows, uint32_t nrows_bin, const SR sr) {
  typedef typename acc_word<T>::type W;
  typedef typename std::conditional<EXACT, unsigned long long, W>::type AW;
  constexpr int TEAMS = BLOCK / TEAM;
  // (EXACT, last bin: 16-byte accumulators for 4 096 mask entries — the work lists and the filter give up LDS for them: 512 k's per round, 8 filter bits per slot)
  constexpr bool EXBIG = EXACT && SLOTS >= 8192;
  constexpr int LCAP = EXBIG ? 512 : (TEAM >= 1024 ? SPG_LIST_BIG_V : (TEAM >= 256 ? SPG_LIST : 64));
  constexpr bool NOVAL = SR::pair_only;                 // the product is a constant: a queued survivor is its column alone
  // the exact table: open addressing, at most SLOTS / 2 mask entries in KS keys (load <= 1/4 in the two small bins, <= 1/2 in the
  // large ones, where the LDS goes to the filter and the queues instead — only survivors of the filter walk its chains now);
  // the accumulators are indexed by the mask position the slot carries and take only SLOTS / 2 words
  constexpr int KS = SLOTS <= 512 ? 2 * SLOTS : SLOTS, ML = SLOTS / 2;
  constexpr int FBITS = (EXBIG ? 8 : (SPG_FBITS_SMALL_V && SLOTS <= 512 ? SPG_FBITS_SMALL_V : SPG_FBITS_V)) * SLOTS, FW = FBITS / 32;    // the filter: >= 64 bits per mask entry (R-MAT-22 triangle count: 8 x SLOTS bits 47.2 ms, 16 x 45.0, 32 x 44.2; 64 / 128 x in the two small bins only: 44.9 / 45.8)
  constexpr int FSH = 32 - __builtin_ctz(FBITS);
  constexpr int WAVES = BLOCK / 64;
  __shared__ uint32_t s_key[TEAMS][KS];
  __shared__ uint16_t s_pos[TEAMS][KS];
  __shared__ AW s_acc[TEAMS][ML];
  __shared__ unsigned long long s_hi[TEAMS][EXACT ? ML : 1];
  __shared__ uint8_t s_flag[TEAMS][ML];
  __shared__ uint32_t s_filt[TEAMS][FW];
  __shared__ uint32_t s_qj[WAVES][SPG_QCAP];
  __shared__ uint32_t s_qp[NOVAL ? 1 : WAVES][NOVAL ? 1 : SPG_QCAP];
  __shared__ uint32_t s_lpa[TEAMS][LCAP], s_lbb[TEAMS][LCAP], s_lbe[TEAMS][LCAP];     // work list: A-entry position, B row begin / end
  __shared__ uint32_t s_cnt[TEAMS][3];                                                  // [0] short rows fill from the front, [1] long rows from the back, [2] huge rows
  constexpr int HCAP = TEAM > 64 ? 64 : 1;                                              // B rows of >= SPG_HUGE entries are walked by the whole team (one wave would hold the others at the barrier)
  __shared__ uint32_t s_hpa[TEAMS][HCAP], s_hbb[TEAMS][HCAP], s_hbe[TEAMS][HCAP];
  const int team = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / TEAM)), t = threadIdx.x % TEAM;      // (TEAM >= 64: the same for every lane of a wave — kept in a scalar register, and with it every LDS base below)
  const int lane16 = t & 15, grp = t >> 4, lane64 = t & 63, wv = t >> 6;
  constexpr int NG = TEAM / 16, NW = TEAM / 64;
  const bool use_a = sr.uses_a(), use_b = sr.uses_u();
  const W idw = to_word<T>(sr.identity);
  int uexp = 0;                                          // EXACT: the unit exponent of the row being formed
  uint32_t* key = s_key[team]; AW* acc = s_acc[team]; unsigned long long* const hiw = s_hi[team]; uint16_t* pos = s_pos[team]; uint8_t* flag = s_flag[team]; uint32_t* filt = s_filt[team];
  uint32_t* lpa = s_lpa[team]; uint32_t* lbb = s_lbb[team]; uint32_t* lbe = s_lbe[team]; uint32_t* cnt = s_cnt[team];
  const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint32_t* qj = s_qj[wave_in_block]; uint32_t* qp = s_qp[NOVAL ? 0 : wave_in_block];
  // a team of one wave (the bin of the shortest mask rows: four rows per block) needs no block barrier: its LDS slices are
  // private and LDS operations of a wave execute in order, so every row runs exactly its own number of rounds
  auto wave_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
  auto team_sync = [&]() {
    if constexpr (TEAM == 64) wave_sync();
    else __syncthreads();
  };
  // one product against the table: column j of B(k,:) at position pb
  auto lookup = [&](const uint32_t j) -> uint32_t {          // the mask position of column j, or ~0
    uint32_t h = hash_col(j, KS - 1);
    uint32_t kk = key[h];
    while (kk != j && kk != HASH_EMPTY) { h = (h + 1) & (KS - 1); kk = key[h]; }
    return kk == j ? (uint32_t)pos[h] : 0xFFFFFFFFu;
  };
  auto hit_at = [&](const uint32_t mp, const uint32_t pb, const T av) {
    if constexpr (EXACT && SPG_EXACT_EXP == 1) atomicAdd((double*)&acc[mp], (double)sr.mult(av, use_b ? a.bval[pb] : T()));          // measurement build: everything of EXACT but the integer add
    else if constexpr (EXACT && SPG_EXACT_EXP == 2) { unsigned long long xl, xh; fx_from_double((double)sr.mult(av, use_b ? a.bval[pb] : T()), uexp, xl, xh); atomicAdd(&acc[mp], xl); atomicAdd(&hiw[mp], xh); }   // measurement build: no returning atomic (carries lost)
    else if constexpr (EXACT) fx_add_lds(fx_lds_addr(&acc[mp]), fx_lds_addr(&hiw[mp]), (double)sr.mult(av, use_b ? a.bval[pb] : T()), uexp);
    else word_combine<T>(sr.add_op(), &acc[mp], sr.mult(av, use_b ? a.bval[pb] : T()));
    flag[mp] = 1;
  };
  auto probe = [&](const uint32_t j, const uint32_t pb, const T av) {
    if constexpr (SPG_EXP == 2) { if (j == 0xFFFFFFF1u) flag[0] = 1; return; }
    const uint32_t mp = lookup(j);
    if (mp != 0xFFFFFFFFu) hit_at(mp, pb, av);
  };
  // Four products of a lane against the table: the four lookups first, then the hits — one per lane and pass.  A hit that reads a value waits for B's value (and,
  // EXACT, runs ~95 instructions) with the whole wave, and at 4 % hits per lane 93 % of the single lookups have one somewhere in the wave: four lookups
  // have their hits in ~1.5 passes instead of 3.7.  pb of product u = pb0 + stride * u.
  [[maybe_unused]] auto probe4 = [&](const uint32_t (&jj)[4], const uint32_t pb0, const uint32_t stride, const uint32_t be, const T av) {
    uint32_t mps[4]; uint32_t pend = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) { mps[u] = pb0 + stride * u < be ? lookup(jj[u]) : 0xFFFFFFFFu; pend |= mps[u] != 0xFFFFFFFFu ? 1u << u : 0u; }
    while (__ballot(pend != 0)) {
      if (pend) {
        const uint32_t u = (uint32_t)__builtin_ctz(pend); pend &= pend - 1u;
        const uint32_t mp = u == 0 ? mps[0] : (u == 1 ? mps[1] : (u == 2 ? mps[2] : mps[3]));
        hit_at(mp, pb0 + stride * u, av);
      }
    }
  };
  // bit FSH.. of the 24-bit product picks the filter bit: the top LW bits the word, the five below them the bit in the word
  constexpr int LW = __builtin_ctz(FW);
  auto filter_hash = [&](const uint32_t j) -> uint32_t { return (uint32_t)__umul24(j, SPG_FILTER_MUL); };      // (the intrinsic returns a signed int: shift the unsigned value)
  auto filter_bit = [&](const uint32_t j) -> uint32_t { return filter_hash(j) >> FSH; };
  auto filter_word = [&](const uint32_t j) -> uint32_t {      // (the index through an opaque bfe, so the address is bfe + lshl_add: the plain shift is rewritten into shift + and + add)
    uint32_t idx; asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(idx) : "v"(filter_hash(j)), "n"(32 - LW), "n"(LW));
    return filt[idx];
  };
  auto filter_pass = [&](const uint32_t j, const uint32_t word) -> bool { return __builtin_amdgcn_ubfe(word, filter_hash(j) >> FSH, 1u) != 0u; };
  // ---- the filtered path (whole waves only: every lane of the wave calls these together) ----
  uint32_t qn = 0;                                         // entries in this wave's queue (wave-uniform)
  auto flush = [&](const uint32_t n, const T av) {         // look the first n (<= 64) queued columns up in the table
    wave_sync();
    if ((uint32_t)lane64 < n) probe(qj[lane64], NOVAL ? 0u : qp[lane64], av);
  };
  auto push = [&](const bool pass, const uint32_t j, const uint32_t pb, const T av) {
    const unsigned long long m = __ballot(pass);
    if (m) {
      const uint32_t q0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)qn);                  // (the count lives in a scalar register: the slot address is one VALU operation)
      const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      uint32_t* const qslot = qj + q0;
      if (pass) { qslot[r] = j; if constexpr (!NOVAL) (qp + q0)[r] = pb; }
      qn = q0 + (uint32_t)__popcll(m);
      if (qn >= 64) {
        flush(64, av);
        const uint32_t rest = qn - 64;                     // < 64: the tail moves to the front
        uint32_t xj = 0, xp = 0;
        if ((uint32_t)lane64 < rest) { xj = qj[64 + lane64]; if constexpr (!NOVAL) xp = qp[64 + lane64]; }
        wave_sync();
        if ((uint32_t)lane64 < rest) { qj[lane64] = xj; if constexpr (!NOVAL) qp[lane64] = xp; }
        qn = rest;
      }
    }
  };
  // four consecutive entries per lane / one entry per lane, `ok` = the entry exists
  auto sift4 = [&](const spg_u4 j4, const uint32_t pb, const T av) {
    const uint32_t w0 = filter_word(j4.x), w1 = filter_word(j4.y), w2 = filter_word(j4.z), w3 = filter_word(j4.w);
    push(filter_pass(j4.x, w0), j4.x, pb, av); push(filter_pass(j4.y, w1), j4.y, pb + 1, av);
    push(filter_pass(j4.z, w2), j4.z, pb + 2, av); push(filter_pass(j4.w, w3), j4.w, pb + 3, av);
  };
  const uint32_t nblk_rows = (nrows_bin + TEAMS - 1) / TEAMS * TEAMS;     // every team runs the same trip count
  for (uint32_t rbase = blockIdx.x * TEAMS; rbase < nblk_rows; rbase += gridDim.x * TEAMS) {
    const uint32_t ridx = rbase + team;
    const bool live = ridx < nrows_bin;
    const uint32_t i = live ? rows[ridx] : 0;
    for (int s2 = t; s2 < KS; s2 += TEAM) key[s2] = HASH_EMPTY;
    if constexpr (EXACT) { uexp = __builtin_amdgcn_readfirstlane(live ? a.rowexp[i] : 0); for (int s2 = t; s2 < ML; s2 += TEAM) { acc[s2] = 0; hiw[s2] = 0; flag[s2] = 0; } }
    else for (int s2 = t; s2 < ML; s2 += TEAM) { acc[s2] = idw; flag[s2] = 0; }
    for (int s2 = t; s2 < FW; s2 += TEAM) filt[s2] = 0;
    team_sync();
    const uint32_t mb = live ? a.mrp[i] : 0, me = live ? a.mrp[i + 1] : 0;
    for (uint32_t p = mb + t; p < me; p += TEAM) {
      if (!spgemm_mask_truth(a.mval, a.mcode, p, a.mstruct)) continue;
      const uint32_t j = a.mcol[p];
      uint32_t h = hash_col(j, KS - 1);
      while (atomicCAS(&key[h], HASH_EMPTY, j) != HASH_EMPTY) h = (h + 1) & (KS - 1);
      pos[h] = (uint16_t)(p - mb);
      const uint32_t fb = filter_bit(j);
      atomicOr(&filt[fb >> 5], 1u << (fb & 31));
    }
    const uint32_t ab = live ? a.arp[i] : 0, ae = live ? a.arp[i + 1] : 0;
    // the longest A row of the block decides the number of rounds, so every team reaches every barrier
    uint32_t maxlen = ae - ab;
    if constexpr (TEAMS > 1 && TEAM != 64) {
      __shared__ uint32_t s_max;
      if (threadIdx.x == 0) s_max = 0;
      __syncthreads();
      if (t == 0) atomicMax(&s_max, maxlen);
      __syncthreads();
      maxlen = s_max;
    }
    for (uint32_t r0 = 0; r0 < maxlen; r0 += LCAP) {
      if (t < 3) cnt[t] = 0;
      team_sync();
      // stage the next LCAP entries of A(i,:) into the short / long lists
      for (uint32_t q = r0 + t; q < r0 + LCAP && ab + q < ae; q += TEAM) {
        const uint32_t pa = ab + q, k = a.acol[pa];
        uint32_t bb = a.brp[k], be = a.brp[k + 1];
        if (be == bb) continue;
        if constexpr (SPG_EXP == 3) { const uint32_t len = be - bb; bb &= 0x7FFFFu; be = bb + len; }      // every B row out of one 2 MiB window: what L2-resident streams would give
        if (TEAM > 64 && be - bb >= SPG_HUGE) {
          const uint32_t hs = atomicAdd(&cnt[2], 1u);
          if (hs < (uint32_t)HCAP) { s_hpa[team][hs] = pa; s_hbb[team][hs] = bb; s_hbe[team][hs] = be; continue; }
        }
        const bool lng = be - bb >= SPG_LONG_V;
        const uint32_t slot = lng ? (LCAP - 1 - atomicAdd(&cnt[1], 1u)) : atomicAdd(&cnt[0], 1u);
        lpa[slot] = pa; lbb[slot] = bb; lbe[slot] = be;
      }
      team_sync();
      const uint32_t nshort = cnt[0], nlong = cnt[1], nhuge = cnt[2] < (uint32_t)HCAP ? cnt[2] : (uint32_t)HCAP;
      // short rows: one 16-lane group each, straight to the table
      for (uint32_t q = grp; q < nshort; q += NG) {
        const T av = use_a ? a.aval[lpa[q]] : T();
        const uint32_t be = lbe[q];
        if constexpr (!NOVAL && SPG_PROBE4_V) {              // (a short row has < 64 entries: four steps of the group, their loads in flight together)
          static_assert(SPG_LONG_V <= 64, "a short B row is at most four steps of a 16-lane group");
          const uint32_t pb0 = lbb[q] + lane16;
          uint32_t jj[4];
#pragma unroll
          for (int u = 0; u < 4; u++) jj[u] = pb0 + 16u * u < be ? spg_ld(a.bcol, pb0 + 16u * u) : 0u;
          probe4(jj, pb0, 16u, be, av);
        } else
        for (uint32_t pb = lbb[q] + lane16; pb < be; pb += 16) {
          const uint32_t j = spg_ld(a.bcol, pb);
          probe(j, pb, av);
        }
      }
      // long rows: one wave each
      for (uint32_t q = wv; q < nlong; q += NW) {
        const uint32_t sl = LCAP - 1 - q;
        const T av = use_a ? a.aval[lpa[sl]] : T();
        const uint32_t be = lbe[sl];
        uint32_t base = lbb[sl];
        if (NOVAL || be - base >= 256) {
          // blocks of 1024 entries as FOUR 16-byte loads per lane (1 KiB per wave instruction, at the row's own 4-byte alignment)
          // (one-wave teams: two register sets, the next block's loads are in flight while this block is sifted — 7.3 -> 6.7 ms for
          // the shortest mask rows of the R-MAT-22 triangle count; in the larger teams the 16 extra registers cost a wave per SIMD
          // and the same change made their kernels 25-55 % slower)
          if constexpr (TEAM == 64) if (base + 1024 <= be) {
            spg_u4 ja[4], jb[4];
            auto ld = [&](spg_u4 (&d)[4], const uint32_t at) {
#pragma unroll
              for (int u = 0; u < 4; u++) d[u] = spg_ld4(a.bcol, at + 256 * u + 4 * lane64);
            };
            auto sift = [&](const spg_u4 (&d)[4], const uint32_t at) {
#pragma unroll
              for (int u = 0; u < 4; u++) sift4(d[u], at + 256 * u + 4 * lane64, av);
            };
            {
              ld(ja, base);
              for (;;) {
                const bool nb = base + 2048 <= be;
                if (nb) ld(jb, base + 1024);
                sift(ja, base); base += 1024;
                if (!nb) break;
                const bool na = base + 2048 <= be;
                if (na) ld(ja, base + 1024);
                sift(jb, base); base += 1024;
                if (!na) break;
              }
            }
          }
          for (; base + 1024 <= be; base += 1024) {
            spg_u4 j4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) j4[u] = spg_ld4(a.bcol, base + 256 * u + 4 * lane64);
#pragma unroll
            for (int u = 0; u < 4; u++) sift4(j4[u], base + 256 * u + 4 * lane64, av);
          }
          for (; base < be; base += 256) {                 // the rest, 64 entries per load, four loads in flight
            uint32_t jj[4], bt[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t pb = base + 64 * u + lane64; jj[u] = spg_ld(a.bcol, pb < be ? pb : be - 1); }
#pragma unroll
            for (int u = 0; u < 4; u++) bt[u] = filter_word(jj[u]);
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t pb = base + 64 * u + lane64; if (base + 64 * u < be) push(filter_pass(jj[u], bt[u]) && pb < be, jj[u], pb, av); }
          }
          if constexpr (!NOVAL) { if (qn) { flush(qn, av); qn = 0; wave_sync(); } }
        } else {
          for (uint32_t pb0 = base + lane64; pb0 < be; pb0 += 256) {
            uint32_t jj[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t pb = pb0 + 64 * u; jj[u] = spg_ld(a.bcol, pb < be ? pb : be - 1); }   // 4 loads in flight
            if constexpr (!NOVAL && SPG_PROBE4_V) probe4(jj, pb0, 64u, be, av);
            else {
#pragma unroll
              for (int u = 0; u < 4; u++) { const uint32_t pb = pb0 + 64 * u; if (pb < be) probe(jj[u], pb, av); }
            }
          }
        }
      }
      // huge rows: the whole team, 4 * TEAM coalesced entries per step (16 bytes per lane)
      if constexpr (TEAM > 64) {
        for (uint32_t q = 0; q < nhuge; q++) {
          const T av = use_a ? a.aval[s_hpa[team][q]] : T();
          const uint32_t be = s_hbe[team][q];
          uint32_t base = s_hbb[team][q];
          constexpr int HD = TEAM >= 512 ? 1 : 2;                        // 16-byte loads per lane and step: 4 * TEAM * HD <= SPG_HUGE entries, so every huge row takes this path
          constexpr uint32_t HS = 4 * TEAM * HD;
          for (; base + HS <= be; base += HS) {
            spg_u4 j4[HD];
#pragma unroll
            for (int u = 0; u < HD; u++) j4[u] = spg_ld4(a.bcol, base + 4 * TEAM * u + 4 * t);
#pragma unroll
            for (int u = 0; u < HD; u++) sift4(j4[u], base + 4 * TEAM * u + 4 * t, av);
          }
          for (; base < be; base += 4 * TEAM) {
            uint32_t jj[4], bt[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t pb = base + TEAM * u + t; jj[u] = spg_ld(a.bcol, pb < be ? pb : be - 1); }
#pragma unroll
            for (int u = 0; u < 4; u++) bt[u] = filter_word(jj[u]);
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint32_t pb = base + TEAM * u + t; if (base + TEAM * u + (t & ~63) < be) push(filter_pass(jj[u], bt[u]) && pb < be, jj[u], pb, av); }
          }
          if constexpr (!NOVAL) { if (qn) { flush(qn, av); qn = 0; wave_sync(); } }
        }
      }
      team_sync();
    }
    if constexpr (NOVAL) {                               // what is still queued belongs to this mask row
      if (qn) { flush(qn, T()); qn = 0; }
      team_sync();
    }
    for (uint32_t mp = t; mp < me - mb; mp += TEAM) if (flag[mp]) {                                                         // by mask position: coalesced
      if constexpr (EXACT) a.cacc[mb + mp] = to_word<T>((T)fx_to_fp<fx_bits<T>::P>(acc[mp], hiw[mp], uexp)); else a.cacc[mb + mp] = acc[mp];
      a.cflag[mb + mp] = 1;
    }
    team_sync();
  }
}

// ---- (1b) huge mask rows: dense position map in HBM, one persistent block per map ----------------------------------
// Nearly every product misses the mask, and a lookup in the 4-bytes-per-column map is a random HBM access.  A bit filter
// in LDS (1 Mbit, column mod 2^20: a mask row of 30 000 entries lets ~2 % of the misses through) answers most products
// without leaving the CU; only the survivors read the map for their exact position.
// Round 3: the hits no longer go to HBM one 8-byte atomic at a time (2e9 of them on R-MAT-22's 567 hub rows: 16 GB written for a
// 0.77 GB result).  The mask row's first LC positions — its lowest columns, the hubs every neighbourhood shares, where nearly all
// hits land — have their accumulators in LDS (native ds atomics; 32-bit counters when the product only counts, as PLUS_PAIR does)
// and reach cacc as ONE plain store each at the end of the row; positions behind LC keep the global atomic.  The LDS they take
// comes out of the filter: 2^18 bits instead of 2^20 (a 30 000-entry mask row lets ~11 % of the misses through to the map).
constexpr uint32_t SPG_FILTER_WORDS = 8192;         // 2^18 bits = 32 KiB of LDS
#ifndef SPG_MAP_LDS_KB_V
#define SPG_MAP_LDS_KB_V 96
#endif
constexpr uint32_t SPG_MAP_LDS_BYTES = SPG_MAP_LDS_KB_V * 1024;   // accumulators of the first positions of the mask row
constexpr uint32_t SPG_MAP_SLICE = SPG_SLICE_V;            // entries of A(i,:) per task (1024: 44.2 ms, 2048: 43.4, 4096: 43.4, 8192: 43.1 for the R-MAT-22 triangle count)
// EXACT (deterministic mode): every accumulator is a 128-bit integer (grb_exact.hpp) — low words, high words and flags of the first LCF positions in the
// same LDS budget, the positions behind them and the hand-over of every task in a.xlo / a.xhi (integer atomics: the slices of a row and the workgroups may
// finish in any order); k_exact_finish rounds the integers into cacc afterwards.
template <class T, class SR, bool CNT32, bool EXACT = false>
__global__ __launch_bounds__(1024) void k_spgemm_masked_map(const SpgemmKArgs<T> a, const uint32_t* __restrict__ rows, uint32_t nrows_bin, uint32_t nslices,
                                                            uint32_t* __restrict__ maps, uint32_t ncols, const SR sr) {
  static_assert(!(CNT32 && EXACT), "counting products are integers");
  typedef typename acc_word<T>::type W;
  typedef typename std::conditional<CNT32, uint32_t, typename std::conditional<EXACT, unsigned long long, W>::type>::type LW;        // what an LDS accumulator holds
  constexpr uint32_t LC = SPG_MAP_LDS_BYTES / sizeof(LW);
  // counters: 0 = no hit yet.  Generic accumulators start at the identity and carry a flag byte each (LCF words + LCF bytes in the same budget)
  constexpr uint32_t LCF = CNT32 ? LC : (uint32_t)(SPG_MAP_LDS_BYTES / ((EXACT ? 16 : sizeof(LW)) + 1));
  __shared__ uint32_t s_filter[SPG_FILTER_WORDS];
  // bit of column j: a multiplicative hash (R-MAT labels are skewed bit by bit: `j mod 2^18` crowds the filter's low words)
  constexpr int FSH = 32 - (__builtin_ctz(SPG_FILTER_WORDS) + 5);
  auto fbit = [](const uint32_t j) -> uint32_t { return (uint32_t)__umul24(j, SPG_FILTER_MUL) >> FSH; };
  __shared__ LW s_acc[LC];
  unsigned long long* const s_hiw = (unsigned long long*)(s_acc + LCF);          // (EXACT only)
  uint8_t* const s_flag = (uint8_t*)(s_acc + (EXACT ? 2 * LCF : LCF));
  int uexp = 0;
  constexpr uint32_t HL = 4096;                            // A-entries whose B row is huge: set aside by the waves, then walked by the whole block
  __shared__ uint32_t s_hl[HL];
  __shared__ uint32_t s_nh;
  uint32_t* map = maps + (size_t)blockIdx.x * ncols;       // zero-initialised; entry = mask position + 1
  const int t = threadIdx.x;
  const bool use_a = sr.uses_a(), use_b = sr.uses_u();
  const W idw = to_word<T>(sr.identity);
  // one B row against the filter and the map: `step` lanes apart, 4 loads in flight per lane
  auto hit = [&](const uint32_t slot1, const uint32_t pb, const T av, const uint32_t mb) {
    const uint32_t mp = slot1 - 1;
    if (mp < LCF) {
      if constexpr (CNT32) atomicAdd(&s_acc[mp], 1u);             // PLUS_PAIR: the product is 1
      else if constexpr (EXACT) { fx_add_lds(fx_lds_addr(&s_acc[mp]), fx_lds_addr(&s_hiw[mp]), (double)sr.mult(av, use_b ? a.bval[pb] : T()), uexp); s_flag[mp] = 1; }
      else { word_combine<T>(sr.add_op(), (W*)&s_acc[mp], sr.mult(av, use_b ? a.bval[pb] : T())); s_flag[mp] = 1; }
    } else {
      if constexpr (EXACT) fx_add(&a.xlo[mb + mp], &a.xhi[mb + mp], (double)sr.mult(av, use_b ? a.bval[pb] : T()), uexp);
      else word_combine<T>(sr.add_op(), &a.cacc[mb + mp], sr.mult(av, use_b ? a.bval[pb] : T()));
      a.cflag[mb + mp] = 1;
    }
  };
  auto walk = [&](uint32_t pa, uint32_t first, uint32_t be, uint32_t step, uint32_t mb) {
    const T av = use_a ? a.aval[pa] : T();
    // whole blocks first: 16 bytes per lane and load, two loads = 8 entries per lane in flight, then up to eight map reads in flight
    uint32_t base = first - (step == 64 ? (uint32_t)(threadIdx.x & 63) : (uint32_t)threadIdx.x);
    const uint32_t ln = step == 64 ? (uint32_t)(threadIdx.x & 63) : (uint32_t)threadIdx.x;
    // (software-pipelined: the next block's B-row loads are issued before this block's map reads are consumed — the kernel runs
    // four waves per SIMD and was parked on two dependent memory round trips per block)
    if (base + 8 * step <= be) {
      spg_u4 j4[2];
#pragma unroll
      for (int u = 0; u < 2; u++) j4[u] = spg_ld4(a.bcol, base + 4 * step * u + 4 * ln);
      for (; base + 8 * step <= be; base += 8 * step) {
        uint32_t ss[8];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const uint32_t jv[4] = {j4[u].x, j4[u].y, j4[u].z, j4[u].w};
#pragma unroll
          for (int c = 0; c < 4; c++) { const uint32_t j = jv[c]; const uint32_t fb = fbit(j); ss[4 * u + c] = ((s_filter[fb >> 5] >> (fb & 31)) & 1u) ? map[j] : 0u; }
        }
        if (base + 16 * step <= be) {
#pragma unroll
          for (int u = 0; u < 2; u++) j4[u] = spg_ld4(a.bcol, base + 8 * step + 4 * step * u + 4 * ln);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) if (ss[u]) hit(ss[u], base + 4 * step * (u >> 2) + 4 * ln + (u & 3), av, mb);
      }
    }
    for (uint32_t pb0 = base + ln; pb0 < be; pb0 += 4 * step) {
      uint32_t ss[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const uint32_t pb = pb0 + step * u; ss[u] = a.bcol[pb < be ? pb : be - 1]; }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t pb = pb0 + step * u; const uint32_t j = ss[u];
        const uint32_t fb = fbit(j); const bool maybe = pb < be && ((s_filter[fb >> 5] >> (fb & 31)) & 1u);
        ss[u] = maybe ? map[j] : 0u;                                     // the survivors: exact position from the map
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t pb = pb0 + step * u;
        if (ss[u]) hit(ss[u], pb, av, mb);
      }
    }
  };
  // a task = SPG_MAP_SLICE consecutive entries of one row's A(i,:): a hub row of 50 000 entries against B rows of thousands is 2e8
  // products — one workgroup alone on it was the tail of the whole product (33 ms for 4 % of the products).  Tasks are numbered
  // slice-major, so the slices of the heaviest rows spread over all workgroups; every task builds the row's map for itself.
  for (uint64_t q = blockIdx.x; q < (uint64_t)nrows_bin * nslices; q += gridDim.x) {
    const uint32_t ridx = (uint32_t)(q % nrows_bin), sl = (uint32_t)(q / nrows_bin);
    const uint32_t i = rows[ridx];
    const uint32_t ab0 = a.arp[i], ae0 = a.arp[i + 1];
    if ((uint64_t)ab0 + (uint64_t)sl * SPG_MAP_SLICE >= ae0) continue;                      // (the whole workgroup takes the same branch)
    const uint32_t ab = ab0 + sl * SPG_MAP_SLICE, ae = ae0 - ab < SPG_MAP_SLICE ? ae0 : ab + SPG_MAP_SLICE;
    const bool shared_row = ae0 - ab0 > SPG_MAP_SLICE;                                      // other workgroups accumulate into the same slots
    const uint32_t mb = a.mrp[i], me = a.mrp[i + 1];
    for (uint32_t w = t; w < SPG_FILTER_WORDS; w += 1024) s_filter[w] = 0;
    const uint32_t nl = (me - mb) < LCF ? (me - mb) : LCF;
    if constexpr (EXACT) uexp = __builtin_amdgcn_readfirstlane(a.rowexp[i]);
    for (uint32_t w = t; w < nl; w += 1024) { if constexpr (CNT32) s_acc[w] = 0; else if constexpr (EXACT) { s_acc[w] = 0; s_hiw[w] = 0; s_flag[w] = 0; } else { s_acc[w] = (LW)idw; s_flag[w] = 0; } }
    if (t == 0) s_nh = 0;
    __syncthreads();
    for (uint32_t p = mb + t; p < me; p += 1024) if (spgemm_mask_truth(a.mval, a.mcode, p, a.mstruct)) {
      const uint32_t j = a.mcol[p];
      map[j] = p - mb + 1; const uint32_t fb = fbit(j); atomicOr(&s_filter[fb >> 5], 1u << (fb & 31));
    }
    __threadfence_block(); __syncthreads();
    // one wave per entry k of A(i,:): these rows have thousands of k's, most with long B rows; the longest ones are kept
    // for the whole block (a wave alone on a 30 000-entry row leaves the other fifteen waiting at the end of the row)
    for (uint32_t pa = ab + (t >> 6); pa < ae; pa += 16) {
      const uint32_t k = a.acol[pa];
      const uint32_t bb = a.brp[k], be = a.brp[k + 1];
      if (be - bb >= SPG_HUBHUGE_V) {
        uint32_t slot = HL;
        if ((t & 63) == 0) slot = atomicAdd(&s_nh, 1u);
        slot = (uint32_t)__shfl((int)slot, 0, 64);
        if (slot < HL) { if ((t & 63) == 0) s_hl[slot] = pa; continue; }
      }
      walk(pa, bb + (t & 63), be, 64, mb);
    }
    __syncthreads();
    const uint32_t nh = s_nh < HL ? s_nh : HL;
    for (uint32_t q = 0; q < nh; q++) {
      const uint32_t pa = s_hl[q], k = a.acol[pa];
      walk(pa, a.brp[k] + t, a.brp[k + 1], 1024, mb);
    }
    __syncthreads();
    // the LDS accumulators leave coalesced: plain stores when this task is the row's only one, else one atomic per touched position
    for (uint32_t w = t; w < nl; w += 1024) {
      if constexpr (CNT32) {
        const uint32_t c = s_acc[w];
        if (c) { if (shared_row) word_combine<T>(B_PLUS, &a.cacc[mb + w], (T)c); else a.cacc[mb + w] = to_word<T>((T)c); a.cflag[mb + w] = 1; }
      } else if constexpr (EXACT) {
        if (s_flag[w]) { fx_add_words(&a.xlo[mb + w], &a.xhi[mb + w], (unsigned long long)s_acc[w], s_hiw[w]); a.cflag[mb + w] = 1; }
      } else if (s_flag[w]) {
        if (shared_row) word_combine<T>(sr.add_op(), &a.cacc[mb + w], from_word<T>((W)s_acc[w])); else a.cacc[mb + w] = (W)s_acc[w];
        a.cflag[mb + w] = 1;
      }
    }
    for (uint32_t p = mb + t; p < me; p += 1024) map[a.mcol[p]] = 0;
    __threadfence_block(); __syncthreads();
  }
}

template <class W> __global__ void k_fill_words(W* p, uint64_t n, W v) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = v;
}

// compaction of the per-mask-entry accumulators into CSR values
template <class T> __global__ void k_compact_acc(uint32_t nrows, const uint32_t* __restrict__ mrp, const uint32_t* __restrict__ mcol,
                                                 const typename acc_word<T>::type* __restrict__ cacc, const uint8_t* __restrict__ cflag,
                                                 const uint32_t* __restrict__ orp, uint32_t* __restrict__ ocol, T* __restrict__ oval) {
  for (uint64_t r = blockIdx.x * 256ull + threadIdx.x; r < nrows; r += (uint64_t)gridDim.x * 256ull) {
    uint32_t w = orp[r];
    for (uint32_t p = mrp[r]; p < mrp[r + 1]; p++) if (cflag[p]) { ocol[w] = mcol[p]; oval[w] = from_word<T>(cacc[p]); w++; }
  }
}

// Rows into the five bins.  A workgroup owns a contiguous range of rows: it counts its rows per bin in LDS (one LDS atomic per wave
// and bin), reserves its part of every list with ONE global atomic per bin, and writes its rows behind that base — 1 280 global atomics
// for any number of rows (one per wave and bin, on five addresses, was 1.2 ms of the R-MAT-22 triangle count: same-address
// atomics complete one at a time).
// ---- (1c) deterministic mode of the masked product (round 5; GRB_MI355X_DETERMINISTIC=1 / GxB_AxB_GUSTAVSON, floating point only) ----------------
// The teams above let every 16-lane group and every wave of a row add into the row's accumulators as their atomics land.  Here ONE group of G lanes owns a
// work item = (row i, slice of at most CAP entries of M(i,:)): the slice's columns (sorted, as every CSR row is) and its accumulators sit in the group's
// own LDS; the group walks the entries k of A(i,:) one after the other, its lanes the entries of B(k,:) — columns of one B row are distinct, so the lanes
// of a step touch distinct accumulators with plain read-add-write, and an accumulator receives its terms in k order in every run, whatever the
// schedule.  A product finds its position by a range check and a binary search of the slice (no hash, no filter).  Mask rows longer than CAP are cut into
// slices that different waves take (each walks all of the row's products and keeps those of its columns): the price of never sharing an accumulator.
// G = 16 for mask rows of <= 32 entries (four rows per wave), G = 64 with slices of 512 for the rest.
constexpr int SPG_ORD_CAP16 = 32, SPG_ORD_CAP64 = 512;
static __global__ void k_ordered_items(const uint32_t* __restrict__ rows, uint32_t nrows_bin, const uint32_t* __restrict__ mrp, uint32_t* __restrict__ item_row,
                                       uint32_t* __restrict__ item_slice, uint32_t* __restrict__ count) {
  for (uint64_t q = blockIdx.x * 256ull + threadIdx.x; q < nrows_bin; q += gridDim.x * 256ull) {
    const uint32_t i = rows[q], ns = (mrp[i + 1] - mrp[i] + SPG_ORD_CAP64 - 1) / SPG_ORD_CAP64;
    const uint32_t base = atomicAdd(count, ns);
    for (uint32_t z = 0; z < ns; z++) { item_row[base + z] = i; item_slice[base + z] = z; }
  }
}
template <class T, class SR, int G>
__global__ __launch_bounds__(256) void k_spgemm_masked_ordered(const SpgemmKArgs<T> a, const uint32_t* __restrict__ item_row, const uint32_t* __restrict__ item_slice,
                                                               uint32_t nitems, const uint32_t* __restrict__ nitems_p, const SR sr) {
  constexpr int GROUPS = 256 / G, CAP = G == 16 ? SPG_ORD_CAP16 : SPG_ORD_CAP64;
  __shared__ uint32_t s_col[GROUPS][CAP];
  __shared__ T s_acc[GROUPS][CAP];
  __shared__ uint8_t s_flag[GROUPS][CAP];          // bit 1: the mask entry is true, bit 0: a product landed
  const int g = threadIdx.x / G, t = threadIdx.x % G;
  uint32_t* col = s_col[g]; T* acc = s_acc[g]; uint8_t* flag = s_flag[g];
  const bool use_a = sr.uses_a(), use_b = sr.uses_u();
  auto group_sync = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };      // a group never spans waves: LDS operations of a wave execute in order
  const uint32_t n = nitems_p ? *nitems_p : nitems;
  for (uint64_t it = (uint64_t)blockIdx.x * GROUPS + g; it < n; it += (uint64_t)gridDim.x * GROUPS) {
    const uint32_t i = item_row[it], sl = item_slice ? item_slice[it] : 0u;
    const uint32_t mb = a.mrp[i] + sl * CAP, mend = a.mrp[i + 1], me = mb + CAP < mend ? mb + CAP : mend, len = me - mb;
    group_sync();
    for (uint32_t q = t; q < len; q += G) {
      col[q] = a.mcol[mb + q]; acc[q] = sr.identity;
      flag[q] = spgemm_mask_truth(a.mval, a.mcode, mb + q, a.mstruct) ? 2 : 0;
    }
    group_sync();
    const uint32_t lo = col[0], hi = col[len - 1];
    const uint32_t ab = a.arp[i], ae = a.arp[i + 1];
    uint32_t bb = 0, be = 0; T av = T();
    if (ab < ae) { const uint32_t k = a.acol[ab]; bb = a.brp[k]; be = a.brp[k + 1]; if (use_a) av = a.aval[ab]; }
    for (uint32_t pa = ab; pa < ae; pa++) {
      uint32_t nbb = 0, nbe = 0; T nav = T();                  // the next entry's B row is looked up while this one is walked
      if (pa + 1 < ae) { const uint32_t k = a.acol[pa + 1]; nbb = a.brp[k]; nbe = a.brp[k + 1]; if (use_a) nav = a.aval[pa + 1]; }
      for (uint32_t pb = bb + t; pb < be; pb += G) {
        const uint32_t j = a.bcol[pb];
        if (j < lo || j > hi) continue;
        uint32_t l = 0, h = len;
        while (h - l > 1) { const uint32_t mid = (l + h) >> 1; if (col[mid] <= j) l = mid; else h = mid; }
        if (col[l] == j && (flag[l] & 2)) { acc[l] = sr.add(acc[l], sr.mult(av, use_b ? a.bval[pb] : T())); flag[l] = 3; }
      }
      asm volatile("" ::: "memory");
      bb = nbb; be = nbe; av = nav;
    }
    group_sync();
    for (uint32_t q = t; q < len; q += G) if (flag[q] & 1) { a.cacc[mb + q] = to_word<T>(acc[q]); a.cflag[mb + q] = 1; }
  }
}

// ---- (1d) the exact accumulators' row units (grb_exact.hpp) ---------------------------------------------------------------------------------------
// |x| as an ordered integer: NaN above Inf above every finite value, so an integer max carries "not finite" along
__device__ __forceinline__ unsigned long long fx_abs_bits(const double v) { return (unsigned long long)__double_as_longlong(v) & 0x7FFFFFFFFFFFFFFFull; }
// largest and smallest non-zero |value| of B as ordered integers, out[0] = max (NaN above Inf above finite), out[1] = min over the non-zero finite values (~0: none)
template <class T> __global__ __launch_bounds__(256) void k_abs_minmax(uint64_t n, const T* __restrict__ val, unsigned long long* __restrict__ out) {
  unsigned long long mx = 0, mn = ~0ull;
  for (uint64_t p = blockIdx.x * 256ull + threadIdx.x; p < n; p += gridDim.x * 256ull) {
    const unsigned long long b = fx_abs_bits((double)val[p]);
    mx = b > mx ? b : mx; if (b && b <= 0x7FEFFFFFFFFFFFFFull) mn = b < mn ? b : mn;
  }
  for (int o = 32; o; o >>= 1) { const unsigned long long x = __shfl_xor(mx, o, 64), y = __shfl_xor(mn, o, 64); mx = x > mx ? x : mx; mn = y < mn ? y : mn; }
  if ((threadIdx.x & 63) == 0) { atomicMax(&out[0], mx); atomicMin(&out[1], mn); }
}
inline bool exact_mult_supported(int mulop) {
  switch (mulop) { case B_TIMES: case B_FIRST: case B_SECOND: case B_PAIR: case B_PLUS: case B_MINUS: case B_RMINUS: case B_MIN: case B_MAX: case B_ANY: return true; default: return false; }
}
// The unit exponent of every output row (grb_exact.hpp), from the largest and smallest non-zero |A(i,k)| of the row and of B as a whole:
//   2^E above the largest |product| the row can form — the multiply applied to the two maxima, in T, rounded as a product is (rounding is monotonic);
//   2^H above the number of products one entry can receive (the entries of A(i,:));   u = E + H - 126.
// A row is EXACT only if no product can have a bit below 2^u: the lowest bit of a product lies at or above 2^(e - 52), e the exponent of the multiply applied
// to the two minima (TIMES), of the smaller minimum (PLUS / MINUS / MIN / MAX / ANY: sums and differences of doubles are multiples of the smaller operand's
// last place), of the one operand read (FIRST / SECOND).  Rows that fail — operands spread over more than ~2^(74-H) — and rows with an Inf / NaN bound
// are marked FX_NO_EXP and formed by k_spgemm_masked_ordered.  256 consecutive rows per workgroup, their entries dealt to the threads 256 apart; a thread
// keeps the extremes of the row it is in and hands them to the row's LDS slot when it moves on (a hub row: one atomic per thread, not one per entry).
template <class T> __global__ __launch_bounds__(256) void k_row_unit_exp(uint32_t nrows, const uint32_t* __restrict__ arp, const T* __restrict__ aval,
                                                                         const unsigned long long* __restrict__ bmm, int mulop, int32_t* __restrict__ out) {
  __shared__ uint32_t s_rp[257]; __shared__ unsigned long long s_max[256], s_min[256];
  const uint32_t r0 = blockIdx.x * 256u, nr = nrows - r0 < 256u ? nrows - r0 : 256u;
  for (uint32_t q = threadIdx.x; q <= nr; q += 256) s_rp[q] = arp[r0 + q];
  s_max[threadIdx.x] = 0; s_min[threadIdx.x] = ~0ull;
  __syncthreads();
  if (aval) {
    const uint32_t e1 = s_rp[nr];
    uint32_t cur = 0, cur_end = 0; unsigned long long mx = 0, mn = ~0ull; bool have = false;
    for (uint32_t p = s_rp[0] + threadIdx.x; p < e1; p += 256) {
      if (!have || p >= cur_end) {
        if (have) { atomicMax(&s_max[cur], mx); atomicMin(&s_min[cur], mn); }
        uint32_t lo = 0, hi = nr;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_rp[mid] <= p) lo = mid; else hi = mid; }
        cur = lo; cur_end = s_rp[lo + 1]; mx = 0; mn = ~0ull; have = true;
      }
      const unsigned long long b = fx_abs_bits((double)aval[p]);
      mx = b > mx ? b : mx; if (b && b <= 0x7FEFFFFFFFFFFFFFull) mn = b < mn ? b : mn;
    }
    if (have) { atomicMax(&s_max[cur], mx); atomicMin(&s_min[cur], mn); }
    __syncthreads();
  }
  const uint32_t r = r0 + threadIdx.x;
  if (r >= nrows) return;
  const uint32_t al = s_rp[threadIdx.x + 1] - s_rp[threadIdx.x];
  if (!al) { out[r] = 0; return; }
  const unsigned long long one = fx_abs_bits(1.0);
  const unsigned long long amx = aval ? s_max[threadIdx.x] : one, amn = aval ? s_min[threadIdx.x] : one, bmx = bmm ? bmm[0] : one, bmn = bmm ? bmm[1] : one;
  auto val_of = [](const unsigned long long b) { return (T)__longlong_as_double((long long)b); };
  auto bits_of = [](const T v) { return fx_abs_bits((double)v); };
  unsigned long long top, low;            // the largest |product|; a value whose exponent bounds the lowest bit of every non-zero product from below
  const bool a_none = amn == ~0ull, b_none = bmn == ~0ull;          // no non-zero finite value on that side
  switch (mulop) {
    case B_TIMES: top = bits_of(val_of(amx) * val_of(bmx)); low = (a_none || b_none) ? ~0ull : bits_of(val_of(amn) * val_of(bmn)); if (!a_none && !b_none && low == 0) low = 1; break;   // (underflow: the last place of the subnormals)
    case B_FIRST: top = amx; low = amn; break;
    case B_SECOND: top = bmx; low = bmn; break;
    case B_PAIR: top = one; low = one; break;
    case B_PLUS: case B_MINUS: case B_RMINUS: top = bits_of(val_of(amx) + val_of(bmx)); low = amn < bmn ? amn : bmn; break;
    default: top = amx > bmx ? amx : bmx; low = amn < bmn ? amn : bmn; break;       // MIN, MAX, ANY
  }
  int32_t u = FX_NO_EXP;
  if (top <= 0x7FEFFFFFFFFFFFFFull && amx <= 0x7FEFFFFFFFFFFFFFull && bmx <= 0x7FEFFFFFFFFFFFFFull) {
    const int E = top ? ilogb(__longlong_as_double((long long)top)) + 2 : -1074;
    u = E + (32 - __clz((int)al)) - 126;
    if (u < -1074) u = -1074;             // (no double has a bit below 2^-1074: a smaller unit would only push subnormal terms beyond the 73-bit shift of fx_from_double)
    if (low != ~0ull) {                    // (no non-zero product at all: every sum is 0)
      int lowbit = ilogb(__longlong_as_double((long long)low)) - 52; if (lowbit < -1074) lowbit = -1074;
      if (lowbit < u) u = FX_NO_EXP;
    }
  }
  out[r] = u;
}
static __global__ void k_rows_without_unit(uint32_t nrows, const int32_t* __restrict__ rowexp, const uint32_t* __restrict__ mrp, const uint32_t* __restrict__ arp,
                                           uint32_t* __restrict__ list, uint32_t* __restrict__ count) {
  for (uint64_t r = blockIdx.x * 256ull + threadIdx.x; r < nrows; r += gridDim.x * 256ull)
    if (rowexp[r] == FX_NO_EXP && mrp[r + 1] > mrp[r] && arp[r + 1] > arp[r]) list[atomicAdd(count, 1u)] = (uint32_t)r;
}
// the rows that accumulated in a.xlo / a.xhi (the HBM-map bin): the integers rounded once into the accumulator words the compaction reads
template <class T> __global__ void k_exact_finish(const uint32_t* __restrict__ rows, uint32_t nrows_bin, const uint32_t* __restrict__ mrp, const int32_t* __restrict__ rowexp,
                                                  const unsigned long long* __restrict__ xlo, const unsigned long long* __restrict__ xhi, const uint8_t* __restrict__ cflag,
                                                  typename acc_word<T>::type* __restrict__ cacc) {
  for (uint32_t q = blockIdx.x; q < nrows_bin; q += gridDim.x) {
    const uint32_t i = rows[q]; const int u = rowexp[i];
    for (uint32_t p = mrp[i] + threadIdx.x; p < mrp[i + 1]; p += blockDim.x) if (cflag[p]) cacc[p] = to_word<T>((T)fx_to_fp<fx_bits<T>::P>(xlo[p], xhi[p], u));
  }
}

constexpr int SPG_BIN_ROWS = 16;      // rows per thread
static __global__ __launch_bounds__(1024) void k_bin_rows(uint32_t nrows, const uint32_t* __restrict__ mrp, const uint32_t* __restrict__ arp, uint32_t* __restrict__ counts,
                                  uint32_t* __restrict__ lists /* 5 x nrows */, uint32_t lim3 /* longest mask row of the last LDS bin */,
                                  const int32_t* __restrict__ rowexp /* exact mode: rows marked FX_NO_EXP are left out */) {
  __shared__ uint32_t s_cnt[5], s_base[5], s_max;
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 5) s_cnt[threadIdx.x] = 0;
  if (threadIdx.x == 5) s_max = 0;
  __syncthreads();
  const uint64_t per_block = 1024ull * SPG_BIN_ROWS;
  const uint64_t r0 = (uint64_t)blockIdx.x * per_block;
  int8_t bin[SPG_BIN_ROWS]; uint32_t rank[SPG_BIN_ROWS]; uint32_t al4 = 0;
#pragma unroll
  for (int it = 0; it < SPG_BIN_ROWS; it++) {
    const uint64_t r = r0 + (uint64_t)it * 1024 + threadIdx.x;
    int b = -1;
    if (r < nrows) {
      const uint32_t ml = mrp[r + 1] - mrp[r], al = arp[r + 1] - arp[r];
      if (ml && al && !(rowexp && rowexp[r] == FX_NO_EXP)) b = ml <= 32 ? 0 : (ml <= 256 ? 1 : (ml <= 1024 ? 2 : (ml <= lim3 ? 3 : 4)));
      if (b == 4 && al > al4) al4 = al;
    }
    bin[it] = (int8_t)b; rank[it] = 0;
    for (int bb = 0; bb < 5; bb++) {
      const unsigned long long m = __ballot(b == bb);
      if (!m) continue;
      const int leader = __builtin_ctzll(m);
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(&s_cnt[bb], (uint32_t)__popcll(m));
      base = __shfl(base, leader, 64);
      if (b == bb) rank[it] = base + __popcll(m & ((1ull << lane) - 1));
    }
  }
  // counts[5]: the longest A row among the hub-bin rows (their A rows are cut into slices shared by all workgroups)
  if (__ballot(al4 != 0)) { al4 = __builtin_amdgcn_wave_reduce_max_u32(al4, 0); if (lane == 0) atomicMax(&s_max, al4); }
  __syncthreads();
  if (threadIdx.x < 5) s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]) : 0u;
  if (threadIdx.x == 5 && s_max) atomicMax(&counts[5], s_max);
  __syncthreads();
#pragma unroll
  for (int it = 0; it < SPG_BIN_ROWS; it++) {
    const int b = bin[it];
    if (b >= 0) lists[(size_t)b * nrows + s_base[b] + rank[it]] = (uint32_t)(r0 + (uint64_t)it * 1024 + threadIdx.x);
  }
}

// entry-parallel compaction of the per-mask-entry accumulators: pos = exclusive scan of the flags
static __global__ void k_flags_to_u32(const uint8_t* __restrict__ flag, uint64_t n, uint32_t* __restrict__ out) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) out[i] = flag[i] ? 1u : 0u;
}
static __global__ void k_rowptr_from_scan(uint32_t nrows, const uint32_t* __restrict__ mrp, const uint32_t* __restrict__ pos, uint32_t total, uint64_t mnz, uint32_t* __restrict__ orp) {
  for (uint64_t r = blockIdx.x * 256ull + threadIdx.x; r <= nrows; r += (uint64_t)gridDim.x * 256ull) {
    const uint32_t p = mrp[r]; orp[r] = p < mnz ? pos[p] : total;
  }
}
template <class T> __global__ void k_scatter_acc(uint64_t mnz, const uint32_t* __restrict__ mcol, const typename acc_word<T>::type* __restrict__ cacc,
                                                 const uint8_t* __restrict__ cflag, const uint32_t* __restrict__ pos, uint32_t* __restrict__ ocol, T* __restrict__ oval) {
  for (uint64_t p = blockIdx.x * 256ull + threadIdx.x; p < mnz; p += gridDim.x * 256ull)
    if (cflag[p]) { const uint32_t w = pos[p]; ocol[w] = mcol[p]; oval[w] = from_word<T>(cacc[p]); }
}
template <class T> void run_spgemm_masked(const SpgemmCall& c, const SemiringDesc& d, DevCSR& out) {
  typedef typename acc_word<T>::type W;
  const DevCSR& A = *c.A; const DevCSR& B = *c.B; const DevCSR& M = *c.M;
  const uint32_t nrows = A.nrows; const uint64_t mnz = M.nnz;
  out.clear(); out.nrows = nrows; out.ncols = B.ncols;
  out.rowptr.alloc(((size_t)nrows + 1) * 4);
  if (!mnz || !A.nnz || !B.nnz) { GRB_HIP(hipMemsetAsync(out.rowptr.p, 0, ((size_t)nrows + 1) * 4, stream())); out.nnz = 0; out.valid = true; return; }
  DevBuf cacc(mnz * sizeof(W)), cflag(mnz), counts(32), lists((size_t)5 * nrows * 4 + 4);
  GRB_HIP(hipMemsetAsync(cflag.p, 0, mnz, stream()));
  GRB_HIP(hipMemsetAsync(counts.p, 0, 32, stream()));
  // Deterministic mode (c.ordered, floating point).  PLUS monoid over a multiply whose size can be bounded from the operands: the usual kernels with EXACT
  // accumulators — 128-bit integers in a per-row unit, order-independent (grb_exact.hpp); rows whose bound is not finite, and every other semiring:
  // k_spgemm_masked_ordered, one group per slice of a mask row with the entries of A(i,:) one after the other.
  constexpr bool is_fp = std::is_floating_point<T>::value;
  const bool exact = is_fp && c.ordered && d.addop == B_PLUS && !d.flip && exact_mult_supported(d.mulop) && !getenv("GRB_MI355X_NO_EXACT");
  DevBuf rowexp, noexp_rows, xlo, xhi, bmax;
  if constexpr (is_fp) if (exact) {
    if (c.bval) {
      const unsigned long long init[2] = {0ull, ~0ull};
      bmax.alloc(16); GRB_HIP(hipMemcpyAsync(bmax.p, init, 16, hipMemcpyHostToDevice, stream()));
      hipLaunchKernelGGL((k_abs_minmax<T>), dim3((unsigned)std::min<uint64_t>((B.nnz + 255) / 256, 2048)), dim3(256), 0, stream(), (uint64_t)B.nnz, (const T*)c.bval, bmax.as<unsigned long long>());
    }
    rowexp.alloc((size_t)nrows * 4 + 4); noexp_rows.alloc((size_t)nrows * 4 + 4);
    hipLaunchKernelGGL((k_row_unit_exp<T>), dim3((nrows + 255) / 256), dim3(256), 0, stream(), nrows, A.rowptr.as<uint32_t>(), (const T*)c.aval,
                       c.bval ? bmax.as<unsigned long long>() : (const unsigned long long*)nullptr, d.mulop, rowexp.as<int32_t>());
    hipLaunchKernelGGL(k_rows_without_unit, dim3((unsigned)std::min<uint64_t>(((uint64_t)nrows + 255) / 256, 4096)), dim3(256), 0, stream(), nrows, rowexp.as<int32_t>(),
                       M.rowptr.as<uint32_t>(), A.rowptr.as<uint32_t>(), noexp_rows.as<uint32_t>(), counts.as<uint32_t>() + 6);
  }
  hipLaunchKernelGGL(k_bin_rows, dim3((unsigned)(((uint64_t)nrows + 1024ull * SPG_BIN_ROWS - 1) / (1024ull * SPG_BIN_ROWS))), dim3(1024), 0, stream(), nrows, M.rowptr.as<uint32_t>(), A.rowptr.as<uint32_t>(), counts.as<uint32_t>(), lists.as<uint32_t>(),
                     4096u, exact ? rowexp.as<int32_t>() : (const int32_t*)nullptr);
  uint32_t hc[8];
  GRB_HIP(hipMemcpyAsync(hc, counts.p, 32, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
  if (exact && hc[4]) { xlo.alloc(mnz * 8 + 8); xhi.alloc(mnz * 8 + 8); GRB_HIP(hipMemsetAsync(xlo.p, 0, mnz * 8, stream())); GRB_HIP(hipMemsetAsync(xhi.p, 0, mnz * 8, stream())); }
  with_semiring<T>(d, [&](auto sr) {
    typedef decltype(sr) SR;
    SpgemmKArgs<T> a{A.rowptr.as<uint32_t>(), A.col.as<uint32_t>(), (const T*)c.aval, B.rowptr.as<uint32_t>(), B.col.as<uint32_t>(), (const T*)c.bval,
                     M.rowptr.as<uint32_t>(), M.col.as<uint32_t>(), M.val.p, c.mcode, c.mstruct, cacc.as<W>(), cflag.as<uint8_t>(),
                     exact ? rowexp.as<int32_t>() : (const int32_t*)nullptr, xlo.as<unsigned long long>(), xhi.as<unsigned long long>()};
    const uint32_t* L = lists.as<uint32_t>();
    // the ordered kernel over (row, slice) work items: `lists_in` = up to four row lists, longest mask rows first; rows of <= 32 mask entries as 16-lane groups
    [[maybe_unused]] auto run_ordered = [&](const uint32_t* const* lists_in, const uint32_t* counts_in, int nlists, const uint32_t* small, uint32_t nsmall) {
      if constexpr (is_fp) {
        uint64_t nbig = 0; for (int q = 0; q < nlists; q++) nbig += counts_in[q];
        const uint64_t cap = nbig + mnz / SPG_ORD_CAP64 + 1;
        DevBuf irow(cap * 4 + 4), islice(cap * 4 + 4), icount(4);
        GRB_HIP(hipMemsetAsync(icount.p, 0, 4, stream()));
        for (int q = 0; q < nlists; q++) if (counts_in[q])
          hipLaunchKernelGGL(k_ordered_items, dim3((unsigned)std::min<uint64_t>(((uint64_t)counts_in[q] + 255) / 256, 4096)), dim3(256), 0, stream(), lists_in[q], counts_in[q],
                             M.rowptr.as<uint32_t>(), irow.as<uint32_t>(), islice.as<uint32_t>(), icount.as<uint32_t>());
        if (nbig) hipLaunchKernelGGL((k_spgemm_masked_ordered<T, SR, 64>), dim3(256 * 6), dim3(256), 0, stream(), a, irow.as<uint32_t>(), islice.as<uint32_t>(), 0u, icount.as<uint32_t>(), sr);
        if (nsmall) hipLaunchKernelGGL((k_spgemm_masked_ordered<T, SR, 16>), dim3((unsigned)std::min<uint64_t>(((uint64_t)nsmall + 15) / 16, 2048)), dim3(256), 0, stream(), a, small,
                                       (const uint32_t*)nullptr, nsmall, (const uint32_t*)nullptr, sr);
        GRB_HIP(hipStreamSynchronize(stream()));          // the item lists go back to the pool when this scope ends
        g_last_plan += std::string("k_spgemm_masked_ordered<") + (sr.is_static ? "static" : "dynamic") + "> rows " + std::to_string(nsmall) + " + " + std::to_string(nbig) + " ";
      }
    };
    // (MIN / MAX monoids give the same bits in any order: the default kernels are their deterministic mode)
    if constexpr (is_fp) if (c.ordered && !exact && d.addop != B_MIN && d.addop != B_MAX) {
      const uint32_t* ls[4] = {L + (size_t)4 * nrows, L + (size_t)3 * nrows, L + (size_t)2 * nrows, L + (size_t)nrows}; const uint32_t cs[4] = {hc[4], hc[3], hc[2], hc[1]};
      run_ordered(ls, cs, 4, L, hc[0]);
      return;
    }
    auto nblocks = [](uint32_t rows, int teams) { uint64_t b = ((uint64_t)rows + teams - 1) / teams; if (b > 256u * 64) b = 256u * 64; if (b < 1) b = 1; return (unsigned)b; };
    // the HBM-map kernel accumulates straight into cacc: start those slots at the identity (before any kernel writes results)
    if (hc[4]) hipLaunchKernelGGL((k_fill_words<W>), dim3(4096), dim3(256), 0, stream(), cacc.as<W>(), mnz, to_word<T>(sr.identity));
    // the bins write disjoint accumulator slots and each is dominated by a few heavy rows: run them concurrently on
    // auxiliary streams (forked from / joined back into the library stream with events) so their tails overlap
    // everything that can throw (allocation of the position maps of the HBM-map bin) happens before the fork; between fork and
    // join only kernel launches are issued, and the guard joins the side streams again if anything unwinds past it — the
    // accumulators return to the pool only after every bin kernel has been ordered before the library stream
    DevBuf maps; unsigned nb_map = 0; uint32_t nslices = 1;
    if (hc[4]) {
      const uint64_t per_map = (uint64_t)B.ncols * 4, budget = 4ull << 30;          // position maps: at most 4 GiB in all
      uint64_t fit = per_map ? budget / per_map : 256; if (fit < 1) fit = 1;
      nslices = (hc[5] + SPG_MAP_SLICE - 1) / SPG_MAP_SLICE; if (nslices < 1) nslices = 1;
      nb_map = (unsigned)std::min<uint64_t>(std::min<uint64_t>((uint64_t)hc[4] * nslices, 256), fit);
      maps.alloc((size_t)nb_map * per_map);
      GRB_HIP(hipMemsetAsync(maps.p, 0, (size_t)nb_map * per_map, stream()));
    }
    AuxStreams& ax = aux_streams();
    struct ForkGuard { AuxStreams& a; bool joined = false; ~ForkGuard() { if (!joined) { try { a.join(stream()); } catch (...) {} (void)hipStreamSynchronize(stream()); } } } guard{ax};
    ax.fork(stream());
    const bool serial = getenv("GRB_MI355X_SPGEMM_SERIAL") != nullptr;      // experiment hook: run the bins one after the other
    hipStream_t bs[4]; for (int q = 0; q < 4; q++) bs[q] = serial ? stream() : ax.s[q];
    constexpr bool can_exact = is_fp && (!SR::is_static || SR::add_code == B_PLUS);
    bool launched_exact = false;
    if constexpr (can_exact) if (exact) {
      launched_exact = true;
      if (hc[0]) hipLaunchKernelGGL((k_spgemm_masked_lds<T, SR, 64, 64, 256, true>), dim3(nblocks(hc[0], 4)), dim3(256), 0, bs[0], a, L, hc[0], sr);
      if (hc[1]) hipLaunchKernelGGL((k_spgemm_masked_lds<T, SR, 512, 256, 256, true>), dim3(nblocks(hc[1], 1)), dim3(256), 0, bs[1], a, L + (size_t)nrows, hc[1], sr);
      if (hc[2]) hipLaunchKernelGGL((k_spgemm_masked_lds<T, SR, 2048, 512, 512, true>), dim3(nblocks(hc[2], 1)), dim3(512), 0, bs[2], a, L + (size_t)2 * nrows, hc[2], sr);
      if (hc[3]) hipLaunchKernelGGL((k_spgemm_masked_lds<T, SR, 8192, 1024, 1024, true>), dim3(nblocks(hc[3], 1)), dim3(1024), 0, bs[3], a, L + (size_t)3 * nrows, hc[3], sr);
      if (hc[4]) {
        hipLaunchKernelGGL((k_spgemm_masked_map<T, SR, false, true>), dim3(nb_map), dim3(1024), 0, stream(), a, L + (size_t)4 * nrows, hc[4], nslices, maps.as<uint32_t>(), B.ncols, sr);
        hipLaunchKernelGGL((k_exact_finish<T>), dim3((unsigned)std::min<uint32_t>(hc[4], 4096u)), dim3(256), 0, stream(), L + (size_t)4 * nrows, hc[4], M.rowptr.as<uint32_t>(), rowexp.as<int32_t>(),
                           xlo.as<unsigned long long>(), xhi.as<unsigned long long>(), cflag.as<uint8_t>(), cacc.as<W>());
      }
    }
    if (!launched_exact) {
    if (hc[0]) hipLaunchKernelGGL((k_spgemm_masked_lds<T, SR, 64, 64, 256>), dim3(nblocks(hc[0], 4)), dim3(256), 0, bs[0], a, L, hc[0], sr);
    if (hc[1]) hipLaunchKernelGGL((k_spgemm_masked_lds<T, SR, 512, 256, 256>), dim3(nblocks(hc[1], 1)), dim3(256), 0, bs[1], a, L + (size_t)nrows, hc[1], sr);
    if (hc[2]) hipLaunchKernelGGL((k_spgemm_masked_lds<T, SR, 2048, 512, 512>), dim3(nblocks(hc[2], 1)), dim3(512), 0, bs[2], a, L + (size_t)2 * nrows, hc[2], sr);
    if (hc[3]) hipLaunchKernelGGL((k_spgemm_masked_lds<T, SR, 8192, 1024, 1024>), dim3(nblocks(hc[3], 1)), dim3(1024), 0, bs[3], a, L + (size_t)3 * nrows, hc[3], sr);
    }
    if (hc[4] && !launched_exact) {
      // a product that only counts (PLUS over PAIR on an integer type): 32-bit LDS counters, 24 576 positions per mask row
      constexpr bool can_count = SR::is_static && std::is_integral<T>::value && sizeof(T) >= 4;
      bool counted = false;
      if constexpr (can_count) if (sr.add_op() == B_PLUS && sr.mul_op() == B_PAIR) {
        hipLaunchKernelGGL((k_spgemm_masked_map<T, SR, true>), dim3(nb_map), dim3(1024), 0, stream(), a, L + (size_t)4 * nrows, hc[4], nslices, maps.as<uint32_t>(), B.ncols, sr); counted = true; }
      if (!counted) hipLaunchKernelGGL((k_spgemm_masked_map<T, SR, false>), dim3(nb_map), dim3(1024), 0, stream(), a, L + (size_t)4 * nrows, hc[4], nslices, maps.as<uint32_t>(), B.ncols, sr);
    }
    ax.join(stream()); guard.joined = true;
    if (hc[4]) GRB_HIP(hipStreamSynchronize(stream()));       // the maps go back to the pool when this scope ends
    g_last_plan += std::string("k_spgemm_masked<") + (sr.is_static ? "static" : "dynamic") + "> bins " + std::to_string(hc[0]) + "/" + std::to_string(hc[1]) + "/" +
                   std::to_string(hc[2]) + "/" + std::to_string(hc[3]) + "/" + std::to_string(hc[4]) + (launched_exact ? " exact " : " ");
    if (launched_exact && hc[6]) {                       // rows with an Inf / NaN bound
      const uint32_t* ls[1] = {noexp_rows.as<uint32_t>()}; const uint32_t cs[1] = {hc[6]};
      run_ordered(ls, cs, 1, nullptr, 0u);
    }
  });
  GRB_HIP(hipGetLastError());
  // compaction (entry-parallel): position of every kept mask entry = exclusive scan of the flags
  {
    auto grid_n = [](uint64_t n) { uint64_t b = (n + 255) / 256; if (b < 1) b = 1; if (b > 8192) b = 8192; return (unsigned)b; };
    DevBuf f32(mnz * 4 + 4), pos(mnz * 4 + 4);
    hipLaunchKernelGGL(k_flags_to_u32, dim3(grid_n(mnz)), dim3(256), 0, stream(), cflag.as<uint8_t>(), mnz, f32.as<uint32_t>());
    exclusive_scan_u32(f32.as<uint32_t>(), pos.as<uint32_t>(), mnz);
    uint32_t lastp = 0; uint8_t lastf = 0;
    GRB_HIP(hipMemcpyAsync(&lastp, pos.as<uint32_t>() + (mnz - 1), 4, hipMemcpyDeviceToHost, stream()));
    GRB_HIP(hipMemcpyAsync(&lastf, cflag.as<uint8_t>() + (mnz - 1), 1, hipMemcpyDeviceToHost, stream()));
    GRB_HIP(hipStreamSynchronize(stream()));
    const uint32_t total = lastp + (lastf ? 1u : 0u);
    out.nnz = total; out.col.alloc((size_t)total * 4 + 4); out.val.alloc((size_t)total * sizeof(T) + 8);
    hipLaunchKernelGGL(k_rowptr_from_scan, dim3(grid_n(nrows + 1)), dim3(256), 0, stream(), nrows, M.rowptr.as<uint32_t>(), pos.as<uint32_t>(), total, mnz, out.rowptr.as<uint32_t>());
    hipLaunchKernelGGL((k_scatter_acc<T>), dim3(grid_n(mnz)), dim3(256), 0, stream(), mnz, M.col.as<uint32_t>(), cacc.as<W>(), cflag.as<uint8_t>(), pos.as<uint32_t>(),
                       out.col.as<uint32_t>(), out.val.as<T>());
  }
  GRB_HIP(hipGetLastError());
  GRB_HIP(hipStreamSynchronize(stream()));
  out.valid = true;
}

// ---- (2) expand / sort / compress ------------------------------------------------------------------------------------------
// products of every row, ub[r] = sum_{k in A(r,:)} nnz(B(k,:)), per ENTRY: the B-row lengths of the entries, an exclusive scan over
// them, and the difference at the row boundaries (a thread walking a 1e5-entry hub row alone was the whole kernel)
static __global__ void k_entry_products(uint64_t nnz, const uint32_t* __restrict__ acol, const uint32_t* __restrict__ brp, unsigned long long* __restrict__ c) {
  for (uint64_t p = blockIdx.x * 256ull + threadIdx.x; p <= nnz; p += gridDim.x * 256ull) { if (p < nnz) { const uint32_t k = acol[p]; c[p] = brp[k + 1] - brp[k]; } else c[p] = 0; }
}
static __global__ void k_row_products(uint32_t nrows, const uint32_t* __restrict__ arp, const unsigned long long* __restrict__ P, unsigned long long* __restrict__ ub) {
  for (uint64_t r = blockIdx.x * 256ull + threadIdx.x; r < nrows; r += (uint64_t)gridDim.x * 256ull) ub[r] = P[arp[r + 1]] - P[arp[r]];
}
static inline void row_upper_bound(const DevCSR& A, const DevCSR& B, unsigned long long* ub) {
  auto grid_n = [](uint64_t n) { uint64_t b = (n + 255) / 256; if (b < 1) b = 1; if (b > 8192) b = 8192; return (unsigned)b; };
  DevBuf c((A.nnz + 1) * 8 + 8), P((A.nnz + 1) * 8 + 8);
  hipLaunchKernelGGL(k_entry_products, dim3(grid_n(A.nnz + 1)), dim3(256), 0, stream(), A.nnz, A.col.as<uint32_t>(), B.rowptr.as<uint32_t>(), c.as<unsigned long long>());
  exclusive_scan_u64((const uint64_t*)c.p, (uint64_t*)P.p, A.nnz + 1);
  hipLaunchKernelGGL(k_row_products, dim3(grid_n(A.nrows)), dim3(256), 0, stream(), A.nrows, A.rowptr.as<uint32_t>(), P.as<unsigned long long>(), ub);
  GRB_HIP(hipStreamSynchronize(stream()));
}

// one wave per row of the chunk: products of row i are written at off[i - r0] in (k, then B-row) order
template <class T, class SR>
__global__ __launch_bounds__(256) void k_expand(uint32_t r0, uint32_t r1, const uint32_t* __restrict__ arp, const uint32_t* __restrict__ acol, const T* __restrict__ aval,
                                                const uint32_t* __restrict__ brp, const uint32_t* __restrict__ bcol, const T* __restrict__ bval,
                                                const unsigned long long* __restrict__ off, unsigned long long* __restrict__ keys, T* __restrict__ vals, const SR sr) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave = (blockIdx.x * 256ull + threadIdx.x) >> 6, nwaves = (uint64_t)gridDim.x * 4;
  const bool use_a = sr.uses_a(), use_b = sr.uses_u();
  for (uint64_t r = r0 + wave; r < r1; r += nwaves) {
    unsigned long long o = off[r - r0];
    for (uint32_t pa = arp[r]; pa < arp[r + 1]; pa++) {
      const uint32_t k = acol[pa]; const T av = use_a ? aval[pa] : T();
      const uint32_t bb = brp[k], be = brp[k + 1];
      for (uint32_t pb = bb + lane; pb < be; pb += 64) {
        keys[o + (pb - bb)] = ((unsigned long long)(r - r0) << 32) | bcol[pb];
        vals[o + (pb - bb)] = sr.mult(av, use_b ? bval[pb] : T());
      }
      o += be - bb;
    }
  }
}
static __global__ void k_iota32(uint32_t* p, uint64_t n) { for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = (uint32_t)i; }
static __global__ void k_heads(const unsigned long long* __restrict__ keys, uint64_t n, uint32_t* __restrict__ head) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
// each segment head reduces its run in sorted (== k) order and appends the entry; rows are counted with atomics
template <class T, class SR>
__global__ void k_compress(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ perm, const T* __restrict__ vals, uint64_t n,
                           const uint32_t* __restrict__ head, const uint32_t* __restrict__ pos, uint32_t out_base, uint32_t r0,
                           uint32_t* __restrict__ ocol, T* __restrict__ oval, uint32_t* __restrict__ rowcount, const SR sr) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
    if (!head[i]) continue;
    const unsigned long long key = keys[i];
    T acc = vals[perm[i]];
    for (uint64_t q = i + 1; q < n && keys[q] == key; q++) acc = sr.add(acc, vals[perm[q]]);
    const uint32_t w = out_base + pos[i];
    ocol[w] = (uint32_t)(key & 0xFFFFFFFFull); oval[w] = acc;
    atomicAdd(&rowcount[r0 + (uint32_t)(key >> 32)], 1u);
  }
}

template <class T> void run_spgemm_esc(const SpgemmCall& c, const SemiringDesc& d, DevCSR& out) {
  const DevCSR& A = *c.A; const DevCSR& B = *c.B;
  const uint32_t nrows = A.nrows;
  out.clear(); out.nrows = nrows; out.ncols = B.ncols;
  out.rowptr.alloc(((size_t)nrows + 1) * 4);
  DevBuf rowcount(((size_t)nrows + 1) * 4);
  GRB_HIP(hipMemsetAsync(rowcount.p, 0, ((size_t)nrows + 1) * 4, stream()));
  auto grid_n = [](uint64_t n) { uint64_t b = (n + 255) / 256; if (b < 1) b = 1; if (b > 8192) b = 8192; return (unsigned)b; };
  // per-row product counts on the host decide the chunking (bounded temporary memory)
  DevBuf ub((size_t)nrows * 8 + 8);
  row_upper_bound(A, B, ub.as<unsigned long long>());
  std::vector<unsigned long long> hub(nrows);
  if (nrows) GRB_HIP(hipMemcpyAsync(hub.data(), ub.p, (size_t)nrows * 8, hipMemcpyDeviceToHost, stream()));
  GRB_HIP(hipStreamSynchronize(stream()));
  const unsigned long long BUDGET = 1ull << 27;      // products per chunk (~3.5 GB of temporaries at 8-byte values)
  struct Piece { DevBuf col, val; uint32_t n; };
  std::vector<Piece> pieces; uint64_t total = 0;
  with_semiring<T>(d, [&](auto sr) {
    typedef decltype(sr) SR;
    uint32_t r0 = 0;
    while (r0 < nrows) {
      uint32_t r1 = r0; unsigned long long P = 0;
      std::vector<unsigned long long> off;
      while (r1 < nrows && (P == 0 || P + hub[r1] <= BUDGET)) { off.push_back(P); P += hub[r1]; r1++; }
      if (P > 0xFFFFFFF0ull) fail(GrB_OUT_OF_MEMORY, "mxm: one output row needs more than 2^32 products; not supported by the expand/sort/compress path");
      if (P) {
        DevBuf doff(off.size() * 8), keys(P * 8), keys2(P * 8), vals(P * sizeof(T)), perm0(P * 4), perm(P * 4), head(P * 4 + 4), pos(P * 4 + 4);
        GRB_HIP(hipMemcpyAsync(doff.p, off.data(), off.size() * 8, hipMemcpyHostToDevice, stream()));
        uint64_t nb = ((uint64_t)(r1 - r0) + 3) / 4; if (nb > 65535u * 4) nb = 65535u * 4; if (nb < 1) nb = 1;
        hipLaunchKernelGGL((k_expand<T, SR>), dim3((unsigned)nb), dim3(256), 0, stream(), r0, r1, A.rowptr.as<uint32_t>(), A.col.as<uint32_t>(), (const T*)c.aval,
                           B.rowptr.as<uint32_t>(), B.col.as<uint32_t>(), (const T*)c.bval, doff.as<unsigned long long>(), keys.as<unsigned long long>(), vals.as<T>(), sr);
        hipLaunchKernelGGL(k_iota32, dim3(grid_n(P)), dim3(256), 0, stream(), perm0.as<uint32_t>(), P);
        int rbits = 1; while ((1ull << rbits) < (unsigned long long)(r1 - r0)) rbits++;
        sort_pairs_u64((const uint64_t*)keys.p, (uint64_t*)keys2.p, perm0.as<uint32_t>(), perm.as<uint32_t>(), P, 32 + rbits);
        hipLaunchKernelGGL(k_heads, dim3(grid_n(P)), dim3(256), 0, stream(), keys2.as<unsigned long long>(), P, head.as<uint32_t>());
        exclusive_scan_u32(head.as<uint32_t>(), pos.as<uint32_t>(), P);
        uint32_t lastpos = 0, lasthead = 0;
        GRB_HIP(hipMemcpyAsync(&lastpos, pos.as<uint32_t>() + (P - 1), 4, hipMemcpyDeviceToHost, stream()));
        GRB_HIP(hipMemcpyAsync(&lasthead, head.as<uint32_t>() + (P - 1), 4, hipMemcpyDeviceToHost, stream()));
        GRB_HIP(hipStreamSynchronize(stream()));
        const uint32_t uniq = lastpos + lasthead;
        Piece pc; pc.n = uniq; pc.col.alloc((size_t)uniq * 4 + 4); pc.val.alloc((size_t)uniq * sizeof(T) + 8);
        hipLaunchKernelGGL((k_compress<T, SR>), dim3(grid_n(P)), dim3(256), 0, stream(), keys2.as<unsigned long long>(), perm.as<uint32_t>(), vals.as<T>(), P,
                           head.as<uint32_t>(), pos.as<uint32_t>(), 0u, r0, (uint32_t*)pc.col.p, (T*)pc.val.p, rowcount.as<uint32_t>(), sr);
        GRB_HIP(hipStreamSynchronize(stream()));
        total += uniq; pieces.push_back(std::move(pc));
      }
      r0 = r1;
    }
    g_last_plan += std::string("spgemm_esc<") + (sr.is_static ? "static" : "dynamic") + "> chunks " + std::to_string(pieces.size()) + " ";
  });
  GRB_HIP(hipGetLastError());
  if (total > 0xFFFFFFF0ull) fail(GrB_INSUFFICIENT_SPACE, "mxm: result has more than 2^32 entries");
  exclusive_scan_u32(rowcount.as<uint32_t>(), out.rowptr.as<uint32_t>(), (uint64_t)nrows + 1);
  out.nnz = total; out.col.alloc(total * 4 + 4); out.val.alloc(total * sizeof(T) + 8);
  uint64_t w = 0;
  for (auto& pc : pieces) {
    if (pc.n) {
      GRB_HIP(hipMemcpyAsync(out.col.as<uint32_t>() + w, pc.col.p, (size_t)pc.n * 4, hipMemcpyDeviceToDevice, stream()));
      GRB_HIP(hipMemcpyAsync(out.val.as<T>() + w, pc.val.p, (size_t)pc.n * sizeof(T), hipMemcpyDeviceToDevice, stream()));
    }
    w += pc.n;
  }
  GRB_HIP(hipStreamSynchronize(stream()));
  out.valid = true;
}

}  // namespace grb
