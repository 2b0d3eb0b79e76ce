// grb_spmv_kernels.hpp — the SpMV / SpMSpV kernels behind GrB_mxv and GrB_vxm  (HBM-bound; no MFMA).
//
//   t(i) = (+)_j  mult(M(i,j), u(j))      M in CSR (u32 rowptr/col, sorted rows), u and t bitmap
//
// Three hand-written kernels, all wave64:
//  A  k_spmv_adaptive   row-block ("CSR-adaptive" style) pull kernel — the FP64 PLUS_TIMES
//                       north-star path.  A block of 256 threads owns a run of consecutive rows
//                       holding <= 2048 entries: col/val are read fully coalesced (every lane a
//                       consecutive entry, 8 independent loads in flight per lane before the
//                       first use), u is gathered (n*8 B = 32 MiB at R-MAT-22: L2/Infinity-Cache
//                       resident), products are staged in LDS and each row is reduced from LDS
//                       in a fixed left-to-right order (deterministic).  Rows longer than a
//                       block are split into 8192-entry parts; the last part to arrive combines
//                       the partials in part order (agent-scope fence + ticket, guide §6 G16).
//  B  k_spmv_rowgroup   G lanes per row with mask skip and monoid-terminal early exit — the
//                       pull step of BFS (complemented visited mask, LOR terminal).
//  C  k_spmspv_push     frontier-driven scatter with atomics — the push step when u is sparse.
// Algorithmic bytes per call: nnz*(sizeof(T)+4) + (nrows+1)*4 + ncols*T + nrows*T.
#pragma once
#include "grb_api.hpp"
#include "grb_device.hpp"
#include "grb_semiring.hpp"
#include "grb_spmv.hpp"

namespace grb {

// Load flavours (experiment knobs GRB_MI355X_GATHER / GRB_MI355X_STREAM): 0 plain, 1 nontemporal, 2 sc1 (L1 bypass)
template <int MODE, class T> __device__ __forceinline__ T ld(const T* p) {
  if constexpr (MODE == 0) return *p;
  else {
    typedef typename std::conditional<sizeof(T) == 8, uint64_t, typename std::conditional<sizeof(T) == 4, uint32_t,
            typename std::conditional<sizeof(T) == 2, uint16_t, uint8_t>::type>::type>::type W;
    W w;
    if constexpr (MODE == 1) w = __builtin_nontemporal_load((const W*)p);
    else w = __hip_atomic_load((const W*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    T t; __builtin_memcpy(&t, &w, sizeof(T)); return t;
  }
}

}  // namespace grb
#include "grb_spmv_wavepipe.hpp"
#include "grb_spmv_xcd.hpp"
namespace grb {

template <class T> struct SpmvKArgs {
  const uint32_t* rowptr; const uint32_t* col; const T* aval;
  const T* uval; const uint8_t* upres; const uint8_t* allow;
  T* tval; uint8_t* tpres;
  const SpmvBlock* blocks; uint32_t* tickets; T* partial; uint8_t* pflag;
  uint32_t nrows;
  uint32_t* any_true; uint32_t any_true_tag;      // BOOL results only (nullptr otherwise): set to the tag when an entry with value true is written
  const uint8_t* fm_val; uint32_t fm_flags;       // SpmvCall::fm_val / fm_flags (row-lane kernel, FUSED instantiation)
  const uint8_t* fm_code;                         // SpmvCall::fm_code (k_spmv_rowlane_k<..., CODE = true>)
  const uint32_t* fe_rowptr; unsigned long long* fe_host;      // SpmvCall::fe_* (row-lane kernel)
  const uint4* heads; const uint32_t* nonempty;                // DevCSR::heads / nonempty (row-lane kernel, FUSED, K rows per lane)
};
// a workgroup's share of the result's summary (SpmvCall::fe_host): edge sum and entry count of the true entries it wrote, stored — each in the
// low half of a 64-bit word whose high half is the product's tag — into ITS OWN pair of page-locked HOST words.  The `q.reduce_bool()` that
// follows spins until every pair carries the tag and adds them up: no device-to-host copy, no stream synchronisation (17 us on this box, six
// times per BFS), and nothing for the workgroups to agree on.  What was tried on the device side first (R-MAT-22, the late levels' 13 us pull):
// a pair of atomics per WAVE on packed pairs — 131 072 atomics on eight cache lines — 261 us; a pair per workgroup on 32 padded pairs + one
// ticket for the last workgroup to report: +28 us for 1024 workgroups; tickets in two levels: +25 us — agent-scope atomics are performed at
// the memory side, and those on one LINE complete one after the other, ~100 ns each.
// One word per workgroup: tag (24 bits) | true entries, saturating at 255 (8 bits: only "any" is asked) | edge sum, saturating (32 bits: a lower bound is
// all the direction choice asks for).  (Two words per workgroup from 1024 workgroups of 256 threads made the pull 7 us longer: 2048 small writes across
// PCIe at the end of a 13 us kernel.  Hence one word, and workgroups of 1024 threads.)
__device__ __forceinline__ void fe_block_report(unsigned long long* __restrict__ host_out, uint32_t slot, uint32_t tag, unsigned long long fe, unsigned long long cnt) {
  __shared__ unsigned long long s_fe[2][16];
  cnt = wave_reduce_add_u64(cnt); fe = wave_reduce_add_u64(fe);
  const int wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) { s_fe[0][wv & 15] = fe; s_fe[1][wv & 15] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    fe = 0; cnt = 0; for (int q = 0; q < nw; q++) { fe += s_fe[0][q]; cnt += s_fe[1][q]; }
    if (fe > 0xFFFFFFFFull) fe = 0xFFFFFFFFull;
    if (cnt > 0xFFull) cnt = 0xFFull;
    __hip_atomic_store(host_out + slot, ((unsigned long long)(tag & 0xFFFFFFu) << 40) | (cnt << 32) | fe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <class T> __device__ __forceinline__ bool spmv_truthy(T v) { if constexpr (is_bool<T>::value) return v.v != 0; else return v != T(); }

// ---- kernel A ---------------------------------------------------------------------------------------------------
template <class T, class SR, bool U_FULL, bool HAS_ALLOW, int GM = 0, int SM = 0>
__global__ __launch_bounds__(SPMV_THREADS) void k_spmv_adaptive(const SpmvKArgs<T> a, const SR sr) {
  __shared__ T s_prod[SPMV_NNZ];
  __shared__ uint8_t s_has[U_FULL ? 4 : SPMV_NNZ];
  __shared__ T s_wave[SPMV_THREADS / 64];
  __shared__ uint8_t s_whas[SPMV_THREADS / 64];
  const int tid = threadIdx.x;
  const SpmvBlock b = a.blocks[blockIdx.x];
  const bool use_a = sr.uses_a(), use_u = sr.uses_u();

  if (b.nparts == 0) {
    const uint32_t r0 = b.row, r1 = b.aux;
    if constexpr (HAS_ALLOW) {
      int any = 0;
      for (uint32_t r = r0 + tid; r < r1; r += SPMV_THREADS) any |= a.allow[r];
      if (!__syncthreads_or(any)) {   // the whole block is masked out: no matrix traffic at all
        for (uint32_t r = r0 + tid; r < r1; r += SPMV_THREADS) a.tpres[r] = 0;
        return;
      }
    }
    const uint32_t p0 = a.rowptr[r0], p1 = a.rowptr[r1];
    const uint32_t cnt = p1 - p0;
    // phase 1: coalesced stream of (col, val), gather u, products to LDS.  Every load is
    // unconditional (tail lanes re-read the block's last entry) so the compiler keeps them
    // branch-free: 16 streaming loads, then 8 gathers, are in flight per lane before the first use.
    if (cnt) {
      uint32_t c[SPMV_UNROLL]; T av[SPMV_UNROLL]; T uv[SPMV_UNROLL]; uint8_t up[SPMV_UNROLL];
#pragma unroll
      for (int u = 0; u < SPMV_UNROLL; u++) {
        const uint32_t k = tid + u * SPMV_THREADS;
        const uint32_t p = p0 + (k < cnt ? k : cnt - 1);
        c[u] = ld<SM>(&a.col[p]);
        av[u] = use_a ? ld<SM>(&a.aval[p]) : T();
      }
#pragma unroll
      for (int u = 0; u < SPMV_UNROLL; u++) {
        uv[u] = use_u ? ld<GM>(&a.uval[c[u]]) : T();
        if constexpr (!U_FULL) up[u] = ld<GM>(&a.upres[c[u]]); else up[u] = 1;
      }
#pragma unroll
      for (int u = 0; u < SPMV_UNROLL; u++) {
        const uint32_t k = tid + u * SPMV_THREADS;
        if (k < cnt) {
          s_prod[k] = sr.mult(av[u], uv[u]);
          if constexpr (!U_FULL) s_has[k] = up[u];
        }
      }
    }
    __syncthreads();
    // phase 2: G lanes per row, G = the largest power of two with rows*G <= 256 (block-uniform).
    // Each lane sums a G-strided slice of the row from LDS (conflict-free), then a fixed
    // shuffle tree combines the G partials: the order is a function of the row block only,
    // so floating-point results are reproducible run to run.
    const uint32_t nr = r1 - r0;
    const uint32_t G = nr >= SPMV_THREADS / 2 ? 1u : (nr <= 4 ? 64u : (1u << (31 - __builtin_clz(SPMV_THREADS / nr))));
    const uint32_t lane = tid & (G - 1), grp = tid / G, ngrp = SPMV_THREADS / G;
    for (uint32_t rb = 0; rb < nr; rb += ngrp) {
      const uint32_t r = r0 + rb + grp;
      const bool live = rb + grp < nr;
      bool ok = live;
      if constexpr (HAS_ALLOW) { if (live) ok = a.allow[r] != 0; }
      uint32_t qb = 0, qe = 0;
      if (ok) { qb = a.rowptr[r] - p0; qe = a.rowptr[r + 1] - p0; }
      T acc = sr.identity; bool has = false;
      for (uint32_t q = qb + lane; q < qe; q += G) {
        if constexpr (U_FULL) { acc = has ? sr.add(acc, s_prod[q]) : s_prod[q]; has = true; }
        else if (s_has[q]) { acc = has ? sr.add(acc, s_prod[q]) : s_prod[q]; has = true; }
      }
      if (G > 1) {
        // lanes without a contribution carry the identity; `has` travels with the value
        for (uint32_t d = G >> 1; d >= 1; d >>= 1) {
          const T ov = shfl_down_t<T>(acc, (int)d);
          const int oh = __shfl_down((int)has, (int)d, 64);
          if (oh) { acc = has ? sr.add(acc, ov) : ov; has = true; }
        }
      }
      if (live && lane == 0) {
        if (has) a.tval[r] = acc;
        a.tpres[r] = has ? 1 : 0;
      }
    }
    return;
  }

  // ---- long row part ----
  const uint32_t row = b.row;
  if constexpr (HAS_ALLOW) { if (!a.allow[row]) { if (b.aux == 0 && tid == 0) a.tpres[row] = 0; return; } }
  const uint32_t rb = a.rowptr[row], re = a.rowptr[row + 1];
  const uint32_t pb = rb + b.aux * SPMV_LONG_CHUNK;
  const uint32_t pe = (re - pb > (uint32_t)SPMV_LONG_CHUNK) ? pb + SPMV_LONG_CHUNK : re;
  T acc = sr.identity; bool has = false;
  for (uint32_t base = pb; base < pe; base += SPMV_NNZ) {
    uint32_t c[SPMV_UNROLL]; T av[SPMV_UNROLL]; T uv[SPMV_UNROLL]; uint8_t up[SPMV_UNROLL];
#pragma unroll
    for (int u = 0; u < SPMV_UNROLL; u++) {
      uint32_t p = base + tid + u * SPMV_THREADS;
      p = p < pe ? p : pe - 1;                       // unconditional loads: tail lanes re-read the last entry
      c[u] = ld<SM>(&a.col[p]);
      av[u] = use_a ? ld<SM>(&a.aval[p]) : T();
    }
#pragma unroll
    for (int u = 0; u < SPMV_UNROLL; u++) {
      uv[u] = use_u ? ld<GM>(&a.uval[c[u]]) : T();
      if constexpr (!U_FULL) up[u] = ld<GM>(&a.upres[c[u]]); else up[u] = 1;
    }
#pragma unroll
    for (int u = 0; u < SPMV_UNROLL; u++) {
      const uint32_t p = base + tid + u * SPMV_THREADS;
      if (p < pe && up[u]) {
        const T m = sr.mult(av[u], uv[u]);
        acc = has ? sr.add(acc, m) : m; has = true;
      }
    }
  }
  // block reduction in a fixed tree: lanes (xor butterfly) -> 4 waves in order
  const T accv = has ? acc : sr.identity;
  const T wsum = wave_reduce_op<T, false>(sr.add_op(), accv);
  const bool whas = __any(has);
  if ((tid & 63) == 0) { s_wave[tid >> 6] = wsum; s_whas[tid >> 6] = whas; }
  __syncthreads();
  if (tid == 0) {
    T r = sr.identity; bool h = false;
    for (int w = 0; w < SPMV_THREADS / 64; w++) if (s_whas[w]) { r = h ? sr.add(r, s_wave[w]) : s_wave[w]; h = true; }
    if (b.nparts == 1) {
      if (h) a.tval[row] = r;
      a.tpres[row] = h ? 1 : 0;
    } else {
      a.partial[b.slot + b.aux] = r; a.pflag[b.slot + b.aux] = h ? 1 : 0;
      __threadfence();                                         // agent-scope release of the partial
      const uint32_t ticket = atomicAdd(&a.tickets[b.slot], 1u);
      if (ticket == b.nparts - 1) {                            // last part to arrive combines, in part order
        __threadfence();                                       // agent-scope acquire
        T tot = sr.identity; bool th = false;
        for (uint32_t p = 0; p < b.nparts; p++) {
          const T pv = __hip_atomic_load(&a.partial[b.slot + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint8_t pf = __hip_atomic_load(&a.pflag[b.slot + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (pf) { tot = th ? sr.add(tot, pv) : pv; th = true; }
        }
        if (th) a.tval[row] = tot;
        a.tpres[row] = th ? 1 : 0;
        a.tickets[b.slot] = 0;                                  // re-arm for the next launch
      }
    }
  }
}

// ---- kernel B: G lanes per row, mask skip, terminal early exit -----------------------------------------------------
template <class T, class SR, int G, bool U_FULL>
__global__ __launch_bounds__(256) void k_spmv_rowgroup(const SpmvKArgs<T> a, const SR sr) {
  const int lane = threadIdx.x & (G - 1);
  const uint64_t group = (blockIdx.x * 256ull + threadIdx.x) / G;
  const uint64_t ngroups = (uint64_t)gridDim.x * 256ull / G;
  const bool use_a = sr.uses_a(), use_u = sr.uses_u();
  for (uint64_t r = group; r < a.nrows; r += ngroups) {
    if (a.allow && !a.allow[r]) { if (lane == 0) a.tpres[r] = 0; continue; }
    const uint32_t pb = a.rowptr[r], pe = a.rowptr[r + 1];
    T acc = sr.identity; bool has = false;
    for (uint32_t p = pb + lane; p < pe; p += G) {
      const uint32_t c = a.col[p];
      bool pr = true;
      if constexpr (!U_FULL) pr = a.upres[c] != 0;
      if (pr) {
        const T m = sr.mult(use_a ? a.aval[p] : T(), use_u ? a.uval[c] : T());
        acc = has ? sr.add(acc, m) : m; has = true;
        if (sr.has_terminal && memcmp_eq(acc, sr.terminal)) break;   // this lane cannot change the result any more
      }
    }
    // reduce the G lanes (identity where a lane saw nothing)
    T v = has ? acc : sr.identity;
    unsigned long long hb = __ballot(has);
    const int gbase = (threadIdx.x & 63) & ~(G - 1);
    const unsigned long long gb = G == 64 ? hb : ((hb >> gbase) & ((1ull << (G & 63)) - 1));
    if (sr.add_op() == B_ANY) { v = shfl_t<T>(v, gbase + (gb ? __builtin_ctzll(gb) : 0)); hb = gb; }        // ANY: the value of a lane that has one, not the identity of one that has none
    else if constexpr (G == 64) { v = wave_reduce_op<T, false>(sr.add_op(), v); }
    else { v = group_reduce_op<T, G, false>(sr.add_op(), v); hb = gb; }
    if (lane == 0) { const bool h = hb != 0; if (h) a.tval[r] = v; a.tpres[r] = h ? 1 : 0; }
  }
}

// ---- kernel B': one LANE per row for the first entries, the wave for what is left ----------------------------------------------
// The masked pull of a BFS level asks, for every unvisited vertex, whether ONE of its neighbours is visited: most rows are
// empty, masked out, or decided by their first few entries.  With 8 lanes per row a wave has 8 rows in flight and every row is
// a chain of four dependent loads (mask byte, row pointers, column, operand byte): 132 us for the 4 M rows of R-MAT-22, 2 TB/s
// on paper and latency in fact.  Here a lane owns a row — 64 chains in flight per wave — and walks its first SPMV_LANE_E
// entries; the rows that are neither finished nor at their monoid's terminal value by then (hub rows) are completed by the
// whole wave, 64 entries per step, starting from the lane's partial result.
constexpr uint32_t SPMV_LANE_E = 8;
template <class T, class SR, bool U_FULL, bool FUSED = false>
__global__ __launch_bounds__(1024) void k_spmv_rowlane(const SpmvKArgs<T> a, const SR sr) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6);
  const bool use_a = sr.uses_a(), use_u = sr.uses_u();
  const uint64_t nround = ((uint64_t)a.nrows + 63) / 64 * 64;
  unsigned long long fe = 0, fcnt = 0;
  for (uint64_t base = wave * 64; base < nround; base += nwaves * 64) {
    const uint64_t r = base + lane;
    const bool valid = r < a.nrows;
    bool allowed;
    if constexpr (FUSED) allowed = valid && ((a.upres[r] != 0 && ((a.fm_flags & 1u) || a.fm_val[r] != 0)) != ((a.fm_flags & 2u) != 0));     // the mask vector itself
    else allowed = valid && (!a.allow || a.allow[r]);
    // operand value at column c: the vector's own byte as BOOL when fused
    auto uat = [&](uint32_t c) __attribute__((always_inline)) -> T { if constexpr (FUSED) { T t; t = T(a.fm_val[c] != 0); return t; } else return a.uval[c]; };
    // (the row pointers are fetched whether or not the row is allowed: one dependent round trip less per wave — the late levels of a BFS,
    //  where half of the 4 M rows are empty and unvisited, are a chain of such trips and little else)
    uint32_t pb = 0, pe = 0;
    if (FUSED ? valid : allowed) { pb = a.rowptr[r]; pe = a.rowptr[r + 1]; }
    T acc = sr.identity; bool has = false, done = !allowed;
    if (allowed) {
      const uint32_t e = pe - pb > SPMV_LANE_E ? pb + SPMV_LANE_E : pe;
      for (uint32_t p = pb; p < e; p++) {
        const uint32_t c = a.col[p];
        bool pr = true;
        if constexpr (!U_FULL) pr = a.upres[c] != 0;
        if (pr) {
          const T m = sr.mult(use_a ? a.aval[p] : T(), use_u ? uat(c) : T());
          acc = has ? sr.add(acc, m) : m; has = true;
          if (sr.has_terminal && memcmp_eq(acc, sr.terminal)) { done = true; break; }
        }
      }
      if (pe - pb <= SPMV_LANE_E) done = true;
    }
    // the unfinished rows, one after the other, by the whole wave
    unsigned long long todo = __ballot(allowed && !done);
    while (todo) {
      const int L = __builtin_ctzll(todo); todo &= todo - 1;
      const uint32_t qb = (uint32_t)__shfl((int)pb, L, 64) + SPMV_LANE_E, qe = (uint32_t)__shfl((int)pe, L, 64);
      T part = sr.identity; bool phas = false;
      for (uint32_t p0 = qb; p0 < qe; p0 += 64) {
        const uint32_t p = p0 + lane;
        if (p < qe) {
          const uint32_t c = a.col[p];
          bool pr = true;
          if constexpr (!U_FULL) pr = a.upres[c] != 0;
          if (pr) { const T m = sr.mult(use_a ? a.aval[p] : T(), use_u ? uat(c) : T()); part = phas ? sr.add(part, m) : m; phas = true; }
        }
        if (sr.has_terminal && __ballot(phas && memcmp_eq(part, sr.terminal))) break;      // some lane is at the terminal value: so is the row
      }
      const unsigned long long hb = __ballot(phas);
      // (ANY keeps "a" value: the tree below would also consider the identity of the lanes that saw nothing — take a real one)
      const T red = sr.add_op() == B_ANY ? shfl_t<T>(part, hb ? __builtin_ctzll(hb) : 0) : wave_reduce_op<T, false>(sr.add_op(), phas ? part : sr.identity);
      if (lane == L && hb) { acc = has ? sr.add(acc, red) : red; has = true; }
    }
    if (valid) { if (allowed && has) a.tval[r] = acc; a.tpres[r] = (allowed && has) ? 1 : 0; }
    const bool tr = valid && allowed && has && spmv_truthy<T>(acc);
    if (a.fe_host) { if (tr) { fe += a.fe_rowptr[r + 1] - a.fe_rowptr[r]; fcnt++; } }      // (the summary's count answers "any true" as well)
    else if (a.any_true) { if (__ballot(tr) && lane == 0) *a.any_true = a.any_true_tag; }      // (same value from every wave: a benign race)
  }
  if (a.fe_host) fe_block_report(a.fe_host, blockIdx.x, a.any_true_tag, fe, fcnt);      // (argument-uniform branch: every thread of the workgroup gets here)
}


constexpr int SPMV_LANE_K = 4;
constexpr uint32_t FE_MAX_BLOCKS = 2048;      // pairs of host words of the result summary (grb_container.cpp allocates them)
// Round 5: a lane owns SPMV_LANE_K rows at once (rows r, r + 64, r + 128, r + 192 of the wave's 256).  A row is a chain of four dependent
// loads — mask byte / row pointers, column, operand byte — and with one row per lane a wave had 64 chains in flight and nothing to do
// while they were: the level-2 pull of the R-MAT-22 BFS took 54 us for 64 MB, the late levels 17 us each for a handful of vertices
// (65 536 waves of one round trip after the other).  With K rows the K loads of every stage are issued together: all of them are
// unconditional (an idle slot reads entry 0 of its array and drops it), so they stand in one basic block.
// CODE (round 6, FUSED only): the operand / mask vector comes with code bytes (bit 0 present, bit 1 present and not zero): a neighbour costs ONE byte gather
// instead of a presence byte and a value byte — the pull is bound by its gathers (~3 CU-clocks each), not by bytes.
template <class T, class SR, bool U_FULL, bool FUSED, int K, bool CODE = false>
__global__ __launch_bounds__(1024) void k_spmv_rowlane_k(const SpmvKArgs<T> a, const SR sr) {
  static_assert(!CODE || FUSED, "code bytes belong to the fused (mask = operand) instantiation");
  const int lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (uint64_t)gridDim.x * (blockDim.x >> 6);
  const bool use_a = sr.uses_a(), use_u = sr.uses_u();
  const uint64_t span = 64ull * K, nround = ((uint64_t)a.nrows + span - 1) / span * span;
  unsigned long long fe = 0, fcnt = 0;
  for (uint64_t base = wave * span; base < nround; base += nwaves * span) {
    uint64_t r[K]; bool valid[K], allowed[K], has[K], done[K]; uint32_t pb[K], pe[K]; T acc[K];
    uint8_t m0[K], m1[K];
    uint32_t e0 = 0;                                                                               // first entry the generic loop below looks at
    bool use_heads = false;
    if constexpr (FUSED) use_heads = a.heads != nullptr;                                           // (argument-uniform)
#pragma unroll
    for (int k = 0; k < K; k++) {
      r[k] = base + 64ull * k + lane; valid[k] = r[k] < a.nrows;
      const uint64_t rr = valid[k] ? r[k] : 0;
      if constexpr (CODE) { const uint8_t q = a.fm_code[rr]; m0[k] = q & 1u; m1[k] = q >> 1; }
      else if constexpr (FUSED) { m0[k] = a.upres[rr]; m1[k] = a.fm_val[rr]; }                  // the mask vector itself
      else { m0[k] = a.allow ? a.allow[rr] : (uint8_t)1; m1[k] = 1; }
      // (without row heads the row pointers are fetched whether or not the row is allowed: one dependent round trip less — the late levels of a
      //  BFS, where half of the 4 M rows are empty and unvisited, are a chain of such trips and little else)
      if (!use_heads) { pb[k] = a.rowptr[rr]; pe[k] = a.rowptr[rr + 1]; } else { pb[k] = 0; pe[k] = 0; }
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
      if constexpr (FUSED) allowed[k] = valid[k] && ((m0[k] != 0 && ((a.fm_flags & 1u) || m1[k] != 0)) != ((a.fm_flags & 2u) != 0));
      else allowed[k] = valid[k] && m0[k] != 0;
      acc[k] = sr.identity; has[k] = false; done[k] = !allowed[k] || (!use_heads && pe[k] == pb[k]);
    }
    if constexpr (FUSED) if (use_heads) {
      // Row heads (DevCSR::heads, round 5): one bit says whether an allowed row has an entry at all — half of R-MAT-22's rows have none, and they stay
      // "unvisited" for the whole search — and one 16-byte word carries the row's first four columns.  The level-2 pull of the BFS, which touched a
      // 128-byte line of the column array for the first entries of each of 2 M rows (65 us: bandwidth), reads 16 bytes per live row; a late level
      // reads the vector, the bits, and the heads of the handful of rows that are still live.
      uint32_t nw[K];
#pragma unroll
      for (int k = 0; k < K; k++) nw[k] = a.nonempty[(valid[k] ? r[k] : 0) >> 5];
      uint4 hd[K];
#pragma unroll
      for (int k = 0; k < K; k++) {
        const bool live = allowed[k] && ((nw[k] >> (r[k] & 31)) & 1u);
        done[k] = !live;
        hd[k] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        if (live) hd[k] = a.heads[r[k]];
      }
#pragma unroll
      for (int e = 0; e < 4; e++) {
        uint32_t c[K]; bool act[K], hv[K]; bool any = false;
#pragma unroll
        for (int k = 0; k < K; k++) {
          const uint32_t raw = e == 0 ? hd[k].x : e == 1 ? hd[k].y : e == 2 ? hd[k].z : hd[k].w;
          act[k] = !done[k] && raw != 0xFFFFFFFFu;
          if (!done[k] && raw == 0xFFFFFFFFu) done[k] = true;                                      // fewer than e + 1 entries: the row is exhausted
          c[k] = act[k] ? (raw & 0x3FFFFFFFu) : 0u;
          hv[k] = (raw >> 31) != 0;
          any = any || act[k];
        }
        if (!__ballot(any)) break;                                                                 // (a late level: most waves have no live row at all)
        uint8_t pr[K], vb[K];
#pragma unroll
        for (int k = 0; k < K; k++) {                                                                // (an idle slot reads position 0 and drops it)
          if constexpr (CODE) { const uint8_t q = a.fm_code[c[k]]; pr[k] = q & 1u; vb[k] = q >> 1; } else { pr[k] = a.upres[c[k]]; vb[k] = a.fm_val[c[k]]; }
        }
#pragma unroll
        for (int k = 0; k < K; k++) {
          if (act[k] && pr[k]) {
            T uvv; uvv = T(vb[k] != 0);
            T avv; avv = T(hv[k]);                                                                    // (the entry's BOOL value rides in bit 31 of its head word)
            const T m = sr.mult(use_a ? avv : T(), uvv);
            acc[k] = has[k] ? sr.add(acc[k], m) : m; has[k] = true;
            if (sr.has_terminal && memcmp_eq(acc[k], sr.terminal)) done[k] = true;
          }
        }
      }
      // rows with more than four entries that are still undecided go on in the column array (their row pointers are fetched now)
#pragma unroll
      for (int k = 0; k < K; k++) {
        const bool more = !done[k] && hd[k].w != 0xFFFFFFFFu && ((hd[k].w >> 30) & 1u);
        if (!more) done[k] = true;
        if (more) { pb[k] = a.rowptr[r[k]]; pe[k] = a.rowptr[r[k] + 1]; }
      }
      e0 = 4;
    }
    // the first SPMV_LANE_E entries of the K rows, entry by entry
    for (uint32_t e = e0; e < SPMV_LANE_E; e++) {
      bool act[K]; bool any = false;
#pragma unroll
      for (int k = 0; k < K; k++) { act[k] = !done[k] && pb[k] + e < pe[k]; any = any || act[k]; }
      if (!__ballot(any)) break;
      uint32_t c[K];
#pragma unroll
      for (int k = 0; k < K; k++) c[k] = a.col[act[k] ? pb[k] + e : 0u];                         // (some lane is active: the matrix has an entry 0)
      uint8_t pr[K]; T uv[K], av[K];
#pragma unroll
      for (int k = 0; k < K; k++) {
        if constexpr (CODE) { const uint8_t q = a.fm_code[c[k]]; pr[k] = q & 1u; T t; t = T((q >> 1) != 0); uv[k] = t; }
        else {
          if constexpr (U_FULL) pr[k] = 1; else pr[k] = a.upres[c[k]];
          if constexpr (FUSED) { T t; t = T(a.fm_val[c[k]] != 0); uv[k] = t; }                     // the vector's own byte as BOOL
          else uv[k] = use_u ? a.uval[c[k]] : T();
        }
        av[k] = use_a ? a.aval[act[k] ? pb[k] + e : 0u] : T();
      }
#pragma unroll
      for (int k = 0; k < K; k++) {
        if (act[k] && pr[k]) {
          const T m = sr.mult(av[k], uv[k]);
          acc[k] = has[k] ? sr.add(acc[k], m) : m; has[k] = true;
          if (sr.has_terminal && memcmp_eq(acc[k], sr.terminal)) done[k] = true;
        }
        if (pb[k] + e + 1 >= pe[k]) done[k] = true;                                                // the row is exhausted
      }
    }
    // the unfinished rows (hub rows), one after the other, by the whole wave
#pragma unroll
    for (int k = 0; k < K; k++) {
      unsigned long long todo = __ballot(!done[k]);
      while (todo) {
        const int L = __builtin_ctzll(todo); todo &= todo - 1;
        const uint32_t qb = (uint32_t)__shfl((int)pb[k], L, 64) + SPMV_LANE_E, qe = (uint32_t)__shfl((int)pe[k], L, 64);
        T part = sr.identity; bool phas = false;
        for (uint32_t p0 = qb; p0 < qe; p0 += 64) {
          const uint32_t p = p0 + lane;
          if (p < qe) {
            const uint32_t c = a.col[p];
            bool pr = true; uint8_t q = 0;
            if constexpr (CODE) { q = a.fm_code[c]; pr = (q & 1u) != 0; }
            else if constexpr (!U_FULL) pr = a.upres[c] != 0;
            if (pr) {
              T u1; if constexpr (CODE) { u1 = T((q >> 1) != 0); } else if constexpr (FUSED) { u1 = T(a.fm_val[c] != 0); } else { u1 = use_u ? a.uval[c] : T(); }
              const T m = sr.mult(use_a ? a.aval[p] : T(), u1); part = phas ? sr.add(part, m) : m; phas = true;
            }
          }
          if (sr.has_terminal && __ballot(phas && memcmp_eq(part, sr.terminal))) break;      // some lane is at the terminal value: so is the row
        }
        const unsigned long long hb = __ballot(phas);
        // (ANY keeps "a" value: the tree below would also consider the identity of the lanes that saw nothing — take a real one)
        const T red = sr.add_op() == B_ANY ? shfl_t<T>(part, hb ? __builtin_ctzll(hb) : 0) : wave_reduce_op<T, false>(sr.add_op(), phas ? part : sr.identity);
        if (lane == L && hb) { acc[k] = has[k] ? sr.add(acc[k], red) : red; has[k] = true; }
      }
    }
    bool wrote_true = false;
#pragma unroll
    for (int k = 0; k < K; k++) {
      const bool w = allowed[k] && has[k];
      if (valid[k]) { if (w) a.tval[r[k]] = acc[k]; a.tpres[r[k]] = w ? 1 : 0; }
      const bool tr = w && spmv_truthy<T>(acc[k]);
      wrote_true = wrote_true || tr;
      if (a.fe_host && tr) { fe += a.fe_rowptr[r[k] + 1] - a.fe_rowptr[r[k]]; fcnt++; }
    }
    if (!a.fe_host && a.any_true) { if (__ballot(wrote_true) && lane == 0) *a.any_true = a.any_true_tag; }      // (same value from every wave: a benign race)
  }
  if (a.fe_host) fe_block_report(a.fe_host, blockIdx.x, a.any_true_tag, fe, fcnt);
}

// ---- kernel C: push.  t is pre-initialised (tpres = 0); entries are claimed with tpres CAS-free flags ---------------
// Scatter with one atomic combine per product.  Used only for semirings whose monoid has a
// native atomic (PLUS on 32/64-bit ints and floats, MIN/MAX on ints, LOR/ANY): chosen by the driver.
template <class T> __device__ __forceinline__ void atomic_combine(int op, T* addr, T v) {
  if constexpr (is_bool<T>::value) {
    // LOR / ANY / PLUS / MAX on BOOL: any contribution sets the byte; benign same-value race
    if (v) addr->v = 1;
  } else if constexpr (std::is_same<T, float>::value || std::is_same<T, double>::value) {
    if (op == B_PLUS) atomicAdd(addr, v);
    else {  // MIN / MAX through integer CAS
      typedef typename std::conditional<sizeof(T) == 8, unsigned long long, unsigned int>::type U;
      U* ua = (U*)addr; U old = *ua, assumed;
      do {
        assumed = old; T cur; memcpy(&cur, &assumed, sizeof(T));
        T nv = apply_binop<T, false>(op, cur, v); U nu; memcpy(&nu, &nv, sizeof(T));
        if (nu == assumed) break;
        old = atomicCAS(ua, assumed, nu);
      } while (old != assumed);
    }
  } else if constexpr (sizeof(T) == 4) {
    typedef typename std::conditional<std::is_signed<T>::value, int, unsigned int>::type A;
    if (op == B_PLUS) atomicAdd((A*)addr, (A)v); else if (op == B_MIN) atomicMin((A*)addr, (A)v);
    else if (op == B_MAX) atomicMax((A*)addr, (A)v); else *addr = v;
  } else if constexpr (sizeof(T) == 8) {
    if (op == B_PLUS) atomicAdd((unsigned long long*)addr, (unsigned long long)v);
    else if (op == B_MIN) { if constexpr (std::is_signed<T>::value) atomicMin((long long*)addr, (long long)v); else atomicMin((unsigned long long*)addr, (unsigned long long)v); }
    else if (op == B_MAX) { if constexpr (std::is_signed<T>::value) atomicMax((long long*)addr, (long long)v); else atomicMax((unsigned long long*)addr, (unsigned long long)v); }
    else *addr = v;
  }
}

constexpr int PUSH_LONG = 4096;
struct SmallList64 { uint32_t n; uint32_t idx[64]; };      // a short list handed over as a kernel argument
__device__ __forceinline__ bool small_list_has(const SmallList64& l, uint32_t j) { bool x = false; for (uint32_t q = 0; q < l.n; q++) x = x || l.idx[q] == j; return x; }
template <class T, class SR>
__global__ __launch_bounds__(256) void k_spmspv_push(const uint32_t* __restrict__ fidx, uint32_t nf, const uint32_t* __restrict__ rowptr,
                                                     const uint32_t* __restrict__ col, const T* __restrict__ aval, const T* __restrict__ uval,
                                                     const uint8_t* __restrict__ allow, T* __restrict__ tval, uint8_t* __restrict__ tpres,
                                                     uint32_t* __restrict__ longlist, const SR sr, uint32_t* __restrict__ any_true = nullptr, const uint32_t any_tag = 0,
                                                     unsigned long long* __restrict__ fe_host = nullptr, const SmallList64 excl = SmallList64{0, {0}}) {
  // one wave per frontier entry: its row of M^T (= CSR row of the stored matrix) is streamed coalesced
  const int lane = threadIdx.x & 63;
  const uint64_t wave = (blockIdx.x * 256ull + threadIdx.x) >> 6;
  const uint64_t nwaves = (uint64_t)gridDim.x * 4;
  const bool use_a = sr.uses_a(), use_u = sr.uses_u();
  unsigned long long fe = 0, fcnt = 0;                                       // (fe_host: nf == 1 — SpmvCall::fe_host)
  for (uint64_t f = wave; f < nf; f += nwaves) {
    const uint32_t i = fidx[f];
    T ui = T(); if (use_u) { if (uval) ui = uval[i]; else ui = T(true); }          // (uval == nullptr: every operand value is true — SpmvCall::excl_small)
    const uint32_t pb = rowptr[i], pe = rowptr[i + 1];
    if (pe - pb > (uint32_t)PUSH_LONG) {          // hub row: handed to the all-blocks kernel below (one wave would take ms)
      if (lane == 0) longlist[1 + atomicAdd(&longlist[0], 1u)] = i;
      continue;
    }
    for (uint32_t p = pb + lane; p < pe; p += 64) {
      const uint32_t j = col[p];
      if (allow && !allow[j]) continue;
      if (excl.n && small_list_has(excl, j)) continue;
      const T m = sr.mult(use_a ? aval[p] : T(), ui);
      atomic_combine<T>(sr.add_op(), &tval[j], m);
      tpres[j] = 1;
      if (any_true && spmv_truthy<T>(m)) *any_true = any_tag;             // BOOL monoids of the push path only OR values in: a true product is a true entry
      if (fe_host && spmv_truthy<T>(m)) { fe += rowptr[j + 1] - rowptr[j]; fcnt++; }
    }
  }
  if (fe_host) fe_block_report(fe_host, blockIdx.x, any_tag, fe, fcnt);                // (one operand entry: one workgroup)
}

// frontier rows longer than PUSH_LONG: every block of the grid takes a slice of each (the list is short: hubs only)
template <class T, class SR>
__global__ __launch_bounds__(256) void k_spmspv_push_long(const uint32_t* __restrict__ longlist, const uint32_t* __restrict__ rowptr,
                                                          const uint32_t* __restrict__ col, const T* __restrict__ aval, const T* __restrict__ uval,
                                                          const uint8_t* __restrict__ allow, T* __restrict__ tval, uint8_t* __restrict__ tpres, const SR sr,
                                                          uint32_t* __restrict__ any_true = nullptr, const uint32_t any_tag = 0, unsigned long long* __restrict__ fe_host = nullptr,
                                                          const SmallList64 excl = SmallList64{0, {0}}) {
  const uint32_t nl = longlist[0];
  const bool use_a = sr.uses_a(), use_u = sr.uses_u();
  unsigned long long fe = 0, fcnt = 0;
  // few long rows: every block takes a slice of each; many: one block per row
  const bool split = nl < gridDim.x / 8;
  for (uint32_t l = split ? 0 : blockIdx.x; l < nl; l += split ? 1 : gridDim.x) {
    const uint32_t i = longlist[1 + l];
    T ui = T(); if (use_u) { if (uval) ui = uval[i]; else ui = T(true); }
    const uint32_t pb = rowptr[i], pe = rowptr[i + 1];
    for (uint64_t p = (uint64_t)pb + (split ? blockIdx.x * 256ull : 0ull) + threadIdx.x; p < pe; p += split ? (uint64_t)gridDim.x * 256ull : 256ull) {
      const uint32_t j = col[p];
      if (allow && !allow[j]) continue;
      if (excl.n && small_list_has(excl, j)) continue;
      const T m = sr.mult(use_a ? aval[p] : T(), ui);
      atomic_combine<T>(sr.add_op(), &tval[j], m);
      tpres[j] = 1;
      if (any_true && spmv_truthy<T>(m)) *any_true = any_tag;
      if (fe_host && spmv_truthy<T>(m)) { fe += rowptr[j + 1] - rowptr[j]; fcnt++; }
    }
  }
  if (fe_host) fe_block_report(fe_host, 1u + blockIdx.x, any_tag, fe, fcnt);             // (pair 0 is the short-row kernel's)
}

// ---- host drivers ---------------------------------------------------------------------------------------------------------
static __global__ void k_col_locality(const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ col, uint32_t nrows, unsigned long long* __restrict__ out) {
  unsigned long long near = 0, seen = 0;
  for (uint64_t r = blockIdx.x * 256ull + threadIdx.x; r < nrows; r += (uint64_t)gridDim.x * 256ull * 16) {      // a sixteenth of the rows, at most 64 entries of each (a thread walks its row alone: 4096 took 3 ms on R-MAT's hub rows)
    const uint32_t b = rowptr[r]; uint32_t e = rowptr[r + 1]; if (e - b > 64u) e = b + 64u;
    for (uint32_t p = b + 1; p < e; p++) near += col[p] - col[p - 1] < 16u;
    if (e > b) seen += e - b - 1;
  }
  near = wave_reduce_add_u64(near); seen = wave_reduce_add_u64(seen);
  if ((threadIdx.x & 63) == 0 && seen) { atomicAdd(out, near); atomicAdd(out + 1, seen); }
}
// share (in %) of the sampled entries whose column lies within 16 of the previous entry of the row; measured once per matrix
static inline int spmv_locality_pct(DevCSR& M) {
  if (M.locality_pct >= 0) return M.locality_pct;
  DevBuf cnt(16); GRB_HIP(hipMemsetAsync(cnt.p, 0, 16, stream()));
  const unsigned nb = (unsigned)((M.nrows / 16 + 255) / 256 + 1);
  hipLaunchKernelGGL(k_col_locality, dim3(nb > 512 ? 512 : nb), dim3(256), 0, stream(), M.rowptr.as<uint32_t>(), M.col.as<uint32_t>(), M.nrows, (unsigned long long*)cnt.p);
  unsigned long long res[2] = {0, 0}; GRB_HIP(hipMemcpyAsync(res, cnt.p, 16, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
  M.locality_pct = res[1] ? (int)(100.0 * (double)res[0] / (double)res[1]) : 0;
  return M.locality_pct;
}

// ---- row heads (DevCSR::heads / nonempty) --------------------------------------------------------------------------------------------------
static __global__ void k_row_heads(uint32_t nrows, const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ col, const uint8_t* __restrict__ bval, uint4* __restrict__ heads, uint32_t* __restrict__ nonempty) {
  const uint64_t r = blockIdx.x * 256ull + threadIdx.x;
  uint32_t b = 0, e = 0;
  if (r < nrows) { b = rowptr[r]; e = rowptr[r + 1]; }
  const unsigned long long m = __ballot(e > b);
  if (r < nrows) {
    auto word = [&](uint32_t p) { return col[p] | ((!bval || bval[p]) ? 0x80000000u : 0u); };
    uint4 h = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    if (e > b) h.x = word(b);
    if (e > b + 1) h.y = word(b + 1);
    if (e > b + 2) h.z = word(b + 2);
    if (e > b + 3) h.w = word(b + 3) | (e > b + 4 ? 0x40000000u : 0u);
    heads[r] = h;
    if ((threadIdx.x & 31) == 0) nonempty[r >> 5] = (uint32_t)(m >> (threadIdx.x & 32));      // (the allocation holds whole words)
  }
}
// bool8_vals: the matrix values as one-byte BOOLs in place (nullptr: the caller's multiplier ignores them)
static inline bool spmv_row_heads(DevCSR& M, const uint8_t* bool8_vals) {
  if (M.heads_valid && (M.heads_vals || !bool8_vals)) return true;
  if (M.ncols >= 0x3FFFFFFFu || !M.nrows) return false;
  if (!M.heads_valid) { M.heads.alloc((size_t)M.nrows * 16 + 16); M.nonempty.alloc(((size_t)M.nrows + 31) / 32 * 4 + 64); }
  hipLaunchKernelGGL(k_row_heads, dim3((unsigned)(((uint64_t)M.nrows + 255) / 256)), dim3(256), 0, stream(), M.nrows, M.rowptr.as<uint32_t>(), M.col.as<uint32_t>(), bool8_vals, (uint4*)M.heads.p, M.nonempty.as<uint32_t>());
  M.heads_valid = true; M.heads_vals = bool8_vals != nullptr;
  return true;
}

template <class T> void run_pull(const SpmvCall& c, const SemiringDesc& d) {
  DevCSR& M = *c.M;
  with_semiring<T>(d, [&](auto sr) {
    typedef decltype(sr) SR;
    SpmvKArgs<T> a{};
    a.rowptr = M.rowptr.as<uint32_t>(); a.col = M.col.as<uint32_t>(); a.aval = (const T*)c.aval;
    a.uval = (const T*)c.uval; a.upres = c.upres; a.allow = c.allow; a.tval = (T*)c.tval; a.tpres = c.tpres; a.nrows = M.nrows; a.any_true = nullptr;
    a.fe_rowptr = nullptr; a.fe_host = nullptr; a.heads = nullptr; a.nonempty = nullptr;
    const bool full = c.upres == nullptr;
    // masked pull with a terminal monoid (BFS) -> row-group kernel with early exit; otherwise the row-block kernel
    const bool prefer_rowgroup = c.method == SPMV_ROWGROUP || (c.method == SPMV_AUTO && (c.allow || c.fm_val) && d.has_terminal);
    if (prefer_rowgroup) {
      const double avg = M.nrows ? (double)M.nnz / M.nrows : 0;
      const int G = avg > 96 ? 64 : 8;        // 8 lanes per row unless rows are long on average: most rows of a power-law graph are short

      uint64_t groups_per_block = 256 / G;
      uint64_t nb = (M.nrows + groups_per_block - 1) / groups_per_block; if (nb < 1) nb = 1; if (nb > 65536) nb = 65536;
#define GRB_LAUNCH_B(GG) \
      if (full) hipLaunchKernelGGL((k_spmv_rowgroup<T, SR, GG, true>), dim3((unsigned)nb), dim3(256), 0, stream(), a, sr); \
      else hipLaunchKernelGGL((k_spmv_rowgroup<T, SR, GG, false>), dim3((unsigned)nb), dim3(256), 0, stream(), a, sr);
      static const bool no_lane = wp_env("GRB_MI355X_NO_ROWLANE", 0) != 0;      // measurement hook
      if (G == 8 && !no_lane && c.method == SPMV_AUTO) {                          // short rows on average: a lane per row (kernel B')
        static const int lane_k = (int)wp_env("GRB_MI355X_LANE_K", SPMV_LANE_K);                    // measurement hook: rows per lane of the fused (BFS) instantiation
        uint64_t nbl = ((uint64_t)M.nrows + 255) / 256; if (nbl < 1) nbl = 1; if (nbl > 65536) nbl = 65536;
        if (c.any_true && c.any_true_done && is_bool<T>::value) {
          a.any_true = c.any_true; a.any_true_tag = c.any_true_tag; *c.any_true_done = true;
          if (c.fe_host && c.fe_done && c.fe_rowptr) { a.fe_rowptr = c.fe_rowptr; a.fe_host = c.fe_host; }      // (honoured by the fused launch below)
        }
        if constexpr (is_bool<T>::value) {
          if (c.fm_val) {                                                       // the mask is the operand itself: no allow / BOOL-value arrays were made
            a.fm_val = c.fm_val; a.fm_flags = c.fm_flags; a.fm_code = c.fm_code;
            const int kk = lane_k >= 4 ? 4 : lane_k >= 2 ? 2 : 1;
            static const uint32_t lane_cap = wp_env("GRB_MI355X_LANE_BLOCKS", 256);                    // measurement hooks: grid cap (the kernel strides), threads per workgroup
            static const uint32_t lane_thr = wp_env("GRB_MI355X_LANE_THREADS", 1024);
            const unsigned thr = lane_thr >= 1024 ? 1024u : lane_thr >= 512 ? 512u : 256u;
            uint64_t nbk = ((uint64_t)M.nrows + (uint64_t)thr * kk - 1) / ((uint64_t)thr * kk); if (nbk < 1) nbk = 1; if (nbk > lane_cap) nbk = lane_cap;
            if (a.fe_host) { if (nbk > FE_MAX_BLOCKS) nbk = FE_MAX_BLOCKS; *c.fe_done = true; *c.fe_nblocks = (uint32_t)nbk; }
            static const bool no_heads = wp_env("GRB_MI355X_NO_ROW_HEADS", 0) != 0;                    // measurement hook: round 5's first half (row pointers + column array)
            // (the value bits need the matrix values as one-byte BOOLs where they lie: a BOOL matrix under a Boolean semiring — the adjacency matrix of the BFS)
            const bool vals_in_place = c.aval == M.val.p;
            if (kk > 1 && !no_heads && (!sr.uses_a() || vals_in_place) && spmv_row_heads(M, sr.uses_a() ? (const uint8_t*)M.val.p : nullptr)) { a.heads = (const uint4*)M.heads.p; a.nonempty = M.nonempty.as<uint32_t>(); }
            if (kk == 4 && a.fm_code) hipLaunchKernelGGL((k_spmv_rowlane_k<T, SR, false, true, 4, true>), dim3((unsigned)nbk), dim3(thr), 0, stream(), a, sr);
            else if (kk == 4) hipLaunchKernelGGL((k_spmv_rowlane_k<T, SR, false, true, 4>), dim3((unsigned)nbk), dim3(thr), 0, stream(), a, sr);
            else if (kk == 2) hipLaunchKernelGGL((k_spmv_rowlane_k<T, SR, false, true, 2>), dim3((unsigned)nbk), dim3(thr), 0, stream(), a, sr);
            else hipLaunchKernelGGL((k_spmv_rowlane<T, SR, false, true>), dim3((unsigned)nbk), dim3(thr), 0, stream(), a, sr);
            g_last_plan += std::string("k_spmv_rowlane<") + (sr.is_static ? "static" : "dynamic") + ",mask=operand" + (kk == 4 && a.fm_code ? ",code bytes" : "") + "> ";
            return;
          }
        }
        a.fe_host = nullptr;                                                     // (the summary rides with the fused instantiation only)
        if (full) hipLaunchKernelGGL((k_spmv_rowlane<T, SR, true>), dim3((unsigned)nbl), dim3(256), 0, stream(), a, sr);
        else hipLaunchKernelGGL((k_spmv_rowlane<T, SR, false>), dim3((unsigned)nbl), dim3(256), 0, stream(), a, sr);
        g_last_plan += std::string("k_spmv_rowlane<") + (sr.is_static ? "static>" : "dynamic>") + " ";
        return;
      }
      if (G == 64) { GRB_LAUNCH_B(64) } else { GRB_LAUNCH_B(8) }
#undef GRB_LAUNCH_B
      g_last_plan += std::string("k_spmv_rowgroup<G=") + std::to_string(G) + (sr.is_static ? ",static>" : ",dynamic>") + " ";
      return;
    }
    // kernel W: full operand, no mask, large matrix
    // ... unless the gathers have locality of their own (banded / mesh-like matrices: consecutive entries of a row read the
    // same 128-byte line of u).  Then the caches serve them, and the row-block kernel — no sub-rows, no merge — is the
    // fast one (measured on a 16-diagonal band of 1.7e7 entries: A 0.035 ms, W 0.058 ms, X 0.072 ms).
    const bool local_gathers = c.method == SPMV_AUTO && M.nnz >= (1u << 20) && spmv_locality_pct(M) >= 50;
    if constexpr (sizeof(T) >= 4) if (!local_gathers) {
      // kernel X (one column panel per XCD) for the big ones; it keeps a panel-major copy of the matrix
      const bool want_x = c.method == SPMV_XCD || (c.method == SPMV_AUTO && M.nnz >= (1u << 22));
      // (32-bit byte offsets into a panel's streams and into u: a panel holds ~nnz/8 entries)
      if (want_x && full && !c.allow && M.ncols < 0x0F000000u && M.nnz >= (uint64_t)WP_ENT * 64 && M.nnz * sizeof(T) < (7ull << 30) && device_cus() > 0) {
        // Plan policy (round 4): kernel X's plan is a panel-major copy of the matrix, 6-7 ms at R-MAT-22 — 27 products.  A caller that
        // multiplies ONCE (the literal `Matrix.mxv`, pygraphblas/matrix.py:2714-2725) must not pay it: the first `GRB_MI355X_XPLAN_AFTER`
        // (default 1) full-operand products of a matrix run kernel W, whose plan is a sampled ranking + one pass (< 1 ms); the product
        // after them builds X's plan, and W's is dropped.  0 = build X's plan at the first product, as rounds 1-3 did.
        // FP32 keeps the eager plan: kernel W adds the 256-entry tasks of a long row one after the other, kernel X in blocks (sub-rows, then
        // the merge) — over the 3.7e5 terms of an R-MAT-25 hub row W's FP32 sum drifts past the 1e-6 the north star allows, X's does not
        // (tests/test_baseline_configs_gpu.py::test_config4_rmat25_pagerank_single_gpu).
        static const uint32_t after_env = wp_env("GRB_MI355X_XPLAN_AFTER", 1);
        const uint32_t after = std::is_same<T, float>::value ? 0u : after_env;
        const XcdPlan* have = static_cast<const XcdPlan*>(M.xcd.get());
        const bool planned = have && have->tsize == (int)sizeof(T);
        if (planned || c.method == SPMV_XCD || M.pipe_uses >= after) {
          if (run_xcd<T>(c, d, device_cus())) {
            if (!planned && M.wp_tsize) { M.wp_rs.reset(); M.wp_hot.reset(); M.wp_pcol.reset(); M.wp_carry.reset(); M.wp_nhot = M.wp_ntasks = 0; M.wp_tsize = 0; }
            return;
          }
        }
        M.pipe_uses++;
      }
      const bool want = c.method == SPMV_WAVEPIPE || c.method == SPMV_XCD || (c.method == SPMV_AUTO && M.nnz >= (1u << 20));
      if (want && full && !c.allow && M.ncols < 0x70000000u && M.nnz >= (uint64_t)WP_ENT && device_cus() > 0) {
        run_wavepipe<T>(c, d, device_cus());
        return;
      }
    }
    spmv_build_plan(M);
    a.blocks = M.plan_blocks.as<SpmvBlock>();
    uint8_t* aux = M.plan_aux.as<uint8_t>();
    const size_t ns = (size_t)M.plan_nlong + 1;
    a.partial = (T*)aux; a.tickets = (uint32_t*)(aux + ns * 8); a.pflag = aux + ns * 8 + ns * 4;
    if (M.plan_nblocks == 0) return;
    const dim3 grid(M.plan_nblocks), block(SPMV_THREADS);
    if constexpr (std::is_same<T, double>::value && SR::is_static) {
      const char* eg = getenv("GRB_MI355X_GATHER"); const char* es = getenv("GRB_MI355X_STREAM");
      const int gm = eg ? atoi(eg) : 0, sm = es ? atoi(es) : 0;
      if (full && !c.allow && (gm || sm)) {
#define GRB_EXP(G_, S_) if (gm == G_ && sm == S_) { hipLaunchKernelGGL((k_spmv_adaptive<T, SR, true, false, G_, S_>), grid, block, 0, stream(), a, sr); g_last_plan += "k_spmv_adaptive<exp g" #G_ " s" #S_ "> "; return; }
        GRB_EXP(0, 1) GRB_EXP(1, 0) GRB_EXP(1, 1) GRB_EXP(2, 0) GRB_EXP(2, 1)
#undef GRB_EXP
      }
    }
    if (full) {
      if (c.allow) hipLaunchKernelGGL((k_spmv_adaptive<T, SR, true, true>), grid, block, 0, stream(), a, sr);
      else hipLaunchKernelGGL((k_spmv_adaptive<T, SR, true, false>), grid, block, 0, stream(), a, sr);
    } else {
      if (c.allow) hipLaunchKernelGGL((k_spmv_adaptive<T, SR, false, true>), grid, block, 0, stream(), a, sr);
      else hipLaunchKernelGGL((k_spmv_adaptive<T, SR, false, false>), grid, block, 0, stream(), a, sr);
    }
    g_last_plan += std::string("k_spmv_adaptive<") + (sr.is_static ? "static" : "dynamic") + (full ? ",full" : ",bitmap") + (c.allow ? ",mask> " : "> ");
  });
}


template <class T> __global__ void k_fill(T* p, uint64_t n, T v) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = v;
}

// zero fill of the presence bytes, identity fill of the values (16 positions per thread), the operand's list and the cleared hub-row counter: one launch
template <class T> __global__ void k_push_init(T* __restrict__ tval, uint8_t* __restrict__ tpres, uint64_t n, T ident, const SmallList64 sl, uint32_t* __restrict__ fidx, uint32_t* __restrict__ longlist) {
  if (blockIdx.x == 0) { if (threadIdx.x < sl.n) fidx[threadIdx.x] = sl.idx[threadIdx.x]; if (threadIdx.x == 64) longlist[0] = 0; }
  const uint64_t n16 = n / 16;
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += gridDim.x * 256ull) {
    ((uint4*)tpres)[i] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 16; q++) tval[i * 16 + q] = ident;
  }
  for (uint64_t i = n16 * 16 + blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) { tpres[i] = 0; tval[i] = ident; }
}
template <class T> void run_push(const SpmvCall& c, const SemiringDesc& d, const uint32_t* fidx, uint64_t u_nvals, uint32_t* longlist) {
  DevCSR& M = *c.M; const uint64_t nout = M.ncols;
  uint32_t* const fidx_w = const_cast<uint32_t*>(fidx);
  auto grid_of = [](uint64_t n) { uint64_t b = (n + 255) / 256; if (b < 1) b = 1; if (b > 4096) b = 4096; return (unsigned)b; };
  with_semiring<T>(d, [&](auto sr) {
    typedef decltype(sr) SR;
    // the accumulator starts at the identity with no entry; with the operand's list known on the host (<= 64 entries, SpmvCall::small_idx) the
    // same launch writes the list and clears the hub-row counter (round 5: six launches -> three for the first level of a BFS)
    if (c.small_idx && c.small_n == u_nvals && u_nvals <= 64) {
      SmallList64 sl; sl.n = (uint32_t)u_nvals; for (uint32_t q = 0; q < sl.n; q++) sl.idx[q] = c.small_idx[q];
      hipLaunchKernelGGL((k_push_init<T>), dim3(grid_of(nout / 16 + 1)), dim3(256), 0, stream(), (T*)c.tval, c.tpres, nout, sr.identity, sl, (uint32_t*)fidx_w, longlist);
    } else if (nout) hipLaunchKernelGGL((k_fill<T>), dim3(grid_of(nout)), dim3(256), 0, stream(), (T*)c.tval, nout, sr.identity);
    uint64_t nb = (u_nvals + 3) / 4; if (nb < 1) nb = 1; if (nb > 16384) nb = 16384;
    SmallList64 excl; excl.n = 0;
    if (c.excl_small && c.small_idx && c.small_n == u_nvals && u_nvals <= 64) { excl.n = (uint32_t)u_nvals; for (uint32_t q = 0; q < excl.n; q++) excl.idx[q] = c.small_idx[q]; }
    uint32_t* any_true = nullptr; unsigned long long* fe_host = nullptr;
    if (c.any_true && c.any_true_done && is_bool<T>::value && (sr.add_op() == B_LOR || sr.add_op() == B_PLUS || sr.add_op() == B_MAX)) {
      any_true = c.any_true; *c.any_true_done = true;
      // one operand entry: the columns of its row are distinct, so the true entries written and the edges leaving them are counted exactly
      if (u_nvals == 1 && c.fe_host && c.fe_done && c.fe_rowptr == M.rowptr.as<uint32_t>()) { fe_host = c.fe_host; *c.fe_done = true; *c.fe_nblocks = 1u + 1024u; }
    }
    hipLaunchKernelGGL((k_spmspv_push<T, SR>), dim3((unsigned)nb), dim3(256), 0, stream(), fidx, (uint32_t)u_nvals,
                       M.rowptr.as<uint32_t>(), M.col.as<uint32_t>(), (const T*)c.aval, (const T*)c.uval, c.allow, (T*)c.tval, c.tpres, longlist, sr, any_true, c.any_true_tag, fe_host, excl);
    hipLaunchKernelGGL((k_spmspv_push_long<T, SR>), dim3(1024), dim3(256), 0, stream(), longlist, M.rowptr.as<uint32_t>(), M.col.as<uint32_t>(),
                       (const T*)c.aval, (const T*)c.uval, c.allow, (T*)c.tval, c.tpres, sr, any_true, c.any_true_tag, fe_host, excl);
    g_last_plan += std::string("k_spmspv_push<") + (sr.is_static ? "static> " : "dynamic> ");
  });
}

}  // namespace grb
