// grb_spmv_xcd.hpp — SpMV kernel "X": kernel W run on eight column panels at once, one panel per XCD.
//
// Why (profiles/spmv_pmc_traffic.json): kernel W's time is its memory-side read traffic, and 60 % of that traffic is
// 128-byte line fills for gathers of u that miss the 4 MiB per-XCD L2 — every XCD sees all 32 MiB of u, so the eight
// L2s cache eight copies of the same hot 4 MiB.  Kernel X gives each XCD its own eighth of u:
//   * u is cut into 128-byte lines; every line (16 FP64 columns) belongs to one panel.  Lines are dealt to the panels so
//     that the panels hold the same number of entries (heaviest lines first to the lightest panel, then in snake order);
//   * the plan stores the matrix panel-major: for each panel the column words and the values of its entries in row-major
//     order, cut into tiles of 256 entries.  A column word is the slot in the panel's LDS table for the panel's H most
//     frequent columns, H + column for the others; bit 31 marks the first entry of a *sub-row* (the entries of one row
//     in one panel; a sub-row also ends at every chunk boundary, so no sub-row spans two chunks of tiles);
//   * a small kernel gathers the contents of the eight LDS tables from u once per call (through the panels' hot-column
//     lists); workgroup b (observed to run on XCD b % 8) runs the tile pipeline (grb_spmv_tiles.hpp) on panel b & 7:
//     table for the hot columns, every other gather reads u itself and touches only the panel's lines, which its XCD's L2
//     keeps — so the aggregate 32 MiB of L2 holds u once and u is never copied or re-ordered per call;
//   * each sub-row's sum goes to a partial array; a merge kernel (2048 rows per workgroup, the eight contiguous runs of
//     their partials accumulated panel after panel in LDS) adds the partials of every row in a fixed order
//     (=> reproducible) and writes y.
// Extra algorithmic cost: one partial (8 B written + read) and one 16-bit row id per sub-row (~7.8 M at R-MAT-22)
// against ~1.3 GB of avoided line fills.  Placement is used for speed only: any other block->XCD mapping is still correct.
//
// The plan (round 2) is built in a handful of passes over the entries, with no full-size sort and two host round trips
// (sizes for the allocations): column counts (LDS-aggregated atomics), the line deal and the per-panel column ranking
// (sorts of `nlines` and `ncols` 32-bit keys), then two sweeps over units of 16384 entries in row-major order — a
// counting sweep (entries per unit and panel) and, after a scan, a scattering sweep that writes every entry to its
// place in its panel (wave ballots rank the entries of a panel inside a step; the order inside a panel stays row-major,
// so the plan is a deterministic function of the matrix) — and two passes over the tiles that number the sub-rows.
#pragma once
#include "grb_spmv_tiles.hpp"
#include "grb_matops.hpp"

namespace grb {

constexpr int XP = 8;                 // physical panels = XCDs: eight entry streams, one tile pipeline each
constexpr int XPMAX = 64;             // virtual panels at most: XP x the sub-panels an XCD's share of the columns is cut into (round 4)
constexpr uint32_t XP_RB = 1024;      // rows per workgroup of the merge kernel
constexpr int XP_CT = 512;            // its threads
constexpr uint32_t XP_UNIT = 16384;   // entries per unit of the plan-building sweeps
constexpr int XP_ST = 512;            // threads of a sweep workgroup
constexpr int XP_SW = XP_ST / 64;     // its waves

struct XcdPlan {          // lives in DevCSR::xcd (type-erased), built once per matrix and value type
  DevBuf hot_cols;        // u32[XP*H]   column held by slot h of panel k's LDS table
  DevBuf pcol, pval;      // u32[.], T[.] panel-major entries (column word, value); panel k starts at tile tbase[k].  pcol is dropped once packed
  DevBuf trow;            // u32[tiles]  sub-row (numbered over all panels, panel after panel) of every tile's first entry
  DevBuf col16, extras, tinfo;   // the 16-bit column plane the kernel streams (grb_spmv_tiles.hpp): u16[.] words, u16[ncold] low halves of the cold columns, u32[2*tiles] {sub-row, first extra}
  uint64_t ncold = 0;
  DevBuf lrow;            // u16[F]      row of every sub-row relative to its block of XP_RB rows — what the merge kernel streams
  DevBuf blockptr;        // u32[(nblocks+1)*XP] first sub-row of panel k in row block b
  // merge in row-major slot order (k_xp_merge): variable row blocks of <= XM_ROWS rows and ~XM_TARGET sub-rows
  DevBuf m_bstart;        // u32[m_nblocks+1] first row of every block
  DevBuf m_blockptr;      // u32[(m_nblocks+1)*XP] first sub-row of panel k in block b
  DevBuf m_slot;          // u16[F]  slot of every sub-row in its block's row-major order (row, then panel, then position in a chain)
  DevBuf m_rowoff;        // u16[nrows] slot of every row's first sub-row in its block
  bool m_wide = false;    // the blocks were cut for k_xp_merge_wide (always with sub-panels; GRB_MI355X_XM_WIDE=1: also without)
  uint32_t m_nblocks = 0, m_slots = 0; bool m_ok = false;      // m_slots: LDS slots a block may need (XM_TARGET + the longest row's sub-rows)
  DevBuf args;            // XtPanel<T>[XP] in HBM; never changes between calls (u and the partial array arrive as kernel arguments)
  DevBuf xhot, partial;   // per-call work buffers kept with the plan (xhot: T[XP*H], the LDS tables' contents)
  uint64_t ne[XPMAX]; uint32_t tbase[XPMAX + 1]; uint32_t ntiles[XPMAX], nhot[XPMAX];      // per stream (NS of them: tile-aligned, one XtPanel and one LDS table each)
  uint64_t F = 0; int tsize = 0; bool has_vals = false; float build_ms = 0;
  int vbytes = 0;          // bytes per value of the panel-major value plane: sizeof(T), or 2 (int16: an integer matrix whose values all fit — pval then holds int16)
  // round 4: S sub-panels per XCD (virtual panel vp = k * S + s lives in physical panel k).  When an XCD's eighth of the operand does not
  // fit its 4 MiB L2 (R-MAT-25 FP32: 16.8 MB; every rank of a row-partitioned run), the lines of u are dealt to XP * S virtual panels and a
  // physical panel's stream holds its S sub-panels one after the other, each in row-major order: the waves of an XCD walk the stream
  // front to back together, so at any time their cold gathers touch ONE sub-panel's lines — which fit.  Only COLD entries are bound to a
  // sub-panel; an entry served by the LDS table (one table per XCD, as before) rides in a sub-panel where its row already has a cold
  // entry, so sub-rows multiply by ~1.3 at S = 4 instead of ~2 (tools/subpanel_model.py).  The tile pipeline knows nothing of this.
  // `own` (GRB_MI355X_XOWN=1): every sub-panel is a stream of its own WITH ITS OWN LDS TABLE — the pipeline's workgroups walk the S
  // streams of their XCD one after the other, refilling the table in between.  S times the table slots: fewer cold gathers (each costs the
  // CU a 128-byte L2->L1 line fill, which is what bounds the pipeline once a third of the entries are cold: R-MAT-25, 37 %), more sub-rows
  // (an entry the table serves is now bound to its column's sub-panel as well).
  int S = 1, NP = XP, NS = XP; bool own = false;
  DevBuf vfirst;          // u32[NP + 1] first sub-row of every virtual panel (sub-rows are numbered in stream order)
  // round 6: the lane-per-piece layout of the same streams (grb_spmv_sell.hpp): steps of 64 column words / values, the partial id of every chunk lane,
  // the work items, the streams' argument block
  DevBuf s_col, s_val, s_perm, s_items, s_args; bool sell = false; uint64_t s_nsteps = 0; uint32_t s_nchunks = 0, s_nitems = 0;
};
extern float g_xcd_plan_build_ms;     // duration of the most recent plan build (grb_spmv.hip; read by GrBX_last_plan_build_ms)
}  // namespace grb
#include "grb_spmv_sell.hpp"
namespace grb {
// does the plan of a matrix of this value type carry the lane-per-piece layout?  (GRB_MI355X_SELL=0: the tile pipeline only)
template <class T> inline bool xp_sell_wanted() { return (sizeof(T) == 4 || sizeof(T) == 8) && wp_env("GRB_MI355X_SELL", 0) != 0; }

// ---- plan pieces ----------------------------------------------------------------------------------------------------------
// column counts = run lengths of the sorted column array.  (Counting with atomics — LDS-aggregated per workgroup, the rest
// straight into HBM — took 4.5-4.7 ms at R-MAT-22 whatever the grid: ~40 M of the 65 M entries belong to columns too cold
// for a workgroup's LDS table, and the chip completes ~9 G device-scope atomics per second.  One 22-bit radix sort of the
// keys and two streaming passes take about a third of that.)
// (k_xp_run_starts / k_xp_run_lengths: grb_spmv_wavepipe.hpp — kernel W's sampled column counts use them too)
// weight of a line of u = entries in its columns
static __global__ void k_xp_line_weights(const uint32_t* __restrict__ cnt, uint32_t n, uint32_t line, uint32_t nlines, uint32_t* __restrict__ negw, uint32_t* __restrict__ id) {
  for (uint32_t l = blockIdx.x * 256 + threadIdx.x; l < nlines; l += gridDim.x * 256) {
    uint64_t w = 0; for (uint32_t j = 0; j < line; j++) { const uint64_t c = (uint64_t)l * line + j; if (c < n) w += cnt[c]; }
    negw[l] = 0xFFFFFFFFu - (uint32_t)(w > 0xFFFFFFFEull ? 0xFFFFFFFEull : w); id[l] = l;
  }
}
// the heavy head of the weight distribution (<= 4096 lines, heaviest first) is balanced exactly: each line to the panel with
// the least load so far (LPT).  One workgroup; the loop itself is serial.
// Deal slot j of the np = XP * S slots is virtual panel (j % XP) * S + j / XP: consecutive slots go to different XCDs.
__device__ __forceinline__ uint32_t xp_vp_of_slot(uint32_t j, uint32_t S) { return (j & (XP - 1)) * S + (j >> 3); }
// (round 4: the deal runs on one WAVE — lane j holds the load of slot j, the lightest slot is a wave minimum + a ballot — instead of one
//  thread comparing the slots one after the other: 1.3 ms -> ~0.1 ms of the plan's 6 ms at R-MAT-22.  Loads are 32-bit: a slot never
//  holds more than the matrix's entries, which a 32-bit count addresses anyway.)
static __global__ __launch_bounds__(64) void k_xp_lpt(const uint32_t* __restrict__ negw_sorted, uint32_t ntop, uint32_t np, uint32_t S, uint8_t* __restrict__ top_panel) {
  __shared__ uint32_t w[4096];
  const uint32_t lane = threadIdx.x;
  for (uint32_t i = lane; i < ntop; i += 64) w[i] = 0xFFFFFFFFu - negw_sorted[i];
  __syncthreads();
  uint32_t load = lane < np ? 0u : 0xFFFFFFFFu;           // (lanes beyond the slots never win)
  for (uint32_t i = 0; i < ntop; i++) {
    const uint32_t m = wave_reduce_dpp<uint32_t, false>(B_MIN, load, 0xFFFFFFFFu);       // (DPP steps; the compiler's wave_reduce builtin walks the lanes one by one: 7.5 ms for this loop)
    const uint32_t best = (uint32_t)__builtin_ctzll(__ballot(load == m));      // the first slot with the least load (ties: lowest index, as the serial loop chose)
    if (lane == best) { const uint32_t nl = load + w[i]; load = nl < load || nl == 0xFFFFFFFFu ? 0xFFFFFFFEu : nl; }
    if (lane == 0) top_panel[i] = (uint8_t)xp_vp_of_slot(best, S);
  }
}
// lines in descending weight: the first `ntop` take the panel the LPT chose for them, the others are dealt in snake order
static __global__ void k_xp_deal_lines(const uint32_t* __restrict__ sorted_line, uint32_t nlines, const uint8_t* __restrict__ top_panel, uint32_t ntop, uint32_t np, uint32_t S,
                                       uint8_t* __restrict__ panel_of_line) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nlines; i += gridDim.x * 256) {
    uint32_t k;
    if (i < ntop) k = top_panel[i]; else { const uint32_t m = (i - ntop) % (2 * np); k = xp_vp_of_slot(m < np ? m : 2 * np - 1 - m, S); }
    panel_of_line[sorted_line[i]] = (uint8_t)k;       // the line's virtual panel
  }
}
// key = panel | descending count (clamped to 20 bits: ties among the rarest and among the very hottest columns do not matter),
// stable sort => a panel's columns in frequency order, ties by index
static __global__ void k_xp_column_keys(const uint32_t* __restrict__ cnt, uint32_t n, uint32_t line, const uint8_t* __restrict__ panel_of_line, uint32_t S, uint32_t* __restrict__ key, uint32_t* __restrict__ colv) {
  for (uint32_t c = blockIdx.x * 256 + threadIdx.x; c < n; c += gridDim.x * 256) {
    const uint32_t w = cnt[c] < 0xFFFFFu ? cnt[c] : 0xFFFFFu;
    key[c] = (((uint32_t)panel_of_line[c / line] / S) << 20) | (0xFFFFFu - w);       // the PHYSICAL panel: one LDS table per XCD
    colv[c] = c;
  }
}
static __global__ void k_xp_panel_starts(const uint32_t* __restrict__ key, uint32_t n, uint32_t* __restrict__ start) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t k = key[i] >> 20;
    if (i == 0 || (key[i - 1] >> 20) != k) start[k] = i;
  }
}
static __global__ void k_xp_fix_starts(uint32_t* __restrict__ start, uint32_t n, uint32_t ns) {      // a panel without columns starts where the next one does
  if (blockIdx.x == 0 && threadIdx.x == 0) { start[ns] = n; for (int k = (int)ns - 1; k >= 0; k--) if (start[k] == 0xFFFFFFFFu) start[k] = start[k + 1]; }
}
// code word of a column (round 4): its slot in its stream's LDS table when it is one of the `hot` most frequent columns there, XT_COLD | column
// otherwise (bit 31 is left for the row-start flag of the entry words); the stream's hot-column list.  The stream of a column comes from the
// line deal (pol[]), not from the word.
static __global__ void k_xp_column_codes(const uint32_t* __restrict__ key, const uint32_t* __restrict__ cols, uint32_t n, uint32_t H, uint32_t hot, const uint32_t* __restrict__ start,
                                         uint32_t* __restrict__ code, uint32_t* __restrict__ hot_cols) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t k = key[i] >> 20, c = cols[i], local = i - start[k];
    code[c] = local < hot ? local : (XT_COLD | c);
    if (local < hot) hot_cols[(size_t)k * H + local] = c;
  }
}
// the hot columns of a stream take their slots in COLUMN order (round 4): the table gather of a call (k_xp_hot_gather: xhot[slot] = u[hot column]) then walks u
// front to back inside every stream — with the slots in order of popularity its 2.5e6 gathers of a 64-table plan were 2.5e6 separate 128-byte lines.
// Every stream contributes exactly H keys (its unused slots sort behind its columns), so sorted position i is slot i - k H of stream k.
static __global__ void k_xp_hot_keys(const uint32_t* __restrict__ hot_cols, const uint32_t* __restrict__ start, uint32_t ns, uint32_t H, uint32_t hot, unsigned long long* __restrict__ key, uint32_t* __restrict__ val) {
  const uint64_t tot = (uint64_t)ns * H;
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < tot; i += gridDim.x * 256ull) {
    const uint32_t k = (uint32_t)(i / H), slot = (uint32_t)(i - (uint64_t)k * H), have = start[k + 1] - start[k];
    const bool valid = slot < hot && slot < have;
    key[i] = ((unsigned long long)k << 32) | (valid ? hot_cols[i] : 0xFFFFFFFFu); val[i] = 0;
  }
}
static __global__ void k_xp_hot_reslot(const unsigned long long* __restrict__ key, uint32_t ns, uint32_t H, uint32_t* __restrict__ code, uint32_t* __restrict__ hot_cols) {
  const uint64_t tot = (uint64_t)ns * H;
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < tot; i += gridDim.x * 256ull) {
    const uint32_t k = (uint32_t)(key[i] >> 32), c = (uint32_t)key[i], slot = (uint32_t)(i - (uint64_t)k * H);
    if (c != 0xFFFFFFFFu) { hot_cols[i] = c; code[c] = slot; } else hot_cols[i] = 0;
  }
}
// row of every entry: the non-empty rows mark their first entry, an inclusive max-scan fills the rest
static __global__ void k_xp_mark_rows(const uint32_t* __restrict__ rowptr, uint32_t nrows, uint32_t* __restrict__ rowidx) {
  for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < nrows; r += gridDim.x * 256) { const uint32_t s = rowptr[r]; if (rowptr[r + 1] > s) rowidx[s] = r; }
}

// The two sweeps over the entries in row-major order, one workgroup per unit of XP_UNIT entries, XP_ST entries per step.
// In a step every wave ballots its lanes panel by panel: the rank of an entry among the entries of its panel in the unit
// follows from the ballots, the counts of the waves before it and the running count of the unit; the row of the entry
// that precedes it in its panel (same wave: a shuffle; earlier: the waves' last rows of the step, then the running
// last row) tells whether it starts a sub-row.
//   SCATTER = false: entries per (panel, unit) -> ne, and the row of the unit's last entry per panel -> lastkey
//                    (index + 1 in the high word, so that an exclusive max-scan finds the nearest earlier unit that has one)
//   SCATTER = true : every entry goes to its place in its panel; bit 31 of the column word = first entry of a sub-row
//                    (the row changes, or a chunk of tiles begins); the rows of those entries go to rowtmp.
// Buckets of the sweeps = the np virtual panels.  With S > 1 sub-panels per XCD the bucket of a COLD entry is its column's (the line
// deal), the bucket of an entry the LDS table serves is a sub-panel of its XCD in which its row has a cold entry (rowmask: bit vp of
// word r = row r has a cold entry in virtual panel vp; lowest such sub-panel) — or, failing that, sub-panel r mod S.
template <class T> struct XpSweep {
  const uint32_t* col; const uint32_t* rowidx; const uint32_t* code; const T* val; uint64_t nnz; uint32_t nunits;
  uint32_t np, S, H, lshift, own; const uint8_t* pol; const unsigned long long* rowmask;
  uint32_t* ne; unsigned long long* lastkey;
  const uint32_t* escan; const unsigned long long* carry;
  uint64_t ebase[XPMAX];            // where virtual panel vp starts in the store (its physical panel's base + the sub-panels before it)
  uint32_t vpoff[XPMAX];            // ... and inside its physical panel's stream (chunks are cut on the physical stream)
  uint32_t chunk_entries[XPMAX];
  uint32_t* pcol; T* pval; uint32_t* rowtmp;
};
template <class T> __device__ __forceinline__ uint32_t xp_bucket(const XpSweep<T>& a, uint32_t code, uint32_t c, uint32_t r) {
  const uint32_t vp = (uint32_t)a.pol[c >> a.lshift];                // the virtual panel of the column's line
  if (a.S == 1u || a.own || (code & XT_COLD)) return vp;
  const uint32_t k = vp / a.S;
  const uint32_t bits = (uint32_t)(a.rowmask[r] >> (k * a.S)) & ((1u << a.S) - 1u);
  return k * a.S + (bits ? (uint32_t)__builtin_ctz(bits) : (r & (a.S - 1u)));
}
static __global__ void k_xp_rowmask(const uint32_t* __restrict__ col, const uint32_t* __restrict__ rowidx, const uint32_t* __restrict__ code, const uint8_t* __restrict__ pol, uint64_t nnz, uint32_t H,
                                    uint32_t lshift, unsigned long long* __restrict__ rowmask) {
  for (uint64_t p = blockIdx.x * 256ull + threadIdx.x; p < nnz; p += gridDim.x * 256ull) {
    const uint32_t c = col[p];
    if (!(code[c] & XT_COLD)) continue;
    const uint32_t r = rowidx[p]; const unsigned long long m = 1ull << pol[c >> lshift];
    // (read at the L2 first: a hub row's 10^5 cold entries would otherwise queue on one address, one atomic per ~80 ns)
    if (!(__hip_atomic_load(&rowmask[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & m)) atomicOr(&rowmask[r], m);
  }
}
template <class T, bool SCATTER>
__global__ __launch_bounds__(XP_ST) void k_xp_sweep(const XpSweep<T> a) {
  __shared__ uint32_t s_cnt[XP_SW][XPMAX], s_last[XP_SW][XPMAX], s_cursor[XPMAX], s_carry[XPMAX];
  const uint32_t u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t XP = a.np;                    // (shadows the constant: the buckets of this kernel are the virtual panels)
  if (tid < XP) {
    s_cursor[tid] = 0; uint32_t cr = WP_NONE;
    if constexpr (SCATTER) {
      const unsigned long long key = a.carry[(size_t)tid * a.nunits + u];
      if (key != 0 && (key >> 32) - 1 >= (unsigned long long)tid * a.nunits) cr = (uint32_t)key;       // the last entry of this panel before the unit
    }
    s_carry[tid] = cr;
  }
  __syncthreads();
  const uint64_t base = (uint64_t)u * XP_UNIT, end = base + XP_UNIT < a.nnz ? base + XP_UNIT : a.nnz;
  for (uint64_t sb = base; sb < end; sb += XP_ST) {
    const uint64_t p = sb + tid; const bool valid = p < end;
    const uint32_t c = valid ? a.col[p] : 0u, code = valid ? a.code[c] : 0u, r = valid ? a.rowidx[p] : 0u;
    const uint32_t k = valid ? xp_bucket(a, code, c, r) : 0xFFu;
    uint32_t myrank = 0; int plane = (int)lane; bool has_prev = false;
    for (uint32_t kk = 0; kk < XP; kk++) {
      const unsigned long long m = __ballot(k == kk);
      if (lane == 0) { s_cnt[w][kk] = (uint32_t)__popcll(m); }
      const int top = m ? 63 - __builtin_clzll(m) : 0;
      const uint32_t lastr = (uint32_t)__builtin_amdgcn_readlane((int)r, top);
      if (lane == 0) s_last[w][kk] = m ? lastr : WP_NONE;
      if (k == kk) {
        const unsigned long long below = m & ((1ull << lane) - 1ull);
        myrank = (uint32_t)__popcll(below);
        if (below) { plane = 63 - __builtin_clzll(below); has_prev = true; }
      }
    }
    const uint32_t prevrow = (uint32_t)__shfl((int)r, plane, 64);
    __syncthreads();
    if (valid) {
      uint32_t off = s_cursor[k], prow = has_prev ? prevrow : s_carry[k];
      for (uint32_t w2 = 0; w2 < w; w2++) { off += s_cnt[w2][k]; if (!has_prev && s_last[w2][k] != WP_NONE) prow = s_last[w2][k]; }
      if constexpr (SCATTER) {
        const uint32_t rel = (a.escan[(size_t)k * a.nunits + u] - a.escan[(size_t)k * a.nunits]) + off + myrank;     // position in the virtual panel
        const bool flag = prow != r || (a.vpoff[k] + rel) % a.chunk_entries[k] == 0;       // (prow == WP_NONE is never a row; chunks are cut on the physical stream)
        const uint64_t dest = a.ebase[k] + rel;
        a.pcol[dest] = code | (flag ? WP_ROWSTART : 0u);
        if (a.pval) a.pval[dest] = a.val[p];
        if (flag) a.rowtmp[dest] = r;
      }
    }
    __syncthreads();
    if (tid < XP) {
      uint32_t add = 0, cr = s_carry[tid];
      for (int w2 = 0; w2 < XP_SW; w2++) { add += s_cnt[w2][tid]; if (s_last[w2][tid] != WP_NONE) cr = s_last[w2][tid]; }
      s_cursor[tid] += add; s_carry[tid] = cr;
    }
    __syncthreads();
  }
  if constexpr (!SCATTER) {
    if (tid < XP) {
      a.ne[(size_t)tid * a.nunits + u] = s_cursor[tid];
      a.lastkey[(size_t)tid * a.nunits + u] = s_carry[tid] != WP_NONE ? (((unsigned long long)tid * a.nunits + u + 1) << 32) | s_carry[tid] : 0ull;
    }
  }
}
static __global__ void k_xp_pick(const uint32_t* __restrict__ escan, uint32_t nunits, uint32_t np, const uint32_t* __restrict__ cstart, uint32_t ns, uint32_t* __restrict__ out) {
  if (threadIdx.x <= np) out[threadIdx.x] = escan[(size_t)threadIdx.x * nunits];           // entries before virtual panel t
  if (threadIdx.x <= ns) out[XPMAX + 1 + threadIdx.x] = cstart[threadIdx.x];                 // columns before stream t (in its frequency order)
}
// first sub-row of every virtual panel = row-start flags before its first entry (a sub-panel may begin in the middle of a tile)
struct XpStarts { unsigned long long v[XPMAX + 1]; };
static __global__ __launch_bounds__(64) void k_xp_vp_first(const uint32_t* __restrict__ pcol, const uint32_t* __restrict__ E, const XpStarts st, uint32_t np, uint32_t ntiles, uint32_t* __restrict__ vfirst) {
  const uint32_t vp = blockIdx.x, lane = threadIdx.x;
  if (vp >= np) { if (lane == 0) vfirst[np] = E[ntiles]; return; }
  const unsigned long long q = st.v[vp]; const uint32_t g = (uint32_t)(q / WP_ENT), off = (uint32_t)(q % WP_ENT);
  uint32_t c = 0;
  if (off)
    for (uint32_t j = 0; j < (uint32_t)WP_PER; j++) { const uint32_t pos = lane * WP_PER + j; if (pos < off) c += pcol[(size_t)g * WP_ENT + pos] >> 31; }
  const uint32_t tot = __builtin_amdgcn_wave_reduce_add_u32(c, 0);
  if (lane == 0) vfirst[vp] = E[g] + tot;
}
// sub-row starts per tile (one wave per tile; the panels' streams are stored back to back in whole tiles, padding is zero)
static __global__ __launch_bounds__(256) void k_xp_tile_flags(const uint32_t* __restrict__ pcol, uint32_t ntiles, uint32_t* __restrict__ tflags) {
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6); g < ntiles; g += gridDim.x * 4) {
    const uint4 wd = *(const uint4*)(pcol + (size_t)g * WP_ENT + lane * 4);
    const uint32_t c = (wd.x >> 31) + (wd.y >> 31) + (wd.z >> 31) + (wd.w >> 31);
    const uint32_t tot = __builtin_amdgcn_wave_reduce_add_u32(c, 0);
    if (lane == 0) tflags[g] = tot;
  }
}
// number the sub-rows: sub-row s (panel after panel, in entry order) starts at the s-th flagged entry
static __global__ __launch_bounds__(256) void k_xp_subrows(const uint32_t* __restrict__ pcol, const uint32_t* __restrict__ rowtmp, const uint32_t* __restrict__ E, uint32_t ntiles,
                                                           uint32_t* __restrict__ trow, uint32_t* __restrict__ subrow_row) {
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6); g < ntiles; g += gridDim.x * 4) {
    const size_t q0 = (size_t)g * WP_ENT + lane * 4;
    const uint4 wd = *(const uint4*)(pcol + q0);
    const uint32_t f[4] = {wd.x >> 31, wd.y >> 31, wd.z >> 31, wd.w >> 31};
    const uint32_t mine = f[0] + f[1] + f[2] + f[3];
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if ((int)lane >= d) incl += o; }
    uint32_t s = E[g] + incl - mine;            // flagged entries before my first one
    if (lane == 0) trow[g] = E[g] - (f[0] ? 0u : 1u);
#pragma unroll
    for (int j = 0; j < 4; j++) if (f[j]) { const uint32_t r = rowtmp[q0 + j]; subrow_row[s] = r; s++; }
  }
}
static __global__ void k_xp_block_starts(const uint32_t* __restrict__ subrow_row, uint32_t nblocks, const uint32_t* __restrict__ vfirst, uint32_t* __restrict__ blockptr) {
  for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < (nblocks + 1) * XP; t += gridDim.x * 256) {
    const uint32_t b = t / XP, k = t % XP; const uint64_t target = (uint64_t)b * XP_RB;
    uint32_t lo = vfirst[k], hi = vfirst[k + 1];               // panel k's sub-rows; first one whose row is >= target
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (subrow_row[mid] < target) lo = mid + 1; else hi = mid; }
    blockptr[t] = lo;
  }
}
// ---- the 16-bit column plane: per tile the number of cold entries, after a scan the packed words and the extras ----
static __global__ __launch_bounds__(256) void k_xc_count(const uint32_t* __restrict__ pcol, uint32_t ntiles, uint32_t H, uint32_t* __restrict__ tcnt) {
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6); g < ntiles; g += gridDim.x * 4) {
    const uint4 wd = *(const uint4*)(pcol + (size_t)g * WP_ENT + lane * 4);
    const uint32_t c = ((wd.x >> XT_COLD_BIT) & 1u) + ((wd.y >> XT_COLD_BIT) & 1u) + ((wd.z >> XT_COLD_BIT) & 1u) + ((wd.w >> XT_COLD_BIT) & 1u);
    const uint32_t tot = __builtin_amdgcn_wave_reduce_add_u32(c, 0);
    if (lane == 0) tcnt[g] = tot;
  }
}
static __global__ __launch_bounds__(256) void k_xc_pack(const uint32_t* __restrict__ pcol, const uint32_t* __restrict__ trow, const uint32_t* __restrict__ tcold, uint32_t ntiles, uint32_t H,
                                                        uint16_t* __restrict__ col16, uint16_t* __restrict__ extras, uint32_t* __restrict__ tinfo) {
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6); g < ntiles; g += gridDim.x * 4) {
    const size_t q0 = (size_t)g * WP_ENT + lane * 4;
    const uint4 wd = *(const uint4*)(pcol + q0);
    const uint32_t w[4] = {wd.x, wd.y, wd.z, wd.w};
    uint32_t mine = 0; bool cold[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { cold[j] = (w[j] & XT_COLD) != 0; mine += cold[j] ? 1u : 0u; }
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if ((int)lane >= d) incl += o; }
    uint32_t x = tcold[g] + incl - mine;
    uint32_t o16[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t v = w[j] & XT_IDXMASK, flag = (w[j] >> 31) << 15;
      if (cold[j]) { const uint32_t c = v; o16[j] = flag | (H + (c >> 16)); extras[x++] = (uint16_t)(c & 0xFFFFu); }
      else o16[j] = flag | v;
    }
    *(uint2*)(col16 + q0) = make_uint2(o16[0] | (o16[1] << 16), o16[2] | (o16[3] << 16));
    if (lane == 0) { tinfo[2 * (size_t)g] = trow[g]; tinfo[2 * (size_t)g + 1] = tcold[g]; }
  }
}
// debugging aid (GRB_MI355X_XC_VERIFY=1 at plan build): decode every tile the way the tile pipeline does and compare with the 32-bit words
static __global__ __launch_bounds__(256) void k_xc_verify(const uint32_t* __restrict__ pcol, const uint16_t* __restrict__ col16, const uint16_t* __restrict__ extras, const uint32_t* __restrict__ tinfo,
                                                          const uint32_t* __restrict__ trow, uint32_t ntiles, uint32_t H, unsigned long long* __restrict__ bad) {
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t g = blockIdx.x * 4 + (threadIdx.x >> 6); g < ntiles; g += gridDim.x * 4) {
    const size_t q0 = (size_t)g * WP_ENT + lane * 4;
    const uint2 hh = *(const uint2*)(col16 + q0);
    const uint32_t h[2] = {hh.x, hh.y};
    uint32_t before = 0, nc = 0;
    for (int u = 0; u < 4; u++) {
      const uint32_t w = (h[u >> 1] >> (16 * (u & 1))) & 0x7FFFu; const bool cold = w >= H;
      const unsigned long long m = __ballot(cold);
      before += __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); nc += cold ? 1u : 0u;
    }
    const uint32_t xr = tinfo[2 * (size_t)g + 1] + before;
    uint32_t x0 = 0, x1 = 0, x2 = 0;
    if (nc) { const uint32_t* xp = (const uint32_t*)extras + (xr >> 1); x0 = xp[0]; x1 = xp[1]; x2 = xp[2]; }
    const uint64_t wa = ((uint64_t)x1 << 32) | x0, wb = ((uint64_t)x2 << 32) | x1;
    uint32_t q = xr & 1u;
    for (int u = 0; u < 4; u++) {
      const uint32_t w16 = (h[u >> 1] >> (16 * (u & 1))) & 0xFFFFu, code = w16 & 0x7FFFu, flag = (w16 & 0x8000u) << 16;
      const bool cold = code >= H;
      const uint32_t lo = (uint32_t)((q < 2u ? wa >> (16u * q) : wb >> (16u * (q - 2u)))) & 0xFFFFu;
      const uint32_t c = flag | (cold ? XT_COLD | (((code - H) << 16) | lo) : code);
      q += cold ? 1u : 0u;
      if (c != pcol[q0 + u]) { const unsigned long long n = atomicAdd(bad, 1ull); if (n == 0) { bad[1] = q0 + u; bad[2] = ((unsigned long long)c << 32) | pcol[q0 + u]; } }
    }
    if (lane == 0 && tinfo[2 * (size_t)g] != trow[g]) atomicAdd(bad + 3, 1ull);
  }
}
// xhot[k*H + h] = u[hot column h of panel k]: the eight LDS tables' contents, gathered once per call (every workgroup of a
// panel then loads its table with coalesced reads; gathering in each of the 32 workgroups cost 7-16 us of L2 traffic)
// (four slots per lane, the four gathers in flight together: one slot per lane — a chain of two dependent loads and nothing else — took 28.7 us for
//  the 2.5e6 slots of a 64-table plan)
template <class T> __global__ __launch_bounds__(256) void k_xp_hot_gather(const T* __restrict__ u, const uint32_t* __restrict__ hot_cols, uint32_t total, T* __restrict__ xhot) {
  for (uint32_t i = (blockIdx.x * 256 + threadIdx.x) * 4u; i < total; i += gridDim.x * 1024u) {
    if (i + 4u <= total) {
      const uint4 c = *(const uint4*)(hot_cols + i);
      const T x0 = u[c.x], x1 = u[c.y], x2 = u[c.z], x3 = u[c.w];
      xhot[i] = x0; xhot[i + 1] = x1; xhot[i + 2] = x2; xhot[i + 3] = x3;
    } else for (uint32_t j = i; j < total; j++) xhot[j] = u[hot_cols[j]];
  }
}
// y(i) = sum of the partials of row i's sub-rows, in panel order.  A workgroup owns XP_RB consecutive rows: their
// sub-rows are one contiguous run in each panel (sub-rows are in row order inside a panel), so the eight runs are read
// with coalesced loads — all eight issued before the first is used: the kernel is a chain of dependent round trips
// otherwise — and accumulated panel after panel in LDS: no per-row index chain, fixed order => reproducible.
// A row that crosses a chunk boundary inside a panel has consecutive sub-rows there.  The plan marks them in the row-id
// stream (XP_CONT: continues the sub-row before it, XP_HEAD: is followed by continuations), so that the common case
// costs nothing: the head adds its continuations, the continuations themselves do nothing.
constexpr uint32_t XP_CONT = 0x8000u, XP_HEAD = 0x4000u, XP_LROW = 0x07FFu;
static_assert(XP_RB <= XP_LROW + 1, "row id inside a block and the two flags share 16 bits");
static __global__ void k_xp_cont_flags(const uint32_t* __restrict__ subrow_row, const uint32_t* __restrict__ vfirst, uint16_t* __restrict__ lrow) {
  uint32_t pb[XP + 1];
#pragma unroll
  for (int k = 0; k <= XP; k++) pb[k] = vfirst[k];                // first sub-row of every panel (S = 1 only: virtual = physical)
  const uint32_t F = pb[XP];
  for (uint32_t s = blockIdx.x * 256 + threadIdx.x; s < F; s += gridDim.x * 256) {
    const uint32_t r = subrow_row[s];
    bool first = false, last = s + 1 == F;                        // first / last sub-row of its panel
#pragma unroll
    for (int k = 0; k <= XP; k++) { first = first || pb[k] == s; last = last || pb[k] == s + 1; }
    const bool cont = !first && subrow_row[s - 1] == r, more = !last && subrow_row[s + 1] == r;
    lrow[s] = (uint16_t)((r % XP_RB) | (cont ? XP_CONT : 0u) | (!cont && more ? XP_HEAD : 0u));
  }
}
// What the kernel costs is latency and LDS instruction issue, not bytes (0.115 GB): a row block is a chain of dependent
// round trips — bounds, data, then per panel an LDS write, a barrier and an LDS read-modify-write — and a CU holds only four
// of these chains at a time (threads and LDS).  So the chain is kept short: all eight runs are loaded before the first is
// used, the panels are folded two per barrier (even panels into one accumulator array, odd ones into another; y = even sum
// + odd sum, a fixed order), and the read-modify-write reads the flag and the sum together instead of one after the other.
// Measured on R-MAT-22 (40 us before): this version 38 us; without the LDS phases it would be 25 us shorter, without its
// stores 6 us, without its loads 13 us.  Rejected: accumulators that start at the monoid's identity plus stamped continuation
// slots (fewer LDS operations per sub-row, but 41 KB of LDS leave three workgroups per CU: 53 us); 320 threads per workgroup
// (46 us: the blocks of the low row ids of un-permuted R-MAT hold 1000+ sub-rows per panel and need more rounds); a
// persistent, software-pipelined version (buffer loads issued a block ahead, s_barrier behind lgkmcnt-only waits so that
// the prefetch survives the barriers: 94-129 us — the same skew, which a static split of the blocks cannot balance).
constexpr int XP_NA = 2;                      // accumulator arrays = panels folded per barrier
template <class T, class SR>
__global__ __launch_bounds__(XP_CT) void k_xp_combine(uint32_t nrows, const uint32_t* __restrict__ blockptr, const uint16_t* __restrict__ subrow_lrow, const T* __restrict__ partial,
                                                      T* __restrict__ y, uint8_t* __restrict__ ypres, const SR sr) {
  __shared__ T acc[XP_NA][XP_RB];
  __shared__ uint8_t has[XP_NA][XP_RB];
  __shared__ T cval[2][XP_NA][XP_CT];         // the partials of the continuation sub-rows of the current panels, by thread
  __shared__ uint8_t ccont[2][XP_NA][XP_CT + 1];   // ... and whether thread t holds one (slot XP_CT: always 0)
  const uint32_t b = blockIdx.x, r0 = b * XP_RB, tid = threadIdx.x;
  for (uint32_t i = tid; i < XP_NA * XP_RB; i += XP_CT) (&has[0][0])[i] = 0;
  if (tid < 2 * XP_NA) ccont[tid / XP_NA][tid % XP_NA][XP_CT] = 0;
  uint32_t lo[XP], hi[XP], longest = 0;
#pragma unroll
  for (int k = 0; k < XP; k++) { lo[k] = blockptr[b * XP + k]; hi[k] = blockptr[(b + 1) * XP + k]; longest = hi[k] - lo[k] > longest ? hi[k] - lo[k] : longest; }
  for (uint32_t base = 0; base < longest; base += XP_CT) {         // (one round unless the block holds more than XP_CT sub-rows of one panel)
    uint32_t w[XP]; T v[XP];
#pragma unroll
    for (int k = 0; k < XP; k++) {
      const uint32_t s = lo[k] + base + tid; const bool ok = s < hi[k];
      w[k] = ok ? (uint32_t)subrow_lrow[s] : XP_CONT; v[k] = ok ? partial[s] : T();
    }
#pragma unroll
    for (int k0 = 0; k0 < XP; k0 += XP_NA) {
      const int buf = (k0 / XP_NA) & 1;
#pragma unroll
      for (int j = 0; j < XP_NA; j++) {
        const int k = k0 + j;
        const bool cont = (w[k] & XP_CONT) != 0 && lo[k] + base + tid < hi[k];
        ccont[buf][j][tid] = cont ? 1 : 0;
        if (cont) cval[buf][j][tid] = v[k];
      }
      __syncthreads();                                              // (also orders these panels' updates of acc behind those of the panels before)
#pragma unroll
      for (int j = 0; j < XP_NA; j++) {
        const int k = k0 + j;
        if (!(w[k] & XP_CONT)) {                                    // one thread per row and panel: no two threads meet on a row of one array
          const uint32_t r = w[k] & XP_LROW; T x = v[k];
          const bool h = has[j][r] != 0; const T a = acc[j][r];     // (a is junk while h is false)
          if (w[k] & XP_HEAD) {                                     // my continuations sit in the threads after me, or (rarely) in the next round
            uint32_t q = tid + 1;
            while (ccont[buf][j][q]) { x = sr.add(x, cval[buf][j][q]); q++; }
            if (q == XP_CT) for (uint32_t g = lo[k] + base + XP_CT; g < hi[k] && (subrow_lrow[g] & XP_CONT); g++) x = sr.add(x, partial[g]);
          }
          acc[j][r] = h ? sr.add(a, x) : x; has[j][r] = 1;
        }
      }
    }
    __syncthreads();
  }
  __syncthreads();
  for (uint32_t i = tid; i < XP_RB; i += XP_CT) {
    const uint32_t r = r0 + i;
    if (r < nrows) {
      bool h = has[0][i] != 0; T x = acc[0][i];
#pragma unroll
      for (int j = 1; j < XP_NA; j++) { const bool hj = has[j][i] != 0; const T xj = acc[j][i]; x = h ? (hj ? sr.add(x, xj) : x) : xj; h = h || hj; }
      y[r] = h ? x : T(); ypres[r] = h ? 1 : 0;
    }
  }
}

// ---- the merge in row-major slot order --------------------------------------------------------------------------------------
// The plan numbers the sub-rows of a block of rows in row-major order (row, then panel, then position in a chain of
// continuations) and cuts the rows into blocks of <= XM_ROWS rows and about XM_TARGET sub-rows (weight of a row = its
// sub-rows + XM_TARGET / XM_ROWS, blocks = equal slices of the weight prefix): a workgroup scatters the partials of its
// block into LDS by slot — one LDS write per sub-row, no read-modify-write, no per-panel phase — and after ONE barrier every
// row adds its consecutive slots in order (fixed order => reproducible) and writes y.  Against k_xp_combine: 2 LDS operations
// per sub-row instead of ~7, one barrier instead of four, blocks of equal weight whatever the labels of the graph.
constexpr uint32_t XM_ROWS = 1024, XM_TARGET = 2048, XM_SLOTS = 2560;   // a block holds < XM_TARGET + (sub-rows of one row) sub-rows: XM_SLOTS of LDS, more (up to 64 KB) when a row has very many
constexpr int XM_CT = 256;
#ifndef XM_INFL
#define XM_INFL 4                     // pairs of loads in flight per thread of the merge kernel (8: no faster)
#endif
static __global__ void k_xm_iota(uint32_t* p, uint64_t n) { for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = (uint32_t)i; }
static __global__ void k_xm_weights(const uint32_t* __restrict__ cnt, uint32_t nrows, uint32_t perrow, uint32_t* __restrict__ w) {
  for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r <= nrows; r += gridDim.x * 256) w[r] = r < nrows ? cnt[r] + perrow : 0u;
}
static __global__ void k_xm_newblock(const uint32_t* __restrict__ P, uint32_t nrows, uint32_t target, uint32_t* __restrict__ f) {
  for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r <= nrows; r += gridDim.x * 256) f[r] = r < nrows && (r == 0 || P[r] / target != P[r - 1] / target) ? 1u : 0u;
}
// bid = exclusive scan of the flags + flag - 1 (the block of row r); first rows of the blocks
static __global__ void k_xm_bstart(const uint32_t* __restrict__ f, const uint32_t* __restrict__ fscan, uint32_t nrows, uint32_t nblocks, uint32_t* __restrict__ bstart) {
  for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r <= nrows; r += gridDim.x * 256) { if (r == nrows) bstart[nblocks] = nrows; else if (f[r]) bstart[fscan[r]] = r; }
}
static __global__ void k_xm_rowoff(const uint32_t* __restrict__ f, const uint32_t* __restrict__ fscan, const uint32_t* __restrict__ bstart, const uint32_t* __restrict__ rfirst, uint32_t nrows,
                                   uint16_t* __restrict__ rowoff) {
  for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < nrows; r += gridDim.x * 256) { const uint32_t b = fscan[r] + f[r] - 1; rowoff[r] = (uint16_t)(rfirst[r] - rfirst[bstart[b]]); }
}
// i-th sub-row in row-major order (sidx[i], of row skey[i]) -> its slot inside its block
static __global__ void k_xm_slots(const uint32_t* __restrict__ skey, const uint32_t* __restrict__ sidx, uint64_t F, const uint32_t* __restrict__ f, const uint32_t* __restrict__ fscan,
                                  const uint32_t* __restrict__ bstart, const uint32_t* __restrict__ rfirst, uint16_t* __restrict__ slot) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < F; i += gridDim.x * 256ull) { const uint32_t r = skey[i], b = fscan[r] + f[r] - 1; slot[sidx[i]] = (uint16_t)((uint32_t)i - rfirst[bstart[b]]); }
}
static __global__ void k_xm_block_starts(const uint32_t* __restrict__ subrow_row, const uint32_t* __restrict__ bstart, uint32_t nblocks, const uint32_t* __restrict__ vfirst, uint32_t np,
                                         uint32_t* __restrict__ blockptr) {
  for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < (nblocks + 1) * np; t += gridDim.x * 256) {
    const uint32_t b = t / np, k = t % np; const uint32_t target = bstart[b];
    uint32_t lo = vfirst[k], hi = vfirst[k + 1];               // virtual panel k's sub-rows (in row order): first one whose row is >= target
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (subrow_row[mid] < target) lo = mid + 1; else hi = mid; }
    blockptr[t] = lo;
  }
}
static __global__ void k_xm_max(const uint32_t* __restrict__ cnt, uint32_t nrows, uint32_t* __restrict__ out) {
  __shared__ uint32_t s_m;
  if (threadIdx.x == 0) s_m = 0;
  __syncthreads();
  uint32_t m = 0;
  for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < nrows; r += gridDim.x * 256) m = cnt[r] > m ? cnt[r] : m;
  m = __builtin_amdgcn_wave_reduce_max_u32(m, 0);
  if ((threadIdx.x & 63) == 0 && m) atomicMax(&s_m, m);
  __syncthreads();
  if (threadIdx.x == 0 && s_m) atomicMax(out, s_m);                  // one device atomic per workgroup
}
// EPI: 0 y = the row sums, ypres = "the row has entries";  1 y(r) = y(r) (+) sum where the row has entries (in place, ypres untouched: the
// accumulate of a product into a full vector with the monoid's operator);  2 y(r) = fill (+) sum / fill, ypres = 1 (the same into a vector
// whose pending `w(:) = fill` was never written: gap/prmark.py:21-23 `r[:] = teleport; r += A' (+).second w` is this one store)
template <class T, class SR, int EPI = 0>
__global__ __launch_bounds__(XM_CT) void k_xp_merge(uint32_t nrows, uint32_t nblocks, const uint32_t* __restrict__ bstart, const uint32_t* __restrict__ blockptr, const uint16_t* __restrict__ slot,
                                                    const uint16_t* __restrict__ rowoff, const T* __restrict__ partial, T* __restrict__ y, uint8_t* __restrict__ ypres, const SR sr, const T fill = T()) {
  extern __shared__ __attribute__((aligned(16))) unsigned char xm_lds[];      // m_slots values: XM_SLOTS unless some row has very many sub-rows
  T* const vals = (T*)xm_lds;
  // workgroup w runs on XCD w % 8 (observed, grb_spmv.hip): give every XCD a contiguous eighth of the row blocks, so that the 128-byte
  // lines two neighbouring blocks share (the ends of their runs of partials, slots, row offsets and y) meet in one L2
  // (the grid is the block count rounded up to a multiple of 8)
  const uint32_t per = gridDim.x / XP, b = (blockIdx.x & (XP - 1)) * per + (blockIdx.x >> 3);
  const uint32_t tid = threadIdx.x;
  if (b >= nblocks) return;
  const uint32_t r0 = bstart[b], r1 = bstart[b + 1];
  uint32_t lo[XP], pre[XP + 1];
  pre[0] = 0;
#pragma unroll
  for (int k = 0; k < XP; k++) { lo[k] = blockptr[b * XP + k]; pre[k + 1] = pre[k] + (blockptr[(b + 1) * XP + k] - lo[k]); }
  const uint32_t total = pre[XP];
  // the block's sub-rows as one index space over the eight runs; XM_INFL pairs of loads in flight per thread before the first LDS write
  for (uint32_t t0 = tid; t0 < total; t0 += XM_INFL * XM_CT) {
    T v[XM_INFL]; uint32_t sl[XM_INFL];
#pragma unroll
    for (int u = 0; u < XM_INFL; u++) {
      const uint32_t t = t0 + u * XM_CT; const bool ok = t < total;
      uint32_t s = 0;
#pragma unroll
      for (int k = 0; k < XP; k++) if (t >= pre[k] && t < pre[k + 1]) s = lo[k] + (t - pre[k]);
      v[u] = ok ? partial[s] : T(); sl[u] = ok ? (uint32_t)slot[s] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int u = 0; u < XM_INFL; u++) if (sl[u] != 0xFFFFFFFFu) vals[sl[u]] = v[u];
  }
  __syncthreads();
  const int lane = tid & 63;
  const uint32_t nr = r1 - r0, nround = (nr + XM_CT - 1) / XM_CT * XM_CT;
  for (uint32_t i = tid; i < nround; i += XM_CT) {
    const bool live = i < nr;
    const uint32_t r = r0 + i;
    const uint32_t o0 = live ? (uint32_t)rowoff[r] : 0u, o1 = live ? (i + 1 < nr ? (uint32_t)rowoff[r + 1] : total) : 0u;
    const bool wide = o1 - o0 > 64u;                                  // a row cut into many sub-rows (a dense row: 8 panels x its chunks): the whole wave adds it
    T acc = T();
    if (o1 > o0 && !wide) { acc = vals[o0]; for (uint32_t q = o0 + 1; q < o1; q++) acc = sr.add(acc, vals[q]); }
    unsigned long long todo = __ballot(wide);
    while (todo) {
      const int L = __builtin_ctzll(todo); todo &= todo - 1;
      const uint32_t q0 = (uint32_t)__shfl((int)o0, L, 64), q1 = (uint32_t)__shfl((int)o1, L, 64);
      T part = sr.identity; bool has = false;
      for (uint32_t q = q0 + lane; q < q1; q += 64) { part = has ? sr.add(part, vals[q]) : vals[q]; has = true; }      // lane-strided, then a fixed tree: reproducible
      const T red = wave_reduce_op<T, false>(sr.add_op(), has ? part : sr.identity);
      if (lane == L) acc = red;
    }
    if constexpr (EPI == 0) { if (live) { y[r] = acc; ypres[r] = o1 > o0 ? 1 : 0; } }
    else if constexpr (EPI == 1) { if (live && o1 > o0) y[r] = sr.add(y[r], acc); }
    else if constexpr (EPI == 2) { if (live) { y[r] = o1 > o0 ? sr.add(fill, acc) : fill; ypres[r] = 1; } }
    else {                                                            // EPI 3: SpmvCall::epi == 3 (y / ypres = the accumulated vector, fill = the threshold)
      if constexpr (std::is_arithmetic<T>::value) {
        if (live && o1 > o0 && (sr.add_op() == B_MIN ? acc < fill : acc > fill)) { if (ypres[r]) y[r] = sr.add(y[r], acc); else { y[r] = acc; ypres[r] = 1; } }
      }
    }
  }
}

// ---- the merge for plans with sub-panels (round 4): np = 16 ... 64 runs per block, rows of 1 ... np (and more) sub-rows ------------------
// What this kernel costs is INSTRUCTIONS (measured at R-MAT-25, a table per sub-panel, 1.08e8 sub-rows, 4.3e4 blocks: ~1080 VALU
// instructions per wave and block, the SIMDs busy, 400-560 us whether blocks are launched one per workgroup or walked by persistent
// workgroups, with the bounds of the next block prefetched or not) — so it is written to issue few:
//   * a thread takes FOUR consecutive positions of the block's index space per step (one 6-step branch-free bisection over the runs'
//     prefix lengths in LDS serves all four when they lie in one run — runs average 128 sub-rows), two steps in flight;
//   * the row sums do not go "one thread per row" (rows have 1 ... np ... sub-rows: a wave would wait for its longest row): the SLOTS
//     are dealt to the threads — thread t scans `chunk` consecutive slots front to back, branch-free: a flag per slot marks where a row
//     begins, the running sum restarts there and is written back to every slot, so a row's sum ends up in its LAST slot; what flows
//     into a chunk from the chunks before it (a row that began earlier) is added to the last slot of the chunk's leading piece after a
//     barrier, by walking back over the chunk totals.  Every order is fixed: reproducible.
//   * persistent workgroups (a few per CU) over an XCD's contiguous eighth of the blocks, the next block's bounds fetched a block ahead.
// (Also measured, no change: the run of a position from a ballot over lane-held prefixes + readlanes instead of the bisection's six dependent LDS
//  reads — 406 against 402 us: the kernel is not bound by that chain either; blocks of 2048 / 8192 sub-rows: 503 / 422 us against 341-400.)
// LDS: m_slots values + m_slots flag bytes.
#ifndef XMW_TARGET_V
#define XMW_TARGET_V 4096          // sub-rows per block of the wide merge (measurement builds: make XTFLAGS=-DXMW_TARGET_V=8192)
#endif
constexpr uint32_t XMW_TARGET = XMW_TARGET_V, XMW_ROWS = XMW_TARGET_V / 2;
constexpr int XMW_RPT = XMW_ROWS / XM_CT;
template <class T, class SR, int EPI = 0>
__global__ __launch_bounds__(XM_CT) void k_xp_merge_wide(uint32_t nrows, uint32_t nblocks, const uint32_t* __restrict__ bstart, const uint32_t* __restrict__ blockptr, const uint16_t* __restrict__ slot,
                                                         const uint16_t* __restrict__ rowoff, const T* __restrict__ partial, T* __restrict__ y, uint8_t* __restrict__ ypres, const SR sr, const T fill,
                                                         const uint32_t np, const uint32_t m_slots) {
  extern __shared__ __attribute__((aligned(16))) unsigned char xm_lds[];
  T* const vals = (T*)xm_lds;
  uint8_t* const flag = xm_lds + (size_t)m_slots * sizeof(T);          // (m_slots is a multiple of 256: the byte array starts 16-byte aligned)
  __shared__ uint32_t s_lo[XPMAX], s_pre[XPMAX + 1];                   // s_pre[k] = sub-rows of the block in the runs before run k; s_pre[np] = all of them; beyond np: 0xFFFFFFFF
  __shared__ T s_tsum[XM_CT];                                          // the running sum at the end of thread t's chunk (since its last row start, or over the whole chunk)
  __shared__ uint8_t s_hasflag[XM_CT];                                 // a row starts inside thread t's chunk
  const uint32_t per = (nblocks + XP - 1) / XP, tid = threadIdx.x, stride = gridDim.x >> 3;
  auto block_of = [&](uint32_t bb) -> uint32_t { const uint32_t b = (blockIdx.x & (XP - 1)) * per + bb; return bb < per && b < nblocks ? b : 0xFFFFFFFFu; };
  uint32_t bb = blockIdx.x >> 3, b = block_of(bb);
  uint32_t n_r0 = 0, n_r1 = 0, n_l0 = 0, n_len = 0;
  auto fetch_bounds = [&](uint32_t bn) __attribute__((always_inline)) {
    if (bn == 0xFFFFFFFFu) return;
    n_r0 = bstart[bn]; n_r1 = bstart[bn + 1];
    if (tid < np) { n_l0 = blockptr[bn * np + tid]; n_len = blockptr[(bn + 1) * np + tid] - n_l0; }
  };
  fetch_bounds(b);
  for (; b != 0xFFFFFFFFu; ) {
  __syncthreads();                                                     // (the LDS arrays of the block before are free)
  const uint32_t r0 = n_r0, r1 = n_r1, nr = r1 - r0;
  if (tid < 64u) {                                                     // the first wave: run lengths, their exclusive prefix by a wave scan
    const bool in = tid < np;
    const uint32_t l0 = in ? n_l0 : 0u, len = in ? n_len : 0u;
    uint32_t incl = len;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if ((int)tid >= d) incl += o; }
    s_lo[tid] = l0; s_pre[tid] = in ? incl - len : 0xFFFFFFFFu;
    if (tid == 63u) { if (np < 64u) s_pre[np] = incl; s_pre[XPMAX] = np < 64u ? 0xFFFFFFFFu : incl; }     // (lanes >= np add nothing: lane 63 holds the total)
  }
  for (uint32_t q = tid * 4u; q < m_slots; q += XM_CT * 4u) *(uint32_t*)(flag + q) = 0u;
  // the offsets of this thread's rows (rows tid, tid + 256, ...): loaded now, used after the partials are in LDS
  uint32_t o0[XMW_RPT], o1[XMW_RPT];
#pragma unroll
  for (int j = 0; j < XMW_RPT; j++) {
    const uint32_t i = tid + (uint32_t)j * XM_CT; const bool live = i < nr;
    o0[j] = live ? (uint32_t)rowoff[r0 + i] : 0u;
    o1[j] = live && i + 1 < nr ? (uint32_t)rowoff[r0 + i + 1] : 0u;   // (the block's last row ends at `total`: patched below)
  }
  __syncthreads();
  const uint32_t total = s_pre[np];
  bb += stride; const uint32_t b_next = block_of(bb);
  fetch_bounds(b_next);                                                // (in flight together with this block's partials; consumed at the top of the next round)
  // phase 1: the block's sub-rows as one index space over the np runs, four consecutive positions per thread and step, two steps in flight
  auto run_of = [&](uint32_t t) __attribute__((always_inline)) -> uint32_t {      // the last run k with s_pre[k] <= t (t < total)
    uint32_t k = 0;
#pragma unroll
    for (uint32_t step = 32u; step; step >>= 1) { const uint32_t cand = k + step; k = s_pre[cand] <= t ? cand : k; }
    return k;
  };
  for (uint32_t base = 0; base < total; base += 2u * XM_CT * 4u) {
    T v[2][4]; uint32_t sl[2][4];
#pragma unroll
    for (int g = 0; g < 2; g++) {
      const uint32_t t = base + (tid + (uint32_t)g * XM_CT) * 4u;
      // (the four positions mostly lie in one run; where they straddle runs the run index ADVANCES — a loop that only the straddling lanes
      //  enter, one or two short rounds.  A second bisection per element for those lanes made every wave pay for it: a wave's 256
      //  positions straddle a run boundary almost always — 67 VALU instructions per sub-row, half of them in that path.)
      uint32_t src[4]; bool ok[4];
      uint32_t k = t < total ? run_of(t) : 0u;
      uint32_t base = s_lo[k] - s_pre[k], end = s_pre[k + 1];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const uint32_t te = t + (uint32_t)e; ok[e] = te < total;
        while (ok[e] && te >= end) { k++; base = s_lo[k] - s_pre[k]; end = s_pre[k + 1]; }
        src[e] = ok[e] ? base + te : 0u;
      }
#pragma unroll
      for (int e = 0; e < 4; e++) { v[g][e] = ok[e] ? partial[src[e]] : T(); sl[g][e] = ok[e] ? (uint32_t)slot[src[e]] : 0xFFFFFFFFu; }
    }
#pragma unroll
    for (int g = 0; g < 2; g++)
#pragma unroll
      for (int e = 0; e < 4; e++) if (sl[g][e] != 0xFFFFFFFFu) vals[sl[g][e]] = v[g][e];
  }
  // ... and the row starts
#pragma unroll
  for (int j = 0; j < XMW_RPT; j++) {
    const uint32_t i = tid + (uint32_t)j * XM_CT;
    if (i < nr && i + 1 == nr) o1[j] = total;
    if (i < nr && o1[j] > o0[j]) flag[o0[j]] = 1;
  }
  __syncthreads();
  // phase 2: thread t scans the slots [q0, q1) of its chunk front to back (an odd chunk length: the threads' accesses fall in different banks)
  const uint32_t chunk = ((total + XM_CT - 1) / XM_CT) | 1u;
  const uint32_t q0 = tid * chunk < total ? tid * chunk : total, q1 = q0 + chunk < total ? q0 + chunk : total;
  uint32_t first = WP_NONE;                                            // the first row start inside my chunk
  {
    T acc = sr.identity;
    for (uint32_t q = q0; q < q1; q++) {
      const T v = vals[q]; const bool f = flag[q] != 0;
      acc = f ? v : sr.add(acc, v);
      vals[q] = acc;
      first = f && first == WP_NONE ? q : first;
    }
    s_tsum[tid] = acc; s_hasflag[tid] = first != WP_NONE ? 1 : 0;
  }
  __syncthreads();
  // a chunk that begins inside a row: its leading piece — up to its first row start, or all of it — continues a row of the chunks before.
  // If the row ENDS with that piece (the next slot starts a row, or the block ends), its last slot must hold the whole row: what the chunks
  // before hold of it (their totals, walking back to the chunk the row starts in) is added there.  If the row goes on, the chunk's total
  // carries it.
  {
    const uint32_t lead_end = first != WP_NONE ? first : q1;
    const bool ends_row = lead_end == total || flag[lead_end < total ? lead_end : 0u] != 0;
    if (q1 > q0 && first != q0 && ends_row) {
      T c = sr.identity;
      for (uint32_t t = tid; t > 0;) { t--; c = sr.add(s_tsum[t], c); if (s_hasflag[t]) break; }
      vals[lead_end - 1u] = sr.add(c, vals[lead_end - 1u]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < XMW_RPT; j++) {
    const uint32_t i = tid + (uint32_t)j * XM_CT;
    if (i < nr) {
      const uint32_t r = r0 + i;
      const bool has = o1[j] > o0[j]; const T acc = has ? vals[o1[j] - 1u] : T();
      if constexpr (EPI == 0) { y[r] = acc; ypres[r] = has ? 1 : 0; }
      else if constexpr (EPI == 1) { if (has) y[r] = sr.add(y[r], acc); }
      else if constexpr (EPI == 2) { y[r] = has ? sr.add(fill, acc) : fill; ypres[r] = 1; }
      else {                                                          // EPI 3: SpmvCall::epi == 3
        if constexpr (std::is_arithmetic<T>::value) {
          if (has && (sr.add_op() == B_MIN ? acc < fill : acc > fill)) { if (ypres[r]) y[r] = sr.add(y[r], acc); else { y[r] = acc; ypres[r] = 1; } }
        }
      }
    }
  }
  b = b_next;
  }     // blocks of this workgroup
}

// which instantiation of the tile pipeline runs.  The product uses the defaults; GRB_MI355X_XT=d<depth>w<waves>[e<exp>] picks
// one of the others in builds with -DXT_VARIANTS (measurement harness, FP64 static semirings only).
struct XtVariant { int depth, waves, exp; };
inline XtVariant xt_variant() {
  XtVariant v{XT_DEPTH, XT_WAVES, 0};
#ifdef XT_VARIANTS
  const char* e = getenv("GRB_MI355X_XT");
  if (e) { int d = v.depth, w = v.waves, x = 0; if (sscanf(e, "d%dw%de%d", &d, &w, &x) >= 2) { v.depth = d; v.waves = w; v.exp = x; } }
#endif
  return v;
}

// sub-panels per XCD (each with its own LDS table).  GRB_MI355X_XS forces a value (1, 2, 4, 8).  Otherwise by the operand's length, from
// the R-MAT measurements of round 4 (profiles/r04_subpanels.txt; ms per product S = 1 -> best S): 4-byte pattern product 2^23 columns
// 0.294 -> 0.275 (S = 2), 2^24 0.702 -> 0.583 (4), 2^25 1.46 -> 0.95 (4; 8: 0.83 with a merge twice as long); 8-byte PLUS_TIMES 2^23
// 0.557 -> 0.570 (worse), 2^24 1.270 -> 1.173 (4).  Below those sizes an XCD's share of the operand fits its L2 and the table covers most
// entries: sub-panels only add sub-rows (R-MAT-22: +8 %).  The candidate is then CHECKED against the matrix (build_xcd_plan): it is taken
// only if the larger tables would serve >= 8 % more of the entries — a matrix without popular columns gains nothing from them.
template <class T> int xp_subpanels(uint64_t ncols, bool* forced_out) {
  const uint32_t forced = wp_env("GRB_MI355X_XS", 0);
  *forced_out = forced == 1 || forced == 2 || forced == 4 || forced == 8;
  if (*forced_out) return (int)forced;
  if (sizeof(T) <= 4 && ncols >= (1ull << 25)) return 8;       // (R-MAT-25 FP32 PageRank: 1.54 ms per iteration with S = 4, 1.45 with S = 8)
  if (ncols >= (1ull << 24)) return 4;
  if (sizeof(T) <= 4 && ncols >= (1ull << 23)) return 2;
  return 1;
}
static __global__ void k_xp_neg(const uint32_t* __restrict__ cnt, uint32_t n, uint32_t* __restrict__ key) {
  for (uint32_t c = blockIdx.x * 256 + threadIdx.x; c < n; c += gridDim.x * 256) key[c] = 0xFFFFFFFFu - cnt[c];
}
static __global__ void k_xp_unneg(uint32_t* __restrict__ key, uint32_t n) {
  for (uint32_t c = blockIdx.x * 256 + threadIdx.x; c < n; c += gridDim.x * 256) key[c] = 0xFFFFFFFFu - key[c];
}
// ---- the narrow value plane (round 6): an integer matrix whose values all fit 16 signed bits keeps them as int16 ---------------------------------
template <class T> static __global__ void k_xt_fits16(const T* __restrict__ v, uint64_t n, uint32_t* __restrict__ flag) {
  int bad = 0;
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
    const T x = v[i];
    if constexpr (std::is_signed<T>::value) bad |= (x < (T)-32768 || x > (T)32767) ? 1 : 0; else bad |= x > (T)32767 ? 1 : 0;
  }
  if (__syncthreads_or(bad) && threadIdx.x == 0) atomicOr(flag, 1u);
}
template <class T> static __global__ void k_xt_narrow16(const T* __restrict__ v, uint64_t n, int16_t* __restrict__ out) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) out[i] = (int16_t)v[i];
}

template <class T> void build_xcd_plan(DevCSR& M, int ncu, bool with_vals, int force_S = 0) {
  auto grid_n = [](uint64_t n) { uint64_t b = (n + 255) / 256; if (b < 1) b = 1; if (b > 8192) b = 8192; return (unsigned)b; };
  hipEvent_t ev0, ev1; GRB_HIP(hipEventCreate(&ev0)); GRB_HIP(hipEventCreate(&ev1)); GRB_HIP(hipEventRecord(ev0, stream()));
  auto* P = new XcdPlan(); M.xcd.reset(P);
  const uint32_t n = M.ncols; const uint64_t nnz = M.nnz;
  constexpr uint32_t H = xt_hot<T>::H;
  const bool sell = xp_sell_wanted<T>();
  // columns a table serves: the lane-per-piece kernel wants the table's last slot free (it holds zero bits: what a cold entry "reads" there)
  const uint32_t HOT = sell && (uint32_t)xt_hot<T>::HOT == H ? H - 1 : (uint32_t)xt_hot<T>::HOT;
  bool forced = false;
  int S = force_S ? force_S : xp_subpanels<T>(n, &forced);
  // 1. column counts; the 128-byte lines of u dealt to the (virtual) panels (equal entry counts); every stream's columns ranked by frequency
  const uint32_t line = 128 / (uint32_t)sizeof(T), nlines = (n + line - 1) / line;
  uint32_t lshift = 0; while ((1u << lshift) < line) lshift++;
  DevBuf cnt((size_t)n * 4 + 4), code((size_t)n * 4 + 4), cstart((XPMAX + 1) * 4), pol((size_t)nlines + 8);
  GRB_HIP(hipMemsetAsync(cnt.p, 0, (size_t)n * 4 + 4, stream()));
  { DevBuf sorted(nnz * 4 + 4), first((size_t)n * 4 + 4);
    int cb = 1; while ((1ull << cb) < (unsigned long long)n) cb++;
    sort_keys_u32(M.col.as<uint32_t>(), sorted.as<uint32_t>(), nnz, cb);
    hipLaunchKernelGGL(k_xp_run_starts, dim3(grid_n(nnz)), dim3(256), 0, stream(), sorted.as<uint32_t>(), nnz, first.as<uint32_t>());
    hipLaunchKernelGGL(k_xp_run_lengths, dim3(grid_n(nnz)), dim3(256), 0, stream(), sorted.as<uint32_t>(), nnz, first.as<uint32_t>(), cnt.as<uint32_t>()); }
  if (S > 1 && !forced && !force_S) {
    // would S times the table slots serve noticeably more entries?  The counts in descending order, their running sum at 8 H and at 8 S H
    // columns (the deal balances the panels, so a stream's H hottest columns are about the matrix's 8 S H hottest ones)
    DevBuf k0((size_t)n * 4 + 4), k1((size_t)n * 4 + 4), ps((size_t)n * 4 + 4);
    hipLaunchKernelGGL(k_xp_neg, dim3(grid_n(n)), dim3(256), 0, stream(), cnt.as<uint32_t>(), n, k0.as<uint32_t>());
    sort_keys_u32(k0.as<uint32_t>(), k1.as<uint32_t>(), n, 32);
    hipLaunchKernelGGL(k_xp_unneg, dim3(grid_n(n)), dim3(256), 0, stream(), k1.as<uint32_t>(), n);
    exclusive_scan_u32(k1.as<uint32_t>(), ps.as<uint32_t>(), n);
    const uint64_t a1 = std::min<uint64_t>((uint64_t)XP * H, n - 1), aS = std::min<uint64_t>((uint64_t)XP * S * H, n - 1);
    uint32_t c1 = 0, cS = 0;
    GRB_HIP(hipMemcpyAsync(&c1, ps.as<uint32_t>() + a1, 4, hipMemcpyDeviceToHost, stream()));
    GRB_HIP(hipMemcpyAsync(&cS, ps.as<uint32_t>() + aS, 4, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
    if ((double)(cS - c1) < 0.08 * (double)nnz) S = 1;
    if (getenv("GRB_MI355X_VERBOSE")) fprintf(stderr, "[grb] xcd plan: tables serve %.1f %% of the entries with one per XCD, %.1f %% with %d per XCD -> S = %d\n", 100.0 * c1 / nnz, 100.0 * cS / nnz, (int)(aS / ((uint64_t)XP * H)), S);
  }
  const uint32_t NP = (uint32_t)(XP * S);
  const bool own = S > 1 && wp_env("GRB_MI355X_XOWN", 1) != 0;      // a table per sub-panel (0: one per XCD, the entries it serves ride with their row's cold ones)
  const uint32_t NS = own ? NP : (uint32_t)XP, vps = NP / NS;       // streams, virtual panels per stream
  P->S = S; P->NP = (int)NP; P->NS = (int)NS; P->own = own;
  P->hot_cols.alloc((size_t)NS * H * 4 + 4);
  GRB_HIP(hipMemsetAsync(P->hot_cols.p, 0, (size_t)NS * H * 4 + 4, stream()));
  {
    DevBuf negw((size_t)nlines * 4 + 4), lid((size_t)nlines * 4 + 4), negw2((size_t)nlines * 4 + 4), lsorted((size_t)nlines * 4 + 4);
    hipLaunchKernelGGL(k_xp_line_weights, dim3(grid_n(nlines)), dim3(256), 0, stream(), cnt.as<uint32_t>(), n, line, nlines, negw.as<uint32_t>(), lid.as<uint32_t>());
    sort_pairs_u32(negw.as<uint32_t>(), negw2.as<uint32_t>(), lid.as<uint32_t>(), lsorted.as<uint32_t>(), nlines, 32);
    const uint32_t ntop = nlines < 4096u ? nlines : 4096u;
    DevBuf dtop((size_t)ntop + 8);
    hipLaunchKernelGGL(k_xp_lpt, dim3(1), dim3(64), 0, stream(), negw2.as<uint32_t>(), ntop, NP, (uint32_t)S, (uint8_t*)dtop.p);
    hipLaunchKernelGGL(k_xp_deal_lines, dim3(grid_n(nlines)), dim3(256), 0, stream(), lsorted.as<uint32_t>(), nlines, (const uint8_t*)dtop.p, ntop, NP, (uint32_t)S, (uint8_t*)pol.p);
    DevBuf k32((size_t)n * 4 + 4), k32o((size_t)n * 4 + 4), cin((size_t)n * 4 + 4), cout((size_t)n * 4 + 4);
    hipLaunchKernelGGL(k_xp_column_keys, dim3(grid_n(n)), dim3(256), 0, stream(), cnt.as<uint32_t>(), n, line, (const uint8_t*)pol.p, vps, k32.as<uint32_t>(), cin.as<uint32_t>());
    int kbits = 20; while ((1u << (kbits - 20)) < NS) kbits++;                 // stream id above the 20 count bits
    sort_pairs_u32(k32.as<uint32_t>(), k32o.as<uint32_t>(), cin.as<uint32_t>(), cout.as<uint32_t>(), n, kbits);
    GRB_HIP(hipMemsetAsync(cstart.p, 0xFF, (XPMAX + 1) * 4, stream()));
    hipLaunchKernelGGL(k_xp_panel_starts, dim3(grid_n(n)), dim3(256), 0, stream(), k32o.as<uint32_t>(), n, cstart.as<uint32_t>());
    hipLaunchKernelGGL(k_xp_fix_starts, dim3(1), dim3(1), 0, stream(), cstart.as<uint32_t>(), n, NS);
    hipLaunchKernelGGL(k_xp_column_codes, dim3(grid_n(n)), dim3(256), 0, stream(), k32o.as<uint32_t>(), cout.as<uint32_t>(), n, H, HOT, cstart.as<uint32_t>(), code.as<uint32_t>(), P->hot_cols.as<uint32_t>());
    if (wp_env("GRB_MI355X_XHOT_BY_COLUMN", 1) != 0) {
      const uint64_t tot = (uint64_t)NS * H;
      DevBuf k64(tot * 8 + 8), k64o(tot * 8 + 8), v32(tot * 4 + 4), v32o(tot * 4 + 4);
      hipLaunchKernelGGL(k_xp_hot_keys, dim3(grid_n(tot)), dim3(256), 0, stream(), P->hot_cols.as<uint32_t>(), cstart.as<uint32_t>(), NS, (uint32_t)H, HOT, (unsigned long long*)k64.p, v32.as<uint32_t>());
      sort_pairs_u64((const uint64_t*)k64.p, (uint64_t*)k64o.p, v32.as<uint32_t>(), v32o.as<uint32_t>(), tot, 32 + 7);
      hipLaunchKernelGGL(k_xp_hot_reslot, dim3(grid_n(tot)), dim3(256), 0, stream(), (const unsigned long long*)k64o.p, NS, (uint32_t)H, code.as<uint32_t>(), P->hot_cols.as<uint32_t>());
    }
  }   // (the temporaries return to the pool; reuse is stream-ordered)
  // 2. row of every entry
  DevBuf rowidx(nnz * 4 + 4);
  GRB_HIP(hipMemsetAsync(rowidx.p, 0, nnz * 4 + 4, stream()));
  hipLaunchKernelGGL(k_xp_mark_rows, dim3(grid_n(M.nrows)), dim3(256), 0, stream(), M.rowptr.as<uint32_t>(), M.nrows, rowidx.as<uint32_t>());
  inclusive_scan_max_u32(rowidx.as<uint32_t>(), rowidx.as<uint32_t>(), nnz);
  // 2b. sub-panels: in which virtual panels every row has cold entries (the entries the LDS tables serve then ride with them)
  DevBuf rowmask;
  if (S > 1 && !own) {
    rowmask.alloc(((size_t)M.nrows + 1) * 8);
    GRB_HIP(hipMemsetAsync(rowmask.p, 0, ((size_t)M.nrows + 1) * 8, stream()));
    hipLaunchKernelGGL(k_xp_rowmask, dim3(grid_n(nnz)), dim3(256), 0, stream(), M.col.as<uint32_t>(), rowidx.as<uint32_t>(), code.as<uint32_t>(), (const uint8_t*)pol.p, nnz, H, lshift,
                       (unsigned long long*)rowmask.p);
  }
  // 3. counting sweep, scans, the sizes (first host round trip)
  const uint32_t nunits = (uint32_t)((nnz + XP_UNIT - 1) / XP_UNIT);
  const size_t nslots = (size_t)NP * nunits;
  DevBuf ne((nslots + 1) * 4), escan((nslots + 1) * 4), lastkey(nslots * 8 + 8), carry(nslots * 8 + 8), picked(2 * (XPMAX + 1) * 4);
  XpSweep<T> sw{};
  sw.col = M.col.as<uint32_t>(); sw.rowidx = rowidx.as<uint32_t>(); sw.code = code.as<uint32_t>(); sw.val = with_vals ? M.val.as<T>() : nullptr; sw.nnz = nnz; sw.nunits = nunits;
  sw.np = NP; sw.S = (uint32_t)S; sw.H = H; sw.lshift = lshift; sw.own = own ? 1u : 0u; sw.pol = (const uint8_t*)pol.p; sw.rowmask = (const unsigned long long*)rowmask.p;
  sw.ne = ne.as<uint32_t>(); sw.lastkey = (unsigned long long*)lastkey.p; sw.escan = escan.as<uint32_t>(); sw.carry = (const unsigned long long*)carry.p;
  GRB_HIP(hipMemsetAsync(ne.as<uint32_t>() + nslots, 0, 4, stream()));
  hipLaunchKernelGGL((k_xp_sweep<T, false>), dim3(nunits), dim3(XP_ST), 0, stream(), sw);
  exclusive_scan_u32(ne.as<uint32_t>(), escan.as<uint32_t>(), nslots + 1);
  exclusive_scan_max_u64((const uint64_t*)lastkey.p, (uint64_t*)carry.p, nslots);
  hipLaunchKernelGGL(k_xp_pick, dim3(1), dim3(128), 0, stream(), escan.as<uint32_t>(), nunits, NP, cstart.as<uint32_t>(), NS, picked.as<uint32_t>());
  uint32_t hp[2 * (XPMAX + 1)];
  GRB_HIP(hipMemcpyAsync(hp, picked.p, sizeof(hp), hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
  const uint32_t wpp = (uint32_t)(ncu / XP) * xt_variant().waves;       // waves per panel
  uint32_t kt[XPMAX]; uint64_t nchunks_total = 0;
  XpStarts vstart{};                                                     // where every virtual panel begins in the store
  P->tbase[0] = 0;
  for (uint32_t k = 0; k < NS; k++) {
    P->ne[k] = hp[(k + 1) * vps] - hp[k * vps];                          // a stream = its virtual panels, one after the other
    P->ntiles[k] = (uint32_t)((P->ne[k] + WP_ENT - 1) / WP_ENT);
    P->tbase[k + 1] = P->tbase[k] + P->ntiles[k];
    const uint32_t nk = hp[XPMAX + 1 + k + 1] - hp[XPMAX + 1 + k]; P->nhot[k] = nk < HOT ? nk : HOT;
    kt[k] = wp_chunk_tasks(P->ntiles[k], wpp);
    nchunks_total += (P->ntiles[k] + kt[k] - 1) / kt[k];
    for (uint32_t sp = 0; sp < vps; sp++) {
      const uint32_t vp = k * vps + sp;
      sw.vpoff[vp] = hp[vp] - hp[k * vps]; sw.ebase[vp] = (uint64_t)P->tbase[k] * WP_ENT + sw.vpoff[vp]; sw.chunk_entries[vp] = kt[k] * (uint32_t)WP_ENT;
      vstart.v[vp] = sw.ebase[vp];
    }
  }
  vstart.v[NP] = (uint64_t)P->tbase[NS] * WP_ENT;
  const uint32_t ntiles = P->tbase[NS];
  const size_t nstore = (size_t)ntiles * WP_ENT + 64;
  // 4. scattering sweep
  DevBuf rowtmp(nstore * 4);
  P->pcol.alloc(nstore * 4);
  if (with_vals) { P->pval.alloc(nstore * sizeof(T)); GRB_HIP(hipMemsetAsync(P->pval.p, 0, nstore * sizeof(T), stream())); }      // (zeroed: the padding behind a panel's end takes part in the range check of the narrow plane)
  GRB_HIP(hipMemsetAsync(P->pcol.p, 0, nstore * 4, stream()));
  sw.pcol = P->pcol.as<uint32_t>(); sw.pval = with_vals ? P->pval.as<T>() : nullptr; sw.rowtmp = rowtmp.as<uint32_t>();
  hipLaunchKernelGGL((k_xp_sweep<T, true>), dim3(nunits), dim3(XP_ST), 0, stream(), sw);
  // 4b. (lane-per-piece layout) the sub-rows cut into pieces of <= SELL_CAP entries / wave slices: more row-start flags, before anything is numbered
  SellStreams sst{}; sst.ns = NS;
  for (uint32_t k = 0; k <= NS; k++) sst.tbase[k] = P->tbase[k];
  for (uint32_t k = 0; k < NS; k++) sst.ne[k] = P->ne[k];
  const unsigned tile_grid = [&] { unsigned nb = (ntiles + 3) / 4; if (nb < 1) nb = 1; if (nb > 16384) nb = 16384; return nb; }();
  if (sell) {
    DevBuf tf0(((size_t)ntiles + 1) * 4), E0(((size_t)ntiles + 1) * 4);
    GRB_HIP(hipMemsetAsync(tf0.as<uint32_t>() + ntiles, 0, 4, stream()));
    hipLaunchKernelGGL(k_xp_tile_flags, dim3(tile_grid), dim3(256), 0, stream(), P->pcol.as<uint32_t>(), ntiles, tf0.as<uint32_t>());
    exclusive_scan_u32(tf0.as<uint32_t>(), E0.as<uint32_t>(), (uint64_t)ntiles + 1);
    uint32_t F0 = 0;
    GRB_HIP(hipMemcpyAsync(&F0, E0.as<uint32_t>() + ntiles, 4, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
    DevBuf ps0((size_t)F0 * 4 + 8), pl0((size_t)F0 * 4 + 8);
    hipLaunchKernelGGL(k_sell_pstart, dim3(tile_grid), dim3(256), 0, stream(), P->pcol.as<uint32_t>(), E0.as<uint32_t>(), ntiles, ps0.as<uint32_t>());
    hipLaunchKernelGGL(k_sell_plen, dim3(grid_n(F0)), dim3(256), 0, stream(), ps0.as<uint32_t>(), (uint64_t)F0, sst, pl0.as<uint32_t>());
    hipLaunchKernelGGL(k_sell_cap, dim3(tile_grid), dim3(256), 0, stream(), P->pcol.as<uint32_t>(), rowtmp.as<uint32_t>(), E0.as<uint32_t>(), ps0.as<uint32_t>(), pl0.as<uint32_t>(), sst, ntiles);
  }
  // 5. sub-rows: starts per tile, scan, the total (second host round trip), then their numbering
  DevBuf tflags(((size_t)ntiles + 1) * 4), E(((size_t)ntiles + 1) * 4);
  GRB_HIP(hipMemsetAsync(tflags.as<uint32_t>() + ntiles, 0, 4, stream()));
  { unsigned nb = (ntiles + 3) / 4; if (nb < 1) nb = 1; if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(k_xp_tile_flags, dim3(nb), dim3(256), 0, stream(), P->pcol.as<uint32_t>(), ntiles, tflags.as<uint32_t>()); }
  exclusive_scan_u32(tflags.as<uint32_t>(), E.as<uint32_t>(), (uint64_t)ntiles + 1);
  uint32_t F32 = 0;
  GRB_HIP(hipMemcpyAsync(&F32, E.as<uint32_t>() + ntiles, 4, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
  P->F = F32;
  DevBuf subrow_row(P->F * 4 + 4);
  P->trow.alloc(((size_t)ntiles + 1) * 4); P->lrow.alloc(P->F * 2 + 4);
  { unsigned nb = (ntiles + 3) / 4; if (nb < 1) nb = 1; if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(k_xp_subrows, dim3(nb), dim3(256), 0, stream(), P->pcol.as<uint32_t>(), rowtmp.as<uint32_t>(), E.as<uint32_t>(), ntiles, P->trow.as<uint32_t>(),
                       subrow_row.as<uint32_t>()); }
  P->vfirst.alloc((XPMAX + 2) * 4);
  hipLaunchKernelGGL(k_xp_vp_first, dim3(NP + 1), dim3(64), 0, stream(), P->pcol.as<uint32_t>(), E.as<uint32_t>(), vstart, NP, ntiles, P->vfirst.as<uint32_t>());
  // 6. row blocks of the per-panel merge kernel (the fallback; eight runs: S = 1 only): where each block's run of sub-rows starts in every panel
  const uint32_t nblocks = (uint32_t)(((uint64_t)M.nrows + XP_RB - 1) / XP_RB);
  if (S == 1) {
    hipLaunchKernelGGL(k_xp_cont_flags, dim3(grid_n(P->F)), dim3(256), 0, stream(), subrow_row.as<uint32_t>(), P->vfirst.as<uint32_t>(), P->lrow.as<uint16_t>());
    P->blockptr.alloc(((size_t)nblocks + 1) * XP * 4 + 4);
    hipLaunchKernelGGL(k_xp_block_starts, dim3(grid_n(((uint64_t)nblocks + 1) * XP)), dim3(256), 0, stream(), subrow_row.as<uint32_t>(), nblocks, P->vfirst.as<uint32_t>(), P->blockptr.as<uint32_t>());
  }
  // 6b. the merge in row-major slot order: sub-rows sorted by row (stable: panel, then chain order inside a row), rows per
  //     sub-row count, variable row blocks, slots and row offsets inside the blocks (third host round trip: the block count)
  {
    const uint32_t nr = M.nrows; const uint64_t F = P->F;
    int rb = 1; while ((1ull << rb) < (unsigned long long)nr) rb++;
    DevBuf sidx0(F * 4 + 4), sidx(F * 4 + 4), skey(F * 4 + 4), rcnt(((size_t)nr + 1) * 4 + 4), first(((size_t)nr + 1) * 4 + 4), rfirst(((size_t)nr + 1) * 4 + 4), w(((size_t)nr + 1) * 4 + 4),
           Pw(((size_t)nr + 1) * 4 + 4), nf(((size_t)nr + 1) * 4 + 4), nfs(((size_t)nr + 1) * 4 + 4), dmax(16);
    hipLaunchKernelGGL(k_xm_iota, dim3(grid_n(F)), dim3(256), 0, stream(), sidx0.as<uint32_t>(), F);
    sort_pairs_u32(subrow_row.as<uint32_t>(), skey.as<uint32_t>(), sidx0.as<uint32_t>(), sidx.as<uint32_t>(), F, rb);
    GRB_HIP(hipMemsetAsync(rcnt.p, 0, ((size_t)nr + 1) * 4, stream())); GRB_HIP(hipMemsetAsync(dmax.p, 0, 16, stream()));
    hipLaunchKernelGGL(k_xp_run_starts, dim3(grid_n(F)), dim3(256), 0, stream(), skey.as<uint32_t>(), F, first.as<uint32_t>());
    hipLaunchKernelGGL(k_xp_run_lengths, dim3(grid_n(F)), dim3(256), 0, stream(), skey.as<uint32_t>(), F, first.as<uint32_t>(), rcnt.as<uint32_t>());
    exclusive_scan_u32(rcnt.as<uint32_t>(), rfirst.as<uint32_t>(), (uint64_t)nr + 1);                 // row-major position of every row's first sub-row
    // block size: XM_TARGET sub-rows in <= XM_ROWS rows for eight runs (k_xp_merge), XMW_TARGET in <= XMW_ROWS with sub-panels (k_xp_merge_wide)
    P->m_wide = NP > (uint32_t)XP || wp_env("GRB_MI355X_XM_WIDE", 0) != 0;
    const uint32_t m_target = !P->m_wide ? XM_TARGET : XMW_TARGET;
    const uint32_t m_perrow = !P->m_wide ? XM_TARGET / XM_ROWS : XMW_TARGET / XMW_ROWS;       // weight of a row beside its sub-rows (caps the rows of a block at target / perrow)
    hipLaunchKernelGGL(k_xm_weights, dim3(grid_n((uint64_t)nr + 1)), dim3(256), 0, stream(), rcnt.as<uint32_t>(), nr, m_perrow, w.as<uint32_t>());
    exclusive_scan_u32(w.as<uint32_t>(), Pw.as<uint32_t>(), (uint64_t)nr + 1);
    hipLaunchKernelGGL(k_xm_newblock, dim3(grid_n((uint64_t)nr + 1)), dim3(256), 0, stream(), Pw.as<uint32_t>(), nr, m_target, nf.as<uint32_t>());
    exclusive_scan_u32(nf.as<uint32_t>(), nfs.as<uint32_t>(), (uint64_t)nr + 1);
    hipLaunchKernelGGL(k_xm_max, dim3(grid_n(nr) > 1024u ? 1024u : grid_n(nr)), dim3(256), 0, stream(), rcnt.as<uint32_t>(), nr, dmax.as<uint32_t>());
    uint32_t hnb = 0, hmax = 0;
    GRB_HIP(hipMemcpyAsync(&hnb, nfs.as<uint32_t>() + nr, 4, hipMemcpyDeviceToHost, stream()));
    GRB_HIP(hipMemcpyAsync(&hmax, dmax.p, 4, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
    const uint64_t need = (uint64_t)m_target + hmax + m_perrow;
    P->m_slots = need <= XM_SLOTS ? XM_SLOTS : (uint32_t)((need + 255) / 256 * 256);
    P->m_nblocks = hnb; P->m_ok = hnb > 0 && (uint64_t)P->m_slots * (sizeof(T) + (P->m_wide ? 1 : 0)) <= 64u * 1024u && P->m_slots < 65536u && F < 0xFFFFFFF0ull;      // (16-bit slots; 64 KB of LDS at most)
    if (P->m_ok) {
      P->m_bstart.alloc(((size_t)hnb + 1) * 4 + 4); P->m_blockptr.alloc(((size_t)hnb + 1) * NP * 4 + 4); P->m_slot.alloc(F * 2 + 4); P->m_rowoff.alloc((size_t)nr * 2 + 4);
      hipLaunchKernelGGL(k_xm_bstart, dim3(grid_n((uint64_t)nr + 1)), dim3(256), 0, stream(), nf.as<uint32_t>(), nfs.as<uint32_t>(), nr, hnb, P->m_bstart.as<uint32_t>());
      hipLaunchKernelGGL(k_xm_rowoff, dim3(grid_n(nr)), dim3(256), 0, stream(), nf.as<uint32_t>(), nfs.as<uint32_t>(), P->m_bstart.as<uint32_t>(), rfirst.as<uint32_t>(), nr, P->m_rowoff.as<uint16_t>());
      hipLaunchKernelGGL(k_xm_slots, dim3(grid_n(F)), dim3(256), 0, stream(), skey.as<uint32_t>(), sidx.as<uint32_t>(), F, nf.as<uint32_t>(), nfs.as<uint32_t>(), P->m_bstart.as<uint32_t>(),
                         rfirst.as<uint32_t>(), P->m_slot.as<uint16_t>());
      hipLaunchKernelGGL(k_xm_block_starts, dim3(grid_n(((uint64_t)hnb + 1) * NP)), dim3(256), 0, stream(), subrow_row.as<uint32_t>(), P->m_bstart.as<uint32_t>(), hnb, P->vfirst.as<uint32_t>(), NP,
                         P->m_blockptr.as<uint32_t>());
    }
  }
  if (S > 1 && !P->m_ok) {        // a row with more sub-rows than a merge block holds: only the eight-run fallback can add it — start over without sub-panels
    GRB_HIP(hipStreamSynchronize(stream())); (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
    build_xcd_plan<T>(M, ncu, with_vals, 1);
    return;
  }
  // 6b'. the lane-per-piece layout (grb_spmv_sell.hpp): pieces sorted by length inside windows, dealt to chunks of 64 lanes, entries scattered to
  //      (chunk, step, lane); two host round trips (the groups' sizes, the streams' step counts)
  uint32_t sell_hs[XPMAX + 2] = {0}, sell_ib[XPMAX + 1] = {0};      // first step / first work item of every stream
  if (sell && P->m_ok && P->F < 0xFFFFFF00ull && ntiles) {
    const uint64_t F = P->F;
    DevBuf pstart(F * 4 + 8), plen(F * 4 + 8), key(F * 8 + 8), keyo(F * 8 + 8), id0(F * 4 + 8), sorted(F * 4 + 8), inv(F * 4 + 8), gstart((2 * XPMAX + 2) * 4);
    hipLaunchKernelGGL(k_sell_pstart, dim3(tile_grid), dim3(256), 0, stream(), P->pcol.as<uint32_t>(), E.as<uint32_t>(), ntiles, pstart.as<uint32_t>());
    hipLaunchKernelGGL(k_sell_plen, dim3(grid_n(F)), dim3(256), 0, stream(), pstart.as<uint32_t>(), F, sst, plen.as<uint32_t>());
    hipLaunchKernelGGL(k_sell_keys, dim3(grid_n(F)), dim3(256), 0, stream(), pstart.as<uint32_t>(), plen.as<uint32_t>(), F, sst, P->vfirst.as<uint32_t>(), vps, (unsigned long long*)key.p, id0.as<uint32_t>());
    sort_pairs_u64((const uint64_t*)key.p, (uint64_t*)keyo.p, id0.as<uint32_t>(), sorted.as<uint32_t>(), F, 47);
    GRB_HIP(hipMemsetAsync(gstart.p, 0xFF, (2 * XPMAX + 2) * 4, stream()));
    hipLaunchKernelGGL(k_sell_groups, dim3(grid_n(F)), dim3(256), 0, stream(), (const unsigned long long*)keyo.p, F, gstart.as<uint32_t>());
    hipLaunchKernelGGL(k_xp_fix_starts, dim3(1), dim3(1), 0, stream(), gstart.as<uint32_t>(), (uint32_t)F, 2 * NS);
    uint32_t hg[2 * XPMAX + 2];
    GRB_HIP(hipMemcpyAsync(hg, gstart.p, (2 * NS + 1) * 4, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
    SellGroups gr{}; gr.ns = NS; uint32_t nchunks = 0;
    for (uint32_t k = 0; k < NS; k++) {
      gr.g1[k] = hg[2 * k]; gr.g0[k] = hg[2 * k + 1]; gr.gend[k] = hg[2 * k + 2];
      gr.nc1[k] = (gr.g0[k] - gr.g1[k] + 63u) / 64u; gr.cbase[k] = nchunks; nchunks += gr.nc1[k] + (gr.gend[k] - gr.g0[k]);
    }
    gr.cbase[NS] = nchunks;
    DevBuf lc(((size_t)nchunks + 1) * 4), cstep(((size_t)nchunks + 1) * 4), pk((XPMAX + 2) * 4);
    GRB_HIP(hipMemsetAsync(lc.as<uint32_t>() + nchunks, 0, 4, stream()));
    { unsigned nb = (nchunks + 3) / 4; if (nb < 1) nb = 1; if (nb > 16384) nb = 16384;
      hipLaunchKernelGGL(k_sell_lc, dim3(nb), dim3(256), 0, stream(), sorted.as<uint32_t>(), plen.as<uint32_t>(), gr, nchunks, lc.as<uint32_t>()); }
    exclusive_scan_u32(lc.as<uint32_t>(), cstep.as<uint32_t>(), (uint64_t)nchunks + 1);
    hipLaunchKernelGGL(k_sell_pick, dim3(1), dim3(128), 0, stream(), cstep.as<uint32_t>(), gr, pk.as<uint32_t>());
    uint32_t hs[XPMAX + 2];
    GRB_HIP(hipMemcpyAsync(hs, pk.p, (NS + 1) * 4, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
    const uint64_t nsteps = hs[NS];
    if (nsteps && nsteps * 256ull < (1ull << 40)) {
      P->s_nsteps = nsteps; P->s_nchunks = nchunks;
      P->s_col.alloc(nsteps * 256 + 64); P->s_perm.alloc((size_t)nchunks * 256 + 64);
      GRB_HIP(hipMemsetAsync(P->s_col.p, 0, nsteps * 256 + 64, stream())); GRB_HIP(hipMemsetAsync(P->s_perm.p, 0xFF, (size_t)nchunks * 256 + 64, stream()));
      if (with_vals) { P->s_val.alloc(nsteps * 64 * sizeof(T) + 64); GRB_HIP(hipMemsetAsync(P->s_val.p, 0, nsteps * 64 * sizeof(T) + 64, stream())); }
      hipLaunchKernelGGL(k_sell_inv, dim3(grid_n(F)), dim3(256), 0, stream(), sorted.as<uint32_t>(), F, inv.as<uint32_t>());
      hipLaunchKernelGGL((k_sell_scatter<T>), dim3(tile_grid), dim3(256), 0, stream(), P->pcol.as<uint32_t>(), with_vals ? P->pval.as<T>() : (const T*)nullptr, E.as<uint32_t>(), pstart.as<uint32_t>(),
                         inv.as<uint32_t>(), lc.as<uint32_t>(), cstep.as<uint32_t>(), sst, gr, ntiles, P->s_col.as<uint32_t>(), with_vals ? P->s_val.as<T>() : (T*)nullptr, P->s_perm.as<uint32_t>());
      { unsigned nb = (nchunks + 3) / 4; if (nb < 1) nb = 1; if (nb > 16384) nb = 16384;
        hipLaunchKernelGGL(k_sell_meta, dim3(nb), dim3(256), 0, stream(), P->s_col.as<uint32_t>(), lc.as<uint32_t>(), cstep.as<uint32_t>(), gr, nchunks); }
      uint32_t ibase[XPMAX + 1], nitems = 0;
      for (uint32_t k = 0; k < NS; k++) { ibase[k] = nitems; nitems += (hs[k + 1] - hs[k] + SELL_ITEM - 1) / SELL_ITEM; }
      ibase[NS] = nitems; P->s_nitems = nitems;
      DevBuf dib((XPMAX + 1) * 4);
      GRB_HIP(hipMemcpyAsync(dib.p, ibase, (NS + 1) * 4, hipMemcpyHostToDevice, stream()));
      P->s_items.alloc(((size_t)nitems + 2) * 4);
      hipLaunchKernelGGL(k_sell_items, dim3(grid_n((uint64_t)nitems + 1)), dim3(256), 0, stream(), cstep.as<uint32_t>(), gr, dib.as<uint32_t>(), nitems, (uint32_t)nsteps, P->s_items.as<uint32_t>());
      memcpy(sell_hs, hs, sizeof(uint32_t) * (NS + 1)); memcpy(sell_ib, ibase, sizeof(uint32_t) * (NS + 1));
      GRB_HIP(hipStreamSynchronize(stream()));      // (ibase lives on this frame)
      P->sell = true;
      if (getenv("GRB_MI355X_VERBOSE")) fprintf(stderr, "[grb] lane-per-piece layout: %llu pieces in %u chunks, %llu steps = %.3f x the entries, %u work items\n", (unsigned long long)F, nchunks,
                                                (unsigned long long)nsteps, (double)nsteps * 64.0 / (double)nnz, nitems);
    }
  }
  // 6c. the 16-bit column plane (fourth host round trip: the number of cold entries); the 32-bit words are dropped
  constexpr bool c16 = xt_fmt<T>::C16;
  const bool keep32 = !c16 || wp_env("GRB_MI355X_XT_KEEP32", 0) != 0;
  if (c16) {
    DevBuf tcnt(((size_t)ntiles + 1) * 4), tcold(((size_t)ntiles + 1) * 4);
    GRB_HIP(hipMemsetAsync(tcnt.as<uint32_t>() + ntiles, 0, 4, stream()));
    unsigned nb = (ntiles + 3) / 4; if (nb < 1) nb = 1; if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(k_xc_count, dim3(nb), dim3(256), 0, stream(), P->pcol.as<uint32_t>(), ntiles, H, tcnt.as<uint32_t>());
    exclusive_scan_u32(tcnt.as<uint32_t>(), tcold.as<uint32_t>(), (uint64_t)ntiles + 1);
    uint32_t nc32 = 0;
    GRB_HIP(hipMemcpyAsync(&nc32, tcold.as<uint32_t>() + ntiles, 4, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
    P->ncold = nc32;
    P->col16.alloc(nstore * 2); P->extras.alloc(P->ncold * 2 + 32); P->tinfo.alloc(((size_t)ntiles + 1) * 8);
    hipLaunchKernelGGL(k_xc_pack, dim3(nb), dim3(256), 0, stream(), P->pcol.as<uint32_t>(), P->trow.as<uint32_t>(), tcold.as<uint32_t>(), ntiles, H, P->col16.as<uint16_t>(), P->extras.as<uint16_t>(),
                       P->tinfo.as<uint32_t>());
    if (wp_env("GRB_MI355X_XC_VERIFY", 0)) {
      DevBuf bad(64); GRB_HIP(hipMemsetAsync(bad.p, 0, 64, stream()));
      hipLaunchKernelGGL(k_xc_verify, dim3(nb), dim3(256), 0, stream(), P->pcol.as<uint32_t>(), P->col16.as<uint16_t>(), P->extras.as<uint16_t>(), P->tinfo.as<uint32_t>(), P->trow.as<uint32_t>(), ntiles, H,
                         (unsigned long long*)bad.p);
      unsigned long long hb[4]; GRB_HIP(hipMemcpyAsync(hb, bad.p, 32, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
      fprintf(stderr, "[grb] 16-bit column plane check: %llu words differ (first at entry %llu: decoded %08llx, stored %08llx), %llu tile infos differ; cold entries %llu of %llu\n", hb[0], hb[1], hb[2] >> 32,
              hb[2] & 0xFFFFFFFFull, hb[3], (unsigned long long)P->ncold, (unsigned long long)nstore);
    }
    if (!keep32) { P->pcol.reset(); P->trow.reset(); }
  }
  // 6c. the narrow value plane: integer values that all fit int16 are kept as int16 (k_spmv_tiles<..., VB = 2>); GRB_MI355X_XT_NARROW=0: never
  P->vbytes = (int)sizeof(T);
  if constexpr (std::is_integral<T>::value && (sizeof(T) == 4 || sizeof(T) == 8)) {
    if (with_vals && !P->sell && wp_env("GRB_MI355X_XT_NARROW", 1) != 0) {
      DevBuf flag(64); GRB_HIP(hipMemsetAsync(flag.p, 0, 64, stream()));
      const unsigned g = (unsigned)std::min<uint64_t>((nstore + 255) / 256, 4096);
      hipLaunchKernelGGL((k_xt_fits16<T>), dim3(g), dim3(256), 0, stream(), P->pval.as<T>(), (uint64_t)nstore, flag.as<uint32_t>());
      uint32_t hf = 1; GRB_HIP(hipMemcpyAsync(&hf, flag.p, 4, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
      if (hf == 0) {
        DevBuf narrow(nstore * 2 + 64);
        hipLaunchKernelGGL((k_xt_narrow16<T>), dim3(g), dim3(256), 0, stream(), P->pval.as<T>(), (uint64_t)nstore, narrow.as<int16_t>());
        P->pval = std::move(narrow); P->vbytes = 2;
      }
    }
  }
  // 7. the panels' argument block and the per-call buffers
  P->args.alloc(XPMAX * sizeof(XtPanel<T>));
  P->xhot.alloc((size_t)NS * H * sizeof(T) + 8); P->partial.alloc(P->F * sizeof(T) + 8);
  XtPanel<T> ha[XPMAX];
  memset((void*)ha, 0, sizeof(ha));
  for (uint32_t k = 0; k < NS; k++) {
    XtPanel<T>& a = ha[k];
    a.pcol = keep32 ? P->pcol.as<uint32_t>() + (size_t)P->tbase[k] * WP_ENT : nullptr; a.aval = with_vals ? (const T*)((const char*)P->pval.p + (size_t)P->tbase[k] * WP_ENT * (size_t)P->vbytes) : nullptr;
    a.trow = keep32 ? P->trow.as<uint32_t>() + P->tbase[k] : nullptr; a.xhot = P->xhot.as<T>() + (size_t)k * H;
    a.col16 = c16 ? P->col16.as<uint16_t>() + (size_t)P->tbase[k] * WP_ENT : nullptr; a.tinfo = c16 ? P->tinfo.as<uint32_t>() + 2 * (size_t)P->tbase[k] : nullptr;
    a.extras = c16 ? P->extras.as<uint16_t>() : nullptr; a.nextras = P->ncold;
    a.nnz = (uint32_t)P->ne[k]; a.ntiles = P->ntiles[k]; a.tiles_per_chunk = kt[k]; a.nhot = P->nhot[k];
    a.static_pct = wp_env("GRB_MI355X_WP_STATIC", WP_STATIC_PCT);
    a.interleave = wp_env("GRB_MI355X_XT_INTERLEAVE", vps > 1 ? 1u : 0u);    // sub-panels inside a stream: the waves of the XCD must walk it front to back together
  }
  if (getenv("GRB_MI355X_VERBOSE"))
    for (uint32_t k = 0; k < NS; k++)
      fprintf(stderr, "[grb] xcd plan stream %u: entries %llu tiles %u chunk %u hot %u (sub-rows in all %llu, chunks %llu, sub-panels per XCD %d, %s)\n", k, (unsigned long long)P->ne[k], P->ntiles[k], kt[k],
              P->nhot[k], (unsigned long long)P->F, (unsigned long long)nchunks_total, S, own ? "a table per sub-panel" : "a table per XCD");
  GRB_HIP(hipMemcpyAsync(P->args.p, ha, sizeof(ha), hipMemcpyHostToDevice, stream()));
  SellPanel<T> sa[XPMAX]; memset((void*)sa, 0, sizeof(sa));
  if (P->sell) {
    for (uint32_t k = 0; k < NS; k++) {
      SellPanel<T>& a = sa[k];
      a.scol = P->s_col.as<uint32_t>() + (size_t)sell_hs[k] * 64; a.sval = with_vals ? P->s_val.as<T>() + (size_t)sell_hs[k] * 64 : nullptr;
      a.item_first = P->s_items.as<uint32_t>() + sell_ib[k]; a.step0 = sell_hs[k]; a.nsteps = sell_hs[k + 1] - sell_hs[k]; a.nitems = sell_ib[k + 1] - sell_ib[k];
      a.xhot = P->xhot.as<T>() + (size_t)k * H; a.nhot = P->nhot[k];
      a.static_pct = wp_env("GRB_MI355X_WP_STATIC", WP_STATIC_PCT); a.interleave = wp_env("GRB_MI355X_XT_INTERLEAVE", vps > 1 ? 1u : 0u);
    }
    P->s_args.alloc(sizeof(sa));
    GRB_HIP(hipMemcpyAsync(P->s_args.p, sa, sizeof(sa), hipMemcpyHostToDevice, stream()));      // (the stream is synchronised below, before this frame goes)
  }
  P->tsize = (int)sizeof(T); P->has_vals = with_vals;
  GRB_HIP(hipEventRecord(ev1, stream()));
  GRB_HIP(hipStreamSynchronize(stream()));
  GRB_HIP(hipEventElapsedTime(&P->build_ms, ev0, ev1)); g_xcd_plan_build_ms = P->build_ms;
  (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
}

template <class T> bool run_xcd(const SpmvCall& c, const SemiringDesc& d, int ncu) {
  DevCSR& M = *c.M;
  if (ncu < XP || ncu % XP) return false;
  if ((uint64_t)M.ncols > xt_hot<T>::MAXCOLS) return false;        // a column must fit a code word (16-bit plane: 2^29 columns and more go to kernel W)
  const bool need_vals = c.aval != nullptr;
  if (need_vals && c.aval != M.val.p) return false;   // the plan's panel-major values are a copy of the stored ones (no typecast)
  auto* P = static_cast<XcdPlan*>(M.xcd.get());
  if (!P || P->tsize != (int)sizeof(T) || (need_vals && !P->has_vals)) { build_xcd_plan<T>(M, ncu, need_vals); P = static_cast<XcdPlan*>(M.xcd.get()); }
  const bool uses_u = d.flip ? binop_uses_x(d.mulop) : binop_uses_y(d.mulop);
  constexpr uint32_t H = xt_hot<T>::H;
  XtCall<T> call{(const T*)c.uval, M.ncols, (uint32_t)(P->NS / XP), P->partial.as<T>()};
  if (uses_u) hipLaunchKernelGGL((k_xp_hot_gather<T>), dim3(((uint32_t)P->NS * H + 1023) / 1024), dim3(256), 0, stream(), (const T*)c.uval, P->hot_cols.as<uint32_t>(), (uint32_t)P->NS * H, P->xhot.as<T>());
  with_semiring<T>(d, [&](auto sr) {
    typedef decltype(sr) SR;
    bool launched = false;
#ifdef XT_VARIANTS
    if constexpr (std::is_same<SR, StaticSR<double, B_PLUS, B_TIMES>>::value) {
      const XtVariant v = xt_variant();
#define XT_TRY(D_, W_, E_) if (!launched && v.depth == D_ && v.waves == W_ && v.exp == E_) { \
        hipLaunchKernelGGL((k_spmv_tiles<T, SR, D_, W_, E_>), dim3(ncu), dim3(W_ * 64), 0, stream(), call, (const XtPanel<T>*)P->args.p, sr); launched = true; }
      XT_TRY(1, 16, 1) XT_TRY(1, 16, 2) XT_TRY(2, 16, 0)
      if (!launched && v.exp == 4 && ((const XcdPlan*)P)->pcol.p) {      // e4: the 32-bit column words (plans built under GRB_MI355X_XT_KEEP32=1)
        hipLaunchKernelGGL((k_spmv_tiles<T, SR, XT_DEPTH, XT_WAVES, 0, false>), dim3(ncu), dim3(XT_WAVES * 64), 0, stream(), call, (const XtPanel<T>*)P->args.p, sr); launched = true; } XT_TRY(1, 8, 0) XT_TRY(2, 8, 0) XT_TRY(3, 8, 0) XT_TRY(2, 8, 2) XT_TRY(3, 8, 2) XT_TRY(4, 8, 0)
#undef XT_TRY
    }
#endif
    if constexpr (sizeof(T) == 4 || sizeof(T) == 8) {
      const bool sell_off = wp_env("GRB_MI355X_SELL_RUN", 1) == 0;            // measurement hook: the tile pipeline on a plan that carries both layouts
      if (!launched && P->sell && !sell_off) {
        hipLaunchKernelGGL((k_spmv_sell<T, SR>), dim3(ncu), dim3(XT_WAVES * 64), 0, stream(), call, (const SellPanel<T>*)P->s_args.p, P->s_perm.as<uint32_t>(), P->s_nchunks, sr);
        launched = true;
      }
    }
    if constexpr (std::is_integral<T>::value && (sizeof(T) == 4 || sizeof(T) == 8)) {
      if (!launched && P->vbytes == 2) {
        hipLaunchKernelGGL((k_spmv_tiles<T, SR, XT_DEPTH, XT_WAVES, 0, xt_fmt<T>::C16, 2>), dim3(ncu), dim3(XT_WAVES * 64), 0, stream(), call, (const XtPanel<T>*)P->args.p, sr);
        launched = true;
      }
    }
    if (!launched) hipLaunchKernelGGL((k_spmv_tiles<T, SR>), dim3(ncu), dim3(XT_WAVES * 64), 0, stream(), call, (const XtPanel<T>*)P->args.p, sr);
    const uint32_t nblocks = (uint32_t)(((uint64_t)M.nrows + XP_RB - 1) / XP_RB);
    static const bool old_merge = wp_env("GRB_MI355X_XP_OLD_MERGE", 0) != 0;       // measurement hook: the per-panel merge kernel
    if (P->m_ok && (!old_merge || P->m_wide)) {
      const dim3 mg((P->m_nblocks + XP - 1) / XP * XP), mb(XM_CT); const size_t ml = (size_t)P->m_slots * sizeof(T);
      const uint32_t np = (uint32_t)P->NP;
      static const uint32_t wg_per_cu = wp_env("GRB_MI355X_XMW_WGS", 6);
      const dim3 mgw(std::min<uint32_t>(mg.x, (uint32_t)ncu * wg_per_cu / XP * XP));             // the wide merge is persistent
#define XM_LAUNCH(EPI_, Y_, YP_, FILL_) { \
        if (P->m_wide) hipLaunchKernelGGL((k_xp_merge_wide<T, SR, EPI_>), mgw, mb, ml + P->m_slots, stream(), M.nrows, P->m_nblocks, P->m_bstart.as<uint32_t>(), P->m_blockptr.as<uint32_t>(), \
                                         P->m_slot.as<uint16_t>(), P->m_rowoff.as<uint16_t>(), P->partial.as<T>(), (Y_), (YP_), sr, (FILL_), np, P->m_slots); \
        else hipLaunchKernelGGL((k_xp_merge<T, SR, EPI_>), mg, mb, ml, stream(), M.nrows, P->m_nblocks, P->m_bstart.as<uint32_t>(), P->m_blockptr.as<uint32_t>(), P->m_slot.as<uint16_t>(), \
                                P->m_rowoff.as<uint16_t>(), P->partial.as<T>(), (Y_), (YP_), sr, (FILL_)); }
      if (c.epi == 1 && c.epi_done) {
        XM_LAUNCH(1, (T*)c.epi_w, (uint8_t*)nullptr, T())
        *c.epi_done = true;
      } else if (c.epi == 2 && c.epi_done) {
        T fill; memcpy(&fill, c.epi_fill, sizeof(T));
        XM_LAUNCH(2, (T*)c.tval, c.tpres, fill)
        *c.epi_done = true;
      } else if (c.epi == 3 && c.epi_done && std::is_arithmetic<T>::value) {
        T th; memcpy(&th, c.epi_fill, sizeof(T));
        XM_LAUNCH(3, (T*)c.epi_w, c.epi_wpres, th)
        *c.epi_done = true;
      } else XM_LAUNCH(0, (T*)c.tval, c.tpres, T())
#undef XM_LAUNCH
    } else
      hipLaunchKernelGGL((k_xp_combine<T, SR>), dim3(nblocks), dim3(XP_CT), 0, stream(), M.nrows, P->blockptr.as<uint32_t>(), P->lrow.as<uint16_t>(), P->partial.as<T>(),
                         (T*)c.tval, c.tpres, sr);
    g_last_plan += std::string("k_spmv_xcd<") + (sr.is_static ? "static" : "dynamic") + (P->sell ? ",lane-per-piece" : "") + (P->vbytes == 2 && P->has_vals ? ",values=int16" : "") + ",subrows=" + std::to_string(P->F) + (P->S > 1 ? ",subpanels=" + std::to_string(P->S) + (P->own ? "/own-tables" : "") : std::string()) + "," + xcd_mapping() + "> ";
  });
  return true;
}

}  // namespace grb
