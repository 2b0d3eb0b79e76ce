// grb_spmv_xcd.hpp — SpMV kernel "X": kernel W run on eight column panels at once, one panel per XCD.
//
// Why (profiles/spmv_pmc_traffic.json): kernel W's time is its memory-side read traffic, and 60 % of that traffic is
// 128-byte line fills for gathers of u that miss the 4 MiB per-XCD L2 — every XCD sees all 32 MiB of u, so the eight
// L2s cache eight copies of the same hot 4 MiB.  Kernel X gives each XCD its own eighth of u:
//   * u is cut into 128-byte lines; every line (16 FP64 columns) belongs to one panel.  Lines are dealt to the panels so
//     that the panels hold the same number of entries (heaviest lines first to the lightest panel, then in snake order);
//   * the plan stores the matrix panel-major: for each panel a CSR over its non-empty *sub-rows* (row fragments), the
//     column words and the values in that order.  A column word is the slot in the panel's LDS table for the panel's
//     H most frequent columns, H + column for the others (bit 31: first entry of a sub-row);
//   * a small kernel gathers the contents of the eight LDS tables from u once per call (through the panels' hot-column
//     lists); workgroup b (observed to run on XCD b % 8) runs the tile pipeline (grb_spmv_tiles.hpp) on panel b & 7:
//     table for the hot columns, every other gather reads u itself and touches only the panel's lines, which its XCD's L2
//     keeps — so the aggregate 32 MiB of L2 holds u once and u is never copied or re-ordered per call;
//   * each sub-row's sum goes to a partial array; a merge kernel (2048 rows per workgroup, the eight contiguous runs of
//     their partials accumulated panel after panel in LDS) adds the <= 8 partials of every row in a fixed order
//     (=> reproducible) and writes y.
// Extra algorithmic cost: one partial (8 B written + read) and one index per sub-row (~7.7 M at R-MAT-22, ~0.2 GB)
// against ~1.3 GB of avoided line fills.  Placement is used for speed only: any other block->XCD mapping is still correct.
#pragma once
#include "grb_spmv_tiles.hpp"
#include "grb_matops.hpp"

namespace grb {

constexpr int XP = 8;      // panels = XCDs

struct XcdPlan {          // lives in DevCSR::xcd (type-erased), built once per matrix and value type
  DevBuf hot_cols;        // u32[8*H]    column held by slot h of panel k's LDS table
  DevBuf pcol, pval;      // u32[.], T[.] panel-major entries (column word, value); panel k starts at ebase[k]
  DevBuf rowptr;          // u32[F + XP] per-panel sub-row pointers, relative to the panel's first entry (F_k + 1 each)
  DevBuf tasks;           // u32 per panel: first sub-row of every 256-entry tile, (ntiles_k + 1) each
  DevBuf subrow_row, blockptr;     // u32[F]: row of every sub-row (panel-major, ascending inside a panel); u32[(nblocks+1)*8]: first sub-row of panel k in row block b
  DevBuf subrow_lrow;              // u16[F]: the same row relative to its block of XP_RB rows — what the merge kernel streams (2 bytes per sub-row instead of 4)
  DevBuf args;            // WpArgs<T>[XP] in HBM; never changes between calls (u and the output arrive as kernel arguments)
  DevBuf carry;           // WpCarry<T>, one per chunk of tasks, panel after panel
  DevBuf xhot, partial, scratch;   // per-call work buffers kept with the plan so the argument block never changes (xhot: T[8*H], the LDS tables' contents)
  uint64_t eoff[XP + 1], soff[XP + 1], toff[XP + 1];
  uint64_t ebase[XP + 1];   // where a panel's entries are stored (eoff padded to 64-entry boundaries: aligned 16-byte loads)
  uint32_t ntasks[XP]; uint32_t maxchunks = 1; uint32_t nhot[XP]; uint64_t F = 0; int tsize = 0;
};

// weight of a line of u = entries in its columns
static __global__ void k_xp_line_weights(const uint32_t* __restrict__ cnt, uint32_t n, uint32_t line, uint32_t nlines, uint32_t* __restrict__ negw, uint32_t* __restrict__ id) {
  for (uint32_t l = blockIdx.x * 256 + threadIdx.x; l < nlines; l += gridDim.x * 256) {
    uint64_t w = 0; for (uint32_t j = 0; j < line; j++) { const uint64_t c = (uint64_t)l * line + j; if (c < n) w += cnt[c]; }
    negw[l] = 0xFFFFFFFFu - (uint32_t)(w > 0xFFFFFFFEull ? 0xFFFFFFFEull : w); id[l] = l;
  }
}
// lines in descending weight: the first `ntop` take the panel the host balanced for them, the others are dealt in snake order
static __global__ void k_xp_deal_lines(const uint32_t* __restrict__ sorted_line, uint32_t nlines, const uint8_t* __restrict__ top_panel, uint32_t ntop, uint8_t* __restrict__ panel_of_line) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < nlines; i += gridDim.x * 256) {
    uint32_t k;
    if (i < ntop) k = top_panel[i]; else { const uint32_t m = (i - ntop) % (2 * XP); k = m < XP ? m : 2 * XP - 1 - m; }
    panel_of_line[sorted_line[i]] = (uint8_t)k;
  }
}
// key = panel | descending count | column: a panel's columns in frequency order (ties by index)
static __global__ void k_xp_column_keys(const uint32_t* __restrict__ cnt, uint32_t n, uint32_t line, const uint8_t* __restrict__ panel_of_line, unsigned long long* __restrict__ key, uint32_t* __restrict__ colv) {
  for (uint32_t c = blockIdx.x * 256 + threadIdx.x; c < n; c += gridDim.x * 256) {
    key[c] = ((unsigned long long)panel_of_line[c / line] << 60) | ((unsigned long long)(0xFFFFFFFFu - cnt[c]) << 28) | c;
    colv[c] = c;
  }
}
static __global__ void k_xp_panel_starts(const unsigned long long* __restrict__ key, uint32_t n, uint32_t* __restrict__ start) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t k = (uint32_t)(key[i] >> 60);
    if (i == 0 || (uint32_t)(key[i - 1] >> 60) != k) start[k] = i;
  }
}
// code word of a column: (slot or H + column) << 3 | panel; the panel's hot-column list
static __global__ void k_xp_column_codes(const unsigned long long* __restrict__ key, const uint32_t* __restrict__ cols, uint32_t n, uint32_t H, uint32_t s0, uint32_t s1, uint32_t s2,
                                         uint32_t s3, uint32_t s4, uint32_t s5, uint32_t s6, uint32_t s7, uint32_t* __restrict__ code, uint32_t* __restrict__ hot_cols) {
  const uint32_t st[XP] = {s0, s1, s2, s3, s4, s5, s6, s7};
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t k = (uint32_t)(key[i] >> 60), c = cols[i], local = i - st[k];
    code[c] = ((local < H ? local : H + c) << 3) | k;
    if (local < H) hot_cols[(size_t)k * H + local] = c;
  }
}
static __global__ void k_xp_panel_keys(const uint32_t* __restrict__ col, uint64_t nnz, const uint32_t* __restrict__ rank, uint32_t* __restrict__ key, uint32_t* __restrict__ idx) {
  for (uint64_t p = blockIdx.x * 256ull + threadIdx.x; p < nnz; p += gridDim.x * 256ull) { key[p] = rank[col[p]] & 7u; idx[p] = (uint32_t)p; }
}
static __global__ void k_xp_hist8(const uint32_t* __restrict__ key, uint64_t nnz, unsigned long long* __restrict__ cnt) {
  __shared__ unsigned int s[XP];
  if (threadIdx.x < XP) s[threadIdx.x] = 0;
  __syncthreads();
  for (uint64_t p = blockIdx.x * 256ull + threadIdx.x; p < nnz; p += gridDim.x * 256ull) atomicAdd(&s[key[p]], 1u);
  __syncthreads();
  if (threadIdx.x < XP && s[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], (unsigned long long)s[threadIdx.x]);
}
template <class T> __global__ void k_xp_gather_entries(const uint32_t* __restrict__ perm, uint64_t nnz, const uint32_t* __restrict__ col, const T* __restrict__ val,
                                                       const uint32_t* __restrict__ rank, const uint32_t* __restrict__ rowidx,
                                                       uint32_t* __restrict__ pcol, T* __restrict__ pval, uint32_t* __restrict__ prow,
                                                       uint64_t e1, uint64_t e2, uint64_t e3, uint64_t e4, uint64_t e5, uint64_t e6, uint64_t e7) {
  for (uint64_t q = blockIdx.x * 256ull + threadIdx.x; q < nnz; q += gridDim.x * 256ull) {
    const uint32_t p = perm[q];
    // panel k is stored from eoff[k] rounded up to a multiple of 64 entries: every panel adds < 64 entries of padding
    uint64_t d = q;
    if (q >= e1) d = q - e1 + ((e1 + 63) & ~63ull);
    const uint64_t b1 = (e1 + 63) & ~63ull, b2 = (b1 + (e2 - e1) + 63) & ~63ull, b3 = (b2 + (e3 - e2) + 63) & ~63ull, b4 = (b3 + (e4 - e3) + 63) & ~63ull,
                   b5 = (b4 + (e5 - e4) + 63) & ~63ull, b6 = (b5 + (e6 - e5) + 63) & ~63ull, b7 = (b6 + (e7 - e6) + 63) & ~63ull;
    if (q >= e7) d = q - e7 + b7; else if (q >= e6) d = q - e6 + b6; else if (q >= e5) d = q - e5 + b5; else if (q >= e4) d = q - e4 + b4;
    else if (q >= e3) d = q - e3 + b3; else if (q >= e2) d = q - e2 + b2; else if (q >= e1) d = q - e1 + b1;
    pcol[d] = rank[col[p]] >> 3; pval[d] = val[p]; prow[q] = rowidx[p];
  }
}
// head[q] = 1 where a new sub-row starts (first entry of a panel, or the row changes)
static __global__ void k_xp_heads(const uint32_t* __restrict__ prow, uint64_t nnz, uint64_t e0, uint64_t e1, uint64_t e2, uint64_t e3, uint64_t e4, uint64_t e5, uint64_t e6, uint64_t e7,
                                  uint32_t* __restrict__ head) {
  for (uint64_t q = blockIdx.x * 256ull + threadIdx.x; q < nnz; q += gridDim.x * 256ull) {
    const bool pstart = q == e0 || q == e1 || q == e2 || q == e3 || q == e4 || q == e5 || q == e6 || q == e7;
    head[q] = (pstart || prow[q] != prow[q - 1]) ? 1u : 0u;
  }
}
// sub-row s (global numbering, panel-major) starts at entry q: rowptr slot s + panel, value relative to the panel
static __global__ void k_xp_subrows(const uint32_t* __restrict__ head, const uint32_t* __restrict__ sidx, const uint32_t* __restrict__ prow, uint64_t q0, uint64_t q1,
                                    uint32_t panel, uint32_t* __restrict__ rowptr, uint32_t* __restrict__ subrow_row) {
  for (uint64_t q = q0 + blockIdx.x * 256ull + threadIdx.x; q < q1; q += gridDim.x * 256ull)
    if (head[q]) { const uint32_t s = sidx[q]; rowptr[s + panel] = (uint32_t)(q - q0); subrow_row[s] = prow[q]; }
}
static __global__ void k_xp_set(uint32_t* p, uint32_t v) { *p = v; }
// xhot[k*H + h] = u[hot column h of panel k]: the eight LDS tables' contents, gathered once per call (every workgroup of a
// panel then loads its table with coalesced reads; gathering in each of the 32 workgroups cost 7-16 us of L2 traffic)
template <class T> __global__ void k_xp_hot_gather(const T* __restrict__ u, const uint32_t* __restrict__ hot_cols, uint32_t total, T* __restrict__ xhot) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) xhot[i] = u[hot_cols[i]];
}
// y(i) = sum of the partials of row i's sub-rows, in panel order.  A workgroup owns XP_RB consecutive rows: their
// sub-rows are one contiguous run in each panel (sub-rows are in row order inside a panel), so the eight runs are read
// with coalesced loads and accumulated panel after panel in LDS — no per-row index chain, fixed order => reproducible.
constexpr uint32_t XP_RB = 2048;
static __global__ void k_xp_local_rows(const uint32_t* __restrict__ subrow_row, uint64_t F, uint16_t* __restrict__ lrow) {
  for (uint64_t s = blockIdx.x * 256ull + threadIdx.x; s < F; s += gridDim.x * 256ull) lrow[s] = (uint16_t)(subrow_row[s] % XP_RB);
}
static __global__ void k_xp_block_starts(const uint32_t* __restrict__ subrow_row, uint32_t nblocks, uint64_t s0, uint64_t s1, uint64_t s2, uint64_t s3, uint64_t s4, uint64_t s5,
                                         uint64_t s6, uint64_t s7, uint64_t s8, uint32_t* __restrict__ blockptr) {
  const uint64_t so[XP + 1] = {s0, s1, s2, s3, s4, s5, s6, s7, s8};
  for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < (nblocks + 1) * XP; t += gridDim.x * 256) {
    const uint32_t b = t / XP, k = t % XP; const uint64_t target = (uint64_t)b * XP_RB;
    uint64_t lo = so[k], hi = so[k + 1];                       // first sub-row of panel k whose row is >= target
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (subrow_row[mid] < target) lo = mid + 1; else hi = mid; }
    blockptr[t] = (uint32_t)lo;
  }
}
template <class T, class SR>
__global__ __launch_bounds__(512) void k_xp_combine(uint32_t nrows, const uint32_t* __restrict__ blockptr, const uint16_t* __restrict__ subrow_lrow, const T* __restrict__ partial,
                                                    T* __restrict__ y, uint8_t* __restrict__ ypres, const SR sr) {
  __shared__ T acc[XP_RB];
  __shared__ uint8_t has[XP_RB];
  const uint32_t b = blockIdx.x, r0 = b * XP_RB;
  for (uint32_t i = threadIdx.x; i < XP_RB; i += 512) has[i] = 0;
  // the loads of all eight runs are independent of the LDS phase: first sub-row of every panel is fetched up front
  uint32_t lo[XP], hi[XP];
#pragma unroll
  for (int k = 0; k < XP; k++) { lo[k] = blockptr[b * XP + k]; hi[k] = blockptr[(b + 1) * XP + k]; }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < XP; k++) {
    for (uint32_t s = lo[k] + threadIdx.x; s < hi[k]; s += 512) {         // one sub-row of a row per panel: no two threads meet on a row
      const uint32_t r = subrow_lrow[s]; const T v = partial[s];
      if (has[r]) acc[r] = sr.add(acc[r], v); else { acc[r] = v; has[r] = 1; }
    }
    __syncthreads();
  }
  for (uint32_t i = threadIdx.x; i < XP_RB; i += 512) {
    const uint32_t r = r0 + i;
    if (r < nrows) { if (has[i]) y[r] = acc[i]; ypres[r] = has[i]; }
  }
}

template <class T> void build_xcd_plan(DevCSR& M, int ncu) {
  auto grid_n = [](uint64_t n) { uint64_t b = (n + 255) / 256; if (b < 1) b = 1; if (b > 8192) b = 8192; return (unsigned)b; };
  auto* P = new XcdPlan(); M.xcd.reset(P);
  const uint32_t n = M.ncols; const uint64_t nnz = M.nnz;
  // 1. deal the 128-byte lines of u to the panels (equal entry counts), rank every panel's columns by frequency
  constexpr uint32_t HH = xt_hot<T>::H;
  const uint32_t line = 128 / (uint32_t)sizeof(T), nlines = (n + line - 1) / line;
  DevBuf cnt((size_t)n * 4 + 4), rank((size_t)n * 4 + 4);
  P->hot_cols.alloc((size_t)XP * HH * 4 + 4);
  GRB_HIP(hipMemsetAsync(cnt.p, 0, (size_t)n * 4 + 4, stream()));
  GRB_HIP(hipMemsetAsync(P->hot_cols.p, 0, (size_t)XP * HH * 4 + 4, stream()));
  hipLaunchKernelGGL(k_wp_col_hist, dim3(grid_n(nnz)), dim3(256), 0, stream(), M.col.as<uint32_t>(), nnz, cnt.as<uint32_t>());
  uint32_t cstart[XP + 1];
  {
    DevBuf negw((size_t)nlines * 4 + 4), lid((size_t)nlines * 4 + 4), negw2((size_t)nlines * 4 + 4), lsorted((size_t)nlines * 4 + 4), pol((size_t)nlines + 8);
    hipLaunchKernelGGL(k_xp_line_weights, dim3(grid_n(nlines)), dim3(256), 0, stream(), cnt.as<uint32_t>(), n, line, nlines, negw.as<uint32_t>(), lid.as<uint32_t>());
    sort_pairs_u32(negw.as<uint32_t>(), negw2.as<uint32_t>(), lid.as<uint32_t>(), lsorted.as<uint32_t>(), nlines, 32);
    const uint32_t ntop = nlines < 4096u ? nlines : 4096u;                 // the heavy head of the distribution is balanced exactly
    std::vector<uint32_t> topw(ntop); std::vector<uint8_t> topk(ntop);
    GRB_HIP(hipMemcpyAsync(topw.data(), negw2.p, (size_t)ntop * 4, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
    uint64_t load[XP] = {0};
    for (uint32_t i = 0; i < ntop; i++) { int best = 0; for (int k = 1; k < XP; k++) if (load[k] < load[best]) best = k; topk[i] = (uint8_t)best; load[best] += 0xFFFFFFFFu - topw[i]; }
    DevBuf dtop((size_t)ntop + 8);
    GRB_HIP(hipMemcpyAsync(dtop.p, topk.data(), ntop, hipMemcpyHostToDevice, stream()));
    hipLaunchKernelGGL(k_xp_deal_lines, dim3(grid_n(nlines)), dim3(256), 0, stream(), lsorted.as<uint32_t>(), nlines, (const uint8_t*)dtop.p, ntop, (uint8_t*)pol.p);
    DevBuf k64((size_t)n * 8 + 8), k64o((size_t)n * 8 + 8), cin((size_t)n * 4 + 4), cout((size_t)n * 4 + 4), dstart(XP * 4);
    hipLaunchKernelGGL(k_xp_column_keys, dim3(grid_n(n)), dim3(256), 0, stream(), cnt.as<uint32_t>(), n, line, (const uint8_t*)pol.p, (unsigned long long*)k64.p, cin.as<uint32_t>());
    sort_pairs_u64((const uint64_t*)k64.p, (uint64_t*)k64o.p, cin.as<uint32_t>(), cout.as<uint32_t>(), n, 64);
    GRB_HIP(hipMemsetAsync(dstart.p, 0xFF, XP * 4, stream()));
    hipLaunchKernelGGL(k_xp_panel_starts, dim3(grid_n(n)), dim3(256), 0, stream(), (const unsigned long long*)k64o.p, n, dstart.as<uint32_t>());
    GRB_HIP(hipMemcpyAsync(cstart, dstart.p, XP * 4, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));   // (also orders the host buffers above)
    cstart[XP] = n;
    for (int k = XP - 1; k >= 0; k--) if (cstart[k] == 0xFFFFFFFFu) cstart[k] = cstart[k + 1];      // a panel without columns
    hipLaunchKernelGGL(k_xp_column_codes, dim3(grid_n(n)), dim3(256), 0, stream(), (const unsigned long long*)k64o.p, cout.as<uint32_t>(), n, HH, cstart[0], cstart[1], cstart[2], cstart[3],
                       cstart[4], cstart[5], cstart[6], cstart[7], rank.as<uint32_t>(), P->hot_cols.as<uint32_t>());
    GRB_HIP(hipStreamSynchronize(stream()));
  }
  for (int k = 0; k < XP; k++) { const uint32_t nk = cstart[k + 1] - cstart[k]; P->nhot[k] = nk < HH ? nk : HH; }
  // 2. entries grouped by panel (stable: row-major order is kept inside a panel)
  DevBuf pk(nnz * 4 + 4), pidx(nnz * 4 + 4), pk2(nnz * 4 + 4), perm(nnz * 4 + 4), rowidx(nnz * 4 + 4), prow(nnz * 4 + 4), hc(XP * 8);
  hipLaunchKernelGGL(k_xp_panel_keys, dim3(grid_n(nnz)), dim3(256), 0, stream(), M.col.as<uint32_t>(), nnz, rank.as<uint32_t>(), pk.as<uint32_t>(), pidx.as<uint32_t>());
  GRB_HIP(hipMemsetAsync(hc.p, 0, XP * 8, stream()));
  hipLaunchKernelGGL(k_xp_hist8, dim3(1024), dim3(256), 0, stream(), pk.as<uint32_t>(), nnz, hc.as<unsigned long long>());
  sort_pairs_u32(pk.as<uint32_t>(), pk2.as<uint32_t>(), pidx.as<uint32_t>(), perm.as<uint32_t>(), nnz, 3);
  unsigned long long hcnt[XP];
  GRB_HIP(hipMemcpyAsync(hcnt, hc.p, XP * 8, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
  P->eoff[0] = 0; for (int k = 0; k < XP; k++) P->eoff[k + 1] = P->eoff[k] + hcnt[k];
  csr_row_indices(M, rowidx.as<uint32_t>());
  P->ebase[0] = 0; for (int k = 0; k < XP; k++) P->ebase[k + 1] = (P->ebase[k] + (P->eoff[k + 1] - P->eoff[k]) + 63) & ~63ull;
  P->pcol.alloc((P->ebase[XP] + 64) * 4 + 4); P->pval.alloc((P->ebase[XP] + 64) * sizeof(T) + 8);
  hipLaunchKernelGGL((k_xp_gather_entries<T>), dim3(grid_n(nnz)), dim3(256), 0, stream(), perm.as<uint32_t>(), nnz, M.col.as<uint32_t>(), M.val.as<T>(),
                     rank.as<uint32_t>(), rowidx.as<uint32_t>(), P->pcol.as<uint32_t>(), P->pval.as<T>(), prow.as<uint32_t>(),
                     P->eoff[1], P->eoff[2], P->eoff[3], P->eoff[4], P->eoff[5], P->eoff[6], P->eoff[7]);
  // 3. sub-rows
  DevBuf head(nnz * 4 + 4), sidx(nnz * 4 + 4);
  hipLaunchKernelGGL(k_xp_heads, dim3(grid_n(nnz)), dim3(256), 0, stream(), prow.as<uint32_t>(), nnz, P->eoff[0], P->eoff[1], P->eoff[2], P->eoff[3], P->eoff[4], P->eoff[5],
                     P->eoff[6], P->eoff[7], head.as<uint32_t>());
  exclusive_scan_u32(head.as<uint32_t>(), sidx.as<uint32_t>(), nnz);
  for (int k = 0; k <= XP; k++) {
    if (P->eoff[k] >= nnz) { uint32_t lp = 0, lh = 0;
      GRB_HIP(hipMemcpyAsync(&lp, sidx.as<uint32_t>() + (nnz - 1), 4, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipMemcpyAsync(&lh, head.as<uint32_t>() + (nnz - 1), 4, hipMemcpyDeviceToHost, stream()));
      GRB_HIP(hipStreamSynchronize(stream())); P->soff[k] = (uint64_t)lp + lh; }
    else { uint32_t v = 0; GRB_HIP(hipMemcpyAsync(&v, sidx.as<uint32_t>() + P->eoff[k], 4, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream())); P->soff[k] = v; }
  }
  P->F = P->soff[XP];
  P->subrow_row.alloc(P->F * 4 + 4);
  P->rowptr.alloc((P->F + XP) * 4 + 4);
  for (int k = 0; k < XP; k++) {
    if (P->eoff[k + 1] > P->eoff[k])
      hipLaunchKernelGGL(k_xp_subrows, dim3(grid_n(P->eoff[k + 1] - P->eoff[k])), dim3(256), 0, stream(), head.as<uint32_t>(), sidx.as<uint32_t>(), prow.as<uint32_t>(),
                         P->eoff[k], P->eoff[k + 1], (uint32_t)k, P->rowptr.as<uint32_t>(), P->subrow_row.as<uint32_t>());
    hipLaunchKernelGGL(k_xp_set, dim3(1), dim3(1), 0, stream(), P->rowptr.as<uint32_t>() + P->soff[k + 1] + k, (uint32_t)(P->eoff[k + 1] - P->eoff[k]));   // end sentinel of panel k
  }
  // 4. row blocks of the merge kernel: where each block's run of sub-rows starts in every panel
  {
    const uint32_t nblocks = (uint32_t)(((uint64_t)M.nrows + XP_RB - 1) / XP_RB);
    P->blockptr.alloc(((size_t)nblocks + 1) * XP * 4 + 4);
    P->subrow_lrow.alloc(P->F * 2 + 4);
    hipLaunchKernelGGL(k_xp_local_rows, dim3(grid_n(P->F)), dim3(256), 0, stream(), P->subrow_row.as<uint32_t>(), P->F, P->subrow_lrow.as<uint16_t>());
    hipLaunchKernelGGL(k_xp_block_starts, dim3(grid_n(((uint64_t)nblocks + 1) * XP)), dim3(256), 0, stream(), P->subrow_row.as<uint32_t>(), nblocks, P->soff[0], P->soff[1], P->soff[2],
                       P->soff[3], P->soff[4], P->soff[5], P->soff[6], P->soff[7], P->soff[8], P->blockptr.as<uint32_t>());
  }
  // 5. tiles of 256 entries per panel and the sub-row each begins in; row-start flags into the column words
  P->toff[0] = 0;
  for (int k = 0; k < XP; k++) {
    const uint64_t ek = P->eoff[k + 1] - P->eoff[k];
    P->ntasks[k] = (uint32_t)((ek + WP_ENT - 1) / WP_ENT);
    P->toff[k + 1] = P->toff[k] + (uint64_t)P->ntasks[k] + 1;
  }
  P->tasks.alloc(P->toff[XP] * 4 + 4);
  for (int k = 0; k < XP; k++) {
    const uint64_t fk = P->soff[k + 1] - P->soff[k];
    hipLaunchKernelGGL(k_xt_tile_rows, dim3(grid_n(P->ntasks[k] + 1)), dim3(256), 0, stream(), P->rowptr.as<uint32_t>() + P->soff[k] + k, (uint32_t)fk, P->ntasks[k],
                       P->tasks.as<uint32_t>() + P->toff[k]);
    hipLaunchKernelGGL(k_wp_mark_row_starts, dim3(grid_n(fk)), dim3(256), 0, stream(), P->rowptr.as<uint32_t>() + P->soff[k] + k, (uint32_t)fk, P->pcol.as<uint32_t>() + P->ebase[k]);
  }
  constexpr uint32_t H = xt_hot<T>::H;
  P->args.alloc(XP * sizeof(WpArgs<T>));
  size_t coff[XP + 1]; coff[0] = 0;
  const uint32_t wpp = (uint32_t)(ncu / XP) * WP_WGS_PER_CU * WP_WAVES;       // waves per panel
  uint32_t kt[XP];
  for (int k = 0; k < XP; k++) { kt[k] = wp_chunk_tasks(P->ntasks[k], wpp); coff[k + 1] = coff[k] + (P->ntasks[k] + kt[k] - 1) / kt[k]; }
  P->carry.alloc((coff[XP] + 1) * sizeof(WpCarry<T>)); P->maxchunks = 1;
  for (int k = 0; k < XP; k++) if (coff[k + 1] - coff[k] > P->maxchunks) P->maxchunks = (uint32_t)(coff[k + 1] - coff[k]);
  P->xhot.alloc((size_t)XP * H * sizeof(T) + 8); P->partial.alloc(P->F * sizeof(T) + 8); P->scratch.alloc(P->F + 8);
  WpArgs<T> ha[XP];
  for (int k = 0; k < XP; k++) {
    WpArgs<T>& a = ha[k];
    const uint32_t fk = (uint32_t)(P->soff[k + 1] - P->soff[k]), ek = (uint32_t)(P->eoff[k + 1] - P->eoff[k]);
    a.rowptr = P->rowptr.as<uint32_t>() + P->soff[k] + k; a.pcol = P->pcol.as<uint32_t>() + P->ebase[k];
    a.aval = P->pval.as<T>() + P->ebase[k];
    a.x = P->xhot.as<T>() + (size_t)k * H; a.xorig = nullptr; a.hot_cols = P->hot_cols.as<uint32_t>() + (size_t)k * H;      // u comes with the launch
    a.trow = P->tasks.as<uint32_t>() + P->toff[k]; a.tent = a.trow;      // first sub-row of every tile
    a.y = P->partial.as<T>() + P->soff[k]; a.ypres = P->scratch.as<uint8_t>() + P->soff[k];
    a.carry = P->carry.as<WpCarry<T>>() + coff[k];
    a.nrows = fk; a.ntasks = P->ntasks[k]; a.nnz = ek; a.tasks_per_chunk = kt[k]; a.static_pct = wp_env("GRB_MI355X_WP_STATIC", WP_STATIC_PCT);
    a.nhot = P->nhot[k]; a.nwarm = H;
  }
  if (getenv("GRB_MI355X_VERBOSE"))
    for (int k = 0; k < XP; k++)
      fprintf(stderr, "[grb] xcd plan panel %d: entries %llu sub-rows %llu tasks %u chunk %u columns %u hot %u\n", k, (unsigned long long)(P->eoff[k + 1] - P->eoff[k]),
              (unsigned long long)(P->soff[k + 1] - P->soff[k]), P->ntasks[k], kt[k], cstart[k + 1] - cstart[k], P->nhot[k]);
  GRB_HIP(hipMemcpyAsync(P->args.p, ha, sizeof(ha), hipMemcpyHostToDevice, stream()));
  P->tsize = (int)sizeof(T);
  GRB_HIP(hipStreamSynchronize(stream()));
}

template <class T> bool run_xcd(const SpmvCall& c, const SemiringDesc& d, int ncu) {
  DevCSR& M = *c.M;
  if (ncu < XP || ncu % XP) return false;
  if (c.aval && c.aval != M.val.p) return false;      // the plan's panel-major values are a copy of the stored ones (no typecast)
  auto* P = static_cast<XcdPlan*>(M.xcd.get());
  if (!P || P->tsize != (int)sizeof(T)) { build_xcd_plan<T>(M, ncu); P = static_cast<XcdPlan*>(M.xcd.get()); }
  const bool uses_u = d.flip ? binop_uses_x(d.mulop) : binop_uses_y(d.mulop);
  WpArgs<T> a0{}; a0.xorig = (const T*)c.uval; a0.nrows = M.ncols;       // the only per-call pointer of the pipeline: u itself (and its length)
  constexpr uint32_t H = xt_hot<T>::H;
  if (uses_u) hipLaunchKernelGGL((k_xp_hot_gather<T>), dim3((XP * H + 255) / 256), dim3(256), 0, stream(), (const T*)c.uval, P->hot_cols.as<uint32_t>(), (uint32_t)(XP * H), P->xhot.as<T>());
  with_semiring<T>(d, [&](auto sr) {
    typedef decltype(sr) SR;
    hipLaunchKernelGGL((k_spmv_tiles<T, SR>), dim3(ncu * WP_WGS_PER_CU), dim3(WP_WAVES * 64), 0, stream(), a0, (const WpArgs<T>*)P->args.p, sr);
    hipLaunchKernelGGL((k_spmv_wavepipe_fixup<T, SR>), dim3((P->maxchunks + 255) / 256, XP), dim3(256), 0, stream(), (const WpCarry<T>*)nullptr, P->maxchunks, (T*)nullptr, (uint8_t*)nullptr,
                       (const WpArgs<T>*)P->args.p, sr);
    const uint32_t nblocks = (uint32_t)(((uint64_t)M.nrows + XP_RB - 1) / XP_RB);
    hipLaunchKernelGGL((k_xp_combine<T, SR>), dim3(nblocks), dim3(512), 0, stream(), M.nrows, P->blockptr.as<uint32_t>(), P->subrow_lrow.as<uint16_t>(), P->partial.as<T>(),
                       (T*)c.tval, c.tpres, sr);
    g_last_plan += std::string("k_spmv_xcd<") + (sr.is_static ? "static" : "dynamic") + ",subrows=" + std::to_string(P->F) + "> ";
  });
  return true;
}

}  // namespace grb
