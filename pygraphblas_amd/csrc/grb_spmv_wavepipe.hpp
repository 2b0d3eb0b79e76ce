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

constexpr int WP_ENT = 512;                 // merge items per task = 8 per lane
constexpr int WP_PER = WP_ENT / 64;
constexpr int WP_SHORT = 24;                // rows longer than this (within one task) are reduced by the whole wave
constexpr int WP_WAVES = 16;                // waves per workgroup (1024 threads)
constexpr int WP_WGS_PER_CU = 1;            // one workgroup per CU (measured: 2 x 768 threads with 256-item tasks spills registers and is slower)
constexpr int WP_LDS_BYTES = 160 * 1024 / WP_WGS_PER_CU;
template <class T> struct wp_hot { static constexpr int H = (WP_LDS_BYTES - WP_WAVES * WP_ENT * (int)sizeof(T)) / (int)sizeof(T); };   // 12288 (8 B) / 24576 (4 B)
constexpr uint32_t WP_NONE = 0xFFFFFFFFu;

template <class T> __device__ __forceinline__ T wp_wave_total(int op, T v, T identity) {
  if constexpr (sizeof(T) >= 4) return wave_reduce_dpp<T, false>(op, v, identity); else return wave_reduce_op<T, false>(op, v);
}

template <class T> struct WpCarry {           // per wave: partial of the row it entered in the middle of / left open
  uint32_t head_row, tail_row; uint8_t head_has, head_done, tail_has, pad; T head_val, tail_val;
};

template <class T> struct WpArgs {
  const uint32_t* rowptr; const uint32_t* pcol; const T* aval; const T* x; const T* xorig; const uint32_t* hot_cols;
  const uint32_t* trow; const uint32_t* tent;      // merge-path task starts: task t begins at (row trow[t], entry tent[t]); [ntasks+1]
  T* y; uint8_t* ypres; WpCarry<T>* carry; uint32_t nrows, ntasks, nnz, tasks_per_wave, nhot, nwarm;
};

// Tasks are equal slices of the merge of {entries} and {row ends} (WP_ENT items each), so a task holds at most WP_ENT
// entries AND completes at most WP_ENT rows: runs of empty or tiny rows cannot unbalance the waves (R-MAT has both
// 10^5-entry rows and long runs of empty rows).  Task t owns entries [tent[t], tent[t+1]) and completes rows
// [trow[t], trow[t+1]); the entries of row trow[t+1] seen so far are carried to the next task.
template <class T, class SR>
__global__ __launch_bounds__(WP_WAVES * 64, WP_WGS_PER_CU * WP_WAVES / 4) void k_spmv_wavepipe(const WpArgs<T> a0, const WpArgs<T>* __restrict__ panels, const SR sr) {
  // panel mode (kernel X, grb_spmv_xcd.hpp): workgroup b works on column panel b % 8 — the XCD it is observed to run on
  const WpArgs<T> a = panels ? panels[blockIdx.x & 7] : a0;
  constexpr int H = wp_hot<T>::H;
  __shared__ T s_hot[H];
  __shared__ T s_prod[WP_WAVES][WP_ENT];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool use_a = sr.uses_a(), use_u = sr.uses_u();
  if (use_u) for (uint32_t h = threadIdx.x; h < a.nhot; h += WP_WAVES * 64) s_hot[h] = a.x[h];      // x is the rank-permuted copy of u
  __syncthreads();
  T* prod = s_prod[wv];
  const uint32_t gw = (panels ? (blockIdx.x >> 3) : blockIdx.x) * WP_WAVES + (uint32_t)__builtin_amdgcn_readfirstlane(wv);
  const uint32_t t0 = gw * a.tasks_per_wave;
  uint32_t t1 = t0 + a.tasks_per_wave; if (t1 > a.ntasks) t1 = a.ntasks;
  WpCarry<T> cr; cr.head_row = cr.tail_row = WP_NONE; cr.head_has = cr.head_done = cr.tail_has = cr.pad = 0; cr.head_val = cr.tail_val = sr.identity;
  if (t0 >= t1) { if (lane == 0) a.carry[gw] = cr; return; }

  T carry = sr.identity; bool carry_has = false, owned = true;        // partial of the row the current task starts in (wave-uniform)
  // stage 1: the coalesced loads of a task (tail lanes re-read its last entry: branch-free) + the row pointers of the
  // first 64 rows it completes
  auto load_task = [&](uint32_t e0, uint32_t cnt, uint32_t r0, uint32_t (&c)[WP_PER], T (&v)[WP_PER], uint32_t& rpa, uint32_t& rpb) {
    const uint32_t last = cnt ? cnt - 1 : 0; const uint32_t base = e0 < a.nnz ? e0 : a.nnz - 1;
#pragma unroll
    for (int u = 0; u < WP_PER; u++) {
      const uint32_t k = lane + u * 64; const uint32_t p = base + (k < cnt ? k : last);
      c[u] = a.pcol[p]; v[u] = use_a ? a.aval[p] : T();
    }
    const uint32_t rq = r0 + lane;
    rpa = a.rowptr[rq < a.nrows ? rq : a.nrows]; rpb = a.rowptr[rq + 1 < a.nrows ? rq + 1 : a.nrows];
  };
  // task descriptors live in registers, one task per lane, 63 tasks + 1 look-ahead at a time: the steady state
  // issues no dependent global load
  for (uint32_t tc = t0; tc < t1; tc += 63) {
    const uint32_t tl = tc + lane <= a.ntasks ? tc + lane : a.ntasks;
    const uint32_t d_r = a.trow[tl], d_e = a.tent[tl];
    const uint32_t nin = t1 - tc < 63u ? t1 - tc : 63u;
    uint32_t cA[WP_PER]; T vA[WP_PER]; uint32_t rpA, rpB;
    {
      const uint32_t r00 = __builtin_amdgcn_readlane(d_r, 0), e00 = __builtin_amdgcn_readlane(d_e, 0), e01 = __builtin_amdgcn_readlane(d_e, 1);
      if (tc == t0) owned = a.rowptr[r00 < a.nrows ? r00 : a.nrows] == e00;   // does this wave see the start of its first row?
      load_task(e00, e01 - e00, r00, cA, vA, rpA, rpB);
    }
    for (uint32_t i = 0; i < nin; i++) {
      const int iu = (int)__builtin_amdgcn_readfirstlane(i);
      const uint32_t r0 = __builtin_amdgcn_readlane(d_r, iu), e0 = __builtin_amdgcn_readlane(d_e, iu);
      const uint32_t r1 = __builtin_amdgcn_readlane(d_r, iu + 1), e1 = __builtin_amdgcn_readlane(d_e, iu + 1);
      const uint32_t cnt = e1 - e0;
      uint32_t cB[WP_PER]; T vB[WP_PER]; uint32_t rpAn = 0, rpBn = 0;
      const bool more = i + 1 < nin;
      if (more) {                                              // next task's loads are in flight while this one gathers and reduces
        const uint32_t e2 = __builtin_amdgcn_readlane(d_e, iu + 2 < 64 ? iu + 2 : 63);
        load_task(e1, e2 - e1, r1, cB, vB, rpAn, rpBn);
      }
      // stage 2: gathers (LDS for hot columns, L2/HBM otherwise), products into the wave's LDS slice
      T uv[WP_PER];
#pragma unroll
      for (int u = 0; u < WP_PER; u++) {
        if (use_u) {
          const uint32_t c = cA[u];                                   // rank of the column if < nwarm (0 = most frequent), else nwarm + column
          const T* base = c < a.nwarm ? a.x : a.xorig - a.nwarm;      // warm: rank-ordered copy of the top of u; cold: u itself
          const T g = base[c >= (uint32_t)H ? c : 0u];
          uv[u] = c < (uint32_t)H ? s_hot[c < (uint32_t)H ? c : 0] : g;
        } else uv[u] = T();
      }
#pragma unroll
      for (int u = 0; u < WP_PER; u++) { const uint32_t k = lane + u * 64; if (k < cnt) prod[k] = sr.mult(vA[u], uv[u]); }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
      // stage 3: rows [r0, r1) end inside this task.  Short segments are summed by one lane in entry order, long ones
      // by the whole wave (64-strided partials + fixed butterfly).  Row r0 first absorbs the carried partial.
      const uint32_t nr = r1 - r0;
      uint32_t tail_start = 0;                                  // task-local offset where the entries of row r1 begin
      for (uint32_t rbase = 0; rbase < nr; rbase += 64) {
        const uint32_t ri = rbase + lane; const bool live = ri < nr; const uint32_t r = r0 + ri;
        uint32_t rs_, re_;
        if (rbase == 0) { rs_ = rpA; re_ = rpB; } else { rs_ = live ? a.rowptr[r] : e1; re_ = live ? a.rowptr[r + 1] : e1; }
        if (rs_ < e0) rs_ = e0;                                 // only row r0 can have started in an earlier task
        if (!live) { rs_ = re_ = e0; }
        const uint32_t qs = rs_ - e0, qe = re_ - e0;
        const bool longrow = qe - qs > (uint32_t)WP_SHORT;
        T acc = sr.identity; bool has = false;
        if (!longrow && qe > qs) {                               // entry order kept; 4 LDS reads in flight per step
          uint32_t q = qs; acc = prod[q++]; has = true;
          for (; q + 4 <= qe; q += 4) { const T p0 = prod[q], p1 = prod[q + 1], p2 = prod[q + 2], p3 = prod[q + 3]; acc = sr.add(sr.add(sr.add(sr.add(acc, p0), p1), p2), p3); }
          for (; q < qe; q++) acc = sr.add(acc, prod[q]);
        }
        unsigned long long lm = __ballot(longrow);
        while (lm) {
          const int j = __builtin_ctzll(lm); lm &= lm - 1;
          const uint32_t js = __shfl(qs, j, 64), je = __shfl(qe, j, 64);
          T pa = sr.identity; bool ph = false;
          for (uint32_t q = js + lane; q < je; q += 64) { pa = ph ? sr.add(pa, prod[q]) : prod[q]; ph = true; }
          const T tot = wp_wave_total<T>(sr.add_op(), ph ? pa : sr.identity, sr.identity);
          if (lane == j) { acc = tot; has = true; }
        }
        if (rbase == 0 && lane == 0 && carry_has) { acc = has ? sr.add(carry, acc) : carry; has = true; }   // carried part comes first
        const bool to_fixup = rbase == 0 && lane == 0 && !owned;      // the row began in another wave's range
        if (live && !to_fixup) { if (has) a.y[r] = acc; a.ypres[r] = has ? 1 : 0; }
        if (rbase == 0 && !owned) {
          cr.head_row = r0; cr.head_val = shfl_t<T>(acc, 0); cr.head_has = (uint8_t)__shfl((int)has, 0, 64); cr.head_done = 1;
        }
        if (rbase + 64 >= nr) tail_start = __shfl(qe, (int)(nr - 1 - rbase), 64);
      }
      if (nr) { carry = sr.identity; carry_has = false; owned = true; }
      // entries [tail_start, cnt) belong to row r1, which ends in a later task: fold them into the carry
      if (tail_start < cnt) {
        T pa = sr.identity; bool ph = false;
        for (uint32_t q = tail_start + lane; q < cnt; q += 64) { pa = ph ? sr.add(pa, prod[q]) : prod[q]; ph = true; }
        const T tot = wp_wave_total<T>(sr.add_op(), ph ? pa : sr.identity, sr.identity);
        carry = carry_has ? sr.add(carry, tot) : tot; carry_has = true;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); __builtin_amdgcn_wave_barrier();   // LDS slice is free for the next task
      if (more) {
#pragma unroll
        for (int u = 0; u < WP_PER; u++) { cA[u] = cB[u]; vA[u] = vB[u]; }
        rpA = rpAn; rpB = rpBn;
      }
    }
  }
  // the row this wave's range ends in (if the range ends strictly inside it, or at its very end without its end marker)
  {
    const uint32_t rend = a.trow[t1], eend = a.tent[t1];
    if (rend < a.nrows && (carry_has || eend > a.rowptr[rend] || !owned)) {
      if (owned) { cr.tail_row = rend; cr.tail_val = carry; cr.tail_has = carry_has; }
      else { cr.head_row = rend; cr.head_val = carry; cr.head_has = carry_has; cr.head_done = 0; }   // the whole range lies inside one row
    }
  }
  if (lane == 0) a.carry[gw] = cr;
}

// combine the partials of rows that span several waves' ranges, in wave order
template <class T, class SR>
__global__ void k_spmv_wavepipe_fixup(const WpCarry<T>* __restrict__ carry0, uint32_t nwaves, T* __restrict__ y0, uint8_t* __restrict__ ypres0,
                                      const WpArgs<T>* __restrict__ panels, const SR sr) {
  const WpCarry<T>* carry = panels ? panels[blockIdx.y].carry : carry0;
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
static __global__ void k_wp_col_hist(const uint32_t* __restrict__ col, uint64_t nnz, uint32_t* __restrict__ cnt) {
  for (uint64_t p = blockIdx.x * 256ull + threadIdx.x; p < nnz; p += gridDim.x * 256ull) atomicAdd(&cnt[col[p]], 1u);
}
static __global__ void k_wp_neg_keys(const uint32_t* __restrict__ cnt, uint32_t n, uint32_t* __restrict__ key, uint32_t* __restrict__ id) {
  for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) { key[j] = 0xFFFFFFFFu - cnt[j]; id[j] = j; }   // ascending sort = descending count, ties by column
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

template <class T> void build_wavepipe_plan(DevCSR& M) {
  constexpr uint32_t H = wp_hot<T>::H;
  auto grid_n = [](uint64_t n) { uint64_t b = (n + 255) / 256; if (b < 1) b = 1; if (b > 8192) b = 8192; return (unsigned)b; };
  const uint32_t n = M.ncols, ntasks = (uint32_t)((M.nnz + M.nrows + WP_ENT - 1) / WP_ENT);
  M.wp_rs.alloc(((size_t)ntasks + 1) * 8);            // trow[ntasks+1] then tent[ntasks+1]
  hipLaunchKernelGGL(k_wp_task_starts, dim3(grid_n(ntasks + 1)), dim3(256), 0, stream(), M.rowptr.as<uint32_t>(), M.nrows, (uint32_t)M.nnz, ntasks,
                     M.wp_rs.as<uint32_t>(), M.wp_rs.as<uint32_t>() + (ntasks + 1));
  DevBuf cnt((size_t)n * 4 + 4), key((size_t)n * 4 + 4), id((size_t)n * 4 + 4), key2((size_t)n * 4 + 4), id2((size_t)n * 4 + 4), rank((size_t)n * 4 + 4);
  GRB_HIP(hipMemsetAsync(cnt.p, 0, (size_t)n * 4 + 4, stream()));
  hipLaunchKernelGGL(k_wp_col_hist, dim3(grid_n(M.nnz)), dim3(256), 0, stream(), M.col.as<uint32_t>(), M.nnz, cnt.as<uint32_t>());
  hipLaunchKernelGGL(k_wp_neg_keys, dim3(grid_n(n)), dim3(256), 0, stream(), cnt.as<uint32_t>(), n, key.as<uint32_t>(), id.as<uint32_t>());
  sort_pairs_u32(key.as<uint32_t>(), key2.as<uint32_t>(), id.as<uint32_t>(), id2.as<uint32_t>(), n, 32);
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
  const uint32_t nwaves = (uint32_t)ncu * WP_WGS_PER_CU * WP_WAVES;
  const uint32_t tpw = (M.wp_ntasks + nwaves - 1) / nwaves;
  if (M.wp_carry.bytes < (size_t)nwaves * sizeof(WpCarry<T>)) M.wp_carry.alloc((size_t)nwaves * sizeof(WpCarry<T>));
  WpArgs<T> a{M.rowptr.as<uint32_t>(), M.wp_pcol.as<uint32_t>(), (const T*)c.aval, (const T*)xp.p, (const T*)c.uval, M.wp_hot.as<uint32_t>(), M.wp_rs.as<uint32_t>(), M.wp_rs.as<uint32_t>() + (M.wp_ntasks + 1),
              (T*)c.tval, c.tpres, M.wp_carry.as<WpCarry<T>>(), M.nrows, M.wp_ntasks, (uint32_t)M.nnz, tpw, M.wp_nhot, M.wp_nwarm};
  with_semiring<T>(d, [&](auto sr) {
    typedef decltype(sr) SR;
    hipLaunchKernelGGL((k_spmv_wavepipe<T, SR>), dim3(ncu * WP_WGS_PER_CU), dim3(WP_WAVES * 64), 0, stream(), a, (const WpArgs<T>*)nullptr, sr);
    hipLaunchKernelGGL((k_spmv_wavepipe_fixup<T, SR>), dim3((nwaves + 255) / 256), dim3(256), 0, stream(), M.wp_carry.as<WpCarry<T>>(), nwaves, (T*)c.tval, c.tpres,
                       (const WpArgs<T>*)nullptr, sr);
    g_last_plan += std::string("k_spmv_wavepipe<") + (sr.is_static ? "static" : "dynamic") + ",hot=" + std::to_string(M.wp_nhot) + "> ";
  });
  return true;
}

}  // namespace grb
