// grb_matops.hip — O(nnz) matrix kernels around the SpGEMM hot path, all on CSR in HBM:
//   * row-wise 3-way merge  C<M,replace> = accum(C, T)   (the GraphBLAS write-back)
//   * row-wise union / intersection (eWiseAdd / eWiseMult)
//   * entry filters (select: tril/triu/diag/offdiag/value tests) and flag compaction
//   * value maps (apply, apply with a bound scalar), row reductions (reduce to vector)
// Two-pass pattern everywhere: count per row -> exclusive scan (rocPRIM) -> fill.  Rows are merged by
// one thread each (sorted column lists), which is simple, exact and order-preserving; the heavy
// lifting of the hot path is in grb_spgemm.hip / grb_spmv_kernels.hpp, not here.
#include "grb_api.hpp"
#include "grb_device.hpp"
#include "grb_matops.hpp"

namespace grb {

static inline unsigned grid_rows(uint64_t n) { uint64_t b = (n + 255) / 256; if (b < 1) b = 1; if (b > 65535u * 16) b = 65535u * 16; return (unsigned)b; }
static inline unsigned grid_n(uint64_t n) { uint64_t b = (n + 255) / 256; if (b < 1) b = 1; if (b > 4096) b = 4096; return (unsigned)b; }

// ---- mask helpers --------------------------------------------------------------------------------------------
// truth of mask entry p (valued mask: stored AND non-zero; structural: stored)
__device__ __forceinline__ bool mask_truth(const void* mval, int mcode, uint32_t p, bool structural) {
  if (structural || !mval) return true;
  switch (type_size(mcode)) {
    case 1: return ((const uint8_t*)mval)[p] != 0;
    case 2: return ((const uint16_t*)mval)[p] != 0;
    case 4: return mcode == T_FP32 ? ((const float*)mval)[p] != 0.0f : ((const uint32_t*)mval)[p] != 0;
    default: return mcode == T_FP64 ? ((const double*)mval)[p] != 0.0 : ((const uint64_t*)mval)[p] != 0;
  }
}

// ---- write-back merge:  out(i,:) from C(i,:), T(i,:), M(i,:) -------------------------------------------------------
// FILL = false: count entries of each output row.  FILL = true: write them at orow[i].
template <class T, bool FILL, bool MATH = false>
__global__ void k_writeback(uint32_t nrows, const uint32_t* __restrict__ crp, const uint32_t* __restrict__ ccol, const T* __restrict__ cval,
                            const uint32_t* __restrict__ trp, const uint32_t* __restrict__ tcol, const T* __restrict__ tval,
                            const uint32_t* __restrict__ mrp, const uint32_t* __restrict__ mcol, const void* __restrict__ mval, int mcode,
                            bool has_mask, bool mstruct, bool mcomp, bool replace, int accum,
                            uint32_t* __restrict__ ocount, const uint32_t* __restrict__ orp, uint32_t* __restrict__ ocol, T* __restrict__ oval) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < nrows; i += (uint64_t)gridDim.x * 256ull) {
    uint32_t pc = crp[i], ec = crp[i + 1], pt = trp[i], et = trp[i + 1];
    uint32_t pm = has_mask ? mrp[i] : 0, em = has_mask ? mrp[i + 1] : 0;
    uint32_t w = FILL ? orp[i] : 0, cnt = 0;
    while (pc < ec || pt < et) {
      const uint32_t jc = pc < ec ? ccol[pc] : 0xFFFFFFFFu, jt = pt < et ? tcol[pt] : 0xFFFFFFFFu;
      const uint32_t j = jc < jt ? jc : jt;
      const bool inc = jc == j, intt = jt == j;
      bool m = true;
      if (has_mask) {
        while (pm < em && mcol[pm] < j) pm++;
        m = pm < em && mcol[pm] == j && mask_truth(mval, mcode, pm, mstruct);
      }
      if (mcomp) m = !m;
      bool outp; T outv = T();
      if (m) {
        if (accum >= 0) {
          if (inc && intt) { outp = true; if (FILL) outv = apply_binop<T, true, MATH>(accum, cval[pc], tval[pt]); }
          else if (intt) { outp = true; if (FILL) outv = tval[pt]; }
          else { outp = true; if (FILL) outv = cval[pc]; }
        } else { outp = intt; if (FILL && intt) outv = tval[pt]; }
      } else if (replace) outp = false;
      else { outp = inc; if (FILL && inc) outv = cval[pc]; }
      if (outp) { if (FILL) { ocol[w] = j; oval[w] = outv; w++; } cnt++; }
      if (inc) pc++;
      if (intt) pt++;
    }
    if (!FILL) ocount[i] = cnt;
  }
}

void csr_writeback(int code, uint32_t nrows, const DevCSR& C, const DevCSR& Tm, const DevCSR* M, int mcode, bool mstruct, bool mcomp,
                   bool replace, int accum, DevCSR& out) {
  out.clear(); out.nrows = nrows; out.ncols = C.ncols;
  DevBuf cnt(((size_t)nrows + 1) * 4);
  out.rowptr.alloc(((size_t)nrows + 1) * 4);
  GRB_HIP(hipMemsetAsync(cnt.p, 0, ((size_t)nrows + 1) * 4, stream()));
  dispatch_type(code, [&]<class T>() {
    hipLaunchKernelGGL((k_writeback<T, false>), dim3(grid_rows(nrows)), dim3(256), 0, stream(), nrows, C.rowptr.as<uint32_t>(), C.col.as<uint32_t>(), C.val.as<T>(),
                       Tm.rowptr.as<uint32_t>(), Tm.col.as<uint32_t>(), Tm.val.as<T>(), M ? M->rowptr.as<uint32_t>() : nullptr, M ? M->col.as<uint32_t>() : nullptr,
                       M ? M->val.p : nullptr, mcode, M != nullptr, mstruct, mcomp, replace, accum, cnt.as<uint32_t>(), nullptr, nullptr, (T*)nullptr);
    exclusive_scan_u32(cnt.as<uint32_t>(), out.rowptr.as<uint32_t>(), (uint64_t)nrows + 1);
    uint32_t total = 0;
    GRB_HIP(hipMemcpyAsync(&total, out.rowptr.as<uint32_t>() + nrows, 4, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
    out.nnz = total; out.col.alloc((size_t)total * 4); out.val.alloc((size_t)total * sizeof(T));
#define GRB_WB_FILL(MATH) hipLaunchKernelGGL((k_writeback<T, true, MATH>), dim3(grid_rows(nrows)), dim3(256), 0, stream(), nrows, C.rowptr.as<uint32_t>(), C.col.as<uint32_t>(), C.val.as<T>(), \
                       Tm.rowptr.as<uint32_t>(), Tm.col.as<uint32_t>(), Tm.val.as<T>(), M ? M->rowptr.as<uint32_t>() : nullptr, M ? M->col.as<uint32_t>() : nullptr, \
                       M ? M->val.p : nullptr, mcode, M != nullptr, mstruct, mcomp, replace, accum, (uint32_t*)nullptr, out.rowptr.as<uint32_t>(), out.col.as<uint32_t>(), out.val.as<T>())
    if (accum >= 0 && binop_needs_math(accum)) GRB_WB_FILL(true); else GRB_WB_FILL(false);
#undef GRB_WB_FILL
  });
  GRB_HIP(hipGetLastError());
  out.valid = true;
}

// ---- element-wise union / intersection -----------------------------------------------------------------------------
template <class T, bool FILL, bool MATH = false>      // MATH: the operator may call into the math library (see grb_ops.hpp)
__global__ void k_ewise(uint32_t nrows, const uint32_t* __restrict__ arp, const uint32_t* __restrict__ acol, const T* __restrict__ aval,
                        const uint32_t* __restrict__ brp, const uint32_t* __restrict__ bcol, const T* __restrict__ bval, int op, bool is_union,
                        uint32_t* __restrict__ ocount, const uint32_t* __restrict__ orp, uint32_t* __restrict__ ocol, T* __restrict__ oval) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < nrows; i += (uint64_t)gridDim.x * 256ull) {
    uint32_t pa = arp[i], ea = arp[i + 1], pb = brp[i], eb = brp[i + 1];
    uint32_t w = FILL ? orp[i] : 0, cnt = 0;
    while (pa < ea || pb < eb) {
      const uint32_t ja = pa < ea ? acol[pa] : 0xFFFFFFFFu, jb = pb < eb ? bcol[pb] : 0xFFFFFFFFu;
      if (ja == jb) { if (FILL) { ocol[w] = ja; oval[w] = apply_binop<T, true, MATH>(op, aval[pa], bval[pb]); w++; } cnt++; pa++; pb++; }
      else if (ja < jb) { if (is_union) { if (FILL) { ocol[w] = ja; oval[w] = aval[pa]; w++; } cnt++; } pa++; }
      else { if (is_union) { if (FILL) { ocol[w] = jb; oval[w] = bval[pb]; w++; } cnt++; } pb++; }
    }
    if (!FILL) ocount[i] = cnt;
  }
}

void csr_ewise(int code, const DevCSR& A, const void* aval, const DevCSR& B, const void* bval, int op, bool is_union, DevCSR& out) {
  const uint32_t nrows = A.nrows;
  out.clear(); out.nrows = nrows; out.ncols = A.ncols;
  DevBuf cnt(((size_t)nrows + 1) * 4);
  out.rowptr.alloc(((size_t)nrows + 1) * 4);
  GRB_HIP(hipMemsetAsync(cnt.p, 0, ((size_t)nrows + 1) * 4, stream()));
  dispatch_type(code, [&]<class T>() {
    hipLaunchKernelGGL((k_ewise<T, false>), dim3(grid_rows(nrows)), dim3(256), 0, stream(), nrows, A.rowptr.as<uint32_t>(), A.col.as<uint32_t>(), (const T*)aval,
                       B.rowptr.as<uint32_t>(), B.col.as<uint32_t>(), (const T*)bval, op, is_union, cnt.as<uint32_t>(), nullptr, nullptr, (T*)nullptr);
    exclusive_scan_u32(cnt.as<uint32_t>(), out.rowptr.as<uint32_t>(), (uint64_t)nrows + 1);
    uint32_t total = 0;
    GRB_HIP(hipMemcpyAsync(&total, out.rowptr.as<uint32_t>() + nrows, 4, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
    out.nnz = total; out.col.alloc((size_t)total * 4); out.val.alloc((size_t)total * sizeof(T));
#define GRB_EW_FILL(MATH) hipLaunchKernelGGL((k_ewise<T, true, MATH>), dim3(grid_rows(nrows)), dim3(256), 0, stream(), nrows, A.rowptr.as<uint32_t>(), A.col.as<uint32_t>(), (const T*)aval, \
                       B.rowptr.as<uint32_t>(), B.col.as<uint32_t>(), (const T*)bval, op, is_union, (uint32_t*)nullptr, out.rowptr.as<uint32_t>(), out.col.as<uint32_t>(), out.val.as<T>())
    if (binop_needs_math(op)) GRB_EW_FILL(true); else GRB_EW_FILL(false);
#undef GRB_EW_FILL
  });
  GRB_HIP(hipGetLastError());
  out.valid = true;
}

// ---- keep-flag compaction of a CSR (select, masked-SpGEMM output, mask application) -------------------------------------
__global__ void k_row_of_entry(const uint32_t* __restrict__ rowptr, uint32_t nrows, uint32_t* __restrict__ rowidx) {
  for (uint64_t r = blockIdx.x * 256ull + threadIdx.x; r < nrows; r += (uint64_t)gridDim.x * 256ull)
    for (uint32_t p = rowptr[r]; p < rowptr[r + 1]; p++) rowidx[p] = (uint32_t)r;
}
__global__ void k_count_kept(const uint32_t* __restrict__ rowptr, uint32_t nrows, const uint8_t* __restrict__ keep, uint32_t* __restrict__ cnt) {
  for (uint64_t r = blockIdx.x * 256ull + threadIdx.x; r < nrows; r += (uint64_t)gridDim.x * 256ull) {
    uint32_t c = 0; for (uint32_t p = rowptr[r]; p < rowptr[r + 1]; p++) c += keep[p] != 0; cnt[r] = c;
  }
}
template <int TS> __global__ void k_compact_rows(const uint32_t* __restrict__ rowptr, uint32_t nrows, const uint8_t* __restrict__ keep,
                                                 const uint32_t* __restrict__ col, const uint8_t* __restrict__ val, const uint32_t* __restrict__ orp,
                                                 uint32_t* __restrict__ ocol, uint8_t* __restrict__ oval) {
  typedef typename std::conditional<TS == 8, uint64_t, typename std::conditional<TS == 4, uint32_t,
          typename std::conditional<TS == 2, uint16_t, uint8_t>::type>::type>::type W;
  for (uint64_t r = blockIdx.x * 256ull + threadIdx.x; r < nrows; r += (uint64_t)gridDim.x * 256ull) {
    uint32_t w = orp[r];
    for (uint32_t p = rowptr[r]; p < rowptr[r + 1]; p++) if (keep[p]) { ocol[w] = col[p]; ((W*)oval)[w] = ((const W*)val)[p]; w++; }
  }
}
void csr_compact(const DevCSR& A, const void* aval, size_t ts, const uint8_t* keep, DevCSR& out) {
  const uint32_t nrows = A.nrows;
  out.clear(); out.nrows = nrows; out.ncols = A.ncols;
  DevBuf cnt(((size_t)nrows + 1) * 4);
  out.rowptr.alloc(((size_t)nrows + 1) * 4);
  GRB_HIP(hipMemsetAsync(cnt.p, 0, ((size_t)nrows + 1) * 4, stream()));
  hipLaunchKernelGGL(k_count_kept, dim3(grid_rows(nrows)), dim3(256), 0, stream(), A.rowptr.as<uint32_t>(), nrows, keep, cnt.as<uint32_t>());
  exclusive_scan_u32(cnt.as<uint32_t>(), out.rowptr.as<uint32_t>(), (uint64_t)nrows + 1);
  uint32_t total = 0;
  GRB_HIP(hipMemcpyAsync(&total, out.rowptr.as<uint32_t>() + nrows, 4, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
  out.nnz = total; out.col.alloc((size_t)total * 4); out.val.alloc((size_t)total * ts);
  const unsigned g = grid_rows(nrows);
#define GRB_COMPACT(TS) hipLaunchKernelGGL((k_compact_rows<TS>), dim3(g), dim3(256), 0, stream(), A.rowptr.as<uint32_t>(), nrows, keep, A.col.as<uint32_t>(), \
                                           (const uint8_t*)aval, out.rowptr.as<uint32_t>(), out.col.as<uint32_t>(), out.val.as<uint8_t>())
  switch (ts) { case 1: GRB_COMPACT(1); break; case 2: GRB_COMPACT(2); break; case 4: GRB_COMPACT(4); break; default: GRB_COMPACT(8); break; }
#undef GRB_COMPACT
  GRB_HIP(hipGetLastError());
  out.valid = true;
}

// positional select flags: keep entry (i,j) by its diagonal index j - i against k
__global__ void k_select_positional(const uint32_t* __restrict__ rowptr, uint32_t nrows, const uint32_t* __restrict__ col, int sel, int64_t k, uint8_t* __restrict__ keep) {
  for (uint64_t r = blockIdx.x * 256ull + threadIdx.x; r < nrows; r += (uint64_t)gridDim.x * 256ull)
    for (uint32_t p = rowptr[r]; p < rowptr[r + 1]; p++) {
      const int64_t d = (int64_t)col[p] - (int64_t)r; bool kp;
      switch (sel) { case SEL_TRIL: kp = d <= k; break; case SEL_TRIU: kp = d >= k; break; case SEL_DIAG: kp = d == k; break; default: kp = d != k; }
      keep[p] = kp ? 1 : 0;
    }
}
void select_positional_flags(const DevCSR& A, int sel, int64_t k, uint8_t* keep) {
  if (!A.nnz) return;
  hipLaunchKernelGGL(k_select_positional, dim3(grid_rows(A.nrows)), dim3(256), 0, stream(), A.rowptr.as<uint32_t>(), A.nrows, A.col.as<uint32_t>(), sel, k, keep);
}

// mask flags for entries of T: keep[p] = mask allows (i, col[p])
__global__ void k_mask_flags(uint32_t nrows, const uint32_t* __restrict__ trp, const uint32_t* __restrict__ tcol, const uint32_t* __restrict__ mrp,
                             const uint32_t* __restrict__ mcol, const void* __restrict__ mval, int mcode, bool mstruct, bool mcomp, uint8_t* __restrict__ keep) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < nrows; i += (uint64_t)gridDim.x * 256ull) {
    uint32_t pm = mrp[i]; const uint32_t em = mrp[i + 1];
    for (uint32_t p = trp[i]; p < trp[i + 1]; p++) {
      const uint32_t j = tcol[p];
      while (pm < em && mcol[pm] < j) pm++;
      bool m = pm < em && mcol[pm] == j && mask_truth(mval, mcode, pm, mstruct);
      keep[p] = (m != mcomp) ? 1 : 0;
    }
  }
}
void mask_flags(const DevCSR& Tm, const DevCSR& M, int mcode, bool mstruct, bool mcomp, uint8_t* keep) {
  if (!Tm.nnz) return;
  hipLaunchKernelGGL(k_mask_flags, dim3(grid_rows(Tm.nrows)), dim3(256), 0, stream(), Tm.nrows, Tm.rowptr.as<uint32_t>(), Tm.col.as<uint32_t>(),
                     M.rowptr.as<uint32_t>(), M.col.as<uint32_t>(), M.val.p, mcode, mstruct, mcomp, keep);
}

// ---- reduce each row with a monoid -> bitmap vector ------------------------------------------------------------------------
template <class T> __global__ void k_reduce_rows(uint32_t nrows, const uint32_t* __restrict__ rp, const T* __restrict__ val, int op, T* __restrict__ tval, uint8_t* __restrict__ tpres) {
  // one 16-lane group per row; fixed tree => deterministic
  const int lane = threadIdx.x & 15;
  const uint64_t grp = (blockIdx.x * 256ull + threadIdx.x) >> 4, ngrp = (uint64_t)gridDim.x * 16;
  for (uint64_t r = grp; r < nrows; r += ngrp) {
    const uint32_t b = rp[r], e = rp[r + 1];
    T acc = T(); bool has = false;
    for (uint32_t p = b + lane; p < e; p += 16) { acc = has ? apply_binop<T, true, false>(op, acc, val[p]) : val[p]; has = true; }
    for (int d = 8; d >= 1; d >>= 1) {
      const T ov = shfl_down_t<T>(acc, d); const int oh = __shfl_down((int)has, d, 64);
      if (oh) { acc = has ? apply_binop<T, true, false>(op, acc, ov) : ov; has = true; }
    }
    if (lane == 0) { if (has) tval[r] = acc; tpres[r] = has ? 1 : 0; }
  }
}
void csr_reduce_rows(int code, const DevCSR& A, const void* aval, int op, void* tval, uint8_t* tpres) {
  if (!A.nrows) return;
  dispatch_type(code, [&]<class T>() {
    uint64_t nb = ((uint64_t)A.nrows + 15) / 16; if (nb > 65535u * 8) nb = 65535u * 8; if (nb < 1) nb = 1;
    hipLaunchKernelGGL((k_reduce_rows<T>), dim3((unsigned)nb), dim3(256), 0, stream(), A.nrows, A.rowptr.as<uint32_t>(), (const T*)aval, op, (T*)tval, tpres);
  });
  GRB_HIP(hipGetLastError());
}

void csr_row_indices(const DevCSR& A, uint32_t* rowidx) {
  if (!A.nnz) return;
  hipLaunchKernelGGL(k_row_of_entry, dim3(grid_rows(A.nrows)), dim3(256), 0, stream(), A.rowptr.as<uint32_t>(), A.nrows, rowidx);
}

}  // namespace grb
