// grb_matops.hip — O(nnz) matrix kernels around the SpGEMM hot path, all on CSR in HBM:
//   * row-wise 3-way merge  C<M,replace> = accum(C, T)   (the GraphBLAS write-back)
//   * row-wise union / intersection (eWiseAdd / eWiseMult)
//   * entry filters (select: tril/triu/diag/offdiag/value tests) and flag compaction
//   * value maps (apply, apply with a bound scalar), row reductions (reduce to vector)
// Entry-parallel throughout since round 2 (flags -> exclusive scan (rocPRIM) -> scatter; union / intersection by one stable
// merge of (row, column) keys): a power-law graph has rows of 10^5 entries, and kernels that give a row to one thread ran at
// 1-2 G entries/s on R-MAT-22.  The heavy lifting of the hot path is in grb_spgemm.hip / grb_spmv_kernels.hpp, not here.
#include "grb_api.hpp"
#include "grb_device.hpp"
#include "grb_atomics.hpp"
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

// ---- entry-parallel building blocks -------------------------------------------------------------------------------------------
// Round 1 merged, filtered and compacted ONE ROW PER THREAD: simple and order-preserving, but a power-law graph has rows of
// 10^5 entries, and a single thread walking one of them is the whole kernel — tril / select 68-100 ms, eWiseAdd 137 ms, a masked
// write-back 179 ms on the 1.3e8 entries of symmetric R-MAT-22 (1-2 G entries/s).  Everything below works per ENTRY:
//   row of an entry     the non-empty rows mark their first entry, an inclusive max-scan fills the rest;
//   compaction          exclusive scan of the keep flags = new position of every kept entry; new rowptr[r] = scan[rowptr[r]];
//   mask membership     every entry binary-searches its row of the mask;
//   union/intersection  (row, column) keys of the two operands — each already sorted — go through ONE merge (rocPRIM, stable:
//                       A before B on equal keys), equal neighbours are the intersection, run heads the union; the number of
//                       merged entries before row r is arp[r] + brp[r], so the new row pointers are a gather from the scan;
//   write-back          Z = accum(C, T) as a union, Z restricted to the mask by compaction, C's entries outside the mask kept
//                       (unless replace) by a second compaction, the two disjoint parts merged.
static __global__ void k_mark_row_starts(const uint32_t* __restrict__ rowptr, uint32_t nrows, uint32_t* __restrict__ rowidx) {
  for (uint64_t r = blockIdx.x * 256ull + threadIdx.x; r < nrows; r += (uint64_t)gridDim.x * 256ull) { const uint32_t s = rowptr[r]; if (rowptr[r + 1] > s) rowidx[s] = (uint32_t)r; }
}
void csr_row_indices(const DevCSR& A, uint32_t* rowidx) {
  if (!A.nnz) return;
  GRB_HIP(hipMemsetAsync(rowidx, 0, A.nnz * 4, stream()));
  hipLaunchKernelGGL(k_mark_row_starts, dim3(grid_n(A.nrows)), dim3(256), 0, stream(), A.rowptr.as<uint32_t>(), A.nrows, rowidx);
  inclusive_scan_max_u32(rowidx, rowidx, A.nnz);
}

static __global__ void k_keep_to_u32(const uint8_t* __restrict__ keep, uint64_t n, uint32_t* __restrict__ out) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i <= n; i += gridDim.x * 256ull) out[i] = (i < n && keep[i]) ? 1u : 0u;
}
static __global__ void k_gather_rowptr(const uint32_t* __restrict__ rowptr, uint32_t nrows, const uint32_t* __restrict__ pos, uint32_t* __restrict__ orp) {
  for (uint64_t r = blockIdx.x * 256ull + threadIdx.x; r <= nrows; r += (uint64_t)gridDim.x * 256ull) orp[r] = pos[rowptr[r]];
}
template <int TS> __global__ void k_compact_entries(uint64_t n, const uint8_t* __restrict__ keep, const uint32_t* __restrict__ pos, const uint32_t* __restrict__ col, const uint8_t* __restrict__ val,
                                                    uint32_t* __restrict__ ocol, uint8_t* __restrict__ oval) {
  typedef typename std::conditional<TS == 8, uint64_t, typename std::conditional<TS == 4, uint32_t, typename std::conditional<TS == 2, uint16_t, uint8_t>::type>::type>::type W;
  for (uint64_t p = blockIdx.x * 256ull + threadIdx.x; p < n; p += gridDim.x * 256ull) if (keep[p]) { const uint32_t w = pos[p]; ocol[w] = col[p]; ((W*)oval)[w] = ((const W*)val)[p]; }
}
void csr_compact(const DevCSR& A, const void* aval, size_t ts, const uint8_t* keep, DevCSR& out) {
  const uint32_t nrows = A.nrows; const uint64_t n = A.nnz;
  out.clear(); out.nrows = nrows; out.ncols = A.ncols;
  out.rowptr.alloc(((size_t)nrows + 1) * 4);
  DevBuf flags((n + 1) * 4 + 4), pos((n + 1) * 4 + 4);
  hipLaunchKernelGGL(k_keep_to_u32, dim3(grid_n(n + 1)), dim3(256), 0, stream(), keep, n, flags.as<uint32_t>());
  exclusive_scan_u32(flags.as<uint32_t>(), pos.as<uint32_t>(), n + 1);
  hipLaunchKernelGGL(k_gather_rowptr, dim3(grid_n((uint64_t)nrows + 1)), dim3(256), 0, stream(), A.rowptr.as<uint32_t>(), nrows, pos.as<uint32_t>(), out.rowptr.as<uint32_t>());
  uint32_t total = 0;
  GRB_HIP(hipMemcpyAsync(&total, pos.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
  out.nnz = total; out.col.alloc((size_t)total * 4 + 4); out.val.alloc((size_t)total * ts + 8);
  if (n) {
#define GRB_COMPACT(TS) hipLaunchKernelGGL((k_compact_entries<TS>), dim3(grid_n(n)), dim3(256), 0, stream(), n, keep, pos.as<uint32_t>(), A.col.as<uint32_t>(), (const uint8_t*)aval, \
                                           out.col.as<uint32_t>(), out.val.as<uint8_t>())
    switch (ts) { case 1: GRB_COMPACT(1); break; case 2: GRB_COMPACT(2); break; case 4: GRB_COMPACT(4); break; default: GRB_COMPACT(8); break; }
#undef GRB_COMPACT
  }
  GRB_HIP(hipGetLastError());
  GRB_HIP(hipStreamSynchronize(stream()));       // flags / pos return to the pool
  out.valid = true;
}

// positional select flags: keep entry (i,j) by its diagonal index j - i against k
static __global__ void k_select_positional(uint64_t n, const uint32_t* __restrict__ rowidx, const uint32_t* __restrict__ col, int sel, int64_t k, uint8_t* __restrict__ keep) {
  for (uint64_t p = blockIdx.x * 256ull + threadIdx.x; p < n; p += gridDim.x * 256ull) {
    const int64_t d = (int64_t)col[p] - (int64_t)rowidx[p]; bool kp;
    switch (sel) { case SEL_TRIL: kp = d <= k; break; case SEL_TRIU: kp = d >= k; break; case SEL_DIAG: kp = d == k; break; default: kp = d != k; }
    keep[p] = kp ? 1 : 0;
  }
}
void select_positional_flags(const DevCSR& A, int sel, int64_t k, uint8_t* keep) {
  if (!A.nnz) return;
  DevBuf rowidx(A.nnz * 4 + 4);
  csr_row_indices(A, rowidx.as<uint32_t>());
  hipLaunchKernelGGL(k_select_positional, dim3(grid_n(A.nnz)), dim3(256), 0, stream(), A.nnz, rowidx.as<uint32_t>(), A.col.as<uint32_t>(), sel, k, keep);
  GRB_HIP(hipStreamSynchronize(stream()));
}

// mask flags for entries of T: keep[p] = mask allows (i, col[p])  (INVERT: the entries the mask does NOT allow)
static __global__ void k_mask_flags(uint64_t n, const uint32_t* __restrict__ rowidx, const uint32_t* __restrict__ tcol, const uint32_t* __restrict__ mrp, const uint32_t* __restrict__ mcol,
                                    const void* __restrict__ mval, int mcode, bool mstruct, bool mcomp, bool invert, uint8_t* __restrict__ keep) {
  for (uint64_t p = blockIdx.x * 256ull + threadIdx.x; p < n; p += gridDim.x * 256ull) {
    const uint32_t i = rowidx[p], j = tcol[p];
    uint32_t lo = mrp[i], hi = mrp[i + 1];
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (mcol[mid] < j) lo = mid + 1; else hi = mid; }
    const bool m = lo < mrp[i + 1] && mcol[lo] == j && mask_truth(mval, mcode, lo, mstruct);
    keep[p] = ((m != mcomp) != invert) ? 1 : 0;
  }
}
static void mask_flags_ex(const DevCSR& Tm, const DevCSR& M, int mcode, bool mstruct, bool mcomp, bool invert, uint8_t* keep) {
  if (!Tm.nnz) return;
  DevBuf rowidx(Tm.nnz * 4 + 4);
  csr_row_indices(Tm, rowidx.as<uint32_t>());
  hipLaunchKernelGGL(k_mask_flags, dim3(grid_n(Tm.nnz)), dim3(256), 0, stream(), Tm.nnz, rowidx.as<uint32_t>(), Tm.col.as<uint32_t>(), M.rowptr.as<uint32_t>(), M.col.as<uint32_t>(), M.val.p, mcode,
                     mstruct, mcomp, invert, keep);
  GRB_HIP(hipStreamSynchronize(stream()));
}
void mask_flags(const DevCSR& Tm, const DevCSR& M, int mcode, bool mstruct, bool mcomp, uint8_t* keep) { mask_flags_ex(Tm, M, mcode, mstruct, mcomp, false, keep); }

// ---- element-wise union / intersection -----------------------------------------------------------------------------------------
void merge_pairs_u64(const uint64_t* k1, const uint64_t* k2, uint64_t* kout, const uint32_t* v1, const uint32_t* v2, uint32_t* vout, uint64_t n1, uint64_t n2);   // grb_prims.hip
static __global__ void k_entry_keys(uint64_t n, const uint32_t* __restrict__ rowidx, const uint32_t* __restrict__ col, uint32_t tag, unsigned long long* __restrict__ key, uint32_t* __restrict__ idx) {
  for (uint64_t p = blockIdx.x * 256ull + threadIdx.x; p < n; p += gridDim.x * 256ull) { key[p] = ((unsigned long long)rowidx[p] << 32) | col[p]; idx[p] = (uint32_t)p | tag; }
}
static __global__ void k_merge_emit(uint64_t n, const unsigned long long* __restrict__ key, bool is_union, uint32_t* __restrict__ e) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i <= n; i += gridDim.x * 256ull) {
    bool out = false;
    if (i < n) { const bool head = i == 0 || key[i] != key[i - 1], both = i + 1 < n && key[i + 1] == key[i]; out = head && (is_union || both); }
    e[i] = out ? 1u : 0u;
  }
}
static __global__ void k_merge_rowptr(const uint32_t* __restrict__ arp, const uint32_t* __restrict__ brp, uint32_t nrows, const uint32_t* __restrict__ pos, uint32_t* __restrict__ orp) {
  for (uint64_t r = blockIdx.x * 256ull + threadIdx.x; r <= nrows; r += (uint64_t)gridDim.x * 256ull) orp[r] = pos[(uint64_t)arp[r] + brp[r]];
}
template <class T, bool MATH> __global__ void k_merge_fill(uint64_t n, const unsigned long long* __restrict__ key, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ e,
                                                           const uint32_t* __restrict__ pos, const T* __restrict__ aval, const T* __restrict__ bval, int op, uint32_t* __restrict__ ocol, T* __restrict__ oval) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) if (e[i]) {
    const uint32_t w = pos[i]; const unsigned long long k = key[i]; const uint32_t x = idx[i];
    const bool both = i + 1 < n && key[i + 1] == k;
    T v;
    if (both) { const uint32_t x2 = idx[i + 1]; const uint32_t xa = (x & 0x80000000u) ? x2 : x, xb = (x & 0x80000000u) ? x : x2;     // (whichever of the two came first)
                v = apply_binop<T, true, MATH>(op, aval[xa & 0x7FFFFFFFu], bval[xb & 0x7FFFFFFFu]); }
    else v = (x & 0x80000000u) ? bval[x & 0x7FFFFFFFu] : aval[x];
    ocol[w] = (uint32_t)k; oval[w] = v;
  }
}
void csr_ewise(int code, const DevCSR& A, const void* aval, const DevCSR& B, const void* bval, int op, bool is_union, DevCSR& out) {
  const uint32_t nrows = A.nrows; const uint64_t na = A.nnz, nb = B.nnz, n = na + nb;
  if (na >= 0x7FFFFFF0ull || nb >= 0x7FFFFFF0ull) fail(GrB_INSUFFICIENT_SPACE, "eWise: more than 2^31 entries in one operand");
  out.clear(); out.nrows = nrows; out.ncols = A.ncols;
  out.rowptr.alloc(((size_t)nrows + 1) * 4);
  if (!n) { GRB_HIP(hipMemsetAsync(out.rowptr.p, 0, ((size_t)nrows + 1) * 4, stream())); out.nnz = 0; out.col.alloc(8); out.val.alloc(8); out.valid = true; return; }
  DevBuf ka(na * 8 + 8), kb(nb * 8 + 8), ia(na * 4 + 4), ib(nb * 4 + 4), km(n * 8 + 8), im(n * 4 + 4), e((n + 1) * 4 + 4), pos((n + 1) * 4 + 4);
  { DevBuf rowidx((na > nb ? na : nb) * 4 + 4);
    if (na) { csr_row_indices(A, rowidx.as<uint32_t>()); hipLaunchKernelGGL(k_entry_keys, dim3(grid_n(na)), dim3(256), 0, stream(), na, rowidx.as<uint32_t>(), A.col.as<uint32_t>(), 0u, (unsigned long long*)ka.p, ia.as<uint32_t>()); }
    if (nb) { csr_row_indices(B, rowidx.as<uint32_t>()); hipLaunchKernelGGL(k_entry_keys, dim3(grid_n(nb)), dim3(256), 0, stream(), nb, rowidx.as<uint32_t>(), B.col.as<uint32_t>(), 0x80000000u, (unsigned long long*)kb.p, ib.as<uint32_t>()); }
    GRB_HIP(hipStreamSynchronize(stream())); }
  merge_pairs_u64((const uint64_t*)ka.p, (const uint64_t*)kb.p, (uint64_t*)km.p, ia.as<uint32_t>(), ib.as<uint32_t>(), im.as<uint32_t>(), na, nb);
  hipLaunchKernelGGL(k_merge_emit, dim3(grid_n(n + 1)), dim3(256), 0, stream(), n, (const unsigned long long*)km.p, is_union, e.as<uint32_t>());
  exclusive_scan_u32(e.as<uint32_t>(), pos.as<uint32_t>(), n + 1);
  hipLaunchKernelGGL(k_merge_rowptr, dim3(grid_n((uint64_t)nrows + 1)), dim3(256), 0, stream(), A.rowptr.as<uint32_t>(), B.rowptr.as<uint32_t>(), nrows, pos.as<uint32_t>(), out.rowptr.as<uint32_t>());
  uint32_t total = 0;
  GRB_HIP(hipMemcpyAsync(&total, pos.as<uint32_t>() + n, 4, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
  out.nnz = total;
  dispatch_type(code, [&]<class T>() {
    out.col.alloc((size_t)total * 4 + 4); out.val.alloc((size_t)total * sizeof(T) + 8);
#define GRB_EW_FILL(MATH) hipLaunchKernelGGL((k_merge_fill<T, MATH>), dim3(grid_n(n)), dim3(256), 0, stream(), n, (const unsigned long long*)km.p, im.as<uint32_t>(), e.as<uint32_t>(), pos.as<uint32_t>(), \
                                             (const T*)aval, (const T*)bval, op, out.col.as<uint32_t>(), out.val.as<T>())
    if (binop_needs_math(op)) GRB_EW_FILL(true); else GRB_EW_FILL(false);
#undef GRB_EW_FILL
  });
  GRB_HIP(hipGetLastError());
  GRB_HIP(hipStreamSynchronize(stream()));       // the temporaries return to the pool
  out.valid = true;
}

// ---- write-back  out = C<M,replace> (+accum) T ------------------------------------------------------------------------------------
static void csr_copy(const DevCSR& S, size_t ts, DevCSR& out) {
  out.clear(); out.nrows = S.nrows; out.ncols = S.ncols; out.nnz = S.nnz;
  out.rowptr.alloc(((size_t)S.nrows + 1) * 4); out.col.alloc(S.nnz * 4 + 4); out.val.alloc(S.nnz * ts + 8);
  GRB_HIP(hipMemcpyAsync(out.rowptr.p, S.rowptr.p, ((size_t)S.nrows + 1) * 4, hipMemcpyDeviceToDevice, stream()));
  if (S.nnz) { GRB_HIP(hipMemcpyAsync(out.col.p, S.col.p, S.nnz * 4, hipMemcpyDeviceToDevice, stream())); GRB_HIP(hipMemcpyAsync(out.val.p, S.val.p, S.nnz * ts, hipMemcpyDeviceToDevice, stream())); }
  out.valid = true;
}
void csr_writeback(int code, uint32_t nrows, const DevCSR& C, const DevCSR& Tm, const DevCSR* M, int mcode, bool mstruct, bool mcomp,
                   bool replace, int accum, DevCSR& out) {
  (void)nrows;
  const size_t ts = (size_t)type_size(code);
  DevCSR Zacc; const DevCSR* Z = &Tm;
  if (accum >= 0) { csr_ewise(code, C, C.val.p, Tm, Tm.val.p, accum, true, Zacc); Z = &Zacc; }       // Z = accum(C, T) on the union of the patterns
  if (!M) {                                         // no mask: everything is allowed (the complemented no-mask case never gets here)
    if (Z == &Zacc) out = std::move(Zacc); else csr_copy(Tm, ts, out);
    return;
  }
  DevCSR Zk;
  { DevBuf keep(Z->nnz + 8); mask_flags_ex(*Z, *M, mcode, mstruct, mcomp, false, keep.as<uint8_t>()); csr_compact(*Z, Z->val.p, ts, keep.as<uint8_t>(), Zk); }
  if (replace || !C.nnz) { out = std::move(Zk); return; }
  DevCSR Ck;
  { DevBuf keep(C.nnz + 8); mask_flags_ex(C, *M, mcode, mstruct, mcomp, true, keep.as<uint8_t>()); csr_compact(C, C.val.p, ts, keep.as<uint8_t>(), Ck); }
  csr_ewise(code, Zk, Zk.val.p, Ck, Ck.val.p, B_FIRST, true, out);              // disjoint patterns: a plain merge
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


// ---- column-wise reduction of a stored-by-row matrix without its transpose (`bcu.reduce_vector(accum=PLUS, out=cent, desc=T0)` at
// the end of gap/bcmark.py: an ns x n batch — building the n x ns transpose by a sort was 60 ms of the 0.28 s algorithm) ----------
// Every entry combines into its column's accumulator with the monoid's atomic (native add / min / max, a CAS loop otherwise).
template <class T> __global__ void k_reduce_cols(uint64_t nnz, const uint32_t* __restrict__ col, const T* __restrict__ val, int op, typename acc_word<T>::type* __restrict__ acc, uint8_t* __restrict__ tpres) {
  for (uint64_t p = blockIdx.x * 256ull + threadIdx.x; p < nnz; p += gridDim.x * 256ull) {
    const uint32_t j = col[p];
    word_combine<T>(op, &acc[j], val[p]);
    if (!tpres[j]) tpres[j] = 1;
  }
}
template <class W> __global__ void k_fill_typed(W* p, uint64_t n, W v) { for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = v; }
template <class T> __global__ void k_words_to_values(uint64_t n, const typename acc_word<T>::type* __restrict__ acc, T* __restrict__ out) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) out[i] = from_word<T>(acc[i]);
}
// The same for a matrix of FEW rows (the ns x n batches of gap/bcmark.py), in a FIXED order: row after row — a row's entries have
// distinct columns, so a launch per row needs no atomics, and column j's terms are added in row order whatever the hardware does
// (a floating-point PLUS stays reproducible: ADVICE round 3).
template <class T> __global__ void k_reduce_cols_row(uint32_t e0, uint32_t e1, const uint32_t* __restrict__ col, const T* __restrict__ val, int op, T* __restrict__ acc, uint8_t* __restrict__ tpres) {
  for (uint64_t p = e0 + blockIdx.x * 256ull + threadIdx.x; p < e1; p += gridDim.x * 256ull) {
    const uint32_t j = col[p];
    acc[j] = tpres[j] ? apply_binop<T, false>(op, acc[j], val[p]) : val[p];
    tpres[j] = 1;
  }
}
bool csr_reduce_cols_few_rows(int code, const DevCSR& A, const void* aval, int op, void* tval, uint8_t* tpres) {
  if (A.nrows > 64 || (code != T_FP32 && code != T_FP64)) return false;
  if (!A.ncols) return true;
  std::vector<uint32_t> rp((size_t)A.nrows + 1);
  GRB_HIP(hipMemcpyAsync(rp.data(), A.rowptr.p, rp.size() * 4, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
  GRB_HIP(hipMemsetAsync(tpres, 0, A.ncols, stream()));
  dispatch_type(code, [&]<class T>() {
    if constexpr (std::is_same<T, float>::value || std::is_same<T, double>::value) {
      GRB_HIP(hipMemsetAsync(tval, 0, (size_t)A.ncols * sizeof(T), stream()));
      for (uint32_t r = 0; r < A.nrows; r++) {
        const uint32_t e0 = rp[r], e1 = rp[r + 1]; if (e1 == e0) continue;
        uint64_t g = ((uint64_t)(e1 - e0) + 255) / 256; if (g > 16384) g = 16384;
        hipLaunchKernelGGL((k_reduce_cols_row<T>), dim3((unsigned)g), dim3(256), 0, stream(), e0, e1, A.col.as<uint32_t>(), (const T*)aval, op, (T*)tval, tpres);
      }
    }
  });
  GRB_HIP(hipGetLastError());
  return true;
}
bool csr_reduce_cols(int code, const DevCSR& A, const void* aval, int op, const void* identity, void* tval, uint8_t* tpres) {
  if (code != T_INT32 && code != T_UINT32 && code != T_INT64 && code != T_UINT64 && code != T_FP32 && code != T_FP64) return false;
  if (!A.ncols) return true;
  GRB_HIP(hipMemsetAsync(tpres, 0, A.ncols, stream()));
  dispatch_type(code, [&]<class T>() {
    if constexpr (sizeof(T) >= 4 && !is_bool<T>::value) {
      typedef typename acc_word<T>::type W;                    // (4- and 8-byte types: the accumulator word IS the value)
      T id; memcpy(&id, identity, sizeof(T));
      uint64_t g = ((uint64_t)A.ncols + 255) / 256; if (g > 8192) g = 8192; if (g < 1) g = 1;
      hipLaunchKernelGGL((k_fill_typed<W>), dim3((unsigned)g), dim3(256), 0, stream(), (W*)tval, (uint64_t)A.ncols, to_word<T>(id));
      uint64_t ge = (A.nnz + 255) / 256; if (ge > 16384) ge = 16384; if (ge < 1) ge = 1;
      if (A.nnz) hipLaunchKernelGGL((k_reduce_cols<T>), dim3((unsigned)ge), dim3(256), 0, stream(), A.nnz, A.col.as<uint32_t>(), (const T*)aval, op, (W*)tval, tpres);
    }
  });
  GRB_HIP(hipGetLastError());
  return true;
}
// position of entry (i, j) in a device CSR, ~0 when it is not stored: one thread walks the row's sorted columns by bisection
static __global__ void k_find_entry(const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ col, uint32_t i, uint32_t j, unsigned long long* __restrict__ out) {
  uint32_t lo = rowptr[i], hi = rowptr[i + 1];
  while (lo < hi) { const uint32_t m = lo + (hi - lo) / 2; if (col[m] < j) lo = m + 1; else hi = m; }
  *out = (lo < rowptr[i + 1] && col[lo] == j) ? (unsigned long long)lo : ~0ull;
}
uint64_t csr_find_entry(const DevCSR& A, uint32_t i, uint32_t j) {
  DevBuf out(8);
  hipLaunchKernelGGL(k_find_entry, dim3(1), dim3(1), 0, stream(), A.rowptr.as<uint32_t>(), A.col.as<uint32_t>(), i, j, (unsigned long long*)out.p);
  unsigned long long* pin = (unsigned long long*)pinned_scratch();
  GRB_HIP(hipMemcpyAsync(pin, out.p, 8, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
  return pin[0];
}

// ---- a full one-valued matrix: `Matrix.dense(T, ns, n)` / `M[:, :] = x` (the ns x n batches of the BC sweeps, gap/bcmark.py:19-20, 48) ----
// Built in HBM by one kernel (round 2 built the three arrays with numpy and uploaded 134 MB per batch: half of the algorithm's time).
template <int TS> __global__ void k_dense_fill(uint32_t nrows, uint32_t ncols, uint64_t total, uint64_t lo, uint64_t hi, uint32_t* __restrict__ rowptr, uint32_t* __restrict__ col, uint8_t* __restrict__ val) {
  for (uint64_t e = blockIdx.x * 256ull + threadIdx.x; e < total; e += gridDim.x * 256ull) {
    col[e] = (uint32_t)(e % ncols);
    if constexpr (TS == 8) ((uint64_t*)val)[e] = lo;
    else if constexpr (TS == 4) ((uint32_t*)val)[e] = (uint32_t)lo;
    else if constexpr (TS == 2) ((uint16_t*)val)[e] = (uint16_t)lo;
    else val[e] = (uint8_t)lo;
    if (e <= nrows) rowptr[e] = (uint32_t)(e * ncols);
  }
  if (total <= nrows) for (uint64_t e = total + blockIdx.x * 256ull + threadIdx.x; e <= nrows; e += gridDim.x * 256ull) rowptr[e] = (uint32_t)(e * ncols);
  (void)hi;
}
void csr_dense_fill(uint32_t nrows, uint32_t ncols, const void* scalar, size_t ts, DevCSR& out) {
  const uint64_t total = (uint64_t)nrows * ncols;
  out.clear(); out.nrows = nrows; out.ncols = ncols; out.nnz = total;
  out.rowptr.alloc(((size_t)nrows + 1) * 4); out.col.alloc(total * 4 + 8); out.val.alloc(total * ts + 8);
  uint64_t lo = 0; memcpy(&lo, scalar, ts < 8 ? ts : 8);
  uint64_t g = ((total > nrows ? total : (uint64_t)nrows + 1) + 255) / 256; if (g < 1) g = 1; if (g > 8192) g = 8192;
#define GRB_DF(TS_) hipLaunchKernelGGL((k_dense_fill<TS_>), dim3((unsigned)g), dim3(256), 0, stream(), nrows, ncols, total, lo, 0ull, out.rowptr.as<uint32_t>(), out.col.as<uint32_t>(), out.val.as<uint8_t>())
  if (ts == 8) GRB_DF(8); else if (ts == 4) GRB_DF(4); else if (ts == 2) GRB_DF(2); else GRB_DF(1);
#undef GRB_DF
  GRB_HIP(hipGetLastError());
  out.valid = true;
}

// ---- positional unary operators (GxB_POSITIONI / I1 / J / J1, round 6; pygraphblas/unaryop.py:55-63): an entry's value becomes its row or column index -----
// which: 0 row, 1 row + 1, 2 column, 3 column + 1.  Z = int32_t / int64_t.
template <class Z> static __global__ void k_u32_to_index(const uint32_t* __restrict__ src, uint64_t n, Z add, Z* __restrict__ dst) {
  for (uint64_t e = blockIdx.x * 256ull + threadIdx.x; e < n; e += gridDim.x * 256ull) dst[e] = (Z)src[e] + add;
}
template <class Z> static __global__ void k_index_fill(uint64_t n, int which, Z* __restrict__ dst) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) dst[i] = which < 2 ? (Z)i + (Z)which : (Z)(which - 2);
}
void csr_position_values(int zcode, const DevCSR& A, int which, void* out) {
  if (!A.nnz) return;
  DevBuf rowidx; const uint32_t* src = A.col.as<uint32_t>();
  if (which < 2) { rowidx.alloc(A.nnz * 4 + 4); csr_row_indices(A, rowidx.as<uint32_t>()); src = rowidx.as<uint32_t>(); }
  const unsigned g = grid_n(A.nnz);
  if (zcode == T_INT32) hipLaunchKernelGGL((k_u32_to_index<int32_t>), dim3(g), dim3(256), 0, stream(), src, A.nnz, (int32_t)(which & 1), (int32_t*)out);
  else hipLaunchKernelGGL((k_u32_to_index<int64_t>), dim3(g), dim3(256), 0, stream(), src, A.nnz, (int64_t)(which & 1), (int64_t*)out);
  GRB_HIP(hipGetLastError());
}
// the same for a vector (an n x 1 column): row = the index, column = 0
void vec_position_values(int zcode, uint64_t n, int which, void* out) {
  if (!n) return;
  const unsigned g = grid_n(n);
  if (zcode == T_INT32) hipLaunchKernelGGL((k_index_fill<int32_t>), dim3(g), dim3(256), 0, stream(), n, which, (int32_t*)out);
  else hipLaunchKernelGGL((k_index_fill<int64_t>), dim3(g), dim3(256), 0, stream(), n, which, (int64_t*)out);
  GRB_HIP(hipGetLastError());
}


}  // namespace grb
