// grb_mxm_rows.cpp — GrB_mxm whose left operand has only a few rows: one vxm per row.
//
// The batched-frontier products of the reference's betweenness centrality (gap/bcmark.py:26-44, 55-58) are
//     frontier<!paths, replace> = frontier (+).first A          frontier: ns x n (ns = 4 sources), paths: DENSE ns x n
// A row-by-row Gustavson over ns output rows cannot fill 256 CUs, and with a complemented (here: dense, valued) mask the
// generic path expands every product before it looks at the mask (measured at R-MAT-22, ns = 4: 1.6-5.6 s per level).
// Each row of such a product is exactly the product the BFS / SSSP loops run — a vector times the matrix under a mask —
// so it goes through GrB_vxm itself: row s of op(A) becomes a bitmap vector, row s of the mask a bitmap mask vector
// (its values and the descriptor's complement / structure flags keep their meaning), the direction choice (push for a thin
// frontier, masked pull with early exit for a wide one) and every semiring come with it, and the ns result vectors are
// compacted into the CSR rows of T.  C<M,replace> = accum(C, T) then runs as for any other mxm.
#include "grb_opcommon.hpp"
#include "grb_matops.hpp"

extern "C" GrB_Info GrB_vxm(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Semiring semiring, const GrB_Vector u, const GrB_Matrix A,
                            const GrB_Descriptor desc);
extern "C" GrB_Info GrB_Vector_eWiseAdd_BinaryOp(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Vector u, const GrB_Vector v, const GrB_Descriptor desc);
extern "C" GrB_Info GrB_Vector_eWiseMult_BinaryOp(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Vector u, const GrB_Vector v, const GrB_Descriptor desc);

namespace grb {

namespace {
// entries of one CSR row -> bitmap (values are moved as raw bytes of the element size)
template <int TS> __global__ void k_row_to_bitmap(const uint32_t* __restrict__ col, const uint8_t* __restrict__ val, uint32_t cnt, uint8_t* __restrict__ dval, uint8_t* __restrict__ dpres) {
  for (uint32_t p = blockIdx.x * 256 + threadIdx.x; p < cnt; p += gridDim.x * 256) {
    const uint32_t c = col[p];
#pragma unroll
    for (int b = 0; b < TS; b++) dval[(size_t)c * TS + b] = val[(size_t)p * TS + b];
    dpres[c] = 1;
  }
}
__global__ void k_pres_to_u32(const uint8_t* __restrict__ pres, uint64_t n, uint32_t* __restrict__ out) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i <= n; i += gridDim.x * 256ull) out[i] = (i < n && pres[i]) ? 1u : 0u;
}
template <int TS> __global__ void k_bitmap_to_row(const uint8_t* __restrict__ pres, const uint8_t* __restrict__ val, const uint32_t* __restrict__ pos, uint64_t n, uint32_t base,
                                                  uint32_t* __restrict__ ocol, uint8_t* __restrict__ oval) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) if (pres[i]) {
    const size_t w = (size_t)base + pos[i]; ocol[w] = (uint32_t)i;
#pragma unroll
    for (int b = 0; b < TS; b++) oval[w * TS + b] = val[i * TS + b];
  }
}
unsigned grid_of(uint64_t n) { uint64_t b = (n + 255) / 256; if (b < 1) b = 1; if (b > 8192) b = 8192; return (unsigned)b; }
template <class F> void by_size(size_t ts, F&& f) {
  switch (ts) { case 1: f(std::integral_constant<int, 1>{}); break; case 2: f(std::integral_constant<int, 2>{}); break;
                case 4: f(std::integral_constant<int, 4>{}); break; default: f(std::integral_constant<int, 8>{}); }
}
GrB_Vector row_vector(const DevCSR& S, GrB_Type type, uint32_t r, const std::vector<uint32_t>& rp, uint64_t n) {
  GrB_Vector v = nullptr; if (GrB_Vector_new(&v, type, n) != GrB_SUCCESS) fail(GrB_OUT_OF_MEMORY, "mxm: row vector");
  const size_t ts = type->size;
  v->dval.alloc(n * ts + 8); v->dpres.alloc(n + 8);
  GRB_HIP(hipMemsetAsync(v->dpres.p, 0, n + 8, stream()));
  const uint32_t b = rp[r], cnt = rp[r + 1] - rp[r];
  if (cnt) by_size(ts, [&](auto TS) {
    hipLaunchKernelGGL((k_row_to_bitmap<decltype(TS)::value>), dim3(grid_of(cnt)), dim3(256), 0, stream(), S.col.as<uint32_t>() + b, (const uint8_t*)S.val.p + (size_t)b * ts, cnt,
                       (uint8_t*)v->dval.p, v->dpres.as<uint8_t>());
  });
  v->dev_valid = true; v->host_valid = false; v->dnvals = cnt; v->dnvals_known = true;
  return v;
}
struct VecGuard { std::vector<GrB_Vector> v; ~VecGuard() { for (auto& x : v) if (x) GrB_Vector_free(&x); } };
}  // namespace

// T = op(A) (+).(x) op(B) restricted by the mask, one GrB_vxm per row of op(A).  `Ad` = device CSR of op(A), of A's type.
bool mxm_few_rows_wanted(const DevCSR& Ad, const DevCSR& Bd) {
  if (getenv("GRB_MI355X_MXM_ROWS")) return atoi(getenv("GRB_MI355X_MXM_ROWS")) != 0;
  return Ad.nrows <= 64 && Bd.nnz >= (1u << 20) && Bd.ncols >= 65536u;
}

void mxm_few_rows(const DevCSR& Ad, GrB_Type atype, GrB_Matrix Mmask, const DescView& dv, GrB_Semiring semiring, GrB_Matrix B, int zcode, DevCSR& T) {
  const uint32_t nr = Ad.nrows; const uint64_t nin = Ad.ncols, nout = dv.tran1 ? B->nrows : B->ncols;
  const size_t zs = type_size(zcode);
  std::vector<uint32_t> arp((size_t)nr + 1), mrp;
  GRB_HIP(hipMemcpyAsync(arp.data(), Ad.rowptr.p, arp.size() * 4, hipMemcpyDeviceToHost, stream()));
  if (Mmask) { mat_to_device(Mmask); mrp.resize((size_t)nr + 1); GRB_HIP(hipMemcpyAsync(mrp.data(), Mmask->csr.rowptr.p, mrp.size() * 4, hipMemcpyDeviceToHost, stream())); }
  GRB_HIP(hipStreamSynchronize(stream()));
  // the descriptor of the per-row products: the mask flags and the B transpose carry over; outputs are fresh vectors
  GrB_Descriptor_opaque d{GRB_MAGIC, 0, (dv.mask_comp ? GrB_COMP : 0) | (dv.mask_struct ? GrB_STRUCTURE : 0), 0, dv.tran1 ? GrB_TRAN : 0, 0, 0, 0, 0.0, false, "mxm_rows"};
  GrB_Type ztype = type_by_code(zcode);
  VecGuard outs; outs.v.resize(nr, nullptr);
  std::vector<uint32_t> cnt(nr, 0);
  std::vector<DevBuf> pos(nr);
  std::string plans;
  for (uint32_t s = 0; s < nr; s++) {
    if (arp[s + 1] == arp[s]) continue;                               // an empty row of op(A) gives an empty row of T
    VecGuard tmp;
    tmp.v.push_back(row_vector(Ad, atype, s, arp, nin));
    GrB_Vector mv = nullptr;
    if (Mmask) { mv = row_vector(Mmask->csr, Mmask->type, s, mrp, nout); tmp.v.push_back(mv); }
    GrB_Vector w = nullptr; if (GrB_Vector_new(&w, ztype, nout) != GrB_SUCCESS) fail(GrB_OUT_OF_MEMORY, "mxm: row result");
    outs.v[s] = w;
    const GrB_Info info = GrB_vxm(w, mv, nullptr, semiring, tmp.v[0], B, &d);
    if (info != GrB_SUCCESS) fail(info, "mxm (row-wise): " + w->err);
    if (s == 0 || plans.empty()) plans = g_last_plan;
    vec_to_device(w);
    pos[s].alloc((nout + 1) * 4 + 4);
    DevBuf flags((nout + 1) * 4 + 4);
    hipLaunchKernelGGL(k_pres_to_u32, dim3(grid_of(nout + 1)), dim3(256), 0, stream(), w->dpres.as<uint8_t>(), nout, flags.as<uint32_t>());
    exclusive_scan_u32(flags.as<uint32_t>(), pos[s].as<uint32_t>(), nout + 1);
    GRB_HIP(hipMemcpyAsync(&cnt[s], pos[s].as<uint32_t>() + nout, 4, hipMemcpyDeviceToHost, stream()));
  }
  GRB_HIP(hipStreamSynchronize(stream()));
  std::vector<uint32_t> trp((size_t)nr + 1, 0);
  uint64_t total = 0; for (uint32_t s = 0; s < nr; s++) { trp[s] = (uint32_t)total; total += cnt[s]; }
  if (total > 0xFFFFFFF0ull) fail(GrB_INSUFFICIENT_SPACE, "mxm: result has more than 2^32 entries");
  trp[nr] = (uint32_t)total;
  T.clear(); T.nrows = nr; T.ncols = (uint32_t)nout; T.nnz = total;
  T.rowptr.alloc(((size_t)nr + 1) * 4); T.col.alloc(total * 4 + 8); T.val.alloc(total * zs + 8);
  GRB_HIP(hipMemcpyAsync(T.rowptr.p, trp.data(), trp.size() * 4, hipMemcpyHostToDevice, stream()));
  for (uint32_t s = 0; s < nr; s++) if (cnt[s]) by_size(zs, [&](auto TS) {
    GrB_Vector w = outs.v[s];
    hipLaunchKernelGGL((k_bitmap_to_row<decltype(TS)::value>), dim3(grid_of(nout)), dim3(256), 0, stream(), w->dpres.as<uint8_t>(), (const uint8_t*)w->dval.p, pos[s].as<uint32_t>(), nout,
                       trp[s], T.col.as<uint32_t>(), (uint8_t*)T.val.p);
  });
  GRB_HIP(hipGetLastError());
  GRB_HIP(hipStreamSynchronize(stream()));                            // (trp lives on the host stack of this call)
  T.valid = true;
  g_last_plan = "mxm_rows<" + std::to_string(nr) + " x vxm> first row: " + plans;
}


// ---- element-wise operations on matrices of a few very long rows ---------------------------------------------------------------
// The matrix eWise / write-back kernels merge one row per wave: right for graphs, hopeless for the ns x n batches of the BC
// sweeps (`bc.emult(paths, DIV, out=W, mask=S[i], desc=R)`, `paths.assign(frontier, accum=PLUS)`, gap/bcmark.py:41-58: 4 rows
// of 4 M entries took 1-4 s each).  Such a matrix is ns bitmap vectors: every row goes through the vector kernel of the same
// operation — mask, accumulator and replace included, so the result row is final — and the rows are compacted back into a CSR.
bool few_long_rows(uint64_t nrows, uint64_t ncols, uint64_t nnz) {
  if (getenv("GRB_MI355X_EWISE_ROWS")) return atoi(getenv("GRB_MI355X_EWISE_ROWS")) != 0;
  return nrows <= 64 && ncols >= 65536u && nnz >= (1u << 18);
}

void ewise_few_rows(GrB_Matrix C, GrB_Matrix Mmask, const DescView& dv, GrB_BinaryOp accum, GrB_BinaryOp op, const DevCSR& Ad, GrB_Type atype, const DevCSR& Bd, GrB_Type btype, bool is_union,
                    DevCSR& T) {
  const uint32_t nr = (uint32_t)C->nrows; const uint64_t n = C->ncols;
  const size_t cs = C->type->size;
  mat_to_device(C); if (Mmask) mat_to_device(Mmask);
  auto fetch = [&](const DevCSR& S) { std::vector<uint32_t> rp((size_t)nr + 1); GRB_HIP(hipMemcpyAsync(rp.data(), S.rowptr.p, rp.size() * 4, hipMemcpyDeviceToHost, stream())); return rp; };
  std::vector<uint32_t> arp = fetch(Ad), brp = fetch(Bd), crp = fetch(C->csr), mrp; if (Mmask) mrp = fetch(Mmask->csr);
  GRB_HIP(hipStreamSynchronize(stream()));
  GrB_Descriptor_opaque d{GRB_MAGIC, dv.replace ? GrB_REPLACE : 0, (dv.mask_comp ? GrB_COMP : 0) | (dv.mask_struct ? GrB_STRUCTURE : 0), 0, 0, 0, 0, 0, 0.0, false, "ewise_rows"};
  VecGuard outs; outs.v.resize(nr, nullptr);
  std::vector<uint32_t> cnt(nr, 0); std::vector<DevBuf> pos(nr);
  for (uint32_t s = 0; s < nr; s++) {
    VecGuard tmp;
    GrB_Vector va = row_vector(Ad, atype, s, arp, n), vb = row_vector(Bd, btype, s, brp, n); tmp.v.push_back(va); tmp.v.push_back(vb);
    GrB_Vector vm = nullptr; if (Mmask) { vm = row_vector(Mmask->csr, Mmask->type, s, mrp, n); tmp.v.push_back(vm); }
    GrB_Vector vc = row_vector(C->csr, C->type, s, crp, n); outs.v[s] = vc;
    const GrB_Info info = is_union ? GrB_Vector_eWiseAdd_BinaryOp(vc, vm, accum, op, va, vb, &d) : GrB_Vector_eWiseMult_BinaryOp(vc, vm, accum, op, va, vb, &d);
    if (info != GrB_SUCCESS) fail(info, "eWise (row-wise): " + vc->err);
    vec_to_device(vc);
    pos[s].alloc((n + 1) * 4 + 4);
    DevBuf flags((n + 1) * 4 + 4);
    hipLaunchKernelGGL(k_pres_to_u32, dim3(grid_of(n + 1)), dim3(256), 0, stream(), vc->dpres.as<uint8_t>(), n, flags.as<uint32_t>());
    exclusive_scan_u32(flags.as<uint32_t>(), pos[s].as<uint32_t>(), n + 1);
    GRB_HIP(hipMemcpyAsync(&cnt[s], pos[s].as<uint32_t>() + n, 4, hipMemcpyDeviceToHost, stream()));
  }
  GRB_HIP(hipStreamSynchronize(stream()));
  std::vector<uint32_t> trp((size_t)nr + 1, 0);
  uint64_t total = 0; for (uint32_t s = 0; s < nr; s++) { trp[s] = (uint32_t)total; total += cnt[s]; }
  if (total > 0xFFFFFFF0ull) fail(GrB_INSUFFICIENT_SPACE, "eWise: result has more than 2^32 entries");
  trp[nr] = (uint32_t)total;
  T.clear(); T.nrows = nr; T.ncols = (uint32_t)n; T.nnz = total;
  T.rowptr.alloc(((size_t)nr + 1) * 4); T.col.alloc(total * 4 + 8); T.val.alloc(total * cs + 8);
  GRB_HIP(hipMemcpyAsync(T.rowptr.p, trp.data(), trp.size() * 4, hipMemcpyHostToDevice, stream()));
  for (uint32_t s = 0; s < nr; s++) if (cnt[s]) by_size(cs, [&](auto TS) {
    GrB_Vector w = outs.v[s];
    hipLaunchKernelGGL((k_bitmap_to_row<decltype(TS)::value>), dim3(grid_of(n)), dim3(256), 0, stream(), w->dpres.as<uint8_t>(), (const uint8_t*)w->dval.p, pos[s].as<uint32_t>(), n, trp[s],
                       T.col.as<uint32_t>(), (uint8_t*)T.val.p);
  });
  GRB_HIP(hipGetLastError());
  GRB_HIP(hipStreamSynchronize(stream()));
  T.valid = true;
  g_last_plan = "ewise_rows<" + std::to_string(nr) + " x vector eWise> ";
}


// ---- batch matrices as bitmaps (round 6) ------------------------------------------------------------------------------------------
// Rounds 3-5 ran every operation of the BC sweeps row by row: each of the ns rows of every operand was scattered from the CSR into a fresh
// bitmap vector, the vector kernel ran, and the result was compacted back into CSR rows behind a host round trip — 15 of the driver's 23 ms
// at R-MAT-22.  An ns x n batch IS a bitmap vector of ns * n positions: element-wise operations with mask,
// accumulator and replace are ONE vector kernel over the flattened arrays, a row of a product is a slice of them, and nothing in the loop
// ever needs the CSR of `paths` / `bc` / `W`.  So a batch result now stays a bitmap (GrB_Matrix_opaque::bm, the only valid form until
// something else asks for the CSR: mat_to_device), inputs are read from their bitmaps (made once from the CSR when they arrive as one).
namespace {
template <int TS> __global__ void k_csr_to_bitmap(const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ col, const uint8_t* __restrict__ val, uint32_t nrows, uint64_t ncols, uint64_t nnz,
                                                  uint8_t* __restrict__ dval, uint8_t* __restrict__ dpres) {
  __shared__ uint32_t rp[66];
  if (threadIdx.x <= nrows) rp[threadIdx.x] = rowptr[threadIdx.x];
  __syncthreads();
  for (uint64_t p = blockIdx.x * 256ull + threadIdx.x; p < nnz; p += gridDim.x * 256ull) {
    uint32_t r = 0; for (uint32_t j = 1; j < nrows; j++) r = p >= rp[j] ? j : r;        // (<= 64 rows)
    const uint64_t i = (uint64_t)r * ncols + col[p];
#pragma unroll
    for (int b = 0; b < TS; b++) dval[i * TS + b] = val[p * TS + b];
    dpres[i] = 1;
  }
}
__global__ void k_bm_rowptr(const uint32_t* __restrict__ pos, uint32_t nrows, uint64_t ncols, uint32_t* __restrict__ rowptr) {
  if (threadIdx.x <= nrows) rowptr[threadIdx.x] = pos[(uint64_t)threadIdx.x * ncols];
}
template <int TS> __global__ void k_bm_compact(const uint8_t* __restrict__ pres, const uint8_t* __restrict__ val, const uint32_t* __restrict__ pos, uint64_t np, uint64_t ncols,
                                               uint32_t* __restrict__ ocol, uint8_t* __restrict__ oval) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < np; i += gridDim.x * 256ull) if (pres[i]) {
    const size_t w = pos[i]; ocol[w] = (uint32_t)(i % ncols);
#pragma unroll
    for (int b = 0; b < TS; b++) oval[w * TS + b] = val[i * TS + b];
  }
}
// a vector object over (a slice of) a bitmap: `own` moves the buffers in (the op may replace them; take them back with vec_release), else they are borrowed
GrB_Vector view_vector(GrB_Type type, uint64_t n, void* val, void* pres, bool known, uint64_t nvals) {
  GrB_Vector v = nullptr; if (GrB_Vector_new(&v, type, n) != GrB_SUCCESS) fail(GrB_OUT_OF_MEMORY, "batch: vector view");
  v->dval.borrow(val, n * type->size); v->dpres.borrow(pres, n);
  v->dev_valid = true; v->host_valid = false; v->dnvals = nvals; v->dnvals_known = known;
  return v;
}
}  // namespace

bool mat_batch_shape(uint64_t nrows, uint64_t ncols, int type_code) {
  return nrows >= 1 && nrows <= 64 && ncols >= 65536u && (ncols & 63u) == 0 && nrows * ncols <= 0xFFFFFFF0ull && type_code < T_FC32 && device_ok();
}
DevBitmap& mat_bitmap(GrB_Matrix A) {
  if (A->bm.valid) return A->bm;
  mat_to_device(A);
  const DevCSR& S = A->csr; const size_t ts = A->type->size; const uint64_t np = (uint64_t)A->nrows * A->ncols;
  A->bm.val.alloc(np * ts + 64); A->bm.pres.alloc(np + 64);
  GRB_HIP(hipMemsetAsync(A->bm.pres.p, 0, np + 64, stream()));
  if (S.nnz) by_size(ts, [&](auto TS) {
    hipLaunchKernelGGL((k_csr_to_bitmap<decltype(TS)::value>), dim3(grid_of(S.nnz)), dim3(256), 0, stream(), S.rowptr.as<uint32_t>(), S.col.as<uint32_t>(), (const uint8_t*)S.val.p, (uint32_t)A->nrows,
                       (uint64_t)A->ncols, (uint64_t)S.nnz, (uint8_t*)A->bm.val.p, A->bm.pres.as<uint8_t>());
  });
  A->bm.valid = true; A->bm.nvals = S.nnz; A->bm.nvals_known = true;
  return A->bm;
}
uint64_t mat_bitmap_nvals(GrB_Matrix A) {
  if (!A->bm.nvals_known) { A->bm.nvals = count_present(A->bm.pres.as<uint8_t>(), (uint64_t)A->nrows * A->ncols); A->bm.nvals_known = true; }
  return A->bm.nvals;
}
void mat_bitmap_to_csr(GrB_Matrix A) {
  const uint64_t np = (uint64_t)A->nrows * A->ncols; const size_t ts = A->type->size;
  DevBuf flags((np + 1) * 4 + 4), pos((np + 1) * 4 + 4);
  hipLaunchKernelGGL(k_pres_to_u32, dim3(grid_of(np + 1)), dim3(256), 0, stream(), A->bm.pres.as<uint8_t>(), np, flags.as<uint32_t>());
  exclusive_scan_u32(flags.as<uint32_t>(), pos.as<uint32_t>(), np + 1);
  uint32_t nnz = 0;
  GRB_HIP(hipMemcpyAsync(&nnz, pos.as<uint32_t>() + np, 4, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
  DevCSR& c = A->csr; c.clear(); A->csc.clear();
  c.nrows = (uint32_t)A->nrows; c.ncols = (uint32_t)A->ncols; c.nnz = nnz;
  c.rowptr.alloc(((size_t)c.nrows + 1) * 4); c.col.alloc((size_t)nnz * 4 + 8); c.val.alloc((size_t)nnz * ts + 8);
  hipLaunchKernelGGL(k_bm_rowptr, dim3(1), dim3(128), 0, stream(), pos.as<uint32_t>(), c.nrows, (uint64_t)A->ncols, c.rowptr.as<uint32_t>());
  if (nnz) by_size(ts, [&](auto TS) {
    hipLaunchKernelGGL((k_bm_compact<decltype(TS)::value>), dim3(grid_of(np)), dim3(256), 0, stream(), A->bm.pres.as<uint8_t>(), (const uint8_t*)A->bm.val.p, pos.as<uint32_t>(), np, (uint64_t)A->ncols,
                       c.col.as<uint32_t>(), (uint8_t*)c.val.p);
  });
  GRB_HIP(hipGetLastError());
  c.valid = true; A->dev_valid = true; A->bm.nvals = nnz; A->bm.nvals_known = true;
}

// C becomes the bitmap (val, pres): the batch operations' way of writing a result
static void adopt_bitmap(GrB_Matrix C, DevBuf&& val, DevBuf&& pres, bool known, uint64_t nvals) {
  mat_invalidate_host(C); C->csc.clear(); C->csr.clear(); C->dev_valid = false; C->iso_full = false;
  C->bm.val = std::move(val); C->bm.pres = std::move(pres); C->bm.valid = true; C->bm.nvals = nvals; C->bm.nvals_known = known;
}

bool batch_wanted(GrB_Matrix C, uint64_t work) {
  if (!mat_batch_shape(C->nrows, C->ncols, C->type->code)) return false;
  if (getenv("GRB_MI355X_BATCH")) return atoi(getenv("GRB_MI355X_BATCH")) != 0;
  return work >= (1u << 18);
}

// C<M, replace> = accum(C, A op B) on batch matrices, element-wise: ONE vector kernel over the nrows * ncols positions
void ewise_batch(GrB_Matrix C, GrB_Matrix Mmask, const DescView& dv, GrB_BinaryOp accum, GrB_BinaryOp op, GrB_Matrix A, GrB_Matrix B, bool is_union) {
  const uint64_t np = (uint64_t)C->nrows * C->ncols;
  if (A != C) mat_bitmap(A); if (B != C) mat_bitmap(B); if (Mmask && Mmask != C) mat_bitmap(Mmask);
  DevBitmap& cb = mat_bitmap(C);                                      // (an empty C: a cleared presence array)
  VecGuard g;
  // the output owns C's buffers for the call (the vector write-back may swap them for the result's); the inputs borrow theirs
  GrB_Vector vc = nullptr; if (GrB_Vector_new(&vc, C->type, np) != GrB_SUCCESS) fail(GrB_OUT_OF_MEMORY, "batch: output view"); g.v.push_back(vc);
  vc->dval = std::move(cb.val); vc->dpres = std::move(cb.pres); vc->dev_valid = true; vc->host_valid = false; vc->dnvals = cb.nvals; vc->dnvals_known = cb.nvals_known;
  cb.valid = false;
  auto view = [&](GrB_Matrix X) -> GrB_Vector {
    if (X == C) return vc;
    GrB_Vector v = view_vector(X->type, np, X->bm.val.p, X->bm.pres.p, X->bm.nvals_known, X->bm.nvals); g.v.push_back(v); return v;
  };
  GrB_Vector va = view(A), vb = B == A ? va : view(B), vm = !Mmask ? nullptr : (Mmask == A ? va : (Mmask == B ? vb : view(Mmask)));
  GrB_Descriptor_opaque d{GRB_MAGIC, dv.replace ? GrB_REPLACE : 0, (dv.mask_comp ? GrB_COMP : 0) | (dv.mask_struct ? GrB_STRUCTURE : 0), 0, 0, 0, 0, 0, 0.0, false, "ewise_batch"};
  const GrB_Info info = is_union ? GrB_Vector_eWiseAdd_BinaryOp(vc, vm, accum, op, va, vb, &d) : GrB_Vector_eWiseMult_BinaryOp(vc, vm, accum, op, va, vb, &d);
  if (info != GrB_SUCCESS) { std::string e = vc->err; fail(info, "eWise (batch): " + e); }
  vec_to_device(vc);                                                  // (completes whatever the non-blocking queue deferred; an empty result gets its cleared bitmap)
  if (vc->dval.borrowed || vc->dpres.borrowed || vc->dval.bytes < np * C->type->size || vc->dpres.bytes < np) fail(GrB_PANIC, "eWise (batch): the result does not own its buffers");
  adopt_bitmap(C, std::move(vc->dval), std::move(vc->dpres), vc->dnvals_known, vc->dnvals);
  g_last_plan = "ewise_batch<" + std::to_string(C->nrows) + " x " + std::to_string(C->ncols) + " as one vector> ";
}

// C = f(A) on a batch matrix, no mask, no accumulator (`frontier.apply(BOOL.ONE, out=s)`, gap/bcmark.py:38-39): the values in one pass, the pattern copied
void apply_batch(GrB_Matrix C, int mode, int opcode, int xcode, const uint8_t* scalar16, GrB_Matrix A) {
  const uint64_t np = (uint64_t)A->nrows * A->ncols;
  DevBitmap& ab = mat_bitmap(A);
  DevBuf ac; const void* av = cast_values(xcode, A->type->code, ab.val.p, np, ac);
  DevBuf out(np * type_size(xcode) + 64), pres(np + 64);
  vec_apply(xcode, np, av, nullptr, mode, opcode, scalar16, out.p, nullptr);
  GRB_HIP(hipMemcpyAsync(pres.p, ab.pres.p, np, hipMemcpyDeviceToDevice, stream()));
  if (xcode != C->type->code) { DevBuf c(np * C->type->size + 64); vec_cast_values(C->type->code, c.p, xcode, out.p, np); out = std::move(c); }
  const bool known = ab.nvals_known; const uint64_t nv = ab.nvals;
  adopt_bitmap(C, std::move(out), std::move(pres), known, nv);
  g_last_plan = "apply_batch ";
}

extern thread_local void* g_mxv_dest_val; extern thread_local uint8_t* g_mxv_dest_pres;      // grb_mxv.cpp
// T = A (+).(x) op(B) under the mask for a batch A (and mask): one GrB_vxm per row on SLICES of the bitmaps, results into the rows of a new bitmap.
// Writes C itself when the write-back is "C becomes T" (no accumulator; replace, no mask, or an empty C) and returns true; otherwise leaves T as a CSR for
// the general write-back and returns false.
bool mxm_batch(GrB_Matrix C, GrB_Matrix A, GrB_Matrix Mmask, const DescView& dv, GrB_BinaryOp accum, GrB_Semiring semiring, GrB_Matrix B, int zcode, DevCSR& T) {
  const uint32_t nr = (uint32_t)A->nrows; const uint64_t nin = A->ncols, nout = dv.tran1 ? B->nrows : B->ncols;
  const size_t zs = type_size(zcode); GrB_Type ztype = type_by_code(zcode);
  DevBitmap& ab = mat_bitmap(A);
  if (Mmask) mat_bitmap(Mmask);
  GrB_Descriptor_opaque d{GRB_MAGIC, 0, (dv.mask_comp ? GrB_COMP : 0) | (dv.mask_struct ? GrB_STRUCTURE : 0), 0, dv.tran1 ? GrB_TRAN : 0, 0, 0, 0, 0.0, false, "mxm_batch"};
  const uint64_t npo = (uint64_t)nr * nout;
  DevBuf tval(npo * zs + 64), tpres(npo + 64);
  std::string plans; uint64_t total = 0; bool known = true;
  for (uint32_t s = 0; s < nr; s++) {
    VecGuard tmp;
    GrB_Vector u = view_vector(A->type, nin, (uint8_t*)ab.val.p + (size_t)s * nin * A->type->size, ab.pres.as<uint8_t>() + (size_t)s * nin, false, 0); tmp.v.push_back(u);
    GrB_Vector mv = nullptr;
    if (Mmask) { mv = view_vector(Mmask->type, nout, (uint8_t*)Mmask->bm.val.p + (size_t)s * nout * Mmask->type->size, Mmask->bm.pres.as<uint8_t>() + (size_t)s * nout, false, 0); tmp.v.push_back(mv); }
    GrB_Vector w = nullptr; if (GrB_Vector_new(&w, ztype, nout) != GrB_SUCCESS) fail(GrB_OUT_OF_MEMORY, "mxm: row result"); tmp.v.push_back(w);
    uint8_t* const row_val = (uint8_t*)tval.p + (size_t)s * nout * zs; uint8_t* const row_pres = tpres.as<uint8_t>() + (size_t)s * nout;
    g_mxv_dest_val = row_val; g_mxv_dest_pres = row_pres;             // the product writes T's row in place (grb_mxv.cpp)
    const GrB_Info info = GrB_vxm(w, mv, nullptr, semiring, u, B, &d);
    g_mxv_dest_val = nullptr; g_mxv_dest_pres = nullptr;
    if (info != GrB_SUCCESS) { std::string e = w->err; fail(info, "mxm (batch): " + e); }
    if (plans.empty()) plans = g_last_plan;
    vec_to_device(w);
    if (w->dval.p != row_val) GRB_HIP(hipMemcpyAsync(row_val, w->dval.p, nout * zs, hipMemcpyDeviceToDevice, stream()));        // (the write-back did more than adopt T: e.g. an empty product)
    if (w->dpres.p != row_pres) GRB_HIP(hipMemcpyAsync(row_pres, w->dpres.p, nout, hipMemcpyDeviceToDevice, stream()));
    if (w->dnvals_known) total += w->dnvals; else known = false;
  }
  g_last_plan = "mxm_batch<" + std::to_string(nr) + " x vxm on bitmap rows> first row: " + plans;
  const bool c_becomes_t = !accum && (!Mmask || dv.replace || mat_nvals(C) == 0) && mat_batch_shape(C->nrows, C->ncols, C->type->code);
  if (c_becomes_t) {
    if (zcode != C->type->code) { DevBuf c(npo * C->type->size + 64); vec_cast_values(C->type->code, c.p, zcode, tval.p, npo); tval = std::move(c); }
    adopt_bitmap(C, std::move(tval), std::move(tpres), known, total);
    return true;
  }
  // the general write-back wants a CSR: a temporary matrix object carries the bitmap through the conversion
  GrB_Matrix_opaque tmpm; tmpm.type = ztype; tmpm.nrows = nr; tmpm.ncols = nout; tmpm.host_valid = false;
  tmpm.bm.val = std::move(tval); tmpm.bm.pres = std::move(tpres); tmpm.bm.valid = true;
  mat_bitmap_to_csr(&tmpm);
  T.clear(); T.nrows = nr; T.ncols = (uint32_t)nout; T.nnz = tmpm.csr.nnz;
  T.rowptr = std::move(tmpm.csr.rowptr); T.col = std::move(tmpm.csr.col); T.val = std::move(tmpm.csr.val); T.valid = true;
  return false;
}

}  // namespace grb
