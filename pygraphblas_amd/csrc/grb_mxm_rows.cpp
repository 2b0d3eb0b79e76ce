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
#include "grb_semiring.hpp"
#include "grb_spmv.hpp"

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
  // (C's buffers travel inside vc for the call: should the operation fail, C is left a valid, EMPTY matrix — the C API leaves an output's content undefined after an
  //  error, not the object)
  auto c_left_empty = [&] { C->hi.clear(); C->hj.clear(); C->hx.clear(); C->pending.clear(); C->host_valid = true; C->iso_full = false; mat_invalidate_device(C); };
  GrB_Info info;
  try { info = is_union ? GrB_Vector_eWiseAdd_BinaryOp(vc, vm, accum, op, va, vb, &d) : GrB_Vector_eWiseMult_BinaryOp(vc, vm, accum, op, va, vb, &d); if (info == GrB_SUCCESS) vec_to_device(vc); }
  catch (...) { c_left_empty(); throw; }
  if (info != GrB_SUCCESS) { std::string e = vc->err; c_left_empty(); fail(info, "eWise (batch): " + e); }
  if (vc->dval.borrowed || vc->dpres.borrowed || vc->dval.bytes < np * C->type->size || vc->dpres.bytes < np) { c_left_empty(); fail(GrB_PANIC, "eWise (batch): the result does not own its buffers"); }
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


// ---- the batch product as ONE pull pass over the matrix (round 6) ---------------------------------------------------------------
// ns masked pulls of the same matrix read it ns times (the bitmap version of the BC driver at R-MAT-22: 20 launches of k_spmv_adaptive, 535 us each =
// 10.7 of 13.5 ms).  All ns rows of the batch are multiplied in ONE pass: the operand rows are interleaved position-major — u(i, 0..NSP) side by side,
// one 16- or 32-byte load per matrix entry, and one byte per position whose bit s says "row s has an entry at i" — the mask rows become one byte per
// output position (bit s: row s may be written there; a position no row may write is skipped before a matrix byte is read: the late levels of a sweep cost
// what their unvisited rows cost), a 16-lane group adds a row of the pull operand for all ns accumulators, rows beyond SPB_LONG entries go to a second
// launch with one workgroup each.  Sums are formed in a fixed order => reproducible.
constexpr uint32_t SPB_LONG = SPMV_NNZ;         // rows beyond a stream block of the row-block plan
template <class T, int NSP> __global__ void k_spb_interleave(const T* __restrict__ uval, const uint8_t* __restrict__ upres, uint32_t ns, uint64_t n, T* __restrict__ ui, uint8_t* __restrict__ upm) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
    uint32_t m = 0;
#pragma unroll
    for (uint32_t sidx = 0; sidx < (uint32_t)NSP; sidx++) {
      const bool in = sidx < ns; const bool pr = in && upres[(uint64_t)sidx * n + i] != 0;
      T v; __builtin_memset(&v, 0, sizeof(T)); if (pr) v = uval[(uint64_t)sidx * n + i];
      ui[i * NSP + sidx] = v; m |= pr ? (1u << sidx) : 0u;
    }
    upm[i] = (uint8_t)m;
  }
}
// bit s of allowm[j]: row s may be written at j — (present && (structural || true-valued)) != complement; no mask: every row
__global__ void k_spb_allow(const uint8_t* __restrict__ mbool, const uint8_t* __restrict__ mpres, uint32_t ns, uint64_t n, int structural, int comp, uint8_t* __restrict__ allowm) {
  for (uint64_t j = blockIdx.x * 256ull + threadIdx.x; j < n; j += gridDim.x * 256ull) {
    uint32_t m = 0;
    for (uint32_t sidx = 0; sidx < ns; sidx++) {
      bool a = true;
      if (mpres) { const uint64_t q = (uint64_t)sidx * n + j; a = (mpres[q] != 0 && (structural || mbool[q] != 0)) != (comp != 0); }
      m |= a ? (1u << sidx) : 0u;
    }
    allowm[j] = (uint8_t)m;
  }
}
// the long rows some batch row may write, cut into PARTS of SPB_PART entries: rec[r] = (row, first item, parts), item[i] = r.  cnt[0] = rows, cnt[1] = items.
// (The slots come from atomics: which slot a row gets differs from run to run, what is computed for it does not.)
constexpr uint32_t SPB_PART = 8192;
__global__ void k_spb_long_rows(const uint32_t* __restrict__ rowptr, const uint8_t* __restrict__ allowm, uint32_t nrows, uint32_t* __restrict__ cnt, uint32_t* __restrict__ rec, uint32_t* __restrict__ item) {
  for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < nrows; j += gridDim.x * 256) {
    const uint32_t len = rowptr[j + 1] - rowptr[j];
    if (len > SPB_LONG && allowm[j]) {
      const uint32_t np = (len + SPB_PART - 1) / SPB_PART, r = atomicAdd(&cnt[0], 1u), base = atomicAdd(&cnt[1], np);
      rec[3 * r] = j; rec[3 * r + 1] = base; rec[3 * r + 2] = np;
      for (uint32_t q = 0; q < np; q++) item[base + q] = r;
    }
  }
}
template <class T, class SR, int NSP> __device__ __forceinline__ void spb_entry(const SR& sr, T a, const T* __restrict__ ui, const uint8_t* __restrict__ upm, uint32_t c, uint32_t am, T (&acc)[NSP], uint32_t& has) {
  const uint32_t pm = (uint32_t)upm[c] & am;
  if (!pm) return;
  T u[NSP];
#pragma unroll
  for (int sidx = 0; sidx < NSP; sidx++) u[sidx] = ui[(size_t)c * NSP + sidx];
#pragma unroll
  for (int sidx = 0; sidx < NSP; sidx++) if (pm & (1u << sidx)) { const T p = sr.mult(a, u[sidx]); acc[sidx] = (has & (1u << sidx)) ? sr.add(acc[sidx], p) : p; }
  has |= pm;
}
// Stream blocks of the pull operand's row-block plan (grb_spmv.hip: consecutive rows holding <= SPMV_NNZ entries): phase 1 reads the block's entries fully
// coalesced — 8 column loads, then 8 presence-byte gathers, then the interleaved operand vectors of the entries some row holds, all in flight before the first
// use — and leaves the NSP products of every entry in LDS; phase 2 gives G lanes to a row (G = the largest power of two with rows * G <= 256), which add their
// strided share per batch row under the output position's allow bits and combine in a fixed shuffle tree.  (The first version — a 16-lane group walking one
// row after the other, four dependent round trips per row — took 1.3 ms per pass at R-MAT-22: as long as the four single-row pulls it replaced.)
template <class T, class SR, int NSP>
__global__ __launch_bounds__(SPMV_THREADS) void k_spb_blocks(const SpmvBlock* __restrict__ blocks, const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ col, const T* __restrict__ aval, uint64_t n,
                                                            const T* __restrict__ ui, const uint8_t* __restrict__ upm, const uint8_t* __restrict__ allowm, T* __restrict__ tval, uint8_t* __restrict__ tpres,
                                                            const SR sr) {
  __shared__ T s_prod[NSP][SPMV_NNZ];
  __shared__ uint8_t s_pm[SPMV_NNZ];
  const int tid = threadIdx.x;
  const SpmvBlock b = blocks[blockIdx.x];
  if (b.nparts != 0) return;                                   // a part of a long row: k_spb_pull_long's
  const uint32_t r0 = b.row, r1 = b.aux;
  {
    int any = 0;
    for (uint32_t r = r0 + tid; r < r1; r += SPMV_THREADS) any |= allowm[r];
    if (!__syncthreads_or(any)) return;                        // no row of the batch may write any of these positions: no matrix traffic (tpres was cleared)
  }
  const uint32_t p0 = rowptr[r0], p1 = rowptr[r1], cnt = p1 - p0;
  if (cnt) {
    uint32_t c[SPMV_UNROLL]; T av[SPMV_UNROLL]; uint8_t pm[SPMV_UNROLL]; T uv[SPMV_UNROLL][NSP];
    const bool use_a = aval != nullptr;
#pragma unroll
    for (int u = 0; u < SPMV_UNROLL; u++) {
      const uint32_t k = tid + u * SPMV_THREADS, p = p0 + (k < cnt ? k : cnt - 1);
      c[u] = col[p]; av[u] = use_a ? aval[p] : T();
    }
#pragma unroll
    for (int u = 0; u < SPMV_UNROLL; u++) pm[u] = upm[c[u]];
#pragma unroll
    for (int u = 0; u < SPMV_UNROLL; u++) {
      if (pm[u]) {
#pragma unroll
        for (int sidx = 0; sidx < NSP; sidx++) uv[u][sidx] = ui[(size_t)c[u] * NSP + sidx];
      } else {
#pragma unroll
        for (int sidx = 0; sidx < NSP; sidx++) uv[u][sidx] = T();
      }
    }
#pragma unroll
    for (int u = 0; u < SPMV_UNROLL; u++) {
      const uint32_t k = tid + u * SPMV_THREADS;
      if (k < cnt) {
        s_pm[k] = pm[u];
#pragma unroll
        for (int sidx = 0; sidx < NSP; sidx++) s_prod[sidx][k] = sr.mult(av[u], uv[u][sidx]);
      }
    }
  }
  __syncthreads();
  const uint32_t nr = r1 - r0;
  const uint32_t G = nr >= SPMV_THREADS / 2 ? 1u : (nr <= 4 ? 64u : (1u << (31 - __builtin_clz(SPMV_THREADS / nr))));
  const uint32_t lane = tid & (G - 1), grp = tid / G, ngrp = SPMV_THREADS / G;
  for (uint32_t rb = 0; rb < nr; rb += ngrp) {
    const uint32_t r = r0 + rb + grp;
    const bool live = rb + grp < nr;
    const uint32_t am = live ? (uint32_t)allowm[r] : 0u;
    uint32_t qb = 0, qe = 0;
    if (am) { qb = rowptr[r] - p0; qe = rowptr[r + 1] - p0; }
    T acc[NSP]; uint32_t has = 0;
#pragma unroll
    for (int sidx = 0; sidx < NSP; sidx++) acc[sidx] = sr.identity;
    for (uint32_t q = qb + lane; q < qe; q += G) {
      const uint32_t pm = (uint32_t)s_pm[q] & am;
#pragma unroll
      for (int sidx = 0; sidx < NSP; sidx++) if (pm & (1u << sidx)) acc[sidx] = (has & (1u << sidx)) ? sr.add(acc[sidx], s_prod[sidx][q]) : s_prod[sidx][q];
      has |= pm;
    }
    if (G > 1) {
      for (uint32_t d = G >> 1; d >= 1; d >>= 1) {
        const uint32_t oh = (uint32_t)__shfl_down((int)has, (int)d, 64);
#pragma unroll
        for (int sidx = 0; sidx < NSP; sidx++) {
          const T ov = shfl_down_t<T>(acc[sidx], (int)d);
          if (oh & (1u << sidx)) acc[sidx] = (has & (1u << sidx)) ? sr.add(acc[sidx], ov) : ov;
        }
        has |= oh;
      }
    }
    if (live && lane == 0) {
#pragma unroll
      for (int sidx = 0; sidx < NSP; sidx++) if (has & (1u << sidx)) { tval[(uint64_t)sidx * n + r] = acc[sidx]; tpres[(uint64_t)sidx * n + r] = 1; }
    }
  }
}
// Long rows (round 6, second half).  The first version gave a long row to ONE workgroup of 256 threads, one dependent column -> presence byte -> operand chain
// per step: the hub rows of R-MAT-22 (1.6e5 entries) took 445 us in the backward level of the BC driver whose mask allows them — a CU forms about one
// scattered 16-byte gather per 3 clocks, whatever is in flight.  Now: parts of SPB_PART entries, a workgroup of 1024 threads per part with four entries per
// thread in flight, the part sums into a small array, a second kernel adds a row's parts in part order (fixed: reproducible).
constexpr uint32_t SPB_LT = 1024, SPB_LU = 4;
template <class T, class SR, int NSP>
__global__ __launch_bounds__(SPB_LT) void k_spb_pull_long(const uint32_t* __restrict__ rowptr, const uint32_t* __restrict__ col, const T* __restrict__ aval, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ rec,
                                                          const uint32_t* __restrict__ item, const T* __restrict__ ui, const uint8_t* __restrict__ upm, const uint8_t* __restrict__ allowm,
                                                          T* __restrict__ part_val, uint32_t* __restrict__ part_has, const SR sr) {
  constexpr uint32_t NW = SPB_LT / 64u;
  __shared__ T s_acc[NW][NSP]; __shared__ uint32_t s_has[NW];
  const uint32_t ni = cnt[1], lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  for (uint32_t k = blockIdx.x; k < ni; k += gridDim.x) {
    const uint32_t r = item[k], j = rec[3 * r], q0 = k - rec[3 * r + 1], am = allowm[j];
    const uint32_t rb = rowptr[j], re = rowptr[j + 1], b = rb + q0 * SPB_PART, e = (re - b > SPB_PART) ? b + SPB_PART : re;
    T acc[NSP]; uint32_t has = 0;
#pragma unroll
    for (int sidx = 0; sidx < NSP; sidx++) acc[sidx] = sr.identity;
    for (uint32_t p0 = b + threadIdx.x; p0 < e; p0 += SPB_LT * SPB_LU) {
      uint32_t c[SPB_LU], pm[SPB_LU]; T a[SPB_LU], u[SPB_LU][NSP];
#pragma unroll
      for (uint32_t q = 0; q < SPB_LU; q++) { const uint32_t p = p0 + q * SPB_LT; const bool ok = p < e; c[q] = ok ? col[p] : 0xFFFFFFFFu; a[q] = (ok && aval) ? aval[p] : T(); }
#pragma unroll
      for (uint32_t q = 0; q < SPB_LU; q++) pm[q] = c[q] != 0xFFFFFFFFu ? ((uint32_t)upm[c[q]] & am) : 0u;
#pragma unroll
      for (uint32_t q = 0; q < SPB_LU; q++) {
#pragma unroll
        for (int sidx = 0; sidx < NSP; sidx++) u[q][sidx] = pm[q] ? ui[(size_t)c[q] * NSP + sidx] : T();
      }
#pragma unroll
      for (uint32_t q = 0; q < SPB_LU; q++) {
#pragma unroll
        for (int sidx = 0; sidx < NSP; sidx++) if (pm[q] & (1u << sidx)) { const T pr = sr.mult(a[q], u[q][sidx]); acc[sidx] = (has & (1u << sidx)) ? sr.add(acc[sidx], pr) : pr; }
        has |= pm[q];
      }
    }
#pragma unroll
    for (int sidx = 0; sidx < NSP; sidx++) {
      T v = (has & (1u << sidx)) ? acc[sidx] : sr.identity;
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v = sr.add(v, shfl_xor_t<T>(v, m));
      acc[sidx] = v;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) has |= (uint32_t)__shfl_xor((int)has, m, 64);
    __syncthreads();
    if (lane == 0) { s_has[wv] = has; for (int sidx = 0; sidx < NSP; sidx++) s_acc[wv][sidx] = acc[sidx]; }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t hall = 0; for (uint32_t w = 0; w < NW; w++) hall |= s_has[w];
      for (int sidx = 0; sidx < NSP; sidx++) {
        T v = sr.identity; bool first = true;
        for (uint32_t w = 0; w < NW; w++) if (s_has[w] & (1u << sidx)) { v = first ? s_acc[w][sidx] : sr.add(v, s_acc[w][sidx]); first = false; }
        part_val[(size_t)k * NSP + sidx] = v;
      }
      part_has[k] = hall;
    }
  }
}
template <class T, class SR, int NSP>
__global__ void k_spb_long_combine(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ rec, const T* __restrict__ part_val, const uint32_t* __restrict__ part_has, uint64_t n,
                                   T* __restrict__ tval, uint8_t* __restrict__ tpres, const SR sr) {
  const uint32_t nrw = cnt[0];
  for (uint32_t r = blockIdx.x * 256 + threadIdx.x; r < nrw; r += gridDim.x * 256) {
    const uint32_t j = rec[3 * r], base = rec[3 * r + 1], np = rec[3 * r + 2];
#pragma unroll
    for (int sidx = 0; sidx < NSP; sidx++) {
      T v = sr.identity; bool any = false;
      for (uint32_t q = 0; q < np; q++) if (part_has[base + q] & (1u << sidx)) { const T pv = part_val[(size_t)(base + q) * NSP + sidx]; v = any ? sr.add(v, pv) : pv; any = true; }
      if (any) { tval[(uint64_t)sidx * n + j] = v; tpres[(uint64_t)sidx * n + j] = 1; }
    }
  }
}
// T (bitmap rows tval / tpres, nr x nout) = batch A (bitmap) (+).(x) op(B) under the mask's rows.  False when this product is not its case.
static bool spmm_pull_batch(GrB_Matrix A, DevBitmap& ab, GrB_Matrix Mmask, const DescView& dv, GrB_Semiring semiring, GrB_Matrix B, int zcode, uint64_t nout, DevBuf& tval, DevBuf& tpres) {
  const uint32_t nr = (uint32_t)A->nrows; const uint64_t nin = A->ncols;
  const size_t zs = type_size(zcode);
  if (nr > 8 || (zs != 4 && zs != 8) || A->type->code != zcode || zcode == T_BOOL || (nr > 4 ? 8u : 4u) * zs > 32) return false;      // (the products of a block in LDS: <= 64 KB)
  const char* e = getenv("GRB_MI355X_SPMM"); if (e && atoi(e) == 0) return false;
  const uint64_t tot = ab.nvals_known ? ab.nvals : count_present(ab.pres.as<uint8_t>(), (uint64_t)nr * nin);
  ab.nvals = tot; ab.nvals_known = true;
  // a thin batch (the first level: one entry per row; the last ones): the rows' own direction choice — a push over a handful of rows — beats a pass over the
  // matrix.  (Measured at R-MAT-22, whole BC driver: one pass for every level 10.2 ms, only for batches of >= 1/256 of the positions 12-13.5 ms, never 13.9 ms: the
  // per-row route pays two counting kernels and a host round trip per row, so the pass wins from a few thousand entries on.)
  if (!(e && atoi(e) == 1) && tot * 4096 < (uint64_t)nr * nin) return false;
  SemiringDesc sd = make_semiring_desc(semiring, /*swap_mult_args=*/true);       // vxm orientation: mult(u(i), B(i, j)); the kernels call mult(matrix value, operand value)
  DevCSR& R = const_cast<DevCSR&>(dv.tran1 ? (mat_to_device(B), B->csr) : mat_csc(B));            // rows of R = output positions
  if (R.nrows != nout || R.ncols != nin) return false;
  spmv_build_plan(R);                                                              // its row blocks (cached with the matrix, shared with kernel A)
  const bool uses_a = sd.flip ? binop_uses_y(sd.mulop) : binop_uses_x(sd.mulop);
  DevBuf acast; const void* av = uses_a ? cast_values(zcode, B->type->code, R.val.p, R.nnz, acast) : nullptr;
  const int NSP = nr <= 4 ? 4 : 8;
  DevBuf ui((size_t)nin * NSP * zs + 64), upm(nin + 64), allowm(nout + 64), mb, lcnt(16);
  const size_t max_long = (size_t)R.nnz / SPB_LONG + 16, max_items = (size_t)R.nnz / SPB_PART + max_long;      // (a long row holds > SPB_LONG entries; its parts: full ones + at most one more)
  DevBuf lrec(max_long * 12 + 64), litem(max_items * 4 + 64), lpval(max_items * NSP * zs + 64), lphas(max_items * 4 + 64);
  const uint8_t* mbool = nullptr; const uint8_t* mpres = nullptr;
  if (Mmask) { mpres = Mmask->bm.pres.as<uint8_t>(); mbool = (const uint8_t*)cast_values(T_BOOL, Mmask->type->code, Mmask->bm.val.p, (uint64_t)nr * nout, mb); }
  hipLaunchKernelGGL(k_spb_allow, dim3(grid_of(nout)), dim3(256), 0, stream(), mbool, mpres, nr, nout, dv.mask_struct ? 1 : 0, dv.mask_comp ? 1 : 0, allowm.as<uint8_t>());
  GRB_HIP(hipMemsetAsync(tpres.p, 0, (size_t)nr * nout, stream())); GRB_HIP(hipMemsetAsync(lcnt.p, 0, 16, stream()));
  hipLaunchKernelGGL(k_spb_long_rows, dim3(grid_of(R.nrows)), dim3(256), 0, stream(), R.rowptr.as<uint32_t>(), allowm.as<uint8_t>(), R.nrows, lcnt.as<uint32_t>(), lrec.as<uint32_t>(), litem.as<uint32_t>());
  bool ran = dispatch_type(zcode, [&]<class T>() {
    if constexpr (sizeof(T) == 4 || sizeof(T) == 8) {
      auto go = [&](auto NSPc) {
        constexpr int N = decltype(NSPc)::value;
        hipLaunchKernelGGL((k_spb_interleave<T, N>), dim3(grid_of(nin)), dim3(256), 0, stream(), (const T*)ab.val.p, ab.pres.as<uint8_t>(), nr, nin, (T*)ui.p, upm.as<uint8_t>());
        with_semiring<T>(sd, [&](auto sr) {
          typedef decltype(sr) SR;
          if (R.plan_nblocks) hipLaunchKernelGGL((k_spb_blocks<T, SR, N>), dim3(R.plan_nblocks), dim3(SPMV_THREADS), 0, stream(), (const SpmvBlock*)R.plan_blocks.p, R.rowptr.as<uint32_t>(), R.col.as<uint32_t>(),
                                                 (const T*)av, nout, (const T*)ui.p, upm.as<uint8_t>(), allowm.as<uint8_t>(), (T*)tval.p, tpres.as<uint8_t>(), sr);
          hipLaunchKernelGGL((k_spb_pull_long<T, SR, N>), dim3(2048), dim3(SPB_LT), 0, stream(), R.rowptr.as<uint32_t>(), R.col.as<uint32_t>(), (const T*)av, lcnt.as<uint32_t>(), lrec.as<uint32_t>(), litem.as<uint32_t>(),
                             (const T*)ui.p, upm.as<uint8_t>(), allowm.as<uint8_t>(), (T*)lpval.p, lphas.as<uint32_t>(), sr);
          hipLaunchKernelGGL((k_spb_long_combine<T, SR, N>), dim3(64), dim3(256), 0, stream(), lcnt.as<uint32_t>(), lrec.as<uint32_t>(), (const T*)lpval.p, lphas.as<uint32_t>(), nout, (T*)tval.p, tpres.as<uint8_t>(), sr);
        });
      };
      if (NSP == 4) go(std::integral_constant<int, 4>{}); else go(std::integral_constant<int, 8>{});
    }
  });
  GRB_HIP(hipGetLastError());
  return ran;
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
  const bool one_pass = spmm_pull_batch(A, ab, Mmask, dv, semiring, B, zcode, nout, tval, tpres);
  if (one_pass) { known = false; plans = "k_spb_blocks (all rows in one pull pass)"; }
  for (uint32_t s = 0; s < nr && !one_pass; s++) {
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
