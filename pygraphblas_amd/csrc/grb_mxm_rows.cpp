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

}  // namespace grb
