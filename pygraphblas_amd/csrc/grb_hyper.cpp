// grb_hyper.cpp — the hot path on hypersparse containers: dimensions beyond the 32-bit device layouts.
//
// The reference's containers default to GxB_INDEX_MAX = 2^60 rows and columns (pygraphblas/matrix.py:167-170,
// `B.resize()` in demo/Intro-Prez.ipynb: "pass no dimension to go hypersparse") and its BFS / shortest-path loops run
// `vxm` on them unchanged.  SuiteSparse keeps a list of the non-empty vectors for such a matrix; the MI355X layouts are a
// CSR over 32-bit row ids and a bitmap vector, which cannot hold 2^60 positions.  What can be held is the set of indices
// that occur: a product only ever touches the rows, columns and inner indices present in one of its operands (or in the
// output and the mask, for the write-back).  So:
//
//     1. per dimension role of the operation, the sorted union of the indices that occur (a "universe");
//     2. every operand relabelled into it — a monotone map, so sorted tuples stay sorted — as an ordinary small container;
//     3. the ordinary operation (HIP kernels, same semiring / mask / accumulator / descriptor) on those;
//     4. the output's indices mapped back.
//
// Steps 1, 2 and 4 are host-side bookkeeping over the entries of the operands (hypersparse containers live on the host
// mirror anyway); the arithmetic is the device's.  Covered: GrB_mxm, GrB_mxv, GrB_vxm and the eWiseAdd / eWiseMult of
// vectors and matrices (what the reference's loops wrap around the products: `w.iseq(v)`, `v @= M`).
#include "grb_opcommon.hpp"
#include <algorithm>

extern "C" {      // the entry points the compacted operations go through (defined in grb_mxv.cpp, grb_matrix_ops.cpp, grb_vector_ops.cpp)
GrB_Info GrB_mxm(GrB_Matrix, const GrB_Matrix, const GrB_BinaryOp, const GrB_Semiring, const GrB_Matrix, const GrB_Matrix, const GrB_Descriptor);
GrB_Info GrB_mxv(GrB_Vector, const GrB_Vector, const GrB_BinaryOp, const GrB_Semiring, const GrB_Matrix, const GrB_Vector, const GrB_Descriptor);
GrB_Info GrB_vxm(GrB_Vector, const GrB_Vector, const GrB_BinaryOp, const GrB_Semiring, const GrB_Vector, const GrB_Matrix, const GrB_Descriptor);
GrB_Info GrB_Vector_eWiseAdd_BinaryOp(GrB_Vector, const GrB_Vector, const GrB_BinaryOp, const GrB_BinaryOp, const GrB_Vector, const GrB_Vector, const GrB_Descriptor);
GrB_Info GrB_Vector_eWiseMult_BinaryOp(GrB_Vector, const GrB_Vector, const GrB_BinaryOp, const GrB_BinaryOp, const GrB_Vector, const GrB_Vector, const GrB_Descriptor);
GrB_Info GrB_Matrix_eWiseAdd_BinaryOp(GrB_Matrix, const GrB_Matrix, const GrB_BinaryOp, const GrB_BinaryOp, const GrB_Matrix, const GrB_Matrix, const GrB_Descriptor);
GrB_Info GrB_Matrix_eWiseMult_BinaryOp(GrB_Matrix, const GrB_Matrix, const GrB_BinaryOp, const GrB_BinaryOp, const GrB_Matrix, const GrB_Matrix, const GrB_Descriptor);
GrB_Info GrB_Matrix_new(GrB_Matrix*, GrB_Type, GrB_Index, GrB_Index);
GrB_Info GrB_Vector_new(GrB_Vector*, GrB_Type, GrB_Index);
GrB_Info GrB_Matrix_free(GrB_Matrix*);
GrB_Info GrB_Vector_free(GrB_Vector*);
}

namespace grb {

bool is_hyper(const GrB_Matrix_opaque* A) { return A && (A->nrows > GRB_DIM_DEVICE_MAX || A->ncols > GRB_DIM_DEVICE_MAX); }
bool is_hyper(const GrB_Vector_opaque* v) { return v && v->n > GRB_DIM_DEVICE_MAX; }

namespace {

struct Universe {
  std::vector<uint64_t> ids;                         // sorted, unique
  void add(const std::vector<GrB_Index>& v) { ids.insert(ids.end(), v.begin(), v.end()); }
  void seal() { std::sort(ids.begin(), ids.end()); ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
                if (ids.size() > GRB_DIM_DEVICE_MAX) fail(GrB_INSUFFICIENT_SPACE, "hypersparse operation: more distinct indices than the device layout holds"); }
  uint64_t pos(uint64_t x) const { return (uint64_t)(std::lower_bound(ids.begin(), ids.end(), x) - ids.begin()); }
  uint64_t dim() const { return ids.empty() ? 1 : ids.size(); }
};

void check_mat(GrB_Matrix A, const char* what) { if (!check_obj(A)) fail(GrB_UNINITIALIZED_OBJECT, std::string(what) + ": uninitialised matrix"); }
struct TempM { GrB_Matrix m = nullptr; ~TempM() { GrB_Matrix_free(&m); } };
struct TempV { GrB_Vector v = nullptr; ~TempV() { GrB_Vector_free(&v); } };

// A relabelled: entry (i, j) -> (R.pos(i), C.pos(j)).  Both maps are monotone, the tuples stay sorted by (row, column).
void compact(TempM& out, GrB_Matrix A, const Universe& R, const Universe& C) {
  if (!A) return;
  mat_to_host(A);
  GrB_Info info = GrB_Matrix_new(&out.m, A->type, R.dim(), C.dim()); if (info) fail(info, "hypersparse operation: temporary matrix");
  GrB_Matrix m = out.m; const size_t nv = A->hi.size();
  m->hi.resize(nv); m->hj.resize(nv); m->hx = A->hx;
  for (size_t k = 0; k < nv; k++) { m->hi[k] = R.pos(A->hi[k]); m->hj[k] = C.pos(A->hj[k]); }
  m->host_valid = true;
}
void compact(TempV& out, GrB_Vector v, const Universe& U) {
  if (!v) return;
  vec_to_host(v);
  GrB_Info info = GrB_Vector_new(&out.v, v->type, U.dim()); if (info) fail(info, "hypersparse operation: temporary vector");
  GrB_Vector w = out.v; const size_t nv = v->hi.size();
  w->hi.resize(nv); w->hx = v->hx;
  for (size_t k = 0; k < nv; k++) w->hi[k] = U.pos(v->hi[k]);
  w->host_valid = true;
}
// the result back in the caller's index space
void expand(GrB_Matrix C, GrB_Matrix c, const Universe& R, const Universe& Cu) {
  mat_to_host(c);
  const size_t nv = c->hi.size();
  C->hi.resize(nv); C->hj.resize(nv); C->hx = c->hx; C->pending.clear();
  for (size_t k = 0; k < nv; k++) { C->hi[k] = R.ids[c->hi[k]]; C->hj[k] = Cu.ids[c->hj[k]]; }
  C->host_valid = true; C->iso_full = false; mat_invalidate_device(C);
}
void expand(GrB_Vector w, GrB_Vector c, const Universe& U) {
  vec_to_host(c);
  const size_t nv = c->hi.size();
  w->hi.resize(nv); w->hx = c->hx; w->pending.clear();
  for (size_t k = 0; k < nv; k++) w->hi[k] = U.ids[c->hi[k]];
  w->host_valid = true; w->iso_full = false; vec_invalidate_device(w);
}
void rows_cols(GrB_Matrix A, bool transposed, Universe& rows, Universe& cols) {     // A's (or A^T's) row and column indices into the two universes
  if (!A) return;
  mat_to_host(A);
  (transposed ? cols : rows).add(A->hi); (transposed ? rows : cols).add(A->hj);
}
void indices(GrB_Vector v, Universe& U) { if (v) { vec_to_host(v); U.add(v->hi); } }
void relay(GrB_Info info, const std::string& err, const char* what) {
  if (info != GrB_SUCCESS) fail(info, err.empty() ? std::string(what) + " on the compacted operands failed" : err);
}

}  // namespace

// w<mask> = accum(w, A u) or accum(w, u A) with any of them hypersparse.  `tran` is the descriptor bit that transposes A
// (INP0 for mxv, INP1 for vxm); the output runs over the rows of the matrix as used (A for mxv, A^T for vxm), the operand over its columns.
void hyper_mxv_like(GrB_Vector w, GrB_Vector mask, GrB_BinaryOp accum, GrB_Semiring semiring, GrB_Matrix A, GrB_Vector u, GrB_Descriptor desc, bool is_vxm) {
  if (!check_obj(A) || !check_obj(u) || (mask && !check_obj(mask))) fail(GrB_UNINITIALIZED_OBJECT, "mxv/vxm: uninitialised operand");
  const DescView dv(desc);
  const bool tranA = is_vxm ? dv.tran1 : dv.tran0;
  // mxv: w(i) over rows of op(A), u over its columns; vxm: w(j) over columns of op(A), u over its rows
  const uint64_t ar = tranA ? A->ncols : A->nrows, ac = tranA ? A->nrows : A->ncols;
  const uint64_t out_n = is_vxm ? ac : ar, in_n = is_vxm ? ar : ac;
  if (u->n != in_n || w->n != out_n || (mask && mask->n != out_n)) fail(GrB_DIMENSION_MISMATCH, "mxv/vxm: dimensions do not conform");
  Universe R, Cc;                       // rows and columns of op(A)
  rows_cols(A, tranA, R, Cc);
  Universe& Out = is_vxm ? Cc : R; Universe& In = is_vxm ? R : Cc;
  indices(w, Out); indices(mask, Out); indices(u, In);
  R.seal(); Cc.seal();
  TempM a; TempV w2, m2, u2;
  if (tranA) compact(a, A, Cc, R); else compact(a, A, R, Cc);          // A itself is stored untransposed: its rows are op(A)'s columns when transposed
  compact(w2, w, Out); compact(m2, mask, Out); compact(u2, u, In);
  const GrB_Info info = is_vxm ? GrB_vxm(w2.v, m2.v, accum, semiring, u2.v, a.m, desc) : GrB_mxv(w2.v, m2.v, accum, semiring, a.m, u2.v, desc);
  relay(info, w2.v->err, "mxv/vxm");
  expand(w, w2.v, Out);
  g_last_plan = "hypersparse<" + std::to_string(R.ids.size()) + "x" + std::to_string(Cc.ids.size()) + "> " + g_last_plan;
}

void hyper_mxm(GrB_Matrix C, GrB_Matrix M, GrB_BinaryOp accum, GrB_Semiring semiring, GrB_Matrix A, GrB_Matrix B, GrB_Descriptor desc) {
  check_mat(A, "mxm"); check_mat(B, "mxm"); if (M) check_mat(M, "mxm");
  const DescView dv(desc);
  const uint64_t ar = dv.tran0 ? A->ncols : A->nrows, ac = dv.tran0 ? A->nrows : A->ncols;
  const uint64_t br = dv.tran1 ? B->ncols : B->nrows, bc = dv.tran1 ? B->nrows : B->ncols;
  if (ac != br || C->nrows != ar || C->ncols != bc || (M && (M->nrows != ar || M->ncols != bc))) fail(GrB_DIMENSION_MISMATCH, "mxm: dimensions do not conform");
  Universe R, K, Cc;                    // rows of C, the inner dimension, columns of C
  rows_cols(A, dv.tran0, R, K); rows_cols(B, dv.tran1, K, Cc); rows_cols(C, false, R, Cc); rows_cols(M, false, R, Cc);
  R.seal(); K.seal(); Cc.seal();
  TempM a, b, c, m;
  if (dv.tran0) compact(a, A, K, R); else compact(a, A, R, K);
  if (dv.tran1) compact(b, B, Cc, K); else compact(b, B, K, Cc);
  compact(c, C, R, Cc); compact(m, M, R, Cc);
  relay(GrB_mxm(c.m, m.m, accum, semiring, a.m, b.m, desc), c.m->err, "mxm");
  expand(C, c.m, R, Cc);
  g_last_plan = "hypersparse<" + std::to_string(R.ids.size()) + "x" + std::to_string(K.ids.size()) + "x" + std::to_string(Cc.ids.size()) + "> " + g_last_plan;
}

void hyper_vec_ewise(GrB_Vector w, GrB_Vector mask, GrB_BinaryOp accum, GrB_BinaryOp op, GrB_Vector u, GrB_Vector v, GrB_Descriptor desc, bool is_union) {
  if (!check_obj(u) || !check_obj(v) || (mask && !check_obj(mask))) fail(GrB_UNINITIALIZED_OBJECT, "eWise: uninitialised operand");
  const uint64_t n = w->n;
  if (u->n != n || v->n != n || (mask && mask->n != n)) fail(GrB_DIMENSION_MISMATCH, "eWise: vector sizes differ");
  Universe U; indices(w, U); indices(mask, U); indices(u, U); indices(v, U); U.seal();
  TempV w2, m2, u2, v2;
  compact(w2, w, U); compact(m2, mask, U); compact(u2, u, U); compact(v2, v, U);
  relay(is_union ? GrB_Vector_eWiseAdd_BinaryOp(w2.v, m2.v, accum, op, u2.v, v2.v, desc) : GrB_Vector_eWiseMult_BinaryOp(w2.v, m2.v, accum, op, u2.v, v2.v, desc), w2.v->err, "eWise");
  expand(w, w2.v, U);
}

void hyper_mat_ewise(GrB_Matrix C, GrB_Matrix M, GrB_BinaryOp accum, GrB_BinaryOp op, GrB_Matrix A, GrB_Matrix B, GrB_Descriptor desc, bool is_union) {
  check_mat(A, "eWise"); check_mat(B, "eWise"); if (M) check_mat(M, "eWise");
  const DescView dv(desc);
  const uint64_t ar = dv.tran0 ? A->ncols : A->nrows, ac = dv.tran0 ? A->nrows : A->ncols;
  const uint64_t br = dv.tran1 ? B->ncols : B->nrows, bc = dv.tran1 ? B->nrows : B->ncols;
  if (ar != br || ac != bc || C->nrows != ar || C->ncols != ac || (M && (M->nrows != ar || M->ncols != ac))) fail(GrB_DIMENSION_MISMATCH, "eWise: dimensions do not conform");
  Universe R, Cc;
  rows_cols(A, dv.tran0, R, Cc); rows_cols(B, dv.tran1, R, Cc); rows_cols(C, false, R, Cc); rows_cols(M, false, R, Cc);
  R.seal(); Cc.seal();
  TempM a, b, c, m;
  if (dv.tran0) compact(a, A, Cc, R); else compact(a, A, R, Cc);
  if (dv.tran1) compact(b, B, Cc, R); else compact(b, B, R, Cc);
  compact(c, C, R, Cc); compact(m, M, R, Cc);
  relay(is_union ? GrB_Matrix_eWiseAdd_BinaryOp(c.m, m.m, accum, op, a.m, b.m, desc) : GrB_Matrix_eWiseMult_BinaryOp(c.m, m.m, accum, op, a.m, b.m, desc), c.m->err, "eWise");
  expand(C, c.m, R, Cc);
}

}  // namespace grb
