// grb_matrix_ops.cpp — GrB_mxm and the matrix operations around it.
//
//   GrB_mxm                 <- lib.GrB_mxm, pygraphblas/matrix.py:2572-2583 (Matrix.mxm)       THE HOT PATH
//   GrB_transpose           <- Matrix.transpose                 (pygraphblas/matrix.py:1003-1062)
//   GrB_Matrix_eWiseAdd/Mult_* <- Matrix.eadd / emult           (pygraphblas/matrix.py:1103-1413)
//   GrB_Matrix_apply, GxB_Matrix_apply_BinaryOp1st/2nd          (pygraphblas/matrix.py:1934-2040)
//   GxB_Matrix_select       <- Matrix.select / tril / triu      (pygraphblas/matrix.py:2042-2200)
//   GrB_Matrix_reduce_Monoid <- Matrix.reduce_vector            (pygraphblas/matrix.py:1861-1932)
//   GrB_Matrix_assign_<T>   <- Matrix.assign_scalar             (pygraphblas/matrix.py:3180-3230)
// Semantics: SURVEY.md Appendix A.  Every operation computes T into a fresh CSR and then performs
// the C<M,replace> = accum(C,T) write-back, so the output may alias any input.
#include "grb_opcommon.hpp"
#include "grb_matops.hpp"

using namespace grb;

namespace grb {
bool mxm_few_rows_wanted(const DevCSR& Ad, const DevCSR& Bd);
void mxm_few_rows(const DevCSR& Ad, GrB_Type atype, GrB_Matrix Mmask, const DescView& dv, GrB_Semiring semiring, GrB_Matrix B, int zcode, DevCSR& T);
bool few_long_rows(uint64_t nrows, uint64_t ncols, uint64_t nnz);
void ewise_few_rows(GrB_Matrix C, GrB_Matrix Mmask, const DescView& dv, GrB_BinaryOp accum, GrB_BinaryOp op, const DevCSR& Ad, GrB_Type atype, const DevCSR& Bd, GrB_Type btype, bool is_union, DevCSR& T);
// round 6: the same batches as bitmaps, all rows in one kernel (grb_mxm_rows.cpp)
bool batch_wanted(GrB_Matrix C, uint64_t work);
void ewise_batch(GrB_Matrix C, GrB_Matrix Mmask, const DescView& dv, GrB_BinaryOp accum, GrB_BinaryOp op, GrB_Matrix A, GrB_Matrix B, bool is_union);
void apply_batch(GrB_Matrix C, int mode, int opcode, int xcode, const uint8_t* scalar16, GrB_Matrix A);
bool mxm_batch(GrB_Matrix C, GrB_Matrix A, GrB_Matrix Mmask, const DescView& dv, GrB_BinaryOp accum, GrB_Semiring semiring, GrB_Matrix B, int zcode, DevCSR& T);
}

namespace {

struct CsrView { const DevCSR* m; DevBuf vals; const void* v; };

// device CSR of op(A) with values cast to `code` (or untouched when `need_vals` is false)
const DevCSR& operand(GrB_Matrix A, bool transpose) { mat_to_device(A); return transpose ? mat_csc(A) : A->csr; }

void adopt(GrB_Matrix C, DevCSR& T, int tcode) {
  // C becomes exactly T (cast values if the types differ)
  if (tcode != C->type->code && T.nnz) { DevBuf c(T.nnz * C->type->size); vec_cast_values(C->type->code, c.p, tcode, T.val.p, T.nnz); T.val = std::move(c); }
  mat_invalidate_host(C); C->csc.clear(); C->csr.clear(); C->bm.clear();
  C->csr.nrows = T.nrows; C->csr.ncols = T.ncols; C->csr.nnz = T.nnz;
  C->csr.rowptr = std::move(T.rowptr); C->csr.col = std::move(T.col); C->csr.val = std::move(T.val);
  if (!C->csr.val.p) C->csr.val.alloc(8);
  if (!C->csr.col.p) C->csr.col.alloc(8);
  C->csr.valid = true; C->dev_valid = true; C->host_valid = false;
}

// C<M,replace> = accum(C, T).  `t_masked`: T already has no entry the mask forbids.
void matrix_write_back(GrB_Matrix C, DevCSR& T, int tcode, GrB_Matrix M, const DescView& dv, GrB_BinaryOp accum, bool t_masked) {
  if (accum) check_binop(accum, "accum");
  if (!M && dv.mask_comp) {      // no mask + complement: nothing may be written
    if (dv.replace) GrB_Matrix_clear(C);
    return;
  }
  const bool c_empty = mat_nvals(C) == 0;
  if (!accum && (!M || (t_masked && (dv.replace || c_empty)))) { adopt(C, T, tcode); return; }
  if (!accum && M && (dv.replace || c_empty)) {
    // filter T by the mask, then adopt
    mat_to_device(M);
    DevBuf keep(T.nnz + 1); DevCSR F;
    mask_flags(T, M->csr, M->type->code, dv.mask_struct, dv.mask_comp, keep.as<uint8_t>());
    csr_compact(T, T.val.p, type_size(tcode), keep.as<uint8_t>(), F);
    adopt(C, F, tcode); return;
  }
  // general three-way merge in the accumulator's domain (or C's type)
  mat_to_device(C); if (M) mat_to_device(M);
  const int ccode = C->type->code, ecode = accum ? accum->xtype->code : ccode;
  DevBuf tc, cc; DevCSR Cv, Tv;     // views with cast values
  const void* tv = cast_values(ecode, tcode, T.val.p, T.nnz, tc);
  const void* cv = cast_values(ecode, ccode, C->csr.val.p, C->csr.nnz, cc);
  // build shallow views that share index arrays: done by temporarily wrapping pointers
  struct Shallow { DevCSR v; ~Shallow() { v.rowptr.p = nullptr; v.col.p = nullptr; v.val.p = nullptr; } } sc, st;
  sc.v.nrows = C->csr.nrows; sc.v.ncols = C->csr.ncols; sc.v.nnz = C->csr.nnz; sc.v.rowptr.p = C->csr.rowptr.p; sc.v.col.p = C->csr.col.p; sc.v.val.p = (void*)cv;
  st.v.nrows = T.nrows; st.v.ncols = T.ncols; st.v.nnz = T.nnz; st.v.rowptr.p = T.rowptr.p; st.v.col.p = T.col.p; st.v.val.p = (void*)tv;
  DevCSR out;
  csr_writeback(ecode, C->csr.nrows, sc.v, st.v, M ? &M->csr : nullptr, M ? M->type->code : 0, dv.mask_struct, dv.mask_comp, dv.replace,
                accum ? accum->opcode : -1, out);
  adopt(C, out, ecode);
}

void check_mat(GrB_Matrix A, const char* what) { if (!check_obj(A)) fail(GrB_UNINITIALIZED_OBJECT, std::string(what) + ": uninitialised matrix"); }

// ---------------------------------------------------------------------------------------------------------------------
void do_mxm(GrB_Matrix C, GrB_Matrix M, GrB_BinaryOp accum, GrB_Semiring semiring, GrB_Matrix A, GrB_Matrix B, GrB_Descriptor desc) {
  need_device();
  if (is_hyper(C) || is_hyper(M) || is_hyper(A) || is_hyper(B)) { hyper_mxm(C, M, accum, semiring, A, B, desc); return; }   // dimensions beyond the device layouts
  check_mat(A, "mxm"); check_mat(B, "mxm"); if (M) check_mat(M, "mxm");
  const DescView dv(desc);
  const uint64_t ar = dv.tran0 ? A->ncols : A->nrows, ac = dv.tran0 ? A->nrows : A->ncols;
  const uint64_t br = dv.tran1 ? B->ncols : B->nrows, bc = dv.tran1 ? B->nrows : B->ncols;
  if (ac != br || C->nrows != ar || C->ncols != bc || (M && (M->nrows != ar || M->ncols != bc))) fail(GrB_DIMENSION_MISMATCH, "mxm: dimensions do not conform");
  SemiringDesc sd = make_semiring_desc(semiring, false);
  g_last_plan.clear();
  if (!M && dv.mask_comp) { if (dv.replace) GrB_Matrix_clear(C); return; }
  // a batch of a few very long rows times a large matrix (the BC sweeps' frontier products): the rows of the batch's BITMAP through GrB_vxm, the result a bitmap
  if (!dv.tran0 && mat_batch_shape(A->nrows, A->ncols, A->type->code) && mat_batch_shape(C->nrows, C->ncols, C->type->code) && (!M || mat_batch_shape(M->nrows, M->ncols, M->type->code)) &&
      !is_hyper(B) && A != B && M != B) {
    const DevCSR& Bd0 = operand(B, false);
    const char* e = getenv("GRB_MI355X_BATCH");
    if (e ? atoi(e) != 0 : (Bd0.nnz >= (1u << 20) && Bd0.ncols >= 65536u)) {
      if (accum) check_binop(accum, "accum");
      DevCSR T;
      if (mxm_batch(C, A, M, dv, accum, semiring, B, sd.zcode, T)) return;
      matrix_write_back(C, T, sd.zcode, M, dv, accum, true); return;
    }
  }
  const DevCSR& Ad = operand(A, dv.tran0); const DevCSR& Bd = operand(B, dv.tran1);
  const bool uses_a = binop_uses_x(sd.mulop), uses_b = binop_uses_y(sd.mulop);
  DevBuf acast, bcast;
  SpgemmCall call{};
  call.A = &Ad; call.B = &Bd;
  call.aval = uses_a ? cast_values(sd.zcode, A->type->code, Ad.val.p, Ad.nnz, acast) : nullptr;
  call.bval = uses_b ? cast_values(sd.zcode, B->type->code, Bd.val.p, Bd.nnz, bcast) : nullptr;
  DevCSR T; bool t_masked = false;
  {
    const bool fp = sd.zcode == T_FP32 || sd.zcode == T_FP64;
    call.ordered = fp && sd.addop != B_MIN && sd.addop != B_MAX && (deterministic_env() || dv.axb == GxB_AxB_GUSTAVSON);      // (MIN / MAX: the same bits in any order)
  }
  if (mxm_few_rows_wanted(Ad, Bd)) {            // a handful of output rows (batched BC frontiers): one vxm per row, see grb_mxm_rows.cpp
    mxm_few_rows(Ad, A->type, M, dv, semiring, B, sd.zcode, T); t_masked = true;
  } else if (M && !dv.mask_comp) {
    mat_to_device(M);
    call.M = &M->csr; call.mcode = M->type->code; call.mstruct = dv.mask_struct;
    spgemm_masked(call, sd, T); t_masked = true;
  } else {
    // no mask, or a complemented one (applied by the write-back): the two-pass LDS-hash Gustavson; expand / sort / compress on
    // request (it forms floating-point sums in a fixed order): GRB_MI355X_SPGEMM=esc or the descriptor's AxB method GxB_AxB_DOT
    // Deterministic mode (round 5): GRB_MI355X_DETERMINISTIC=1 or the descriptor's GxB_AxB_GUSTAVSON — the two-pass product with its dense path's ordered
    // walk (floating-point sums bit-reproducible from run to run at ~1.15 x the time, grb_spgemm_hash.hpp); results too wide for that path (> 2^20 columns)
    // take expand / sort / compress.  Integer / Boolean monoids are exact in any order: nothing changes for them.
    const char* e = getenv("GRB_MI355X_SPGEMM");
    if ((e && !strcmp(e, "esc")) || dv.axb == GxB_AxB_DOT || (call.ordered && Bd.ncols > (1u << 20))) spgemm_esc(call, sd, T); else spgemm_hash(call, sd, T);
  }
  matrix_write_back(C, T, sd.zcode, M, dv, accum, t_masked);
}

void do_transpose(GrB_Matrix C, GrB_Matrix M, GrB_BinaryOp accum, GrB_Matrix A, GrB_Descriptor desc) {
  need_device(); check_mat(A, "transpose"); if (M) check_mat(M, "transpose");
  const DescView dv(desc);
  // desc.INP0 = TRAN transposes the input first: the result is then A itself
  const bool tr = !dv.tran0;
  const uint64_t r = tr ? A->ncols : A->nrows, c = tr ? A->nrows : A->ncols;
  if (C->nrows != r || C->ncols != c || (M && (M->nrows != r || M->ncols != c))) fail(GrB_DIMENSION_MISMATCH, "transpose: dimensions do not conform");
  const DevCSR& S = operand(A, tr);
  DevCSR T; const size_t ts = A->type->size;
  T.nrows = S.nrows; T.ncols = S.ncols; T.nnz = S.nnz;
  T.rowptr.alloc(((size_t)S.nrows + 1) * 4); T.col.alloc(S.nnz * 4 + 4); T.val.alloc(S.nnz * ts + 8);
  GRB_HIP(hipMemcpyAsync(T.rowptr.p, S.rowptr.p, ((size_t)S.nrows + 1) * 4, hipMemcpyDeviceToDevice, stream()));
  if (S.nnz) { GRB_HIP(hipMemcpyAsync(T.col.p, S.col.p, S.nnz * 4, hipMemcpyDeviceToDevice, stream()));
               GRB_HIP(hipMemcpyAsync(T.val.p, S.val.p, S.nnz * ts, hipMemcpyDeviceToDevice, stream())); }
  T.valid = true;
  matrix_write_back(C, T, A->type->code, M, dv, accum, false);
}

void do_ewise(GrB_Matrix C, GrB_Matrix M, GrB_BinaryOp accum, GrB_BinaryOp op, GrB_Matrix A, GrB_Matrix B, GrB_Descriptor desc, bool is_union) {
  need_device(); check_mat(A, "eWise"); check_mat(B, "eWise"); if (M) check_mat(M, "eWise");
  check_binop(op, "eWise");
  if (is_hyper(C) || is_hyper(M) || is_hyper(A) || is_hyper(B)) { hyper_mat_ewise(C, M, accum, op, A, B, desc, is_union); return; }
  const DescView dv(desc);
  const uint64_t ar = dv.tran0 ? A->ncols : A->nrows, ac = dv.tran0 ? A->nrows : A->ncols;
  const uint64_t br = dv.tran1 ? B->ncols : B->nrows, bc = dv.tran1 ? B->nrows : B->ncols;
  if (ar != br || ac != bc || C->nrows != ar || C->ncols != ac || (M && (M->nrows != ar || M->ncols != ac))) fail(GrB_DIMENSION_MISMATCH, "eWise: dimensions do not conform");
  if (!M && dv.mask_comp) { if (dv.replace) GrB_Matrix_clear(C); return; }
  if (!dv.tran0 && !dv.tran1 && batch_wanted(C, mat_nvals(A) + mat_nvals(B)) && A->type->code < T_FC32 && B->type->code < T_FC32 && (!M || M->type->code < T_FC32)) {
    if (accum) check_binop(accum, "accum");
    ewise_batch(C, M, dv, accum, op, A, B, is_union); return;      // a batch of a few very long rows (BC sweeps): its bitmap as ONE vector through the vector kernel
  }
  const DevCSR& Ad = operand(A, dv.tran0); const DevCSR& Bd = operand(B, dv.tran1);
  if (few_long_rows(C->nrows, C->ncols, Ad.nnz + Bd.nnz)) {          // a batch of a few very long rows (BC sweeps): row by row through the vector kernels
    DevCSR T; ewise_few_rows(C, M, dv, accum, op, Ad, A->type, Bd, B->type, is_union, T);
    adopt(C, T, C->type->code); return;
  }
  const int xc = op->xtype->code;
  DevBuf acast, bcast;
  const void* av = cast_values(xc, A->type->code, Ad.val.p, Ad.nnz, acast);
  const void* bv = cast_values(xc, B->type->code, Bd.val.p, Bd.nnz, bcast);
  DevCSR T;
  csr_ewise(xc, Ad, av, Bd, bv, op->opcode, is_union, T);
  matrix_write_back(C, T, xc, M, dv, accum, false);
}

// mode 0 unary, 1 bind-first, 2 bind-second
void do_apply(GrB_Matrix C, GrB_Matrix M, GrB_BinaryOp accum, int mode, int opcode, int xcode, const void* scalar, int scode, GrB_Matrix A, GrB_Descriptor desc) {
  need_device(); check_mat(A, "apply"); if (M) check_mat(M, "apply");
  const DescView dv(desc);
  const uint64_t r = dv.tran0 ? A->ncols : A->nrows, c = dv.tran0 ? A->nrows : A->ncols;
  if (C->nrows != r || C->ncols != c || (M && (M->nrows != r || M->ncols != c))) fail(GrB_DIMENSION_MISMATCH, "apply: dimensions do not conform");
  if (mode == 0 && opcode >= U_POSITIONI && opcode <= U_POSITIONJ1) {      // positional: T has op(A)'s pattern, the values are the entries' row / column indices in the operator's type
    if (!M && dv.mask_comp) { if (dv.replace) GrB_Matrix_clear(C); return; }
    const DevCSR& S = operand(A, dv.tran0);
    DevCSR T; T.nrows = S.nrows; T.ncols = S.ncols; T.nnz = S.nnz;
    T.rowptr.alloc(((size_t)S.nrows + 1) * 4); T.col.alloc(S.nnz * 4 + 4); T.val.alloc(S.nnz * type_size(xcode) + 8);
    GRB_HIP(hipMemcpyAsync(T.rowptr.p, S.rowptr.p, ((size_t)S.nrows + 1) * 4, hipMemcpyDeviceToDevice, stream()));
    if (S.nnz) GRB_HIP(hipMemcpyAsync(T.col.p, S.col.p, S.nnz * 4, hipMemcpyDeviceToDevice, stream()));
    csr_position_values(xcode, S, opcode - U_POSITIONI, T.val.p);
    T.valid = true;
    matrix_write_back(C, T, xcode, M, dv, accum, false);
    return;
  }
  if (!M && !accum && !dv.mask_comp && !dv.tran0 && A->bm.valid && !A->host_valid && batch_wanted(C, mat_nvals(A)) && A->type->code < T_FC32) {      // a batch that lives as a bitmap stays one
    uint8_t s16[16] = {0}; if (scalar) cast_scalar(xcode, s16, scode, scalar);
    apply_batch(C, mode, opcode, xcode, s16, A); return;
  }
  const DevCSR& S = operand(A, dv.tran0);
  DevCSR T; T.nrows = S.nrows; T.ncols = S.ncols; T.nnz = S.nnz;
  T.rowptr.alloc(((size_t)S.nrows + 1) * 4); T.col.alloc(S.nnz * 4 + 4); T.val.alloc(S.nnz * type_size(xcode) + 8);
  GRB_HIP(hipMemcpyAsync(T.rowptr.p, S.rowptr.p, ((size_t)S.nrows + 1) * 4, hipMemcpyDeviceToDevice, stream()));
  if (S.nnz) GRB_HIP(hipMemcpyAsync(T.col.p, S.col.p, S.nnz * 4, hipMemcpyDeviceToDevice, stream()));
  DevBuf ac; const void* av = cast_values(xcode, A->type->code, S.val.p, S.nnz, ac);
  uint8_t s[16] = {0}; if (scalar) cast_scalar(xcode, s, scode, scalar);
  vec_apply(xcode, S.nnz, av, nullptr, mode, opcode, s, T.val.p, nullptr);
  T.valid = true;
  matrix_write_back(C, T, xcode, M, dv, accum, false);
}

void do_select(GrB_Matrix C, GrB_Matrix M, GrB_BinaryOp accum, GxB_SelectOp op, GrB_Matrix A, GxB_Scalar thunk, GrB_Descriptor desc) {
  need_device(); check_mat(A, "select"); if (M) check_mat(M, "select");
  if (!check_obj(op)) fail(GrB_UNINITIALIZED_OBJECT, "select: operator");
  if (op->opcode == SEL_USER) not_implemented("user-defined select operator");
  const DescView dv(desc);
  const uint64_t r = dv.tran0 ? A->ncols : A->nrows, c = dv.tran0 ? A->nrows : A->ncols;
  if (C->nrows != r || C->ncols != c || (M && (M->nrows != r || M->ncols != c))) fail(GrB_DIMENSION_MISMATCH, "select: dimensions do not conform");
  const DevCSR& S = operand(A, dv.tran0);
  const int acode = A->type->code;
  int64_t k = 0; uint8_t th[16] = {0};
  if (thunk && check_obj(thunk) && thunk->has) { cast_scalar(T_INT64, &k, thunk->type->code, thunk->x); cast_scalar(acode, th, thunk->type->code, thunk->x); }
  DevBuf keep(S.nnz + 1); DevCSR T;
  if (op->opcode <= SEL_OFFDIAG) select_positional_flags(S, op->opcode, k, keep.as<uint8_t>());
  else select_value_flags(acode, S.nnz, S.val.p, nullptr, op->opcode, th, keep.as<uint8_t>());
  csr_compact(S, S.val.p, A->type->size, keep.as<uint8_t>(), T);
  matrix_write_back(C, T, acode, M, dv, accum, false);
}

void do_reduce_vector(GrB_Vector w, GrB_Vector mask, GrB_BinaryOp accum, GrB_Monoid monoid, GrB_Matrix A, GrB_Descriptor desc) {
  need_device(); check_mat(A, "reduce");
  if (!check_obj(monoid)) fail(GrB_UNINITIALIZED_OBJECT, "reduce: monoid"); check_binop(monoid->op, "monoid");
  if (mask && !check_obj(mask)) fail(GrB_UNINITIALIZED_OBJECT, "reduce: mask");
  const DescView dv(desc);
  const uint64_t r = dv.tran0 ? A->ncols : A->nrows;
  if (w->n != r || (mask && mask->n != r)) fail(GrB_DIMENSION_MISMATCH, "reduce: dimensions do not conform");
  DevBuf allow_buf; bool nothing = false;
  const uint8_t* allow = vector_allow(mask, dv, r, allow_buf, &nothing);
  if (nothing) { if (dv.replace) GrB_Vector_clear(w); return; }
  const int mc = monoid->op->ztype->code;
  DevBuf ac, tval(r * type_size(mc) + 8), tpres(r + 1);
  // the columns of a large matrix whose transpose is not at hand (desc T0 on a by-row matrix): no transpose is built for this, every
  // entry combines into its column's accumulator.  (Small matrices keep the row-wise reduction of the transpose: its fixed order
  // is what the reference's docstring values were computed with.)
  bool done = false;
  if (dv.tran0) {
    mat_to_device(A);
    // (only where the order the entries land in cannot show: integer / Boolean monoids and MIN / MAX.  A floating-point PLUS or TIMES keeps
    //  ONE fixed-order algorithm — the row reduction of the transpose — whatever the cache holds: the same call must not return different
    //  bits depending on whether an earlier operation happened to build the transpose)
    const int rop = monoid->op->opcode;
    const bool order_free = !(mc == T_FP32 || mc == T_FP64) || rop == B_MIN || rop == B_MAX || rop == B_ANY;
    // (taken WHETHER OR NOT a cached transpose exists: its row reduction adds in another order, and the same call must not return other bits because an
    //  earlier product happened to build the transpose)
    if (A->csr.nnz >= (1u << 20) && !order_free && A->csr.nrows <= 64) {      // a few long rows (a batch of the BC sweeps): row after row, no atomics, fixed order
      const void* av = cast_values(mc, A->type->code, A->csr.val.p, A->csr.nnz, ac);
      done = csr_reduce_cols_few_rows(mc, A->csr, av, rop, tval.p, tpres.as<uint8_t>());
    }
    if (!done && !A->csc.valid && A->csr.nnz >= (1u << 20) && order_free) {
      const void* av = cast_values(mc, A->type->code, A->csr.val.p, A->csr.nnz, ac);
      uint8_t id[16]; memcpy(id, monoid->identity, 16);
      fp_minmax_identity(mc, monoid->op->opcode, id);          // FP MIN / MAX start from NaN = from the column's first value (the one NaN rule, grb_opcommon.hpp)
      done = csr_reduce_cols(mc, A->csr, av, monoid->op->opcode, id, tval.p, tpres.as<uint8_t>());
    }
  }
  if (!done) {
    const DevCSR& S = operand(A, dv.tran0);
    const void* av = cast_values(mc, A->type->code, S.val.p, S.nnz, ac);
    csr_reduce_rows(mc, S, av, monoid->op->opcode, tval.p, tpres.as<uint8_t>());
  }
  vector_write_back(w, mc, tval, tpres, allow, accum, dv.replace, false);
}

// C<M>(I,J) = accum(C(I,J), x): built as a T with the scalar at every (i,j) of I x J, then assign semantics
void do_assign_scalar(GrB_Matrix C, GrB_Matrix M, GrB_BinaryOp accum, const void* x, int xcode, const GrB_Index* I, GrB_Index ni, const GrB_Index* J, GrB_Index nj, GrB_Descriptor desc) {
  // no HBM layout for this container (hypersparse dimensions, or complex entries): bookkeeping on the host mirror
  if (C->nrows > GRB_DIM_DEVICE_MAX || C->ncols > GRB_DIM_DEVICE_MAX || C->type->code >= T_FC32) { host_assign_scalar(C, M, accum, x, xcode, I, ni, J, nj, desc); return; }
  need_device(); if (M) check_mat(M, "assign");
  const DescView dv(desc);
  if (M && (M->nrows != C->nrows || M->ncols != C->ncols)) fail(GrB_DIMENSION_MISMATCH, "assign: mask dimensions");
  if (I == GrB_ALL && J == GrB_ALL && !M && !accum && !dv.mask_comp && C->nrows * C->ncols <= 0xFFFFFFF0ull && C->nrows * C->ncols > 0) {
    // every position of C: the full one-valued matrix, written by one kernel (`Matrix.dense`, `M[:, :] = x`)
    uint8_t s0[16]; cast_scalar(C->type->code, s0, xcode, x);
    DevCSR T; csr_dense_fill((uint32_t)C->nrows, (uint32_t)C->ncols, s0, C->type->size, T);
    adopt(C, T, C->type->code);
    return;
  }
  // (indices are validated as 64-bit values before they are narrowed to the device layout's 32 bits)
  const std::vector<uint64_t> rows64 = expand_index_list(I, ni, C->nrows, "assign (rows)"), cols64 = expand_index_list(J, nj, C->ncols, "assign (columns)");
  std::vector<uint32_t> rows(rows64.begin(), rows64.end()), cols(cols64.begin(), cols64.end());
  std::sort(cols.begin(), cols.end()); cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
  std::vector<uint8_t> inrow(C->nrows ? C->nrows : 1, 0); for (auto r : rows) inrow[r] = 1;
  const uint64_t total = 0; (void)total;
  uint64_t nsel = 0; for (uint64_t i = 0; i < C->nrows; i++) nsel += inrow[i];
  if (nsel * cols.size() > 0xFFFFFFF0ull) fail(GrB_OUT_OF_MEMORY, "assign: region too large");
  // the scalar block T (host-built CSR, uploaded)
  const int ccode = C->type->code; const size_t ts = C->type->size;
  uint8_t s[16]; cast_scalar(ccode, s, xcode, x);
  std::vector<uint32_t> rp(C->nrows + 1, 0), cc; std::vector<uint8_t> vv;
  cc.reserve(nsel * cols.size()); vv.reserve(nsel * cols.size() * ts);
  for (uint64_t i = 0; i < C->nrows; i++) {
    if (inrow[i]) { cc.insert(cc.end(), cols.begin(), cols.end()); for (size_t q = 0; q < cols.size(); q++) vv.insert(vv.end(), s, s + ts); }
    rp[i + 1] = (uint32_t)cc.size();
  }
  DevCSR T; T.nrows = (uint32_t)C->nrows; T.ncols = (uint32_t)C->ncols; T.nnz = cc.size();
  T.rowptr.alloc(rp.size() * 4); T.col.alloc(cc.size() * 4 + 4); T.val.alloc(vv.size() + 8);
  GRB_HIP(hipMemcpyAsync(T.rowptr.p, rp.data(), rp.size() * 4, hipMemcpyHostToDevice, stream()));
  if (!cc.empty()) { GRB_HIP(hipMemcpyAsync(T.col.p, cc.data(), cc.size() * 4, hipMemcpyHostToDevice, stream()));
                     GRB_HIP(hipMemcpyAsync(T.val.p, vv.data(), vv.size(), hipMemcpyHostToDevice, stream())); }
  GRB_HIP(hipStreamSynchronize(stream())); T.valid = true;
  // assign keeps entries of C outside the region: Z = C with the region overwritten (or accumulated).
  // Expressed with the write-back merge by using FIRST/SECOND-style accumulation: with no accum the
  // region entries replace C's, entries of C outside stay => that is accum = SECOND on the union.
  GrB_BinaryOp_opaque second{GRB_MAGIC, B_SECOND, C->type, C->type, C->type, "assign_second", nullptr};
  matrix_write_back(C, T, ccode, M, dv, accum ? accum : &second, false);
}

}  // namespace

#define MAT_GUARD(C) if (!(C)) return GrB_NULL_POINTER; if (!check_obj(C)) return GrB_UNINITIALIZED_OBJECT

extern "C" {

GrB_Info GrB_mxm(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_Semiring semiring, const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc) {
  MAT_GUARD(C); if (!A || !B || !semiring) return GrB_NULL_POINTER;
  return guarded(C, [&] { do_mxm(C, Mask, accum, semiring, A, B, desc); });
}
GrB_Info GrB_transpose(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_Matrix A, const GrB_Descriptor desc) {
  MAT_GUARD(C); if (!A) return GrB_NULL_POINTER; return guarded(C, [&] { do_transpose(C, Mask, accum, A, desc); });
}
GrB_Info GrB_Matrix_eWiseAdd_BinaryOp(GrB_Matrix C, const GrB_Matrix M, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc) {
  MAT_GUARD(C); if (!op || !A || !B) return GrB_NULL_POINTER; return guarded(C, [&] { do_ewise(C, M, accum, op, A, B, desc, true); }); }
GrB_Info GrB_Matrix_eWiseAdd_Monoid(GrB_Matrix C, const GrB_Matrix M, const GrB_BinaryOp accum, const GrB_Monoid op, const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc) {
  MAT_GUARD(C); if (!op || !A || !B) return GrB_NULL_POINTER; if (!check_obj(op)) return GrB_UNINITIALIZED_OBJECT; return guarded(C, [&] { do_ewise(C, M, accum, op->op, A, B, desc, true); }); }
GrB_Info GrB_Matrix_eWiseAdd_Semiring(GrB_Matrix C, const GrB_Matrix M, const GrB_BinaryOp accum, const GrB_Semiring op, const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc) {
  MAT_GUARD(C); if (!op || !A || !B) return GrB_NULL_POINTER; if (!check_obj(op)) return GrB_UNINITIALIZED_OBJECT; return guarded(C, [&] { do_ewise(C, M, accum, op->add->op, A, B, desc, true); }); }
GrB_Info GrB_Matrix_eWiseMult_BinaryOp(GrB_Matrix C, const GrB_Matrix M, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc) {
  MAT_GUARD(C); if (!op || !A || !B) return GrB_NULL_POINTER; return guarded(C, [&] { do_ewise(C, M, accum, op, A, B, desc, false); }); }
GrB_Info GrB_Matrix_eWiseMult_Monoid(GrB_Matrix C, const GrB_Matrix M, const GrB_BinaryOp accum, const GrB_Monoid op, const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc) {
  MAT_GUARD(C); if (!op || !A || !B) return GrB_NULL_POINTER; if (!check_obj(op)) return GrB_UNINITIALIZED_OBJECT; return guarded(C, [&] { do_ewise(C, M, accum, op->op, A, B, desc, false); }); }
GrB_Info GrB_Matrix_eWiseMult_Semiring(GrB_Matrix C, const GrB_Matrix M, const GrB_BinaryOp accum, const GrB_Semiring op, const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc) {
  MAT_GUARD(C); if (!op || !A || !B) return GrB_NULL_POINTER; if (!check_obj(op)) return GrB_UNINITIALIZED_OBJECT; return guarded(C, [&] { do_ewise(C, M, accum, op->mul, A, B, desc, false); }); }
GrB_Info GrB_Matrix_apply(GrB_Matrix C, const GrB_Matrix M, const GrB_BinaryOp accum, const GrB_UnaryOp op, const GrB_Matrix A, const GrB_Descriptor desc) {
  MAT_GUARD(C); if (!op || !A) return GrB_NULL_POINTER; if (!check_obj(op)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(C, [&] { if (op->opcode >= U_USER) not_implemented("user-defined unary operator"); do_apply(C, M, accum, 0, op->opcode, op->xtype->code, nullptr, 0, A, desc); });
}
GrB_Info GxB_Matrix_select(GrB_Matrix C, const GrB_Matrix M, const GrB_BinaryOp accum, const GxB_SelectOp op, const GrB_Matrix A, const GxB_Scalar thunk, const GrB_Descriptor desc) {
  MAT_GUARD(C); if (!op || !A) return GrB_NULL_POINTER; return guarded(C, [&] { do_select(C, M, accum, op, A, thunk, desc); });
}
GrB_Info GrB_Matrix_reduce_Monoid(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Monoid monoid, const GrB_Matrix A, const GrB_Descriptor desc) {
  if (!w || !A || !monoid) return GrB_NULL_POINTER; if (!check_obj(w)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(w, [&] { do_reduce_vector(w, mask, accum, monoid, A, desc); });
}

#define GRB_TYPED_MATOPS(SUF, CT, CODE) \
  GrB_Info GrB_Matrix_assign_##SUF(GrB_Matrix C, const GrB_Matrix M, const GrB_BinaryOp accum, CT x, const GrB_Index* I, GrB_Index ni, const GrB_Index* J, GrB_Index nj, const GrB_Descriptor desc) { \
    MAT_GUARD(C); return guarded(C, [&] { do_assign_scalar(C, M, accum, &x, CODE, I, ni, J, nj, desc); }); } \
  GrB_Info GxB_Matrix_apply_BinaryOp1st_##SUF(GrB_Matrix C, const GrB_Matrix M, const GrB_BinaryOp accum, const GrB_BinaryOp op, CT x, const GrB_Matrix A, const GrB_Descriptor desc) { \
    MAT_GUARD(C); if (!op || !A) return GrB_NULL_POINTER; return guarded(C, [&] { check_binop(op, "apply"); do_apply(C, M, accum, 1, op->opcode, op->xtype->code, &x, CODE, A, desc); }); } \
  GrB_Info GxB_Matrix_apply_BinaryOp2nd_##SUF(GrB_Matrix C, const GrB_Matrix M, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Matrix A, CT y, const GrB_Descriptor desc) { \
    MAT_GUARD(C); if (!op || !A) return GrB_NULL_POINTER; return guarded(C, [&] { check_binop(op, "apply"); do_apply(C, M, accum, 2, op->opcode, op->xtype->code, &y, CODE, A, desc); }); }
GRB_TYPED_MATOPS(BOOL, bool, T_BOOL) GRB_TYPED_MATOPS(INT8, int8_t, T_INT8) GRB_TYPED_MATOPS(UINT8, uint8_t, T_UINT8)
GRB_TYPED_MATOPS(INT16, int16_t, T_INT16) GRB_TYPED_MATOPS(UINT16, uint16_t, T_UINT16) GRB_TYPED_MATOPS(INT32, int32_t, T_INT32)
GRB_TYPED_MATOPS(UINT32, uint32_t, T_UINT32) GRB_TYPED_MATOPS(INT64, int64_t, T_INT64) GRB_TYPED_MATOPS(UINT64, uint64_t, T_UINT64)
GRB_TYPED_MATOPS(FP32, float, T_FP32) GRB_TYPED_MATOPS(FP64, double, T_FP64)

}  // extern "C"
