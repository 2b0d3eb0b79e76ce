// grb_vector_ops.cpp — the O(n) vector operations the reference's BFS / PageRank loops wrap around
// the hot path, kept in HBM so an iteration never leaves the device:
//   GrB_Vector_reduce_<T>            <- Vector.reduce_bool/int/float   (pygraphblas/vector.py:1101-1202)
//   GrB_Matrix_reduce_<T>            <- Matrix.reduce_int              (pygraphblas/matrix.py:1782-1804)
//   GrB_Vector_assign_<T>            <- Vector.assign_scalar           (pygraphblas/vector.py:1494-1524)
//   GrB_Vector_eWiseAdd/eWiseMult_*  <- Vector.eadd / emult            (pygraphblas/vector.py:604-833)
//   GrB_Vector_apply, GxB_Vector_apply_BinaryOp1st/2nd, GxB_Vector_select  (pygraphblas/vector.py:1204-1340)
#include <algorithm>
#include "grb_opcommon.hpp"
#include "grb_lazy.hpp"
#include "grb_matops.hpp"

using namespace grb;

// scalar c = accum ? accum(c, s) : s, with s in type scode and c in type ccode (host)
static void scalar_accum(void* c, int ccode, const void* s, int scode, GrB_BinaryOp accum) {
  if (!accum) { cast_scalar(ccode, c, scode, s); return; }
  check_binop(accum, "accum");
  const int ac = accum->xtype->code; uint8_t a[16], b[16], z[16];
  cast_scalar(ac, a, ccode, c); cast_scalar(ac, b, scode, s);
  dispatch_type(ac, [&]<class T>() { T x, y; memcpy(&x, a, sizeof(T)); memcpy(&y, b, sizeof(T)); T r = apply_binop<T>(accum->opcode, x, y); memcpy(z, &r, sizeof(T)); });
  cast_scalar(ccode, c, ac, z);
}

// a NaN result of an FP MIN / MAX reduction that started from NaN: every value was NaN — or there was none, and the answer is the monoid's identity
static void nan_or_empty(uint8_t* r, int mc, GrB_Monoid monoid, const uint8_t* pres, uint64_t n) {
  bool is_nan = false;
  if (mc == T_FP32) { float f; memcpy(&f, r, 4); is_nan = f != f; } else { double f; memcpy(&f, r, 8); is_nan = f != f; }
  if (is_nan && (n == 0 || (pres && count_present(pres, n) == 0))) memcpy(r, monoid->identity, 16);
}

static void reduce_common(void* c, int ccode, GrB_BinaryOp accum, GrB_Monoid monoid, int vcode, const void* val, const uint8_t* pres, uint64_t n) {
  need_device();
  if (!check_obj(monoid)) fail(GrB_UNINITIALIZED_OBJECT, "monoid is not initialised");
  check_binop(monoid->op, "monoid");
  const int mc = monoid->op->ztype->code;
  uint8_t r[16];
  const int mop = monoid->op->opcode;
  uint8_t id[16]; memcpy(id, monoid->identity, 16);
  const bool nan_id = fp_minmax_identity(mc, mop, id);                        // FP MIN / MAX: start from NaN = from the first value (grb_opcommon.hpp)
  if (mc == T_FP64 && vcode == T_FP32 && (mop == B_PLUS || mop == B_MIN || mop == B_MAX || mop == B_TIMES)) {
    reduce_values_f32_f64(n, val, pres, mop, id, r);                        // no cast pass: the first reduction level widens on the fly
  } else {
    DevBuf tmp; const void* v = cast_values(mc, vcode, val, n, tmp);
    reduce_values(mc, n, v, pres, mop, id, r);
  }
  if (nan_id) nan_or_empty(r, mc, monoid, pres, n);
  scalar_accum(c, ccode, r, mc, accum);
}

// a container whose dimensions exceed the device layout (hypersparse, GxB_INDEX_MAX by default: `Matrix.sparse(INT8)`,
// pygraphblas/matrix.py:1785-1793) still reduces in HBM: a reduction needs the stored values only, not their coordinates
static void reduce_host_values(void* c, int ccode, GrB_BinaryOp accum, GrB_Monoid monoid, GrB_Type type, const std::vector<uint8_t>& hx) {
  if (type->code >= T_FC32) fail(GrB_DOMAIN_MISMATCH, "reduce: complex values are out of scope");
  need_device();
  const uint64_t n = hx.size() / type->size; DevBuf d; d.alloc(hx.empty() ? 8 : hx.size());
  if (n) { GRB_HIP(hipMemcpyAsync(d.p, hx.data(), hx.size(), hipMemcpyHostToDevice, stream())); GRB_HIP(hipStreamSynchronize(stream())); }
  reduce_common(c, ccode, accum, monoid, type->code, d.p, nullptr, n);
}

static GrB_Info vec_reduce(void* c, int ccode, GrB_BinaryOp accum, GrB_Monoid monoid, GrB_Vector u) {
  if (!c || !u || !monoid) return GrB_NULL_POINTER; if (!check_obj(u)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(u, [&] {
    if (u->n > GRB_DIM_DEVICE_MAX) { vec_to_host(u); reduce_host_values(c, ccode, accum, monoid, u->type, u->hx); return; }
    if (u->lazy == 2 && check_obj(monoid) && check_obj(monoid->op) && monoid->op->opcode < B_FIRSTI && monoid->op->xtype == monoid->op->ytype) {
      // u is the result of queued element-wise operations: their one kernel reduces it on the way (`t -= r; abs(t); reduce_float()`)
      const int mc = monoid->op->ztype->code; uint8_t r[16] = {0}, id[16]; memcpy(id, monoid->identity, 16);
      const bool nan_id = fp_minmax_identity(mc, monoid->op->opcode, id);
      if (lazy_reduce(u, monoid->op->opcode, mc, id, r, !nan_id)) {
        if (nan_id) nan_or_empty(r, mc, monoid, u->dpres.as<uint8_t>(), u->n);      // (u is materialised now: may_keep was false)
        scalar_accum(c, ccode, r, mc, accum); return;
      }
    }
    if (u->lor_state && u->type->code == T_BOOL && check_obj(monoid) && check_obj(monoid->op) && monoid->op->opcode == B_LOR && monoid->op->ztype->code == T_BOOL) {
      bool any = false;                                  // the kernel that produced u already noted it (grb_mxv.cpp)
      if (any_true_lookup(u, &any)) { const uint8_t r = any ? 1 : 0; scalar_accum(c, ccode, &r, T_BOOL, accum); return; }
    }
    if (u->host_valid && !u->lazy && !u->q_reads && !u->iso_full && u->type->code == T_BOOL && check_obj(monoid) && check_obj(monoid->op) &&
        (monoid->op->opcode == B_LOR || monoid->op->opcode == B_LAND) && monoid->op->ztype->code == T_BOOL) {
      // the entries are on the host (`q[start] = True` in front of the loop): LOR / LAND over them needs no device at all
      vec_host_assemble(u);
      const bool want = monoid->op->opcode == B_LOR; bool hit = false;
      for (uint8_t x : u->hx) if ((x != 0) == want) { hit = true; break; }
      const uint8_t r = want ? hit : !hit; scalar_accum(c, ccode, &r, T_BOOL, accum); return;
    }
    vec_to_device(u); reduce_common(c, ccode, accum, monoid, u->type->code, u->dval.p, u->dpres.as<uint8_t>(), u->n); });
}
static GrB_Info mat_reduce(void* c, int ccode, GrB_BinaryOp accum, GrB_Monoid monoid, GrB_Matrix A) {
  if (!c || !A || !monoid) return GrB_NULL_POINTER; if (!check_obj(A)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(A, [&] {
    if (A->nrows > GRB_DIM_DEVICE_MAX || A->ncols > GRB_DIM_DEVICE_MAX) { mat_to_host(A); reduce_host_values(c, ccode, accum, monoid, A->type, A->hx); return; }
    mat_to_device(A); reduce_common(c, ccode, accum, monoid, A->type->code, A->csr.val.p, nullptr, A->csr.nnz); });
}

static void vec_assign(GrB_Vector w, GrB_Vector mask, GrB_BinaryOp accum, const void* x, int xcode, const GrB_Index* I, GrB_Index ni, GrB_Descriptor desc) {
  if (w->n > GRB_DIM_DEVICE_MAX || w->type->code >= T_FC32) { host_assign_scalar(w, mask, accum, x, xcode, I, ni, desc); return; }   // no HBM layout: host mirror
  need_device();
  if (mask && !check_obj(mask)) fail(GrB_UNINITIALIZED_OBJECT, "assign: mask is not initialised");
  const DescView dv(desc);
  if (mask && mask->n != w->n) fail(GrB_DIMENSION_MISMATCH, "assign: mask size");
  if (accum) check_binop(accum, "accum");
  const uint64_t n = w->n;
  if (I == GrB_ALL && mask && mask != w && n) {
    // every index under a mask (`v.assign_scalar(level, mask=q)`, the BFS loop): the kernel reads the mask vector itself
    const int wcode0 = w->type->code; const int ecode0 = accum ? accum->xtype->code : wcode0;
    if (ecode0 == wcode0) {
      vec_to_device(mask); vec_to_device(w);
      uint8_t s0[16]; cast_scalar(wcode0, s0, xcode, x);
      // when both the mask's and w's entries are known as short lists (the first level of a BFS: `v[q] = 1` with q = {start}), so are
      // the result's: w's entries and the positions the mask allows
      std::vector<uint32_t> merged; bool keep_list = false, truthy = false;
      if (!accum && !dv.replace && !dv.mask_comp && w->small_valid && mask->small_valid && (dv.mask_struct || mask->small_truthy)) {
        merged.resize(w->small_idx.size() + mask->small_idx.size());
        merged.resize(std::set_union(w->small_idx.begin(), w->small_idx.end(), mask->small_idx.begin(), mask->small_idx.end(), merged.begin()) - merged.begin());
        bool snz = false; for (size_t b = 0; b < type_size(wcode0); b++) snz = snz || s0[b] != 0;
        keep_list = merged.size() <= 64; truthy = snz && (w->small_truthy || w->small_idx.empty());
      }
      // (Round 5, measured and dropped: recording this assign and letting the masked pull that follows apply it on its way over the rows — one launch and
      //  one dependent kernel less per BFS level.  The pull then looks at q beside v wherever it gathers an operand entry: +2 byte gathers per entry
      //  made the level-2 pull of the R-MAT-22 BFS 65 -> 106 us and the whole loop 283 -> 341 us.)
      // a one-byte-typed output (the level vector of a BFS): its code bytes come out of the same pass (the pull that follows gathers ONE byte per neighbour)
      uint8_t* code_out = nullptr;
      if (type_size(wcode0) == 1 && n >= (1u << 16)) { if (!w->dcode.p || w->dcode.bytes < n + 16) w->dcode.alloc(n + 16); code_out = w->dcode.as<uint8_t>(); }
      const bool coded = vec_assign_scalar_masked(wcode0, n, w->dval.p, w->dpres.as<uint8_t>(), mask->type->code, mask->dval.p, mask->dpres.as<uint8_t>(), dv.mask_struct, dv.mask_comp, s0,
                               accum ? accum->opcode : -1, dv.replace, code_out);
      vec_invalidate_host(w);
      w->code_valid = coded;
      if (dv.replace) { w->fe_lb = 0; w->fe_lb_key = 0; }
      // the positions the mask allows are entries of w now: when the mask carries a bound of the edges leaving its TRUE entries (left by the
      // product that made it, any_true_lookup), w's entries include them — `v[q] = level`: the next product's direction choice needs no count
      // (a bound that counts every PRESENT entry of the mask says nothing about the positions a valued mask allows: ADVICE round 5)
      if (!dv.mask_comp && !dv.replace && (dv.mask_struct || mask->fe_lb_true) && mask->fe_lb_key && mask->fe_lb && mask->lazy == 0 && (w->fe_lb_key != mask->fe_lb_key || w->fe_lb < mask->fe_lb)) { w->fe_lb = mask->fe_lb; w->fe_lb_key = mask->fe_lb_key; w->fe_lb_true = false; }
      w->dnvals_known = false; w->dnvals = 0;
      if (keep_list) { w->small_idx.swap(merged); w->small_valid = true; w->small_truthy = truthy; w->dnvals = w->small_idx.size(); w->dnvals_known = true; }
      return;
    }
  }
  DevBuf allow_buf, region; bool nothing = false;
  const uint8_t* allow = vector_allow(mask, dv, n, allow_buf, &nothing);
  if (nothing) { if (dv.replace) { GrB_Vector_clear(w); } return; }
  if (I == GrB_ALL && !mask && !accum) {                     // `w(:) = s`: a note on the vector in non-blocking mode (grb_lazy.cpp)
    uint8_t sw[16] = {0}; cast_scalar(w->type->code, sw, xcode, x);
    if (lazy_fill(w, sw)) return;
  }
  vec_to_device(w);
  const uint8_t* reg = nullptr;
  if (I != GrB_ALL) {
    std::vector<uint8_t> h(n ? n : 1, 0);
    if (ni) for (uint64_t i : expand_index_list(I, ni, n, "assign")) h[i] = 1;
    region.alloc(n ? n : 1);
    GRB_HIP(hipMemcpyAsync(region.p, h.data(), n, hipMemcpyHostToDevice, stream())); GRB_HIP(hipStreamSynchronize(stream()));
    reg = region.as<uint8_t>();
  }
  const int wcode = w->type->code;
  const int ecode = accum ? accum->xtype->code : wcode;
  uint8_t s[16]; cast_scalar(ecode, s, xcode, x);
  if (ecode == wcode) {
    vec_assign_scalar(wcode, n, w->dval.p, w->dpres.as<uint8_t>(), allow, reg, s, accum ? accum->opcode : -1, dv.replace);
  } else {
    DevBuf wc(n * type_size(ecode) + 1);
    vec_cast_values(ecode, wc.p, wcode, w->dval.p, n);
    vec_assign_scalar(ecode, n, wc.p, w->dpres.as<uint8_t>(), allow, reg, s, accum->opcode, dv.replace);
    vec_cast_values(wcode, w->dval.p, ecode, wc.p, n);
  }
  vec_invalidate_host(w);
  if (dv.replace) { w->fe_lb = 0; w->fe_lb_key = 0; }                      // (without replace a scalar assign only adds entries: the bound stays)
  w->dnvals_known = !allow && !reg; w->dnvals = w->dnvals_known ? n : 0;       // every index, no mask: the vector is full now
}

static void vec_ewise_op(GrB_Vector w, GrB_Vector mask, GrB_BinaryOp accum, GrB_BinaryOp op, GrB_Vector u, GrB_Vector v,
                         GrB_Descriptor desc, bool is_union) {
  need_device();
  if (!check_obj(u) || !check_obj(v) || (mask && !check_obj(mask))) fail(GrB_UNINITIALIZED_OBJECT, "eWise: uninitialised operand");
  check_binop(op, "eWise");
  if (is_hyper(w)) { hyper_vec_ewise(w, mask, accum, op, u, v, desc, is_union); return; }      // a size beyond the device layout
  const DescView dv(desc); const uint64_t n = w->n;
  if (u->n != n || v->n != n || (mask && mask->n != n)) fail(GrB_DIMENSION_MISMATCH, "eWise: vector sizes differ");
  if (!mask && dv.mask_comp) { if (dv.replace) GrB_Vector_clear(w); return; }      // no mask, complemented: nothing may be written
  if (!mask && !accum && lazy_ewise(w, op, u, v, is_union)) return;      // queued: runs fused with its neighbours when a result is looked at
  const int xc = op->xtype->code;
  // one pass when everything works in w's own type (round 6): the mask read in place, T(i) in a register, the write-back in the same store
  {
    static const bool off = getenv("GRB_MI355X_EWISE_FUSED") && atoi(getenv("GRB_MI355X_EWISE_FUSED")) == 0;      // measurement / test hook
    const int wc = w->type->code;
    const bool same = xc == wc && op->ytype->code == wc && op->ztype->code == wc && u->type->code == wc && v->type->code == wc && wc < T_FC32 &&
                      (!accum || (check_obj(accum) && accum->xtype->code == wc && accum->ytype->code == wc && accum->ztype->code == wc)) &&
                      (!mask || mask->type->code < T_FC32);
    if (!off && same && (mask || accum) && n) {
      if (accum) check_binop(accum, "accum");
      vec_to_device(u); vec_to_device(v); if (mask) vec_to_device(mask); vec_to_device(w);
      const bool uf = u->dnvals_known && u->dnvals == n, vf = v->dnvals_known && v->dnvals == n, wf = w->dnvals_known && w->dnvals == n;
      const bool t_full = is_union ? (uf || vf) : (uf && vf);
      const bool stays_full = !mask && accum && (wf || t_full);
      vec_ewise_fused(wc, n, u->dval.p, u->dpres.as<uint8_t>(), v->dval.p, v->dpres.as<uint8_t>(), op->opcode, is_union,
                      mask ? mask->type->code : 0, mask ? mask->dval.p : nullptr, mask ? mask->dpres.as<uint8_t>() : nullptr, dv.mask_struct, dv.mask_comp,
                      accum ? accum->opcode : -1, dv.replace, w->dval.p, w->dpres.as<uint8_t>());
      vec_invalidate_host(w);
      w->fe_lb = 0; w->fe_lb_key = 0;
      w->dnvals_known = stays_full; w->dnvals = stays_full ? n : 0;
      return;
    }
  }
  DevBuf allow_buf; bool nothing = false;
  const uint8_t* allow = vector_allow(mask, dv, n, allow_buf, &nothing);
  if (nothing) { if (dv.replace) GrB_Vector_clear(w); return; }
  vec_to_device(u); vec_to_device(v);
  DevBuf uc, vc, tval(n * type_size(xc) + 1), tpres(n + 1);
  const void* uv = cast_values(xc, u->type->code, u->dval.p, n, uc);
  const void* vv = cast_values(xc, v->type->code, v->dval.p, n, vc);
  vec_ewise(xc, n, uv, u->dpres.as<uint8_t>(), vv, v->dpres.as<uint8_t>(), op->opcode, is_union, tval.p, tpres.as<uint8_t>());
  // a comparison yields 0/1 in the operand type: identical to BOOL after the typecast into w
  const bool uf = u->dnvals_known && u->dnvals == n, vf = v->dnvals_known && v->dnvals == n;
  const uint64_t tn = (is_union ? (uf || vf) : (uf && vf)) ? n : ~0ull;          // a full operand makes the union full, two make the intersection full
  vector_write_back(w, xc, tval, tpres, allow, accum, dv.replace, /*t_only_allowed=*/false, tn);
}

// mode 0: unary op; 1: z = f(s, x); 2: z = f(x, s)
static void vec_apply_op(GrB_Vector w, GrB_Vector mask, GrB_BinaryOp accum, int mode, int opcode, int opxcode, int opzcode, const void* scalar, int scode,
                         GrB_Vector u, GrB_Descriptor desc) {
  need_device();
  if (!check_obj(u) || (mask && !check_obj(mask))) fail(GrB_UNINITIALIZED_OBJECT, "apply: uninitialised operand");
  const DescView dv(desc); const uint64_t n = w->n;
  if (u->n != n || (mask && mask->n != n)) fail(GrB_DIMENSION_MISMATCH, "apply: vector sizes differ");
  DevBuf allow_buf; bool nothing = false;
  const uint8_t* allow = vector_allow(mask, dv, n, allow_buf, &nothing);
  if (nothing) { if (dv.replace) GrB_Vector_clear(w); return; }
  uint8_t s[16] = {0}; if (scalar) cast_scalar(opxcode, s, scode, scalar);
  if (!mask && !accum && lazy_apply(w, mode, opcode, opxcode, opzcode, s, u)) return;
  vec_to_device(u);
  DevBuf uc, tval(n * type_size(opxcode) + 1), tpres(n + 1);
  const void* uv = cast_values(opxcode, u->type->code, u->dval.p, n, uc);
  vec_apply(opxcode, n, uv, u->dpres.as<uint8_t>(), mode, opcode, s, tval.p, tpres.as<uint8_t>());
  vector_write_back(w, opxcode, tval, tpres, allow, accum, dv.replace, false, u->dnvals_known ? u->dnvals : ~0ull);     // apply keeps the pattern
}

// positional unary operators on a vector (an n x 1 column): the pattern of u, the values are the index (which 0 / 1: + 1) or the column 0 (2 / 3: + 1)
static void vec_position_op(GrB_Vector w, GrB_Vector mask, GrB_BinaryOp accum, int which, int zcode, GrB_Vector u, GrB_Descriptor desc) {
  need_device();
  if (!check_obj(u) || (mask && !check_obj(mask))) fail(GrB_UNINITIALIZED_OBJECT, "apply: uninitialised operand");
  const DescView dv(desc); const uint64_t n = w->n;
  if (u->n != n || (mask && mask->n != n)) fail(GrB_DIMENSION_MISMATCH, "apply: vector sizes differ");
  DevBuf allow_buf; bool nothing = false;
  const uint8_t* allow = vector_allow(mask, dv, n, allow_buf, &nothing);
  if (nothing) { if (dv.replace) GrB_Vector_clear(w); return; }
  vec_to_device(u);
  DevBuf tval(n * type_size(zcode) + 1), tpres(n + 1);
  vec_position_values(zcode, n, which, tval.p);
  if (n) GRB_HIP(hipMemcpyAsync(tpres.p, u->dpres.p, n, hipMemcpyDeviceToDevice, stream()));
  vector_write_back(w, zcode, tval, tpres, allow, accum, dv.replace, false, u->dnvals_known ? u->dnvals : ~0ull);
}

#define VEC_GUARD(w) if (!(w)) return GrB_NULL_POINTER; if (!check_obj(w)) return GrB_UNINITIALIZED_OBJECT

extern "C" {

GrB_Info GrB_Vector_eWiseAdd_BinaryOp(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Vector u, const GrB_Vector v, const GrB_Descriptor desc) {
  VEC_GUARD(w); if (!op || !u || !v) return GrB_NULL_POINTER; return guarded(w, [&] { vec_ewise_op(w, mask, accum, op, u, v, desc, true); }); }
GrB_Info GrB_Vector_eWiseAdd_Monoid(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Monoid op, const GrB_Vector u, const GrB_Vector v, const GrB_Descriptor desc) {
  VEC_GUARD(w); if (!op || !u || !v) return GrB_NULL_POINTER; if (!check_obj(op)) return GrB_UNINITIALIZED_OBJECT; return guarded(w, [&] { vec_ewise_op(w, mask, accum, op->op, u, v, desc, true); }); }
GrB_Info GrB_Vector_eWiseAdd_Semiring(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Semiring op, const GrB_Vector u, const GrB_Vector v, const GrB_Descriptor desc) {
  VEC_GUARD(w); if (!op || !u || !v) return GrB_NULL_POINTER; if (!check_obj(op)) return GrB_UNINITIALIZED_OBJECT; return guarded(w, [&] { vec_ewise_op(w, mask, accum, op->add->op, u, v, desc, true); }); }
GrB_Info GrB_Vector_eWiseMult_BinaryOp(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Vector u, const GrB_Vector v, const GrB_Descriptor desc) {
  VEC_GUARD(w); if (!op || !u || !v) return GrB_NULL_POINTER; return guarded(w, [&] { vec_ewise_op(w, mask, accum, op, u, v, desc, false); }); }
GrB_Info GrB_Vector_eWiseMult_Monoid(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Monoid op, const GrB_Vector u, const GrB_Vector v, const GrB_Descriptor desc) {
  VEC_GUARD(w); if (!op || !u || !v) return GrB_NULL_POINTER; if (!check_obj(op)) return GrB_UNINITIALIZED_OBJECT; return guarded(w, [&] { vec_ewise_op(w, mask, accum, op->op, u, v, desc, false); }); }
GrB_Info GrB_Vector_eWiseMult_Semiring(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Semiring op, const GrB_Vector u, const GrB_Vector v, const GrB_Descriptor desc) {
  VEC_GUARD(w); if (!op || !u || !v) return GrB_NULL_POINTER; if (!check_obj(op)) return GrB_UNINITIALIZED_OBJECT; return guarded(w, [&] { vec_ewise_op(w, mask, accum, op->mul, u, v, desc, false); }); }

// "Same size, same pattern, equal values" of two vectors of ONE built-in real type as a single pass (what pygraphblas/vector.py:188-235 `Vector.iseq` composes from
// five calls).  GrB_NO_VALUE: not this function's case (types differ, complex, a size beyond the device layout, no device) — the caller composes it as before.
GrB_Info GrBX_Vector_iseq(bool* equal, const GrB_Vector u, const GrB_Vector v) {
  if (!equal || !u || !v) return GrB_NULL_POINTER; if (!check_obj(u) || !check_obj(v)) return GrB_UNINITIALIZED_OBJECT;
  if (u->type != v->type || u->type->code >= T_FC32 || is_hyper(u) || is_hyper(v) || !device_ok() || u->iso_full || v->iso_full) return GrB_NO_VALUE;
  if (u->n != v->n) { *equal = false; return GrB_SUCCESS; }
  if (u == v) { *equal = true; return GrB_SUCCESS; }
  return guarded(const_cast<GrB_Vector>(u), [&] {
    vec_to_device(const_cast<GrB_Vector>(u)); vec_to_device(const_cast<GrB_Vector>(v));
    if (u->dnvals_known && v->dnvals_known && u->dnvals != v->dnvals) { *equal = false; return; }
    *equal = vec_iseq_mismatches(u->type->code, u->n, u->dval.p, u->dpres.as<uint8_t>(), v->dval.p, v->dpres.as<uint8_t>()) == 0;
  });
}

GrB_Info GrB_Vector_apply(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_UnaryOp op, const GrB_Vector u, const GrB_Descriptor desc) {
  VEC_GUARD(w); if (!op || !u) return GrB_NULL_POINTER; if (!check_obj(op)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(w, [&] { if (op->opcode >= U_USER) not_implemented("user-defined unary operator");
    if (op->opcode >= U_POSITIONI) { vec_position_op(w, mask, accum, op->opcode - U_POSITIONI, op->ztype->code, u, desc); return; }
    vec_apply_op(w, mask, accum, 0, op->opcode, op->xtype->code, op->ztype->code, nullptr, 0, u, desc); });
}

GrB_Info GxB_Vector_select(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GxB_SelectOp op, const GrB_Vector u, const GxB_Scalar thunk, const GrB_Descriptor desc) {
  VEC_GUARD(w); if (!op || !u) return GrB_NULL_POINTER; if (!check_obj(op) || !check_obj(u)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(w, [&] {
    need_device();
    const DescView dv(desc); const uint64_t n = w->n;
    if (u->n != n || (mask && mask->n != n)) fail(GrB_DIMENSION_MISMATCH, "select: vector sizes differ");
    if (op->opcode == SEL_USER) not_implemented("user-defined select operator");
    DevBuf allow_buf; bool nothing = false;
    const uint8_t* allow = vector_allow(mask, dv, n, allow_buf, &nothing);
    if (nothing) { if (dv.replace) GrB_Vector_clear(w); return; }
    vec_to_device(u);
    const int uc = u->type->code; const size_t ts = u->type->size;
    DevBuf tval(n * ts + 1), tpres(n + 1);
    GRB_HIP(hipMemcpyAsync(tval.p, u->dval.p, n * ts, hipMemcpyDeviceToDevice, stream()));
    int64_t k = 0; uint8_t th[16] = {0};
    if (thunk && check_obj(thunk) && thunk->has) { cast_scalar(T_INT64, &k, thunk->type->code, thunk->x); cast_scalar(uc, th, thunk->type->code, thunk->x); }
    if (op->opcode <= SEL_OFFDIAG) {
      // positional on an n x 1 column: entry (i, 0);  tril: 0 <= i + k ... keep iff j - i <= k etc.
      std::vector<uint8_t> up(n), keep(n);
      GRB_HIP(hipMemcpyAsync(up.data(), u->dpres.p, n, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
      for (uint64_t i = 0; i < n; i++) { const int64_t d = 0 - (int64_t)i; bool kp = false;
        switch (op->opcode) { case SEL_TRIL: kp = d <= k; break; case SEL_TRIU: kp = d >= k; break; case SEL_DIAG: kp = d == k; break; default: kp = d != k; }
        keep[i] = up[i] && kp; }
      GRB_HIP(hipMemcpyAsync(tpres.p, keep.data(), n, hipMemcpyHostToDevice, stream())); GRB_HIP(hipStreamSynchronize(stream()));
    } else {
      select_value_flags(uc, n, u->dval.p, u->dpres.as<uint8_t>(), op->opcode, th, tpres.as<uint8_t>());
    }
    vector_write_back(w, uc, tval, tpres, allow, accum, dv.replace, false);
  });
}

#define GRB_TYPED_VECOPS(SUF, CT, CODE) \
  GrB_Info GrB_Vector_reduce_##SUF(CT* c, const GrB_BinaryOp accum, const GrB_Monoid monoid, const GrB_Vector u, const GrB_Descriptor desc) { (void)desc; return vec_reduce(c, CODE, accum, monoid, u); } \
  GrB_Info GrB_Matrix_reduce_##SUF(CT* c, const GrB_BinaryOp accum, const GrB_Monoid monoid, const GrB_Matrix A, const GrB_Descriptor desc) { (void)desc; return mat_reduce(c, CODE, accum, monoid, A); } \
  GrB_Info GrB_Vector_assign_##SUF(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, CT x, const GrB_Index* I, GrB_Index ni, const GrB_Descriptor desc) { \
    VEC_GUARD(w); return guarded(w, [&] { vec_assign(w, mask, accum, &x, CODE, I, ni, desc); }); } \
  GrB_Info GxB_Vector_apply_BinaryOp1st_##SUF(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, CT x, const GrB_Vector u, const GrB_Descriptor desc) { \
    VEC_GUARD(w); if (!op || !u) return GrB_NULL_POINTER; return guarded(w, [&] { check_binop(op, "apply"); vec_apply_op(w, mask, accum, 1, op->opcode, op->xtype->code, op->ztype->code, &x, CODE, u, desc); }); } \
  GrB_Info GxB_Vector_apply_BinaryOp2nd_##SUF(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Vector u, CT y, const GrB_Descriptor desc) { \
    VEC_GUARD(w); if (!op || !u) return GrB_NULL_POINTER; return guarded(w, [&] { check_binop(op, "apply"); vec_apply_op(w, mask, accum, 2, op->opcode, op->xtype->code, op->ztype->code, &y, CODE, u, desc); }); }
GRB_TYPED_VECOPS(BOOL, bool, T_BOOL) GRB_TYPED_VECOPS(INT8, int8_t, T_INT8) GRB_TYPED_VECOPS(UINT8, uint8_t, T_UINT8)
GRB_TYPED_VECOPS(INT16, int16_t, T_INT16) GRB_TYPED_VECOPS(UINT16, uint16_t, T_UINT16) GRB_TYPED_VECOPS(INT32, int32_t, T_INT32)
GRB_TYPED_VECOPS(UINT32, uint32_t, T_UINT32) GRB_TYPED_VECOPS(INT64, int64_t, T_INT64) GRB_TYPED_VECOPS(UINT64, uint64_t, T_UINT64)
GRB_TYPED_VECOPS(FP32, float, T_FP32) GRB_TYPED_VECOPS(FP64, double, T_FP64)

GrB_Info GxB_Vector_fprint(GrB_Vector v, const char* name, int pr, FILE* f) {
  if (!v) return GrB_NULL_POINTER; if (!check_obj(v)) return GrB_UNINITIALIZED_OBJECT; if (pr <= 0) return GrB_SUCCESS;
  return guarded(v, [&] { fprintf(f ? f : stdout, "\n  GraphBLAS vector: %s  type %s  size %llu  nvals %llu  (MI355X bitmap layout)\n", name ? name : "",
                                 v->type->name, (unsigned long long)v->n, (unsigned long long)vec_nvals(v)); });
}
GrB_Info GxB_Matrix_fprint(GrB_Matrix A, const char* name, int pr, FILE* f) {
  if (!A) return GrB_NULL_POINTER; if (!check_obj(A)) return GrB_UNINITIALIZED_OBJECT; if (pr <= 0) return GrB_SUCCESS;
  return guarded(A, [&] { fprintf(f ? f : stdout, "\n  GraphBLAS matrix: %s  type %s  %llu-by-%llu  nvals %llu  (MI355X CSR layout)\n", name ? name : "",
                                 A->type->name, (unsigned long long)A->nrows, (unsigned long long)A->ncols, (unsigned long long)mat_nvals(A)); });
}

}  // extern "C"
