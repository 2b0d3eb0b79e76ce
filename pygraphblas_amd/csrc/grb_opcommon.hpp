// grb_opcommon.hpp — pieces every GraphBLAS operation driver shares: operator validation, operand
// typecasts into the operator's domain, mask -> allow bytes, and the final
// C<M,replace> = accum(C, T) write-back for vectors.
#pragma once
#include "grb_api.hpp"
#include "grb_device.hpp"
#include "grb_semiring.hpp"
#include <vector>

namespace grb {

inline void not_implemented(const std::string& what) { fail(GrB_INVALID_VALUE, "not implemented in the MI355X backend: " + what); }

// built-in, same-type binary operator (x, y, z all one real type) or a comparison whose inputs share a type
inline void check_binop(GrB_BinaryOp op, const char* what) {
  if (!check_obj(op)) fail(GrB_UNINITIALIZED_OBJECT, std::string(what) + " operator is not initialised");
  if (op->opcode >= B_FIRSTI) not_implemented(std::string("positional / user-defined operator ") + op->name);
  if (op->xtype != op->ytype) not_implemented(std::string("mixed-type operator ") + op->name);
}

// ONE rule for NaN under a floating-point MIN / MAX monoid, on every path: the operator is fmin / fmax (a NaN operand is omitted, as in
// SuiteSparse), and an entry all of whose products are NaN is NaN — what a reduction that starts from its first product gives
// (the oracle's rule).  Kernels that start their accumulators from the monoid's identity (push SpMSpV, hash tables, padding lanes
// of wave reductions) get the same answer when that identity is NaN instead of +-inf: fmin(NaN, x) = x, fmin(NaN, NaN) = NaN — NaN
// IS the identity of fmin / fmax.  (Round 2: +-inf there dropped such an entry's NaN on some paths and kept it on others.)
inline bool fp_minmax_identity(int zcode, int addop, uint8_t* identity) {
  if (!(addop == B_MIN || addop == B_MAX)) return false;
  if (zcode == T_FP32) { const float q = std::numeric_limits<float>::quiet_NaN(); memcpy(identity, &q, 4); return true; }
  if (zcode == T_FP64) { const double q = std::numeric_limits<double>::quiet_NaN(); memcpy(identity, &q, 8); return true; }
  return false;
}

inline SemiringDesc make_semiring_desc(GrB_Semiring s, bool swap_mult_args) {
  if (!check_obj(s)) fail(GrB_UNINITIALIZED_OBJECT, "semiring is not initialised");
  check_binop(s->mul, "multiply"); check_binop(s->add->op, "monoid");
  if (s->mul->xtype != s->mul->ztype) not_implemented(std::string("semiring with a comparison multiplier: ") + s->name);
  if (!semiring_op_supported(s->mul->opcode, s->mul->ztype->code)) not_implemented(std::string("semiring multiplier ") + s->mul->name + " (math-library and bit-index operators run only in apply / eWise)");
  if (!semiring_op_supported(s->add->op->opcode, s->add->op->ztype->code)) not_implemented(std::string("semiring monoid ") + s->add->op->name);
  SemiringDesc d{};
  d.zcode = s->add->op->ztype->code; d.addop = s->add->op->opcode; d.mulop = s->mul->opcode; d.flip = false;
  if (swap_mult_args) { int m; if (mirror_binop(d.mulop, &m)) d.mulop = m; else d.flip = true; }
  memcpy(d.identity, s->add->identity, 16); memcpy(d.terminal, s->add->terminal, 16); d.has_terminal = s->add->has_terminal;
  fp_minmax_identity(d.zcode, d.addop, d.identity);
  return d;
}

// An index list argument (I, ni) of extract / assign as explicit 64-bit indices, validated against `dim` BEFORE any
// narrowing: GrB_ALL, an explicit list, or SuiteSparse's GxB_RANGE / GxB_STRIDE / GxB_BACKWARDS encodings of ni
// (I = {begin, end[, stride]}, `end` inclusive).
constexpr uint64_t GXB_RANGE = 0x7FFFFFFFFFFFFFFFull, GXB_STRIDE = GXB_RANGE - 1, GXB_BACKWARDS = GXB_RANGE - 2;
inline std::vector<uint64_t> expand_index_list(const GrB_Index* I, GrB_Index ni, uint64_t dim, const char* what) {
  std::vector<uint64_t> out;
  // an explicit list of positions is only ever built for lists that fit in memory: GrB_ALL or an open-ended range over a
  // hypersparse dimension (2^60) is refused with an error instead of an allocation failure
  constexpr uint64_t LIST_MAX = 1ull << 28;
  auto too_many = [&](uint64_t k) { fail(GrB_INSUFFICIENT_SPACE, std::string(what) + ": an index list of " + std::to_string(k) + " positions (more than 2^28) is not materialised; slice hypersparse containers with explicit indices"); };
  if (I == GrB_ALL) { if (dim > LIST_MAX) too_many(dim); out.resize(dim); for (uint64_t i = 0; i < dim; i++) out[i] = i; return out; }
  if (!I) fail(GrB_NULL_POINTER, std::string(what) + ": index list is NULL");
  auto oob = [&]() { fail(GrB_INDEX_OUT_OF_BOUNDS, std::string(what) + ": index out of bounds"); };
  if (ni == GXB_RANGE || ni == GXB_STRIDE || ni == GXB_BACKWARDS) {
    const uint64_t b = I[0], e = I[1], st = ni == GXB_RANGE ? 1 : I[2];
    if (st == 0) return out;
    { const uint64_t lo = ni == GXB_BACKWARDS ? e : b, hi = ni == GXB_BACKWARDS ? b : e; if (hi >= lo && (hi - lo) / st >= LIST_MAX) too_many((hi - lo) / st + 1); }
    if (ni == GXB_BACKWARDS) { if (b >= dim && b >= e) oob(); for (uint64_t i = b; i + 1 > e; i -= st) { if (i >= dim) oob(); out.push_back(i); if (i < st) break; } }
    else for (uint64_t i = b; i <= e; i += st) { if (i >= dim) oob(); out.push_back(i); }
    return out;
  }
  if (ni > (1ull << 40)) fail(GrB_INVALID_VALUE, std::string(what) + ": index count is not plausible");
  out.assign(I, I + ni);
  for (uint64_t v : out) if (v >= dim) oob();
  return out;
}

// how many positions an index argument names (GrB_ALL, an explicit list, or a GxB_RANGE / GxB_STRIDE / GxB_BACKWARDS triple whose
// `ni` is a sentinel, not a count), without building the list
inline double index_count(const GrB_Index* I, GrB_Index ni, uint64_t dim) {
  if (I == GrB_ALL) return (double)dim;
  if (!I) return 0.0;
  if (ni == GXB_RANGE || ni == GXB_STRIDE || ni == GXB_BACKWARDS) {
    const uint64_t b = I[0], e = I[1], st = ni == GXB_RANGE ? 1 : I[2];
    if (st == 0) return 0.0;
    const uint64_t lo = ni == GXB_BACKWARDS ? e : b, hi0 = ni == GXB_BACKWARDS ? b : e;
    if (hi0 < lo) return 0.0;
    const uint64_t hi = hi0 >= dim && dim ? dim - 1 : hi0;          // (an end beyond the dimension is an error raised later; count what could exist)
    return hi < lo ? 0.0 : (double)((hi - lo) / st + 1);
  }
  return (double)ni;
}

// values of a device array in another type: returns `src` itself when no cast is needed, else fills `tmp`
inline const void* cast_values(int dst_code, int src_code, const void* src, uint64_t n, DevBuf& tmp) {
  if (dst_code == src_code || n == 0) return src;
  tmp.alloc(n * type_size(dst_code));
  vec_cast_values(dst_code, tmp.p, src_code, src, n);
  return tmp.p;
}

// mask vector -> allow bytes.  Returns nullptr (= everything allowed) when there is no mask and no
// complement; sets *nothing when there is no mask but the complement flag is set.
inline const uint8_t* vector_allow(GrB_Vector mask, const DescView& dv, uint64_t n, DevBuf& tmp, bool* nothing) {
  *nothing = false;
  if (!mask) { if (dv.mask_comp) *nothing = true; return nullptr; }
  vec_to_device(mask);
  tmp.alloc(n ? n : 1);
  build_allow(n, mask->type->code, mask->dval.p, mask->dpres.as<uint8_t>(), dv.mask_struct, dv.mask_comp, tmp.as<uint8_t>());
  return tmp.as<uint8_t>();
}

// Write T (bitmap tval/tpres of type tcode; buffers are consumed) into w under mask/accum/replace.
// `t_only_allowed`: T has no entries where the mask forbids writing (true for the kernels here).
// `t_nvals`: the entry count of T when the caller knows it (~0 otherwise) — carried into w where the result's count follows
// from it, so that the next product does not have to count on the device and wait for the answer.
inline void vector_write_back(GrB_Vector w, int tcode, DevBuf& tval, DevBuf& tpres, const uint8_t* allow, GrB_BinaryOp accum,
                              bool replace, bool t_only_allowed, uint64_t t_nvals = ~0ull) {
  const uint64_t n = w->n; const int wcode = w->type->code;
  w->fe_lb = 0; w->fe_lb_key = 0;                       // entries may disappear: the bound of grb_mxv.cpp's direction choice is void
  const bool w_empty = w->lazy ? false : (w->host_valid ? (vec_nvals(w) == 0) : (w->dnvals_known && w->dnvals == 0));   // (deferred work pending on w: not known, and not worth completing for this)
  if (accum) check_binop(accum, "accum");
  if (!accum && (!allow || (t_only_allowed && (replace || w_empty)))) {
    // w becomes exactly T: adopt the buffers (typecast the values if the output type differs)
    if (tcode != wcode) { DevBuf c(n * w->type->size + 1); vec_cast_values(wcode, c.p, tcode, tval.p, n); tval = std::move(c); }
    vec_invalidate_host(w);
    w->dval = std::move(tval); w->dpres = std::move(tpres); w->dev_valid = true;
    w->dnvals_known = t_nvals != ~0ull; w->dnvals = w->dnvals_known ? t_nvals : 0;
    return;
  }
  vec_to_device(w);
  // no mask + accum: the result is the union of w and T — still full when w or T was
  const bool stays_full = !allow && accum && ((w->dnvals_known && w->dnvals == n) || t_nvals == n);
  // the epilogue runs in the accumulator's domain (or w's type without one)
  const int ecode = accum ? accum->xtype->code : wcode;
  DevBuf tc, wc;
  const void* tv = cast_values(ecode, tcode, tval.p, n, tc);
  void* wv = w->dval.p;
  if (ecode != wcode) { wc.alloc(n * type_size(ecode) + 1); vec_cast_values(ecode, wc.p, wcode, w->dval.p, n); wv = wc.p; }
  vec_epilogue(ecode, n, wv, w->dpres.as<uint8_t>(), tv, tpres.as<uint8_t>(), allow, accum ? accum->opcode : -1, replace);
  if (ecode != wcode) vec_cast_values(wcode, w->dval.p, ecode, wv, n);
  vec_invalidate_host(w);
  w->dnvals_known = stays_full; w->dnvals = stays_full ? n : 0;   // (temporaries return to the pool; reuse is stream-ordered)
}

}  // namespace grb
