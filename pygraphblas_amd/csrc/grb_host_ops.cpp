// grb_host_ops.cpp — index-list extract / assign and kronecker, on the host mirror.
//
//   GrB_Vector_extract, GrB_Col_extract, GrB_Matrix_extract     <- Vector / Matrix slicing  (pygraphblas/vector.py:1526-1573, matrix.py:2807-2990)
//   GrB_Vector_assign, GrB_Row_assign, GrB_Col_assign, GrB_Matrix_assign <- slice assignment (vector.py:1447-1492, matrix.py:2992-3130)
//   GrB_Matrix_kronecker_BinaryOp                               <- Matrix.kronecker         (matrix.py:2728-2805)
//   GxB_{Matrix,Vector}_apply_BinaryOp1st/2nd                   <- apply_first / apply_second with a Scalar (matrix.py:1999-2004, 2034-2039)
//
// None of these is on the hot path (mxm / mxv / vxm and the O(n) / O(nnz) operations of the BFS, PageRank and
// triangle-count loops, all HIP): they are the element-wise container surface — notebook slicing `M[2]`, `v[1:3]`,
// `M[1, :] = v` — beside setElement / extractElement, and like those they edit the host mirror of the container (sorted
// tuples); the HBM image is rebuilt lazily the next time a kernel needs it.  Semantics: C API 1.3: T is
// formed, then C<M,replace> = accum(C, T); for assign the mask spans all of C (GrB_assign), for row / column assign only that
// row / column.
#include "grb_opcommon.hpp"
#include <array>
#include <map>

// the typed apply entry points (grb_matrix_ops.cpp / grb_vector_ops.cpp) that the GxB_Scalar forms forward to
#define GRB_FOR_TYPES(X) X(BOOL, bool, T_BOOL) X(INT8, int8_t, T_INT8) X(UINT8, uint8_t, T_UINT8) X(INT16, int16_t, T_INT16) X(UINT16, uint16_t, T_UINT16) X(INT32, int32_t, T_INT32) \
  X(UINT32, uint32_t, T_UINT32) X(INT64, int64_t, T_INT64) X(UINT64, uint64_t, T_UINT64) X(FP32, float, T_FP32) X(FP64, double, T_FP64)
extern "C" {
#define GRB_DECL(SUF, CT, CODE) \
  GrB_Info GxB_Matrix_apply_BinaryOp1st_##SUF(GrB_Matrix, const GrB_Matrix, const GrB_BinaryOp, const GrB_BinaryOp, CT, const GrB_Matrix, const GrB_Descriptor); \
  GrB_Info GxB_Matrix_apply_BinaryOp2nd_##SUF(GrB_Matrix, const GrB_Matrix, const GrB_BinaryOp, const GrB_BinaryOp, const GrB_Matrix, CT, const GrB_Descriptor); \
  GrB_Info GxB_Vector_apply_BinaryOp1st_##SUF(GrB_Vector, const GrB_Vector, const GrB_BinaryOp, const GrB_BinaryOp, CT, const GrB_Vector, const GrB_Descriptor); \
  GrB_Info GxB_Vector_apply_BinaryOp2nd_##SUF(GrB_Vector, const GrB_Vector, const GrB_BinaryOp, const GrB_BinaryOp, const GrB_Vector, CT, const GrB_Descriptor);
GRB_FOR_TYPES(GRB_DECL)
#undef GRB_DECL
GrB_Info GrB_Matrix_eWiseAdd_BinaryOp(GrB_Matrix, const GrB_Matrix, const GrB_BinaryOp, const GrB_BinaryOp, const GrB_Matrix, const GrB_Matrix, const GrB_Descriptor);
GrB_Info GrB_Vector_eWiseAdd_BinaryOp(GrB_Vector, const GrB_Vector, const GrB_BinaryOp, const GrB_BinaryOp, const GrB_Vector, const GrB_Vector, const GrB_Descriptor);
}

using namespace grb;

namespace {

typedef std::array<uint8_t, 16> Val;
typedef std::map<std::pair<uint64_t, uint64_t>, Val> Map;     // (row, column) -> value bytes in some type

Map load(GrB_Matrix A, bool transpose) {
  mat_to_host(A); Map m; const size_t ts = A->type->size;
  for (size_t p = 0; p < A->hi.size(); p++) { Val v{}; memcpy(v.data(), &A->hx[p * ts], ts); m[transpose ? std::make_pair(A->hj[p], A->hi[p]) : std::make_pair(A->hi[p], A->hj[p])] = v; }
  return m;
}
Map load(GrB_Vector u) {
  vec_to_host(u); Map m; const size_t ts = u->type->size;
  for (size_t p = 0; p < u->hi.size(); p++) { Val v{}; memcpy(v.data(), &u->hx[p * ts], ts); m[{u->hi[p], 0}] = v; }
  return m;
}
void store(GrB_Matrix C, const Map& m) {
  const size_t ts = C->type->size;
  C->hi.clear(); C->hj.clear(); C->hx.clear(); C->pending.clear();
  for (auto& kv : m) { C->hi.push_back(kv.first.first); C->hj.push_back(kv.first.second); C->hx.insert(C->hx.end(), kv.second.data(), kv.second.data() + ts); }
  C->host_valid = true; mat_invalidate_device(C);
}
void store(GrB_Vector w, const Map& m) {
  const size_t ts = w->type->size;
  w->hi.clear(); w->hx.clear(); w->pending.clear();
  for (auto& kv : m) { w->hi.push_back(kv.first.first); w->hx.insert(w->hx.end(), kv.second.data(), kv.second.data() + ts); }
  w->host_valid = true; vec_invalidate_device(w);
}
Val cast(int dst, int src, const Val& v) { Val o{}; cast_scalar(dst, o.data(), src, v.data()); return o; }
bool truth(int code, const Val& v) { Val b = cast(T_BOOL, code, v); return b[0] != 0; }
Val combine(GrB_BinaryOp op, int ccode, const Val& c, int tcode, const Val& t) {          // accum(c, t) in the operator's domain, result in C's type
  const int ac = op->xtype->code; if (ac >= T_FC32) fail(GrB_DOMAIN_MISMATCH, "operators on complex values are out of scope");
  Val a = cast(ac, ccode, c), b = cast(ac, tcode, t), z{};
  dispatch_type(ac, [&]<class T>() { T x, y; memcpy(&x, a.data(), sizeof(T)); memcpy(&y, b.data(), sizeof(T)); T r = apply_binop<T>(op->opcode, x, y); memcpy(z.data(), &r, sizeof(T)); });
  return cast(ccode, op->ztype->code, z);
}
struct MaskView { const Map* m; int code; bool structural, comp, present;
  bool allows(const std::pair<uint64_t, uint64_t>& p) const {
    if (!present) return !comp;
    auto it = m->find(p); const bool t = it != m->end() && (structural || truth(code, it->second));
    return t != comp;
  } };

// C<M,replace> = accum(C, T) on maps.  `in_scope(p)`: positions the mask / replace step may touch (all of C, or one row / column)
template <class Scope> void write_back(Map& C, int ccode, const Map& T, int tcode, const MaskView& mk, bool replace, GrB_BinaryOp accum, Scope in_scope) {
  if (accum) check_binop(accum, "accum");
  Map Z;
  if (accum) {
    Z = C;
    for (auto& kv : T) { auto it = Z.find(kv.first); if (it == Z.end()) Z[kv.first] = cast(ccode, tcode, kv.second); else it->second = combine(accum, ccode, it->second, tcode, kv.second); }
  } else for (auto& kv : T) Z[kv.first] = cast(ccode, tcode, kv.second);
  Map out;
  for (auto& kv : C) { const bool scope = in_scope(kv.first); if (!scope || (!mk.allows(kv.first) && !replace)) out[kv.first] = kv.second; }   // what survives untouched
  for (auto& kv : Z) if (in_scope(kv.first) && mk.allows(kv.first)) out[kv.first] = kv.second;
  if (accum) { /* outside the scope Z == C: already kept */ }
  C.swap(out);
}
auto everywhere = [](const std::pair<uint64_t, uint64_t>&) { return true; };

std::vector<uint64_t> indices(const GrB_Index* I, GrB_Index ni, uint64_t dim, const char* what) { return expand_index_list(I, ni, dim, what); }
// an index list that is never materialised when it is GrB_ALL (the identity map): the default-dimension (2^60) hypersparse
// containers are sliced with it (`H[1]`, `H[:, 2]`), and a vector of `dim` indices cannot exist there
struct Sel {
  bool all = false; uint64_t dim = 0; std::vector<uint64_t> v;
  Sel(const GrB_Index* I, GrB_Index ni, uint64_t d, const char* what) : all(I == GrB_ALL), dim(d) { if (!all) v = expand_index_list(I, ni, d, what); }
  uint64_t size() const { return all ? dim : v.size(); }
  bool increasing() const { if (all) return true; for (size_t k = 1; k < v.size(); k++) if (v[k] <= v[k - 1]) return false; return true; }
  // position of source index x in an increasing list, or ~0
  uint64_t find_sorted(uint64_t x) const { if (all) return x < dim ? x : ~0ull; const auto a = std::lower_bound(v.begin(), v.end(), x); return (a == v.end() || *a != x) ? ~0ull : (uint64_t)(a - v.begin()); }
};
void check_m(GrB_Matrix A, const char* w) { if (!check_obj(A)) fail(GrB_UNINITIALIZED_OBJECT, std::string(w) + ": uninitialised matrix"); }
void check_v(GrB_Vector A, const char* w) { if (!check_obj(A)) fail(GrB_UNINITIALIZED_OBJECT, std::string(w) + ": uninitialised vector"); }

// ---- the common slices on the sorted tuples themselves (no map of the whole matrix): `M[i]`, `M[:, j]`, `M[i] = v`, `M[a:b, c:d]` ----
typedef std::vector<std::pair<uint64_t, Val>> Line;                     // (index, value bytes), sorted by index
Val val_at(const GrB_Matrix A, size_t p) { Val v{}; memcpy(v.data(), &A->hx[p * A->type->size], A->type->size); return v; }
std::pair<size_t, size_t> row_range(const GrB_Matrix A, uint64_t i) {
  const auto lo = std::lower_bound(A->hi.begin(), A->hi.end(), i), hi = std::upper_bound(lo, A->hi.end(), i);
  return {(size_t)(lo - A->hi.begin()), (size_t)(hi - A->hi.begin())};
}
// column j of op(A): a row of A is a contiguous run of its tuples, a column is picked out in one pass
Line line_of(GrB_Matrix A, bool transposed, uint64_t j) {
  mat_to_host(A); Line out;
  if (transposed) { const auto r = row_range(A, j); out.reserve(r.second - r.first); for (size_t p = r.first; p < r.second; p++) out.push_back({A->hj[p], val_at(A, p)}); }
  else for (size_t p = 0; p < A->hj.size(); p++) if (A->hj[p] == j) out.push_back({A->hi[p], val_at(A, p)});
  return out;
}
// row i (or column i) of C := `line` (values already in C's type)
void replace_line(GrB_Matrix C, bool is_row, uint64_t i, const Line& line) {
  const size_t ts = C->type->size;
  if (is_row) {
    const auto r = row_range(C, i);
    std::vector<GrB_Index> nj; std::vector<uint8_t> nx; nj.reserve(line.size()); nx.reserve(line.size() * ts);
    for (auto& e : line) { nj.push_back(e.first); nx.insert(nx.end(), e.second.data(), e.second.data() + ts); }
    C->hi.erase(C->hi.begin() + r.first, C->hi.begin() + r.second); C->hi.insert(C->hi.begin() + r.first, line.size(), i);
    C->hj.erase(C->hj.begin() + r.first, C->hj.begin() + r.second); C->hj.insert(C->hj.begin() + r.first, nj.begin(), nj.end());
    C->hx.erase(C->hx.begin() + r.first * ts, C->hx.begin() + r.second * ts); C->hx.insert(C->hx.begin() + r.first * ts, nx.begin(), nx.end());
  } else {
    const size_t n = C->hi.size();
    std::vector<GrB_Index> ni, nj; std::vector<uint8_t> nx; ni.reserve(n + line.size()); nj.reserve(n + line.size()); nx.reserve((n + line.size()) * ts);
    size_t q = 0;
    auto emit = [&](const std::pair<uint64_t, Val>& e) { ni.push_back(e.first); nj.push_back(i); nx.insert(nx.end(), e.second.data(), e.second.data() + ts); };
    for (size_t p = 0; p < n; p++) {
      while (q < line.size() && (line[q].first < C->hi[p] || (line[q].first == C->hi[p] && i < C->hj[p]))) emit(line[q++]);
      if (C->hj[p] == i) continue;                                      // the column's old entry
      ni.push_back(C->hi[p]); nj.push_back(C->hj[p]); nx.insert(nx.end(), &C->hx[p * ts], &C->hx[p * ts] + ts);
    }
    while (q < line.size()) emit(line[q++]);
    C->hi.swap(ni); C->hj.swap(nj); C->hx.swap(nx);
  }
  C->host_valid = true; mat_invalidate_device(C);
}
// the line C gets from `C(line) = accum(C(line), u)` over ALL positions, no mask: u's entries (cast), united with C's under accum
Line assigned_line(const Line& cur, int ccode, GrB_Vector u, GrB_BinaryOp accum) {
  if (accum) check_binop(accum, "accum");
  vec_to_host(u); const size_t us = u->type->size; const int ucode = u->type->code;
  Line out; out.reserve(cur.size() + u->hi.size());
  size_t a = 0, b = 0;
  auto uval = [&](size_t k) { Val v{}; memcpy(v.data(), &u->hx[k * us], us); return v; };
  while (a < cur.size() || b < u->hi.size()) {
    if (b >= u->hi.size() || (a < cur.size() && cur[a].first < u->hi[b])) { if (accum) out.push_back(cur[a]); a++; }          // C only: kept under accum, deleted without
    else if (a >= cur.size() || u->hi[b] < cur[a].first) { out.push_back({u->hi[b], cast(ccode, ucode, uval(b))}); b++; }
    else { out.push_back({u->hi[b], accum ? combine(accum, ccode, cur[a].second, ucode, uval(b)) : cast(ccode, ucode, uval(b))}); a++; b++; }
  }
  return out;
}
bool strictly_increasing(const std::vector<uint64_t>& v) { for (size_t k = 1; k < v.size(); k++) if (v[k] <= v[k - 1]) return false; return true; }

}  // namespace

extern "C" {

GrB_Info GrB_Vector_extract(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Vector u, const GrB_Index* I, GrB_Index ni, const GrB_Descriptor desc) {
  if (!w || !u) return GrB_NULL_POINTER; if (!check_obj(w)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(w, [&] {
    check_v(u, "extract"); if (mask) check_v(mask, "extract");
    const DescView dv(desc);
    const Sel idx(I, ni, u->n, "extract");
    if (w->n != idx.size() || (mask && mask->n != w->n)) fail(GrB_DIMENSION_MISMATCH, "extract: output size must equal the number of indices");
    Map U = load(u), T, C = load(w), Mm; if (mask) Mm = load(mask);
    if (idx.all) T = U;
    else for (size_t k = 0; k < idx.v.size(); k++) { auto it = U.find({idx.v[k], 0}); if (it != U.end()) T[{k, 0}] = it->second; }
    write_back(C, w->type->code, T, u->type->code, MaskView{&Mm, mask ? mask->type->code : 0, dv.mask_struct, dv.mask_comp, mask != nullptr}, dv.replace, accum, everywhere);
    store(w, C);
  });
}

GrB_Info GrB_Col_extract(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Matrix A, const GrB_Index* I, GrB_Index ni, GrB_Index j, const GrB_Descriptor desc) {
  if (!w || !A) return GrB_NULL_POINTER; if (!check_obj(w)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(w, [&] {
    check_m(A, "extract"); if (mask) check_v(mask, "extract");
    const DescView dv(desc);
    const uint64_t ar = dv.tran0 ? A->ncols : A->nrows, ac = dv.tran0 ? A->nrows : A->ncols;
    if (j >= ac) fail(GrB_INVALID_INDEX, "extract: column index out of range");
    const Sel idx(I, ni, ar, "extract");
    if (w->n != idx.size() || (mask && mask->n != w->n)) fail(GrB_DIMENSION_MISMATCH, "extract: output size must equal the number of indices");
    const Line col = line_of(A, dv.tran0, j);                             // (index in op(A)'s column j, value), sorted
    Map T, C = load(w), Mm; if (mask) Mm = load(mask);
    if (idx.all) for (auto& e : col) T[{e.first, 0}] = e.second;
    else for (size_t k = 0; k < idx.v.size(); k++) {
      auto it = std::lower_bound(col.begin(), col.end(), idx.v[k], [](const std::pair<uint64_t, Val>& e, uint64_t x) { return e.first < x; });
      if (it != col.end() && it->first == idx.v[k]) T[{k, 0}] = it->second;
    }
    write_back(C, w->type->code, T, A->type->code, MaskView{&Mm, mask ? mask->type->code : 0, dv.mask_struct, dv.mask_comp, mask != nullptr}, dv.replace, accum, everywhere);
    store(w, C);
  });
}

GrB_Info GrB_Matrix_extract(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_Matrix A, const GrB_Index* I, GrB_Index ni, const GrB_Index* J, GrB_Index nj,
                            const GrB_Descriptor desc) {
  if (!C || !A) return GrB_NULL_POINTER; if (!check_obj(C)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(C, [&] {
    check_m(A, "extract"); if (Mask) check_m(Mask, "extract");
    const DescView dv(desc);
    const uint64_t ar = dv.tran0 ? A->ncols : A->nrows, ac = dv.tran0 ? A->nrows : A->ncols;
    const Sel ri(I, ni, ar, "extract"), ci(J, nj, ac, "extract");
    if (C->nrows != ri.size() || C->ncols != ci.size() || (Mask && (Mask->nrows != C->nrows || Mask->ncols != C->ncols))) fail(GrB_DIMENSION_MISMATCH, "extract: output shape must be |I| x |J|");
    if (!Mask && !dv.mask_comp && !accum && ri.increasing() && ci.increasing()) {       // a slice `A[a:b, c:d]` into a fresh output: one pass over A's tuples
      mat_to_host(A); const size_t ts = A->type->size; const int acode = A->type->code, ccode = C->type->code;
      struct E { uint64_t i, j; Val v; }; std::vector<E> out;
      for (size_t p = 0; p < A->hi.size(); p++) {
        const uint64_t si = dv.tran0 ? A->hj[p] : A->hi[p], sj = dv.tran0 ? A->hi[p] : A->hj[p];
        const uint64_t a = ri.find_sorted(si); if (a == ~0ull) continue;
        const uint64_t b = ci.find_sorted(sj); if (b == ~0ull) continue;
        Val v{}; memcpy(v.data(), &A->hx[p * ts], ts);
        out.push_back({a, b, cast(ccode, acode, v)});
      }
      if (dv.tran0) std::sort(out.begin(), out.end(), [](const E& x, const E& y) { return x.i != y.i ? x.i < y.i : x.j < y.j; });
      const size_t cs = C->type->size;
      C->hi.clear(); C->hj.clear(); C->hx.clear(); C->pending.clear(); C->iso_full = false;
      C->hi.reserve(out.size()); C->hj.reserve(out.size()); C->hx.reserve(out.size() * cs);
      for (auto& e : out) { C->hi.push_back(e.i); C->hj.push_back(e.j); C->hx.insert(C->hx.end(), e.v.data(), e.v.data() + cs); }
      C->host_valid = true; mat_invalidate_device(C);
      return;
    }
    Map Am = load(A, dv.tran0), T, Cm = load(C, false), Mm; if (Mask) Mm = load(Mask, false);
    std::multimap<uint64_t, uint64_t> rpos, cpos;                       // source index -> output positions (an index may repeat; GrB_ALL is the identity and is not listed)
    if (!ri.all) for (size_t k = 0; k < ri.v.size(); k++) rpos.insert({ri.v[k], k});
    if (!ci.all) for (size_t k = 0; k < ci.v.size(); k++) cpos.insert({ci.v[k], k});
    for (auto& kv : Am) {
      std::vector<uint64_t> ra, ca;
      if (ri.all) ra.push_back(kv.first.first); else { auto rr = rpos.equal_range(kv.first.first); for (auto a = rr.first; a != rr.second; ++a) ra.push_back(a->second); }
      if (ci.all) ca.push_back(kv.first.second); else { auto cc = cpos.equal_range(kv.first.second); for (auto b = cc.first; b != cc.second; ++b) ca.push_back(b->second); }
      for (uint64_t a : ra) for (uint64_t b : ca) T[{a, b}] = kv.second;
    }
    write_back(Cm, C->type->code, T, A->type->code, MaskView{&Mm, Mask ? Mask->type->code : 0, dv.mask_struct, dv.mask_comp, Mask != nullptr}, dv.replace, accum, everywhere);
    store(C, Cm);
  });
}

// ---- assign: C(I,J)<M> = accum(C(I,J), A) --------------------------------------------------------------------------------------
}  // extern "C"
namespace {
// the region update: positions of the index grid take A's entry (or lose theirs), combined with accum when given
void region_update(Map& C, int ccode, const Map& A, int acode, const std::vector<uint64_t>& ri, const std::vector<uint64_t>& ci, GrB_BinaryOp accum, const MaskView& mk, bool replace,
                   bool whole_mask /* GrB_assign: mask and replace span all of C */, bool row_scope, bool col_scope, uint64_t fixed) {
  if (accum) check_binop(accum, "accum");
  Map out;
  auto in_region = [&](const std::pair<uint64_t, uint64_t>&) { return false; };   (void)in_region;
  // new values over the region
  std::map<std::pair<uint64_t, uint64_t>, std::pair<bool, Val>> upd;               // position -> (has value, value in C's type)
  for (size_t a = 0; a < ri.size(); a++) for (size_t b = 0; b < ci.size(); b++) {
    const std::pair<uint64_t, uint64_t> p{ri[a], ci[b]};
    auto ia = A.find({a, b}); auto ic = C.find(p);
    if (accum) {
      if (ia != A.end() && ic != C.end()) upd[p] = {true, combine(accum, ccode, ic->second, acode, ia->second)};
      else if (ia != A.end()) upd[p] = {true, cast(ccode, acode, ia->second)};
      else if (ic != C.end()) upd[p] = {true, ic->second};
      else upd[p] = {false, Val{}};
    } else upd[p] = ia != A.end() ? std::make_pair(true, cast(ccode, acode, ia->second)) : std::make_pair(false, Val{});
  }
  auto in_scope = [&](const std::pair<uint64_t, uint64_t>& p) { return whole_mask || (row_scope && p.first == fixed) || (col_scope && p.second == fixed); };
  auto mkey = [&](const std::pair<uint64_t, uint64_t>& p) { return row_scope ? std::make_pair(p.second, (uint64_t)0) : (col_scope ? std::make_pair(p.first, (uint64_t)0) : p); };
  for (auto& kv : C) {
    const bool touched = upd.count(kv.first) != 0;
    const bool allowed = mk.allows(mkey(kv.first));
    if (touched && allowed) continue;                                    // replaced below
    if (!allowed && replace && in_scope(kv.first)) continue;             // deleted by replace
    out[kv.first] = kv.second;
  }
  for (auto& kv : upd) if (kv.second.first && mk.allows(mkey(kv.first))) out[kv.first] = kv.second.second;
  C.swap(out);
}
}  // namespace
// the whole-container fast paths forward to eWiseAdd without the caller's descriptor: only when it asks for nothing they would
// drop (a complemented mask without a mask object allows no writes at all; an invalid descriptor is the general path's error)
static inline bool plain_desc(GrB_Descriptor d) { return !d || (check_obj(d) && (d->mask & GrB_COMP) == 0); }
extern "C" {

GrB_Info GrB_Vector_assign(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Vector u, const GrB_Index* I, GrB_Index ni, const GrB_Descriptor desc) {
  if (!w || !u) return GrB_NULL_POINTER; if (!check_obj(w)) return GrB_UNINITIALIZED_OBJECT;
  if (I == GrB_ALL && !mask && plain_desc(desc) && accum && check_obj(u) && device_ok() && u->n == w->n) return GrB_Vector_eWiseAdd_BinaryOp(w, nullptr, nullptr, accum, w, u, nullptr);   // w = accum(w, u), in HBM
  return guarded(w, [&] {
    check_v(u, "assign"); if (mask) check_v(mask, "assign");
    const DescView dv(desc);
    const auto idx = indices(I, ni, w->n, "assign");
    if (u->n != idx.size() || (mask && mask->n != w->n)) fail(GrB_DIMENSION_MISMATCH, "assign: the vector's size must equal the number of indices");
    Map C = load(w), U = load(u), Mm; if (mask) Mm = load(mask);
    region_update(C, w->type->code, U, u->type->code, idx, std::vector<uint64_t>{0}, accum, MaskView{&Mm, mask ? mask->type->code : 0, dv.mask_struct, dv.mask_comp, mask != nullptr}, dv.replace,
                  true, false, false, 0);
    store(w, C);
  });
}

GrB_Info GrB_Matrix_assign(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_Matrix A, const GrB_Index* I, GrB_Index ni, const GrB_Index* J, GrB_Index nj,
                           const GrB_Descriptor desc) {
  if (!C || !A) return GrB_NULL_POINTER; if (!check_obj(C)) return GrB_UNINITIALIZED_OBJECT;
  // the whole container, no mask, with an accumulator (`paths.assign_matrix(frontier, accum=PLUS)`, gap/bcmark.py:41): that is
  // C = accum(C, A) on the union of the patterns — the eWiseAdd kernel, in HBM (a dense ns x n `paths` must not visit the host)
  if (I == GrB_ALL && J == GrB_ALL && !Mask && plain_desc(desc) && accum && check_obj(A) && device_ok()) {      // (a complemented mask without a mask object allows nothing: not this path)
    const DescView dv0(desc);
    if (!dv0.tran0 && A->nrows == C->nrows && A->ncols == C->ncols) return GrB_Matrix_eWiseAdd_BinaryOp(C, nullptr, nullptr, accum, C, A, nullptr);
  }
  return guarded(C, [&] {
    check_m(A, "assign"); if (Mask) check_m(Mask, "assign");
    const DescView dv(desc);
    const auto ri = indices(I, ni, C->nrows, "assign"), ci = indices(J, nj, C->ncols, "assign");
    const uint64_t ar = dv.tran0 ? A->ncols : A->nrows, ac = dv.tran0 ? A->nrows : A->ncols;
    if (ar != ri.size() || ac != ci.size() || (Mask && (Mask->nrows != C->nrows || Mask->ncols != C->ncols))) fail(GrB_DIMENSION_MISMATCH, "assign: the matrix must be |I| x |J|");
    if (!Mask && !dv.mask_comp && !dv.tran0 && C != A && strictly_increasing(ri) && strictly_increasing(ci) && C->type->code < T_FC32 && A->type->code < T_FC32) {
      // `M[a:b, c:d] = A` with increasing index lists: A's tuples stay sorted when they move to (I[i], J[j]) — one merge of C's tuples
      // (without those of the region, unless an accumulator keeps them) with A's
      if (accum) check_binop(accum, "accum");
      mat_to_host(C); mat_to_host(A);
      const size_t cs = C->type->size, as = A->type->size; const int ccode = C->type->code, acode = A->type->code;
      auto in_region = [&](uint64_t i, uint64_t j) { return std::binary_search(ri.begin(), ri.end(), i) && std::binary_search(ci.begin(), ci.end(), j); };
      std::vector<GrB_Index> ni2, nj2; std::vector<uint8_t> nx;
      const size_t nc = C->hi.size(), na = A->hi.size();
      ni2.reserve(nc + na); nj2.reserve(nc + na); nx.reserve((nc + na) * cs);
      auto put = [&](uint64_t i, uint64_t j, const Val& v) { ni2.push_back(i); nj2.push_back(j); nx.insert(nx.end(), v.data(), v.data() + cs); };
      size_t p = 0, q = 0;
      while (p < nc || q < na) {
        const uint64_t ai = q < na ? ri[A->hi[q]] : 0, aj = q < na ? ci[A->hj[q]] : 0;
        const bool take_c = q >= na || (p < nc && (C->hi[p] < ai || (C->hi[p] == ai && C->hj[p] < aj)));
        if (take_c) {                                                   // an entry of C with no counterpart in A: kept outside the region, and inside it under an accumulator
          if (accum || !in_region(C->hi[p], C->hj[p])) put(C->hi[p], C->hj[p], val_at(C, p));
          p++;
        } else {
          Val av{}; memcpy(av.data(), &A->hx[q * as], as);
          const bool both = p < nc && C->hi[p] == ai && C->hj[p] == aj;
          put(ai, aj, both && accum ? combine(accum, ccode, val_at(C, p), acode, av) : cast(ccode, acode, av));
          if (both) p++;
          q++;
        }
      }
      C->hi.swap(ni2); C->hj.swap(nj2); C->hx.swap(nx); C->host_valid = true; mat_invalidate_device(C);
      return;
    }
    Map Cm = load(C, false), Am = load(A, dv.tran0), Mm; if (Mask) Mm = load(Mask, false);
    region_update(Cm, C->type->code, Am, A->type->code, ri, ci, accum, MaskView{&Mm, Mask ? Mask->type->code : 0, dv.mask_struct, dv.mask_comp, Mask != nullptr}, dv.replace, true, false, false, 0);
    store(C, Cm);
  });
}

GrB_Info GrB_Row_assign(GrB_Matrix C, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Vector u, GrB_Index i, const GrB_Index* J, GrB_Index nj, const GrB_Descriptor desc) {
  if (!C || !u) return GrB_NULL_POINTER; if (!check_obj(C)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(C, [&] {
    check_v(u, "assign"); if (mask) check_v(mask, "assign");
    const DescView dv(desc);
    if (i >= C->nrows) fail(GrB_INVALID_INDEX, "assign: row index out of range");
    if (!mask && !dv.mask_comp && J == GrB_ALL && u->n == C->ncols && C->type->code < T_FC32 && u->type->code < T_FC32) {   // `M[i] = v`: the row's run of tuples is replaced in place (GrB_ALL is never listed: C may be 2^60 wide)
      mat_to_host(C);
      const auto r = row_range(C, i); Line cur; cur.reserve(r.second - r.first);
      for (size_t p = r.first; p < r.second; p++) cur.push_back({C->hj[p], val_at(C, p)});
      replace_line(C, true, i, assigned_line(cur, C->type->code, u, accum));
      return;
    }
    if (J == GrB_ALL && u->n != C->ncols) fail(GrB_DIMENSION_MISMATCH, "assign: the vector's size must equal the number of column indices");   // (before the list is asked for: GrB_ALL over 2^60 columns is never materialised)
    const auto ci = indices(J, nj, C->ncols, "assign");
    if (u->n != ci.size() || (mask && mask->n != C->ncols)) fail(GrB_DIMENSION_MISMATCH, "assign: the vector's size must equal the number of column indices");
    Map Cm = load(C, false), U = load(u), Ut, Mm; if (mask) Mm = load(mask);
    for (auto& kv : U) Ut[{0, kv.first.first}] = kv.second;             // the vector as a 1 x nj row
    region_update(Cm, C->type->code, Ut, u->type->code, std::vector<uint64_t>{i}, ci, accum, MaskView{&Mm, mask ? mask->type->code : 0, dv.mask_struct, dv.mask_comp, mask != nullptr}, dv.replace,
                  false, true, false, i);
    store(C, Cm);
  });
}

GrB_Info GrB_Col_assign(GrB_Matrix C, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Vector u, const GrB_Index* I, GrB_Index ni, GrB_Index j, const GrB_Descriptor desc) {
  if (!C || !u) return GrB_NULL_POINTER; if (!check_obj(C)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(C, [&] {
    check_v(u, "assign"); if (mask) check_v(mask, "assign");
    const DescView dv(desc);
    if (j >= C->ncols) fail(GrB_INVALID_INDEX, "assign: column index out of range");
    if (!mask && !dv.mask_comp && I == GrB_ALL && u->n == C->nrows && C->type->code < T_FC32 && u->type->code < T_FC32) {   // `M[:, j] = v`: one merge pass over the tuples (GrB_ALL is never listed)
      replace_line(C, false, j, assigned_line(line_of(C, false, j), C->type->code, u, accum));
      return;
    }
    if (I == GrB_ALL && u->n != C->nrows) fail(GrB_DIMENSION_MISMATCH, "assign: the vector's size must equal the number of row indices");   // (as in GrB_Row_assign: before the list is asked for)
    const auto ri = indices(I, ni, C->nrows, "assign");
    if (u->n != ri.size() || (mask && mask->n != C->nrows)) fail(GrB_DIMENSION_MISMATCH, "assign: the vector's size must equal the number of row indices");
    Map Cm = load(C, false), U = load(u), Mm; if (mask) Mm = load(mask);
    region_update(Cm, C->type->code, U, u->type->code, ri, std::vector<uint64_t>{j}, accum, MaskView{&Mm, mask ? mask->type->code : 0, dv.mask_struct, dv.mask_comp, mask != nullptr}, dv.replace,
                  false, false, true, j);
    store(C, Cm);
  });
}

GrB_Info GrB_Matrix_kronecker_BinaryOp(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc) {
  if (!C || !A || !B || !op) return GrB_NULL_POINTER; if (!check_obj(C)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(C, [&] {
    check_m(A, "kronecker"); check_m(B, "kronecker"); if (Mask) check_m(Mask, "kronecker"); check_binop(op, "kronecker");
    const DescView dv(desc);
    const uint64_t ar = dv.tran0 ? A->ncols : A->nrows, ac = dv.tran0 ? A->nrows : A->ncols, br = dv.tran1 ? B->ncols : B->nrows, bc = dv.tran1 ? B->nrows : B->ncols;
    if (C->nrows != ar * br || C->ncols != ac * bc || (Mask && (Mask->nrows != C->nrows || Mask->ncols != C->ncols))) fail(GrB_DIMENSION_MISMATCH, "kronecker: dimensions do not conform");
    Map Am = load(A, dv.tran0), Bm = load(B, dv.tran1), T, Cm = load(C, false), Mm; if (Mask) Mm = load(Mask, false);
    const int oc = op->xtype->code, zc = op->ztype->code;
    for (auto& a : Am) for (auto& b : Bm) {
      Val x = cast(oc, A->type->code, a.second), y = cast(oc, B->type->code, b.second), z{};
      dispatch_type(oc, [&]<class T>() { T p, q; memcpy(&p, x.data(), sizeof(T)); memcpy(&q, y.data(), sizeof(T)); T r = apply_binop<T>(op->opcode, p, q); memcpy(z.data(), &r, sizeof(T)); });
      T[{a.first.first * br + b.first.first, a.first.second * bc + b.first.second}] = z;
    }
    (void)zc;
    write_back(Cm, C->type->code, T, oc, MaskView{&Mm, Mask ? Mask->type->code : 0, dv.mask_struct, dv.mask_comp, Mask != nullptr}, dv.replace, accum, everywhere);
    store(C, Cm);
  });
}


// apply with the bound operand in a GxB_Scalar: forwarded to the typed entry point of the scalar's type
#define GRB_CASE1(SUF, CT, CODE) case CODE: { CT v; memcpy(&v, x->x, sizeof(CT)); return FN1(SUF)(C, M, accum, op, v, A, desc); }
#define GRB_CASE2(SUF, CT, CODE) case CODE: { CT v; memcpy(&v, x->x, sizeof(CT)); return FN2(SUF)(C, M, accum, op, A, v, desc); }
#define GRB_SCALAR_OK(x) if (!x) return GrB_NULL_POINTER; if (!check_obj(x)) return GrB_UNINITIALIZED_OBJECT; if (!x->has) return GrB_INVALID_VALUE;
GrB_Info GxB_Matrix_apply_BinaryOp1st(GrB_Matrix C, const GrB_Matrix M, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GxB_Scalar x, const GrB_Matrix A, const GrB_Descriptor desc) {
  GRB_SCALAR_OK(x)
#define FN1(SUF) GxB_Matrix_apply_BinaryOp1st_##SUF
  switch (x->type->code) { GRB_FOR_TYPES(GRB_CASE1) default: return GrB_DOMAIN_MISMATCH; }
#undef FN1
}
GrB_Info GxB_Matrix_apply_BinaryOp2nd(GrB_Matrix C, const GrB_Matrix M, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Matrix A, const GxB_Scalar x, const GrB_Descriptor desc) {
  GRB_SCALAR_OK(x)
#define FN2(SUF) GxB_Matrix_apply_BinaryOp2nd_##SUF
  switch (x->type->code) { GRB_FOR_TYPES(GRB_CASE2) default: return GrB_DOMAIN_MISMATCH; }
#undef FN2
}
GrB_Info GxB_Vector_apply_BinaryOp1st(GrB_Vector C, const GrB_Vector M, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GxB_Scalar x, const GrB_Vector A, const GrB_Descriptor desc) {
  GRB_SCALAR_OK(x)
#define FN1(SUF) GxB_Vector_apply_BinaryOp1st_##SUF
  switch (x->type->code) { GRB_FOR_TYPES(GRB_CASE1) default: return GrB_DOMAIN_MISMATCH; }
#undef FN1
}
GrB_Info GxB_Vector_apply_BinaryOp2nd(GrB_Vector C, const GrB_Vector M, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Vector A, const GxB_Scalar x, const GrB_Descriptor desc) {
  GRB_SCALAR_OK(x)
#define FN2(SUF) GxB_Vector_apply_BinaryOp2nd_##SUF
  switch (x->type->code) { GRB_FOR_TYPES(GRB_CASE2) default: return GrB_DOMAIN_MISMATCH; }
#undef FN2
}

double GxB_ALWAYS_HYPER = 1.0, GxB_NEVER_HYPER = -1.0, GxB_HYPER_DEFAULT = 0.0625;      // hyper_switch settings (stored options only)


// ---- a scalar into a region, on the host mirror ------------------------------------------------------------------
// Used for containers that have no HBM layout: complex ones (`Matrix.dense(FC64, 10, 10)` is `M[:, :] = 0j`,
// pygraphblas/matrix.py:225-230; tests/test_matrix.py:853) and those whose dimensions exceed the 32-bit device layout
// (the hypersparse default, GxB_INDEX_MAX: `Matrix.sparse(float, fill=3.14, mask=mask)`, pygraphblas/matrix.py:154-162,
// `Matrix.iso(3)`, :234-266).  Container bookkeeping, like setElement — never on the mxm / mxv / vxm path.
typedef struct { float re, im; } GxB_FC32_t;
typedef struct { double re, im; } GxB_FC64_t;
}  // extern "C"
namespace grb {
void host_assign_scalar(GrB_Matrix C, GrB_Matrix Mask, GrB_BinaryOp accum, const void* x, int xcode, const GrB_Index* I, GrB_Index ni, const GrB_Index* J, GrB_Index nj, GrB_Descriptor desc) {
  if (Mask) check_m(Mask, "assign");
  const DescView dv(desc);
  if (Mask && (Mask->nrows != C->nrows || Mask->ncols != C->ncols)) fail(GrB_DIMENSION_MISMATCH, "assign: mask dimensions");
  const bool all = (I == GrB_ALL && J == GrB_ALL);
  const int ccode = C->type->code;
  Val v{}; memcpy(v.data(), x, (size_t)type_size(xcode));
  const double positions = index_count(I, ni, C->nrows) * index_count(J, nj, C->ncols);      // (ni / nj may be a range sentinel, not a count)
  if (all && Mask && !dv.mask_comp) {                    // the result's pattern is bounded by the mask's
    Map Cm = load(C, false), Mm = load(Mask, false), T;
    const MaskView mk{&Mm, Mask->type->code, dv.mask_struct, false, true};
    for (auto& kv : Mm) if (mk.allows(kv.first)) T[kv.first] = v;
    write_back(Cm, ccode, T, xcode, mk, dv.replace, accum, everywhere);
    store(C, Cm); return;
  }
  if (all && !Mask && positions > 16777216.0) {          // every position of a container too large to enumerate: one stored value
    if (accum && mat_nvals(C) != 0) fail(GrB_INSUFFICIENT_SPACE, "assign: accumulating a scalar into every position of a container of this dimension");
    GrB_Matrix_clear(C); C->iso_full = true; memset(C->iso_val, 0, 16); cast_scalar(ccode, C->iso_val, xcode, x); return;
  }
  if (positions > 16777216.0) fail(GrB_INSUFFICIENT_SPACE, "assign: a scalar over more than 2^24 positions of a host-side container");
  const auto ri = indices(I, ni, C->nrows, "assign"), ci = indices(J, nj, C->ncols, "assign");
  Map Cm = load(C, false), Am, Mm; if (Mask) Mm = load(Mask, false);
  for (size_t a = 0; a < ri.size(); a++) for (size_t b = 0; b < ci.size(); b++) Am[{a, b}] = v;
  region_update(Cm, ccode, Am, xcode, ri, ci, accum, MaskView{&Mm, Mask ? Mask->type->code : 0, dv.mask_struct, dv.mask_comp, Mask != nullptr}, dv.replace, true, false, false, 0);
  store(C, Cm);
}
void host_assign_scalar(GrB_Vector w, GrB_Vector mask, GrB_BinaryOp accum, const void* x, int xcode, const GrB_Index* I, GrB_Index ni, GrB_Descriptor desc) {
  if (mask) check_v(mask, "assign");
  const DescView dv(desc);
  if (mask && mask->n != w->n) fail(GrB_DIMENSION_MISMATCH, "assign: mask size");
  const bool all = (I == GrB_ALL);
  const int wcode = w->type->code;
  Val v{}; memcpy(v.data(), x, (size_t)type_size(xcode));
  const double positions = index_count(I, ni, w->n);
  if (all && mask && !dv.mask_comp) {
    Map C = load(w), Mm = load(mask), T;
    const MaskView mk{&Mm, mask->type->code, dv.mask_struct, false, true};
    for (auto& kv : Mm) if (mk.allows(kv.first)) T[kv.first] = v;
    write_back(C, wcode, T, xcode, mk, dv.replace, accum, everywhere);
    store(w, C); return;
  }
  if (all && !mask && positions > 16777216.0) {
    if (accum && vec_nvals(w) != 0) fail(GrB_INSUFFICIENT_SPACE, "assign: accumulating a scalar into every position of a container of this dimension");
    GrB_Vector_clear(w); w->iso_full = true; memset(w->iso_val, 0, 16); cast_scalar(wcode, w->iso_val, xcode, x); return;
  }
  if (positions > 16777216.0) fail(GrB_INSUFFICIENT_SPACE, "assign: a scalar over more than 2^24 positions of a host-side container");
  const auto idx = indices(I, ni, w->n, "assign");
  Map C = load(w), U, Mm; if (mask) Mm = load(mask);
  for (size_t a = 0; a < idx.size(); a++) U[{a, 0}] = v;
  region_update(C, wcode, U, xcode, idx, std::vector<uint64_t>{0}, accum, MaskView{&Mm, mask ? mask->type->code : 0, dv.mask_struct, dv.mask_comp, mask != nullptr}, dv.replace, true, false, false, 0);
  store(w, C);
}
}  // namespace grb
extern "C" {
static GrB_Info mat_assign_complex(GrB_Matrix C, GrB_Matrix Mask, GrB_BinaryOp accum, const void* x, int xcode, const GrB_Index* I, GrB_Index ni, const GrB_Index* J, GrB_Index nj, GrB_Descriptor desc) {
  if (!C) return GrB_NULL_POINTER; if (!check_obj(C)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(C, [&] { host_assign_scalar(C, Mask, accum, x, xcode, I, ni, J, nj, desc); });
}
static GrB_Info vec_assign_complex(GrB_Vector w, GrB_Vector mask, GrB_BinaryOp accum, const void* x, int xcode, const GrB_Index* I, GrB_Index ni, GrB_Descriptor desc) {
  if (!w) return GrB_NULL_POINTER; if (!check_obj(w)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(w, [&] { host_assign_scalar(w, mask, accum, x, xcode, I, ni, desc); });
}
GrB_Info GxB_Matrix_assign_FC32(GrB_Matrix C, const GrB_Matrix M, const GrB_BinaryOp accum, GxB_FC32_t x, const GrB_Index* I, GrB_Index ni, const GrB_Index* J, GrB_Index nj, const GrB_Descriptor desc) { return mat_assign_complex(C, M, accum, &x, T_FC32, I, ni, J, nj, desc); }
GrB_Info GxB_Matrix_assign_FC64(GrB_Matrix C, const GrB_Matrix M, const GrB_BinaryOp accum, GxB_FC64_t x, const GrB_Index* I, GrB_Index ni, const GrB_Index* J, GrB_Index nj, const GrB_Descriptor desc) { return mat_assign_complex(C, M, accum, &x, T_FC64, I, ni, J, nj, desc); }
GrB_Info GxB_Vector_assign_FC32(GrB_Vector w, const GrB_Vector m, const GrB_BinaryOp accum, GxB_FC32_t x, const GrB_Index* I, GrB_Index ni, const GrB_Descriptor desc) { return vec_assign_complex(w, m, accum, &x, T_FC32, I, ni, desc); }
GrB_Info GxB_Vector_assign_FC64(GrB_Vector w, const GrB_Vector m, const GrB_BinaryOp accum, GxB_FC64_t x, const GrB_Index* I, GrB_Index ni, const GrB_Descriptor desc) { return vec_assign_complex(w, m, accum, &x, T_FC64, I, ni, desc); }

// ---- GxB_Matrix_diag / GxB_Vector_diag (Matrix.from_diag, Matrix.vector_diag: pygraphblas/matrix.py:333-375, 2225-2277) ----
GrB_Info GxB_Matrix_diag(GrB_Matrix C, const GrB_Vector v, int64_t k, const GrB_Descriptor desc) {
  (void)desc; if (!C || !v) return GrB_NULL_POINTER; if (!check_obj(C)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(C, [&] {
    check_v(v, "diag");
    const uint64_t ak = (uint64_t)(k < 0 ? -k : k), n = v->n + ak;
    if (C->nrows != n || C->ncols != n) fail(GrB_DIMENSION_MISMATCH, "diag: C must be square of dimension size(v) + |k|");
    Map V = load(v), out;
    for (auto& kv : V) { const uint64_t i = kv.first.first; out[k >= 0 ? std::make_pair(i, i + ak) : std::make_pair(i + ak, i)] = cast(C->type->code, v->type->code, kv.second); }
    store(C, out);
  });
}
GrB_Info GxB_Vector_diag(GrB_Vector v, const GrB_Matrix A, int64_t k, const GrB_Descriptor desc) {
  (void)desc; if (!v || !A) return GrB_NULL_POINTER; if (!check_obj(v)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(v, [&] {
    check_m(A, "diag");
    const uint64_t m = A->nrows, n = A->ncols; uint64_t len = 0;
    if (k >= 0 && (uint64_t)k < n) len = std::min(m, n - (uint64_t)k);
    else if (k < 0 && (uint64_t)(-k) < m) len = std::min(m - (uint64_t)(-k), n);
    if (v->n != len) fail(GrB_DIMENSION_MISMATCH, "diag: the vector must have the length of the k-th diagonal");
    Map Am = load(A, false), out;
    for (auto& kv : Am) {
      const uint64_t i = kv.first.first, j = kv.first.second;
      if (k >= 0 ? (j >= i && j - i == (uint64_t)k) : (i > j && i - j == (uint64_t)(-k))) out[{k >= 0 ? i : j, 0}] = cast(v->type->code, A->type->code, kv.second);
    }
    store(v, out);
  });
}
}  // extern "C"
