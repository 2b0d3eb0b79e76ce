/* grb_oracle.c — CPU restatement of the GraphBLAS mxm / mxv / vxm semantics.   TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle of the MI355X backend.  It may be loaded only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product path
 * (pygraphblas_amd/ never imports it; the C-ABI library does not link it).
 *
 * What it restates.  The reference (Graphegon/pygraphblas) holds no arithmetic of its own: its
 * Matrix.mxm / Matrix.mxv / Vector.vxm (pygraphblas/matrix.py:2553-2584, :2693-2726,
 * pygraphblas/vector.py:942-971) forward to GrB_mxm / GrB_mxv / GrB_vxm of
 * SuiteSparse:GraphBLAS (third-party, unpinned in setup.py:21, v5.1.x era per
 * build-wheels.sh:13; NOT present in /root/reference nor on this machine).  The algorithm
 * restated here is therefore the published GraphBLAS C API 1.3 definition of those operations
 * (SURVEY.md Appendix A, items 1-7):
 *     T = op(A) (+).(x) op(B)                    entry exists iff some k has both operands stored
 *     Z = accum ? accum(C, T) on the union : T
 *     C<M, replace> = Z                          valued / structural / complemented mask
 * with typecasting as C does except float->int saturation, NaN->0 and x->BOOL == (x != 0).
 * PINNING: the oracle is checked against every golden vector the reference's own tests and
 * doctests hold for this path (tests/golden/reference_vectors.json, transcribed with file:line
 * citations from tests/test_matrix.py:249-306, tests/test_vector.py:298-315,
 * tests/test_descriptor.py:13-30 and the doctests in matrix.py/vector.py), and against
 * scipy.sparse / networkx as independent second opinions (tests/test_oracle.py).
 * Parity with SuiteSparse at R-MAT scale is unpinned by anything runnable here:
 * for BOOL/INT semirings the result is mathematically unique; for FP the comparison is
 * 1e-6 relative.
 *
 * Design: deliberately naive and different from the GPU code — one generic value union, one
 * operator switch, a dense-accumulator Gustavson product on 64-bit-index CSR.  The `fast_*`
 * entry points at the end are straight typed loops (OpenMP) used as the timed CPU baseline; the
 * tests check them against the generic path.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { T_BOOL = 0, T_INT8, T_UINT8, T_INT16, T_UINT16, T_INT32, T_UINT32, T_INT64, T_UINT64, T_FP32, T_FP64 };
enum { B_FIRST = 0, B_SECOND, B_PAIR, B_ANY, B_MIN, B_MAX, B_PLUS, B_MINUS, B_RMINUS, B_TIMES, B_DIV, B_RDIV, B_POW,
       B_ISEQ, B_ISNE, B_ISGT, B_ISLT, B_ISGE, B_ISLE, B_LOR, B_LAND, B_LXOR, B_EQ, B_NE, B_GT, B_LT, B_GE, B_LE, B_LXNOR };
enum { F_REPLACE = 1, F_MASK_COMP = 2, F_MASK_STRUCT = 4, F_TRAN_A = 8, F_TRAN_B = 16 };

typedef struct { int64_t i; uint64_t u; double f; } val_t;   /* i: signed+bool, u: unsigned, f: floating */

static int is_signed_t(int t) { return t == T_INT8 || t == T_INT16 || t == T_INT32 || t == T_INT64; }
static int is_unsigned_t(int t) { return t == T_UINT8 || t == T_UINT16 || t == T_UINT32 || t == T_UINT64; }
static int is_float_t(int t) { return t == T_FP32 || t == T_FP64; }
static size_t tsize(int t) {
  switch (t) { case T_BOOL: case T_INT8: case T_UINT8: return 1; case T_INT16: case T_UINT16: return 2;
               case T_INT32: case T_UINT32: case T_FP32: return 4; default: return 8; }
}
static int tbits(int t) { return (int)tsize(t) * 8; }

/* wrap a wide value to the width of type t (two's complement) */
static val_t norm(int t, val_t v) {
  val_t r = {0, 0, 0.0};
  if (t == T_BOOL) { r.i = v.i != 0; }
  else if (is_signed_t(t)) { int b = tbits(t); uint64_t m = b == 64 ? ~0ull : ((1ull << b) - 1); uint64_t x = (uint64_t)v.i & m;
    if (b < 64 && (x >> (b - 1))) { x |= ~m; }
    r.i = (int64_t)x; }
  else if (is_unsigned_t(t)) { int b = tbits(t); uint64_t m = b == 64 ? ~0ull : ((1ull << b) - 1); r.u = v.u & m; }
  else if (t == T_FP32) { r.f = (double)(float)v.f; }
  else r.f = v.f;
  return r;
}
static val_t load(int t, const void* p, uint64_t k) {
  val_t v = {0, 0, 0.0};
  switch (t) {
    case T_BOOL: v.i = ((const uint8_t*)p)[k] != 0; break;
    case T_INT8: v.i = ((const int8_t*)p)[k]; break;   case T_UINT8: v.u = ((const uint8_t*)p)[k]; break;
    case T_INT16: v.i = ((const int16_t*)p)[k]; break; case T_UINT16: v.u = ((const uint16_t*)p)[k]; break;
    case T_INT32: v.i = ((const int32_t*)p)[k]; break; case T_UINT32: v.u = ((const uint32_t*)p)[k]; break;
    case T_INT64: v.i = ((const int64_t*)p)[k]; break; case T_UINT64: v.u = ((const uint64_t*)p)[k]; break;
    case T_FP32: v.f = ((const float*)p)[k]; break;    case T_FP64: v.f = ((const double*)p)[k]; break;
  }
  return v;
}
static void store(int t, void* p, uint64_t k, val_t v) {
  switch (t) {
    case T_BOOL: ((uint8_t*)p)[k] = v.i != 0; break;
    case T_INT8: ((int8_t*)p)[k] = (int8_t)v.i; break;     case T_UINT8: ((uint8_t*)p)[k] = (uint8_t)v.u; break;
    case T_INT16: ((int16_t*)p)[k] = (int16_t)v.i; break;  case T_UINT16: ((uint16_t*)p)[k] = (uint16_t)v.u; break;
    case T_INT32: ((int32_t*)p)[k] = (int32_t)v.i; break;  case T_UINT32: ((uint32_t*)p)[k] = (uint32_t)v.u; break;
    case T_INT64: ((int64_t*)p)[k] = v.i; break;           case T_UINT64: ((uint64_t*)p)[k] = v.u; break;
    case T_FP32: ((float*)p)[k] = (float)v.f; break;       case T_FP64: ((double*)p)[k] = v.f; break;
  }
}

/* GraphBLAS typecast */
static val_t cast(int dt, int st, val_t v) {
  val_t r = {0, 0, 0.0};
  if (dt == st) return v;
  if (dt == T_BOOL) { r.i = is_float_t(st) ? (v.f != 0.0) : (is_unsigned_t(st) ? v.u != 0 : v.i != 0); return r; }
  if (is_float_t(st)) {
    if (is_float_t(dt)) { r.f = v.f; return norm(dt, r); }
    if (isnan(v.f)) return r;                                   /* NaN -> 0 */
    if (is_signed_t(dt)) {                                      /* saturate */
      int b = tbits(dt); double lo = -ldexp(1.0, b - 1), hi = ldexp(1.0, b - 1);
      if (v.f <= lo) r.i = b == 64 ? INT64_MIN : -((int64_t)1 << (b - 1));
      else if (v.f >= hi) r.i = b == 64 ? INT64_MAX : (((int64_t)1 << (b - 1)) - 1);
      else r.i = (int64_t)v.f;
      return r;
    }
    { int b = tbits(dt); double hi = ldexp(1.0, b);
      if (v.f <= 0.0) r.u = 0; else if (v.f >= hi) r.u = b == 64 ? UINT64_MAX : ((1ull << b) - 1); else r.u = (uint64_t)v.f;
      return r; }
  }
  /* integer / bool source */
  if (is_float_t(dt)) { r.f = is_unsigned_t(st) ? (double)v.u : (double)v.i; return norm(dt, r); }
  { uint64_t bitsv = is_unsigned_t(st) ? v.u : (uint64_t)v.i;   /* C integer conversion = reinterpret modulo 2^k */
    r.i = (int64_t)bitsv; r.u = bitsv; return norm(dt, r); }
}

static val_t from_bool(int t, int b) { val_t r = {0, 0, 0.0}; if (is_float_t(t)) r.f = b; else if (is_unsigned_t(t)) r.u = (uint64_t)b; else r.i = b; return r; }
static int truth(int t, val_t v) { return is_float_t(t) ? v.f != 0.0 : (is_unsigned_t(t) ? v.u != 0 : v.i != 0); }
static int cmp(int t, val_t x, val_t y) {   /* -1, 0, 1; unordered (NaN) -> 2 */
  if (is_float_t(t)) { if (isnan(x.f) || isnan(y.f)) return 2; return x.f < y.f ? -1 : (x.f > y.f ? 1 : 0); }
  if (is_unsigned_t(t)) return x.u < y.u ? -1 : (x.u > y.u ? 1 : 0);
  return x.i < y.i ? -1 : (x.i > y.i ? 1 : 0);
}

/* z = op(x, y), all in type t.  BOOL renames arithmetic to logic as SuiteSparse does
 * (PLUS=MAX=LOR, TIMES=MIN=LAND, MINUS=RMINUS=NE=LXOR, DIV=FIRST, RDIV=SECOND, EQ=ISEQ=LXNOR ...). */
static val_t binop(int op, int t, val_t x, val_t y) {
  val_t r = {0, 0, 0.0};
  if (t == T_BOOL) {
    int a = x.i != 0, b = y.i != 0, z = 0;
    switch (op) {
      case B_FIRST: case B_DIV: z = a; break;
      case B_SECOND: case B_RDIV: case B_ANY: z = b; break;
      case B_PAIR: z = 1; break;
      case B_MIN: case B_TIMES: case B_LAND: z = a && b; break;
      case B_MAX: case B_PLUS: case B_LOR: z = a || b; break;
      case B_MINUS: case B_RMINUS: case B_ISNE: case B_NE: case B_LXOR: z = a != b; break;
      case B_ISEQ: case B_EQ: case B_LXNOR: z = a == b; break;
      case B_ISGT: case B_GT: z = a && !b; break;
      case B_ISLT: case B_LT: z = !a && b; break;
      case B_ISGE: case B_GE: case B_POW: z = a || !b; break;
      case B_ISLE: case B_LE: z = !a || b; break;
    }
    r.i = z; return r;
  }
  switch (op) {
    case B_FIRST: return x;
    case B_SECOND: case B_ANY: return y;
    case B_PAIR: return from_bool(t, 1);
    case B_MIN: if (is_float_t(t)) { r.f = fmin(x.f, y.f); return r; } return cmp(t, x, y) <= 0 ? x : y;
    case B_MAX: if (is_float_t(t)) { r.f = fmax(x.f, y.f); return r; } return cmp(t, x, y) >= 0 ? x : y;
    case B_PLUS: r.f = x.f + y.f; r.i = (int64_t)((uint64_t)x.i + (uint64_t)y.i); r.u = x.u + y.u; return norm(t, r);
    case B_MINUS: r.f = x.f - y.f; r.i = (int64_t)((uint64_t)x.i - (uint64_t)y.i); r.u = x.u - y.u; return norm(t, r);
    case B_RMINUS: r.f = y.f - x.f; r.i = (int64_t)((uint64_t)y.i - (uint64_t)x.i); r.u = y.u - x.u; return norm(t, r);
    case B_TIMES: r.f = x.f * y.f; r.i = (int64_t)((uint64_t)x.i * (uint64_t)y.i); r.u = x.u * y.u; return norm(t, r);
    case B_DIV: case B_RDIV: {
      val_t n = op == B_DIV ? x : y, d = op == B_DIV ? y : x;
      if (is_float_t(t)) { r.f = n.f / d.f; return norm(t, r); }
      if (is_unsigned_t(t)) { if (d.u == 0) { r.u = n.u == 0 ? 0 : UINT64_MAX; return norm(t, r); } r.u = n.u / d.u; return r; }
      if (d.i == -1) { r.i = (int64_t)(0 - (uint64_t)n.i); return norm(t, r); }
      if (d.i == 0) { int b = tbits(t); int64_t mx = b == 64 ? INT64_MAX : (((int64_t)1 << (b - 1)) - 1);
        r.i = n.i == 0 ? 0 : (n.i < 0 ? -mx - 1 : mx); return r; }
      r.i = n.i / d.i; return r; }
    case B_ISEQ: case B_EQ: return from_bool(t, cmp(t, x, y) == 0);
    case B_ISNE: case B_NE: return from_bool(t, cmp(t, x, y) != 0);
    case B_ISGT: case B_GT: return from_bool(t, cmp(t, x, y) == 1);
    case B_ISLT: case B_LT: return from_bool(t, cmp(t, x, y) == -1);
    case B_ISGE: case B_GE: { int c = cmp(t, x, y); return from_bool(t, c == 0 || c == 1); }
    case B_ISLE: case B_LE: { int c = cmp(t, x, y); return from_bool(t, c == 0 || c == -1); }
    case B_LOR: return from_bool(t, truth(t, x) || truth(t, y));
    case B_LAND: return from_bool(t, truth(t, x) && truth(t, y));
    case B_LXOR: return from_bool(t, truth(t, x) != truth(t, y));
  }
  return r;
}

/* ---- CSR with 64-bit indices ------------------------------------------------------------------ */
typedef struct { int type; uint64_t nrows, ncols, nvals; uint64_t* rp; uint64_t* col; void* val; } csr_t;

static void csr_free(csr_t* m) { free(m->rp); free(m->col); free(m->val); m->rp = m->col = NULL; m->val = NULL; }

/* tuples (any order, no duplicates) -> CSR with sorted rows; optionally transposed */
static int csr_from_tuples(csr_t* m, int type, uint64_t nrows, uint64_t ncols, uint64_t n, const uint64_t* I, const uint64_t* J,
                           const void* X, int transpose) {
  size_t ts = tsize(type);
  if (transpose) { const uint64_t* t = I; I = J; J = t; uint64_t d = nrows; nrows = ncols; ncols = d; }
  m->type = type; m->nrows = nrows; m->ncols = ncols; m->nvals = n;
  m->rp = (uint64_t*)calloc(nrows + 2, 8); m->col = (uint64_t*)malloc((n + 1) * 8); m->val = malloc((n + 1) * ts);
  if (!m->rp || !m->col || !m->val) return 1;
  for (uint64_t k = 0; k < n; k++) m->rp[I[k] + 1]++;
  for (uint64_t r = 0; r < nrows; r++) m->rp[r + 1] += m->rp[r];
  uint64_t* fill = (uint64_t*)malloc((nrows + 1) * 8); memcpy(fill, m->rp, (nrows + 1) * 8);
  for (uint64_t k = 0; k < n; k++) { uint64_t p = fill[I[k]]++; m->col[p] = J[k]; memcpy((char*)m->val + p * ts, (const char*)X + k * ts, ts); }
  free(fill);
  /* insertion sort inside each row (rows are short in the test sizes; bench inputs arrive sorted) */
  char tmp[8];
  for (uint64_t r = 0; r < nrows; r++)
    for (uint64_t p = m->rp[r] + 1; p < m->rp[r + 1]; p++) {
      uint64_t c = m->col[p]; memcpy(tmp, (char*)m->val + p * ts, ts); uint64_t q = p;
      while (q > m->rp[r] && m->col[q - 1] > c) { m->col[q] = m->col[q - 1]; memcpy((char*)m->val + q * ts, (char*)m->val + (q - 1) * ts, ts); q--; }
      m->col[q] = c; memcpy((char*)m->val + q * ts, tmp, ts);
    }
  return 0;
}

/* ---- the operation ------------------------------------------------------------------------------ */
/* C<M,replace> = accum(C, op(A) add.mul op(B)).  Inputs as tuples; output tuples (row-major sorted)
 * are malloc'ed here and released with oracle_free.  Returns 0, or 8 = dimension mismatch. */
int oracle_mxm(int ctype, uint64_t cnrows, uint64_t cncols, uint64_t cn, const uint64_t* CI, const uint64_t* CJ, const void* CX,
               int mtype, uint64_t mn, const uint64_t* MI, const uint64_t* MJ, const void* MX,
               int accum_op, int accum_type, int add_op, int mul_op, int sr_type,
               int atype, uint64_t anrows, uint64_t ancols, uint64_t an, const uint64_t* AI, const uint64_t* AJ, const void* AX,
               int btype, uint64_t bnrows, uint64_t bncols, uint64_t bn, const uint64_t* BI, const uint64_t* BJ, const void* BX,
               int flags, uint64_t* out_n, uint64_t** OI, uint64_t** OJ, void** OX) {
  csr_t A, B, C, M; memset(&M, 0, sizeof M);
  if (csr_from_tuples(&A, atype, anrows, ancols, an, AI, AJ, AX, flags & F_TRAN_A)) return 10;
  if (csr_from_tuples(&B, btype, bnrows, bncols, bn, BI, BJ, BX, flags & F_TRAN_B)) return 10;
  if (A.ncols != B.nrows || cnrows != A.nrows || cncols != B.ncols) { csr_free(&A); csr_free(&B); return 8; }
  csr_from_tuples(&C, ctype, cnrows, cncols, cn, CI, CJ, CX, 0);
  const int has_mask = mtype >= 0;
  if (has_mask) csr_from_tuples(&M, mtype, cnrows, cncols, mn, MI, MJ, MX, 0);
  const uint64_t nr = cnrows, nc = cncols; const size_t cts = tsize(ctype);
  /* dense accumulators for one output row */
  val_t* tv = (val_t*)malloc((nc + 1) * sizeof(val_t)); uint8_t* tp = (uint8_t*)calloc(nc + 1, 1);
  val_t* cv = (val_t*)malloc((nc + 1) * sizeof(val_t)); uint8_t* cp = (uint8_t*)calloc(nc + 1, 1);
  uint8_t* mp = (uint8_t*)calloc(nc + 1, 1);
  uint64_t cap = cn + 16, on = 0;
  uint64_t* oi = (uint64_t*)malloc(cap * 8); uint64_t* oj = (uint64_t*)malloc(cap * 8); char* ox = (char*)malloc(cap * cts);
  uint64_t* touched = (uint64_t*)malloc((nc + 1) * 8);
  for (uint64_t i = 0; i < nr; i++) {
    /* T(i,:) — SURVEY.md App. A item 2: operands cast to the multiplier's input type (== sr_type here) */
    uint64_t nt = 0;
    for (uint64_t pa = A.rp[i]; pa < A.rp[i + 1]; pa++) {
      uint64_t k = A.col[pa]; val_t a = cast(sr_type, atype, load(atype, A.val, pa));
      for (uint64_t pb = B.rp[k]; pb < B.rp[k + 1]; pb++) {
        uint64_t j = B.col[pb]; val_t b = cast(sr_type, btype, load(btype, B.val, pb));
        val_t m = binop(mul_op, sr_type, a, b);
        if (tp[j]) tv[j] = binop(add_op, sr_type, tv[j], m); else { tp[j] = 1; tv[j] = m; touched[nt++] = j; }
      }
    }
    for (uint64_t p = C.rp[i]; p < C.rp[i + 1]; p++) { cp[C.col[p]] = 1; cv[C.col[p]] = load(ctype, C.val, p); }
    if (has_mask) for (uint64_t p = M.rp[i]; p < M.rp[i + 1]; p++)
      mp[M.col[p]] = (flags & F_MASK_STRUCT) ? 1 : (uint8_t)truth(mtype, load(mtype, M.val, p));
    for (uint64_t j = 0; j < nc; j++) {
      /* item 3: Z = accum(C, T) on the union, or T */
      int zp; val_t z = {0, 0, 0.0};
      if (accum_op >= 0) {
        if (cp[j] && tp[j]) { zp = 1; z = cast(ctype, accum_type, binop(accum_op, accum_type, cast(accum_type, ctype, cv[j]), cast(accum_type, sr_type, tv[j]))); }
        else if (tp[j]) { zp = 1; z = cast(ctype, sr_type, tv[j]); }
        else { zp = cp[j]; z = cv[j]; }
      } else { zp = tp[j]; if (zp) z = cast(ctype, sr_type, tv[j]); }
      /* item 4: mask and replace */
      int m = has_mask ? mp[j] : 1; if (flags & F_MASK_COMP) m = !m;
      int outp; val_t outv = z;
      if (m) outp = zp; else if (flags & F_REPLACE) outp = 0; else { outp = cp[j]; outv = cv[j]; }
      if (outp) {
        if (on == cap) { cap *= 2; oi = (uint64_t*)realloc(oi, cap * 8); oj = (uint64_t*)realloc(oj, cap * 8); ox = (char*)realloc(ox, cap * cts); }
        oi[on] = i; oj[on] = j; store(ctype, ox, on, outv); on++;
      }
    }
    for (uint64_t q = 0; q < nt; q++) tp[touched[q]] = 0;
    for (uint64_t p = C.rp[i]; p < C.rp[i + 1]; p++) cp[C.col[p]] = 0;
    if (has_mask) for (uint64_t p = M.rp[i]; p < M.rp[i + 1]; p++) mp[M.col[p]] = 0;
  }
  free(tv); free(tp); free(cv); free(cp); free(mp); free(touched);
  csr_free(&A); csr_free(&B); csr_free(&C); if (has_mask) csr_free(&M);
  *out_n = on; *OI = oi; *OJ = oj; *OX = ox;
  return 0;
}

void oracle_free(void* p) { free(p); }

/* scalar reduce of values with a monoid (Matrix.reduce_int: pygraphblas/matrix.py:1782-1804) */
void oracle_reduce(int type, uint64_t n, const void* X, int op, void* out) {
  val_t acc = {0, 0, 0.0}; int first = 1;
  for (uint64_t k = 0; k < n; k++) { val_t v = load(type, X, k); if (first) { acc = v; first = 0; } else acc = binop(op, type, acc, v); }
  if (first) {   /* identity */
    if (op == B_MIN) { if (is_float_t(type)) acc.f = INFINITY; else if (is_unsigned_t(type)) acc.u = UINT64_MAX; else acc.i = INT64_MAX; acc = norm(type, acc);
      if (is_signed_t(type)) { int b = tbits(type); acc.i = b == 64 ? INT64_MAX : (((int64_t)1 << (b - 1)) - 1); } }
    else if (op == B_MAX) { if (is_float_t(type)) acc.f = -INFINITY; else if (is_signed_t(type)) { int b = tbits(type); acc.i = b == 64 ? INT64_MIN : -((int64_t)1 << (b - 1)); } }
    else if (op == B_TIMES || op == B_LAND || op == B_LXNOR || op == B_EQ) acc = from_bool(type, 1);
  }
  store(type, out, 0, acc);
}

/* ============================ typed fast paths: the timed CPU baseline ============================ */
/* y = A x over PLUS_TIMES, CSR u32; ypres[i] = row i non-empty.  One OpenMP thread per row chunk. */
/* rows [*r0, *r1) of piece q of P pieces holding about the same number of entries each: R-MAT rows differ by five orders
   of magnitude in length, so equal row counts per thread leave one thread with the hub rows (measured: 82 ms per pass on
   128 threads with 1024-row chunks, 10x slower than this) */
static void spmv_piece(const uint32_t* rp, uint32_t nrows, int q, int P, uint32_t* r0, uint32_t* r1) {
  const uint64_t nnz = rp[nrows], lo_t = nnz * (uint64_t)q / (uint64_t)P, hi_t = nnz * (uint64_t)(q + 1) / (uint64_t)P;
  uint32_t lo = 0, hi = nrows;                       /* first row whose start is >= lo_t */
  while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (rp[mid] < lo_t) lo = mid + 1; else hi = mid; }
  *r0 = q == 0 ? 0 : lo;
  lo = 0; hi = nrows;
  while (lo < hi) { const uint32_t mid = lo + (hi - lo) / 2; if (rp[mid] < hi_t) lo = mid + 1; else hi = mid; }
  *r1 = q == P - 1 ? nrows : lo;
}
void fast_spmv_plus_times_fp64(uint32_t nrows, const uint32_t* rp, const uint32_t* col, const double* val, const double* x,
                               double* y, uint8_t* ypres) {
  const int P = omp_get_max_threads() * 8;
#pragma omp parallel for schedule(dynamic, 1)
  for (int q = 0; q < P; q++) {
    uint32_t r0, r1; spmv_piece(rp, nrows, q, P, &r0, &r1);
    for (uint32_t i = r0; i < r1; i++) {
      double s = 0.0; const uint32_t b = rp[i], e = rp[i + 1];
      for (uint32_t p = b; p < e; p++) s += val[p] * x[col[p]];
      y[i] = s; ypres[i] = e > b;
    }
  }
}
void fast_spmv_plus_times_fp32(uint32_t nrows, const uint32_t* rp, const uint32_t* col, const float* val, const float* x,
                               float* y, uint8_t* ypres) {
  const int P = omp_get_max_threads() * 8;
#pragma omp parallel for schedule(dynamic, 1)
  for (int q = 0; q < P; q++) {
    uint32_t r0, r1; spmv_piece(rp, nrows, q, P, &r0, &r1);
    for (uint32_t i = r0; i < r1; i++) {
      float s = 0.0f; const uint32_t b = rp[i], e = rp[i + 1];
      for (uint32_t p = b; p < e; p++) s += val[p] * x[col[p]];
      y[i] = s; ypres[i] = e > b;
    }
  }
}
/* y = A x over PLUS_SECOND on a pattern (PageRank inner product, gap/prmark.py:22) */
void fast_spmv_plus_second_fp32(uint32_t nrows, const uint32_t* rp, const uint32_t* col, const float* x, float* y, uint8_t* ypres) {
  const int P = omp_get_max_threads() * 8;
#pragma omp parallel for schedule(dynamic, 1)
  for (int q = 0; q < P; q++) {
    uint32_t r0, r1; spmv_piece(rp, nrows, q, P, &r0, &r1);
    for (uint32_t i = r0; i < r1; i++) {
      float s = 0.0f; const uint32_t b = rp[i], e = rp[i + 1];
      for (uint32_t p = b; p < e; p++) s += x[col[p]];
      y[i] = s; ypres[i] = e > b;
    }
  }
}
/* the same product with the row sums formed in double and rounded to float once: the "compensated" form SURVEY.md §8c asks FP
 * comparisons to be made against (the reference's own summation order is unspecified; a sequential float sum over a hub's
 * 1.6e5 terms is itself off by ~1e-5 relative) */
void fast_spmv_plus_second_fp32_wide(uint32_t nrows, const uint32_t* rp, const uint32_t* col, const float* x, float* y, uint8_t* ypres) {
  const int P = omp_get_max_threads() * 8;
#pragma omp parallel for schedule(dynamic, 1)
  for (int q = 0; q < P; q++) {
    uint32_t r0, r1; spmv_piece(rp, nrows, q, P, &r0, &r1);
    for (uint32_t i = r0; i < r1; i++) {
      double s = 0.0; const uint32_t b = rp[i], e = rp[i + 1];
      for (uint32_t p = b; p < e; p++) s += (double)x[col[p]];
      y[i] = (float)s; ypres[i] = e > b;
    }
  }
}
/* the same pattern product in FP64 throughout: the reference iteration of the PageRank parity tests (every rank value of the
 * FP32 loop must lie within 1e-6 of this one, and stop after the same number of iterations) */
void fast_spmv_plus_second_fp64(uint32_t nrows, const uint32_t* rp, const uint32_t* col, const double* x, double* y, uint8_t* ypres) {
  const int P = omp_get_max_threads() * 8;
#pragma omp parallel for schedule(dynamic, 1)
  for (int q = 0; q < P; q++) {
    uint32_t r0, r1; spmv_piece(rp, nrows, q, P, &r0, &r1);
    for (uint32_t i = r0; i < r1; i++) {
      double s = 0.0; const uint32_t b = rp[i], e = rp[i + 1];
      for (uint32_t p = b; p < e; p++) s += x[col[p]];
      y[i] = s; ypres[i] = e > b;
    }
  }
}
/* sum over (i,k) in L, of |L(k,:) ∩ L(i,:)|  ==  reduce(L.mxm(L, PLUS_PAIR, mask=L))  (demo/TriangleCentrality.ipynb cell 17) */
int64_t fast_tricount_LL_maskL(uint32_t n, const uint32_t* rp, const uint32_t* col) {
  int64_t total = 0;
#pragma omp parallel reduction(+ : total)
  {
    uint8_t* mark = (uint8_t*)calloc((size_t)n + 1, 1);
#pragma omp for schedule(dynamic, 256)
    for (int64_t i = 0; i < (int64_t)n; i++) {
      for (uint32_t p = rp[i]; p < rp[i + 1]; p++) mark[col[p]] = 1;
      for (uint32_t p = rp[i]; p < rp[i + 1]; p++) { const uint32_t k = col[p];
        for (uint32_t q = rp[k]; q < rp[k + 1]; q++) total += mark[col[q]]; }
      for (uint32_t p = rp[i]; p < rp[i + 1]; p++) mark[col[p]] = 0;
    }
    free(mark);
  }
  return total;
}
/* level-synchronous BFS written exactly as the reference loop does it
 * (demo/Introduction-to-GraphBLAS-with-Python.ipynb cell 31): levels start at 1, 0 = unreached.
 * q<!v,replace> = v lor.land A  ==  "unvisited j with some visited in-neighbour".  Returns depth. */
int fast_bfs_levels(uint32_t n, const uint32_t* rp, const uint32_t* col, uint32_t src, uint8_t* level) {
  memset(level, 0, n); uint8_t* q = (uint8_t*)calloc(n, 1); uint8_t* nq = (uint8_t*)calloc(n, 1);
  q[src] = 1; int lev = 1; int any = 1;
  while (any && lev <= 255) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) if (q[i]) level[i] = (uint8_t)lev;
    memset(nq, 0, n); any = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(| : any)
    for (int64_t i = 0; i < (int64_t)n; i++) if (q[i])           /* push from the newest frontier: same result set */
      for (uint32_t p = rp[i]; p < rp[i + 1]; p++) { const uint32_t j = col[p]; if (!level[j]) { nq[j] = 1; any = 1; } }
    uint8_t* t = q; q = nq; nq = t; lev++;
  }
  free(q); free(nq); return lev - 1;
}
/* single-source shortest paths written exactly as the reference's loop does it (demo/Intro-Prez.ipynb:1034-1045, the doctest
 * pygraphblas/vector.py:883-885 `with Accum(INT64.min): o @= M`):
 *     v = sparse; v[src] = 0;  repeat { w = dup(v);  v<accum MIN> = v MIN_PLUS A;  } until w.iseq(v)
 * one Jacobi sweep per iteration (every product of a sweep reads the distances of the previous one), so the number of
 * sweeps is the reference's too.  A is CSR u32; the sweep pulls along the transpose (built here by a counting sort), which
 * needs no atomics: t(j) = min_i v(i) + A(i,j), v(j) = min(v(j), t(j)) on the union of the patterns.
 * INT64 sums wrap modulo 2^64 like the C operator; FP64 MIN is fmin (omits NaN).  Returns the number of sweeps done
 * (the last one changed nothing). */
#define FAST_SSSP(NAME, T, ADD, LESS)                                                                                      \
int NAME(uint32_t n, const uint32_t* rp, const uint32_t* col, const T* val, uint32_t src, int max_sweeps, T* dist, uint8_t* pres) { \
  const uint64_t nnz = rp[n];                                                                                              \
  uint32_t* tp = (uint32_t*)calloc((size_t)n + 2, 4); uint32_t* ti = (uint32_t*)malloc((nnz + 1) * 4); T* tv = (T*)malloc((nnz + 1) * sizeof(T)); \
  for (uint64_t p = 0; p < nnz; p++) tp[col[p] + 2]++;                                                                     \
  for (uint32_t j = 0; j < n; j++) tp[j + 2] += tp[j + 1];                                                                 \
  for (uint32_t i = 0; i < n; i++) for (uint32_t p = rp[i]; p < rp[i + 1]; p++) { const uint32_t q = tp[col[p] + 1]++; ti[q] = i; tv[q] = val[p]; } \
  T* nd = (T*)malloc((size_t)n * sizeof(T)); uint8_t* np_ = (uint8_t*)malloc(n);                                           \
  memset(pres, 0, n); memset(dist, 0, (size_t)n * sizeof(T)); pres[src] = 1; dist[src] = 0;                                \
  int sweeps = 0, changed = 1;                                                                                             \
  while (changed && sweeps < max_sweeps) {                                                                                 \
    changed = 0;                                                                                                           \
    _Pragma("omp parallel for schedule(dynamic, 4096) reduction(| : changed)")                                             \
    for (int64_t j = 0; j < (int64_t)n; j++) {                                                                             \
      T best = dist[j]; uint8_t has = pres[j];                                                                             \
      for (uint32_t q = tp[j]; q < tp[j + 1]; q++) { const uint32_t i = ti[q]; if (!pres[i]) continue;                     \
        const T c = ADD(dist[i], tv[q]); if (!has) { best = c; has = 1; } else if (LESS(c, best)) best = c; }              \
      nd[j] = best; np_[j] = has;                                                                                          \
      if (has != pres[j] || (has && memcmp(&best, &dist[j], sizeof(T)) != 0)) changed = 1;                                 \
    }                                                                                                                      \
    memcpy(dist, nd, (size_t)n * sizeof(T)); memcpy(pres, np_, n); sweeps++;                                               \
  }                                                                                                                        \
  free(tp); free(ti); free(tv); free(nd); free(np_); return sweeps;                                                        \
}
#define SSSP_ADD_I64(a, b) ((int64_t)((uint64_t)(a) + (uint64_t)(b)))
#define SSSP_ADD_F64(a, b) ((a) + (b))
#define SSSP_LESS(a, b) ((a) < (b))
FAST_SSSP(fast_sssp_min_plus_int64, int64_t, SSSP_ADD_I64, SSSP_LESS)
FAST_SSSP(fast_sssp_min_plus_fp64, double, SSSP_ADD_F64, SSSP_LESS)
/* C(i,j) for every (i,j) in L of  C<L> = L (+).(x) L  — the per-entry form of the triangle count (demo/TriangleCentrality.ipynb cell 17,
 * pygraphblas/matrix.py:2401-2584 with mask = L): val == NULL  -> PLUS_PAIR (C(i,j) = |L(i,:) ∩ L(:,j)| as a double, exact below 2^53),
 * val != NULL -> PLUS_TIMES on doubles, products added in ascending k (the Gustavson order).  out / has are aligned with L's entries:
 * has[p] = 0 where no product met mask entry p (C has no entry there). */
void fast_masked_mxm_LL(uint32_t n, const uint32_t* rp, const uint32_t* col, const double* val, double* out, uint8_t* has) {
#pragma omp parallel
  {
    uint32_t* pos = (uint32_t*)calloc((size_t)n + 1, 4);
#pragma omp for schedule(dynamic, 256)
    for (int64_t i = 0; i < (int64_t)n; i++) {
      for (uint32_t p = rp[i]; p < rp[i + 1]; p++) { pos[col[p]] = p + 1; out[p] = 0.0; has[p] = 0; }
      for (uint32_t p = rp[i]; p < rp[i + 1]; p++) { const uint32_t k = col[p]; const double a = val ? val[p] : 1.0;
        for (uint32_t q = rp[k]; q < rp[k + 1]; q++) { const uint32_t t = pos[col[q]]; if (t) { out[t - 1] += val ? a * val[q] : 1.0; has[t - 1] = 1; } } }
      for (uint32_t p = rp[i]; p < rp[i + 1]; p++) pos[col[p]] = 0;
    }
    free(pos);
  }
}
/* The batched betweenness centrality of the reference's GAP driver, gap/bcmark.py:16-67, statement for statement on dense ns x n
 * batches of doubles with presence bytes (the driver computes in FP32; this is what its arithmetic means):
 *   paths = 0; paths[s, src_s] = 1; frontier[s, src_s] = 1
 *   frontier<!paths,replace> = frontier (+).first A                       (pull along AT: rows of the transpose)
 *   while frontier has entries: S[d] = pattern(frontier); paths += frontier; frontier<!paths,replace> = frontier (+).first A
 *   bcu = 1; for i = depth-1 .. 1:  W<S[i],replace> = bcu ./ paths;  W<S[i-1],replace> = W (+).first AT;  bcu += W .* paths
 *   centrality(j) = -ns + sum_s bcu(s, j)
 * A = (rp, col), AT = (rpT, colT), both CSR of the directed graph.  level_nvals[d] = entries of the frontier at level d (d < max_levels).
 * Returns the depth (number of levels whose frontier had entries). */
int fast_bc_batch(uint32_t n, const uint32_t* rp, const uint32_t* col, const uint32_t* rpT, const uint32_t* colT, const uint32_t* sources, int ns,
                  double* cent, int64_t* level_nvals, int max_levels) {
  const size_t N = (size_t)ns * n;
  double* paths = (double*)calloc(N, 8); double* f = (double*)calloc(N, 8); double* g = (double*)calloc(N, 8);
  uint8_t* fp = (uint8_t*)calloc(N, 1); uint8_t* gp = (uint8_t*)calloc(N, 1);
  uint8_t** S = (uint8_t**)calloc((size_t)max_levels + 1, sizeof(uint8_t*));
  for (int s = 0; s < ns; s++) { paths[(size_t)s * n + sources[s]] = 1.0; f[(size_t)s * n + sources[s]] = 1.0; fp[(size_t)s * n + sources[s]] = 1; }
  int depth = 0;
  for (int round = 0;; round++) {
    /* g<!paths,replace> = f (+).first A : g(s,j) = sum over k in AT(j,:) with f(s,k) present, where paths(s,j) == 0 */
    int64_t nv = 0;
#pragma omp parallel for schedule(dynamic, 4096) reduction(+ : nv)
    for (int64_t j = 0; j < (int64_t)n; j++)
      for (int s = 0; s < ns; s++) {
        const size_t o = (size_t)s * n; double acc = 0.0; uint8_t has = 0;
        if (paths[o + j] == 0.0) for (uint32_t q = rpT[j]; q < rpT[j + 1]; q++) { const uint32_t k = colT[q]; if (fp[o + k]) { acc += f[o + k]; has = 1; } }
        g[o + j] = acc; gp[o + j] = has; nv += has;
      }
    { double* t = f; f = g; g = t; uint8_t* tp = fp; fp = gp; gp = tp; }
    if (round > 0) depth = round;       /* (the first product precedes the loop of the driver) */
    if (nv == 0 || depth >= max_levels) break;
    if (depth < max_levels) level_nvals[depth] = nv;
    S[depth] = (uint8_t*)malloc(N); memcpy(S[depth], fp, N);
    for (size_t t = 0; t < N; t++) if (fp[t]) paths[t] += f[t];
  }
  /* at this point S[0..depth-1] hold the frontiers' patterns (S[d] = the frontier found by product d) */
  double* bcu = (double*)malloc(N * 8); for (size_t t = 0; t < N; t++) bcu[t] = 1.0;
  double* W = f; uint8_t* Wp = fp; double* W2 = g; uint8_t* W2p = gp;
  for (int i = depth - 1; i > 0; i--) {
    for (size_t t = 0; t < N; t++) { Wp[t] = S[i][t]; W[t] = S[i][t] ? bcu[t] / paths[t] : 0.0; }
    /* W2<S[i-1],replace> = W (+).first AT : W2(s,j) = sum over k in A(j,:) with W(s,k) present */
#pragma omp parallel for schedule(dynamic, 4096)
    for (int64_t j = 0; j < (int64_t)n; j++)
      for (int s = 0; s < ns; s++) {
        const size_t o = (size_t)s * n; double acc = 0.0; uint8_t has = 0;
        if (S[i - 1][o + j]) for (uint32_t q = rp[j]; q < rp[j + 1]; q++) { const uint32_t k = col[q]; if (Wp[o + k]) { acc += W[o + k]; has = 1; } }
        W2[o + j] = acc; W2p[o + j] = has;
      }
    for (size_t t = 0; t < N; t++) if (W2p[t]) bcu[t] += W2[t] * paths[t];
  }
  for (uint32_t j = 0; j < n; j++) { double c = -(double)ns; for (int s = 0; s < ns; s++) c += bcu[(size_t)s * n + j]; cent[j] = c; }
  for (int d = 0; d <= max_levels; d++) free(S[d]);
  free(S); free(paths); free(f); free(g); free(fp); free(gp); free(bcu);
  return depth;
}
/* Rows `rows[0..ns)` of the UNMASKED product C = A (+).(x) A over PLUS_TIMES FP64 (lib.GrB_mxm with mask = NULL,
 * pygraphblas/matrix.py:2572-2583; C API 1.3 / SURVEY.md Appendix A item 2: C(i,j) exists iff some k has A(i,k) and A(k,j) stored),
 * Gustavson row by row with a dense accumulator, products added in ascending k.  Two calls: with out_col == NULL it fills
 * counts[s] = entries of row rows[s] and products[s] = sum_k nnz(A(k,:)); with out_col / out_val it writes row s, columns ascending,
 * at offsets[s].  Used by the tests and bench.py to check SAMPLED rows of a product too large for any host (A @ A on R-MAT-18:
 * 3e9 entries), and timed as the CPU baseline of that product. */
void fast_mxm_rows_plus_times_fp64(uint32_t n, const uint32_t* rp, const uint32_t* col, const double* val, uint32_t ns, const uint32_t* rows,
                                   int64_t* counts, int64_t* products, const int64_t* offsets, uint32_t* out_col, double* out_val) {
#pragma omp parallel
  {
    double* acc = (double*)calloc((size_t)n, 8); uint8_t* has = (uint8_t*)calloc((size_t)n, 1);
#pragma omp for schedule(dynamic, 1)
    for (int64_t s = 0; s < (int64_t)ns; s++) {
      const uint32_t i = rows[s]; int64_t cnt = 0, prod = 0; uint32_t lo = n, hi = 0;
      for (uint32_t p = rp[i]; p < rp[i + 1]; p++) { const uint32_t k = col[p]; const double a = val[p]; prod += rp[k + 1] - rp[k];
        for (uint32_t q = rp[k]; q < rp[k + 1]; q++) { const uint32_t j = col[q]; acc[j] += a * val[q]; if (!has[j]) { has[j] = 1; cnt++; if (j < lo) lo = j; if (j > hi) hi = j; } } }
      if (!out_col) { counts[s] = cnt; products[s] = prod; }
      int64_t o = out_col ? offsets[s] : 0;
      if (cnt) for (uint32_t j = lo; j <= hi; j++) if (has[j]) { if (out_col) { out_col[o] = j; out_val[o] = acc[j]; o++; } acc[j] = 0.0; has[j] = 0; }
    }
    free(acc); free(has);
  }
}
int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void oracle_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
