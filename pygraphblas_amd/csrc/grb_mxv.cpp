// grb_mxv.cpp — GrB_mxv and GrB_vxm: the SpMV half of the hot path.
//
// Replaces lib.GrB_mxv (reference call site pygraphblas/matrix.py:2714-2725, Matrix.mxv) and
// lib.GrB_vxm (pygraphblas/vector.py:960-970, Vector.vxm); semantics per SURVEY.md App. A:
//   w<mask,replace> = accum(w, op(A) (+).(x) u)         mxv: multiply is A(i,j) (x) u(j),  desc.INP0 transposes A
//   w'<mask',replace> = accum(w', u' (+).(x) op(A))     vxm: multiply is u(i) (x) A(i,j),  desc.INP1 transposes A
// Both are executed in "mxv orientation" t = M (+).(x) u with M = op(A) (mxv) or op(A)^T (vxm);
// whichever of CSR(A) / CSR(A^T) the chosen direction needs is taken from the matrix (the transpose
// is built once on the device and cached).
#include "grb_opcommon.hpp"
#include "grb_spmv.hpp"
#include "grb_lazy.hpp"
#include <cmath>
#include <algorithm>

using namespace grb;

static int g_force_method = SPMV_AUTO;   // test hook: GRB_MI355X_SPMV=adaptive|rowgroup|push
// Where the next product on this thread writes T (round 6, grb_mxm_rows.cpp): a row of a batch matrix's bitmap.  The kernels write the row sums and presence
// bytes there instead of into fresh buffers, and when the write-back makes w exactly T — the batch products' case — w ends up as a VIEW of that row: the
// ns x (n values + n bytes) copies per product of the first bitmap version (3.8 of 14 ms of the BC driver at R-MAT-22) are gone.  One-shot: cleared by the call.
namespace grb { thread_local void* g_mxv_dest_val = nullptr; thread_local uint8_t* g_mxv_dest_pres = nullptr; }

static void mxv_like(GrB_Vector w, GrB_Vector mask, GrB_BinaryOp accum, GrB_Semiring semiring, GrB_Matrix A, GrB_Vector u,
                     GrB_Descriptor desc, bool is_vxm) {
  need_device();
  if (!check_obj(w) || !check_obj(A) || !check_obj(u) || (mask && !check_obj(mask)))
    fail(GrB_UNINITIALIZED_OBJECT, "mxv/vxm: uninitialised operand");
  if (is_hyper(A) || is_hyper(w) || is_hyper(u) || is_hyper(mask)) { hyper_mxv_like(w, mask, accum, semiring, A, u, desc, is_vxm); return; }   // dimensions beyond the device layouts
  if (w->q_reads || w->lazy == 2) vec_gate(w);                // deferred element-wise work on the output completes first (a pending fill, lazy == 1, is dealt with below)
  const DescView dv(desc);
  const bool useT = is_vxm ? !dv.tran1 : dv.tran0;           // M = useT ? A^T : A
  const uint64_t mr = useT ? A->ncols : A->nrows, mc = useT ? A->nrows : A->ncols;
  if (u->n != mc || w->n != mr || (mask && mask->n != mr)) fail(GrB_DIMENSION_MISMATCH, "mxv/vxm: dimensions do not conform");
  SemiringDesc sd = make_semiring_desc(semiring, /*swap_mult_args=*/is_vxm);
  const char* env = getenv("GRB_MI355X_SPMV");
  int method = g_force_method;
  if (env) method = !strcmp(env, "adaptive") ? SPMV_ADAPTIVE : !strcmp(env, "rowgroup") ? SPMV_ROWGROUP : !strcmp(env, "push") ? SPMV_PUSH : !strcmp(env, "wavepipe") ? SPMV_WAVEPIPE : !strcmp(env, "xcd") ? SPMV_XCD : SPMV_AUTO;
  g_last_plan.clear();

  DevBuf allow_buf, ubool; bool nothing = false;
  const uint8_t* allow = nullptr;
  // the mask is also the operand and the semiring is Boolean (`v.vxm(A, mask=v, desc=RC)` on the UINT8 level vector of a BFS): its allow
  // bytes and its values as BOOL come out of ONE pass instead of a k_allow and a k_cast — made below, once the direction is known: the
  // masked pull of a one-byte vector reads the vector itself and needs neither (SpmvCall::fm_val)
  const bool mask_is_u = mask && mask == u && sd.zcode == T_BOOL && u->type->code != T_BOOL && mr;
  if (mask_is_u) vec_to_device(u);
  else allow = vector_allow(mask, dv, mr, allow_buf, &nothing);
  if (nothing) {   // no mask + complement: nothing may be written; replace clears w
    if (dv.replace) { w->hi.clear(); w->hx.clear(); w->pending.clear(); w->host_valid = true; vec_invalidate_device(w); }
    return;
  }
  mat_to_device(A); vec_to_device(u);
  // the entry count of u and (for the direction choice below) the edges leaving it come from one kernel and one round
  // trip when the count is not known yet — the usual state inside a BFS loop, where u was just updated under a mask
  uint64_t fe_cached = ~0ull;
  bool stays_pull = false;            // the operand was too heavy for a push step when it was last counted and has only grown since
  // (masked products — the frontier-like operands of a BFS — and, since round 4, unmasked ones whose operand has an unknown count: the
  //  sweeps of the shortest-path loop, whose operand only gains entries; never build a transpose just for this)
  if (method == SPMV_AUTO && spmspv_push_supported(sd) && u->n && (useT || A->csc.valid)) {
    const DevCSR& P0 = useT ? A->csr : mat_csc(A);
    if (u->fe_lb_key == P0.rowptr.serial && P0.rowptr.serial && u->fe_lb * 16 >= P0.nnz + 16) stays_pull = true;      // no kernel, no host round trip (whether or not the count is known: `w.iseq(v)` of the shortest-path loop counts v)
    else if (!u->dnvals_known) {
      uint64_t cnt = 0;
      fe_cached = frontier_edges_and_count(u->dpres.as<uint8_t>(), P0.rowptr.as<uint32_t>(), u->n, &cnt);
      u->dnvals = cnt; u->dnvals_known = true; u->fe_lb = fe_cached; u->fe_lb_key = P0.rowptr.serial; u->fe_lb_true = false;
    }
  }
  const uint64_t u_nvals = stays_pull && !u->dnvals_known ? (u->n ? u->n - 1 : 0) : vec_dev_nvals(u);             // (not counted: treated as "has holes")
  const bool u_full = u_nvals == u->n;

  // does the multiply read the matrix / vector values at all?
  const bool uses_a = sd.flip ? binop_uses_y(sd.mulop) : binop_uses_x(sd.mulop);
  const bool uses_u = sd.flip ? binop_uses_x(sd.mulop) : binop_uses_y(sd.mulop);

  // direction (Beamer-style): push walks the rows of the frontier, pull scans the rows of the output.  Push is
  // considered only for a sparse operand and a monoid with a native atomic; it is taken when the edges leaving the
  // frontier (an exact count on the device) are < 1/16 of all entries.
  bool push = false;
  // an operand of at most 64 entries known as a list on the host (the first level of a BFS): push without counting its edges — 64 rows,
  // the long ones split over all workgroups, are never worth a pull over every row of a large matrix
  const bool tiny = u->small_valid && u->small_idx.size() == u_nvals && u_nvals <= 64 && (uint64_t)A->csr.nnz >= (1u << 20) && (useT || A->csc.valid);
  if (method == SPMV_PUSH) push = spmspv_push_supported(sd);
  else if (method == SPMV_AUTO && !stays_pull && spmspv_push_supported(sd) && !u_full && u_nvals * 16 < (uint64_t)A->csr.nnz + 16) {
    if (tiny) push = true;
    else {
      const DevCSR& P = useT ? A->csr : mat_csc(A);
      const uint64_t fe = fe_cached != ~0ull ? fe_cached : frontier_edges(u->dpres.as<uint8_t>(), P.rowptr.as<uint32_t>(), u->n);
      push = fe * 16 < P.nnz + 16;
      u->fe_lb = fe; u->fe_lb_key = P.rowptr.serial; u->fe_lb_true = false;                       // (exact now, a lower bound while entries are only added)
    }
  }

  bool fused_mask = false, excl_small = false;
  if (mask_is_u) {
    // (the pull operand — for vxm the cached transpose, built on first need — is looked at only when the product pulls: a BFS that
    //  stays in push for every level never pays for a transpose it does not use)
    fused_mask = !push && !accum && dv.replace && type_size(u->type->code) == 1 && spmv_rowlane_applies(useT ? mat_csc(A) : A->csr, sd, method);
    // the first level of a BFS: the operand is a short list known on the host, it is also the (complemented) mask and every listed value is true — the
    // push kernels skip the listed positions themselves and multiply by `true`: no allow bytes, no BOOL copy of the operand (a pass over all positions)
    // (only for `w<!u, replace> = ...` without an accumulator, like the fused pull: with `allow == nullptr` the write-back makes w exactly T, which is
    //  right only when the listed positions are to be deleted anyway and nothing of w's old content survives — ADVICE round 5)
    excl_small = !fused_mask && push && tiny && !accum && dv.replace && dv.mask_comp && u->small_truthy;
    if (!fused_mask && !excl_small) {
      allow_buf.alloc(mr); ubool.alloc(mr + 1);
      build_allow_and_bool(mr, u->type->code, u->dval.p, u->dpres.as<uint8_t>(), dv.mask_struct, dv.mask_comp, allow_buf.as<uint8_t>(), ubool.as<uint8_t>());
      allow = allow_buf.as<uint8_t>();
    }
  }
  const size_t zs = type_size(sd.zcode);
  DevBuf tval, tpres, ucast, acast;
  if (g_mxv_dest_val && g_mxv_dest_pres) { tval.borrow(g_mxv_dest_val, mr * zs); tpres.borrow(g_mxv_dest_pres, mr); } else { tval.alloc(mr * zs + 1); tpres.alloc(mr + 1); }
  g_mxv_dest_val = nullptr; g_mxv_dest_pres = nullptr;
  // An operand with holes whose product is accumulated into a full vector with the monoid's own operator, no mask (PageRank:
  // r<accum PLUS> += A' (+).second w, w = t / d has no entry for dangling vertices — gap/prmark.py:21-23): where T has no
  // entry the output keeps its value, and where T's entries would come from absent operand entries only, accumulating the
  // monoid's identity keeps it too.  So the pattern of T does not matter, and the holes can be filled with a value z for
  // which mult(a, z) is the identity — z = identity for SECOND (any monoid), 0 for integer PLUS_TIMES, false for LOR_LAND —
  // in the same pass that casts the operand; the product then runs as a full-operand one (kernel X / W instead of the
  // bitmap variant of the row-block kernel: 0.60 -> 0.17 ms at R-MAT-22).
  // w is full: resident in HBM with every position present, or `w(:) = s` was requested and not written yet (non-blocking mode, grb_lazy.cpp)
  const bool w_fill = w->lazy == 1 && w != u && w != mask;
  const bool w_full = w_fill || (w->lazy == 0 && w->dev_valid && !w->host_valid && w->dnvals_known && w->dnvals == w->n && w != u);
  const bool accum_is_monoid = !mask && accum && check_obj(accum) && accum->opcode == sd.addop && accum->xtype->code == sd.zcode && accum->ytype->code == sd.zcode &&
                               accum->ztype->code == sd.zcode && w->type->code == sd.zcode;
  bool fill_holes = false;
  if (!push && !u_full && accum_is_monoid && method == SPMV_AUTO && !sd.flip && w_full) {
    const bool is_int = sd.zcode != T_FP32 && sd.zcode != T_FP64 && sd.zcode != T_BOOL;
    // (the accumulator must leave w alone when it meets the identity: true for these monoid operators, not for ANY)
    const bool neutral = sd.addop == B_PLUS || sd.addop == B_TIMES || sd.addop == B_MIN || sd.addop == B_MAX || sd.addop == B_LOR || sd.addop == B_LAND || sd.addop == B_LXOR;
    fill_holes = neutral && (sd.mulop == B_SECOND || (sd.mulop == B_TIMES && sd.addop == B_PLUS && is_int) || (sd.mulop == B_LAND && sd.addop == B_LOR && sd.zcode == T_BOOL));
  }
  // "Big holes": a MIN_PLUS / MAX_PLUS product over an operand with holes and no mask (the sweeps of the reference's shortest-path
  // loop, `v<accum MIN> = v MIN_PLUS A`: v has no entry for the vertices not reached yet).  The full-operand pipeline kernels cannot
  // skip absent entries, and no value z makes a + z the monoid's identity for every a.  But when the values are small against the
  // type's range, a BIG fill does the same job exactly: every sum that touches a hole lands beyond a threshold no real sum can
  // reach, so "T(i) is an entry" is "T(i) is on the near side of the threshold" — one pass over the result.  Conditions (else the
  // bitmap variant of the row-block kernel runs, as before): |A's values| and |u's values| below a quarter of BIG (integers: BIG =
  // 2^(bits-2), so nothing wraps; floating point: BIG = infinity and every value finite, sums not overflowing), measured once per
  // matrix and once per call.  R-MAT-22 INT64: 1.0 -> 0.3 ms per sweep.
  bool big_holes = false; uint8_t big_fill[16] = {0}, big_thresh[16] = {0}; double big_uabs = 0, big_aabs = 0;
  if (!push && !u_full && !fill_holes && !allow && method == SPMV_AUTO && !sd.flip && sd.mulop == B_PLUS && (sd.addop == B_MIN || sd.addop == B_MAX) &&
      (sd.zcode == T_INT32 || sd.zcode == T_INT64 || sd.zcode == T_FP32 || sd.zcode == T_FP64) && A->type->code == sd.zcode && u->type->code == sd.zcode) {
    DevCSR& R = useT ? const_cast<DevCSR&>(mat_csc(A)) : A->csr;
    if (R.nnz >= (1u << 20) && u_nvals * 64 >= u->n) {                       // a product the pipeline kernels take, an operand that is not nearly empty
      if (R.range_state == 0) {
        uint8_t mn[8], mx[8]; uint64_t bad = 0, cnt = 0;
        R.range_state = 2;
        if (value_range(sd.zcode, R.nnz, R.val.p, nullptr, mn, mx, &bad, &cnt) && bad == 0 && cnt) {
          double a = 0, b = 0;
          if (sd.zcode == T_INT32) { int32_t x, y; memcpy(&x, mn, 4); memcpy(&y, mx, 4); a = (double)x; b = (double)y; }
          else if (sd.zcode == T_INT64) { int64_t x, y; memcpy(&x, mn, 8); memcpy(&y, mx, 8); a = (double)x; b = (double)y; }
          else if (sd.zcode == T_FP32) { float x, y; memcpy(&x, mn, 4); memcpy(&y, mx, 4); a = x; b = y; }
          else { memcpy(&a, mn, 8); memcpy(&b, mx, 8); }
          R.range_abs = std::max(std::fabs(a), std::fabs(b)); R.range_state = 1;
        }
      }
      if (R.range_state == 1) {
        uint8_t mn[8], mx[8]; uint64_t bad = 0, cnt = 0;
        const bool bound_known = u->abs_bound >= 0 && u->lazy == 0 && u->dev_valid;          // left by the previous sweep: no kernel, no read-back
        if (bound_known || (value_range(sd.zcode, u->n, u->dval.p, u->dpres.as<uint8_t>(), mn, mx, &bad, &cnt) && bad == 0 && cnt)) {
          double a = 0, b = 0; const bool is_min = sd.addop == B_MIN;
          if (bound_known) a = b = u->abs_bound;
          else if (sd.zcode == T_INT32) { int32_t x, y; memcpy(&x, mn, 4); memcpy(&y, mx, 4); a = (double)x; b = (double)y; }
          else if (sd.zcode == T_INT64) { int64_t x, y; memcpy(&x, mn, 8); memcpy(&y, mx, 8); a = (double)x; b = (double)y; }
          else if (sd.zcode == T_FP32) { float x, y; memcpy(&x, mn, 4); memcpy(&y, mx, 4); a = x; b = y; }
          else { memcpy(&a, mn, 8); memcpy(&b, mx, 8); }
          const double uabs = std::max(std::fabs(a), std::fabs(b));
          big_uabs = uabs; big_aabs = R.range_abs;
          if (sd.zcode == T_INT32 && R.range_abs < 268435456.0 && uabs < 268435456.0) {            // 2^28: real sums within +-2^29, hole sums beyond +-(2^30 - 2^28)
            const int32_t f = is_min ? (1 << 30) : -(1 << 30), th = is_min ? (1 << 29) + (1 << 28) : -((1 << 29) + (1 << 28));
            memcpy(big_fill, &f, 4); memcpy(big_thresh, &th, 4); big_holes = true;
          } else if (sd.zcode == T_INT64 && R.range_abs < 1.15e18 && uabs < 1.15e18) {               // < 2^60
            const int64_t f = is_min ? (1ll << 62) : -(1ll << 62), th = is_min ? (1ll << 61) + (1ll << 60) : -((1ll << 61) + (1ll << 60));
            memcpy(big_fill, &f, 8); memcpy(big_thresh, &th, 8); big_holes = true;
          } else if (sd.zcode == T_FP32 && R.range_abs < 8e37 && uabs < 8e37) {                       // sums stay finite
            const float f = is_min ? INFINITY : -INFINITY; memcpy(big_fill, &f, 4); memcpy(big_thresh, &f, 4); big_holes = true;
          } else if (sd.zcode == T_FP64 && R.range_abs < 4e307 && uabs < 4e307) {
            const double f = is_min ? (double)INFINITY : -(double)INFINITY; memcpy(big_fill, &f, 8); memcpy(big_thresh, &f, 8); big_holes = true;
          }
        }
      }
    }
  }
  const void* uval = nullptr;
  const bool zero_fill = [&] { for (size_t b = 0; b < zs; b++) if (sd.identity[b]) return false; return true; }();
  if (uses_u && fill_holes && u->holes_zero && u->type->code == sd.zcode && zero_fill) {
    uval = u->dval.p;                                     // written by the element-wise chain kernel with zeros in the holes: no pass at all
  } else if (big_holes) {
    // the fill goes into u's OWN buffer (the values of absent positions are nobody's business: holes_zero is dropped) and is remembered: the
    // sweeps of the shortest-path loop only add entries — real values, written by the merge's store — so from the second sweep on the operand
    // is ready as it stands (a cast-and-fill pass over the vector per sweep before: 18 us of 357 at R-MAT-22)
    // (the operand is an INPUT: its stored entries never change, only the bytes behind its holes do — a write all the same.  Not while queued work
    //  still reads the vector, nor while an exchange writes into vector buffers on the second stream: then the fill goes into a copy, as before round 4)
    if (u->holes_big && memcmp(u->holes_big_val, big_fill, zs) == 0) uval = u->dval.p;
    else if (u->q_reads || dist_exchange_pending()) {
      ucast.alloc(u->n * zs + 1);
      vec_cast_fill_values(sd.zcode, ucast.p, u->type->code, u->dval.p, u->dpres.as<uint8_t>(), u->n, big_fill);
      uval = ucast.p;
    } else {
      vec_cast_fill_values(sd.zcode, u->dval.p, u->type->code, u->dval.p, u->dpres.as<uint8_t>(), u->n, big_fill);      // (same type: big_holes requires it; element-wise, in place)
      u->holes_zero = false; u->holes_big = true; memcpy(u->holes_big_val, big_fill, 16);
      uval = u->dval.p;
    }
  } else if (uses_u && fill_holes) {
    ucast.alloc(u->n * zs + 1);
    vec_cast_fill_values(sd.zcode, ucast.p, u->type->code, u->dval.p, u->dpres.as<uint8_t>(), u->n, sd.identity);   // (identity of PLUS / LOR is the 0 / false the two other cases need)
    uval = ucast.p;
  } else if (uses_u && (fused_mask || excl_small)) uval = nullptr;  // (the kernel reads the vector's own bytes / takes every operand value as true)
  else if (uses_u) uval = ubool.p ? ubool.p : cast_values(sd.zcode, u->type->code, u->dval.p, u->n, ucast);

  SpmvCall call{};
  call.uval = uval; call.allow = allow; call.tval = tval.p; call.tpres = tpres.as<uint8_t>(); call.method = method;
  // a BOOL result that replaces w: the kernel notes whether it wrote a true value, and the `q.reduce_bool()` that follows a BFS
  // level (tests/test_bfs.py loop: `while q.reduce_bool() and level <= n`) reads that word instead of scanning q
  bool any_done = false;
  bool fe_done = false; uint64_t fe_key = 0; uint32_t fe_nblocks = 0;
  if (sd.zcode == T_BOOL && w->type->code == T_BOOL && !accum && method == SPMV_AUTO) {
    call.any_true = any_true_acquire(&call.any_true_tag); call.any_true_done = &any_done;
    // ... and, on a square matrix, the edges leaving the result's true entries in the row pointers the direction choice above counts in:
    // the product after `v[q] = level` then knows its operand is too heavy for a push step without counting (SpmvCall::fe_slots)
    if (mr == mc && (useT || A->csc.valid)) {
      const DevCSR& P0 = useT ? A->csr : mat_csc(A);
      if (P0.nrows == mr && P0.rowptr.serial) { call.fe_rowptr = P0.rowptr.as<uint32_t>(); call.fe_host = fe_summary_host(); call.fe_done = &fe_done; call.fe_nblocks = &fe_nblocks; fe_key = P0.rowptr.serial; }
    }
  }
  const void* const tkey = tval.p;
  if (push) {
    // push walks rows of M^T:  M^T = useT ? A : A^T
    DevCSR& P = useT ? A->csr : const_cast<DevCSR&>(mat_csc(A));
    call.M = &P; call.upres = u->dpres.as<uint8_t>();
    if (u->small_valid && u->small_idx.size() == u_nvals) { call.small_idx = u->small_idx.data(); call.small_n = (uint32_t)u_nvals; call.excl_small = excl_small; }
    call.aval = uses_a ? cast_values(sd.zcode, A->type->code, P.val.p, P.nnz, acast) : nullptr;
    spmspv_push(call, sd, u_nvals);
  } else {
    DevCSR& R = useT ? const_cast<DevCSR&>(mat_csc(A)) : A->csr;
    call.M = &R; call.upres = (u_full || fill_holes || big_holes) ? nullptr : u->dpres.as<uint8_t>();
    if (fused_mask) {
      call.upres = u->dpres.as<uint8_t>(); call.fm_val = u->dval.as<uint8_t>(); call.fm_flags = (uint8_t)((dv.mask_struct ? 1 : 0) | (dv.mask_comp ? 2 : 0));
      // the vector's code bytes (one gather per neighbour instead of two): left behind by the masked assign that wrote it (`v[q] = level`), else made by one pass
      // over the vector when the matrix is large enough for the pull to repay it (GRB_MI355X_CODE_BYTES=0: never)
      const char* ce = getenv("GRB_MI355X_CODE_BYTES"); const bool code_off = ce && atoi(ce) == 0;      // (read per call: a test hook)
      if (!code_off && u->n >= (1u << 16) && R.nnz >= (1u << 20)) {
        if (!u->code_valid) {
          if (!u->dcode.p || u->dcode.bytes < u->n + 16) u->dcode.alloc(u->n + 16);
          vec_code_bytes(u->n, u->dval.as<uint8_t>(), u->dpres.as<uint8_t>(), u->dcode.as<uint8_t>());
          u->code_valid = true;
        }
        call.fm_code = u->dcode.as<uint8_t>();
      }
    }
    call.aval = uses_a ? cast_values(sd.zcode, A->type->code, R.val.p, R.nnz, acast) : nullptr;
    // `w += M (+).(x) u` with the monoid's own operator into a full w, no mask: the kernel that writes the row sums can apply the
    // accumulator in the same store — and when w is a fill that was never written (`r[:] = teleport` before the product of
    // gap/prmark.py:21-23), the fill folds into that store too and w is never read
    bool epi_done = false;
    if (accum_is_monoid && w_full && method == SPMV_AUTO && !big_holes) {       // (with big holes every row has a sum: the threshold must see it first — epi 3)
      call.epi = w_fill ? 2 : 1; call.epi_w = w_fill ? nullptr : w->dval.p; call.epi_done = &epi_done;
      if (w_fill) memcpy(call.epi_fill, w->lazy_fill, 16);
    }
    // big holes + `w<accum MIN> = ...` with the monoid's own operator (the sweeps of the shortest-path loop, `v.vxm(A, MIN_PLUS, accum=MIN, out=v)`):
    // the merge kernel applies the threshold and the accumulator in its store — no threshold pass over T, no accumulate epilogue, no T at all
    const bool big_epi = big_holes && accum_is_monoid && method == SPMV_AUTO && w->lazy == 0 && !w->q_reads;
    if (big_epi) {
      vec_to_device(w);
      call.epi = 3; call.epi_w = w->dval.p; call.epi_wpres = w->dpres.as<uint8_t>(); memcpy(call.epi_fill, big_thresh, 16); call.epi_done = &epi_done;
    }
    spmv_pull(call, sd);
    if (epi_done && call.epi == 3) {
      const bool keep_big = w == u && u->holes_big;                       // (the store wrote real values into former holes, nothing else: the others still hold the fill)
      vec_invalidate_host(w);
      w->holes_big = keep_big;
      w->dnvals_known = false; w->dnvals = 0; w->holes_zero = false;      // (entries were only added: the lower bound fe_lb of the edges leaving them stays valid — the next sweep needs no count)
      w->abs_bound = big_uabs + big_aabs;       // every sum that passed the threshold is within |u| + |A|; MIN / MAX select among such values and w's own (w is u, or held values of an earlier sweep)
      if (w != u) w->abs_bound = -1;
      return;
    }
    if (epi_done) {
      if (w_fill) {
        lazy_fill_consumed(w);
        w->dval = std::move(tval); w->dpres = std::move(tpres);
        w->dev_valid = true; w->host_valid = false; w->hi.clear(); w->hx.clear(); w->pending.clear();
      } else vec_invalidate_host(w);
      w->dnvals = w->n; w->dnvals_known = true; w->fe_lb = 0; w->fe_lb_key = 0; w->holes_zero = false; w->holes_big = false;
      return;
    }
  }
  if (big_holes) big_to_absent(sd.zcode, mr, tval.p, tpres.as<uint8_t>(), big_thresh, sd.addop == B_MIN);       // sums made of fill values only are no entries
  const bool w_is_u = w == u; const bool w_was_empty = !w_is_u && w->lazy == 0 && w->dnvals_known && w->dnvals == 0 && !w->host_valid;
  vector_write_back(w, sd.zcode, tval, tpres, allow, accum, dv.replace, /*t_only_allowed=*/true);
  // every sum that survived the threshold is within |u| + |A|: the bound of the result when w held nothing else (w was empty / is
  // replaced as a whole), or when w is u itself and the accumulator SELECTS one of its arguments (`v<accum MIN> = v MIN_PLUS A`: MIN,
  // MAX, FIRST, SECOND, ANY).  An arithmetic accumulator (PLUS, TIMES ...) can leave |w| beyond it: then no bound is recorded and the
  // next sweep measures the range again.
  const bool accum_selects = accum && check_obj(accum) && (accum->opcode == B_MIN || accum->opcode == B_MAX || accum->opcode == B_FIRST || accum->opcode == B_SECOND || accum->opcode == B_ANY);
  if (big_holes && w->type->code == sd.zcode && (!accum || w_was_empty || (w_is_u && accum_selects))) w->abs_bound = big_uabs + big_aabs;
  if (any_done) any_true_written(w->lazy == 0 && w->dev_valid && w->dval.p == tkey ? w : nullptr, tkey, call.any_true_tag, fe_done ? fe_key : 0, fe_nblocks);       // (adopted as they are: w is exactly T)
}

extern "C" {

GrB_Info GrB_mxv(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Semiring semiring,
                 const GrB_Matrix A, const GrB_Vector u, const GrB_Descriptor desc) {
  if (!w || !A || !u || !semiring) return GrB_NULL_POINTER;
  if (!check_obj(w)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(w, [&] { mxv_like(w, mask, accum, semiring, A, u, desc, false); });
}

GrB_Info GrB_vxm(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Semiring semiring,
                 const GrB_Vector u, const GrB_Matrix A, const GrB_Descriptor desc) {
  if (!w || !A || !u || !semiring) return GrB_NULL_POINTER;
  if (!check_obj(w)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(w, [&] { mxv_like(w, mask, accum, semiring, A, u, desc, true); });
}

}  // extern "C"
