// grb_container.cpp — GrB_Matrix / GrB_Vector / GxB_Scalar objects and their host<->HBM mirrors.
//
// Replaces the object-model half of the `lib.` surface that pygraphblas calls:
//   GrB_Matrix_new/free/dup/clear/nrows/ncols/nvals/wait/error/resize/removeElement
//        (reference call sites: pygraphblas/matrix.py:99-117, 171-175, 627-760, 3353)
//   GrB_Matrix_build_<T> / setElement_<T> / extractElement_<T> / extractTuples_<T>
//        (pygraphblas/matrix.py:325-330, 1503-1529, 3180-3308; types.py:87-110)
//   the Vector and Scalar equivalents (pygraphblas/vector.py:60-95, scalar.py:20-92).
// Ownership: the library allocates in *_new and frees in *_free(&h); free of NULL / already freed
// handles is a successful no-op (reference __del__ checks the return code).
#include "grb_api.hpp"
#include "grb_device.hpp"
#include "grb_matops.hpp"
#include <algorithm>
#include <numeric>
#include <string.h>
#include <stdarg.h>
#include <stdio.h>

namespace grb {

// ---- host assemble: apply pending setElement/removeElement in program order ---------------------
static const char* const ISO_MSG = "a full one-valued container of this dimension can be read element-wise only";
void mat_host_assemble(GrB_Matrix A) {
  if (A->iso_full) fail(GrB_INSUFFICIENT_SPACE, ISO_MSG);
  if (A->pending.empty()) return;
  const size_t ts = A->type->size;
  auto& P = A->pending;
  if (P.size() <= 8) {            // a few edits (`M[i, j] = x` then a read): in place, one memmove each, instead of rebuilding the tuple arrays
    for (const auto& e : P) {
      size_t lo = 0, hi = A->hi.size();
      while (lo < hi) { const size_t m = (lo + hi) / 2; if (A->hi[m] < e.i || (A->hi[m] == e.i && A->hj[m] < e.j)) lo = m + 1; else hi = m; }
      const bool found = lo < A->hi.size() && A->hi[lo] == e.i && A->hj[lo] == e.j;
      if (e.del) { if (found) { A->hi.erase(A->hi.begin() + lo); A->hj.erase(A->hj.begin() + lo); A->hx.erase(A->hx.begin() + lo * ts, A->hx.begin() + (lo + 1) * ts); } }
      else if (found) memcpy(&A->hx[lo * ts], e.x, ts);
      else { A->hi.insert(A->hi.begin() + lo, e.i); A->hj.insert(A->hj.begin() + lo, e.j); A->hx.insert(A->hx.begin() + lo * ts, e.x, e.x + ts); }
    }
    P.clear(); return;
  }
  std::vector<uint32_t> ord(P.size());
  std::iota(ord.begin(), ord.end(), 0u);
  std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) {
    return P[a].i != P[b].i ? P[a].i < P[b].i : P[a].j < P[b].j; });
  std::vector<GrB_Index> ni, nj; std::vector<uint8_t> nx;
  ni.reserve(A->hi.size() + P.size()); nj.reserve(A->hi.size() + P.size()); nx.reserve((A->hi.size() + P.size()) * ts);
  size_t b = 0, nb = A->hi.size(), k = 0;
  auto push_base = [&](size_t q) { ni.push_back(A->hi[q]); nj.push_back(A->hj[q]); nx.insert(nx.end(), &A->hx[q * ts], &A->hx[q * ts] + ts); };
  while (k < ord.size()) {
    size_t e = k;   // [k, e] = run of edits to the same coordinate; the last one wins
    while (e + 1 < ord.size() && P[ord[e + 1]].i == P[ord[k]].i && P[ord[e + 1]].j == P[ord[k]].j) e++;
    const auto& last = P[ord[e]];
    while (b < nb && (A->hi[b] < last.i || (A->hi[b] == last.i && A->hj[b] < last.j))) push_base(b++);
    if (b < nb && A->hi[b] == last.i && A->hj[b] == last.j) b++;   // replaced or deleted
    if (!last.del) { ni.push_back(last.i); nj.push_back(last.j); nx.insert(nx.end(), last.x, last.x + ts); }
    k = e + 1;
  }
  while (b < nb) push_base(b++);
  A->hi.swap(ni); A->hj.swap(nj); A->hx.swap(nx); P.clear(); P.shrink_to_fit();
}

void mat_invalidate_device(GrB_Matrix A) { A->dev_valid = false; A->csr.clear(); A->csc.clear(); A->bm.clear(); }
void mat_invalidate_host(GrB_Matrix A) {
  A->host_valid = false; A->hi.clear(); A->hj.clear(); A->hx.clear(); A->pending.clear(); A->dev_elem_ops = 0;
  A->hi.shrink_to_fit(); A->hj.shrink_to_fit(); A->hx.shrink_to_fit();
  A->csc.clear(); A->csr.has_plan = false; A->csr.plan_blocks.reset(); A->csr.plan_aux.reset();
  A->csr.wp_rs.reset(); A->csr.wp_hot.reset(); A->csr.wp_pcol.reset(); A->csr.wp_tsize = 0; A->csr.xcd.reset();
  A->csr.heads_valid = false; A->csr.range_state = 0;      // (the device copy was rewritten: whatever was derived from its values goes with the plans)
}

void mat_to_host(GrB_Matrix A) {
  if (A->iso_full) fail(GrB_INSUFFICIENT_SPACE, ISO_MSG);
  if (A->host_valid) { mat_host_assemble(A); return; }
  if (mat_bitmap_only(A)) mat_to_device(A);                  // a batch matrix that lives as a bitmap: its CSR first
  // download the device CSR and expand to sorted tuples
  const DevCSR& c = A->csr; const size_t ts = A->type->size;
  std::vector<uint32_t> rp(c.nrows + 1), col(c.nnz);
  A->hx.resize(c.nnz * ts);
  GRB_HIP(hipMemcpyAsync(rp.data(), c.rowptr.p, (c.nrows + 1) * 4ull, hipMemcpyDeviceToHost, stream()));
  if (c.nnz) {
    GRB_HIP(hipMemcpyAsync(col.data(), c.col.p, c.nnz * 4ull, hipMemcpyDeviceToHost, stream()));
    GRB_HIP(hipMemcpyAsync(A->hx.data(), c.val.p, c.nnz * ts, hipMemcpyDeviceToHost, stream()));
  }
  GRB_HIP(hipStreamSynchronize(stream()));
  A->hi.resize(c.nnz); A->hj.resize(c.nnz);
  for (uint32_t r = 0; r < c.nrows; r++)
    for (uint32_t p = rp[r]; p < rp[r + 1]; p++) { A->hi[p] = r; A->hj[p] = col[p]; }
  A->host_valid = true;
}

void mat_to_device(GrB_Matrix A) {
  if (A->dev_valid) return;
  if (mat_bitmap_only(A)) { mat_bitmap_to_csr(A); return; }
  if (A->type->code >= T_FC32) fail(GrB_DOMAIN_MISMATCH, "complex matrices are host-side containers here: no device arithmetic on them");
  need_device();
  mat_host_assemble(A);
  if (A->nrows > GRB_DIM_DEVICE_MAX || A->ncols > GRB_DIM_DEVICE_MAX)
    fail(GrB_INSUFFICIENT_SPACE, "matrix dimension exceeds the 32-bit index range of the HBM CSR layout");
  const uint64_t nnz = A->hi.size(); const size_t ts = A->type->size;
  if (nnz > 0xFFFFFFF0ull) fail(GrB_INSUFFICIENT_SPACE, "more than 2^32 entries in one device matrix");
  DevCSR& c = A->csr; c.clear();
  c.nrows = (uint32_t)A->nrows; c.ncols = (uint32_t)A->ncols; c.nnz = nnz;
  std::vector<uint32_t> rp((size_t)c.nrows + 1, 0), col(nnz);
  for (uint64_t p = 0; p < nnz; p++) { rp[A->hi[p] + 1]++; col[p] = (uint32_t)A->hj[p]; }
  for (uint32_t r = 0; r < c.nrows; r++) rp[r + 1] += rp[r];
  c.rowptr.alloc(((size_t)c.nrows + 1) * 4); c.col.alloc(nnz * 4); c.val.alloc(nnz * ts);
  GRB_HIP(hipMemcpyAsync(c.rowptr.p, rp.data(), rp.size() * 4, hipMemcpyHostToDevice, stream()));
  if (nnz) {
    GRB_HIP(hipMemcpyAsync(c.col.p, col.data(), nnz * 4, hipMemcpyHostToDevice, stream()));
    GRB_HIP(hipMemcpyAsync(c.val.p, A->hx.data(), nnz * ts, hipMemcpyHostToDevice, stream()));
  }
  GRB_HIP(hipStreamSynchronize(stream()));
  c.valid = true; A->dev_valid = true;
}

uint64_t mat_nvals(GrB_Matrix A) {
  if (A->iso_full) { const unsigned __int128 t = (unsigned __int128)A->nrows * A->ncols; return t > UINT64_MAX ? UINT64_MAX : (uint64_t)t; }
  if (A->host_valid) { mat_host_assemble(A); return A->hi.size(); }
  if (mat_bitmap_only(A)) return mat_bitmap_nvals(A);
  return A->csr.nnz;
}

const DevCSR& mat_csc(GrB_Matrix A) {
  mat_to_device(A);
  if (!A->csc.valid) csr_transpose(A->csr, A->type->size, A->csc);
  return A->csc;
}

// ---- vectors ---------------------------------------------------------------------------------------
void vec_host_assemble(GrB_Vector v) {
  if (v->iso_full) fail(GrB_INSUFFICIENT_SPACE, ISO_MSG);
  if (v->pending.empty()) return;
  const size_t ts = v->type->size; auto& P = v->pending;
  std::vector<uint32_t> ord(P.size()); std::iota(ord.begin(), ord.end(), 0u);
  std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return P[a].i < P[b].i; });
  std::vector<GrB_Index> ni; std::vector<uint8_t> nx;
  size_t b = 0, nb = v->hi.size(), k = 0;
  auto push_base = [&](size_t q) { ni.push_back(v->hi[q]); nx.insert(nx.end(), &v->hx[q * ts], &v->hx[q * ts] + ts); };
  while (k < ord.size()) {
    size_t e = k; while (e + 1 < ord.size() && P[ord[e + 1]].i == P[ord[k]].i) e++;
    const auto& last = P[ord[e]];
    while (b < nb && v->hi[b] < last.i) push_base(b++);
    if (b < nb && v->hi[b] == last.i) b++;
    if (!last.del) { ni.push_back(last.i); nx.insert(nx.end(), last.x, last.x + ts); }
    k = e + 1;
  }
  while (b < nb) push_base(b++);
  v->hi.swap(ni); v->hx.swap(nx); P.clear(); P.shrink_to_fit();
}
// ---- "any stored value true" of a BOOL product result (grb_spmv.hpp: SpmvCall::any_true) -----------------------------------
// One device word per thread.  Every product brings a fresh non-zero tag and its kernels store that tag, so the word never has
// to be cleared: it answers "true" for the vector that holds the tag it currently shows, and only the latest product's vector
// may ask (GrB_Vector_reduce_BOOL with LOR then reads four bytes instead of launching a kernel over the vector).
// Round 5: the summary of a BOOL result's true entries (SpmvCall::fe_host) — every workgroup of the product stores its (edge sum, entry count),
// tagged, into its own pair of page-locked HOST words, and the lookup spins until all pairs carry the product's tag: no copy, no stream
// synchronisation.
namespace {
constexpr size_t FE_HOST_PAIRS = 2048;      // (grb_spmv_kernels.hpp: FE_MAX_BLOCKS)
struct AnyTrue { DevBuf word; GrB_Vector owner = nullptr; uint32_t tag = 0; bool fe_has = false; uint64_t fe_key = 0; uint32_t fe_nblocks = 0; unsigned long long* host = nullptr; unsigned long long* host_dev = nullptr; };
thread_local AnyTrue t_any;
}
uint32_t* any_true_acquire(uint32_t* tag) {
  AnyTrue& s = t_any;
  if (!s.word.p) {
    s.word.alloc(64); GRB_HIP(hipMemsetAsync(s.word.p, 0, 64, stream()));
    void* h = nullptr; GRB_HIP(hipHostMalloc(&h, FE_HOST_PAIRS * 8, hipHostMallocMapped | hipHostMallocCoherent)); memset(h, 0, FE_HOST_PAIRS * 8); s.host = (unsigned long long*)h;      // (lives as long as the thread's pinned scratch: never freed)
    void* dp = nullptr; if (hipHostGetDevicePointer(&dp, h, 0) != hipSuccess) { (void)hipGetLastError(); dp = h; }
    s.host_dev = (unsigned long long*)dp;
    if (const char* e = getenv("GRB_MI355X_SUMMARY_TAG0")) s.tag = (uint32_t)strtoul(e, nullptr, 0);      // test hook: start the tags just below the 24-bit wrap
  }
  if ((++s.tag & 0xFFFFFFu) == 0) {                   // (the low 24 bits travel in the host words of the summary: never zero)
    s.tag++;
    // the 24-bit tags start over: a word that a product 2^24 calls ago left unread (its summary was never asked for, and no product since had as many
    // workgroups) must not pass for the coming product's — once per 16.7 million products, wait for the stream and wipe the words
    GRB_HIP(hipStreamSynchronize(stream())); memset(s.host, 0, FE_HOST_PAIRS * 8);
  }
  s.owner = nullptr; s.fe_has = false; *tag = s.tag;
  return s.word.as<uint32_t>();
}
unsigned long long* fe_summary_host() { return t_any.host_dev; }      // (after any_true_acquire)
void any_true_written(GrB_Vector w, const void* key, uint32_t tag, uint64_t fe_key, uint32_t fe_nblocks) {
  AnyTrue& s = t_any;
  s.owner = w; s.fe_has = fe_key != 0 && w != nullptr && fe_nblocks > 0 && fe_nblocks <= FE_HOST_PAIRS; s.fe_key = fe_key; s.fe_nblocks = fe_nblocks;
  if (w) { w->lor_state = 1; w->lor_key = key; w->lor_tag = tag; }
}
bool any_true_lookup(GrB_Vector u, bool* value) {
  if (!u->lor_state || u->lazy || u->q_reads || !u->dev_valid || u->dval.p != u->lor_key) return false;
  if (u->lor_state == 1) {
    AnyTrue& s = t_any;
    if (s.owner != u || s.tag != u->lor_tag) { u->lor_state = 0; return false; }
    if (s.fe_has) {
      // every workgroup stores  tag (24 bits) | true entries, saturating (8) | edge sum, saturating (32)  into its host word: wait for all of them,
      // and clear what was read (a word is never mistaken for a later product's with the same 24-bit tag)
      volatile unsigned long long* h = s.host; const unsigned long long want = (unsigned long long)(u->lor_tag & 0xFFFFFFu);
      unsigned long long fe = 0, cnt = 0; uint32_t i = 0; const uint32_t nb = s.fe_nblocks;
      for (int round = 0; round < 2 && i < nb; round++) {
        for (uint64_t spin = 0; spin < (1ull << 22) && i < nb;) {
          const unsigned long long a = h[i];
          if ((a >> 40) == want && a != 0) { fe += a & 0xFFFFFFFFull; cnt += (a >> 32) & 0xFFull; h[i] = 0; i++; } else { spin++; __builtin_ia32_pause(); }
        }
        if (i < nb) GRB_HIP(hipStreamSynchronize(stream()));      // (far beyond any product's time: let a failed launch report itself, then look once more)
      }
      s.owner = nullptr; s.fe_has = false;
      if (i < nb) { u->lor_state = 0; return false; }
      u->lor_state = cnt ? 3 : 2;
      u->fe_lb = fe; u->fe_lb_key = s.fe_key; u->fe_lb_true = true;      // the edges leaving u's true entries (exact when the kernel counted: a lower bound is all the direction choice asks for)
    } else {
      uint32_t* pin = (uint32_t*)pinned_scratch();
      GRB_HIP(hipMemcpyAsync(pin, s.word.p, 4, hipMemcpyDeviceToHost, stream()));
      GRB_HIP(hipStreamSynchronize(stream()));
      u->lor_state = pin[0] == u->lor_tag ? 3 : 2; s.owner = nullptr;
    }
  }
  *value = u->lor_state == 3; return true;
}

void vec_invalidate_device(GrB_Vector v) { vec_overwritten(v); v->holes_zero = false; v->holes_big = false; v->lor_state = 0; v->abs_bound = -1; v->small_valid = false; v->code_valid = false; v->dev_valid = false; v->dval.reset(); v->dpres.reset(); v->dnvals = 0; v->dnvals_known = true; v->fe_lb = 0; v->fe_lb_key = 0; }
void vec_invalidate_host(GrB_Vector v) {
  vec_overwritten(v); v->holes_zero = false; v->holes_big = false; v->lor_state = 0; v->abs_bound = -1; v->small_valid = false; v->code_valid = false; v->dev_elem_ops = 0;
  v->host_valid = false; v->hi.clear(); v->hx.clear(); v->pending.clear(); v->hi.shrink_to_fit(); v->hx.shrink_to_fit();
}
void vec_to_host(GrB_Vector v) {
  vec_gate(v);
  if (v->iso_full) fail(GrB_INSUFFICIENT_SPACE, ISO_MSG);
  if (v->host_valid) { vec_host_assemble(v); return; }
  const size_t ts = v->type->size; const uint64_t n = v->n;
  std::vector<uint8_t> val(n * ts), pres(n);
  if (n) {
    GRB_HIP(hipMemcpyAsync(val.data(), v->dval.p, n * ts, hipMemcpyDeviceToHost, stream()));
    GRB_HIP(hipMemcpyAsync(pres.data(), v->dpres.p, n, hipMemcpyDeviceToHost, stream()));
    GRB_HIP(hipStreamSynchronize(stream()));
  }
  v->hi.clear(); v->hx.clear();
  for (uint64_t i = 0; i < n; i++) if (pres[i]) { v->hi.push_back(i); v->hx.insert(v->hx.end(), &val[i * ts], &val[i * ts] + ts); }
  v->host_valid = true;
}
void vec_to_device(GrB_Vector v) {
  vec_gate(v);
  if (v->dev_valid) return;
  if (v->type->code >= T_FC32) fail(GrB_DOMAIN_MISMATCH, "complex vectors are host-side containers here: no device arithmetic on them");
  need_device();
  vec_host_assemble(v);
  if (v->n > GRB_DIM_DEVICE_MAX) fail(GrB_INSUFFICIENT_SPACE, "vector length exceeds the 32-bit index range of the HBM bitmap layout");
  const size_t ts = v->type->size; const uint64_t n = v->n;
  v->dval.alloc(n * ts ? n * ts : 1); v->dpres.alloc(n ? n : 1);
  if (n && v->hi.size() * 16 <= n) {
    // few entries (an empty output vector, the one-entry frontier of a BFS): zero the bitmap in HBM and scatter the entries —
    // staging n zero bytes on the host and copying them cost 2 x 15 ms per PageRank run at R-MAT-25 and half of a BFS at R-MAT-22
    const uint32_t k = (uint32_t)v->hi.size();
    if (k <= 16 && ts <= 8) init_entries_small(k, v->hi.data(), v->hx.data(), ts, v->dval.p, v->dpres.as<uint8_t>(), n);      // zeros + the entries, one launch
    else { GRB_HIP(hipMemsetAsync(v->dval.p, 0, n * ts, stream())); GRB_HIP(hipMemsetAsync(v->dpres.p, 0, n, stream())); }
    if (k <= 16 && ts <= 8) {}
    else if (k) {
      std::vector<uint32_t> i32(k); for (uint32_t e = 0; e < k; e++) i32[e] = (uint32_t)v->hi[e];
      DevBuf di((size_t)k * 4), dx((size_t)k * ts);
      GRB_HIP(hipMemcpyAsync(di.p, i32.data(), (size_t)k * 4, hipMemcpyHostToDevice, stream()));
      GRB_HIP(hipMemcpyAsync(dx.p, v->hx.data(), (size_t)k * ts, hipMemcpyHostToDevice, stream()));
      scatter_entries(k, di.as<uint32_t>(), dx.p, ts, v->dval.p, v->dpres.as<uint8_t>());
      GRB_HIP(hipStreamSynchronize(stream()));                          // the host staging vectors go out of scope
    }
    v->dnvals = k; v->dnvals_known = true; v->dev_valid = true;
    if (k <= 64) {                                                       // the entries as a list (see small_idx)
      v->small_idx.assign(k, 0); bool truthy = true;
      for (uint32_t e = 0; e < k; e++) {
        v->small_idx[e] = (uint32_t)v->hi[e];
        // truth as the mask kernels test it — value != 0 in the vector's type: the bytes of an FP -0.0 are not all zero, the value is false
        bool nz = false;
        const uint8_t* px = &v->hx[(size_t)e * ts];
        if (v->type->code == T_FP32) { float f; memcpy(&f, px, 4); nz = f != 0.0f; }
        else if (v->type->code == T_FP64) { double f; memcpy(&f, px, 8); nz = f != 0.0; }
        else for (size_t b = 0; b < ts; b++) nz = nz || px[b] != 0;
        truthy = truthy && nz;
      }
      v->small_valid = true; v->small_truthy = truthy;
    }
    return;
  }
  std::vector<uint8_t> val(n * ts, 0), pres(n, 0);
  for (size_t k = 0; k < v->hi.size(); k++) { pres[v->hi[k]] = 1; memcpy(&val[v->hi[k] * ts], &v->hx[k * ts], ts); }
  if (n) {
    GRB_HIP(hipMemcpyAsync(v->dval.p, val.data(), n * ts, hipMemcpyHostToDevice, stream()));
    GRB_HIP(hipMemcpyAsync(v->dpres.p, pres.data(), n, hipMemcpyHostToDevice, stream()));
    GRB_HIP(hipStreamSynchronize(stream()));
  }
  v->dnvals = v->hi.size(); v->dnvals_known = true; v->dev_valid = true;
}
uint64_t vec_dev_nvals(GrB_Vector v) {
  vec_gate(v);
  if (!v->dnvals_known) { v->dnvals = count_present(v->dpres.as<uint8_t>(), v->n); v->dnvals_known = true; }
  return v->dnvals;
}
uint64_t vec_nvals(GrB_Vector v) {
  vec_gate(v);
  if (v->iso_full) return v->n;
  if (v->host_valid) { vec_host_assemble(v); return v->hi.size(); }
  return vec_dev_nvals(v);
}

// sort + combine duplicates for build(); keys are (i,j) pairs (j == 0 for vectors)
static void build_tuples(GrB_Type type, const GrB_Index* I, const GrB_Index* J, const void* X, int xcode, GrB_Index n,
                         GrB_Index nrows, GrB_Index ncols, GrB_BinaryOp dup, std::vector<GrB_Index>& hi,
                         std::vector<GrB_Index>* hj, std::vector<uint8_t>& hx) {
  const size_t ts = type->size; const size_t xs = type_size(xcode);
  for (GrB_Index k = 0; k < n; k++) {
    if (I[k] >= nrows || (J && J[k] >= ncols)) fail(GrB_INDEX_OUT_OF_BOUNDS, "build: index out of bounds");
  }
  std::vector<uint64_t> ord(n); std::iota(ord.begin(), ord.end(), 0ull);
  std::stable_sort(ord.begin(), ord.end(), [&](uint64_t a, uint64_t b) {
    if (I[a] != I[b]) return I[a] < I[b]; return J ? J[a] < J[b] : false; });
  hi.clear(); if (hj) hj->clear(); hx.clear();
  hi.reserve(n); if (hj) hj->reserve(n); hx.reserve(n * ts);
  uint8_t cur[16], nxt[16];
  for (GrB_Index k = 0; k < n;) {
    GrB_Index e = k; const uint64_t o = ord[k];
    cast_scalar(type->code, cur, xcode, (const uint8_t*)X + o * xs);
    while (e + 1 < n && I[ord[e + 1]] == I[o] && (!J || J[ord[e + 1]] == J[o])) {
      e++;
      cast_scalar(type->code, nxt, xcode, (const uint8_t*)X + ord[e] * xs);
      if (!dup) fail(GrB_INVALID_VALUE, "build: duplicate index and no dup operator");
      if (type->code >= T_FC32) fail(GrB_DOMAIN_MISMATCH, "build: combining duplicate complex entries is out of scope");
      dispatch_type(type->code, [&]<class T>() { T a, b; memcpy(&a, cur, sizeof(T)); memcpy(&b, nxt, sizeof(T));
        T z = apply_binop<T>(dup->opcode, a, b); memcpy(cur, &z, sizeof(T)); });
    }
    hi.push_back(I[o]); if (hj) hj->push_back(J[o]); hx.insert(hx.end(), cur, cur + ts);
    k = e + 1;
  }
}

}  // namespace grb

using namespace grb;

#define CHECK_MAT(A) do { if (!(A)) return GrB_NULL_POINTER; if (!check_obj(A)) return GrB_UNINITIALIZED_OBJECT; } while (0)
#define CHECK_VEC(v) CHECK_MAT(v)

extern "C" {

// =================================== Matrix ========================================================
GrB_Info GrB_Matrix_new(GrB_Matrix* A, GrB_Type type, GrB_Index nrows, GrB_Index ncols) {
  if (!A) return GrB_NULL_POINTER; *A = nullptr;
  if (!check_obj(type)) return GrB_UNINITIALIZED_OBJECT;
  if (type->code >= T_UDT) return GrB_DOMAIN_MISMATCH;    // user types: out of scope; complex: host-side containers only
  if (nrows > GXB_INDEX_MAX || ncols > GXB_INDEX_MAX) return GrB_INVALID_VALUE;
  auto* m = new (std::nothrow) GrB_Matrix_opaque(); if (!m) return GrB_OUT_OF_MEMORY;
  m->type = type; m->nrows = nrows; m->ncols = ncols; *A = m; return GrB_SUCCESS;
}
GrB_Info GrB_Matrix_free(GrB_Matrix* A) {
  if (!A || !*A) return GrB_SUCCESS;
  if (check_obj(*A)) { (*A)->magic = GRB_FREED; delete *A; }
  *A = nullptr; return GrB_SUCCESS;
}
GrB_Info GrB_Matrix_dup(GrB_Matrix* C, const GrB_Matrix A) {
  if (!C) return GrB_NULL_POINTER; CHECK_MAT(A);
  GrB_Matrix m = nullptr; GrB_Info info = GrB_Matrix_new(&m, A->type, A->nrows, A->ncols); if (info) return info;
  info = guarded(A, [&] {
    m->format = A->format; m->sparsity_control = A->sparsity_control; m->hyper_switch = A->hyper_switch;
    if (A->iso_full) { m->iso_full = true; memcpy(m->iso_val, A->iso_val, 16); }
    else if (A->host_valid) { mat_host_assemble(A); m->hi = A->hi; m->hj = A->hj; m->hx = A->hx; m->host_valid = true; }
    else if (mat_bitmap_only(A)) {                             // a batch matrix that lives as a bitmap: so does its copy
      const size_t np = (size_t)A->nrows * A->ncols, ts = A->type->size;
      m->bm.val.alloc(np * ts + 64); m->bm.pres.alloc(np + 64);
      GRB_HIP(hipMemcpyAsync(m->bm.val.p, A->bm.val.p, np * ts, hipMemcpyDeviceToDevice, stream()));
      GRB_HIP(hipMemcpyAsync(m->bm.pres.p, A->bm.pres.p, np, hipMemcpyDeviceToDevice, stream()));
      m->bm.valid = true; m->bm.nvals = A->bm.nvals; m->bm.nvals_known = A->bm.nvals_known; m->host_valid = false; m->dev_valid = false;
    } else {
      const DevCSR& s = A->csr; DevCSR& d = m->csr; const size_t ts = A->type->size;
      d.nrows = s.nrows; d.ncols = s.ncols; d.nnz = s.nnz;
      d.rowptr.alloc(((size_t)s.nrows + 1) * 4); d.col.alloc(s.nnz * 4); d.val.alloc(s.nnz * ts);
      GRB_HIP(hipMemcpyAsync(d.rowptr.p, s.rowptr.p, ((size_t)s.nrows + 1) * 4, hipMemcpyDeviceToDevice, stream()));
      if (s.nnz) { GRB_HIP(hipMemcpyAsync(d.col.p, s.col.p, s.nnz * 4, hipMemcpyDeviceToDevice, stream()));
                   GRB_HIP(hipMemcpyAsync(d.val.p, s.val.p, s.nnz * ts, hipMemcpyDeviceToDevice, stream())); }
      d.valid = true; m->dev_valid = true; m->host_valid = false;
    }
  });
  if (info) { GrB_Matrix_free(&m); return info; }
  *C = m; return GrB_SUCCESS;
}
GrB_Info GrB_Matrix_clear(GrB_Matrix A) {
  CHECK_MAT(A); A->hi.clear(); A->hj.clear(); A->hx.clear(); A->pending.clear(); A->host_valid = true; A->iso_full = false; mat_invalidate_device(A); return GrB_SUCCESS;
}
GrB_Info GrB_Matrix_nrows(GrB_Index* n, const GrB_Matrix A) { if (!n) return GrB_NULL_POINTER; CHECK_MAT(A); *n = A->nrows; return GrB_SUCCESS; }
GrB_Info GrB_Matrix_ncols(GrB_Index* n, const GrB_Matrix A) { if (!n) return GrB_NULL_POINTER; CHECK_MAT(A); *n = A->ncols; return GrB_SUCCESS; }
GrB_Info GrB_Matrix_nvals(GrB_Index* n, const GrB_Matrix A) {
  if (!n) return GrB_NULL_POINTER; CHECK_MAT(A); return guarded(A, [&] { *n = mat_nvals(A); });
}
GrB_Info GrB_Matrix_wait(GrB_Matrix* A) {
  if (!A) return GrB_NULL_POINTER; CHECK_MAT(*A);
  return guarded(*A, [&] { if ((*A)->host_valid && !(*A)->iso_full) mat_host_assemble(*A); if (device_ok()) GRB_HIP(hipStreamSynchronize(stream())); });
}
// the reference asks `self` for the message even when the failing object was the output (pygraphblas/matrix.py:43-51)
GrB_Info GrB_Matrix_error(const char** s, const GrB_Matrix A) { if (!s) return GrB_NULL_POINTER; CHECK_MAT(A); *s = A->err.empty() ? g_last_error.c_str() : A->err.c_str(); return GrB_SUCCESS; }
GrB_Info GxB_Matrix_type(GrB_Type* t, const GrB_Matrix A) { if (!t) return GrB_NULL_POINTER; CHECK_MAT(A); *t = A->type; return GrB_SUCCESS; }
GrB_Info GrB_Matrix_resize(GrB_Matrix A, GrB_Index nr, GrB_Index nc) {
  CHECK_MAT(A); if (nr > GXB_INDEX_MAX || nc > GXB_INDEX_MAX) return GrB_INVALID_VALUE;
  return guarded(A, [&] {
    mat_to_host(A); const size_t ts = A->type->size; size_t w = 0;
    for (size_t k = 0; k < A->hi.size(); k++) if (A->hi[k] < nr && A->hj[k] < nc) {
      if (w != k) { A->hi[w] = A->hi[k]; A->hj[w] = A->hj[k]; memmove(&A->hx[w * ts], &A->hx[k * ts], ts); } w++; }
    A->hi.resize(w); A->hj.resize(w); A->hx.resize(w * ts); A->nrows = nr; A->ncols = nc; mat_invalidate_device(A);
  });
}
GrB_Info GrB_Matrix_removeElement(GrB_Matrix A, GrB_Index i, GrB_Index j) {
  CHECK_MAT(A); if (i >= A->nrows || j >= A->ncols) return GrB_INVALID_INDEX;
  return guarded(A, [&] { mat_to_host(A); GrB_Matrix_opaque::Pending p{i, j, true, {0}}; A->pending.push_back(p); mat_invalidate_device(A); });
}
GrB_Info GxB_Matrix_Option_set(GrB_Matrix A, int field, ...) {
  CHECK_MAT(A); va_list ap; va_start(ap, field); GrB_Info info = GrB_SUCCESS;
  switch (field) {
    case 1: A->format = va_arg(ap, int); break;
    case 0: A->hyper_switch = va_arg(ap, double); break;
    case 32: A->sparsity_control = va_arg(ap, int); break;
    case 34: (void)va_arg(ap, double); break;
    default: info = GrB_INVALID_VALUE;
  }
  va_end(ap); return info;
}
GrB_Info GxB_Matrix_Option_get(GrB_Matrix A, int field, ...) {
  CHECK_MAT(A); va_list ap; va_start(ap, field); GrB_Info info = GrB_SUCCESS;
  switch (field) {
    case 1: { int* p = va_arg(ap, int*); if (p) *p = A->format; break; }
    case 0: { double* p = va_arg(ap, double*); if (p) *p = A->hyper_switch; break; }
    case 32: { int* p = va_arg(ap, int*); if (p) *p = A->sparsity_control; break; }
    case 33: { int* p = va_arg(ap, int*); if (p) *p = A->iso_full ? 8 : (A->nrows > GRB_DIM_DEVICE_MAX || A->hyper_switch >= 1.0 || mat_nvals(A) == 0) ? 1 : 2; break; }  // what SuiteSparse would report: hypersparse for huge or empty matrices and under hyper_switch = GxB_ALWAYS_HYPER (a stored option here), else sparse
    case 34: { double* p = va_arg(ap, double*); if (p) *p = 0.04; break; }
    default: info = GrB_INVALID_VALUE;
  }
  va_end(ap); return info;
}
GrB_Info GxB_Matrix_memoryUsage(size_t* size, const GrB_Matrix A) {
  if (!size) return GrB_NULL_POINTER; CHECK_MAT(A);
  *size = sizeof(*A) + A->hi.capacity() * 16 + A->hx.capacity() + A->csr.rowptr.bytes + A->csr.col.bytes + A->csr.val.bytes +
          A->csc.rowptr.bytes + A->csc.col.bytes + A->csc.val.bytes; return GrB_SUCCESS;
}

static GrB_Info mat_build(GrB_Matrix C, const GrB_Index* I, const GrB_Index* J, const void* X, int xcode, GrB_Index n, GrB_BinaryOp dup) {
  CHECK_MAT(C); if (n && (!I || !J || !X)) return GrB_NULL_POINTER;
  if (dup && !check_obj(dup)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(C, [&] {
    if (mat_nvals(C) != 0) fail(GrB_OUTPUT_NOT_EMPTY, "build: output already has entries");
    build_tuples(C->type, I, J, X, xcode, n, C->nrows, C->ncols, dup, C->hi, &C->hj, C->hx);
    C->host_valid = true; mat_invalidate_device(C);
  });
}
// a matrix that lives in HBM only and is large (`paths[i, source] = 1` on a dense ns x n batch, gap/bcmark.py:23: bringing 134 MB to
// the host mirror for one value was a third of the algorithm): the entry is looked up on the device; an existing one is read or
// overwritten in place (what changes with the values — the transpose cache, a kernel-X plan's copy — is dropped)
// (a caller that touches many elements one by one is better served by the host mirror: one transfer, then no round trips)
static bool mat_device_only(GrB_Matrix A) { return A->dev_elem_ops++ < 32 && device_ok() && A->dev_valid && !A->host_valid && A->pending.empty() && !A->iso_full && A->csr.valid && A->csr.nnz >= (1u << 16) && A->type->code < T_FC32; }
static GrB_Info mat_set(GrB_Matrix C, const void* x, int xcode, GrB_Index i, GrB_Index j) {
  CHECK_MAT(C); if (i >= C->nrows || j >= C->ncols) return GrB_INVALID_INDEX;
  return guarded(C, [&] {
    if (mat_device_only(C)) {
      const uint64_t pos = csr_find_entry(C->csr, (uint32_t)i, (uint32_t)j);
      if (pos != ~0ull) {
        uint8_t* pin = (uint8_t*)pinned_scratch() + 64; cast_scalar(C->type->code, pin, xcode, x);
        GRB_HIP(hipMemcpyAsync((uint8_t*)C->csr.val.p + pos * C->type->size, pin, C->type->size, hipMemcpyHostToDevice, stream()));
        GRB_HIP(hipStreamSynchronize(stream()));
 C->csc.clear(); C->csr.xcd.reset(); C->csr.range_state = 0; C->bm.clear();
        C->csr.heads_valid = false;      // the row heads carry the BOOL value of a row's first four entries
        return;
      }
    }
    if (C->iso_full) fail(GrB_INSUFFICIENT_SPACE, ISO_MSG);
    if (!C->host_valid) mat_to_host(C);
    GrB_Matrix_opaque::Pending p{i, j, false, {0}}; cast_scalar(C->type->code, p.x, xcode, x);
    C->pending.push_back(p); mat_invalidate_device(C);
  });
}
static size_t mat_find(GrB_Matrix A, GrB_Index i, GrB_Index j) {   // index into host tuples or SIZE_MAX
  size_t lo = 0, hi = A->hi.size();
  while (lo < hi) { size_t m = (lo + hi) / 2; if (A->hi[m] < i || (A->hi[m] == i && A->hj[m] < j)) lo = m + 1; else hi = m; }
  return (lo < A->hi.size() && A->hi[lo] == i && A->hj[lo] == j) ? lo : SIZE_MAX;
}
static GrB_Info mat_get(void* x, int xcode, GrB_Matrix A, GrB_Index i, GrB_Index j) {
  if (!x) return GrB_NULL_POINTER; CHECK_MAT(A); if (i >= A->nrows || j >= A->ncols) return GrB_INVALID_INDEX;
  GrB_Info r = GrB_SUCCESS;
  GrB_Info info = guarded(A, [&] {
    if (A->iso_full) { cast_scalar(xcode, x, A->type->code, A->iso_val); return; }
    if (mat_device_only(A)) {
      const uint64_t pos = csr_find_entry(A->csr, (uint32_t)i, (uint32_t)j);
      if (pos == ~0ull) { r = GrB_NO_VALUE; return; }
      uint8_t* pin = (uint8_t*)pinned_scratch() + 64;
      GRB_HIP(hipMemcpyAsync(pin, (const uint8_t*)A->csr.val.p + pos * A->type->size, A->type->size, hipMemcpyDeviceToHost, stream())); GRB_HIP(hipStreamSynchronize(stream()));
      cast_scalar(xcode, x, A->type->code, pin); return;
    }
    mat_to_host(A); size_t k = mat_find(A, i, j);
    if (k == SIZE_MAX) r = GrB_NO_VALUE; else cast_scalar(xcode, x, A->type->code, &A->hx[k * A->type->size]); });
  return info ? info : r;
}
static GrB_Info mat_tuples(GrB_Index* I, GrB_Index* J, void* X, int xcode, GrB_Index* n, GrB_Matrix A) {
  if (!n) return GrB_NULL_POINTER; CHECK_MAT(A);
  return guarded(A, [&] {
    mat_to_host(A); const GrB_Index nv = A->hi.size();
    if ((I || J || X) && *n < nv) fail(GrB_INSUFFICIENT_SPACE, "extractTuples: output arrays too small");
    const size_t ts = A->type->size, xs = type_size(xcode);
    for (GrB_Index k = 0; k < nv; k++) {
      if (I) I[k] = A->hi[k]; if (J) J[k] = A->hj[k];
      if (X) cast_scalar(xcode, (uint8_t*)X + k * xs, A->type->code, &A->hx[k * ts]);
    }
    *n = nv;
  });
}

// =================================== Vector ========================================================
GrB_Info GrB_Vector_new(GrB_Vector* v, GrB_Type type, GrB_Index n) {
  if (!v) return GrB_NULL_POINTER; *v = nullptr;
  if (!check_obj(type)) return GrB_UNINITIALIZED_OBJECT;
  if (type->code >= T_UDT) return GrB_DOMAIN_MISMATCH;
  if (n > GXB_INDEX_MAX) return GrB_INVALID_VALUE;
  auto* w = new (std::nothrow) GrB_Vector_opaque(); if (!w) return GrB_OUT_OF_MEMORY;
  w->type = type; w->n = n; *v = w; return GrB_SUCCESS;
}
GrB_Info GrB_Vector_free(GrB_Vector* v) {
  if (!v || !*v) return GrB_SUCCESS;
  if (check_obj(*v)) {
    if ((*v)->lazy | (*v)->q_reads) { GrB_Vector x = *v; (void)guarded(x, [&] { vec_overwritten(x); }); }   // deferred work that reads it runs first; work that only wrote it is dropped
    (*v)->magic = GRB_FREED; delete *v; }
  *v = nullptr; return GrB_SUCCESS;
}
GrB_Info GrB_Vector_dup(GrB_Vector* w, const GrB_Vector u) {
  if (!w) return GrB_NULL_POINTER; CHECK_VEC(u);
  GrB_Vector r = nullptr; GrB_Info info = GrB_Vector_new(&r, u->type, u->n); if (info) return info;
  info = guarded(u, [&] {
    vec_gate(u);
    if (u->iso_full) { r->iso_full = true; memcpy(r->iso_val, u->iso_val, 16); }
    else if (u->host_valid) { vec_host_assemble(u); r->hi = u->hi; r->hx = u->hx; }
    else {
      const size_t ts = u->type->size;
      r->dval.alloc(u->n * ts ? u->n * ts : 1); r->dpres.alloc(u->n ? u->n : 1);
      if (u->n) dev_copy2(r->dval.p, u->dval.p, u->n * ts, r->dpres.p, u->dpres.p, u->n);
      r->dnvals = u->dnvals; r->dnvals_known = u->dnvals_known; r->dev_valid = true; r->host_valid = false;
    }
  });
  if (info) { GrB_Vector_free(&r); return info; }
  *w = r; return GrB_SUCCESS;
}
GrB_Info GrB_Vector_clear(GrB_Vector v) { CHECK_VEC(v); if (v->lazy | v->q_reads) { GrB_Info e = guarded(v, [&] { vec_overwritten(v); }); if (e) return e; } v->hi.clear(); v->hx.clear(); v->pending.clear(); v->host_valid = true; v->iso_full = false; vec_invalidate_device(v); return GrB_SUCCESS; }
GrB_Info GrB_Vector_size(GrB_Index* n, const GrB_Vector v) { if (!n) return GrB_NULL_POINTER; CHECK_VEC(v); *n = v->n; return GrB_SUCCESS; }
GrB_Info GrB_Vector_nvals(GrB_Index* n, const GrB_Vector v) { if (!n) return GrB_NULL_POINTER; CHECK_VEC(v); return guarded(v, [&] { *n = vec_nvals(v); }); }
GrB_Info GrB_Vector_wait(GrB_Vector* v) {
  if (!v) return GrB_NULL_POINTER; CHECK_VEC(*v);
  return guarded(*v, [&] { vec_gate(*v); if ((*v)->host_valid) vec_host_assemble(*v); if (device_ok()) GRB_HIP(hipStreamSynchronize(stream())); });
}
GrB_Info GrB_Vector_error(const char** s, const GrB_Vector v) { if (!s) return GrB_NULL_POINTER; CHECK_VEC(v); *s = v->err.empty() ? g_last_error.c_str() : v->err.c_str(); return GrB_SUCCESS; }
GrB_Info GxB_Vector_type(GrB_Type* t, const GrB_Vector v) { if (!t) return GrB_NULL_POINTER; CHECK_VEC(v); *t = v->type; return GrB_SUCCESS; }
GrB_Info GrB_Vector_resize(GrB_Vector v, GrB_Index n) {
  CHECK_VEC(v); if (n > GXB_INDEX_MAX) return GrB_INVALID_VALUE;
  return guarded(v, [&] { vec_to_host(v); const size_t ts = v->type->size; size_t w = 0;
    while (w < v->hi.size() && v->hi[w] < n) w++;
    v->hi.resize(w); v->hx.resize(w * ts); v->n = n; vec_invalidate_device(v); });
}
GrB_Info GrB_Vector_removeElement(GrB_Vector v, GrB_Index i) {
  CHECK_VEC(v); if (i >= v->n) return GrB_INVALID_INDEX;
  return guarded(v, [&] { vec_to_host(v); GrB_Vector_opaque::Pending p{i, true, {0}}; v->pending.push_back(p); vec_invalidate_device(v); });
}
GrB_Info GxB_Vector_Option_set(GrB_Vector v, int field, ...) {
  CHECK_VEC(v); va_list ap; va_start(ap, field); GrB_Info info = GrB_SUCCESS;
  switch (field) { case 32: v->sparsity_control = va_arg(ap, int); break; case 34: (void)va_arg(ap, double); break;
                   case 1: (void)va_arg(ap, int); break; case 0: (void)va_arg(ap, double); break; default: info = GrB_INVALID_VALUE; }
  va_end(ap); return info;
}
GrB_Info GxB_Vector_Option_get(GrB_Vector v, int field, ...) {
  CHECK_VEC(v); va_list ap; va_start(ap, field); GrB_Info info = GrB_SUCCESS;
  switch (field) {
    case 32: { int* p = va_arg(ap, int*); if (p) *p = v->sparsity_control; break; }
    case 33: { int* p = va_arg(ap, int*); (void)guarded(v, [&] { vec_gate(v); }); if (p) *p = v->dev_valid && !v->host_valid ? (vec_dev_nvals(v) == v->n ? 8 : 4) : 2; break; }
    case 1: { int* p = va_arg(ap, int*); if (p) *p = 1; break; }
    case 34: { double* p = va_arg(ap, double*); if (p) *p = 0.04; break; }
    default: info = GrB_INVALID_VALUE;
  }
  va_end(ap); return info;
}
GrB_Info GxB_Vector_memoryUsage(size_t* size, const GrB_Vector v) {
  if (!size) return GrB_NULL_POINTER; CHECK_VEC(v); *size = sizeof(*v) + v->hi.capacity() * 8 + v->hx.capacity() + v->dval.bytes + v->dpres.bytes; return GrB_SUCCESS;
}
static GrB_Info vec_build(GrB_Vector w, const GrB_Index* I, const void* X, int xcode, GrB_Index n, GrB_BinaryOp dup) {
  CHECK_VEC(w); if (n && (!I || !X)) return GrB_NULL_POINTER; if (dup && !check_obj(dup)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(w, [&] {
    if (vec_nvals(w) != 0) fail(GrB_OUTPUT_NOT_EMPTY, "build: output already has entries");
    build_tuples(w->type, I, nullptr, X, xcode, n, w->n, 1, dup, w->hi, nullptr, w->hx);
    w->host_valid = true; vec_invalidate_device(w);
  });
}
static GrB_Info vec_set(GrB_Vector w, const void* x, int xcode, GrB_Index i) {
  CHECK_VEC(w); if (i >= w->n) return GrB_INVALID_INDEX;
  return guarded(w, [&] { if (w->iso_full) fail(GrB_INSUFFICIENT_SPACE, ISO_MSG); if (!w->host_valid) vec_to_host(w);
    GrB_Vector_opaque::Pending p{i, false, {0}}; cast_scalar(w->type->code, p.x, xcode, x); w->pending.push_back(p); vec_invalidate_device(w); });
}
static GrB_Info vec_get(void* x, int xcode, GrB_Vector v, GrB_Index i) {
  if (!x) return GrB_NULL_POINTER; CHECK_VEC(v); if (i >= v->n) return GrB_INVALID_INDEX;
  GrB_Info r = GrB_SUCCESS;
  GrB_Info info = guarded(v, [&] {
    if (v->iso_full) { cast_scalar(xcode, x, v->type->code, v->iso_val); return; }
    vec_gate(v);
    // a large vector that lives in HBM only (`r[vertex]` after a PageRank): the presence byte and the value come over alone — a few
    // dozen reads in a row, then the host mirror takes over (one transfer, no more round trips)
    if (device_ok() && v->dev_valid && !v->host_valid && v->pending.empty() && v->n >= (1u << 16) && v->type->code < T_FC32 && v->dev_elem_ops++ < 32) {
      uint8_t* pin = (uint8_t*)pinned_scratch() + 64; const size_t ts = v->type->size;
      GRB_HIP(hipMemcpyAsync(pin, v->dpres.as<uint8_t>() + i, 1, hipMemcpyDeviceToHost, stream()));
      GRB_HIP(hipMemcpyAsync(pin + 16, (const uint8_t*)v->dval.p + i * ts, ts, hipMemcpyDeviceToHost, stream()));
      GRB_HIP(hipStreamSynchronize(stream()));
      if (!pin[0]) r = GrB_NO_VALUE; else cast_scalar(xcode, x, v->type->code, pin + 16);
      return;
    }
    vec_to_host(v);
    auto it = std::lower_bound(v->hi.begin(), v->hi.end(), i);
    if (it == v->hi.end() || *it != i) r = GrB_NO_VALUE;
    else cast_scalar(xcode, x, v->type->code, &v->hx[(it - v->hi.begin()) * v->type->size]); });
  return info ? info : r;
}
static GrB_Info vec_tuples(GrB_Index* I, void* X, int xcode, GrB_Index* n, GrB_Vector v) {
  if (!n) return GrB_NULL_POINTER; CHECK_VEC(v);
  return guarded(v, [&] {
    vec_to_host(v); const GrB_Index nv = v->hi.size();
    if ((I || X) && *n < nv) fail(GrB_INSUFFICIENT_SPACE, "extractTuples: output arrays too small");
    const size_t ts = v->type->size, xs = type_size(xcode);
    for (GrB_Index k = 0; k < nv; k++) { if (I) I[k] = v->hi[k]; if (X) cast_scalar(xcode, (uint8_t*)X + k * xs, v->type->code, &v->hx[k * ts]); }
    *n = nv;
  });
}

// =================================== Scalar ========================================================
GrB_Info GxB_Scalar_new(GxB_Scalar* s, GrB_Type type) {
  if (!s) return GrB_NULL_POINTER; *s = nullptr; if (!check_obj(type)) return GrB_UNINITIALIZED_OBJECT;
  if (type->code >= T_UDT) return GrB_DOMAIN_MISMATCH;
  auto* r = new (std::nothrow) GxB_Scalar_opaque(); if (!r) return GrB_OUT_OF_MEMORY; r->type = type; *s = r; return GrB_SUCCESS;
}
GrB_Info GxB_Scalar_dup(GxB_Scalar* s, const GxB_Scalar t) {
  if (!s) return GrB_NULL_POINTER; CHECK_MAT(t); auto* r = new (std::nothrow) GxB_Scalar_opaque(); if (!r) return GrB_OUT_OF_MEMORY;
  r->type = t->type; r->has = t->has; memcpy(r->x, t->x, 16); *s = r; return GrB_SUCCESS;
}
GrB_Info GxB_Scalar_clear(GxB_Scalar s) { CHECK_MAT(s); s->has = false; return GrB_SUCCESS; }
GrB_Info GxB_Scalar_free(GxB_Scalar* s) { if (!s || !*s) return GrB_SUCCESS; if (check_obj(*s)) { (*s)->magic = GRB_FREED; delete *s; } *s = nullptr; return GrB_SUCCESS; }
GrB_Info GxB_Scalar_nvals(GrB_Index* n, const GxB_Scalar s) { if (!n) return GrB_NULL_POINTER; CHECK_MAT(s); *n = s->has ? 1 : 0; return GrB_SUCCESS; }
GrB_Info GxB_Scalar_wait(GxB_Scalar* s) { if (!s) return GrB_NULL_POINTER; CHECK_MAT(*s); return GrB_SUCCESS; }
GrB_Info GxB_Scalar_type(GrB_Type* t, const GxB_Scalar s) { if (!t) return GrB_NULL_POINTER; CHECK_MAT(s); *t = s->type; return GrB_SUCCESS; }
static GrB_Info scalar_set(GxB_Scalar s, const void* x, int xcode) { CHECK_MAT(s); cast_scalar(s->type->code, s->x, xcode, x); s->has = true; return GrB_SUCCESS; }
static GrB_Info scalar_get(void* x, int xcode, GxB_Scalar s) { if (!x) return GrB_NULL_POINTER; CHECK_MAT(s); if (!s->has) return GrB_NO_VALUE; cast_scalar(xcode, x, s->type->code, s->x); return GrB_SUCCESS; }

// ---- typed families ---------------------------------------------------------------------------------
#define GRB_TYPED_CONTAINER(SUF, CT, CODE) \
  GrB_Info GrB_Matrix_build_##SUF(GrB_Matrix C, const GrB_Index* I, const GrB_Index* J, const CT* X, GrB_Index n, const GrB_BinaryOp dup) { return mat_build(C, I, J, X, CODE, n, dup); } \
  GrB_Info GrB_Matrix_setElement_##SUF(GrB_Matrix C, CT x, GrB_Index i, GrB_Index j) { return mat_set(C, &x, CODE, i, j); } \
  GrB_Info GrB_Matrix_extractElement_##SUF(CT* x, const GrB_Matrix A, GrB_Index i, GrB_Index j) { return mat_get(x, CODE, A, i, j); } \
  GrB_Info GrB_Matrix_extractTuples_##SUF(GrB_Index* I, GrB_Index* J, CT* X, GrB_Index* n, const GrB_Matrix A) { return mat_tuples(I, J, X, CODE, n, A); } \
  GrB_Info GrB_Vector_build_##SUF(GrB_Vector w, const GrB_Index* I, const CT* X, GrB_Index n, const GrB_BinaryOp dup) { return vec_build(w, I, X, CODE, n, dup); } \
  GrB_Info GrB_Vector_setElement_##SUF(GrB_Vector w, CT x, GrB_Index i) { return vec_set(w, &x, CODE, i); } \
  GrB_Info GrB_Vector_extractElement_##SUF(CT* x, const GrB_Vector v, GrB_Index i) { return vec_get(x, CODE, v, i); } \
  GrB_Info GrB_Vector_extractTuples_##SUF(GrB_Index* I, CT* X, GrB_Index* n, const GrB_Vector v) { return vec_tuples(I, X, CODE, n, v); } \
  GrB_Info GxB_Scalar_setElement_##SUF(GxB_Scalar s, CT x) { return scalar_set(s, &x, CODE); } \
  GrB_Info GxB_Scalar_extractElement_##SUF(CT* x, const GxB_Scalar s) { return scalar_get(x, CODE, s); }
GRB_TYPED_CONTAINER(BOOL, bool, T_BOOL) GRB_TYPED_CONTAINER(INT8, int8_t, T_INT8) GRB_TYPED_CONTAINER(UINT8, uint8_t, T_UINT8)
GRB_TYPED_CONTAINER(INT16, int16_t, T_INT16) GRB_TYPED_CONTAINER(UINT16, uint16_t, T_UINT16) GRB_TYPED_CONTAINER(INT32, int32_t, T_INT32)
GRB_TYPED_CONTAINER(UINT32, uint32_t, T_UINT32) GRB_TYPED_CONTAINER(INT64, int64_t, T_INT64) GRB_TYPED_CONTAINER(UINT64, uint64_t, T_UINT64)
GRB_TYPED_CONTAINER(FP32, float, T_FP32) GRB_TYPED_CONTAINER(FP64, double, T_FP64)

// complex entries: stored, set, read and listed on the host mirror (the reference's type registry, from_lists with
// complex values and Matrix.dense(FC64) need that much: tests/test_matrix.py:56-57,853); arithmetic on them is not offered
typedef struct { float re, im; } GxB_FC32_t;
typedef struct { double re, im; } GxB_FC64_t;
#define GRB_COMPLEX_CONTAINER(SUF, CT, CODE) \
  GrB_Info GxB_Matrix_build_##SUF(GrB_Matrix C, const GrB_Index* I, const GrB_Index* J, const CT* X, GrB_Index n, const GrB_BinaryOp dup) { return mat_build(C, I, J, X, CODE, n, dup); } \
  GrB_Info GxB_Matrix_setElement_##SUF(GrB_Matrix C, CT x, GrB_Index i, GrB_Index j) { return mat_set(C, &x, CODE, i, j); } \
  GrB_Info GxB_Matrix_extractElement_##SUF(CT* x, const GrB_Matrix A, GrB_Index i, GrB_Index j) { return mat_get(x, CODE, A, i, j); } \
  GrB_Info GxB_Matrix_extractTuples_##SUF(GrB_Index* I, GrB_Index* J, CT* X, GrB_Index* n, const GrB_Matrix A) { return mat_tuples(I, J, X, CODE, n, A); } \
  GrB_Info GxB_Vector_build_##SUF(GrB_Vector w, const GrB_Index* I, const CT* X, GrB_Index n, const GrB_BinaryOp dup) { return vec_build(w, I, X, CODE, n, dup); } \
  GrB_Info GxB_Vector_setElement_##SUF(GrB_Vector w, CT x, GrB_Index i) { return vec_set(w, &x, CODE, i); } \
  GrB_Info GxB_Vector_extractElement_##SUF(CT* x, const GrB_Vector v, GrB_Index i) { return vec_get(x, CODE, v, i); } \
  GrB_Info GxB_Vector_extractTuples_##SUF(GrB_Index* I, CT* X, GrB_Index* n, const GrB_Vector v) { return vec_tuples(I, X, CODE, n, v); } \
  GrB_Info GxB_Scalar_setElement_##SUF(GxB_Scalar s, CT x) { return scalar_set(s, &x, CODE); } \
  GrB_Info GxB_Scalar_extractElement_##SUF(CT* x, const GxB_Scalar s) { return scalar_get(x, CODE, s); }
GRB_COMPLEX_CONTAINER(FC32, GxB_FC32_t, T_FC32) GRB_COMPLEX_CONTAINER(FC64, GxB_FC64_t, T_FC64)

// =================================== bulk / device import-export ===================================
GrB_Info GrBX_Matrix_import_CSR(GrB_Matrix* A, GrB_Type type, GrB_Index nrows, GrB_Index ncols, GrB_Index nvals,
                                const uint32_t* rowptr, const uint32_t* colidx, const void* values, int location) {
  if (!A || !rowptr || (nvals && (!colidx || !values))) return GrB_NULL_POINTER;
  GrB_Matrix m = nullptr; GrB_Info info = GrB_Matrix_new(&m, type, nrows, ncols); if (info) return info;
  info = guarded(m, [&] {
    need_device();
    if (nrows > GRB_DIM_DEVICE_MAX || ncols > GRB_DIM_DEVICE_MAX || nvals > 0xFFFFFFF0ull) fail(GrB_INSUFFICIENT_SPACE, "import_CSR: 32-bit index range exceeded");
    DevCSR& c = m->csr; const size_t ts = type->size;
    c.nrows = (uint32_t)nrows; c.ncols = (uint32_t)ncols; c.nnz = nvals;
    c.rowptr.alloc((nrows + 1) * 4); c.col.alloc(nvals * 4); c.val.alloc(nvals * ts);
    hipMemcpyKind k = location ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    GRB_HIP(hipMemcpyAsync(c.rowptr.p, rowptr, (nrows + 1) * 4, k, stream()));
    if (nvals) { GRB_HIP(hipMemcpyAsync(c.col.p, colidx, nvals * 4, k, stream())); GRB_HIP(hipMemcpyAsync(c.val.p, values, nvals * ts, k, stream())); }
    GRB_HIP(hipStreamSynchronize(stream()));
    c.valid = true; m->dev_valid = true; m->host_valid = false;
  });
  if (info) { GrB_Matrix_free(&m); return info; }
  *A = m; return GrB_SUCCESS;
}
GrB_Info GrBX_Matrix_export_CSR(const GrB_Matrix A, uint32_t* rowptr, uint32_t* colidx, void* values, int location) {
  CHECK_MAT(A);
  return guarded(A, [&] {
    mat_to_device(A); const DevCSR& c = A->csr; const size_t ts = A->type->size;
    hipMemcpyKind k = location ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (rowptr) GRB_HIP(hipMemcpyAsync(rowptr, c.rowptr.p, ((size_t)c.nrows + 1) * 4, k, stream()));
    if (colidx && c.nnz) GRB_HIP(hipMemcpyAsync(colidx, c.col.p, c.nnz * 4, k, stream()));
    if (values && c.nnz) GRB_HIP(hipMemcpyAsync(values, c.val.p, c.nnz * ts, k, stream()));
    GRB_HIP(hipStreamSynchronize(stream()));
  });
}
GrB_Info GrBX_Vector_import_Bitmap(GrB_Vector* v, GrB_Type type, GrB_Index n, const void* values, const uint8_t* present, int location) {
  if (!v || (n && !values)) return GrB_NULL_POINTER;
  GrB_Vector w = nullptr; GrB_Info info = GrB_Vector_new(&w, type, n); if (info) return info;
  info = guarded(w, [&] {
    need_device(); if (n > GRB_DIM_DEVICE_MAX) fail(GrB_INSUFFICIENT_SPACE, "import: vector too long");
    const size_t ts = type->size; hipMemcpyKind k = location ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    w->dval.alloc(n * ts ? n * ts : 1); w->dpres.alloc(n ? n : 1);
    if (n) {
      GRB_HIP(hipMemcpyAsync(w->dval.p, values, n * ts, k, stream()));
      if (present) GRB_HIP(hipMemcpyAsync(w->dpres.p, present, n, k, stream()));
      else GRB_HIP(hipMemsetAsync(w->dpres.p, 1, n, stream()));
    }
    w->dev_valid = true; w->host_valid = false;
    w->dnvals = n; w->dnvals_known = !present;
    GRB_HIP(hipStreamSynchronize(stream()));
  });
  if (info) { GrB_Vector_free(&w); return info; }
  *v = w; return GrB_SUCCESS;
}
GrB_Info GrBX_Vector_import_Full(GrB_Vector* v, GrB_Type type, GrB_Index n, const void* values, int location) {
  return GrBX_Vector_import_Bitmap(v, type, n, values, nullptr, location);
}
GrB_Info GrBX_Vector_export_Bitmap(const GrB_Vector v, void* values, uint8_t* present, int location) {
  CHECK_VEC(v);
  return guarded(v, [&] {
    vec_to_device(v); const size_t ts = v->type->size; hipMemcpyKind k = location ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (values && v->n) GRB_HIP(hipMemcpyAsync(values, v->dval.p, v->n * ts, k, stream()));
    if (present && v->n) GRB_HIP(hipMemcpyAsync(present, v->dpres.p, v->n, k, stream()));
    GRB_HIP(hipStreamSynchronize(stream()));
  });
}
GrB_Info GrBX_Vector_device_view(GrB_Vector v, void** values, uint8_t** present, GrB_Index* nvals) {
  CHECK_VEC(v);
  return guarded(v, [&] { vec_to_device(v); if (values) *values = v->dval.p; if (present) *present = v->dpres.as<uint8_t>(); if (nvals) *nvals = vec_dev_nvals(v); });
}
GrB_Info GrBX_Vector_device_touch(GrB_Vector v) {
  CHECK_VEC(v);
  return guarded(v, [&] { if (!v->dev_valid) fail(GrB_INVALID_VALUE, "device_touch: vector has no device view");
    vec_invalidate_host(v); v->dnvals_known = false; });
}

}  // extern "C"
