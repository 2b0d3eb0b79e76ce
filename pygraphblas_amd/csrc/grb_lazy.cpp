// grb_lazy.cpp — non-blocking execution of the O(n) vector operations around the hot path.
//
// The reference initialises the library GrB_NONBLOCKING (pygraphblas/__init__.py:251-256) and observes results only through
// `nvals`, `reduce_*`, `extract*`, `wait` (gap/prmark.py:26, 51; SURVEY.md §8b "work may be deferred, results must be as-if
// sequential").  This file uses that licence for the vector operations of the reference's loops (gap/prmark.py:17-29):
//
//   * `w(:) = s` over every index, no mask, no accumulator      -> a note on the vector (lazy == 1), no kernel.  A product
//        `w += A (+).(x) u` with the monoid's own operator then folds the value into its store (grb_mxv.cpp), anything else
//        that touches w writes the fill first.
//   * eWiseAdd / eWiseMult / apply without mask and accumulator, every operand of the operator's type
//        -> a node of a short queue (<= 4 steps over <= 4 stored operands, each step reading stored vectors or the result of
//        the step before it).  The queue runs as ONE kernel (k_vec_chain, grb_lazy_kernels.hip) when any vector it involves
//        is next accessed (vec_gate in grb_container.cpp) — `t -= r; t = abs(t); t.reduce_float()` is one pass that also
//        produces the reduction, `w = t / d` writes zeros into the positions without an entry so that the product can gather
//        from it without a cast-and-fill pass.
//
// Errors that depend only on the arguments (dimensions, domains, uninitialised objects) are raised by the entry points before
// anything is queued, exactly as in blocking mode.  GRB_MI355X_BLOCKING=1 (or GrB_init(GrB_BLOCKING)) turns all of this off.
#include "grb_opcommon.hpp"
#include "grb_lazy.hpp"
#include <mutex>

namespace grb {

static bool g_nonblocking = false;
static int g_env_blocking = -1;
void set_nonblocking(bool on) { g_nonblocking = on; }
bool nonblocking() {
  if (g_env_blocking < 0) { const char* e = getenv("GRB_MI355X_BLOCKING"); g_env_blocking = (e && atoi(e) != 0) ? 1 : 0; }
  return g_nonblocking && !g_env_blocking;
}

static bool g_env_store_reduced() { static const int e = getenv("GRB_MI355X_LAZY_STORE_REDUCED") ? atoi(getenv("GRB_MI355X_LAZY_STORE_REDUCED")) : 0; return e != 0; }      // 1: a reduced chain is stored at once (round 3's behaviour)

void vec_chain_launch(const ChainLaunch& L, void* red_result) {
  if (!L.n || !L.nsteps) return;
  dispatch_type(L.tcode, [&]<class T>() { vec_chain_launch_t<T>(L, red_result); });
}

namespace {

struct Node {
  int kind;                 // 0: eWise (union / intersection), 1: apply (unary, or binary with a bound scalar)
  GrB_Vector out;           // nullptr once the result was overwritten or its vector freed before the queue ran
  GrB_Vector in[2];         // stored operands (nullptr when the operand is the previous step's result)
  bool prev[2];             // operand k is the result of the step before this one
  int op, mode; bool is_union; uint8_t scalar[16];
};
std::vector<Node> g_q;
int g_q_type = -1; uint64_t g_q_n = 0;
bool g_flushing = false;
bool g_q_kept = false;          // the queue as it stands was already run once for a reduction only (run_queue(keep)): a second reduction of it stores
std::recursive_mutex g_mu;
uint64_t g_stat_chains = 0, g_stat_nodes = 0, g_stat_fills_folded = 0, g_stat_reduces_fused = 0;

bool is_full(GrB_Vector v) { return v->dnvals_known && v->dnvals == v->n; }

void release_reads() { for (auto& nd : g_q) for (int k = 0; k < 2; k++) if (nd.in[k]) nd.in[k]->q_reads = 0; }

// run the queue as one kernel; `red`: also reduce the last step's result.  `keep` (with `red`): only the reduction is wanted now — nothing is
// stored and the queue stays as it is.  `t -= r; t = abs(t); t.reduce_float()` (gap/prmark.py:24-26) then never writes t at all: the next
// iteration assigns the vector as a whole (`r[:] = teleport` after the swap) and vec_overwritten() drops the two steps — a third of that pass's
// traffic.  Whoever looks at t first runs the queue in full.
void run_queue(const ChainReduce* red, void* red_result, bool keep = false) {
  if (g_q.empty()) return;
  g_flushing = true;
  // if anything below throws (an allocation, a launch): the queue is abandoned as a whole — every vector it would have written keeps its stored value
  // (what lazy == 2 describes anyway), no node keeps a pointer that a later GrB_free could leave dangling, and the caller sees the error
  struct Restore {
    bool committed = false;
    ~Restore() {
      g_flushing = false;
      if (!committed) { release_reads(); for (auto& nd : g_q) if (nd.out && nd.out->lazy == 2) { GrB_Vector w = nd.out; w->lazy = 0; if (!w->dev_valid) { w->hi.clear(); w->hx.clear(); w->pending.clear(); w->host_valid = true; } }      // (a result that never had buffers: empty, as after a failed call)
        g_q.clear(); g_q_type = -1; g_q_n = 0; g_q_kept = false; }
    }
  } restore;
  ChainLaunch L{};
  L.tcode = g_q_type; L.n = g_q_n; L.nsteps = (int)g_q.size();
  std::vector<GrB_Vector> ext;
  auto ext_slot = [&](GrB_Vector v) { for (size_t k = 0; k < ext.size(); k++) if (ext[k] == v) return (int)k; ext.push_back(v); return (int)ext.size() - 1; };
  // which step writes each output vector last: only that store happens
  std::vector<int> last_writer(g_q.size(), 1);
  for (size_t i = 0; i < g_q.size(); i++) for (size_t j = i + 1; j < g_q.size(); j++) if (g_q[i].out && g_q[j].out == g_q[i].out) last_writer[i] = 0;
  bool full_prev = false, math = false;
  std::vector<GrB_Vector> outs; std::vector<bool> out_full;
  for (size_t i = 0; i < g_q.size(); i++) {
    Node& nd = g_q[i]; ChainStepDesc& st = L.st[i];
    st.kind = nd.kind; st.op = nd.op; st.mode = nd.mode; st.is_union = nd.is_union ? 1 : 0; memcpy(st.scalar, nd.scalar, 16);
    bool f[2] = {false, false};
    for (int k = 0; k < (nd.kind == 0 ? 2 : 1); k++) {
      if (nd.prev[k]) { st.src[k] = CHAIN_PREV; f[k] = full_prev; }
      else { st.src[k] = ext_slot(nd.in[k]); f[k] = is_full(nd.in[k]); }
    }
    if (nd.kind == 0) { full_prev = nd.is_union ? (f[0] || f[1]) : (f[0] && f[1]); math = math || binop_needs_math(nd.op); }
    else { full_prev = f[0]; math = math || (nd.mode == 0 ? unop_needs_math_host(nd.op) : binop_needs_math(nd.op)); }
    st.out = -1;
    if (nd.out && last_writer[i] && !keep) { st.out = (int)outs.size(); outs.push_back(nd.out); out_full.push_back(full_prev); }
  }
  L.math = math;
  L.next = (int)ext.size();
  for (size_t k = 0; k < ext.size(); k++) {
    GrB_Vector v = ext[k];
    L.ev[k] = v->dval.p; L.ep[k] = is_full(v) ? nullptr : v->dpres.as<uint8_t>();   // a full operand: its presence bytes are not read
  }
  const size_t ts = type_size(g_q_type);
  std::vector<DevBuf> nval(outs.size()), npres(outs.size());
  for (size_t o = 0; o < outs.size(); o++) {
    GrB_Vector w = outs[o];
    bool in_place = false;
    for (GrB_Vector e : ext) if (e == w) in_place = true;                // (element-wise: every stored operand of element i is read before anything of element i is written)
    if (in_place) {
      L.ov[o] = w->dval.p;
      // a full result into a full vector's own buffers: the presence bytes are all ones already
      L.op[o] = (out_full[o] && is_full(w)) ? nullptr : w->dpres.as<uint8_t>();
    } else {
      nval[o].alloc(g_q_n * ts + 8); npres[o].alloc(g_q_n + 8);
      L.ov[o] = nval[o].p; L.op[o] = npres[o].as<uint8_t>();
    }
  }
  L.nout = (int)outs.size();
  if (red) L.red = *red;
  vec_chain_launch(L, red_result);
  g_stat_chains++; if (red) g_stat_reduces_fused++;
  if (!g_q_kept) g_stat_nodes += g_q.size();                     // (the operations of a kept queue are counted once, whichever run stores them)
  restore.committed = true;
  if (keep) { g_q_kept = true; return; }
  g_q_kept = false;
  release_reads();
  for (size_t o = 0; o < outs.size(); o++) {
    GrB_Vector w = outs[o];
    if (nval[o].p) { w->dval = std::move(nval[o]); w->dpres = std::move(npres[o]); }
    w->lazy = 0; w->dev_valid = true; w->host_valid = false; w->hi.clear(); w->hx.clear(); w->pending.clear();
    w->dnvals_known = out_full[o]; w->dnvals = out_full[o] ? w->n : 0;
    w->holes_zero = true; w->holes_big = false; w->fe_lb = 0; w->fe_lb_key = 0; w->lor_state = 0; w->abs_bound = -1; w->small_valid = false; w->code_valid = false;
  }
  g_q.clear(); g_q_type = -1; g_q_n = 0;
}

void materialise_fill(GrB_Vector w) {
  const uint64_t n = w->n; const size_t ts = w->type->size;
  w->lazy = 0;
  w->dval.alloc(n * ts ? n * ts : 1); w->dpres.alloc(n ? n : 1);
  vec_assign_scalar(w->type->code, n, w->dval.p, w->dpres.as<uint8_t>(), nullptr, nullptr, w->lazy_fill, -1, false);
  w->dev_valid = true; w->host_valid = false; w->dnvals = n; w->dnvals_known = true; w->holes_zero = false; w->holes_big = false; w->lor_state = 0; w->abs_bound = -1; w->small_valid = false; w->code_valid = false;
}

}  // namespace

void lazy_flush() { std::lock_guard<std::recursive_mutex> lk(g_mu); if (!g_flushing) run_queue(nullptr, nullptr); }

void vec_resolve(GrB_Vector v) {
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (g_flushing) return;
  if (v->lazy == 2 || v->q_reads) run_queue(nullptr, nullptr);
  if (v->lazy == 1) materialise_fill(v);
}

// drop the steps nothing needs any more: a step is needed when its result is stored, or when the step after it is needed and reads it
static void prune_queue() {
  if (g_q.empty()) return;
  std::vector<char> need(g_q.size(), 0);
  for (size_t i = g_q.size(); i-- > 0;) {
    const bool next_reads = i + 1 < g_q.size() && need[i + 1] && (g_q[i + 1].prev[0] || g_q[i + 1].prev[1]);
    need[i] = g_q[i].out != nullptr || next_reads;
  }
  size_t w = 0;
  for (size_t i = 0; i < g_q.size(); i++) {
    if (need[i]) { if (w != i) g_q[w] = g_q[i]; w++; }
    else for (int k = 0; k < 2; k++) if (g_q[i].in[k] && g_q[i].in[k]->q_reads) g_q[i].in[k]->q_reads--;
  }
  g_q.resize(w);
  if (g_q.size() != need.size()) g_q_kept = false;                 // (a pruned queue is another queue)
  if (g_q.empty()) { g_q_type = -1; g_q_n = 0; }
}

void vec_overwritten(GrB_Vector v) {
  if (!(v->lazy | v->q_reads)) return;
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (g_flushing) return;
  if (v->lazy == 2) {                                   // queued work produced it: the stores are dropped (later steps may still use the value in registers),
    for (auto& nd : g_q) if (nd.out == v) nd.out = nullptr;      // and with them the steps nothing else needs — which may have been the only readers of v's stored value
    prune_queue();
  }
  if (v->q_reads) run_queue(nullptr, nullptr);          // queued work still reads the value that is about to go: it runs first
  v->lazy = 0;
}

// ---- `w(:) = s` -------------------------------------------------------------------------------------------------------------
bool lazy_fill(GrB_Vector w, const void* s_in_w_type) {
  if (!nonblocking() || !device_ok() || w->n == 0 || w->n > GRB_DIM_DEVICE_MAX || w->type->code >= T_FC32) return false;
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (g_flushing) return false;
  vec_overwritten(w);
  // the buffers go back to the pool: whoever consumes the fill allocates the result
  w->hi.clear(); w->hx.clear(); w->pending.clear(); w->host_valid = false; w->iso_full = false;
  w->dev_valid = false; w->dval.reset(); w->dpres.reset(); w->dnvals = 0; w->dnvals_known = false; w->fe_lb = 0; w->fe_lb_key = 0; w->holes_zero = false; w->holes_big = false;
  w->lazy = 1; memcpy(w->lazy_fill, s_in_w_type, 16); w->lor_state = 0; w->abs_bound = -1; w->small_valid = false; w->code_valid = false;
  return true;
}
void lazy_fill_consumed(GrB_Vector w) { w->lazy = 0; g_stat_fills_folded++; }

// ---- element-wise nodes ------------------------------------------------------------------------------------------------------
static bool enqueue(Node nd, GrB_Vector w, int tcode) {
  const uint64_t n = w->n;
  // operands: the previous step's result, or a stored vector (brought to HBM now; it stays untouched until the queue ran)
  const int nin = nd.kind == 0 ? 2 : 1;
  for (int pass = 0; pass < 2; pass++) {
    bool need_flush = false;
    if (!g_q.empty() && (g_q_type != tcode || g_q_n != n || g_q.size() >= CHAIN_MAX_STEPS)) need_flush = true;
    for (int k = 0; k < nin && !need_flush; k++) {
      GrB_Vector x = nd.in[k];
      if (x->lazy == 2) { if (g_q.empty() || g_q.back().out != x) need_flush = true; }      // produced by an older step (or a dropped one): only the last result is in reach
    }
    if (!need_flush) {
      // distinct stored operands of the whole queue
      std::vector<GrB_Vector> ext;
      auto add = [&](GrB_Vector v) { for (auto e : ext) if (e == v) return; ext.push_back(v); };
      for (auto& q : g_q) for (int k = 0; k < 2; k++) if (q.in[k]) add(q.in[k]);
      for (int k = 0; k < nin; k++) if (nd.in[k]->lazy != 2) add(nd.in[k]);
      size_t nouts = 1; { std::vector<GrB_Vector> os; for (auto& q : g_q) if (q.out) { bool f = false; for (auto o : os) if (o == q.out) f = true; if (!f) os.push_back(q.out); } nouts = os.size() + 1; }
      if (ext.size() > CHAIN_MAX_IN || nouts > CHAIN_MAX_OUT) need_flush = true;
    }
    if (!need_flush) break;
    if (pass == 1) return false;
    run_queue(nullptr, nullptr);
  }
  for (int k = 0; k < nin; k++) {
    GrB_Vector x = nd.in[k];
    if (x->lazy == 2) { nd.prev[k] = true; nd.in[k] = nullptr; }
    else { nd.prev[k] = false; if (!x->dev_valid || x->lazy == 1) vec_to_device(x); }      // (a pending fill of x is written here; an operand the queue already reads is resident)
  }
  for (int k = 0; k < nin; k++) if (nd.in[k]) nd.in[k]->q_reads++;
  // the output: what it held is replaced as a whole — unless it is also a stored operand of the queue, its buffers are not needed
  if (w->lazy == 1) w->lazy = 0;
  else if (w->lazy == 2) { for (auto& q : g_q) if (q.out == w) q.out = nullptr; }
  // (while lazy == 2, dnvals / holes_zero keep describing the STORED value — the queue reads it; nobody else can without passing vec_gate)
  if (!w->q_reads) { w->hi.clear(); w->hx.clear(); w->pending.clear(); w->host_valid = false; w->iso_full = false; w->dev_valid = false; w->dval.reset(); w->dpres.reset();
                     w->dnvals = 0; w->dnvals_known = false; w->holes_zero = false; w->holes_big = false; }
  else { w->host_valid = false; w->hi.clear(); w->hx.clear(); w->pending.clear(); }
  w->fe_lb = 0; w->fe_lb_key = 0; w->lor_state = 0; w->abs_bound = -1; w->small_valid = false; w->code_valid = false;
  w->lazy = 2;
  nd.out = w;
  g_q.push_back(nd); g_q_type = tcode; g_q_n = n; g_q_kept = false;
  return true;
}

static bool chainable(GrB_Vector w, uint64_t n) {
  return nonblocking() && device_ok() && !g_flushing && n > 0 && n <= GRB_DIM_DEVICE_MAX && vec_chain_type_supported(w->type->code);
}

bool lazy_ewise(GrB_Vector w, GrB_BinaryOp op, GrB_Vector u, GrB_Vector v, bool is_union) {
  const int tc = w->type->code;
  if (!chainable(w, w->n)) return false;
  if (op->xtype->code != tc || op->ytype->code != tc || op->ztype->code != tc || u->type->code != tc || v->type->code != tc) return false;
  if (op->opcode > B_LXOR || op->opcode == B_POW) return false;      // the chain kernel carries the compact operator switch (FIRST .. LXOR)
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (g_flushing) return false;
  Node nd{}; nd.kind = 0; nd.in[0] = u; nd.in[1] = v; nd.op = op->opcode; nd.is_union = is_union;
  return enqueue(nd, w, tc);
}

bool lazy_apply(GrB_Vector w, int mode, int opcode, int xcode, int zcode, const void* scalar_in_x_type, GrB_Vector u) {
  const int tc = w->type->code;
  if (!chainable(w, w->n)) return false;
  if (xcode != tc || zcode != tc || u->type->code != tc) return false;
  if (mode == 0 ? opcode > U_BNOT : (opcode > B_LXOR || opcode == B_POW)) return false;
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (g_flushing) return false;
  Node nd{}; nd.kind = 1; nd.in[0] = u; nd.in[1] = nullptr; nd.op = opcode; nd.mode = mode;
  if (scalar_in_x_type) memcpy(nd.scalar, scalar_in_x_type, 16);
  return enqueue(nd, w, tc);
}

// `reduce(u)` when u is the result of the queue's last step: the chain kernel reduces it on the way
bool lazy_reduce(GrB_Vector u, int mop, int mcode, const void* identity, void* result_in_mcode, bool may_keep) {
  if (u->lazy != 2) return false;
  std::lock_guard<std::recursive_mutex> lk(g_mu);
  if (g_flushing || g_q.empty() || g_q.back().out != u) return false;
  const int tc = g_q_type;
  const bool same = mcode == tc, widen = (tc == T_FP32 && mcode == T_FP64);
  if (!(same || widen)) return false;
  if (!(mop == B_PLUS || mop == B_MIN || mop == B_MAX || mop == B_TIMES || mop == B_LOR || mop == B_LAND || mop == B_LXOR || mop == B_ANY)) return false;
  if (tc == T_BOOL && !(mop == B_LOR || mop == B_LAND || mop == B_LXOR)) return false;
  ChainReduce r{}; r.on = 1; r.op = mop; r.widen = widen ? 1 : 0; memcpy(r.identity, identity, 16);
  // (a queue that was already run for a reduction and is reduced AGAIN — a norm, then another look — stores this time: a third pass is the most it costs)
  run_queue(&r, result_in_mcode, may_keep && !g_env_store_reduced() && !g_q_kept);
  return true;
}

}  // namespace grb

extern "C" GrB_Info GrBX_lazy_stats(uint64_t* chains, uint64_t* nodes, uint64_t* fills_folded, uint64_t* reduces_fused) {
  if (chains) *chains = grb::g_stat_chains; if (nodes) *nodes = grb::g_stat_nodes;
  if (fills_folded) *fills_folded = grb::g_stat_fills_folded; if (reduces_fused) *reduces_fused = grb::g_stat_reduces_fused;
  return GrB_SUCCESS;
}
