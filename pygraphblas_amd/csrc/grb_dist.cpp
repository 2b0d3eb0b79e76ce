// grb_dist.cpp — the exchange steps of the row-partitioned hot path, inside the library (RCCL, bound at first use: see below).
//
// The reference has no distributed code; BASELINE.json's north star partitions the matrix by row
// blocks across the GPUs of one node — one process per GPU — and names the exchange: an allgatherv of the operand /
// frontier vector over xGMI before each product, plus the all-reduce of a scalar (PageRank's residual, the triangle
// count).  Any host that binds the C ABI gets it (GrBX_dist_* in include/grb_mi355x.h); torch is not involved.
//
//   * ncclCommInitRank from a 128-byte id the host passes around however it likes (GrBX_dist_unique_id on rank 0).
//   * GrBX_Vector_allgatherv_start(full, local, bounds, presence): rank p owns full[bounds[p], bounds[p+1]).  The own slice is
//     copied on the compute stream; the slices of the other ranks arrive by grouped ncclSend/ncclRecv — every pair of
//     GPUs exchanges directly over its own xGMI link (MI355X: fully connected, 7 links per GPU; a ring would put the whole
//     vector on every link) — on a second HIP stream, so the caller can run the part of the product that needs only local
//     columns (the diagonal block of its row block) while the rest of the vector is in flight.  GrBX_dist_wait() makes the
//     compute stream wait for it.
//   * GrBX_Vector_allgatherv_bits: a BOOL frontier travels as one bit per vertex (n/8 bytes instead of 2n).
//   * GrBX_dist_allreduce: a few host scalars through a device staging buffer and ncclAllReduce.
// Without GrBX_dist_init the process is a world of one and every call degenerates to the local copy.
//
// RCCL is bound at the first GrBX_dist_* call that needs it (dlopen + dlsym of the eight entry points used), not at load
// time: a process that also hosts PyTorch already carries PyTorch's own copy of librccl, and two copies of the library in
// one symbol namespace abort at exit (double free in their static destructors).  The copy that is already loaded is used
// if there is one, /opt/rocm/lib/librccl.so.1 otherwise.
#include "grb_api.hpp"
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <string.h>

using namespace grb;

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
} R;
void rccl_bind() {
  if (R.h) return;
  const char* env = getenv("GRB_MI355X_RCCL");
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  if (env && *env) {
    // an explicit library (tests/libfake_rccl.so: two ranks on one GPU) is bound as named — never replaced by the copy of librccl a
    // process that hosts PyTorch already carries.  dlsym on ITS handle finds its own definitions first.
    h = dlopen(env, RTLD_LAZY | RTLD_LOCAL);
    if (!h) fail(GrB_PANIC, std::string("GRB_MI355X_RCCL: cannot load ") + env + ": " + dlerror());
  }
  for (int pass = 0; pass < 2 && !h; pass++)                 // first whatever copy the process already holds, then a fresh load
    for (const char* nm : names) { h = dlopen(nm, RTLD_LAZY | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0)); if (h) break; }
  if (!h) fail(GrB_PANIC, std::string("RCCL not found: ") + dlerror());
#define GRB_SYM(F) *(void**)&R.F = dlsym(h, "nccl" #F); if (!R.F) fail(GrB_PANIC, "RCCL lacks nccl" #F)
  GRB_SYM(GetUniqueId); GRB_SYM(CommInitRank); GRB_SYM(CommDestroy); GRB_SYM(GroupStart); GRB_SYM(GroupEnd); GRB_SYM(Send); GRB_SYM(Recv);
  GRB_SYM(AllReduce); GRB_SYM(GetErrorString);
#undef GRB_SYM
  R.h = h;
}
ncclComm_t g_comm = nullptr;
int g_rank = 0, g_world = 1;
hipStream_t g_cstream = nullptr;          // the exchange runs here
hipEvent_t g_ev_ready = nullptr, g_ev_done = nullptr;
bool g_pending = false;                   // an exchange was started and not waited for

void nccl_check(ncclResult_t r, const char* what) {
  if (r != ncclSuccess) fail(GrB_PANIC, std::string(what) + ": " + R.GetErrorString(r));
}
void ensure_streams() {
  if (g_cstream) return;
  GRB_HIP(hipStreamCreateWithFlags(&g_cstream, hipStreamNonBlocking));
  GRB_HIP(hipEventCreateWithFlags(&g_ev_ready, hipEventDisableTiming));
  GRB_HIP(hipEventCreateWithFlags(&g_ev_done, hipEventDisableTiming));
}
void check_bounds(const GrB_Index* bounds, GrB_Index n) {
  if (!bounds) fail(GrB_NULL_POINTER, "allgatherv: bounds is NULL");
  if (bounds[0] != 0 || bounds[g_world] != n) fail(GrB_INVALID_VALUE, "allgatherv: bounds must run from 0 to the vector's length");
  for (int p = 0; p < g_world; p++) if (bounds[p] > bounds[p + 1]) fail(GrB_INVALID_VALUE, "allgatherv: bounds must be non-decreasing");
}

// bit p of out[] = vertex p of the slice holds `true`
__global__ void k_pack_bits(const uint8_t* __restrict__ val, const uint8_t* __restrict__ pres, uint64_t len, uint8_t* __restrict__ out) {
  const uint64_t nbytes = (len + 7) / 8;
  for (uint64_t b = blockIdx.x * 256ull + threadIdx.x; b < nbytes; b += gridDim.x * 256ull) {
    uint32_t x = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { const uint64_t i = b * 8 + j; if (i < len && pres[i] && val[i]) x |= 1u << j; }
    out[b] = (uint8_t)x;
  }
}
__global__ void k_unpack_bits(const uint8_t* __restrict__ bits, uint64_t len, uint8_t* __restrict__ val, uint8_t* __restrict__ pres) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < len; i += gridDim.x * 256ull) {
    const uint8_t t = (bits[i >> 3] >> (i & 7)) & 1u; val[i] = t; pres[i] = t;
  }
}
unsigned grid_for(uint64_t n) { uint64_t b = (n + 255) / 256; if (b < 1) b = 1; if (b > 8192) b = 8192; return (unsigned)b; }

bool nccl_type(int code, ncclDataType_t* t) {
  switch (code) {
    case T_BOOL: case T_UINT8: *t = ncclUint8; return true;
    case T_INT8: *t = ncclInt8; return true;
    case T_INT32: *t = ncclInt32; return true;   case T_UINT32: *t = ncclUint32; return true;
    case T_INT64: *t = ncclInt64; return true;   case T_UINT64: *t = ncclUint64; return true;
    case T_FP32: *t = ncclFloat32; return true;  case T_FP64: *t = ncclFloat64; return true;
    default: return false;
  }
}
}  // namespace
namespace grb { bool dist_exchange_pending() { return g_pending; } }      // an exchange is writing into a vector's buffers on the second stream (grb_mxv.cpp: no in-place writes into operands meanwhile)

extern "C" {

GrB_Info GrBX_dist_unique_id(void* id, int len) {
  if (!id) return GrB_NULL_POINTER;
  if (len < (int)NCCL_UNIQUE_ID_BYTES) return GrB_INSUFFICIENT_SPACE;
  return guarded((GrB_Vector) nullptr, [&] {
    rccl_bind();
    ncclUniqueId u; nccl_check(R.GetUniqueId(&u), "ncclGetUniqueId"); memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  });
}

GrB_Info GrBX_dist_init(int rank, int world, const void* id, int len) {
  if (world < 1 || rank < 0 || rank >= world) return GrB_INVALID_VALUE;
  if (world > 1 && (!id || len < (int)NCCL_UNIQUE_ID_BYTES)) return GrB_NULL_POINTER;
  return guarded((GrB_Vector) nullptr, [&] {
    need_device();
    if (g_comm) fail(GrB_INVALID_VALUE, "GrBX_dist_init: already initialised");
    ensure_streams(); rccl_bind();
    ncclUniqueId u; memset(&u, 0, sizeof u);
    if (id) memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES); else nccl_check(R.GetUniqueId(&u), "ncclGetUniqueId");
    nccl_check(R.CommInitRank(&g_comm, world, u, rank), "ncclCommInitRank");
    g_rank = rank; g_world = world;
  });
}

GrB_Info GrBX_dist_finalize(void) {
  return guarded((GrB_Vector) nullptr, [&] {
    if (g_cstream) (void)hipStreamSynchronize(g_cstream);
    if (g_comm) { (void)R.CommDestroy(g_comm); g_comm = nullptr; }
    g_rank = 0; g_world = 1; g_pending = false;
  });
}

GrB_Info GrBX_dist_info(int* rank, int* world) { if (rank) *rank = g_rank; if (world) *world = g_world; return GrB_SUCCESS; }

// which RCCL-ABI library carries the exchange: the file that defines the bound ncclSend ("" while nothing is bound yet)
GrB_Info GrBX_dist_transport(char* buf, int len) {
  if (!buf || len <= 0) return GrB_NULL_POINTER;
  buf[0] = 0;
  Dl_info di;
  if (R.h && R.Send && dladdr((void*)R.Send, &di) && di.dli_fname) snprintf(buf, (size_t)len, "%s", di.dli_fname);
  return GrB_SUCCESS;
}

// full[bounds[p], bounds[p+1]) <- rank p's `local` (length bounds[p+1]-bounds[p]) for every p.  `local` may be NULL when the
// caller wrote its slice through the device view of `full` already.  presence != 0: the presence bytes travel too (operands
// with holes); presence == 0: `full` is treated as all-present afterwards (its presence bytes are set once by the caller).
GrB_Info GrBX_Vector_allgatherv_start(GrB_Vector full, const GrB_Vector local, const GrB_Index* bounds, int presence) {
  if (!full) return GrB_NULL_POINTER;
  if (!check_obj(full) || (local && !check_obj(local))) return GrB_UNINITIALIZED_OBJECT;
  return guarded(full, [&] {
    need_device();
    check_bounds(bounds, full->n);
    if (g_pending) fail(GrB_INVALID_VALUE, "allgatherv: the previous exchange has not been waited for (GrBX_dist_wait)");
    const size_t ts = full->type->size;
    const GrB_Index r0 = bounds[g_rank], r1 = bounds[g_rank + 1];
    vec_to_device(full);
    if (local) {
      if (local->type != full->type) fail(GrB_DOMAIN_MISMATCH, "allgatherv: local and full vectors differ in type");
      if (local->n != r1 - r0) fail(GrB_DIMENSION_MISMATCH, "allgatherv: the local vector's length is not this rank's slice");
      vec_to_device(local);
      if (r1 > r0) {
        GRB_HIP(hipMemcpyAsync((char*)full->dval.p + r0 * ts, local->dval.p, (r1 - r0) * ts, hipMemcpyDeviceToDevice, stream()));
        if (presence) GRB_HIP(hipMemcpyAsync(full->dpres.as<uint8_t>() + r0, local->dpres.p, r1 - r0, hipMemcpyDeviceToDevice, stream()));
      }
    }
    vec_invalidate_host(full);
    // entry count of the gathered vector: all-present without presence bytes.  With them the true count is the sum of the ranks'
    // counts, which nobody needs on the hot path; the vector is marked "not full" (n - 1) instead of "unknown", so that the
    // products that follow neither recount it (a kernel and a host round trip, over slices that may still be in flight) nor skip
    // the presence bytes.  GrBX_Vector_device_touch re-establishes the exact count when a caller wants it.
    if (presence) { full->dnvals = full->n ? full->n - 1 : 0; full->dnvals_known = true; } else { full->dnvals = full->n; full->dnvals_known = true; }
    if (g_world == 1) return;
    if (!g_comm) fail(GrB_INVALID_VALUE, "allgatherv: GrBX_dist_init has not been called");
    ensure_streams();
    GRB_HIP(hipEventRecord(g_ev_ready, stream()));            // the own slice is complete
    GRB_HIP(hipStreamWaitEvent(g_cstream, g_ev_ready, 0));
    nccl_check(R.GroupStart(), "ncclGroupStart");
    for (int d = 1; d < g_world; d++) {                        // peers in rotating order: rank r talks to r+d and r-d in step d
      const int to = (g_rank + d) % g_world, from = (g_rank - d + g_world) % g_world;
      if (r1 > r0) {
        nccl_check(R.Send((const char*)full->dval.p + r0 * ts, (r1 - r0) * ts, ncclUint8, to, g_comm, g_cstream), "ncclSend");
        if (presence) nccl_check(R.Send(full->dpres.as<uint8_t>() + r0, r1 - r0, ncclUint8, to, g_comm, g_cstream), "ncclSend");
      }
      const GrB_Index f0 = bounds[from], f1 = bounds[from + 1];
      if (f1 > f0) {
        nccl_check(R.Recv((char*)full->dval.p + f0 * ts, (f1 - f0) * ts, ncclUint8, from, g_comm, g_cstream), "ncclRecv");
        if (presence) nccl_check(R.Recv(full->dpres.as<uint8_t>() + f0, f1 - f0, ncclUint8, from, g_comm, g_cstream), "ncclRecv");
      }
    }
    nccl_check(R.GroupEnd(), "ncclGroupEnd");
    GRB_HIP(hipEventRecord(g_ev_done, g_cstream));
    g_pending = true;
  });
}

GrB_Info GrBX_dist_wait(void) {
  return guarded((GrB_Vector) nullptr, [&] {
    if (!g_pending) return;
    GRB_HIP(hipStreamWaitEvent(stream(), g_ev_done, 0));
    g_pending = false;
  });
}

GrB_Info GrBX_Vector_allgatherv(GrB_Vector full, const GrB_Vector local, const GrB_Index* bounds, int presence) {
  GrB_Info info = GrBX_Vector_allgatherv_start(full, local, bounds, presence);
  return info != GrB_SUCCESS ? info : GrBX_dist_wait();
}

// The BOOL frontier of a partitioned BFS: vertex i of `full` becomes (true, present) iff some rank's `local` holds true there.
// Each rank packs its slice to bits (slice p starts at byte offset sum_{q<p} ceil(len_q / 8)), the bytes are exchanged,
// and one kernel per slice expands them into the bitmap layout: n/8 bytes on the links instead of 2n.
GrB_Info GrBX_Vector_allgatherv_bits(GrB_Vector full, const GrB_Vector local, const GrB_Index* bounds) {
  if (!full || !local) return GrB_NULL_POINTER;
  if (!check_obj(full) || !check_obj(local)) return GrB_UNINITIALIZED_OBJECT;
  return guarded(full, [&] {
    need_device();
    check_bounds(bounds, full->n);
    if (full->type->code != T_BOOL || local->type->code != T_BOOL) fail(GrB_DOMAIN_MISMATCH, "allgatherv_bits: BOOL vectors only");
    const GrB_Index r0 = bounds[g_rank], r1 = bounds[g_rank + 1];
    if (local->n != r1 - r0) fail(GrB_DIMENSION_MISMATCH, "allgatherv_bits: the local vector's length is not this rank's slice");
    if (g_pending) fail(GrB_INVALID_VALUE, "allgatherv_bits: an exchange is still in flight (GrBX_dist_wait)");
    vec_to_device(full); vec_to_device(local);
    std::vector<uint64_t> off(g_world + 1, 0);
    for (int p = 0; p < g_world; p++) off[p + 1] = off[p] + (bounds[p + 1] - bounds[p] + 7) / 8;
    DevBuf bits(off[g_world] + 8);
    hipLaunchKernelGGL(k_pack_bits, dim3(grid_for((r1 - r0 + 7) / 8)), dim3(256), 0, stream(), local->dval.as<uint8_t>(), local->dpres.as<uint8_t>(), (uint64_t)(r1 - r0),
                       bits.as<uint8_t>() + off[g_rank]);
    if (g_world > 1) {
      if (!g_comm) fail(GrB_INVALID_VALUE, "allgatherv_bits: GrBX_dist_init has not been called");
      nccl_check(R.GroupStart(), "ncclGroupStart");          // (on the compute stream: nothing to overlap with, the next step needs the frontier)
      for (int d = 1; d < g_world; d++) {
        const int to = (g_rank + d) % g_world, from = (g_rank - d + g_world) % g_world;
        if (off[g_rank + 1] > off[g_rank]) nccl_check(R.Send(bits.as<uint8_t>() + off[g_rank], off[g_rank + 1] - off[g_rank], ncclUint8, to, g_comm, stream()), "ncclSend");
        if (off[from + 1] > off[from]) nccl_check(R.Recv(bits.as<uint8_t>() + off[from], off[from + 1] - off[from], ncclUint8, from, g_comm, stream()), "ncclRecv");
      }
      nccl_check(R.GroupEnd(), "ncclGroupEnd");
    }
    for (int p = 0; p < g_world; p++) {
      const uint64_t len = bounds[p + 1] - bounds[p];
      if (len) hipLaunchKernelGGL(k_unpack_bits, dim3(grid_for(len)), dim3(256), 0, stream(), bits.as<uint8_t>() + off[p], len, full->dval.as<uint8_t>() + bounds[p],
                                  full->dpres.as<uint8_t>() + bounds[p]);
    }
    GRB_HIP(hipGetLastError());
    vec_invalidate_host(full); full->dnvals_known = false; full->fe_lb = 0; full->fe_lb_key = 0;
  });
}

// buf[0..count) (host, `type`) <- reduction over all ranks with the monoid's operator (PLUS, MIN, MAX, TIMES; LOR / LAND on BOOL)
GrB_Info GrBX_dist_allreduce(void* buf, GrB_Index count, GrB_Type type, GrB_BinaryOp op) {
  if (!buf || !type || !op) return GrB_NULL_POINTER;
  if (!check_obj(type) || !check_obj(op)) return GrB_UNINITIALIZED_OBJECT;
  return guarded((GrB_Vector) nullptr, [&] {
    if (g_world == 1 && !g_comm) return;            // (a world of one with a communicator still goes through RCCL: the GPU test of this path)
    need_device();
    if (!g_comm) fail(GrB_INVALID_VALUE, "allreduce: GrBX_dist_init has not been called");
    ncclDataType_t dt; if (!nccl_type(type->code, &dt)) fail(GrB_DOMAIN_MISMATCH, "allreduce: type not supported");
    ncclRedOp_t ro;
    const bool isbool = type->code == T_BOOL;
    switch (op->opcode) {
      case B_PLUS: ro = isbool ? ncclMax : ncclSum; break;
      case B_TIMES: ro = isbool ? ncclMin : ncclProd; break;
      case B_MIN: ro = ncclMin; break;   case B_MAX: ro = ncclMax; break;
      case B_LOR: if (!isbool) fail(GrB_DOMAIN_MISMATCH, "allreduce: LOR needs BOOL"); ro = ncclMax; break;
      case B_LAND: if (!isbool) fail(GrB_DOMAIN_MISMATCH, "allreduce: LAND needs BOOL"); ro = ncclMin; break;
      default: fail(GrB_DOMAIN_MISMATCH, "allreduce: operator not supported (PLUS, TIMES, MIN, MAX, LOR, LAND)");
    }
    const size_t bytes = count * type->size;
    DevBuf d(bytes + 8);
    GRB_HIP(hipMemcpyAsync(d.p, buf, bytes, hipMemcpyHostToDevice, stream()));
    nccl_check(R.AllReduce(d.p, d.p, count, dt, ro, g_comm, stream()), "ncclAllReduce");
    GRB_HIP(hipMemcpyAsync(buf, d.p, bytes, hipMemcpyDeviceToHost, stream()));
    GRB_HIP(hipStreamSynchronize(stream()));
  });
}

}  // extern "C"
