// fake_rccl.cpp — TEST INFRASTRUCTURE, not part of the product: a librccl-shaped shared library whose nine entry points move
// DEVICE pointers between processes that share ONE GPU, so that the library's own exchange path (pygraphblas_amd/csrc/grb_dist.cpp:
// grouped ncclSend / ncclRecv on a second HIP stream, the ready / done events, presence bytes, the bit frontier, ncclAllReduce) runs
// with two real ranks on a one-GPU box — everything except xGMI.  RCCL itself refuses two ranks on one device.
//
//   GRB_MI355X_RCCL=tests/libfake_rccl.so   makes grb_dist.cpp bind this file instead of librccl (rccl_bind()).
//
// How it moves data: the ranks meet in a POSIX shared-memory control block named by the 128-byte "unique id".  A send exports the
// allocation that holds the buffer as a hipIpc memory handle (+ offset, length) into the slot (from, to, k) — k = the k-th send to that
// peer inside the group; the matching receive opens the handle, copies device-to-device ON THE STREAM IT WAS GIVEN, synchronises that
// stream and marks the slot consumed; the sender returns from ncclGroupEnd once its sends were consumed.  An all-reduce stages the
// ranks' buffers through the control block and every rank reduces them in rank order.
// With more than two ranks a buffer would be imported by several peers at once, which the dmabuf IPC of this driver refuses
// (hipIpcOpenMemHandle: invalid device pointer — an exported handle serves one importer): there the bytes are staged through a POSIX
// shared-memory file per transfer (device -> shm by the sender, shm -> device on the receiver's stream).  Matching, ordering and the
// library's code path are the same; only the copy engine differs.
// Unlike RCCL the calls are host-synchronous (the stream is drained at ncclGroupEnd): ordering bugs that only show with asynchronous
// progress are not provoked, data movement and matching are.  Every wait has a deadline (FAKE_RCCL_TIMEOUT_S, default 60 s) and fails
// with ncclSystemError instead of hanging the box.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <vector>

namespace {
constexpr int MAXR = 8, MAXK = 8, AR_BYTES = 1 << 16;
struct Slot {
  volatile uint64_t posted, done;       // sequence numbers of the last transfer posted by the sender / consumed by the receiver
  hipIpcMemHandle_t h; uint64_t offset, bytes;
  int staged; char shm[96];            // staged != 0: the bytes are in the shared-memory file `shm` instead of behind the handle
};
struct Ctl {
  volatile int arrived, left;
  volatile int bar_count; volatile int bar_gen;
  int world;
  Slot p2p[MAXR][MAXR][MAXK];
  unsigned char ar[MAXR][AR_BYTES];
};
struct Comm {
  Ctl* c = nullptr; int rank = 0, world = 1; char name[64] = {0};
  uint64_t scount[MAXR][MAXK] = {{0}}, rcount[MAXR][MAXK] = {{0}};
};
struct Op { bool send; void* buf; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

double now_s() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
double timeout_s() { const char* e = getenv("FAKE_RCCL_TIMEOUT_S"); return e && *e ? atof(e) : 60.0; }
template <class F> bool wait_until(F&& f) {
  const double t0 = now_s(), lim = timeout_s();
  while (!f()) { __sync_synchronize(); if (now_s() - t0 > lim) return false; usleep(20); }
  __sync_synchronize();
  return true;
}
bool barrier(Comm* m) {
  Ctl* c = m->c;
  const int gen = c->bar_gen;
  if (__sync_add_and_fetch(&c->bar_count, 1) == m->world) { c->bar_count = 0; __sync_synchronize(); __sync_add_and_fetch(&c->bar_gen, 1); return true; }
  return wait_until([&] { return c->bar_gen != gen; });
}
size_t dt_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}
template <class T> void reduce_t(T* acc, const T* x, size_t n, ncclRedOp_t op) {
  for (size_t i = 0; i < n; i++) {
    switch (op) {
      case ncclSum: acc[i] = (T)(acc[i] + x[i]); break;
      case ncclProd: acc[i] = (T)(acc[i] * x[i]); break;
      case ncclMax: acc[i] = x[i] > acc[i] ? x[i] : acc[i]; break;
      case ncclMin: acc[i] = x[i] < acc[i] ? x[i] : acc[i]; break;
      default: break;
    }
  }
}
bool reduce_any(void* acc, const void* x, size_t n, ncclDataType_t t, ncclRedOp_t op) {
  switch (t) {
    case ncclInt8: reduce_t((int8_t*)acc, (const int8_t*)x, n, op); return true;
    case ncclUint8: reduce_t((uint8_t*)acc, (const uint8_t*)x, n, op); return true;
    case ncclInt32: reduce_t((int32_t*)acc, (const int32_t*)x, n, op); return true;
    case ncclUint32: reduce_t((uint32_t*)acc, (const uint32_t*)x, n, op); return true;
    case ncclInt64: reduce_t((int64_t*)acc, (const int64_t*)x, n, op); return true;
    case ncclUint64: reduce_t((uint64_t*)acc, (const uint64_t*)x, n, op); return true;
    case ncclFloat32: reduce_t((float*)acc, (const float*)x, n, op); return true;
    case ncclFloat64: reduce_t((double*)acc, (const double*)x, n, op); return true;
    default: return false;
  }
}
#define FK_HIP(e) do { hipError_t fk_e = (e); if (fk_e != hipSuccess) { fprintf(stderr, "[fake_rccl] %s -> %s\n", #e, hipGetErrorString(fk_e)); return ncclUnhandledCudaError; } } while (0)

ncclResult_t run_group() {
  std::vector<Op> ops; ops.swap(t_ops);
  if (ops.empty()) return ncclSuccess;
  // everything queued before the exchange on the streams it uses is complete before a buffer is exported
  for (const Op& o : ops) FK_HIP(hipStreamSynchronize(o.stream));
  int ks[MAXR] = {0}, kr[MAXR] = {0};
  struct Posted { Slot* s; uint64_t n; };
  std::vector<Posted> mine;
  for (const Op& o : ops) if (o.send) {            // 1. post every send
    Comm* m = o.comm; const int k = ks[o.peer]++;
    if (k >= MAXK) { fprintf(stderr, "[fake_rccl] more than %d sends to one peer in a group\n", MAXK); return ncclInvalidUsage; }
    Slot* s = &m->c->p2p[m->rank][o.peer][k];
    const uint64_t n = ++m->scount[o.peer][k];
    if (!wait_until([&] { return s->done == n - 1; })) { fprintf(stderr, "[fake_rccl] rank %d: the previous send to %d was never consumed\n", m->rank, o.peer); return ncclSystemError; }
    if (m->world > 2) {                              // several importers per buffer: stage through shared memory
      snprintf(s->shm, sizeof s->shm, "%s_x%d_%d_%d_%llu", m->name, m->rank, o.peer, k, (unsigned long long)n);
      const int fd = shm_open(s->shm, O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0 || ftruncate(fd, (off_t)(o.bytes ? o.bytes : 1)) != 0) { perror("[fake_rccl] shm staging"); if (fd >= 0) close(fd); return ncclSystemError; }
      void* hp = mmap(nullptr, o.bytes ? o.bytes : 1, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); close(fd);
      if (hp == MAP_FAILED) { perror("[fake_rccl] mmap staging"); return ncclSystemError; }
      FK_HIP(hipMemcpy(hp, o.buf, o.bytes, hipMemcpyDeviceToHost));
      munmap(hp, o.bytes ? o.bytes : 1);
      s->staged = 1; s->offset = 0; s->bytes = o.bytes;
    } else {
      hipDeviceptr_t base = nullptr; size_t span = 0;
      FK_HIP(hipMemGetAddressRange(&base, &span, (hipDeviceptr_t)o.buf));
      FK_HIP(hipIpcGetMemHandle(&s->h, base));
      s->staged = 0; s->offset = (uint64_t)((const char*)o.buf - (const char*)base); s->bytes = o.bytes;
    }
    __sync_synchronize(); s->posted = n; __sync_synchronize();
    mine.push_back({s, n});
  }
  std::vector<hipStream_t> rstreams;
  struct Got { Slot* s; uint64_t n; void* mapped; void* host; size_t host_bytes; };
  std::vector<Got> got;
  for (const Op& o : ops) if (!o.send) {           // 2. every receive: open the peer's allocation, copy on the caller's stream
    Comm* m = o.comm; const int k = kr[o.peer]++;
    if (k >= MAXK) return ncclInvalidUsage;
    Slot* s = &m->c->p2p[o.peer][m->rank][k];
    const uint64_t n = ++m->rcount[o.peer][k];
    if (!wait_until([&] { return s->posted == n; })) { fprintf(stderr, "[fake_rccl] rank %d: no matching send from %d (receive %d of the group)\n", m->rank, o.peer, k); return ncclSystemError; }
    if (s->bytes != o.bytes) { fprintf(stderr, "[fake_rccl] rank %d: receive of %zu bytes from %d meets a send of %llu\n", m->rank, o.bytes, o.peer, (unsigned long long)s->bytes); return ncclInvalidArgument; }
    void* mapped = nullptr; void* hp = nullptr;
    if (s->staged) {
      const int fd = shm_open(s->shm, O_RDWR, 0600);
      if (fd < 0) { perror("[fake_rccl] shm staging (open)"); return ncclSystemError; }
      hp = mmap(nullptr, o.bytes ? o.bytes : 1, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); close(fd);
      if (hp == MAP_FAILED) { perror("[fake_rccl] mmap staging (open)"); return ncclSystemError; }
      FK_HIP(hipMemcpyAsync(o.buf, hp, o.bytes, hipMemcpyHostToDevice, o.stream));
    } else {
      hipIpcMemHandle_t h; memcpy(&h, (const void*)&s->h, sizeof h);
      FK_HIP(hipIpcOpenMemHandle(&mapped, h, hipIpcMemLazyEnablePeerAccess));
      FK_HIP(hipMemcpyAsync(o.buf, (const char*)mapped + s->offset, o.bytes, hipMemcpyDeviceToDevice, o.stream));
    }
    rstreams.push_back(o.stream); got.push_back({s, n, mapped, hp, o.bytes ? o.bytes : 1});
  }
  for (hipStream_t st : rstreams) FK_HIP(hipStreamSynchronize(st));
  for (const Got& g : got) {
    if (g.mapped) FK_HIP(hipIpcCloseMemHandle(g.mapped));
    if (g.host) { munmap(g.host, g.host_bytes); shm_unlink(g.s->shm); }
    __sync_synchronize(); g.s->done = g.n;
  }
  __sync_synchronize();
  for (const Posted& p : mine)                      // 3. my buffers may be reused once the receivers have copied them
    if (!wait_until([&] { return p.s->done == p.n; })) { fprintf(stderr, "[fake_rccl] a send was never consumed\n"); return ncclSystemError; }
  return ncclSuccess;
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "/fake_rccl_%d_%llx", (int)getpid(), (unsigned long long)(now_s() * 1e6));
  const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) { perror("[fake_rccl] shm_open(create)"); return ncclSystemError; }
  if (ftruncate(fd, sizeof(Ctl)) != 0) { perror("[fake_rccl] ftruncate"); close(fd); return ncclSystemError; }      // zero-filled
  close(fd);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  Comm* m = new Comm();
  m->rank = rank; m->world = nranks; memcpy(m->name, id.internal, sizeof m->name - 1);
  const int fd = shm_open(m->name, O_RDWR, 0600);
  if (fd < 0) { perror("[fake_rccl] shm_open"); delete m; return ncclSystemError; }
  void* p = mmap(nullptr, sizeof(Ctl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { perror("[fake_rccl] mmap"); delete m; return ncclSystemError; }
  m->c = (Ctl*)p; m->c->world = nranks;
  __sync_add_and_fetch(&m->c->arrived, 1);
  if (!wait_until([&] { return m->c->arrived >= nranks; })) { fprintf(stderr, "[fake_rccl] rank %d: only %d of %d ranks arrived\n", rank, m->c->arrived, nranks); munmap(p, sizeof(Ctl)); delete m; return ncclSystemError; }
  *comm = (ncclComm_t)m;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm* m = (Comm*)comm;
  if (!m) return ncclSuccess;
  if (__sync_add_and_fetch(&m->c->left, 1) == m->world) shm_unlink(m->name);
  munmap((void*)m->c, sizeof(Ctl));
  delete m;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() { t_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
  if (t_depth <= 0) return ncclInvalidUsage;
  if (--t_depth > 0) return ncclSuccess;
  return run_group();
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  Comm* m = (Comm*)comm;
  if (!m || peer < 0 || peer >= m->world || peer == m->rank || !dt_size(datatype)) return ncclInvalidArgument;
  t_ops.push_back({true, (void*)sendbuff, count * dt_size(datatype), peer, m, stream});
  return t_depth > 0 ? ncclSuccess : run_group();
}
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  Comm* m = (Comm*)comm;
  if (!m || peer < 0 || peer >= m->world || peer == m->rank || !dt_size(datatype)) return ncclInvalidArgument;
  t_ops.push_back({false, recvbuff, count * dt_size(datatype), peer, m, stream});
  return t_depth > 0 ? ncclSuccess : run_group();
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
  Comm* m = (Comm*)comm;
  const size_t bytes = count * dt_size(datatype);
  if (!m || !dt_size(datatype) || bytes > (size_t)AR_BYTES) return ncclInvalidArgument;
  FK_HIP(hipStreamSynchronize(stream));
  FK_HIP(hipMemcpy((void*)m->c->ar[m->rank], sendbuff, bytes, hipMemcpyDeviceToHost));
  __sync_synchronize();
  if (!barrier(m)) return ncclSystemError;
  std::vector<unsigned char> acc(bytes ? bytes : 1);
  memcpy(acc.data(), (const void*)m->c->ar[0], bytes);
  for (int r = 1; r < m->world; r++) if (!reduce_any(acc.data(), (const void*)m->c->ar[r], count, datatype, op)) return ncclInvalidArgument;   // rank order: same bits on every rank
  if (!barrier(m)) return ncclSystemError;             // every rank has read the staging rows
  FK_HIP(hipMemcpy(recvbuff, acc.data(), bytes, hipMemcpyHostToDevice));
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "fake_rccl: success";
    case ncclUnhandledCudaError: return "fake_rccl: a HIP call failed (see stderr)";
    case ncclSystemError: return "fake_rccl: rendezvous failed or timed out (see stderr)";
    case ncclInvalidArgument: return "fake_rccl: invalid argument";
    case ncclInvalidUsage: return "fake_rccl: invalid usage";
    default: return "fake_rccl: error";
  }
}

// marker the tests read to be sure which library the product bound
int fake_rccl_marker(void) { return 355; }

}  // extern "C"
