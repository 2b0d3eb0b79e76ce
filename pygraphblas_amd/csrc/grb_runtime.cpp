// grb_runtime.cpp — process-wide state: HIP device, stream, pooled HBM allocator, timers,
// and the built-in object registry.  (MI355X: one process drives one GPU; multi-GPU runs are
// one process per GPU, see pygraphblas_amd/dist.py.)
#include <atomic>
#include "grb_internal.hpp"
#include "grb_api.hpp"
#include "grb_lazy.hpp"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <unordered_map>

#include "registry_gen.inc"

static const uint64_t grb_all_sentinel = 0;
extern "C" const uint64_t* GrB_ALL = &grb_all_sentinel;

namespace grb {

static std::mutex& g_mu = *new std::mutex();      // (immortal, like the pool's registries below)
static bool g_inited = false, g_device_ok = false;
static std::string g_device_err = "GrB_init has not been called";
static hipStream_t g_stream = 0;
static hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;
static int g_cus = 0;
static int g_nthreads = 1; static double g_chunk = 65536; static int g_burble = 0;
static double g_hyper_switch = 0.0625; static int g_format = 0;
static double g_bitmap_switch[8] = {0.04, 0.05, 0.06, 0.08, 0.10, 0.20, 0.30, 0.40};
thread_local std::string g_last_plan;      // (per thread: "the last call" is the calling thread's)
thread_local std::string g_last_error;

// ---- pooled allocator: power-of-two-ish size classes, blocks are never split -------------------
// (never destroyed: the scratch buffers some kernels' launchers keep per thread — thread_local DevBufs — are released by destructors that run at thread or
//  process exit, possibly after this translation unit's statics are gone and after GrB_finalize returned the cached blocks; they only re-enter the free list)
static std::multimap<size_t, void*>& g_free = *new std::multimap<size_t, void*>();                  // size class -> block
static std::unordered_map<void*, size_t>& g_live = *new std::unordered_map<void*, size_t>();        // block -> size class
static size_t g_in_use = 0, g_cached = 0;

static size_t size_class(size_t n) {
  if (n < 512) return 512;
  // round up to a multiple of 1/8 of the enclosing power of two: <= 12.5 % slack
  size_t p = 1; while (p < n) p <<= 1;
  size_t step = p >> 4; if (step < 512) step = 512;
  return (n + step - 1) / step * step;
}

[[noreturn]] void fail(int info, const std::string& msg) { throw GrbError{info, msg}; }

bool device_ok() { return g_device_ok; }
const char* device_error() { return g_device_err.c_str(); }
hipStream_t stream() { return g_stream; }

void* dev_alloc(size_t bytes) {
  if (!g_device_ok) fail(GrB_PANIC, std::string("no HIP device available: ") + g_device_err);
  size_t sc = size_class(bytes);
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_free.find(sc);
  void* p = nullptr;
  if (it != g_free.end()) { p = it->second; g_free.erase(it); g_cached -= sc; }
  else {
    hipError_t e = hipMalloc(&p, sc);
    if (e != hipSuccess) {
      // drop the cache and retry once
      for (auto& kv : g_free) (void)hipFree(kv.second);
      g_free.clear(); g_cached = 0; (void)hipGetLastError();
      e = hipMalloc(&p, sc);
      if (e != hipSuccess) { (void)hipGetLastError(); fail(GrB_OUT_OF_MEMORY, "hipMalloc failed for " + std::to_string(sc) + " bytes"); }
    }
  }
  g_live[p] = sc; g_in_use += sc;
  return p;
}

uint64_t dev_alloc_serial() { static std::atomic<uint64_t> g_serial{0}; return ++g_serial; }

void dev_free(void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_live.find(p);
  if (it == g_live.end()) return;
  size_t sc = it->second; g_live.erase(it); g_in_use -= sc;
  // The pool keeps at most `cap` bytes of free blocks (round 6; default: half of the device's memory, GRB_MI355X_POOL_MAX_GB): a large block that would
  // take it beyond goes back to the driver at once.  (Unbounded, a process that had once formed a 43 GB product kept 285 of the 288 GB for good, and a second
  // process on the same GPU — a test's subprocess, another rank's — could not allocate at all.)  hipFree waits for the device: a block still read by
  // queued kernels is safe, and only blocks of >= 256 MB ever take this path.
  static const size_t cap = [] {
    const char* e = getenv("GRB_MI355X_POOL_MAX_GB"); if (e && *e) return (size_t)atof(e) * (size_t)(1ull << 30);
    size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); tot = 64ull << 30; }
    return tot / 2;
  }();
  if (sc >= (256ull << 20) && g_cached + sc > cap) { (void)hipFree(p); return; }
  // stream-ordered reuse is safe: every kernel runs on the one library stream
  g_free.emplace(sc, p); g_cached += sc;
}

void dev_pool_release() {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_device_ok) return;
  (void)hipStreamSynchronize(g_stream);
  for (auto& kv : g_free) (void)hipFree(kv.second);
  g_free.clear(); g_cached = 0;
}
size_t dev_bytes_in_use() { return g_in_use; }
size_t dev_bytes_cached() { return g_cached; }
int device_cus() { return g_cus; }

bool check_obj(const void* p) { return p && *(const uint64_t*)p == GRB_MAGIC; }
GrB_Type type_by_code(int code) {
  for (auto* t : all_types) if (t->code == code) return t;
  return nullptr;
}

// complex values exist as stored container entries only: real <-> complex casts follow the C rules
// SuiteSparse applies (real part kept, zero imaginary part added)
static void cast_complex(int dst_code, void* dst, int src_code, const void* src) {
  double re = 0, im = 0;
  if (src_code == T_FC32) { float f[2]; memcpy(f, src, 8); re = f[0]; im = f[1]; }
  else if (src_code == T_FC64) { double d[2]; memcpy(d, src, 16); re = d[0]; im = d[1]; }
  else if (src_code < T_FC32) cast_scalar(T_FP64, &re, src_code, src);
  else fail(GrB_DOMAIN_MISMATCH, "user-defined types are out of scope");
  if (dst_code == T_FC32) { float f[2] = {(float)re, (float)im}; memcpy(dst, f, 8); }
  else if (dst_code == T_FC64) { double d[2] = {re, im}; memcpy(dst, d, 16); }
  else if (dst_code == T_BOOL) { const uint8_t b = (re != 0 || im != 0); memcpy(dst, &b, 1); }
  else if (dst_code < T_FC32) cast_scalar(dst_code, dst, T_FP64, &re);
  else fail(GrB_DOMAIN_MISMATCH, "user-defined types are out of scope");
}

void cast_scalar(int dst_code, void* dst, int src_code, const void* src) {
  if (dst_code >= T_FC32 || src_code >= T_FC32) { cast_complex(dst_code, dst, src_code, src); return; }
  dispatch_type(src_code, [&]<class S>() {
    S s; memcpy(&s, src, sizeof(S));
    dispatch_type(dst_code, [&]<class D>() { D d = cast_to<D, S>(s); memcpy(dst, &d, sizeof(D)); });
  });
}

static void init_registry() {
  for (auto* t : all_types) t->size = (size_t)type_size(t->code);
  for (auto* m : all_monoids) {
    int code = m->op->ztype->code;
    dispatch_type(code, [&]<class T>() {
      T id = monoid_identity<T>(m->op->opcode); memcpy(m->identity, &id, sizeof(T));
      T term; if (monoid_terminal<T>(m->op->opcode, &term)) { m->has_terminal = true; memcpy(m->terminal, &term, sizeof(T)); }
    });
  }
}

static GrB_Info do_init() {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_inited) return GrB_SUCCESS;
  init_registry();
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) {
    (void)hipGetLastError();
    g_device_ok = false;
    g_device_err = e != hipSuccess ? hipGetErrorString(e) : "hipGetDeviceCount returned 0 devices";
  } else {
    int dev = 0;
    const char* lr = getenv("GRB_MI355X_DEVICE");
    if (lr) dev = atoi(lr); else { const char* l2 = getenv("LOCAL_RANK"); if (l2) dev = atoi(l2) % n; }
    e = hipSetDevice(dev);
    if (e == hipSuccess) e = hipEventCreate(&g_ev0);
    if (e == hipSuccess) e = hipEventCreate(&g_ev1);
    if (e == hipSuccess) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, dev) == hipSuccess) g_cus = pr.multiProcessorCount; }
    g_device_ok = (e == hipSuccess);
    if (!g_device_ok) { g_device_err = hipGetErrorString(e); (void)hipGetLastError(); }
    else g_device_err.clear();
  }
  g_inited = true;
  return GrB_SUCCESS;
}

}  // namespace grb

using namespace grb;

extern "C" {

GrB_Info GrB_init(int mode) { set_nonblocking(mode == 0 /* GrB_NONBLOCKING */); return do_init(); }
GrB_Info GxB_init(int mode, void* (*m)(size_t), void* (*c)(size_t, size_t), void* (*r)(void*, size_t),
                  void (*f)(void*), bool ts) { set_nonblocking(mode == 0); (void)m; (void)c; (void)r; (void)f; (void)ts; return do_init(); }
GrB_Info GrB_finalize(void) { if (g_device_ok) { try { lazy_flush(); } catch (...) {} } dev_pool_release(); return GrB_SUCCESS; }
GrB_Info GrB_getVersion(unsigned int* v, unsigned int* s) { if (v) *v = 1; if (s) *s = 3; return GrB_SUCCESS; }

GrB_Info GxB_Global_Option_set(int field, ...) {
  va_list ap; va_start(ap, field); GrB_Info info = GrB_SUCCESS;
  switch (field) {
    case 5: g_nthreads = va_arg(ap, int); break;
    case 7: g_chunk = va_arg(ap, double); break;
    case 99: g_burble = va_arg(ap, int); break;
    case 0: g_hyper_switch = va_arg(ap, double); break;
    case 1: g_format = va_arg(ap, int); break;
    case 34: { double* p = va_arg(ap, double*); if (p) for (int i = 0; i < 8; i++) g_bitmap_switch[i] = p[i]; break; }
    default: info = GrB_INVALID_VALUE;
  }
  va_end(ap); return info;
}
GrB_Info GxB_Global_Option_get(int field, ...) {
  va_list ap; va_start(ap, field); GrB_Info info = GrB_SUCCESS;
  switch (field) {
    case 5: { int* p = va_arg(ap, int*); if (p) *p = g_nthreads; break; }
    case 7: { double* p = va_arg(ap, double*); if (p) *p = g_chunk; break; }
    case 99: { bool* p = va_arg(ap, bool*); if (p) *p = g_burble != 0; break; }
    case 0: { double* p = va_arg(ap, double*); if (p) *p = g_hyper_switch; break; }
    case 1: { int* p = va_arg(ap, int*); if (p) *p = g_format; break; }
    case 34: { double* p = va_arg(ap, double*); if (p) for (int i = 0; i < 8; i++) p[i] = g_bitmap_switch[i]; break; }
    default: info = GrB_INVALID_VALUE;
  }
  va_end(ap); return info;
}

// ---- introspection ------------------------------------------------------------------------------
GrB_Info GxB_Semiring_add(GrB_Monoid* add, GrB_Semiring s) { if (!add) return GrB_NULL_POINTER; if (!check_obj(s)) return GrB_UNINITIALIZED_OBJECT; *add = s->add; return GrB_SUCCESS; }
GrB_Info GxB_Semiring_multiply(GrB_BinaryOp* mul, GrB_Semiring s) { if (!mul) return GrB_NULL_POINTER; if (!check_obj(s)) return GrB_UNINITIALIZED_OBJECT; *mul = s->mul; return GrB_SUCCESS; }
GrB_Info GxB_Monoid_operator(GrB_BinaryOp* op, GrB_Monoid m) { if (!op) return GrB_NULL_POINTER; if (!check_obj(m)) return GrB_UNINITIALIZED_OBJECT; *op = m->op; return GrB_SUCCESS; }
GrB_Info GxB_BinaryOp_ztype(GrB_Type* t, GrB_BinaryOp op) { if (!t) return GrB_NULL_POINTER; if (!check_obj(op)) return GrB_UNINITIALIZED_OBJECT; *t = op->ztype; return GrB_SUCCESS; }
GrB_Info GxB_BinaryOp_xtype(GrB_Type* t, GrB_BinaryOp op) { if (!t) return GrB_NULL_POINTER; if (!check_obj(op)) return GrB_UNINITIALIZED_OBJECT; *t = op->xtype; return GrB_SUCCESS; }
GrB_Info GxB_BinaryOp_ytype(GrB_Type* t, GrB_BinaryOp op) { if (!t) return GrB_NULL_POINTER; if (!check_obj(op)) return GrB_UNINITIALIZED_OBJECT; *t = op->ytype; return GrB_SUCCESS; }
GrB_Info GxB_UnaryOp_ztype(GrB_Type* t, GrB_UnaryOp op) { if (!t) return GrB_NULL_POINTER; if (!check_obj(op)) return GrB_UNINITIALIZED_OBJECT; *t = op->ztype; return GrB_SUCCESS; }
GrB_Info GxB_UnaryOp_xtype(GrB_Type* t, GrB_UnaryOp op) { if (!t) return GrB_NULL_POINTER; if (!check_obj(op)) return GrB_UNINITIALIZED_OBJECT; *t = op->xtype; return GrB_SUCCESS; }
GrB_Info GxB_Type_size(size_t* size, GrB_Type t) { if (!size) return GrB_NULL_POINTER; if (!check_obj(t)) return GrB_UNINITIALIZED_OBJECT; *size = t->size; return GrB_SUCCESS; }

static void print_value(FILE* f, int code, const void* p) {
  dispatch_type(code, [&]<class T>() {
    T v; memcpy(&v, p, sizeof(T));
    if constexpr (is_bool<T>::value) fprintf(f, "%d", (int)(bool)v);
    else if constexpr (std::is_floating_point<T>::value) fprintf(f, "%g", (double)v);
    else if constexpr (std::is_signed<T>::value) fprintf(f, "%lld", (long long)v);
    else fprintf(f, "%llu", (unsigned long long)v);
  });
}
static FILE* outf(FILE* f) { return f ? f : stdout; }
GrB_Info GxB_BinaryOp_fprint(GrB_BinaryOp op, const char* name, int pr, FILE* f) {
  if (!check_obj(op)) return GrB_UNINITIALIZED_OBJECT; if (pr <= 0) return GrB_SUCCESS;
  fprintf(outf(f), "\n    GraphBLAS BinaryOp: %s (built-in) z=%s(x,y)  x:%s y:%s z:%s\n", name ? name : "", op->name,
          op->xtype->name, op->ytype->name, op->ztype->name); return GrB_SUCCESS;
}
GrB_Info GxB_UnaryOp_fprint(GrB_UnaryOp op, const char* name, int pr, FILE* f) {
  if (!check_obj(op)) return GrB_UNINITIALIZED_OBJECT; if (pr <= 0) return GrB_SUCCESS;
  fprintf(outf(f), "\n    GraphBLAS UnaryOp: %s (built-in) z=%s(x)  x:%s z:%s\n", name ? name : "", op->name,
          op->xtype->name, op->ztype->name); return GrB_SUCCESS;
}
GrB_Info GxB_Monoid_fprint(GrB_Monoid m, const char* name, int pr, FILE* f) {
  if (!check_obj(m)) return GrB_UNINITIALIZED_OBJECT; if (pr <= 0) return GrB_SUCCESS;
  FILE* o = outf(f);
  fprintf(o, "\n    GraphBLAS Monoid: %s (%s) op %s identity: [ ", name ? name : "", m->builtin ? "built-in" : "user", m->op->name);
  print_value(o, m->op->ztype->code, m->identity); fprintf(o, " ]");
  if (m->has_terminal) { fprintf(o, " terminal: [ "); print_value(o, m->op->ztype->code, m->terminal); fprintf(o, " ]"); }
  fprintf(o, "\n"); return GrB_SUCCESS;
}
GrB_Info GxB_Semiring_fprint(GrB_Semiring s, const char* name, int pr, FILE* f) {
  if (!check_obj(s)) return GrB_UNINITIALIZED_OBJECT; if (pr <= 0) return GrB_SUCCESS;
  fprintf(outf(f), "\n    GraphBLAS Semiring: %s (%s) add %s multiply %s\n", name ? name : "", s->builtin ? "built-in" : "user",
          s->add->op->name, s->mul->name); return GrB_SUCCESS;
}
GrB_Info GxB_SelectOp_fprint(GxB_SelectOp op, const char* name, int pr, FILE* f) {
  if (!check_obj(op)) return GrB_UNINITIALIZED_OBJECT; if (pr <= 0) return GrB_SUCCESS;
  fprintf(outf(f), "\n    GraphBLAS SelectOp: %s: %s\n", name ? name : "", op->name); return GrB_SUCCESS;
}

// ---- descriptors ----------------------------------------------------------------------------------
GrB_Info GrB_Descriptor_new(GrB_Descriptor* d) {
  if (!d) return GrB_NULL_POINTER;
  *d = new GrB_Descriptor_opaque{GRB_MAGIC, 0, 0, 0, 0, 0, 0, 0, 0.0, false, "user"}; return GrB_SUCCESS;
}
static GrB_Info desc_set(GrB_Descriptor d, int field, int val) {
  if (!check_obj(d)) return GrB_UNINITIALIZED_OBJECT;
  if (d->builtin) return GrB_INVALID_VALUE;
  switch (field) {
    case GrB_OUTP: if (val != GxB_DEFAULT && val != GrB_REPLACE) return GrB_INVALID_VALUE; d->outp = val; break;
    case GrB_MASK:
      if (val == GxB_DEFAULT) d->mask = 0;
      else if (val == GrB_COMP || val == GrB_STRUCTURE) d->mask |= val;   // fields accumulate, as in SuiteSparse
      else if (val == GrB_COMP + GrB_STRUCTURE) d->mask = val;
      else return GrB_INVALID_VALUE;
      break;
    case GrB_INP0: if (val != GxB_DEFAULT && val != GrB_TRAN) return GrB_INVALID_VALUE; d->inp0 = val; break;
    case GrB_INP1: if (val != GxB_DEFAULT && val != GrB_TRAN) return GrB_INVALID_VALUE; d->inp1 = val; break;
    case GxB_AxB_METHOD: d->axb = val; break;
    case GxB_DESCRIPTOR_NTHREADS: d->nthreads = val; break;
    case GxB_SORT: d->sort = val; break;
    default: return GrB_INVALID_VALUE;
  }
  return GrB_SUCCESS;
}
GrB_Info GrB_Descriptor_set(GrB_Descriptor d, int field, int val) { return desc_set(d, field, val); }
GrB_Info GxB_Desc_set(GrB_Descriptor d, int field, ...) {
  va_list ap; va_start(ap, field); GrB_Info info;
  if (field == GxB_DESCRIPTOR_CHUNK) { if (!check_obj(d)) info = GrB_UNINITIALIZED_OBJECT; else { d->chunk = va_arg(ap, double); info = GrB_SUCCESS; } }
  else info = desc_set(d, field, va_arg(ap, int));
  va_end(ap); return info;
}
GrB_Info GxB_Desc_get(GrB_Descriptor d, int field, ...) {
  va_list ap; va_start(ap, field); GrB_Info info = GrB_SUCCESS;
  if (d && !check_obj(d)) { va_end(ap); return GrB_UNINITIALIZED_OBJECT; }
  if (field == GxB_DESCRIPTOR_CHUNK) { double* p = va_arg(ap, double*); if (p) *p = d ? d->chunk : 0; }
  else {
    int* p = va_arg(ap, int*); int v = 0;
    switch (field) {
      case GrB_OUTP: v = d ? d->outp : 0; break; case GrB_MASK: v = d ? d->mask : 0; break;
      case GrB_INP0: v = d ? d->inp0 : 0; break; case GrB_INP1: v = d ? d->inp1 : 0; break;
      case GxB_AxB_METHOD: v = d ? d->axb : 0; break; case GxB_DESCRIPTOR_NTHREADS: v = d ? d->nthreads : 0; break;
      case GxB_SORT: v = d ? d->sort : 0; break; default: info = GrB_INVALID_VALUE;
    }
    if (p) *p = v;
  }
  va_end(ap); return info;
}
GrB_Info GrB_Descriptor_free(GrB_Descriptor* d) {
  if (!d || !*d) return GrB_SUCCESS;
  if (check_obj(*d) && !(*d)->builtin) { (*d)->magic = GRB_FREED; delete *d; *d = nullptr; }
  return GrB_SUCCESS;   // predefined descriptors: no-op (reference frees them from __del__, descriptor.py:76-78)
}

// ---- algebra objects made of built-ins --------------------------------------------------------------
GrB_Info GrB_Semiring_new(GrB_Semiring* s, GrB_Monoid add, GrB_BinaryOp mul) {
  if (!s) return GrB_NULL_POINTER; if (!check_obj(add) || !check_obj(mul)) return GrB_UNINITIALIZED_OBJECT;
  if (mul->ztype != add->op->ztype) return GrB_DOMAIN_MISMATCH;
  auto* r = new GrB_Semiring_opaque{GRB_MAGIC, add, mul, "", false};
  snprintf(r->name, sizeof r->name, "user_%s_%s", add->op->name, mul->name); *s = r; return GrB_SUCCESS;
}
GrB_Info GrB_Semiring_free(GrB_Semiring* s) { if (s && *s && check_obj(*s) && !(*s)->builtin) { (*s)->magic = GRB_FREED; delete *s; *s = nullptr; } return GrB_SUCCESS; }
GrB_Info GrB_Monoid_free(GrB_Monoid* m) { if (m && *m && check_obj(*m) && !(*m)->builtin) { (*m)->magic = GRB_FREED; delete *m; *m = nullptr; } return GrB_SUCCESS; }

static GrB_Info monoid_new(GrB_Monoid* m, GrB_BinaryOp op, int code, const void* identity) {
  if (!m) return GrB_NULL_POINTER; if (!check_obj(op)) return GrB_UNINITIALIZED_OBJECT;
  if (op->xtype != op->ztype || op->ytype != op->ztype) return GrB_DOMAIN_MISMATCH;
  auto* r = new GrB_Monoid_opaque{GRB_MAGIC, op, {0}, false, {0}, "", false};
  cast_scalar(op->ztype->code, r->identity, code, identity);
  snprintf(r->name, sizeof r->name, "user_%s", op->name); *m = r; return GrB_SUCCESS;
}
#define GRB_MONOID_NEW(SUF, CT, CODE) \
  GrB_Info GrB_Monoid_new_##SUF(GrB_Monoid* m, GrB_BinaryOp op, CT id) { return monoid_new(m, op, CODE, &id); }
GRB_MONOID_NEW(BOOL, bool, T_BOOL) GRB_MONOID_NEW(INT8, int8_t, T_INT8) GRB_MONOID_NEW(UINT8, uint8_t, T_UINT8)
GRB_MONOID_NEW(INT16, int16_t, T_INT16) GRB_MONOID_NEW(UINT16, uint16_t, T_UINT16) GRB_MONOID_NEW(INT32, int32_t, T_INT32)
GRB_MONOID_NEW(UINT32, uint32_t, T_UINT32) GRB_MONOID_NEW(INT64, int64_t, T_INT64) GRB_MONOID_NEW(UINT64, uint64_t, T_UINT64)
GRB_MONOID_NEW(FP32, float, T_FP32) GRB_MONOID_NEW(FP64, double, T_FP64)

// ---- backend extensions -----------------------------------------------------------------------------
// (deferred vector operations — grb_lazy.cpp — are completed first: they belong to the work these calls order or measure)
static GrB_Info flush_deferred() { try { lazy_flush(); return GrB_SUCCESS; } catch (const GrbError& e) { g_last_error = e.msg; return e.info; } catch (...) { return GrB_PANIC; } }
GrB_Info GrBX_set_stream(void* s) { if (g_device_ok) { const GrB_Info i = flush_deferred(); if (i) return i; } g_stream = (hipStream_t)s; return GrB_SUCCESS; }
GrB_Info GrBX_device_synchronize(void) {
  if (!g_device_ok) return GrB_PANIC;
  const GrB_Info i = flush_deferred(); if (i) return i;
  return hipStreamSynchronize(g_stream) == hipSuccess ? GrB_SUCCESS : GrB_PANIC;
}
GrB_Info GrBX_timer_start(void) { if (!g_device_ok) return GrB_PANIC; return hipEventRecord(g_ev0, g_stream) == hipSuccess ? GrB_SUCCESS : GrB_PANIC; }
GrB_Info GrBX_timer_stop(float* ms) {
  if (!g_device_ok) return GrB_PANIC;
  { const GrB_Info i = flush_deferred(); if (i) return i; }
  if (hipEventRecord(g_ev1, g_stream) != hipSuccess) return GrB_PANIC;
  if (hipEventSynchronize(g_ev1) != hipSuccess) return GrB_PANIC;
  float t = 0; if (hipEventElapsedTime(&t, g_ev0, g_ev1) != hipSuccess) return GrB_PANIC;
  if (ms) *ms = t; return GrB_SUCCESS;
}
GrB_Info GrBX_device_info(char* name, int len, int* cus, size_t* hbm) {
  if (!g_device_ok) { if (name && len > 0) snprintf(name, len, "none (%s)", g_device_err.c_str()); if (cus) *cus = 0; if (hbm) *hbm = 0; return GrB_NO_VALUE; }
  hipDeviceProp_t p; int dev = 0; (void)hipGetDevice(&dev);
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return GrB_PANIC;
  if (name && len > 0) snprintf(name, len, "%s %s", p.name, p.gcnArchName);
  if (cus) *cus = p.multiProcessorCount; if (hbm) *hbm = p.totalGlobalMem; return GrB_SUCCESS;
}
GrB_Info GrBX_memory_in_use(size_t* b) { if (b) *b = g_in_use; return GrB_SUCCESS; }
GrB_Info GrBX_last_kernel_plan(char* buf, int len) { if (buf && len > 0) snprintf(buf, len, "%s", g_last_plan.c_str()); return GrB_SUCCESS; }

}  // extern "C"

// ---- the exact accumulators of the masked product's deterministic mode on the host (grb_exact.hpp): what the CPU suite checks against math.fsum ----------
#include "grb_exact.hpp"
extern "C" GrB_Info GrBX_exact_sum_host(const double* terms, uint64_t n, int significand_bits, double* sum, int* unit_exp_out) {
  if (!terms || !sum || (significand_bits != 53 && significand_bits != 24)) return GrB_NULL_POINTER;
  double bound = 0.0;
  for (uint64_t q = 0; q < n; q++) { const double a = fabs(terms[q]); if (!(a <= bound)) bound = a; }       // NaN sticks
  if (!(bound <= std::numeric_limits<double>::max())) return GrB_INVALID_VALUE;
  int H = 0; while ((1ull << H) <= n) H++;
  int u = (bound > 0.0 ? ilogb(bound) + 2 : -1074) + H - 126;
  if (u < -1074) u = -1074;               // (as k_row_unit_exp: no double has a bit below 2^-1074)
  unsigned long long lo = 0, hi = 0;
  for (uint64_t q = 0; q < n; q++) {
    unsigned long long xl, xh; grb::fx_from_double(terms[q], u, xl, xh);
    const unsigned long long old = lo; lo += xl; hi += xh + (lo < old ? 1ull : 0ull);
  }
  *sum = significand_bits == 53 ? grb::fx_to_fp<53>(lo, hi, u) : grb::fx_to_fp<24>(lo, hi, u);
  if (unit_exp_out) *unit_exp_out = u;
  return GrB_SUCCESS;
}
