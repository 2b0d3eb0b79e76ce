// grb_chain_jit.cpp — deferred element-wise chains compiled as they stand (round 5).
//
// The queue of grb_lazy.cpp runs as ONE kernel, k_vec_chain (grb_lazy_inst.hip) — an interpreter: per pack of four positions it walks the
// step descriptors through scalar compares and branches, ~100 instructions per position, bound by instruction issue and not by the HBM (2.5 and
// 4.9 TB/s for the two passes of a PageRank iteration at 2^25 vertices).  Round 4 compiled the two shapes of gap/prmark.py:21-26 ahead of time
// (k_vec_chain<..., SPEC>): fast, and benchmark-shaped.  This file is the general answer: a chain that is seen a second time is turned into the
// HIP source of a kernel that does exactly its steps — operators, operand slots, union / intersection and which operands are full are constants
// of the text, scalars and pointers stay arguments — compiled with hipRTC (bound at first use, like RCCL in grb_dist.cpp), cached by the
// chain's signature for the life of the process, and launched on the library's stream with the interpreter's grid.  Same loads (one 16-byte
// pack per operand and lane, the next pack in flight while this one is worked on), same stores, same per-workgroup partials of a reduction
// (the host folds them in index order, grb_lazy_inst.hip): only the steps differ — straight-line code instead of descriptor walks.
//
// Round 5 compiled floating-point chains only.  Round 6: the 4- and 8-byte INTEGER types as well — the rules of grb_ops.hpp that are not plain C travel as
// helper functions in the generated text (two's-complement wrap-around through the unsigned type, SuiteSparse's integer division: x / 0 saturates, 0 / 0 = 0,
// x / -1 = -x without the INT_MIN trap) — a CODE-OBJECT CACHE ON DISK (GRB_MI355X_CACHE_DIR, else $XDG_CACHE_HOME or ~/.cache, /grb_mi355x/chain-<hash>.co,
// keyed by the generated source + the device's architecture + the hipRTC version: a second process compiles nothing), and the ~0.3 s compilation runs
// OUTSIDE the table's lock (another thread's chains launch meanwhile; the same chain arriving during its own compilation takes the interpreter once more).
// No hipRTC on the machine, or a compile error: the interpreter (or, for the two shapes of gap/prmark.py, the ahead-of-time kernel) runs, as before.  The two ahead-of-time
// shapes are compiled like any other chain at their second appearance — the PageRank loop runs through the general mechanism, the k_vec_chain<..., SPEC>
// kernels are what runs before that and without hipRTC.  GRB_MI355X_CHAIN_JIT=0 turns the compiler off, =2 compiles at the first sight, =3 keeps the
// ahead-of-time shapes (round 4's behaviour; tests/test_nonblocking_gpu.py compares them).
#include "grb_opcommon.hpp"
#include "grb_lazy.hpp"
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <map>
#include <mutex>
#include <sstream>

namespace grb {
namespace {

typedef struct _hiprtcProgram* hiprtcProgram;
struct Rtc {
  void* h = nullptr; bool tried = false;
  int (*CreateProgram)(hiprtcProgram*, const char*, const char*, int, const char**, const char**) = nullptr;
  int (*CompileProgram)(hiprtcProgram, int, const char**) = nullptr;
  int (*GetCodeSize)(hiprtcProgram, size_t*) = nullptr;
  int (*GetCode)(hiprtcProgram, char*) = nullptr;
  int (*GetProgramLogSize)(hiprtcProgram, size_t*) = nullptr;
  int (*GetProgramLog)(hiprtcProgram, char*) = nullptr;
  int (*DestroyProgram)(hiprtcProgram*) = nullptr;
  int (*Version)(int*, int*) = nullptr;
  bool bind() {
    if (tried) return h != nullptr;
    tried = true;
    for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) { h = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) return false;
#define GRB_RTC(N) N = (decltype(N))dlsym(h, "hiprtc" #N); if (!N) { h = nullptr; return false; }
    GRB_RTC(CreateProgram) GRB_RTC(CompileProgram) GRB_RTC(GetCodeSize) GRB_RTC(GetCode) GRB_RTC(GetProgramLogSize) GRB_RTC(GetProgramLog) GRB_RTC(DestroyProgram)
#undef GRB_RTC
    Version = (decltype(Version))dlsym(h, "hiprtcVersion");          // (optional: part of the disk cache's key)
    return true;
  }
};
Rtc g_rtc;

struct Entry { int uses = 0; bool failed = false, compiling = false; hipModule_t mod = nullptr; hipFunction_t fn = nullptr; };
std::map<std::string, Entry> g_cache;          // (std::map: a reference to an entry stays valid while other entries come and go)
std::mutex g_mu;
std::atomic<uint64_t> g_stat_compiled{0}, g_stat_launched{0}, g_stat_from_disk{0};

// the value type of a chain: what the generated text needs to know about it
struct TypeInfo { const char* name; const char* uname; bool is_float, is_signed, f32; int size; const char* tmin; const char* tmax; };
bool type_info(int code, TypeInfo& t) {
  switch (code) {
    case T_FP32: t = {"float", "float", true, true, true, 4, "", ""}; return true;
    case T_FP64: t = {"double", "double", true, true, false, 8, "", ""}; return true;
    case T_INT32: t = {"int", "unsigned int", false, true, false, 4, "(-2147483647 - 1)", "2147483647"}; return true;
    case T_UINT32: t = {"unsigned int", "unsigned int", false, false, false, 4, "0u", "4294967295u"}; return true;
    case T_INT64: t = {"long long", "unsigned long long", false, true, false, 8, "(-9223372036854775807ll - 1)", "9223372036854775807ll"}; return true;
    case T_UINT64: t = {"unsigned long long", "unsigned long long", false, false, false, 8, "0ull", "18446744073709551615ull"}; return true;
    default: return false;
  }
}

int jit_mode() { const char* e = getenv("GRB_MI355X_CHAIN_JIT"); return e ? atoi(e) : 1; }      // (read per call: a test hook)

const char* bin_expr_int(int op) {                  // z = f(x, y) on an integer T: grb_ops.hpp apply_binop<T> (wrap_add / wrap_sub / wrap_mul / int_div)
  switch (op) {
    case B_FIRST: return "x"; case B_SECOND: case B_ANY: return "y"; case B_PAIR: return "(T)1";
    case B_MIN: return "(x < y ? x : y)"; case B_MAX: return "(x > y ? x : y)";
    case B_PLUS: return "(T)((U)x + (U)y)"; case B_MINUS: return "(T)((U)x - (U)y)"; case B_RMINUS: return "(T)((U)y - (U)x)"; case B_TIMES: return "(T)((U)x * (U)y)";
    case B_DIV: return "grb_idiv(x, y)"; case B_RDIV: return "grb_idiv(y, x)";
    case B_ISEQ: return "(T)(x == y)"; case B_ISNE: return "(T)(x != y)"; case B_ISGT: return "(T)(x > y)"; case B_ISLT: return "(T)(x < y)";
    case B_ISGE: return "(T)(x >= y)"; case B_ISLE: return "(T)(x <= y)";
    case B_LOR: return "(T)((x != 0) || (y != 0))"; case B_LAND: return "(T)((x != 0) && (y != 0))"; case B_LXOR: return "(T)((x != 0) != (y != 0))";
    default: return nullptr;
  }
}
const char* un_expr_int(int op, bool is_signed) {   // apply_unop<T> for an integer T
  switch (op) {
    case U_IDENTITY: return "x"; case U_AINV: return "(T)((U)0 - (U)x)"; case U_MINV: return "grb_idiv((T)1, x)"; case U_LNOT: return "(T)(x == 0)";
    case U_ONE: return "(T)1"; case U_ABS: return is_signed ? "(x < 0 ? (T)((U)0 - (U)x) : x)" : "x"; case U_BNOT: return "(T)~x";
    default: return nullptr;
  }
}
const char* red_expr_int(int op) {                  // the monoid on R = T
  switch (op) {
    case B_PLUS: return "(R)((U)a + (U)b)"; case B_TIMES: return "(R)((U)a * (U)b)"; case B_MIN: return "(a < b ? a : b)"; case B_MAX: return "(a > b ? a : b)";
    case B_LOR: return "(R)((a != 0) || (b != 0))"; case B_LAND: return "(R)((a != 0) && (b != 0))"; case B_LXOR: return "(R)((a != 0) != (b != 0))"; case B_ANY: return "b";
    default: return nullptr;
  }
}
const char* bin_expr(int op, bool f32) {           // z = f(x, y) on T: grb_ops.hpp apply_binop<T> for floating-point T
  switch (op) {
    case B_FIRST: return "x"; case B_SECOND: case B_ANY: return "y"; case B_PAIR: return "(T)1";
    case B_MIN: return f32 ? "fminf(x, y)" : "fmin(x, y)"; case B_MAX: return f32 ? "fmaxf(x, y)" : "fmax(x, y)";
    case B_PLUS: return "x + y"; case B_MINUS: return "x - y"; case B_RMINUS: return "y - x"; case B_TIMES: return "x * y";
    case B_DIV: return "x / y"; case B_RDIV: return "y / x";
    case B_ISEQ: return "(T)(x == y)"; case B_ISNE: return "(T)(x != y)"; case B_ISGT: return "(T)(x > y)"; case B_ISLT: return "(T)(x < y)";
    case B_ISGE: return "(T)(x >= y)"; case B_ISLE: return "(T)(x <= y)";
    case B_LOR: return "(T)((x != 0) || (y != 0))"; case B_LAND: return "(T)((x != 0) && (y != 0))"; case B_LXOR: return "(T)((x != 0) != (y != 0))";
    default: return nullptr;
  }
}
const char* un_expr(int op) {                       // apply_unop<T> for floating-point T
  switch (op) {
    case U_IDENTITY: case U_BNOT: return "x"; case U_AINV: return "(T)0 - x"; case U_MINV: return "(T)1 / x"; case U_LNOT: return "(T)(x == 0)";
    case U_ONE: return "(T)1"; case U_ABS: return "(T)fabs((double)x)";
    default: return nullptr;
  }
}
const char* red_expr(int op, bool f32) {            // the monoid on R (apply_binop<R>)
  switch (op) {
    case B_PLUS: return "a + b"; case B_TIMES: return "a * b"; case B_MIN: return f32 ? "fminf(a, b)" : "fmin(a, b)"; case B_MAX: return f32 ? "fmaxf(a, b)" : "fmax(a, b)";
    case B_LOR: return "(R)((a != 0) || (b != 0))"; case B_LAND: return "(R)((a != 0) && (b != 0))"; case B_LXOR: return "(R)((a != 0) != (b != 0))"; case B_ANY: return "b";
    default: return nullptr;
  }
}

// the chain as HIP source; false when a step has no expression here
bool generate(const ChainLaunch& L, const TypeInfo& ti, int red /* 0 none, 1 in T, 2 FP32 widened to FP64 */, std::string& src) {
  std::ostringstream o;
  const int nin = L.next, nout = L.nout, ns = L.nsteps;
  const bool f32 = ti.f32, isf = ti.is_float;
  const bool r32 = f32 && red != 2;
  if (!isf && red == 2) return false;
  o << "typedef " << ti.name << " T; typedef " << ti.uname << " U; typedef " << (isf ? (r32 ? "float" : "double") : ti.name) << " R;\n";
  if (!isf) {
    if (ti.is_signed) o << "__device__ inline T grb_idiv(T x, T y) { if (y == (T)-1) return (T)((U)0 - (U)x); if (y == 0) return x == 0 ? (T)0 : (x < 0 ? (T)" << ti.tmin << " : (T)" << ti.tmax << "); return (T)(x / y); }\n";
    else o << "__device__ inline T grb_idiv(T x, T y) { if (y == 0) return x == 0 ? (T)0 : (T)" << ti.tmax << "; return (T)(x / y); }\n";
  }
  o << "struct __attribute__((aligned(" << 4 * ti.size << "))) P4 { T v[4]; }; struct __attribute__((aligned(4))) B4 { unsigned char v[4]; };\n"
       "extern \"C\" __global__ void __launch_bounds__(256) grb_chain(";
  for (int k = 0; k < nin; k++) o << "const T* e" << k << ", const unsigned char* p" << k << ", ";
  for (int k = 0; k < nout; k++) o << "T* o" << k << ", unsigned char* q" << k << ", ";      // (no restrict: an output may be an operand's own buffers)
  for (int s = 0; s < ns; s++) o << "T sc" << s << ", ";
  o << "unsigned long long n, R rid, R* partial) {\n  R racc = rid;\n  const unsigned long long stride = (unsigned long long)gridDim.x * 1024ull;\n";
  for (int k = 0; k < nin; k++) { o << "  P4 ve" << k << ", vn" << k << ";"; if (L.ep[k]) o << " B4 vp" << k << ", vq" << k << ";"; o << "\n"; }
  o << "  const unsigned long long b0 = ((unsigned long long)blockIdx.x * 256ull + threadIdx.x) * 4ull;\n  if (b0 + 4 <= n) {";
  for (int k = 0; k < nin; k++) { o << " ve" << k << " = *(const P4*)(e" << k << " + b0);"; if (L.ep[k]) o << " vp" << k << " = *(const B4*)(p" << k << " + b0);"; }
  o << " }\n  for (unsigned long long base = b0; base < n; base += stride) {\n    const int nv = n - base >= 4ull ? 4 : (int)(n - base);\n";
  for (int k = 0; k < nin; k++) o << "    T x" << k << "[4]; bool h" << k << "[4];\n";
  o << "    if (nv == 4) {\n      const unsigned long long nb = base + stride, sb = nb + 4 <= n ? nb : base;\n";
  for (int k = 0; k < nin; k++) { o << "      vn" << k << " = *(const P4*)(e" << k << " + sb);"; if (L.ep[k]) o << " vq" << k << " = *(const B4*)(p" << k << " + sb);"; o << "\n"; }
  o << "#pragma unroll\n      for (int h = 0; h < 4; h++) {";
  for (int k = 0; k < nin; k++) { o << " x" << k << "[h] = ve" << k << ".v[h]; h" << k << "[h] = "; if (L.ep[k]) o << "vp" << k << ".v[h] != 0;"; else o << "true;"; }
  o << " }\n";
  for (int k = 0; k < nin; k++) { o << "      ve" << k << " = vn" << k << ";"; if (L.ep[k]) o << " vp" << k << " = vq" << k << ";"; o << "\n"; }
  o << "    } else {\n#pragma unroll\n      for (int h = 0; h < 4; h++) { const unsigned long long i = h < nv ? base + h : base;";
  for (int k = 0; k < nin; k++) { o << " x" << k << "[h] = e" << k << "[i]; h" << k << "[h] = "; if (L.ep[k]) o << "p" << k << "[i] != 0;"; else o << "true;"; }
  o << " }\n    }\n    T acc[4]; bool ap[4];\n";
  for (int k = 0; k < nout; k++) o << "    T w" << k << "[4]; bool g" << k << "[4];\n";
  o << "#pragma unroll\n    for (int h = 0; h < 4; h++) {\n      acc[h] = (T)0; ap[h] = false;\n";
  for (int k = 0; k < nout; k++) o << "      w" << k << "[h] = (T)0; g" << k << "[h] = false;\n";
  for (int s = 0; s < ns; s++) {
    const ChainStepDesc& st = L.st[s];
    auto val = [&](int slot) { std::ostringstream t; if (slot == CHAIN_PREV) t << "acc[h]"; else t << "x" << slot << "[h]"; return t.str(); };
    auto has = [&](int slot) { std::ostringstream t; if (slot == CHAIN_PREV) t << "ap[h]"; else t << "h" << slot << "[h]"; return t.str(); };
    o << "      {";
    if (st.kind == 1 && st.mode == 0) {
      const char* e = isf ? un_expr(st.op) : un_expr_int(st.op, ti.is_signed); if (!e) return false;
      o << " const T x = " << val(st.src[0]) << "; const bool xp = " << has(st.src[0]) << "; const T z = " << e << "; ap[h] = xp; acc[h] = xp ? z : (T)0;";
    } else {
      const char* e = isf ? bin_expr(st.op, f32) : bin_expr_int(st.op); if (!e) return false;
      if (st.kind == 0) {
        o << " const T x = " << val(st.src[0]) << ", y = " << val(st.src[1]) << "; const bool xp = " << has(st.src[0]) << ", yp = " << has(st.src[1]) << "; const T z = " << e << ";";
        if (st.is_union) o << " const bool zp = xp || yp; const T v = (xp && yp) ? z : (xp ? x : y);";
        else o << " const bool zp = xp && yp; const T v = z;";
        o << " ap[h] = zp; acc[h] = zp ? v : (T)0;";
      } else if (st.mode == 1) {          // z = f(s, x): the scalar takes the first seat
        o << " const T y = " << val(st.src[0]) << ", x = sc" << s << "; const bool yp = " << has(st.src[0]) << "; const T z = " << e << "; ap[h] = yp; acc[h] = yp ? z : (T)0;";
      } else {                            // z = f(x, s)
        o << " const T x = " << val(st.src[0]) << ", y = sc" << s << "; const bool xp = " << has(st.src[0]) << "; const T z = " << e << "; ap[h] = xp; acc[h] = xp ? z : (T)0;";
      }
    }
    if (st.out >= 0 && st.out < nout) o << " w" << st.out << "[h] = acc[h]; g" << st.out << "[h] = ap[h];";
    o << " }\n";
  }
  if (red) { const char* e = isf ? red_expr(L.red.op, r32) : red_expr_int(L.red.op); if (!e) return false; o << "      if (h < nv && ap[h]) { const R a = racc, b = (R)acc[h]; racc = " << e << "; }\n"; }
  o << "    }\n";
  for (int k = 0; k < nout; k++) {
    o << "    if (nv == 4) { P4 t; B4 u;\n#pragma unroll\n      for (int h = 0; h < 4; h++) { t.v[h] = w" << k << "[h]; u.v[h] = g" << k << "[h] ? 1 : 0; }\n      *(P4*)(o" << k << " + base) = t; if (q" << k
      << ") *(B4*)(q" << k << " + base) = u;\n    } else { for (int h = 0; h < nv; h++) { o" << k << "[base + h] = w" << k << "[h]; if (q" << k << ") q" << k << "[base + h] = g" << k << "[h] ? 1 : 0; } }\n";
  }
  o << "  }\n";
  if (red) {
    const char* e = isf ? red_expr(L.red.op, r32) : red_expr_int(L.red.op);
    // lanes -> wave (a fixed butterfly) -> workgroup (its four waves in order) -> one partial per workgroup, as the interpreter leaves them
    o << "  __shared__ R sh[4];\n  for (int d = 32; d; d >>= 1) { const R a = racc, b = __shfl_xor(racc, d, 64); racc = " << e << "; }\n"
         "  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = racc;\n  __syncthreads();\n"
         "  if (threadIdx.x == 0) { R a = sh[0]; for (int w = 1; w < 4; w++) { const R b = sh[w]; a = " << e << "; } partial[blockIdx.x] = a; }\n";
  }
  o << "}\n";
  src = o.str();
  return true;
}

std::string signature(const ChainLaunch& L, int tcode, int red) {
  std::ostringstream k;
  k << 't' << tcode << 'r' << red << ':' << L.next << ':' << L.nout << ':' << L.nsteps << ':' << (red ? L.red.op : -1);
  for (int i = 0; i < L.next; i++) k << (L.ep[i] ? 'b' : 'F');
  for (int s = 0; s < L.nsteps; s++) { const ChainStepDesc& st = L.st[s]; k << '|' << st.kind << ',' << st.op << ',' << st.mode << ',' << st.is_union << ',' << st.src[0] << ',' << st.src[1] << ',' << st.out; }
  return k.str();
}

}  // namespace

// ---- the code-object cache on disk ------------------------------------------------------------------------------------------
static uint64_t fnv1a(const std::string& t) { uint64_t h = 1469598103934665603ull; for (unsigned char c : t) { h ^= c; h *= 1099511628211ull; } return h; }
static std::string cache_dir() {
  const char* e = getenv("GRB_MI355X_CACHE_DIR");
  std::string d;
  if (e && *e) d = e;
  else { const char* x = getenv("XDG_CACHE_HOME"); const char* h = getenv("HOME"); if (x && *x) d = std::string(x) + "/grb_mi355x"; else if (h && *h) d = std::string(h) + "/.cache/grb_mi355x"; else return std::string(); }
  if (d == "off" || d == "0") return std::string();
  // mkdir -p (two levels are enough for ~/.cache/grb_mi355x)
  const size_t cut = d.find_last_of('/'); if (cut != std::string::npos && cut > 0) (void)mkdir(d.substr(0, cut).c_str(), 0700);
  if (mkdir(d.c_str(), 0700) != 0 && errno != EEXIST) return std::string();
  return d;
}
static const std::string& device_arch() {          // "gfx950" of the device the library runs on (hipGetDeviceProperties: gcnArchName up to its feature flags)
  static std::string arch = [] {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return std::string("gfx950"); }
    std::string a = prop.gcnArchName; const size_t c = a.find(':'); if (c != std::string::npos) a.resize(c);
    return a.empty() ? std::string("gfx950") : a;
  }();
  return arch;
}
static bool read_file(const std::string& path, std::vector<char>& out) {
  FILE* f = fopen(path.c_str(), "rb"); if (!f) return false;
  fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
  bool ok = n > 0; if (ok) { out.resize((size_t)n); ok = fread(out.data(), 1, (size_t)n, f) == (size_t)n; }
  fclose(f); return ok;
}
static void write_file_atomically(const std::string& path, const std::vector<char>& data) {
  const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  FILE* f = fopen(tmp.c_str(), "wb"); if (!f) return;
  const bool ok = fwrite(data.data(), 1, data.size(), f) == data.size();
  fclose(f);
  if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)unlink(tmp.c_str());
}

// source -> loaded function (from the disk cache, else through hipRTC); runs WITHOUT the table's lock
static bool build_kernel(const std::string& src, hipModule_t* mod, hipFunction_t* fn, bool* from_disk) {
  *from_disk = false; *mod = nullptr; *fn = nullptr;
  const std::string& arch = device_arch();
  int vmaj = 0, vmin = 0; if (g_rtc.Version) (void)g_rtc.Version(&vmaj, &vmin);
  const std::string dir = cache_dir();
  char name[64]; snprintf(name, sizeof(name), "/chain-%016llx.co", (unsigned long long)fnv1a(src + "|" + arch + "|" + std::to_string(vmaj) + "." + std::to_string(vmin) + "|O3,fp-contract=off"));
  const std::string path = dir.empty() ? std::string() : dir + name;
  std::vector<char> code;
  if (!path.empty() && read_file(path, code)) {
    if (hipModuleLoadData(mod, code.data()) == hipSuccess && hipModuleGetFunction(fn, *mod, "grb_chain") == hipSuccess) { *from_disk = true; return true; }
    (void)hipGetLastError(); if (*mod) { (void)hipModuleUnload(*mod); *mod = nullptr; } *fn = nullptr;      // (a stale or truncated file: compile and overwrite it)
    code.clear();
  }
  hiprtcProgram prog = nullptr;
  const std::string archopt = "--offload-arch=" + arch;
  const char* opts[] = {archopt.c_str(), "-O3", "-ffp-contract=off"};        // (no fused multiply-adds the ahead-of-time kernels would not form either: their steps are separate operator calls)
  bool ok = g_rtc.CreateProgram(&prog, src.c_str(), "grb_chain.hip", 0, nullptr, nullptr) == 0 && g_rtc.CompileProgram(prog, 3, opts) == 0;
  if (ok) { size_t sz = 0; ok = g_rtc.GetCodeSize(prog, &sz) == 0 && sz > 0; if (ok) { code.resize(sz); ok = g_rtc.GetCode(prog, code.data()) == 0; } }
  else if (prog && getenv("GRB_MI355X_VERBOSE")) { size_t ls = 0; if (g_rtc.GetProgramLogSize(prog, &ls) == 0 && ls > 1) { std::vector<char> log(ls); g_rtc.GetProgramLog(prog, log.data()); fprintf(stderr, "grb chain jit: %s\n%s\n", log.data(), src.c_str()); } }
  if (prog) g_rtc.DestroyProgram(&prog);
  if (ok) {
    ok = hipModuleLoadData(mod, code.data()) == hipSuccess && hipModuleGetFunction(fn, *mod, "grb_chain") == hipSuccess;
    if (!ok) { (void)hipGetLastError(); if (*mod) { (void)hipModuleUnload(*mod); *mod = nullptr; } *fn = nullptr; }      // (the module was leaked when only the lookup failed)
  }
  if (ok && !path.empty()) write_file_atomically(path, code);
  return ok;
}

// Launches the chain through its compiled kernel when there is one (or when this call is the one that compiles it); false: the caller runs the
// interpreter.  `grid` workgroups of 256 threads; `partial` receives the per-workgroup partials of a reduction.  `tcode`: the chain's value type.
bool chain_jit_launch(const ChainLaunch& L, int tcode, int red, const void* rid, void* partial, unsigned grid, bool replaces_spec) {
  const int mode = jit_mode();
  if (mode == 0 || (replaces_spec && mode == 3) || !L.nsteps || L.math) return false;
  TypeInfo ti; if (!type_info(tcode, ti)) return false;
  const std::string key = signature(L, tcode, red);
  hipFunction_t fn = nullptr;
  {
    std::unique_lock<std::mutex> lk(g_mu);
    Entry& en = g_cache[key];
    en.uses++;
    if (en.failed || en.compiling) return false;                     // (compiling: another thread is at it — this call takes the interpreter once more)
    if (!en.fn) {
      if (en.uses < 2 && mode != 2) return false;                    // a chain seen once is not worth a compilation (~0.3 s): the interpreter runs it
      std::string src;
      if (!g_rtc.bind() || !generate(L, ti, red, src)) { en.failed = true; return false; }
      en.compiling = true;
      lk.unlock();                                                   // the compilation (or the read of its cached code object) holds no lock
      hipModule_t mod = nullptr; hipFunction_t f = nullptr; bool from_disk = false;
      const bool ok = build_kernel(src, &mod, &f, &from_disk);
      lk.lock();
      Entry& en2 = g_cache[key];
      en2.compiling = false;
      if (!ok) { en2.failed = true; return false; }
      en2.mod = mod; en2.fn = f;
      if (from_disk) g_stat_from_disk++; else g_stat_compiled++;
    }
    fn = g_cache[key].fn;
  }
  // arguments in the order of the generated signature (scalars and the reduction's identity as raw words of the type's size)
  const void* ev[CHAIN_MAX_IN]; const uint8_t* ep[CHAIN_MAX_IN]; void* ov[CHAIN_MAX_OUT]; uint8_t* op[CHAIN_MAX_OUT];
  uint32_t s32[CHAIN_MAX_STEPS]; uint64_t s64[CHAIN_MAX_STEPS]; unsigned long long n = L.n; uint32_t rid32 = 0; uint64_t rid64 = 0;
  std::vector<void*> args;
  for (int k = 0; k < L.next; k++) { ev[k] = L.ev[k]; ep[k] = L.ep[k]; args.push_back(&ev[k]); args.push_back(&ep[k]); }
  for (int k = 0; k < L.nout; k++) { ov[k] = L.ov[k]; op[k] = L.op[k]; args.push_back(&ov[k]); args.push_back(&op[k]); }
  for (int s = 0; s < L.nsteps; s++) { if (ti.size == 4) { memcpy(&s32[s], L.st[s].scalar, 4); args.push_back(&s32[s]); } else { memcpy(&s64[s], L.st[s].scalar, 8); args.push_back(&s64[s]); } }
  args.push_back(&n);
  const bool rsmall = ti.size == 4 && red != 2;
  if (rsmall) { if (rid) memcpy(&rid32, rid, 4); args.push_back(&rid32); } else { if (rid) memcpy(&rid64, rid, 8); args.push_back(&rid64); }
  args.push_back(&partial);
  if (hipModuleLaunchKernel(fn, grid, 1, 1, 256, 1, 1, 0, stream(), args.data(), nullptr) != hipSuccess) {
    (void)hipGetLastError(); std::lock_guard<std::mutex> lk(g_mu); g_cache[key].failed = true; return false;
  }
  g_stat_launched++;
  return true;
}

}  // namespace grb

extern "C" GrB_Info GrBX_chain_jit_stats(uint64_t* compiled, uint64_t* launched) {
  if (compiled) *compiled = grb::g_stat_compiled.load(); if (launched) *launched = grb::g_stat_launched.load();
  return GrB_SUCCESS;
}
extern "C" GrB_Info GrBX_chain_jit_stats2(uint64_t* compiled, uint64_t* launched, uint64_t* loaded_from_disk) {
  if (compiled) *compiled = grb::g_stat_compiled.load(); if (launched) *launched = grb::g_stat_launched.load(); if (loaded_from_disk) *loaded_from_disk = grb::g_stat_from_disk.load();
  return GrB_SUCCESS;
}
