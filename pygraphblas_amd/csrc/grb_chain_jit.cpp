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
// Floating-point types only (their operators are plain C expressions — the definitions of grb_ops.hpp: fmin / fmax, IEEE division, 0 / 1 for the
// comparisons); integer chains keep the interpreter, whose integer division and wrap-around rules live in grb_ops.hpp.  No hipRTC on the
// machine, or a compile error: the interpreter (or, for the two shapes of gap/prmark.py, the ahead-of-time kernel) runs, as before.  The two ahead-of-time
// shapes are compiled like any other chain at their second appearance — the PageRank loop runs through the general mechanism, the k_vec_chain<..., SPEC>
// kernels are what runs before that and without hipRTC.  GRB_MI355X_CHAIN_JIT=0 turns the compiler off, =2 compiles at the first sight, =3 keeps the
// ahead-of-time shapes (round 4's behaviour; tests/test_nonblocking_gpu.py compares them).
#include "grb_opcommon.hpp"
#include "grb_lazy.hpp"
#include <dlfcn.h>
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
  bool bind() {
    if (tried) return h != nullptr;
    tried = true;
    for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) { h = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    if (!h) return false;
#define GRB_RTC(N) N = (decltype(N))dlsym(h, "hiprtc" #N); if (!N) { h = nullptr; return false; }
    GRB_RTC(CreateProgram) GRB_RTC(CompileProgram) GRB_RTC(GetCodeSize) GRB_RTC(GetCode) GRB_RTC(GetProgramLogSize) GRB_RTC(GetProgramLog) GRB_RTC(DestroyProgram)
#undef GRB_RTC
    return true;
  }
};
Rtc g_rtc;

struct Entry { int uses = 0; bool failed = false; hipModule_t mod = nullptr; hipFunction_t fn = nullptr; };
std::map<std::string, Entry> g_cache;
std::mutex g_mu;
uint64_t g_stat_compiled = 0, g_stat_launched = 0;

int jit_mode() { const char* e = getenv("GRB_MI355X_CHAIN_JIT"); return e ? atoi(e) : 1; }      // (read per call: a test hook)

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
bool generate(const ChainLaunch& L, bool f32, int red /* 0 none, 1 in T, 2 FP32 widened to FP64 */, std::string& src) {
  std::ostringstream o;
  const int nin = L.next, nout = L.nout, ns = L.nsteps;
  const bool r32 = f32 && red != 2;
  o << "typedef " << (f32 ? "float" : "double") << " T; typedef " << (r32 ? "float" : "double") << " R;\n"
       "struct __attribute__((aligned(16))) P4 { T v[4]; }; struct __attribute__((aligned(4))) B4 { unsigned char v[4]; };\n"
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
      const char* e = un_expr(st.op); if (!e) return false;
      o << " const T x = " << val(st.src[0]) << "; const bool xp = " << has(st.src[0]) << "; const T z = " << e << "; ap[h] = xp; acc[h] = xp ? z : (T)0;";
    } else {
      const char* e = bin_expr(st.op, f32); if (!e) return false;
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
  if (red) { const char* e = red_expr(L.red.op, r32); if (!e) return false; o << "      if (h < nv && ap[h]) { const R a = racc, b = (R)acc[h]; racc = " << e << "; }\n"; }
  o << "    }\n";
  for (int k = 0; k < nout; k++) {
    o << "    if (nv == 4) { P4 t; B4 u;\n#pragma unroll\n      for (int h = 0; h < 4; h++) { t.v[h] = w" << k << "[h]; u.v[h] = g" << k << "[h] ? 1 : 0; }\n      *(P4*)(o" << k << " + base) = t; if (q" << k
      << ") *(B4*)(q" << k << " + base) = u;\n    } else { for (int h = 0; h < nv; h++) { o" << k << "[base + h] = w" << k << "[h]; if (q" << k << ") q" << k << "[base + h] = g" << k << "[h] ? 1 : 0; } }\n";
  }
  o << "  }\n";
  if (red) {
    const char* e = red_expr(L.red.op, r32);
    // lanes -> wave (a fixed butterfly) -> workgroup (its four waves in order) -> one partial per workgroup, as the interpreter leaves them
    o << "  __shared__ R sh[4];\n  for (int d = 32; d; d >>= 1) { const R a = racc, b = __shfl_xor(racc, d, 64); racc = " << e << "; }\n"
         "  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = racc;\n  __syncthreads();\n"
         "  if (threadIdx.x == 0) { R a = sh[0]; for (int w = 1; w < 4; w++) { const R b = sh[w]; a = " << e << "; } partial[blockIdx.x] = a; }\n";
  }
  o << "}\n";
  src = o.str();
  return true;
}

std::string signature(const ChainLaunch& L, bool f32, int red) {
  std::ostringstream k;
  k << (f32 ? 'f' : 'd') << red << ':' << L.next << ':' << L.nout << ':' << L.nsteps << ':' << (red ? L.red.op : -1);
  for (int i = 0; i < L.next; i++) k << (L.ep[i] ? 'b' : 'F');
  for (int s = 0; s < L.nsteps; s++) { const ChainStepDesc& st = L.st[s]; k << '|' << st.kind << ',' << st.op << ',' << st.mode << ',' << st.is_union << ',' << st.src[0] << ',' << st.src[1] << ',' << st.out; }
  return k.str();
}

}  // namespace

// Launches the chain through its compiled kernel when there is one (or when this call is the one that compiles it); false: the caller runs the
// interpreter.  `grid` workgroups of 256 threads; `partial` receives the per-workgroup partials of a reduction.
bool chain_jit_launch(const ChainLaunch& L, bool f32, int red, const void* rid, void* partial, unsigned grid, bool replaces_spec) {
  const int mode = jit_mode();
  if (mode == 0 || (replaces_spec && mode == 3) || !L.nsteps || L.math) return false;
  std::lock_guard<std::mutex> lk(g_mu);
  const std::string key = signature(L, f32, red);
  Entry& en = g_cache[key];
  en.uses++;
  if (en.failed) return false;
  if (!en.fn) {
    if (en.uses < 2 && mode != 2) return false;                    // a chain seen once is not worth a compilation (~0.3 s): the interpreter runs it
    std::string src;
    if (!g_rtc.bind() || !generate(L, f32, red, src)) { en.failed = true; return false; }
    hiprtcProgram prog = nullptr;
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off"};        // (no fused multiply-adds the ahead-of-time kernels would not form either: their steps are separate operator calls)
    bool ok = g_rtc.CreateProgram(&prog, src.c_str(), "grb_chain.hip", 0, nullptr, nullptr) == 0 && g_rtc.CompileProgram(prog, 3, opts) == 0;
    std::vector<char> code;
    if (ok) { size_t sz = 0; ok = g_rtc.GetCodeSize(prog, &sz) == 0 && sz > 0; if (ok) { code.resize(sz); ok = g_rtc.GetCode(prog, code.data()) == 0; } }
    else if (prog && getenv("GRB_MI355X_VERBOSE")) { size_t ls = 0; if (g_rtc.GetProgramLogSize(prog, &ls) == 0 && ls > 1) { std::vector<char> log(ls); g_rtc.GetProgramLog(prog, log.data()); fprintf(stderr, "grb chain jit: %s\n%s\n", log.data(), src.c_str()); } }
    if (prog) g_rtc.DestroyProgram(&prog);
    if (ok) ok = hipModuleLoadData(&en.mod, code.data()) == hipSuccess && hipModuleGetFunction(&en.fn, en.mod, "grb_chain") == hipSuccess;
    if (!ok) { (void)hipGetLastError(); en.failed = true; en.fn = nullptr; return false; }
    g_stat_compiled++;
  }
  // arguments in the order of the generated signature
  const void* ev[CHAIN_MAX_IN]; const uint8_t* ep[CHAIN_MAX_IN]; void* ov[CHAIN_MAX_OUT]; uint8_t* op[CHAIN_MAX_OUT];
  float s32[CHAIN_MAX_STEPS]; double s64[CHAIN_MAX_STEPS]; unsigned long long n = L.n; float rid32 = 0; double rid64 = 0;
  std::vector<void*> args;
  for (int k = 0; k < L.next; k++) { ev[k] = L.ev[k]; ep[k] = L.ep[k]; args.push_back(&ev[k]); args.push_back(&ep[k]); }
  for (int k = 0; k < L.nout; k++) { ov[k] = L.ov[k]; op[k] = L.op[k]; args.push_back(&ov[k]); args.push_back(&op[k]); }
  for (int s = 0; s < L.nsteps; s++) { if (f32) { memcpy(&s32[s], L.st[s].scalar, 4); args.push_back(&s32[s]); } else { memcpy(&s64[s], L.st[s].scalar, 8); args.push_back(&s64[s]); } }
  args.push_back(&n);
  if (f32 && red != 2) { if (rid) memcpy(&rid32, rid, 4); args.push_back(&rid32); } else { if (rid) memcpy(&rid64, rid, 8); args.push_back(&rid64); }
  args.push_back(&partial);
  if (hipModuleLaunchKernel(en.fn, grid, 1, 1, 256, 1, 1, 0, stream(), args.data(), nullptr) != hipSuccess) { (void)hipGetLastError(); en.failed = true; return false; }
  g_stat_launched++;
  return true;
}

}  // namespace grb

extern "C" GrB_Info GrBX_chain_jit_stats(uint64_t* compiled, uint64_t* launched) {
  if (compiled) *compiled = grb::g_stat_compiled; if (launched) *launched = grb::g_stat_launched;
  return GrB_SUCCESS;
}
