// grb_spmv.hip — the SpMV / SpMSpV kernels behind GrB_mxv and GrB_vxm  (HBM-bound; no MFMA).
//
//   t(i) = (+)_j  mult(M(i,j), u(j))      M in CSR (u32 rowptr/col, sorted rows), u and t bitmap
//
// Three hand-written kernels, all wave64:
//  A  k_spmv_adaptive   row-block ("CSR-adaptive" style) pull kernel — the FP64 PLUS_TIMES
//                       north-star path.  A block of 256 threads owns a run of consecutive rows
//                       holding <= 2048 entries: col/val are read fully coalesced (every lane a
//                       consecutive entry, 8 independent loads in flight per lane before the
//                       first use), u is gathered (n*8 B = 32 MiB at R-MAT-22: L2/Infinity-Cache
//                       resident), products are staged in LDS and each row is reduced from LDS
//                       in a fixed left-to-right order (deterministic).  Rows longer than a
//                       block are split into 8192-entry parts; the last part to arrive combines
//                       the partials in part order (agent-scope fence + ticket, guide §6 G16).
//  B  k_spmv_rowgroup   G lanes per row with mask skip and monoid-terminal early exit — the
//                       pull step of BFS (complemented visited mask, LOR terminal).
//  C  k_spmspv_push     frontier-driven scatter with atomics — the push step when u is sparse.
// Algorithmic bytes per call: nnz*(sizeof(T)+4) + (nrows+1)*4 + ncols*T + nrows*T.
#include "grb_api.hpp"
#include "grb_device.hpp"
#include "grb_semiring.hpp"
#include "grb_spmv.hpp"
#include <vector>

namespace grb {

float g_xcd_plan_build_ms = 0;

// per-type kernel instantiations live in grb_spmv_inst.hip (compiled once per value type, in parallel)
template <class T> void run_pull(const SpmvCall& c, const SemiringDesc& d);
template <class T> void run_push(const SpmvCall& c, const SemiringDesc& d, const uint32_t* fidx, uint64_t u_nvals, uint32_t* longlist);

// ---- plan: greedy row blocking, once per matrix -------------------------------------------------------------
void spmv_build_plan(DevCSR& M) {
  if (M.has_plan) return;
  std::vector<uint32_t> rp((size_t)M.nrows + 1);
  GRB_HIP(hipMemcpyAsync(rp.data(), M.rowptr.p, rp.size() * 4, hipMemcpyDeviceToHost, stream()));
  GRB_HIP(hipStreamSynchronize(stream()));
  std::vector<SpmvBlock> blocks; blocks.reserve(M.nnz / SPMV_NNZ * 2 + 16);
  uint32_t nslots = 0, nlong = 0;
  uint32_t r = 0; const uint32_t n = M.nrows;
  while (r < n) {
    const uint32_t len = rp[r + 1] - rp[r];
    if (len > (uint32_t)SPMV_NNZ) {
      const uint32_t parts = (len + SPMV_LONG_CHUNK - 1) / SPMV_LONG_CHUNK;
      for (uint32_t p = 0; p < parts; p++) blocks.push_back({r, p, parts, nslots});
      if (parts > 1) { nslots += parts; nlong++; }
      r++;
    } else {
      uint32_t e = r, tot = 0;
      while (e < n && e - r < (uint32_t)SPMV_MAX_ROWS) {
        const uint32_t l = rp[e + 1] - rp[e];
        if (tot + l > (uint32_t)SPMV_NNZ) break;
        tot += l; e++;
      }
      blocks.push_back({r, e, 0, 0});
      r = e;
    }
  }
  M.plan_nblocks = (uint32_t)blocks.size(); M.plan_nlong = nslots;
  M.plan_blocks.alloc(blocks.size() * sizeof(SpmvBlock) + 16);
  GRB_HIP(hipMemcpyAsync(M.plan_blocks.p, blocks.data(), blocks.size() * sizeof(SpmvBlock), hipMemcpyHostToDevice, stream()));
  // aux: [partials 8 B * nslots] [tickets u32 * nslots] [partial flags u8 * nslots]  (tickets indexed by `slot`; the 8-byte
  // partials come first so that they are 8-byte aligned whatever nslots is)
  const size_t aux = (size_t)(nslots + 1) * 8 + (size_t)(nslots + 1) * 4 + (size_t)(nslots + 1);
  M.plan_aux.alloc(aux);
  GRB_HIP(hipMemsetAsync(M.plan_aux.p, 0, aux, stream()));
  GRB_HIP(hipStreamSynchronize(stream()));
  M.has_plan = true;
}

bool spmv_rowlane_applies(const DevCSR& M, const SemiringDesc& d, int method) {
  if (method != SPMV_AUTO || !d.has_terminal) return false;
  const double avg = M.nrows ? (double)M.nnz / M.nrows : 0;
  static const bool no_lane = getenv("GRB_MI355X_NO_ROWLANE") && atoi(getenv("GRB_MI355X_NO_ROWLANE")) != 0;
  return avg <= 96 && !no_lane;
}

void spmv_pull(const SpmvCall& c, const SemiringDesc& d) {
  dispatch_type(d.zcode, [&]<class T>() { run_pull<T>(c, d); });
  GRB_HIP(hipGetLastError());
}

bool spmspv_push_supported(const SemiringDesc& d) {
  const int ts = type_size(d.zcode);
  if (d.zcode == T_BOOL) return d.addop == B_LOR || d.addop == B_ANY || d.addop == B_PLUS || d.addop == B_MAX;
  if (ts < 4) return false;
  if (d.addop == B_PLUS && (d.zcode == T_FP32 || d.zcode == T_FP64) && deterministic_env()) return false;      // floating-point sums by atomics: the pull kernels add in a fixed order
  return d.addop == B_PLUS || d.addop == B_MIN || d.addop == B_MAX;
}

// gather the indices of present entries of a bitmap vector (ascending) -> fidx, returns count
__global__ void k_flag_to_u32(const uint8_t* __restrict__ pres, uint64_t n, uint32_t* __restrict__ out) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) out[i] = pres[i] ? 1u : 0u;
}
__global__ void k_compact_idx(const uint8_t* __restrict__ pres, const uint32_t* __restrict__ pos, uint64_t n, uint32_t* __restrict__ out) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) if (pres[i]) out[pos[i]] = (uint32_t)i;
}

void spmspv_push(const SpmvCall& c, const SemiringDesc& d, uint64_t u_nvals) {
  // c.M here is the CSR whose ROWS are indexed like u (i.e. the transpose of the pull operand)
  DevCSR& M = *c.M; const uint64_t nu = M.nrows, nout = M.ncols;
  auto grid_of = [](uint64_t n) { uint64_t b = (n + 255) / 256; if (b < 1) b = 1; if (b > 4096) b = 4096; return (unsigned)b; };
  DevBuf flags, pos, fidx((u_nvals + 1) * 4), longlist((u_nvals + 2) * 4);
  if (c.small_idx && c.small_n == u_nvals && u_nvals <= 64) {
    // the frontier is known on the host: its list travels as kernel arguments of run_push's first launch (k_push_init), which also clears
    // the presence bytes and the hub-row counter
  } else {
    GRB_HIP(hipMemsetAsync(longlist.p, 0, 4, stream()));
    flags.alloc(nu * 4 + 4); pos.alloc(nu * 4 + 4);
    hipLaunchKernelGGL(k_flag_to_u32, dim3(grid_of(nu)), dim3(256), 0, stream(), c.upres, nu, flags.as<uint32_t>());
    exclusive_scan_u32(flags.as<uint32_t>(), pos.as<uint32_t>(), nu);
    hipLaunchKernelGGL(k_compact_idx, dim3(grid_of(nu)), dim3(256), 0, stream(), c.upres, pos.as<uint32_t>(), nu, fidx.as<uint32_t>());
    GRB_HIP(hipMemsetAsync(c.tpres, 0, nout ? nout : 1, stream()));
  }
  dispatch_type(d.zcode, [&]<class T>() { run_push<T>(c, d, fidx.as<uint32_t>(), u_nvals, longlist.as<uint32_t>()); });
  GRB_HIP(hipGetLastError());
}

}  // namespace grb

// ---- how a full-chip launch maps workgroups to XCDs ------------------------------------------------------------------------
// Kernel X gives workgroup b the column panel b % 8 and is fast when that workgroup runs on XCD b % 8 (every XCD's L2 then
// holds one eighth of the operand).  The round-robin dispatch is observed, not documented, and other partition modes deal
// differently — so it is probed once (XCC_ID of every workgroup of a one-workgroup-per-CU launch) and reported with the plan.
namespace grb {
static __global__ __launch_bounds__(1024) void k_xcc_probe(uint32_t* __restrict__ out) {
  uint32_t x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = x & 0xFu;
}
const std::string& xcd_mapping() {
  static std::string res;
  static const bool reprobe = getenv("GRB_MI355X_XCD_REPROBE") != nullptr;      // measurement hook: probe at every call (tools/xcd_probe.py)
  if (reprobe) res.clear();
  if (!res.empty() || !device_ok()) return res;
  const int ncu = device_cus();
  if (ncu <= 0) { res = "unknown"; return res; }
  DevBuf d((size_t)ncu * 4);
  hipLaunchKernelGGL(k_xcc_probe, dim3(ncu), dim3(1024), 0, stream(), d.as<uint32_t>());
  std::vector<uint32_t> h(ncu);
  if (hipMemcpyAsync(h.data(), d.p, (size_t)ncu * 4, hipMemcpyDeviceToHost, stream()) != hipSuccess || hipStreamSynchronize(stream()) != hipSuccess) { (void)hipGetLastError(); res = "unknown"; return res; }
  int match = 0; uint32_t seen = 0;
  for (int b = 0; b < ncu; b++) { match += (h[b] == (uint32_t)(b & 7)); seen |= 1u << h[b]; }
  const int nx = __builtin_popcount(seen);
  if (match == ncu) { res = "roundrobin8"; return res; }
  // a rotated deal (workgroup b on XCD (b + r) % 8: the dispatcher started at another XCD) serves kernel X as well: every panel still
  // has one XCD to itself for the launch
  for (int r = 1; r < 8; r++) {
    int m = 0; for (int b = 0; b < ncu; b++) m += (h[b] == (uint32_t)((b + r) & 7));
    if (m == ncu) { res = "roundrobin8+" + std::to_string(r); return res; }
  }
  res = "xcds=" + std::to_string(nx) + ",workgroups_on_xcd_b%8=" + std::to_string(match) + "/" + std::to_string(ncu);
  return res;
}
}  // namespace grb

extern "C" GrB_Info GrBX_last_plan_build_ms(float* ms) { if (!ms) return GrB_NULL_POINTER; *ms = grb::g_xcd_plan_build_ms; return GrB_SUCCESS; }
extern "C" GrB_Info GrBX_xcd_mapping(char* buf, int len) {
  if (!buf || len <= 0) return GrB_NULL_POINTER;
  if (!grb::device_ok()) { snprintf(buf, len, "no device"); return GrB_NO_VALUE; }
  try { snprintf(buf, len, "%s", grb::xcd_mapping().c_str()); } catch (...) { snprintf(buf, len, "unknown"); }
  return GrB_SUCCESS;
}
