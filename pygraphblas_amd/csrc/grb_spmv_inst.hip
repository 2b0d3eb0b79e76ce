// grb_spmv_inst.hip — explicit instantiation of the SpMV kernels for ONE value type.
// The Makefile compiles this file once per built-in type (-DGRB_INST_TYPE=...), in parallel.
#include "grb_spmv_kernels.hpp"
namespace grb {
using std::int8_t; using std::uint8_t; using std::int16_t; using std::uint16_t; using std::int32_t; using std::uint32_t; using std::int64_t; using std::uint64_t;
template void run_pull<GRB_INST_TYPE>(const SpmvCall&, const SemiringDesc&);
template void run_push<GRB_INST_TYPE>(const SpmvCall&, const SemiringDesc&, const uint32_t*, uint64_t, uint32_t*);
}

#ifdef WP_PROFILE
// experiment only: read and reset the phase counters of kernel W (this TU's copy)
extern "C" int GrBX_wp_prof_read(unsigned long long* out8) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out8, HIP_SYMBOL(grb::g_wp_prof), 8192 * 8);
  return 0;
}
#endif

#ifdef XT_PROFILE
// measurement build only: read the per-wave phase counters of the tile pipeline (this translation unit's copy: one per value type)
#define XT_CAT2(a, b) a##b
#define XT_CAT(a, b) XT_CAT2(a, b)
extern "C" int XT_CAT(GrBX_xt_prof_read_, GRB_INST_TYPE)(unsigned long long* out) {
  hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(grb::g_xt_prof), 4096 * 8 * 8);
}
#endif
