// grb_spgemm_inst.hip — explicit instantiation of the SpGEMM kernels for ONE value type (-DGRB_INST_TYPE=...).
#include "grb_spgemm_kernels.hpp"
#include "grb_spgemm_hash.hpp"
namespace grb {
using std::int8_t; using std::uint8_t; using std::int16_t; using std::uint16_t; using std::int32_t; using std::uint32_t; using std::int64_t; using std::uint64_t;
template void run_spgemm_masked<GRB_INST_TYPE>(const SpgemmCall&, const SemiringDesc&, DevCSR&);
template void run_spgemm_esc<GRB_INST_TYPE>(const SpgemmCall&, const SemiringDesc&, DevCSR&);
template void run_spgemm_hash<GRB_INST_TYPE>(const SpgemmCall&, const SemiringDesc&, DevCSR&);
}

#ifdef SPA_PROFILE
// measurement build only: the per-workgroup phase counters of the last k_spgemm_spa_numeric launch (this translation unit's copy: one per value type)
#define SPA_CAT2(a, b) a##b
#define SPA_CAT(a, b) SPA_CAT2(a, b)
extern "C" int SPA_CAT(GrBX_spa_prof_read_, GRB_INST_TYPE)(unsigned long long* out) {
  hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(grb::g_spa_prof), 1024 * 16 * 8);
}
#endif
