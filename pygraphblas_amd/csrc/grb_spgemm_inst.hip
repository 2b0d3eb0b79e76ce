// grb_spgemm_inst.hip — explicit instantiation of the SpGEMM kernels for ONE value type (-DGRB_INST_TYPE=...).
#include "grb_spgemm_kernels.hpp"
#include "grb_spgemm_hash.hpp"
namespace grb {
using std::int8_t; using std::uint8_t; using std::int16_t; using std::uint16_t; using std::int32_t; using std::uint32_t; using std::int64_t; using std::uint64_t;
template void run_spgemm_masked<GRB_INST_TYPE>(const SpgemmCall&, const SemiringDesc&, DevCSR&);
template void run_spgemm_esc<GRB_INST_TYPE>(const SpgemmCall&, const SemiringDesc&, DevCSR&);
template void run_spgemm_hash<GRB_INST_TYPE>(const SpgemmCall&, const SemiringDesc&, DevCSR&);
}
