// grb_spgemm.hip — type dispatch for the SpGEMM kernels (instantiated per value type in grb_spgemm_inst.hip).
#include "grb_api.hpp"
#include "grb_matops.hpp"
#include "grb_spgemm_kernels_fwd.hpp"

namespace grb {
AuxStreams& aux_streams() { static AuxStreams a; return a; }
template <class T> void run_spgemm_masked(const SpgemmCall& c, const SemiringDesc& d, DevCSR& out);
template <class T> void run_spgemm_esc(const SpgemmCall& c, const SemiringDesc& d, DevCSR& out);
template <class T> void run_spgemm_hash(const SpgemmCall& c, const SemiringDesc& d, DevCSR& out);

void spgemm_masked(const SpgemmCall& c, const SemiringDesc& d, DevCSR& out) {
  dispatch_type(d.zcode, [&]<class T>() { run_spgemm_masked<T>(c, d, out); });
}
void spgemm_esc(const SpgemmCall& c, const SemiringDesc& d, DevCSR& out) {
  dispatch_type(d.zcode, [&]<class T>() { run_spgemm_esc<T>(c, d, out); });
}
void spgemm_hash(const SpgemmCall& c, const SemiringDesc& d, DevCSR& out) {
  dispatch_type(d.zcode, [&]<class T>() { run_spgemm_hash<T>(c, d, out); });
}
}  // namespace grb
