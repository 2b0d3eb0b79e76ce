// grb_spgemm_kernels_fwd.hpp — pieces shared between the SpGEMM dispatcher and the per-type kernel instantiations.
#pragma once
#include "grb_api.hpp"
namespace grb {
// four non-blocking side streams, forked from and joined into the library stream with events
struct AuxStreams {
  hipStream_t s[4] = {nullptr, nullptr, nullptr, nullptr}; hipEvent_t e0 = nullptr, e[4] = {nullptr, nullptr, nullptr, nullptr}; bool ok = false;
  void init() { if (ok) return; for (int i = 0; i < 4; i++) { GRB_HIP(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking)); GRB_HIP(hipEventCreateWithFlags(&e[i], hipEventDisableTiming)); }
                GRB_HIP(hipEventCreateWithFlags(&e0, hipEventDisableTiming)); ok = true; }
  void fork(hipStream_t main) { init(); GRB_HIP(hipEventRecord(e0, main)); for (int i = 0; i < 4; i++) GRB_HIP(hipStreamWaitEvent(s[i], e0, 0)); }
  void join(hipStream_t main) { for (int i = 0; i < 4; i++) { GRB_HIP(hipEventRecord(e[i], s[i])); GRB_HIP(hipStreamWaitEvent(main, e[i], 0)); } }
};
AuxStreams& aux_streams();      // defined in grb_spgemm.hip
}  // namespace grb
