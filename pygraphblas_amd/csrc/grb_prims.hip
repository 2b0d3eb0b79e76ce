// grb_prims.hip — device-wide scan and radix sort, delegated to rocPRIM (header-only, ships with
// ROCm).  These are support primitives for format conversion (CSR build, transpose, compaction);
// the hot-path kernels (grb_spmv.hip, grb_spgemm.hip) are hand-written.
#include "grb_api.hpp"
#include "grb_device.hpp"
#include <rocprim/rocprim.hpp>

namespace grb {

void exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint64_t n) {
  if (!n) return;
  size_t tmp = 0;
  GRB_HIP(rocprim::exclusive_scan(nullptr, tmp, in, out, (uint32_t)0, (size_t)n, rocprim::plus<uint32_t>(), stream()));
  DevBuf t(tmp ? tmp : 16);
  GRB_HIP(rocprim::exclusive_scan(t.p, tmp, in, out, (uint32_t)0, (size_t)n, rocprim::plus<uint32_t>(), stream()));
}

void inclusive_scan_max_u32(const uint32_t* in, uint32_t* out, uint64_t n) {
  if (!n) return;
  size_t tmp = 0;
  GRB_HIP(rocprim::inclusive_scan(nullptr, tmp, in, out, (size_t)n, rocprim::maximum<uint32_t>(), stream()));
  DevBuf t(tmp ? tmp : 16);
  GRB_HIP(rocprim::inclusive_scan(t.p, tmp, in, out, (size_t)n, rocprim::maximum<uint32_t>(), stream()));
}

void exclusive_scan_u64(const uint64_t* in, uint64_t* out, uint64_t n) {
  if (!n) return;
  size_t tmp = 0;
  GRB_HIP(rocprim::exclusive_scan(nullptr, tmp, in, out, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), stream()));
  DevBuf t(tmp ? tmp : 16);
  GRB_HIP(rocprim::exclusive_scan(t.p, tmp, in, out, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), stream()));
}

void exclusive_scan_max_u64(const uint64_t* in, uint64_t* out, uint64_t n) {
  if (!n) return;
  size_t tmp = 0;
  GRB_HIP(rocprim::exclusive_scan(nullptr, tmp, in, out, (uint64_t)0, (size_t)n, rocprim::maximum<uint64_t>(), stream()));
  DevBuf t(tmp ? tmp : 16);
  GRB_HIP(rocprim::exclusive_scan(t.p, tmp, in, out, (uint64_t)0, (size_t)n, rocprim::maximum<uint64_t>(), stream()));
}

void sort_pairs_u32(const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, uint64_t n, int end_bit) {
  if (!n) return;
  size_t tmp = 0;
  GRB_HIP(rocprim::radix_sort_pairs(nullptr, tmp, kin, kout, vin, vout, (size_t)n, 0u, (unsigned)end_bit, stream()));
  DevBuf t(tmp ? tmp : 16);
  GRB_HIP(rocprim::radix_sort_pairs(t.p, tmp, kin, kout, vin, vout, (size_t)n, 0u, (unsigned)end_bit, stream()));
}

void sort_pairs_u64(const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout, uint64_t n, int end_bit) {
  if (!n) return;
  size_t tmp = 0;
  GRB_HIP(rocprim::radix_sort_pairs(nullptr, tmp, kin, kout, vin, vout, (size_t)n, 0u, (unsigned)end_bit, stream()));
  DevBuf t(tmp ? tmp : 16);
  GRB_HIP(rocprim::radix_sort_pairs(t.p, tmp, kin, kout, vin, vout, (size_t)n, 0u, (unsigned)end_bit, stream()));
}

void sort_keys_u32(const uint32_t* kin, uint32_t* kout, uint64_t n, int end_bit) {
  if (!n) return;
  size_t tmp = 0;
  GRB_HIP(rocprim::radix_sort_keys(nullptr, tmp, kin, kout, (size_t)n, 0u, (unsigned)end_bit, stream()));
  DevBuf t(tmp ? tmp : 16);
  GRB_HIP(rocprim::radix_sort_keys(t.p, tmp, kin, kout, (size_t)n, 0u, (unsigned)end_bit, stream()));
}

// stable merge of two sorted key sequences with their values (equal keys: the first input's entries first)
void merge_pairs_u64(const uint64_t* k1, const uint64_t* k2, uint64_t* kout, const uint32_t* v1, const uint32_t* v2, uint32_t* vout, uint64_t n1, uint64_t n2) {
  if (!(n1 + n2)) return;
  size_t tmp = 0;
  GRB_HIP(rocprim::merge(nullptr, tmp, k1, k2, kout, v1, v2, vout, (size_t)n1, (size_t)n2, rocprim::less<uint64_t>(), stream()));
  DevBuf t(tmp ? tmp : 16);
  GRB_HIP(rocprim::merge(t.p, tmp, k1, k2, kout, v1, v2, vout, (size_t)n1, (size_t)n2, rocprim::less<uint64_t>(), stream()));
}

// every segment [begins[s], ends[s]) sorted by key on its own (rows of a CSR put in column order; ends = begins + 1 for all rows,
// or a separate array in which the rows that need no sorting are empty)
void segmented_sort_pairs_u32(const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, uint64_t n, uint32_t nseg, const uint32_t* begins, const uint32_t* ends, int end_bit) {
  if (!n || !nseg) return;
  size_t tmp = 0;
  GRB_HIP(rocprim::segmented_radix_sort_pairs(nullptr, tmp, kin, kout, vin, vout, (unsigned)n, (unsigned)nseg, begins, ends, 0u, (unsigned)end_bit, stream()));
  DevBuf t(tmp ? tmp : 16);
  GRB_HIP(rocprim::segmented_radix_sort_pairs(t.p, tmp, kin, kout, vin, vout, (unsigned)n, (unsigned)nseg, begins, ends, 0u, (unsigned)end_bit, stream()));
}

}  // namespace grb
