// grb_transpose.hip — CSR -> CSR of the transpose, entirely in HBM.
// Used for GrB_vxm / desc.T0 (the pull kernels want rows of the effective left operand;
// gap/prmark.py:50 stores the matrix by column for exactly this reason),
// for GrB_transpose, and cached on the matrix after the first use.
//   1. expand rowptr to a row index per entry and take col as the sort key
//   2. stable radix sort of (col -> entry position): within one column the original row-major
//      order, i.e. ascending rows, is preserved, so the output rows are sorted
//   3. gather values / row ids through the permutation; histogram + scan gives the new rowptr
#include "grb_api.hpp"
#include "grb_device.hpp"

namespace grb {

// row of every entry (round 6): the non-empty rows mark their first entry, an inclusive max-scan fills the rest.  (A thread per row writing its entries one
// after the other took 7.2 ms on the symmetric R-MAT-22 — its hub rows hold 10^5 entries — of the 28 ms a BFS's first run spent building the transpose.)
__global__ void k_mark_row_starts(const uint32_t* __restrict__ rowptr, uint32_t nrows, uint32_t* __restrict__ rowidx) {
  for (uint64_t r = blockIdx.x * 256ull + threadIdx.x; r < nrows; r += gridDim.x * 256ull) { const uint32_t b = rowptr[r]; if (rowptr[r + 1] > b) rowidx[b] = (uint32_t)r; }
}
// row pointers of the transpose from the SORTED column keys (round 6): position i starts the run of key k[i]; every key between the one before and k[i] is an
// empty row that starts there too.  (One atomicAdd per entry — 1.3e8 device-scope atomics — took 12.6 ms of those 28.)
__global__ void k_rowptr_from_sorted(const uint32_t* __restrict__ k, uint64_t n, uint32_t nkeys, uint32_t* __restrict__ rowptr) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i <= n; i += gridDim.x * 256ull) {
    if (i == n) { for (uint32_t c = (n ? k[n - 1] + 1 : 0u); c <= nkeys; c++) rowptr[c] = (uint32_t)n; continue; }
    const uint32_t cur = k[i];
    if (i == 0) { for (uint32_t c = 0; c <= cur; c++) rowptr[c] = 0; }
    else { const uint32_t prev = k[i - 1]; if (prev != cur) for (uint32_t c = prev + 1; c <= cur; c++) rowptr[c] = (uint32_t)i; }
  }
}
__global__ void k_iota(uint32_t* p, uint64_t n) {
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = (uint32_t)i;
}
template <int TS> __global__ void k_gather_perm(const uint32_t* __restrict__ perm, uint64_t n, const uint32_t* __restrict__ rowidx,
                                                const uint8_t* __restrict__ val, uint32_t* __restrict__ ocol, uint8_t* __restrict__ oval) {
  typedef typename std::conditional<TS == 8, uint64_t, typename std::conditional<TS == 4, uint32_t,
          typename std::conditional<TS == 2, uint16_t, uint8_t>::type>::type>::type W;
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
    const uint32_t p = perm[i];
    ocol[i] = rowidx[p];
    ((W*)oval)[i] = ((const W*)val)[p];
  }
}

static inline int grid_of(uint64_t n) { uint64_t b = (n + 255) / 256; if (b < 1) b = 1; if (b > 4096) b = 4096; return (int)b; }

void csr_transpose(const DevCSR& A, size_t ts, DevCSR& At) {
  At.clear();
  At.nrows = A.ncols; At.ncols = A.nrows; At.nnz = A.nnz;
  const uint64_t nnz = A.nnz;
  At.rowptr.alloc(((size_t)At.nrows + 1) * 4); At.col.alloc(nnz * 4); At.val.alloc(nnz * ts);
  GRB_HIP(hipMemsetAsync(At.rowptr.p, 0, ((size_t)At.nrows + 1) * 4, stream()));
  if (nnz) {
    DevBuf rowidx(nnz * 4 + 4), perm_in(nnz * 4), perm(nnz * 4), keys_out(nnz * 4);
    GRB_HIP(hipMemsetAsync(rowidx.p, 0, nnz * 4 + 4, stream()));
    hipLaunchKernelGGL(k_mark_row_starts, dim3(grid_of(A.nrows)), dim3(256), 0, stream(), A.rowptr.as<uint32_t>(), A.nrows, rowidx.as<uint32_t>());
    inclusive_scan_max_u32(rowidx.as<uint32_t>(), rowidx.as<uint32_t>(), nnz);
    hipLaunchKernelGGL(k_iota, dim3(grid_of(nnz)), dim3(256), 0, stream(), perm_in.as<uint32_t>(), nnz);
    int bits = 1; while (bits < 32 && (1ull << bits) < (uint64_t)A.ncols) bits++;
    sort_pairs_u32(A.col.as<uint32_t>(), keys_out.as<uint32_t>(), perm_in.as<uint32_t>(), perm.as<uint32_t>(), nnz, bits);
    hipLaunchKernelGGL(k_rowptr_from_sorted, dim3(grid_of(nnz + 1)), dim3(256), 0, stream(), keys_out.as<uint32_t>(), nnz, At.nrows, At.rowptr.as<uint32_t>());
    const int g = grid_of(nnz);
    switch (ts) {
      case 1: hipLaunchKernelGGL((k_gather_perm<1>), dim3(g), dim3(256), 0, stream(), perm.as<uint32_t>(), nnz, rowidx.as<uint32_t>(), A.val.as<uint8_t>(), At.col.as<uint32_t>(), At.val.as<uint8_t>()); break;
      case 2: hipLaunchKernelGGL((k_gather_perm<2>), dim3(g), dim3(256), 0, stream(), perm.as<uint32_t>(), nnz, rowidx.as<uint32_t>(), A.val.as<uint8_t>(), At.col.as<uint32_t>(), At.val.as<uint8_t>()); break;
      case 4: hipLaunchKernelGGL((k_gather_perm<4>), dim3(g), dim3(256), 0, stream(), perm.as<uint32_t>(), nnz, rowidx.as<uint32_t>(), A.val.as<uint8_t>(), At.col.as<uint32_t>(), At.val.as<uint8_t>()); break;
      default: hipLaunchKernelGGL((k_gather_perm<8>), dim3(g), dim3(256), 0, stream(), perm.as<uint32_t>(), nnz, rowidx.as<uint32_t>(), A.val.as<uint8_t>(), At.col.as<uint32_t>(), At.val.as<uint8_t>()); break;
    }
    GRB_HIP(hipStreamSynchronize(stream()));   // temporaries are released on scope exit; the pool is stream-ordered
  }
  At.valid = true;
}

}  // namespace grb
