// grb_spmv.hpp — host interface of the SpMV / SpMSpV kernels (grb_spmv.hip).
#pragma once
#include "grb_internal.hpp"
#include "grb_semiring.hpp"

namespace grb {

constexpr int SPMV_THREADS = 256;
constexpr int SPMV_NNZ = 2048;          // entries per stream block  (LDS: 2048 * sizeof(T) <= 16 KiB)
constexpr int SPMV_UNROLL = SPMV_NNZ / SPMV_THREADS;
constexpr int SPMV_LONG_CHUNK = 8192;   // entries per part of a long row
constexpr int SPMV_MAX_ROWS = 1024;     // rows per stream block (bounds runs of empty rows)

struct SpmvBlock { uint32_t row, aux, nparts, slot; };
// stream block : rows [row, aux), nparts == 0
// long-row part: row, aux = part index, nparts >= 1, slot = first partial slot of this row

enum SpmvMethod { SPMV_AUTO = 0, SPMV_ADAPTIVE = 1, SPMV_ROWGROUP = 2, SPMV_PUSH = 3, SPMV_WAVEPIPE = 4, SPMV_XCD = 5 };

struct SpmvCall {
  DevCSR* M;                // pull: rows of M index the output.  push: rows of M index the input u.
  const void* aval;         // M's values already in the semiring type (nullptr when the multiply ignores them)
  const void* uval;         // u values in the semiring type (dense array of length ncols(M) / nrows(M) for push)
  const uint8_t* upres;     // presence bytes, nullptr when every entry of u is present (pull only)
  const uint8_t* allow;     // per-output "mask allows writing" bytes, nullptr = all allowed
  void* tval; uint8_t* tpres;   // output bitmap vector (semiring type)
  int method = SPMV_AUTO;
  // optional fused epilogue `w = accum(w, t)` with accum = the monoid's own operator, no mask, w full (grb_mxv.cpp):
  //   epi == 1: w's values are `epi_w` (in place: rows with products become w (+) sum, the others stay), presence bytes untouched
  //   epi == 2: w is the pending fill `epi_fill` everywhere (`w(:) = s` not yet written): tval = s (+) sum / s, tpres = 1
  // A kernel that honours it sets *epi_done; otherwise the product lands in tval/tpres as usual and the caller runs the epilogue.
  //   epi == 3: "big holes" (grb_mxv.cpp): `w<accum MIN / MAX> = ...` with the monoid's own operator into w = (epi_w, epi_wpres), which has holes:
  //             a row sum on the far side of the threshold `epi_fill` was made of fill values only and is no entry; a real one is combined
  //             with w(r) or becomes its entry.  In place (the sweeps of the reference's shortest-path loop: no threshold pass, no epilogue).
  int epi = 0; void* epi_w = nullptr; uint8_t* epi_wpres = nullptr; uint8_t epi_fill[16] = {0}; bool* epi_done = nullptr;
  // optional summary of a BOOL result: a device word the kernel sets to `any_true_tag` when it writes an entry whose value is true
  // (`while q.reduce_bool()` of a BFS loop then needs no kernel of its own; a fresh tag per product, so the word is never
  // cleared).  A kernel that honours it sets *any_true_done.
  uint32_t* any_true = nullptr; uint32_t any_true_tag = 0; bool* any_true_done = nullptr;
  // ... and, for a square matrix, the edges that leave the result's TRUE entries in the push orientation (round 5): `fe_rowptr` are the row
  // pointers the direction choice of the NEXT product would count them in (grb_mxv.cpp); every workgroup of the kernel stores its (edge sum,
  // entry count), tagged with `any_true_tag`, into its own pair of the page-locked host words `fe_host` (<= 2048 pairs).  The masked pull
  // counts exactly; the push kernels only when the operand has ONE entry (the columns of one row are distinct: no product lands twice).
  // A kernel that honours it sets *fe_done and *fe_nblocks (the pairs to expect).  The BFS loop's `q.reduce_bool()` spins on the host words —
  // no device-to-host copy, no stream synchronisation — `v[q] = level` hands the edge sum on to v, and the level-2 product needs no counting
  // kernel and no read-back of its own.
  const uint32_t* fe_rowptr = nullptr; unsigned long long* fe_host = nullptr; bool* fe_done = nullptr; uint32_t* fe_nblocks = nullptr;
  // The mask IS the operand, of a one-byte type, under a Boolean semiring (`v.vxm(A, mask=v, out=q, desc=RC)` on the level vector of the
  // reference's BFS loop): the row-lane kernel can read the vector itself — allowed(r) = (pres[r] && (structural || val[r])) != complement,
  // operand value = (val[c] != 0) — where the caller would otherwise make allow bytes and BOOL values in a pass of its own (11 us per
  // level at R-MAT-22).  fm_val = the vector's raw value bytes, upres its presence bytes, fm_flags bit 0 structural, bit 1 complement.
  const uint8_t* fm_val = nullptr; uint8_t fm_flags = 0;
  const uint8_t* fm_code = nullptr;      // (round 6) the vector's code bytes — bit 0 present, bit 1 present and not zero — when it has them: ONE gather per neighbour instead of two
  // push: the operand's entries as a host list of <= 64 ascending indices (GrB_Vector_opaque::small_idx) — no frontier compaction
  const uint32_t* small_idx = nullptr; uint32_t small_n = 0;
  // push, round 5: the mask is that same short list, complemented, every listed value true (the first level of a BFS: `v.vxm(A, mask=v, desc=RC)` with v =
  // {start}): the kernels skip the listed positions themselves and take the operand's values as `true` — no allow bytes, no BOOL copy of the operand
  // (a pass over all n positions and its launch)
  bool excl_small = false;
};
bool spmv_rowlane_applies(const DevCSR& M, const SemiringDesc& d, int method);      // would a masked pull of this matrix run the row-lane kernel?

const std::string& xcd_mapping();   // "roundrobin8" when workgroup b of a full-chip launch runs on XCD b % 8 (probed once)
void spmv_build_plan(DevCSR& M);
void spmv_pull(const SpmvCall& c, const SemiringDesc& d);
bool spmspv_push_supported(const SemiringDesc& d);
void spmspv_push(const SpmvCall& c, const SemiringDesc& d, uint64_t u_nvals);

}  // namespace grb
