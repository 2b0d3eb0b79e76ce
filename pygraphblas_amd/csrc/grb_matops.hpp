// grb_matops.hpp — host interface of the O(nnz) matrix kernels (grb_matops.hip) and SpGEMM (grb_spgemm.hip).
#pragma once
#include "grb_internal.hpp"
#include "grb_semiring.hpp"

namespace grb {

// out = C<M,replace> (+accum) T  — all three in one value type `code`; M may be nullptr
void csr_writeback(int code, uint32_t nrows, const DevCSR& C, const DevCSR& Tm, const DevCSR* M, int mcode, bool mstruct, bool mcomp,
                   bool replace, int accum, DevCSR& out);
void csr_ewise(int code, const DevCSR& A, const void* aval, const DevCSR& B, const void* bval, int op, bool is_union, DevCSR& out);
void csr_compact(const DevCSR& A, const void* aval, size_t ts, const uint8_t* keep, DevCSR& out);
void select_positional_flags(const DevCSR& A, int sel, int64_t k, uint8_t* keep);
void mask_flags(const DevCSR& Tm, const DevCSR& M, int mcode, bool mstruct, bool mcomp, uint8_t* keep);
void csr_reduce_rows(int code, const DevCSR& A, const void* aval, int op, void* tval, uint8_t* tpres);
// the same per COLUMN of A, without the transpose: every entry combines into its column's accumulator (atomics); false for 1- and 2-byte types
bool csr_reduce_cols(int code, const DevCSR& A, const void* aval, int op, const void* identity, void* tval, uint8_t* tpres);
bool csr_reduce_cols_few_rows(int code, const DevCSR& A, const void* aval, int op, void* tval, uint8_t* tpres);   // <= 64 rows, FP types: row after row, fixed order, no atomics
uint64_t csr_find_entry(const DevCSR& A, uint32_t i, uint32_t j);        // position of (i, j) in col / val, ~0 when not stored (one host round trip)
void csr_row_indices(const DevCSR& A, uint32_t* rowidx);
void csr_dense_fill(uint32_t nrows, uint32_t ncols, const void* scalar, size_t ts, DevCSR& out);     // every position holds `scalar`: rowptr[i] = i ncols, col[e] = e mod ncols
// positional unary operators (GxB_POSITIONI / I1 / J / J1): the row (which 0 / 1: + 1) or column (2 / 3) index of every entry of A as INT32 / INT64 values; the same for the n positions of a vector
void csr_position_values(int zcode, const DevCSR& A, int which, void* out);
void vec_position_values(int zcode, uint64_t n, int which, void* out);

// ---- SpGEMM ----------------------------------------------------------------------------------------------
struct SpgemmCall {
  const DevCSR* A; const void* aval;     // values already in the semiring type (nullptr: multiply ignores them)
  const DevCSR* B; const void* bval;
  const DevCSR* M; int mcode; bool mstruct;   // non-complemented mask for the masked kernel (nullptr otherwise)
  bool ordered = false;                       // spgemm_hash: floating-point sums in a fixed order (GRB_MI355X_DETERMINISTIC=1 / GxB_AxB_GUSTAVSON)
};
// T<M> = A (+).(x) B restricted to the entries the mask allows; T's pattern is a subset of M's
void spgemm_masked(const SpgemmCall& c, const SemiringDesc& d, DevCSR& out);
// T = A (+).(x) B by expand / sort / compress (deterministic summation order)
void spgemm_esc(const SpgemmCall& c, const SemiringDesc& d, DevCSR& out);
// T = A (+).(x) B by a two-pass (symbolic + numeric) Gustavson with LDS hash accumulators (grb_spgemm_hash.hpp)
void spgemm_hash(const SpgemmCall& c, const SemiringDesc& d, DevCSR& out);

}  // namespace grb
