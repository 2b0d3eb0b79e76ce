// grb_internal.hpp — object model of the MI355X GraphBLAS backend (not part of the C ABI).
//
// Layout in HBM:
//   Matrix : CSR  rowptr u32[nrows+1] | col u32[nnz] (sorted within a row) | val T[nnz]
//            + a lazily built, cached CSR of the transpose ("CSC view") and SpMV row-block plan.
//   Vector : bitmap  val T[n] | present u8[n]   (+ nvals); sparse index lists are transient.
// Host mirrors (sorted tuples) exist only for element-wise host access (setElement /
// extractTuples) and are synchronised lazily in either direction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include <mutex>
#include <stdlib.h>
#include <memory>
#include "grb_ops.hpp"

typedef uint64_t GrB_Index;

// GrB_Info: positive v1.3 / SuiteSparse-5 numbering (reference: pygraphblas/base.py:189-203)
enum GrB_Info_e : int {
  GrB_SUCCESS = 0, GrB_NO_VALUE = 1, GrB_UNINITIALIZED_OBJECT = 2, GrB_INVALID_OBJECT = 3,
  GrB_NULL_POINTER = 4, GrB_INVALID_VALUE = 5, GrB_INVALID_INDEX = 6, GrB_DOMAIN_MISMATCH = 7,
  GrB_DIMENSION_MISMATCH = 8, GrB_OUTPUT_NOT_EMPTY = 9, GrB_OUT_OF_MEMORY = 10,
  GrB_INSUFFICIENT_SPACE = 11, GrB_INDEX_OUT_OF_BOUNDS = 12, GrB_PANIC = 13
};
typedef int GrB_Info;

// descriptor fields / values (SuiteSparse v5.1 numbering)
enum { GrB_OUTP = 0, GrB_MASK = 1, GrB_INP0 = 2, GrB_INP1 = 3,
       GxB_DESCRIPTOR_NTHREADS = 5, GxB_DESCRIPTOR_CHUNK = 7, GxB_DESCRIPTOR_GPU_CONTROL = 21,
       GxB_DESCRIPTOR_GPU_CHUNK = 22, GxB_AxB_METHOD = 1000, GxB_SORT = 35 };
enum { GxB_DEFAULT = 0, GrB_REPLACE = 1, GrB_COMP = 2, GrB_TRAN = 3, GrB_STRUCTURE = 4,
       GxB_AxB_GUSTAVSON = 1001, GxB_AxB_DOT = 1003, GxB_AxB_HASH = 1004, GxB_AxB_SAXPY = 1005 };

#define GRB_MAGIC 0x72657473786f62ULL
#define GRB_FREED 0x6c6c756e786f62ULL
#define GRB_DIM_DEVICE_MAX 0xFFFFFFF0ULL   // device containers use 32-bit indices
#define GXB_INDEX_MAX ((GrB_Index)1 << 60)

struct GrB_Type_opaque { uint64_t magic; int code; size_t size; char name[32]; };
struct GrB_UnaryOp_opaque { uint64_t magic; int opcode; GrB_Type_opaque* xtype; GrB_Type_opaque* ztype; char name[40]; void* fn; };
struct GrB_BinaryOp_opaque { uint64_t magic; int opcode; GrB_Type_opaque* xtype; GrB_Type_opaque* ytype; GrB_Type_opaque* ztype; char name[40]; void* fn; };
struct GrB_Monoid_opaque { uint64_t magic; GrB_BinaryOp_opaque* op; uint8_t identity[16]; bool has_terminal; uint8_t terminal[16]; char name[48]; bool builtin; };
struct GrB_Semiring_opaque { uint64_t magic; GrB_Monoid_opaque* add; GrB_BinaryOp_opaque* mul; char name[56]; bool builtin; };
struct GrB_Descriptor_opaque { uint64_t magic; int outp, mask, inp0, inp1, axb, nthreads, sort; double chunk; bool builtin; char name[16]; };
struct GxB_SelectOp_opaque { uint64_t magic; int opcode; char name[24]; void* fn; GrB_Type_opaque* xtype; GrB_Type_opaque* ttype; };

typedef GrB_Type_opaque* GrB_Type;
typedef GrB_UnaryOp_opaque* GrB_UnaryOp;
typedef GrB_BinaryOp_opaque* GrB_BinaryOp;
typedef GrB_Monoid_opaque* GrB_Monoid;
typedef GrB_Semiring_opaque* GrB_Semiring;
typedef GrB_Descriptor_opaque* GrB_Descriptor;
typedef GxB_SelectOp_opaque* GxB_SelectOp;

enum SelectCode { SEL_TRIL = 0, SEL_TRIU, SEL_DIAG, SEL_OFFDIAG, SEL_NONZERO, SEL_EQ_ZERO, SEL_GT_ZERO,
                  SEL_GE_ZERO, SEL_LT_ZERO, SEL_LE_ZERO, SEL_NE_THUNK, SEL_EQ_THUNK, SEL_GT_THUNK,
                  SEL_GE_THUNK, SEL_LT_THUNK, SEL_LE_THUNK, SEL_USER };

namespace grb {

// ---- device memory ----------------------------------------------------------------------
bool device_ok();                 // a HIP device was found at init
const char* device_error();       // why not
hipStream_t stream();             // the stream every kernel of the library is launched on
void* dev_alloc(size_t bytes);    // pooled hipMalloc; throws GrbError(OUT_OF_MEMORY/PANIC)
void dev_free(void* p);
void dev_pool_release();          // return cached blocks to the driver
size_t dev_bytes_in_use();
int device_cus();                 // compute units of the device (256 on MI355X)

struct GrbError { int info; std::string msg; };
[[noreturn]] void fail(int info, const std::string& msg);
#define GRB_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
  ::grb::fail(GrB_PANIC, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

// owning device buffer
uint64_t dev_alloc_serial();      // a number no other allocation of this process gets (the pool hands the same addresses out again)
struct DevBuf {
  void* p = nullptr; size_t bytes = 0;
  uint64_t serial = 0;              // identifies this allocation where a raw address could be a recycled one (Vector::fe_lb_key)
  bool borrowed = false;            // a view of memory another DevBuf owns (a row of a batch matrix's bitmap handed to a vector kernel, grb_mxm_rows.cpp): never freed here
  DevBuf() {}
  explicit DevBuf(size_t n) { alloc(n); }
  DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes), serial(o.serial), borrowed(o.borrowed) { o.p = nullptr; o.bytes = 0; o.serial = 0; o.borrowed = false; }
  DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { reset(); p = o.p; bytes = o.bytes; serial = o.serial; borrowed = o.borrowed; o.p = nullptr; o.bytes = 0; o.serial = 0; o.borrowed = false; } return *this; }
  ~DevBuf() { reset(); }
  void alloc(size_t n) { reset(); if (n) { p = dev_alloc(n); bytes = n; serial = dev_alloc_serial(); } }
  void borrow(void* q, size_t n) { reset(); p = q; bytes = n; serial = dev_alloc_serial(); borrowed = true; }
  void reset() { if (p && !borrowed) dev_free(p); p = nullptr; bytes = 0; serial = 0; borrowed = false; }
  template <class T> T* as() const { return (T*)p; }
};

// ---- device CSR ---------------------------------------------------------------------------
struct DevCSR {
  uint32_t nrows = 0, ncols = 0; uint64_t nnz = 0;
  DevBuf rowptr, col, val;      // u32[nrows+1], u32[nnz], T[nnz]
  // SpMV row-block plan (grb_spmv.hip): blocks of consecutive rows holding <= SPMV_BLOCK_NNZ
  // entries, long rows split into parts.  Built on first mxv, cached with the matrix.
  DevBuf plan_blocks; uint32_t plan_nblocks = 0; DevBuf plan_aux; uint32_t plan_nlong = 0;
  bool has_plan = false;
  int range_state = 0;          // largest |value| of the stored values: 0 not measured, 1 = range_abs holds it, 2 = not usable (NaN / infinity / a type without a range)
  double range_abs = 0;         // (as a double: an upper bound is all the caller needs — grb_mxv.cpp "big holes")
  int locality_pct = -1;        // share of entries whose column is < 16 behind its predecessor in the row (measured once, -1 = not yet): gathers with locality need no panels
  // kernel-W plan (grb_spmv_wavepipe.hpp): per-task first row, hot-column list, remapped column array, per-wave carries
  DevBuf wp_rs, wp_hot, wp_pcol, wp_carry; uint32_t wp_nhot = 0, wp_ntasks = 0, wp_nwarm = 0; int wp_tsize = 0;
  std::shared_ptr<void> xcd;    // kernel-X plan (grb_spmv_xcd.hpp: XcdPlan), panel-major copy of the matrix
  // row heads (grb_spmv_kernels.hpp, round 5): for the masked pull of a BFS level — per row its first four entries as one 16-byte word (per entry:
  // bits 29..0 the column, bit 31 its BOOL value, in the fourth bit 30 = the row has more entries; absent = 0xFFFFFFFF) and one bit per row "has an
  // entry": a level then costs what its unvisited NON-EMPTY rows cost, and a row decided by its first entries never touches the column array.
  // Built at the first masked pull of a matrix with < 2^30 - 1 columns; heads_vals: the value bits come from one-byte BOOL values (else they are 1).
  DevBuf heads, nonempty; bool heads_valid = false, heads_vals = false;
  uint32_t pipe_uses = 0;       // full-operand pull products this matrix has served: the first runs kernel W (cheap plan), kernel X's plan is built for the second
  bool valid = false;
  void clear() { rowptr.reset(); col.reset(); val.reset(); plan_blocks.reset(); plan_aux.reset();
                 wp_rs.reset(); wp_hot.reset(); wp_pcol.reset(); wp_carry.reset(); wp_nhot = wp_ntasks = 0; wp_tsize = 0; xcd.reset(); pipe_uses = 0; heads.reset(); nonempty.reset(); heads_valid = false;
                 nnz = 0; has_plan = false; locality_pct = -1; range_state = 0; valid = false; plan_nblocks = plan_nlong = 0; }
};

}  // namespace grb

// A matrix of a few very long rows (the ns x n batches of the reference's betweenness centrality, gap/bcmark.py:16-67) as a BITMAP: val T[nrows x ncols]
// | pres u8[nrows x ncols], row-major — i.e. ns bitmap vectors back to back, or one bitmap vector of nrows x ncols positions.  Round 6: such a matrix may
// live in this form alone (dev_valid == host_valid == false, bm.valid == true): the batch operations (grb_mxm_rows.cpp) read and write it directly, and the
// CSR is made from it only when something else asks (mat_to_device).
namespace grb { struct DevBitmap { DevBuf val, pres; bool valid = false; uint64_t nvals = 0; bool nvals_known = false;
                                  void clear() { val.reset(); pres.reset(); valid = false; nvals = 0; nvals_known = false; } }; }
struct GrB_Matrix_opaque {
  uint64_t magic = GRB_MAGIC;
  GrB_Type type = nullptr;
  GrB_Index nrows = 0, ncols = 0;
  // host mirror: tuples sorted by (i, j), values packed at type->size each
  bool host_valid = true;
  std::vector<GrB_Index> hi, hj; std::vector<uint8_t> hx;
  // pending host edits (setElement / removeElement), applied in order at assemble time
  struct Pending { GrB_Index i, j; bool del; uint8_t x[16]; };
  std::vector<Pending> pending;
  // a full, one-valued ("iso") container of a dimension beyond both layouts — what `Matrix.dense(T)` / `Matrix.iso(x)` make with
  // the default GxB_INDEX_MAX dimensions (pygraphblas/matrix.py:220-230, 234-266): it can be read element-wise, nothing more
  bool iso_full = false; uint8_t iso_val[16] = {0};
  // device
  bool dev_valid = false;
  uint32_t dev_elem_ops = 0; // element reads / writes served on the device in a row (grb_container.cpp: after a few dozen the host mirror takes over)
  grb::DevCSR csr;          // by-row
  grb::DevCSR csc;          // CSR of the transpose (cached; invalidated with csr)
  grb::DevBitmap bm;        // the bitmap form of a batch matrix (cached beside the CSR, or the only valid form)
  int format = 0;           // GxB_BY_ROW(0) / GxB_BY_COL(1): stored option only
  int sparsity_control = 15;
  double hyper_switch = 0.0625;
  std::string err;
};

struct GrB_Vector_opaque {
  uint64_t magic = GRB_MAGIC;
  GrB_Type type = nullptr;
  GrB_Index n = 0;
  bool host_valid = true;
  std::vector<GrB_Index> hi; std::vector<uint8_t> hx;
  struct Pending { GrB_Index i; bool del; uint8_t x[16]; };
  std::vector<Pending> pending;
  bool iso_full = false; uint8_t iso_val[16] = {0};   // as for matrices: `Vector.iso(x)` of the default size
  bool dev_valid = false;
  grb::DevBuf dval, dpres;   // T[n], u8[n]
  uint64_t dnvals = 0; bool dnvals_known = true;   // entry count of the device bitmap, recounted lazily
  // a lower bound of the edges that leave the vector's entries in one matrix (keyed by its row-pointer buffer), valid while entries are
  // only added (scalar assign under a mask without replace — the `v[q] = level` of a BFS loop): a masked product whose operand was
  // already too heavy for a push step needs no recount to stay a pull step.  A stale value can only cost speed, never correctness.
  uint64_t fe_lb = 0; uint64_t fe_lb_key = 0; bool fe_lb_true = false;   // fe_lb_true: the bound counts the edges of the TRUE entries only (the product's summary), not of every present one      // (key: the serial of the matrix's row-pointer allocation, 0 = none)
  // ---- non-blocking state (grb_lazy.cpp; the library is initialised GrB_NONBLOCKING by the reference, pygraphblas/__init__.py:251-256) ----
  // lazy == 1: `w(:) = lazy_fill` over every index was requested and nothing has been written yet (no buffers): a product that
  //            accumulates into w with its monoid's operator folds the fill into its own store; anything else materialises it.
  // lazy == 2: w is the output of queued element-wise operations (its stored value, if any, is the old one).
  // q_reads:   queued operations that read this vector's stored value — it must not change before they ran.
  // Every access goes through vec_gate() (vec_to_device / vec_to_host / vec_nvals / the entry points of grb_container.cpp).
  int lazy = 0; uint8_t lazy_fill[16] = {0}; int q_reads = 0;
  bool holes_zero = false;     // the device values of absent positions are all-zero bits (written so by the element-wise chain kernel)
  bool holes_big = false; uint8_t holes_big_val[16] = {0};      // ... or all hold this value of the vector's type (the BIG fill of a MIN_PLUS / MAX_PLUS sweep, grb_mxv.cpp; round 4).  Reset wherever holes_zero is.
  // BOOL vectors: "is any stored value true", recorded by the product kernel that wrote the device buffers `lor_key`
  // (0 unknown, 1 in the device word of grb_container.cpp's any_true_*, 2 false, 3 true) — `while q.reduce_bool()` of a BFS loop
  uint8_t lor_state = 0; uint32_t lor_tag = 0; const void* lor_key = nullptr;
  // an upper bound of |value| over the stored entries, left behind by the "big holes" product that wrote them (grb_mxv.cpp: the next
  // sweep of a shortest-path loop needs no range kernel and no read-back); < 0 = unknown.  Reset wherever lor_state is.
  double abs_bound = -1;
  // the device image's entries as a list, when it has at most 64 of them and the list is known for free (uploaded from a small host mirror;
  // `v(q) = s` over the entries of such a q): the first level of a BFS then pushes from the list — no frontier compaction (a flag pass,
  // a scan and a scatter over all n positions), no counting kernel, no read-back.  small_truthy: every listed value is non-zero.
  // Reset wherever lor_state is.
  std::vector<uint32_t> small_idx; bool small_valid = false, small_truthy = false;
  // one CODE byte per position of a one-byte-typed vector (round 6): bit 0 = present, bit 1 = present and its value is not zero — what the masked pull of a BFS
  // level gathers per neighbour as ONE byte instead of a presence byte and a value byte (grb_spmv_kernels.hpp: k_spmv_rowlane_k<..., CODE>).  Written for free by
  // the masked scalar assign that precedes the product (`v[q] = level`), else by one pass before the pull.  Reset wherever lor_state is.
  grb::DevBuf dcode; bool code_valid = false;
  uint32_t dev_elem_ops = 0;   // element reads served on the device in a row (grb_container.cpp: after a few dozen the host mirror takes over)
  int sparsity_control = 15;
  std::string err;
};

struct GxB_Scalar_opaque {
  uint64_t magic = GRB_MAGIC;
  GrB_Type type = nullptr; bool has = false; uint8_t x[16] = {0};
  std::string err;
};

typedef GrB_Matrix_opaque* GrB_Matrix;
typedef GrB_Vector_opaque* GrB_Vector;
typedef GxB_Scalar_opaque* GxB_Scalar;

namespace grb {

// ---- containers (grb_container.cpp) ----------------------------------------------------------
void mat_host_assemble(GrB_Matrix A);     // apply pending edits to the host mirror
void mat_to_host(GrB_Matrix A);           // ensure host mirror valid (downloads if needed)
void mat_to_device(GrB_Matrix A);         // ensure device CSR valid (assembles + uploads)
void mat_invalidate_host(GrB_Matrix A);   // device was written
void mat_invalidate_device(GrB_Matrix A); // host was written
uint64_t mat_nvals(GrB_Matrix A);
// batch matrices (a few very long rows) in bitmap form — grb_mxm_rows.cpp
bool mat_batch_shape(uint64_t nrows, uint64_t ncols, int type_code);   // <= 64 rows of >= 65536 (a multiple of 64) columns, a non-complex type, < 2^32 positions
grb::DevBitmap& mat_bitmap(GrB_Matrix A);          // ensure the bitmap form is valid (made from the CSR when it is not)
void mat_bitmap_to_csr(GrB_Matrix A);              // ... and the CSR from the bitmap (what mat_to_device does for a matrix that lives as a bitmap)
uint64_t mat_bitmap_nvals(GrB_Matrix A);           // entries of the bitmap (counted once, then remembered)
inline bool mat_bitmap_only(GrB_Matrix A) { return A->bm.valid && !A->dev_valid && !A->host_valid; }
// hypersparse containers (a dimension beyond the device layouts): the products and eWise operations on the index sets that occur (grb_hyper.cpp)
bool is_hyper(const GrB_Matrix_opaque* A); bool is_hyper(const GrB_Vector_opaque* v);
void hyper_mxv_like(GrB_Vector w, GrB_Vector mask, GrB_BinaryOp accum, GrB_Semiring semiring, GrB_Matrix A, GrB_Vector u, GrB_Descriptor desc, bool is_vxm);
void hyper_mxm(GrB_Matrix C, GrB_Matrix M, GrB_BinaryOp accum, GrB_Semiring semiring, GrB_Matrix A, GrB_Matrix B, GrB_Descriptor desc);
void hyper_vec_ewise(GrB_Vector w, GrB_Vector mask, GrB_BinaryOp accum, GrB_BinaryOp op, GrB_Vector u, GrB_Vector v, GrB_Descriptor desc, bool is_union);
void hyper_mat_ewise(GrB_Matrix C, GrB_Matrix M, GrB_BinaryOp accum, GrB_BinaryOp op, GrB_Matrix A, GrB_Matrix B, GrB_Descriptor desc, bool is_union);
// scalar into a region on the host mirror (grb_host_ops.cpp): for complex containers and for dimensions beyond the device layout
void host_assign_scalar(GrB_Matrix C, GrB_Matrix Mask, GrB_BinaryOp accum, const void* x, int xcode, const GrB_Index* I, GrB_Index ni, const GrB_Index* J, GrB_Index nj, GrB_Descriptor desc);
void host_assign_scalar(GrB_Vector w, GrB_Vector mask, GrB_BinaryOp accum, const void* x, int xcode, const GrB_Index* I, GrB_Index ni, GrB_Descriptor desc);
void vec_resolve(GrB_Vector v);           // complete deferred work that involves v (grb_lazy.cpp)
inline void vec_gate(GrB_Vector v) { if (v->lazy | v->q_reads) vec_resolve(v); }
void vec_overwritten(GrB_Vector v);       // v's value is about to be replaced as a whole: deferred work that only produced it is dropped
uint32_t* any_true_acquire(uint32_t* tag);        // a device word a BOOL product kernel sets to the (fresh, non-zero) tag when it writes a true value (see lor_state)
void any_true_written(GrB_Vector w_or_null, const void* key, uint32_t tag, uint64_t fe_key = 0, uint32_t fe_nblocks = 0);    // a kernel honoured it; w's device buffers `key` are the result it describes (nullptr: nobody's); fe_key != 0: the kernel also filled the edge summary (counted in the row pointers with that serial)
bool dist_exchange_pending();                   // grb_dist.cpp: GrBX_Vector_allgatherv_start without its GrBX_dist_wait yet
// GRB_MI355X_DETERMINISTIC=1: floating-point PLUS results are the same bits in every run (mxm: grb_matrix_ops.cpp; vxm / mxv: no push step, whose atomics land in any order)
inline bool deterministic_env() { const char* e = getenv("GRB_MI355X_DETERMINISTIC"); return e && atoi(e) != 0; }
unsigned long long* fe_summary_host();      // SpmvCall::fe_host: the page-locked pairs of the result summary (device address), for the product that just called any_true_acquire
bool any_true_lookup(GrB_Vector u, bool* value);
bool nonblocking();                       // GrB_init(GrB_NONBLOCKING) and not GRB_MI355X_BLOCKING=1
void vec_host_assemble(GrB_Vector v);
void vec_to_host(GrB_Vector v);
void vec_to_device(GrB_Vector v);
void vec_invalidate_host(GrB_Vector v);
void vec_invalidate_device(GrB_Vector v);
uint64_t vec_nvals(GrB_Vector v);
uint64_t vec_dev_nvals(GrB_Vector v);   // entry count of the (valid) device bitmap
const DevCSR& mat_csc(GrB_Matrix A);      // cached CSR of A^T (device)

bool check_obj(const void* p);            // magic check
GrB_Type type_by_code(int code);

// scalar conversion between any two built-in real types through host code
void cast_scalar(int dst_code, void* dst, int src_code, const void* src);

// ---- kernels launched from the op drivers (see the respective .hip files) -----------------------
// grb_transpose.hip
void csr_transpose(const DevCSR& A, size_t tsize, DevCSR& At);
// grb_vecops.hip
void vec_cast_values(int dst_code, void* dst, int src_code, const void* src, uint64_t n);
void vec_cast_fill_values(int dst_code, void* dst, int src_code, const void* src, const uint8_t* pres, uint64_t n, const void* fill);
void build_allow(uint64_t n, int mcode, const void* mval, const uint8_t* mpres, bool structural,
                 bool complement, uint8_t* allow);
uint64_t count_present(const uint8_t* pres, uint64_t n);
void init_entries_small(uint32_t k, const uint64_t* idx_host, const uint8_t* vals_host, size_t ts, void* val, uint8_t* pres, uint64_t n);      // zero both arrays of an n-vector and write k <= 16 entries: one launch   // k <= 16, ts <= 8: entries passed as kernel arguments (no staging, no synchronisation)
void write_small_list(uint32_t k, const uint32_t* idx_host, uint32_t* out_dev);   // k <= 64 indices passed as kernel arguments
void scatter_entries(uint32_t k, const uint32_t* idx_dev, const void* vals_dev, size_t ts, void* val, uint8_t* pres);   // val[idx[e]] = vals[e], pres[idx[e]] = 1
uint64_t frontier_edges(const uint8_t* pres, const uint32_t* rowptr, uint64_t n);
uint64_t frontier_edges_and_count(const uint8_t* pres, const uint32_t* rowptr, uint64_t n, uint64_t* count);

}  // namespace grb
