#!/usr/bin/env python3
"""Generate the built-in object registry of libgrb_mi355x and the public C header.

Outputs (both committed):
  include/grb_mi355x.h                      public C ABI (also the text the CFFI shim cdef()s)
  pygraphblas_amd/csrc/registry_gen.inc     definitions of every built-in type / operator /
                                            monoid / semiring / descriptor handle

The *names* are the GraphBLAS C API 1.3 + SuiteSparse v5.1 names that pygraphblas discovers by
regex over dir(lib) (reference: pygraphblas/semiring.py:87-121, monoid.py:81-93,
binaryop.py:104-112, unaryop.py:55-63, descriptor.py:148-182, types.py:182-342).  Nothing here is
derived from SuiteSparse sources; the operator set is enumerated from those regexes.
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REAL = ["BOOL", "INT8", "UINT8", "INT16", "UINT16", "INT32", "UINT32", "INT64", "UINT64", "FP32", "FP64"]
NUM = REAL[1:]
INTS = ["INT8", "UINT8", "INT16", "UINT16", "INT32", "UINT32", "INT64", "UINT64"]
UINTS = ["UINT8", "UINT16", "UINT32", "UINT64"]
FLOATS = ["FP32", "FP64"]
CTYPE = {"BOOL": "bool", "INT8": "int8_t", "UINT8": "uint8_t", "INT16": "int16_t", "UINT16": "uint16_t",
         "INT32": "int32_t", "UINT32": "uint32_t", "INT64": "int64_t", "UINT64": "uint64_t",
         "FP32": "float", "FP64": "double"}

types = [("GrB_" + t, "T_" + t) for t in REAL] + [("GxB_FC32", "T_FC32"), ("GxB_FC64", "T_FC64")]

unops = []   # (cname, opcode, xtype, ztype)
binops = []  # (cname, opcode, xtype, ytype, ztype)
monoids = []  # (cname, binop cname)
semirings = []  # (cname, monoid cname, binop cname)
_bin_by_key = {}
_mon_by_key = {}


def add_binop(cname, op, x, y, z):
    binops.append((cname, op, x, y, z))
    _bin_by_key.setdefault((op, x), cname)


# ---- unary ops -----------------------------------------------------------------------------
for t in REAL:
    for pre, op in [("GrB", "IDENTITY"), ("GrB", "AINV"), ("GrB", "MINV"), ("GrB", "ABS"), ("GxB", "ABS"),
                    ("GxB", "ONE"), ("GxB", "LNOT")]:
        unops.append((f"{pre}_{op}_{t}", "U_" + op, t, t))
unops.append(("GrB_LNOT", "U_LNOT", "BOOL", "BOOL"))
for t in INTS:
    unops.append((f"GrB_BNOT_{t}", "U_BNOT", t, t))
for t in FLOATS:
    for op in ["SQRT", "LOG", "EXP", "LOG2", "SIN", "COS", "TAN", "ACOS", "ASIN", "ATAN", "SINH", "COSH", "TANH",
               "ACOSH", "ASINH", "ATANH", "SIGNUM", "CEIL", "FLOOR", "ROUND", "TRUNC", "EXP2", "EXPM1", "LOG10",
               "LOG1P", "LGAMMA", "TGAMMA", "ERF", "ERFC", "FREXPX", "FREXPE"]:
        unops.append((f"GxB_{op}_{t}", "U_" + op, t, t))
    for op in ["ISINF", "ISNAN", "ISFINITE"]:
        unops.append((f"GxB_{op}_{t}", "U_" + op, t, "BOOL"))
# positional unary operators (round 6; pygraphblas/unaryop.py:55-63 lists them): the result is the entry's row / column index (0- or 1-based), whatever the input holds
for t in ["INT32", "INT64"]:
    for op in ["POSITIONI", "POSITIONI1", "POSITIONJ", "POSITIONJ1"]:
        unops.append((f"GxB_{op}_{t}", "U_" + op, t, t))

# ---- binary ops ----------------------------------------------------------------------------
for t in REAL:
    for op in ["FIRST", "SECOND", "MIN", "MAX", "PLUS", "MINUS", "TIMES", "DIV"]:
        add_binop(f"GrB_{op}_{t}", "B_" + op, t, t, t)
    for op in ["EQ", "NE", "GT", "LT", "GE", "LE"]:
        add_binop(f"GrB_{op}_{t}", "B_" + op, t, t, "BOOL")
    for op in ["PAIR", "ANY", "RMINUS", "RDIV", "POW", "ISEQ", "ISNE", "ISGT", "ISLT", "ISGE", "ISLE",
               "LOR", "LAND", "LXOR"]:
        add_binop(f"GxB_{op}_{t}", "B_" + op, t, t, t)
for op in ["LOR", "LAND", "LXOR", "LXNOR"]:
    add_binop(f"GrB_{op}", "B_" + op, "BOOL", "BOOL", "BOOL")
_bin_by_key[("B_LXNOR", "BOOL")] = "GrB_LXNOR"
for t in INTS:
    for op in ["BOR", "BAND", "BXOR", "BXNOR"]:
        add_binop(f"GrB_{op}_{t}", "B_" + op, t, t, t)
    for op in ["BGET", "BSET", "BCLR"]:
        add_binop(f"GxB_{op}_{t}", "B_" + op, t, t, t)
for t in FLOATS:
    for op in ["ATAN2", "HYPOT", "FMOD", "REMAINDER", "COPYSIGN", "LDEXP"]:
        add_binop(f"GxB_{op}_{t}", "B_" + op, t, t, t)


def bin_of(op, t):
    return _bin_by_key[("B_" + op, t)]


# ---- monoids -------------------------------------------------------------------------------
def add_monoid(cname, op, t):
    monoids.append((cname, bin_of(op, t)))
    _mon_by_key.setdefault((op, t), cname)


for t in NUM:
    for op in ["MIN", "MAX", "PLUS", "TIMES"]:
        add_monoid(f"GrB_{op}_MONOID_{t}", op, t)
        add_monoid(f"GxB_{op}_{t}_MONOID", op, t)
    add_monoid(f"GxB_ANY_{t}_MONOID", "ANY", t)
for op in ["LOR", "LAND", "LXOR", "LXNOR"]:
    add_monoid(f"GrB_{op}_MONOID_BOOL", op, "BOOL")
for op in ["ANY", "LOR", "LAND", "LXOR"]:
    add_monoid(f"GxB_{op}_BOOL_MONOID", op, "BOOL")
monoids.append(("GxB_EQ_BOOL_MONOID", "GrB_LXNOR"))
_mon_by_key[("EQ", "BOOL")] = "GxB_EQ_BOOL_MONOID"
for t in UINTS:
    for op in ["BOR", "BAND", "BXOR", "BXNOR"]:
        add_monoid(f"GxB_{op}_{t}_MONOID", op, t)


def mon_of(op, t):
    return _mon_by_key[(op, t)]


# ---- semirings (same-type multiply; comparison-multiply semirings are not declared) ---------
MULS = ["FIRST", "SECOND", "PAIR", "MIN", "MAX", "PLUS", "MINUS", "RMINUS", "TIMES", "DIV", "RDIV",
        "ISEQ", "ISNE", "ISGT", "ISLT", "ISGE", "ISLE", "LOR", "LAND", "LXOR"]
for t in NUM:
    for add in ["MIN", "MAX", "PLUS", "TIMES", "ANY"]:
        for mul in MULS:
            semirings.append((f"GxB_{add}_{mul}_{t}", mon_of(add, t), bin_of(mul, t)))
    for add, mul in [("PLUS", "TIMES"), ("PLUS", "MIN"), ("MIN", "PLUS"), ("MIN", "TIMES"), ("MIN", "FIRST"),
                     ("MIN", "SECOND"), ("MIN", "MAX"), ("MAX", "PLUS"), ("MAX", "TIMES"), ("MAX", "FIRST"),
                     ("MAX", "SECOND"), ("MAX", "MIN")]:
        semirings.append((f"GrB_{add}_{mul}_SEMIRING_{t}", mon_of(add, t), bin_of(mul, t)))
BOOL_MULS = ["FIRST", "SECOND", "PAIR", "LOR", "LAND", "LXOR", "EQ", "GT", "LT", "GE", "LE"]
for add in ["LOR", "LAND", "LXOR", "EQ", "ANY"]:
    for mul in BOOL_MULS:
        semirings.append((f"GxB_{add}_{mul}_BOOL", mon_of(add, "BOOL"), bin_of(mul, "BOOL")))
for add, mul in [("LOR", "LAND"), ("LAND", "LOR"), ("LXOR", "LAND")]:
    semirings.append((f"GrB_{add}_{mul}_SEMIRING_BOOL", mon_of(add, "BOOL"), bin_of(mul, "BOOL")))
semirings.append(("GrB_LXNOR_LOR_SEMIRING_BOOL", "GrB_LXNOR_MONOID_BOOL", "GrB_LOR"))

# ---- descriptors ---------------------------------------------------------------------------
descs = []
for r in ["", "R"]:
    for s in ["", "S"]:
        for c in ["", "C"]:
            for tt in ["", "T0", "T1", "T0T1"]:
                nm = r + s + c + tt
                if nm:
                    descs.append(nm)
assert len(descs) == 31

selectops = ["TRIL", "TRIU", "DIAG", "OFFDIAG", "NONZERO", "EQ_ZERO", "GT_ZERO", "GE_ZERO", "LT_ZERO", "LE_ZERO",
             "NE_THUNK", "EQ_THUNK", "GT_THUNK", "GE_THUNK", "LT_THUNK", "LE_THUNK"]

# ============================================================================================
# registry_gen.inc
# ============================================================================================
inc = ["// GENERATED by tools/gen_api.py — do not edit.\n"]
for cname, code in types:
    inc.append(f'static GrB_Type_opaque ty_{cname} = {{GRB_MAGIC, grb::{code}, 0, "{cname}"}};\n'
               f'extern "C" GrB_Type {cname} = &ty_{cname};\n')


def ty(t):
    return "&ty_GrB_" + t


for cname, op, x, z in unops:
    inc.append(f'static GrB_UnaryOp_opaque uo_{cname} = {{GRB_MAGIC, grb::{op}, {ty(x)}, {ty(z)}, "{cname}", nullptr}};\n'
               f'extern "C" GrB_UnaryOp {cname} = &uo_{cname};\n')
for cname, op, x, y, z in binops:
    inc.append(f'static GrB_BinaryOp_opaque bo_{cname} = {{GRB_MAGIC, grb::{op}, {ty(x)}, {ty(y)}, {ty(z)}, "{cname}", nullptr}};\n'
               f'extern "C" GrB_BinaryOp {cname} = &bo_{cname};\n')
for cname, b in monoids:
    inc.append(f'static GrB_Monoid_opaque mo_{cname} = {{GRB_MAGIC, &bo_{b}, {{0}}, false, {{0}}, "{cname}", true}};\n'
               f'extern "C" GrB_Monoid {cname} = &mo_{cname};\n')
for cname, m, b in semirings:
    inc.append(f'static GrB_Semiring_opaque sr_{cname} = {{GRB_MAGIC, &mo_{m}, &bo_{b}, "{cname}", true}};\n'
               f'extern "C" GrB_Semiring {cname} = &sr_{cname};\n')
for nm in descs:
    outp = "GrB_REPLACE" if "R" in nm.replace("T0", "").replace("T1", "") else "GxB_DEFAULT"
    core = nm.replace("T0", "").replace("T1", "")
    mask = ("GrB_STRUCTURE" if "S" in core else "0") + " + " + ("GrB_COMP" if "C" in core else "0")
    inp0 = "GrB_TRAN" if "T0" in nm else "GxB_DEFAULT"
    inp1 = "GrB_TRAN" if "T1" in nm else "GxB_DEFAULT"
    inc.append(f'static GrB_Descriptor_opaque de_{nm} = {{GRB_MAGIC, {outp}, {mask}, {inp0}, {inp1}, 0, 0, 0, 0.0, true, "{nm}"}};\n'
               f'extern "C" GrB_Descriptor GrB_DESC_{nm} = &de_{nm};\n')
for i, nm in enumerate(selectops):
    inc.append(f'static GxB_SelectOp_opaque so_{nm} = {{GRB_MAGIC, SEL_{nm}, "GxB_{nm}", nullptr, nullptr, nullptr}};\n'
               f'extern "C" GxB_SelectOp GxB_{nm} = &so_{nm};\n')
inc.append("static GrB_Type_opaque* const all_types[] = {" + ", ".join(f"&ty_{c}" for c, _ in types) + "};\n")
inc.append("static GrB_Monoid_opaque* const all_monoids[] = {" + ", ".join(f"&mo_{c}" for c, _ in monoids) + "};\n")
inc.append("static GrB_BinaryOp_opaque* const all_binops[] = {" + ", ".join(f"&bo_{c[0]}" for c in binops) + "};\n")
inc.append("static GrB_UnaryOp_opaque* const all_unops[] = {" + ", ".join(f"&uo_{c[0]}" for c in unops) + "};\n")
inc.append("static GrB_Semiring_opaque* const all_semirings[] = {" + ", ".join(f"&sr_{c[0]}" for c in semirings) + "};\n")
with open(os.path.join(ROOT, "pygraphblas_amd/csrc/registry_gen.inc"), "w") as f:
    f.write("".join(inc))

# ============================================================================================
# include/grb_mi355x.h
# ============================================================================================
H = []
H.append("""/* grb_mi355x.h — C ABI of libgrb_mi355x.so, an MI355X (gfx950) GraphBLAS hot-path backend.
 *
 * GENERATED by tools/gen_api.py.  The file is deliberately plain C declarations without
 * preprocessor logic so that a CFFI `ffi.cdef()` can consume it verbatim (shim/).
 *
 * What each group replaces in the reference (Graphegon/pygraphblas, files under /root/reference):
 *   GrB_mxm   <- lib.GrB_mxm   called at pygraphblas/matrix.py:2572-2583   (Matrix.mxm)
 *   GrB_mxv   <- lib.GrB_mxv   called at pygraphblas/matrix.py:2714-2725   (Matrix.mxv)
 *   GrB_vxm   <- lib.GrB_vxm   called at pygraphblas/vector.py:960-970     (Vector.vxm)
 *   GrB_Matrix_reduce_<T> <- pygraphblas/matrix.py:1799-1803 (Matrix.reduce_int et al.)
 *   GrB_Vector_reduce_<T> <- pygraphblas/vector.py:1132-1202
 *   object model (new, free, build, setElement, extractElement, extractTuples, nvals, ...) <- the `lib.` calls listed by
 *       grep -oh "lib\\.G[rx]B_[A-Za-z0-9_]*" pygraphblas/*.py  (SURVEY.md §8b)
 *   error convention: positive GrB_Info codes 0..13 as mapped at pygraphblas/base.py:189-203.
 * Signatures are GraphBLAS C API 1.3 with SuiteSparse v5.1-era GxB_ extensions.
 * GrBX_* functions are extensions of this backend (bulk/device import-export, timing).
 */
typedef uint64_t GrB_Index;
typedef int GrB_Info;
typedef int GrB_Mode;
typedef int GrB_Desc_Field;
typedef int GrB_Desc_Value;
typedef int GxB_Option_Field;
typedef int GxB_Format_Value;
typedef void (*GxB_unary_function)(void *, const void *);
typedef void (*GxB_binary_function)(void *, const void *, const void *);
typedef bool (*GxB_select_function)(GrB_Index i, GrB_Index j, const void *x, const void *thunk);
typedef struct GrB_Type_opaque *GrB_Type;
typedef struct GrB_UnaryOp_opaque *GrB_UnaryOp;
typedef struct GrB_BinaryOp_opaque *GrB_BinaryOp;
typedef struct GrB_Monoid_opaque *GrB_Monoid;
typedef struct GrB_Semiring_opaque *GrB_Semiring;
typedef struct GrB_Descriptor_opaque *GrB_Descriptor;
typedef struct GxB_SelectOp_opaque *GxB_SelectOp;
typedef struct GxB_Scalar_opaque *GxB_Scalar;
typedef struct GrB_Vector_opaque *GrB_Vector;
typedef struct GrB_Matrix_opaque *GrB_Matrix;

/* GrB_Info values */
#define GrB_SUCCESS 0
#define GrB_NO_VALUE 1
#define GrB_UNINITIALIZED_OBJECT 2
#define GrB_INVALID_OBJECT 3
#define GrB_NULL_POINTER 4
#define GrB_INVALID_VALUE 5
#define GrB_INVALID_INDEX 6
#define GrB_DOMAIN_MISMATCH 7
#define GrB_DIMENSION_MISMATCH 8
#define GrB_OUTPUT_NOT_EMPTY 9
#define GrB_OUT_OF_MEMORY 10
#define GrB_INSUFFICIENT_SPACE 11
#define GrB_INDEX_OUT_OF_BOUNDS 12
#define GrB_PANIC 13
/* modes */
#define GrB_NONBLOCKING 0
#define GrB_BLOCKING 1
/* descriptor fields and values */
#define GrB_OUTP 0
#define GrB_MASK 1
#define GrB_INP0 2
#define GrB_INP1 3
#define GxB_DESCRIPTOR_NTHREADS 5
#define GxB_DESCRIPTOR_CHUNK 7
#define GxB_AxB_METHOD 1000
#define GxB_SORT 35
#define GxB_DEFAULT 0
#define GrB_REPLACE 1
#define GrB_COMP 2
#define GrB_TRAN 3
#define GrB_STRUCTURE 4
#define GxB_AxB_GUSTAVSON 1001
#define GxB_AxB_DOT 1003
#define GxB_AxB_HASH 1004
#define GxB_AxB_SAXPY 1005
/* options */
#define GxB_HYPER_SWITCH 0
#define GxB_BITMAP_SWITCH 34
#define GxB_FORMAT 1
#define GxB_GLOBAL_NTHREADS 5
#define GxB_GLOBAL_CHUNK 7
#define GxB_BURBLE 99
#define GxB_SPARSITY_STATUS 33
#define GxB_SPARSITY_CONTROL 32
#define GxB_BY_ROW 0
#define GxB_BY_COL 1
#define GxB_HYPERSPARSE 1
#define GxB_SPARSE 2
#define GxB_BITMAP 4
#define GxB_FULL 8
#define GxB_AUTO_SPARSITY 15
#define GxB_RANGE 9223372036854775807
#define GxB_STRIDE 9223372036854775806
#define GxB_BACKWARDS 9223372036854775805
#define GxB_BEGIN 0
#define GxB_END 1
#define GxB_INC 2
#define GxB_INDEX_MAX 1152921504606846976
#define GxB_IMPLEMENTATION_MAJOR 0
#define GxB_IMPLEMENTATION_MINOR 1
#define GxB_IMPLEMENTATION_SUB 0
#define GxB_SPEC_MAJOR 1
#define GxB_SPEC_MINOR 3
#define GxB_SPEC_SUB 0

extern const uint64_t *GrB_ALL;

""")
H.append("/* ---- built-in types ---- */\n")
for cname, _ in types:
    H.append(f"extern GrB_Type {cname};\n")
H.append("/* ---- built-in unary operators ---- */\n")
for c in unops:
    H.append(f"extern GrB_UnaryOp {c[0]};\n")
H.append("/* ---- built-in binary operators ---- */\n")
for c in binops:
    H.append(f"extern GrB_BinaryOp {c[0]};\n")
H.append("/* ---- built-in monoids ---- */\n")
for c in monoids:
    H.append(f"extern GrB_Monoid {c[0]};\n")
H.append("/* ---- built-in semirings ---- */\n")
for c in semirings:
    H.append(f"extern GrB_Semiring {c[0]};\n")
H.append("/* ---- predefined descriptors ---- */\n")
for nm in descs:
    H.append(f"extern GrB_Descriptor GrB_DESC_{nm};\n")
H.append("/* ---- select operators ---- */\n")
for nm in selectops:
    H.append(f"extern GxB_SelectOp GxB_{nm};\n")

H.append("""
/* ---- context ---- */
GrB_Info GrB_init(int mode);
GrB_Info GxB_init(int mode, void *(*user_malloc)(size_t), void *(*user_calloc)(size_t, size_t),
                  void *(*user_realloc)(void *, size_t), void (*user_free)(void *), bool user_malloc_is_thread_safe);
GrB_Info GrB_finalize(void);
GrB_Info GrB_getVersion(unsigned int *version, unsigned int *subversion);
GrB_Info GxB_Global_Option_set(int field, ...);
GrB_Info GxB_Global_Option_get(int field, ...);

/* ---- introspection of operator objects ---- */
GrB_Info GxB_Semiring_add(GrB_Monoid *add, GrB_Semiring semiring);
GrB_Info GxB_Semiring_multiply(GrB_BinaryOp *multiply, GrB_Semiring semiring);
GrB_Info GxB_Monoid_operator(GrB_BinaryOp *op, GrB_Monoid monoid);
GrB_Info GxB_BinaryOp_ztype(GrB_Type *ztype, GrB_BinaryOp op);
GrB_Info GxB_BinaryOp_xtype(GrB_Type *xtype, GrB_BinaryOp op);
GrB_Info GxB_BinaryOp_ytype(GrB_Type *ytype, GrB_BinaryOp op);
GrB_Info GxB_UnaryOp_ztype(GrB_Type *ztype, GrB_UnaryOp op);
GrB_Info GxB_UnaryOp_xtype(GrB_Type *xtype, GrB_UnaryOp op);
GrB_Info GxB_Type_size(size_t *size, GrB_Type type);
GrB_Info GxB_Semiring_fprint(GrB_Semiring semiring, const char *name, int pr, FILE *f);
GrB_Info GxB_Monoid_fprint(GrB_Monoid monoid, const char *name, int pr, FILE *f);
GrB_Info GxB_BinaryOp_fprint(GrB_BinaryOp op, const char *name, int pr, FILE *f);
GrB_Info GxB_UnaryOp_fprint(GrB_UnaryOp op, const char *name, int pr, FILE *f);
GrB_Info GxB_SelectOp_fprint(GxB_SelectOp op, const char *name, int pr, FILE *f);
GrB_Info GxB_Matrix_fprint(GrB_Matrix A, const char *name, int pr, FILE *f);
GrB_Info GxB_Vector_fprint(GrB_Vector v, const char *name, int pr, FILE *f);

/* ---- descriptors ---- */
GrB_Info GrB_Descriptor_new(GrB_Descriptor *descriptor);
GrB_Info GrB_Descriptor_set(GrB_Descriptor desc, int field, int val);
GrB_Info GxB_Desc_get(GrB_Descriptor desc, int field, ...);
GrB_Info GxB_Desc_set(GrB_Descriptor desc, int field, ...);
GrB_Info GrB_Descriptor_free(GrB_Descriptor *descriptor);

/* ---- user-defined algebra made of built-in operators ---- */
GrB_Info GrB_Semiring_new(GrB_Semiring *semiring, GrB_Monoid add, GrB_BinaryOp multiply);
GrB_Info GrB_Semiring_free(GrB_Semiring *semiring);
GrB_Info GrB_Monoid_free(GrB_Monoid *monoid);

/* ---- matrices ---- */
GrB_Info GrB_Matrix_new(GrB_Matrix *A, GrB_Type type, GrB_Index nrows, GrB_Index ncols);
GrB_Info GrB_Matrix_dup(GrB_Matrix *C, const GrB_Matrix A);
GrB_Info GrB_Matrix_clear(GrB_Matrix A);
GrB_Info GrB_Matrix_free(GrB_Matrix *A);
GrB_Info GrB_Matrix_nrows(GrB_Index *nrows, const GrB_Matrix A);
GrB_Info GrB_Matrix_ncols(GrB_Index *ncols, const GrB_Matrix A);
GrB_Info GrB_Matrix_nvals(GrB_Index *nvals, const GrB_Matrix A);
GrB_Info GrB_Matrix_wait(GrB_Matrix *A);
GrB_Info GrB_Matrix_error(const char **error, const GrB_Matrix A);
GrB_Info GrB_Matrix_resize(GrB_Matrix A, GrB_Index nrows_new, GrB_Index ncols_new);
GrB_Info GrB_Matrix_removeElement(GrB_Matrix C, GrB_Index i, GrB_Index j);
GrB_Info GxB_Matrix_type(GrB_Type *type, const GrB_Matrix A);
GrB_Info GxB_Matrix_Option_set(GrB_Matrix A, int field, ...);
GrB_Info GxB_Matrix_Option_get(GrB_Matrix A, int field, ...);
GrB_Info GxB_Matrix_memoryUsage(size_t *size, const GrB_Matrix A);

/* ---- vectors ---- */
GrB_Info GrB_Vector_new(GrB_Vector *v, GrB_Type type, GrB_Index n);
GrB_Info GrB_Vector_dup(GrB_Vector *w, const GrB_Vector u);
GrB_Info GrB_Vector_clear(GrB_Vector v);
GrB_Info GrB_Vector_free(GrB_Vector *v);
GrB_Info GrB_Vector_size(GrB_Index *n, const GrB_Vector v);
GrB_Info GrB_Vector_nvals(GrB_Index *nvals, const GrB_Vector v);
GrB_Info GrB_Vector_wait(GrB_Vector *v);
GrB_Info GrB_Vector_error(const char **error, const GrB_Vector v);
GrB_Info GrB_Vector_resize(GrB_Vector v, GrB_Index n_new);
GrB_Info GrB_Vector_removeElement(GrB_Vector v, GrB_Index i);
GrB_Info GxB_Vector_type(GrB_Type *type, const GrB_Vector v);
GrB_Info GxB_Vector_Option_set(GrB_Vector v, int field, ...);
GrB_Info GxB_Vector_Option_get(GrB_Vector v, int field, ...);
GrB_Info GxB_Vector_memoryUsage(size_t *size, const GrB_Vector v);

/* ---- scalars ---- */
GrB_Info GxB_Scalar_new(GxB_Scalar *s, GrB_Type type);
GrB_Info GxB_Scalar_dup(GxB_Scalar *s, const GxB_Scalar t);
GrB_Info GxB_Scalar_clear(GxB_Scalar s);
GrB_Info GxB_Scalar_free(GxB_Scalar *s);
GrB_Info GxB_Scalar_nvals(GrB_Index *nvals, const GxB_Scalar s);
GrB_Info GxB_Scalar_wait(GxB_Scalar *s);
GrB_Info GxB_Scalar_type(GrB_Type *type, const GxB_Scalar s);

/* ==== THE HOT PATH (HIP kernels; fail with GrB_PANIC when no gfx950 device is present) ==== */
GrB_Info GrB_mxm(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_Semiring semiring,
                 const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc);
GrB_Info GrB_mxv(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Semiring semiring,
                 const GrB_Matrix A, const GrB_Vector u, const GrB_Descriptor desc);
GrB_Info GrB_vxm(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Semiring semiring,
                 const GrB_Vector u, const GrB_Matrix A, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_reduce_Monoid(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum,
                 const GrB_Monoid monoid, const GrB_Matrix A, const GrB_Descriptor desc);
GrB_Info GrB_transpose(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_Matrix A,
                 const GrB_Descriptor desc);

/* ---- O(n) / O(nnz) companions of the hot path ("next" rows of SURVEY.md §8f) ---- */
GrB_Info GrB_Vector_eWiseAdd_BinaryOp(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum,
                 const GrB_BinaryOp op, const GrB_Vector u, const GrB_Vector v, const GrB_Descriptor desc);
GrB_Info GrB_Vector_eWiseAdd_Monoid(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum,
                 const GrB_Monoid op, const GrB_Vector u, const GrB_Vector v, const GrB_Descriptor desc);
GrB_Info GrB_Vector_eWiseAdd_Semiring(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum,
                 const GrB_Semiring op, const GrB_Vector u, const GrB_Vector v, const GrB_Descriptor desc);
GrB_Info GrB_Vector_eWiseMult_BinaryOp(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum,
                 const GrB_BinaryOp op, const GrB_Vector u, const GrB_Vector v, const GrB_Descriptor desc);
GrB_Info GrB_Vector_eWiseMult_Monoid(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum,
                 const GrB_Monoid op, const GrB_Vector u, const GrB_Vector v, const GrB_Descriptor desc);
GrB_Info GrB_Vector_eWiseMult_Semiring(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum,
                 const GrB_Semiring op, const GrB_Vector u, const GrB_Vector v, const GrB_Descriptor desc);
GrB_Info GrB_Vector_apply(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_UnaryOp op,
                 const GrB_Vector u, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_eWiseAdd_BinaryOp(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum,
                 const GrB_BinaryOp op, const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_eWiseAdd_Monoid(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum,
                 const GrB_Monoid op, const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_eWiseAdd_Semiring(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum,
                 const GrB_Semiring op, const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_eWiseMult_BinaryOp(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum,
                 const GrB_BinaryOp op, const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_eWiseMult_Monoid(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum,
                 const GrB_Monoid op, const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_eWiseMult_Semiring(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum,
                 const GrB_Semiring op, const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_apply(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_UnaryOp op,
                 const GrB_Matrix A, const GrB_Descriptor desc);
GrB_Info GxB_Matrix_select(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GxB_SelectOp op,
                 const GrB_Matrix A, const GxB_Scalar Thunk, const GrB_Descriptor desc);
GrB_Info GxB_Vector_select(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GxB_SelectOp op,
                 const GrB_Vector u, const GxB_Scalar Thunk, const GrB_Descriptor desc);

/* ---- backend extensions: bulk / device-resident import-export, timing, stream ---- */
/* location: 0 = host pointers, 1 = device (HBM) pointers.  Arrays are copied. */
GrB_Info GrBX_Matrix_import_CSR(GrB_Matrix *A, GrB_Type type, GrB_Index nrows, GrB_Index ncols, GrB_Index nvals,
                 const uint32_t *rowptr, const uint32_t *colidx, const void *values, int location);
GrB_Info GrBX_Matrix_export_CSR(const GrB_Matrix A, uint32_t *rowptr, uint32_t *colidx, void *values, int location);
GrB_Info GrBX_Vector_import_Full(GrB_Vector *v, GrB_Type type, GrB_Index n, const void *values, int location);
GrB_Info GrBX_Vector_import_Bitmap(GrB_Vector *v, GrB_Type type, GrB_Index n, const void *values,
                 const uint8_t *present, int location);
GrB_Info GrBX_Vector_export_Bitmap(const GrB_Vector v, void *values, uint8_t *present, int location);
/* raw HBM addresses of a vector's value / presence arrays (valid until the vector is next written) */
GrB_Info GrBX_Vector_device_view(GrB_Vector v, void **values, uint8_t **present, GrB_Index *nvals);
GrB_Info GrBX_Vector_device_touch(GrB_Vector v);   /* caller wrote through the device view: recount entries */
GrB_Info GrBX_set_stream(void *hip_stream);        /* all kernels are launched on this HIP stream (default: 0) */
GrB_Info GrBX_device_synchronize(void);
GrB_Info GrBX_timer_start(void);                   /* hipEventRecord on the library stream */
GrB_Info GrBX_timer_stop(float *milliseconds);     /* hipEventRecord + synchronize + elapsed */
GrB_Info GrBX_device_info(char *name, int name_len, int *compute_units, size_t *hbm_bytes);
GrB_Info GrBX_memory_in_use(size_t *bytes);
GrB_Info GrBX_last_kernel_plan(char *buf, int len); /* which kernels the last hot-path call launched */
GrB_Info GrBX_last_plan_build_ms(float *milliseconds); /* device time of the most recent SpMV plan build (kernel X), 0 if none */
GrB_Info GrBX_lazy_stats(uint64_t *chains, uint64_t *nodes, uint64_t *fills_folded, uint64_t *reduces_fused); /* non-blocking mode: element-wise chain kernels run, operations they carried, `w(:) = s` fills folded into a product's store, reductions fused into a chain */
GrB_Info GrBX_chain_jit_stats(uint64_t *compiled, uint64_t *launched); /* deferred element-wise chains compiled with hipRTC (grb_chain_jit.cpp): kernels compiled, launches through them */
GrB_Info GrBX_chain_jit_stats2(uint64_t *compiled, uint64_t *launched, uint64_t *loaded_from_disk); /* ... and the kernels whose code object came from the disk cache (GRB_MI355X_CACHE_DIR) instead of a compilation */
GrB_Info GrBX_exact_sum_host(const double *terms, uint64_t n, int significand_bits, double *sum, int *unit_exp_out); /* the 128-bit integer accumulation of the masked product's deterministic mode (grb_exact.hpp) run on the host over `terms`: the exact sum rounded once to 53 / 24 bits */
GrB_Info GrBX_Vector_iseq(bool *equal, const GrB_Vector u, const GrB_Vector v); /* same size, pattern and values of two vectors of one built-in real type in ONE pass (pygraphblas/vector.py:188-235 Vector.iseq composes it from five calls); GrB_NO_VALUE when that is not the case at hand (types differ, complex, hypersparse): compose it then */
GrB_Info GrBX_xcd_mapping(char *buf, int len);      /* how workgroups of a full-chip launch map to XCDs ("roundrobin8", or what was observed) */
/* The exchange steps of the row-partitioned path (one process per GPU; RCCL over xGMI; grb_dist.cpp).  The reference has no
 * distributed code: these replace nothing in it, they are what BASELINE.json's north star adds (SURVEY.md section 8e). */
/* Index-list extract / assign, kronecker and apply with a GxB_Scalar operand: the element-wise container surface, computed on the
 * host mirror like setElement (grb_host_ops.cpp) — not part of the HIP hot path. */
GrB_Info GrB_Vector_extract(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Vector u, const GrB_Index *I, GrB_Index ni, const GrB_Descriptor desc);
GrB_Info GrB_Col_extract(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Matrix A, const GrB_Index *I, GrB_Index ni, GrB_Index j, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_extract(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_Matrix A, const GrB_Index *I, GrB_Index ni, const GrB_Index *J, GrB_Index nj, const GrB_Descriptor desc);
GrB_Info GrB_Vector_assign(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Vector u, const GrB_Index *I, GrB_Index ni, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_assign(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_Matrix A, const GrB_Index *I, GrB_Index ni, const GrB_Index *J, GrB_Index nj, const GrB_Descriptor desc);
GrB_Info GrB_Row_assign(GrB_Matrix C, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Vector u, GrB_Index i, const GrB_Index *J, GrB_Index nj, const GrB_Descriptor desc);
GrB_Info GrB_Col_assign(GrB_Matrix C, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_Vector u, const GrB_Index *I, GrB_Index ni, GrB_Index j, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_kronecker_BinaryOp(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Matrix A, const GrB_Matrix B, const GrB_Descriptor desc);
GrB_Info GxB_Matrix_diag(GrB_Matrix C, const GrB_Vector v, int64_t k, const GrB_Descriptor desc);   /* Matrix.from_diag, pygraphblas/matrix.py:333-375 (host mirror) */
GrB_Info GxB_Vector_diag(GrB_Vector v, const GrB_Matrix A, int64_t k, const GrB_Descriptor desc);   /* Matrix.vector_diag, pygraphblas/matrix.py:2225-2277 (host mirror) */
GrB_Info GxB_Matrix_apply_BinaryOp1st(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GxB_Scalar x, const GrB_Matrix A, const GrB_Descriptor desc);
GrB_Info GxB_Matrix_apply_BinaryOp2nd(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Matrix A, const GxB_Scalar y, const GrB_Descriptor desc);
GrB_Info GxB_Vector_apply_BinaryOp1st(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GxB_Scalar x, const GrB_Vector u, const GrB_Descriptor desc);
GrB_Info GxB_Vector_apply_BinaryOp2nd(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Vector u, const GxB_Scalar y, const GrB_Descriptor desc);
extern double GxB_ALWAYS_HYPER;
extern double GxB_NEVER_HYPER;
extern double GxB_HYPER_DEFAULT;
GrB_Info GrBX_dist_unique_id(void *id, int len);      /* 128 bytes, made on one rank and handed to all (any channel) */
GrB_Info GrBX_dist_init(int rank, int world, const void *id, int len);   /* ncclCommInitRank on this process's GPU */
GrB_Info GrBX_dist_finalize(void);
GrB_Info GrBX_dist_info(int *rank, int *world);
GrB_Info GrBX_dist_transport(char *buf, int len);      /* file name of the RCCL-ABI library bound for the exchange (librccl, or a test stand-in named by GRB_MI355X_RCCL); "" before the first use */
GrB_Info GrBX_Vector_allgatherv_start(GrB_Vector full, const GrB_Vector local, const GrB_Index *bounds, int presence);
GrB_Info GrBX_dist_wait(void);                        /* the compute stream waits for the exchange started above */
GrB_Info GrBX_Vector_allgatherv(GrB_Vector full, const GrB_Vector local, const GrB_Index *bounds, int presence);
GrB_Info GrBX_Vector_allgatherv_bits(GrB_Vector full, const GrB_Vector local, const GrB_Index *bounds);   /* BOOL frontier, 1 bit per vertex */
GrB_Info GrBX_dist_allreduce(void *host_buf, GrB_Index count, GrB_Type type, GrB_BinaryOp op);
""")

# typed families
for t in REAL:
    c = CTYPE[t]
    H.append(f"""
GrB_Info GrB_Matrix_build_{t}(GrB_Matrix C, const GrB_Index *I, const GrB_Index *J, const {c} *X, GrB_Index nvals, const GrB_BinaryOp dup);
GrB_Info GrB_Matrix_setElement_{t}(GrB_Matrix C, {c} x, GrB_Index i, GrB_Index j);
GrB_Info GrB_Matrix_extractElement_{t}({c} *x, const GrB_Matrix A, GrB_Index i, GrB_Index j);
GrB_Info GrB_Matrix_extractTuples_{t}(GrB_Index *I, GrB_Index *J, {c} *X, GrB_Index *nvals, const GrB_Matrix A);
GrB_Info GrB_Matrix_reduce_{t}({c} *c, const GrB_BinaryOp accum, const GrB_Monoid monoid, const GrB_Matrix A, const GrB_Descriptor desc);
GrB_Info GrB_Vector_build_{t}(GrB_Vector w, const GrB_Index *I, const {c} *X, GrB_Index nvals, const GrB_BinaryOp dup);
GrB_Info GrB_Vector_setElement_{t}(GrB_Vector w, {c} x, GrB_Index i);
GrB_Info GrB_Vector_extractElement_{t}({c} *x, const GrB_Vector v, GrB_Index i);
GrB_Info GrB_Vector_extractTuples_{t}(GrB_Index *I, {c} *X, GrB_Index *nvals, const GrB_Vector v);
GrB_Info GrB_Vector_reduce_{t}({c} *c, const GrB_BinaryOp accum, const GrB_Monoid monoid, const GrB_Vector u, const GrB_Descriptor desc);
GrB_Info GrB_Vector_assign_{t}(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, {c} x, const GrB_Index *I, GrB_Index ni, const GrB_Descriptor desc);
GrB_Info GrB_Matrix_assign_{t}(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, {c} x, const GrB_Index *I, GrB_Index ni, const GrB_Index *J, GrB_Index nj, const GrB_Descriptor desc);
GrB_Info GxB_Vector_apply_BinaryOp1st_{t}(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, {c} x, const GrB_Vector u, const GrB_Descriptor desc);
GrB_Info GxB_Vector_apply_BinaryOp2nd_{t}(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Vector u, {c} y, const GrB_Descriptor desc);
GrB_Info GxB_Matrix_apply_BinaryOp1st_{t}(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, {c} x, const GrB_Matrix A, const GrB_Descriptor desc);
GrB_Info GxB_Matrix_apply_BinaryOp2nd_{t}(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Matrix A, {c} y, const GrB_Descriptor desc);
GrB_Info GxB_Scalar_setElement_{t}(GxB_Scalar s, {c} x);
GrB_Info GxB_Scalar_extractElement_{t}({c} *x, const GxB_Scalar s);
GrB_Info GrB_Monoid_new_{t}(GrB_Monoid *monoid, GrB_BinaryOp op, {c} identity);
""")
# complex types: the handles and the 17 typed entry points exist so that the reference's type registry imports
# (pygraphblas/types.py:87-110 resolves them for all 13 types).  Complex containers are kept on the host: entries can be
# built, set, read, listed and assigned; everything that would compute on them returns GrB_DOMAIN_MISMATCH
H.append("""
/* ---- complex types: host-side containers (build / set / extract / assign a scalar); arithmetic on them
 *      (reduce, apply, monoids, every device operation) returns GrB_DOMAIN_MISMATCH ---- */
typedef struct { float re; float im; } GxB_FC32_t;
typedef struct { double re; double im; } GxB_FC64_t;
""")
for t, c in (("FC32", "GxB_FC32_t"), ("FC64", "GxB_FC64_t")):
    H.append(f"""
GrB_Info GxB_Matrix_build_{t}(GrB_Matrix C, const GrB_Index *I, const GrB_Index *J, const {c} *X, GrB_Index nvals, const GrB_BinaryOp dup);
GrB_Info GxB_Vector_build_{t}(GrB_Vector w, const GrB_Index *I, const {c} *X, GrB_Index nvals, const GrB_BinaryOp dup);
GrB_Info GxB_Matrix_setElement_{t}(GrB_Matrix C, {c} x, GrB_Index i, GrB_Index j);
GrB_Info GxB_Matrix_extractElement_{t}({c} *x, const GrB_Matrix A, GrB_Index i, GrB_Index j);
GrB_Info GxB_Matrix_extractTuples_{t}(GrB_Index *I, GrB_Index *J, {c} *X, GrB_Index *nvals, const GrB_Matrix A);
GrB_Info GxB_Matrix_reduce_{t}({c} *c, const GrB_BinaryOp accum, const GrB_Monoid monoid, const GrB_Matrix A, const GrB_Descriptor desc);
GrB_Info GxB_Vector_setElement_{t}(GrB_Vector w, {c} x, GrB_Index i);
GrB_Info GxB_Vector_extractElement_{t}({c} *x, const GrB_Vector v, GrB_Index i);
GrB_Info GxB_Vector_extractTuples_{t}(GrB_Index *I, {c} *X, GrB_Index *nvals, const GrB_Vector v);
GrB_Info GxB_Vector_reduce_{t}({c} *c, const GrB_BinaryOp accum, const GrB_Monoid monoid, const GrB_Vector u, const GrB_Descriptor desc);
GrB_Info GxB_Vector_assign_{t}(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, {c} x, const GrB_Index *I, GrB_Index ni, const GrB_Descriptor desc);
GrB_Info GxB_Matrix_assign_{t}(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, {c} x, const GrB_Index *I, GrB_Index ni, const GrB_Index *J, GrB_Index nj, const GrB_Descriptor desc);
GrB_Info GxB_Vector_apply_BinaryOp1st_{t}(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, {c} x, const GrB_Vector u, const GrB_Descriptor desc);
GrB_Info GxB_Vector_apply_BinaryOp2nd_{t}(GrB_Vector w, const GrB_Vector mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Vector u, {c} y, const GrB_Descriptor desc);
GrB_Info GxB_Matrix_apply_BinaryOp1st_{t}(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, {c} x, const GrB_Matrix A, const GrB_Descriptor desc);
GrB_Info GxB_Matrix_apply_BinaryOp2nd_{t}(GrB_Matrix C, const GrB_Matrix Mask, const GrB_BinaryOp accum, const GrB_BinaryOp op, const GrB_Matrix A, {c} y, const GrB_Descriptor desc);
GrB_Info GxB_Scalar_setElement_{t}(GxB_Scalar s, {c} x);
GrB_Info GxB_Scalar_extractElement_{t}({c} *x, const GxB_Scalar s);
GrB_Info GxB_Monoid_new_{t}(GrB_Monoid *monoid, GrB_BinaryOp op, {c} identity);
""")
with open(os.path.join(ROOT, "include/grb_mi355x.h"), "w") as f:
    f.write("".join(H))
print(f"types={len(types)} unops={len(unops)} binops={len(binops)} monoids={len(monoids)} "
      f"semirings={len(semirings)} descs={len(descs)}")
