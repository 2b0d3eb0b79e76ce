// grb_complex_stubs.cpp — the FC32 / FC64 typed entry points that would COMPUTE on complex values.  They exist so
// that the reference's type registry (pygraphblas/types.py:87-110 resolves 17 typed functions for each of its 13
// types) can import; complex arithmetic is out of scope for the MI355X backend and each reports GrB_DOMAIN_MISMATCH.
// (Storing complex entries — build / setElement / extractElement / extractTuples / assign of a scalar — is host-side
// container work and lives in grb_container.cpp and grb_host_ops.cpp.)
#include "grb_api.hpp"

typedef struct { float re; float im; } GxB_FC32_t;
typedef struct { double re; double im; } GxB_FC64_t;

extern "C" {
#define GRB_COMPLEX_STUBS(SUF, CT) \
  GrB_Info GxB_Matrix_reduce_##SUF(CT*, const GrB_BinaryOp, const GrB_Monoid, const GrB_Matrix, const GrB_Descriptor) { return GrB_DOMAIN_MISMATCH; } \
  GrB_Info GxB_Vector_reduce_##SUF(CT*, const GrB_BinaryOp, const GrB_Monoid, const GrB_Vector, const GrB_Descriptor) { return GrB_DOMAIN_MISMATCH; } \
  GrB_Info GxB_Vector_apply_BinaryOp1st_##SUF(GrB_Vector, const GrB_Vector, const GrB_BinaryOp, const GrB_BinaryOp, CT, const GrB_Vector, const GrB_Descriptor) { return GrB_DOMAIN_MISMATCH; } \
  GrB_Info GxB_Vector_apply_BinaryOp2nd_##SUF(GrB_Vector, const GrB_Vector, const GrB_BinaryOp, const GrB_BinaryOp, const GrB_Vector, CT, const GrB_Descriptor) { return GrB_DOMAIN_MISMATCH; } \
  GrB_Info GxB_Matrix_apply_BinaryOp1st_##SUF(GrB_Matrix, const GrB_Matrix, const GrB_BinaryOp, const GrB_BinaryOp, CT, const GrB_Matrix, const GrB_Descriptor) { return GrB_DOMAIN_MISMATCH; } \
  GrB_Info GxB_Matrix_apply_BinaryOp2nd_##SUF(GrB_Matrix, const GrB_Matrix, const GrB_BinaryOp, const GrB_BinaryOp, const GrB_Matrix, CT, const GrB_Descriptor) { return GrB_DOMAIN_MISMATCH; } \
  GrB_Info GxB_Monoid_new_##SUF(GrB_Monoid*, GrB_BinaryOp, CT) { return GrB_DOMAIN_MISMATCH; }
GRB_COMPLEX_STUBS(FC32, GxB_FC32_t)
GRB_COMPLEX_STUBS(FC64, GxB_FC64_t)
}
