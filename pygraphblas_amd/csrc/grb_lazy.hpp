// grb_lazy.hpp — interface between the deferred-operation queue (grb_lazy.cpp) and its one kernel (grb_lazy_kernels.hip).
#pragma once
#include "grb_internal.hpp"

namespace grb {

constexpr int CHAIN_MAX_STEPS = 4, CHAIN_MAX_IN = 4, CHAIN_MAX_OUT = 4, CHAIN_PREV = 7;

struct ChainStepDesc {
  int kind;            // 0: eWise, 1: apply
  int op, mode;        // operator code; apply: 0 unary, 1 z = f(s, x), 2 z = f(x, s)
  int is_union;
  int src[2];          // stored operand slot 0..3, or CHAIN_PREV = the result of the step before
  int out;             // output slot that receives this step's result, or -1
  uint8_t scalar[16];
};
struct ChainReduce { int on, op, widen; uint8_t identity[16]; };      // widen: FP32 values reduced in FP64 (`reduce_float` of an FP32 vector)
struct ChainLaunch {
  int tcode; uint64_t n; int nsteps, next, nout; bool math;
  const void* ev[CHAIN_MAX_IN]; const uint8_t* ep[CHAIN_MAX_IN];      // stored operands: values, presence bytes (nullptr = every position present)
  void* ov[CHAIN_MAX_OUT]; uint8_t* op[CHAIN_MAX_OUT];                // outputs: values (zeros where absent), presence bytes (nullptr = not written: full result in place)
  ChainStepDesc st[CHAIN_MAX_STEPS];
  ChainReduce red;
};
// one pass over the n positions; with red.on the reduction of the last step's result lands in `red_result` (host, monoid's type)
template <class T> void vec_chain_launch_t(const ChainLaunch& L, void* red_result);      // grb_lazy_inst.hip, one instantiation per type
inline bool vec_chain_type_supported(int tcode) { return type_size(tcode) >= 4 && tcode < T_FC32; }   // 4- and 8-byte real types (the loops' FP32 / FP64 / INT32 / INT64 vectors)
void vec_chain_launch(const ChainLaunch& L, void* red_result);       // grb_lazy.cpp: dispatch on the value type
inline bool unop_needs_math_host(int op) { return op >= U_SQRT && op <= U_ISFINITE; }

// grb_chain_jit.cpp: the chain through the kernel hipRTC compiled for its steps (FP32 / FP64); false = run the interpreter.  red: 0 none, 1 in T, 2 FP32 widened
bool chain_jit_launch(const ChainLaunch& L, int tcode, int red, const void* rid, void* partial, unsigned grid, bool replaces_spec);      // tcode: the chain's value type (FP32 / FP64 and the 4- and 8-byte integers)
void set_nonblocking(bool on);
void lazy_flush();
// each returns false when the operation cannot be deferred (the caller then runs it the blocking way)
bool lazy_fill(GrB_Vector w, const void* s_in_w_type);
void lazy_fill_consumed(GrB_Vector w);
bool lazy_ewise(GrB_Vector w, GrB_BinaryOp op, GrB_Vector u, GrB_Vector v, bool is_union);
bool lazy_apply(GrB_Vector w, int mode, int opcode, int xcode, int zcode, const void* scalar_in_x_type, GrB_Vector u);
bool lazy_reduce(GrB_Vector u, int mop, int mcode, const void* identity, void* result_in_mcode, bool may_keep);      // may_keep: u may stay unmaterialised (the caller does not look at its buffers)

}  // namespace grb
