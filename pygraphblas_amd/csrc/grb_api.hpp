// grb_api.hpp — helpers for writing extern "C" entry points.
#pragma once
#include "grb_internal.hpp"
#include <new>

extern "C" {
extern const uint64_t* GrB_ALL;
GrB_Info GrB_Vector_clear(GrB_Vector v);
GrB_Info GrB_Matrix_clear(GrB_Matrix A);
GrB_Info GrB_Matrix_new(GrB_Matrix* A, GrB_Type type, GrB_Index nrows, GrB_Index ncols);
GrB_Info GrB_Matrix_free(GrB_Matrix* A);
GrB_Info GrB_Vector_new(GrB_Vector* v, GrB_Type type, GrB_Index n);
GrB_Info GrB_Vector_free(GrB_Vector* v);
}

namespace grb {
extern thread_local std::string g_last_plan;
extern thread_local std::string g_last_error;   // most recent failure message of this thread (for *_error on another operand)  // human-readable list of kernels launched by the last hot-path call

// Run `body`; translate C++ failures to GrB_Info and remember the message on `obj` (if it has .err).
template <class Obj, class F> inline GrB_Info guarded(Obj* obj, F&& body) {
  try {
    body();
    return GrB_SUCCESS;
  } catch (const GrbError& e) {
    if (obj) obj->err = e.msg;
    g_last_error = e.msg;
    return e.info;
  } catch (const std::bad_alloc&) {
    if (obj) obj->err = "host allocation failed";
    return GrB_OUT_OF_MEMORY;
  } catch (const std::exception& e) {
    if (obj) obj->err = e.what();
    return GrB_PANIC;
  }
}

struct DescView {
  bool replace = false, mask_comp = false, mask_struct = false, tran0 = false, tran1 = false;
  int axb = 0;
  explicit DescView(GrB_Descriptor d) {
    if (!d) return;
    if (!check_obj(d)) fail(GrB_UNINITIALIZED_OBJECT, "descriptor is not initialised");
    replace = d->outp == GrB_REPLACE;
    mask_comp = (d->mask & GrB_COMP) != 0;
    mask_struct = (d->mask & GrB_STRUCTURE) != 0;
    tran0 = d->inp0 == GrB_TRAN; tran1 = d->inp1 == GrB_TRAN; axb = d->axb;
  }
};

inline void need_device() {
  if (!device_ok()) fail(GrB_PANIC, std::string("the MI355X HIP backend has no device: ") + device_error());
}

}  // namespace grb
