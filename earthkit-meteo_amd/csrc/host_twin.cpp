// Host twin of the kernels' per-point math -- TEST INFRASTRUCTURE ONLY.
//
// Compiles ops.hpp / thermo_math.hpp with g++ so the exact formulas the gfx950
// kernels run can be checked against the golden vectors in a container without
// a GPU.  The product (ekm_hip) never loads this library and has no CPU path.
#include <cstddef>
#include <cstdio>

#include "ops.hpp"

template <class Op, class T>
static int host_map(const T* const* ins, T* const* outs, size_t n, double rp) {
  for (size_t i = 0; i < n; ++i) {
    T x[Op::NIN], y[Op::NOUT];
    for (int k = 0; k < Op::NIN; ++k) x[k] = ins[k][i];
    Op::template apply<T>(x, y, T(rp));
    for (int k = 0; k < Op::NOUT; ++k) outs[k][i] = y[k];
  }
  return 0;
}

static int host_bad_enum(const char*, ...) { return -3; }

#include "gen/host_entries.inc"
