// Host twin of the kernels' per-point math -- TEST INFRASTRUCTURE ONLY.
//
// Compiles ops.hpp / thermo_math.hpp with g++ so the exact formulas the gfx950
// kernels run can be checked against the golden vectors in a container without
// a GPU.  The product (ekm_hip) never loads this library and has no CPU path.
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ops.hpp"

// Ops that keep a per-workgroup LDS table on the device (ops.hpp::OpTable, the bisection lattice) take the SAME path
// here -- table filled once, OpTable<Op>::apply per point -- so the table arithmetic the kernels run is what the golden
// vectors check.  EKM_TWIN_TABLE_FREE=1 selects Op::apply, the table-free statement of the same search;
// EKM_TWIN_BISECT_EXACT=1 makes every step of the tree walk take the reference's own residual (the kernels' tuning
// parameter bisect_exact), against which tests/test_hosttwin_fuzz.py compares the default walk bit for bit.
template <class Op, class T>
static int host_map(const T* const* ins, T* const* outs, size_t n, double rp) {
  std::vector<T> tab;
  bool all_exact = false;
  if constexpr (ekm::OpTable<Op>::elems > 0) {
    const char* ex = std::getenv("EKM_TWIN_BISECT_EXACT");
    all_exact = ex && ex[0] == '1';
    const char* env = std::getenv("EKM_TWIN_TABLE_FREE");
    if (!(env && env[0] == '1')) {
      tab.resize(ekm::OpTable<Op>::template count<T>());
      ekm::OpTable<Op>::template fill<T>(tab.data(), 0, 1);
    }
  }
  // fp64: the kernels' two passes (map_kernel.hpp::apply_points) -- fdouble first, whose primitives poison to NaN where
  // the plain ones apply an IEEE special-operand fix-up, then plain double for a point with a non-finite output.
  // EKM_TWIN_PLAIN_F64=1 skips the first pass.
  const char* plain = std::getenv("EKM_TWIN_PLAIN_F64");  // read per call: tests compare the two in one process
  const bool two_pass = !(plain && plain[0] == '1');
  (void)all_exact;
  for (size_t i = 0; i < n; ++i) {
    T x[Op::NIN], y[Op::NOUT];
    for (int k = 0; k < Op::NIN; ++k) x[k] = ins[k][i];
    if constexpr (sizeof(T) == 8) {
      if (two_pass) {
        ekm::fdouble xf[Op::NIN], yf[Op::NOUT];
        for (int k = 0; k < Op::NIN; ++k) xf[k] = ekm::fdouble(x[k]);
        if constexpr (ekm::OpTable<Op>::elems > 0) {
          if (!tab.empty())
            ekm::OpTable<Op>::template apply<ekm::fdouble>(xf, yf, ekm::fdouble(rp), reinterpret_cast<const ekm::fdouble*>(tab.data()), all_exact);
          else
            Op::template apply<ekm::fdouble>(xf, yf, ekm::fdouble(rp));
        } else {
          Op::template apply<ekm::fdouble>(xf, yf, ekm::fdouble(rp));
        }
        double yv[Op::NOUT];
        for (int k = 0; k < Op::NOUT; ++k) yv[k] = yf[k].v;
        if (!ekm::two_pass_redo_needed<Op>(x, yv)) {  // every output finite, or non-finite because an input it depends on is NaN
          for (int k = 0; k < Op::NOUT; ++k) outs[k][i] = yf[k].v;
          continue;
        }
      }
    }
    if constexpr (ekm::OpTable<Op>::elems > 0) {
      if (!tab.empty())
        ekm::OpTable<Op>::template apply<T>(x, y, T(rp), tab.data(), all_exact);
      else
        Op::template apply<T>(x, y, T(rp));
    } else {
      Op::template apply<T>(x, y, T(rp));
    }
    for (int k = 0; k < Op::NOUT; ++k) outs[k][i] = y[k];
  }
  return 0;
}

static int host_bad_enum(const char*, ...) { return -3; }

#include "gen/host_entries.inc"
