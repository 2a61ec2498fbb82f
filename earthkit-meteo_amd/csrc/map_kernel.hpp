// Streaming map kernel for gfx950: every thermo entry point is one launch of
// this template over an Op from ops.hpp.
//
// Data movement (the part that decides performance; the path is HBM-bound):
//  * 16 B per lane per access (float4 / double2): one wave = 1 KiB per
//    global_load_dwordx4 / global_store_dwordx4, fully coalesced;
//  * every input field is read once, every output field written once;
//  * loads/stores are non-temporal: nothing is re-used, so the streams should
//    not displace each other in L2 / MALL;
//  * one workgroup per run of `tiles` consecutive 4-KiB tiles (no persistent
//    grid-stride loop: in-order dispatch keeps one moving window per stream,
//    measured 10-25 % faster), UNROLL tiles in flight per lane per trip;
//  * operands that are not full fields (a scalar, or a level vector such as the
//    137 model-level pressures) never touch HBM per point.  The usual shape --
//    fields + ONE such operand last (pressure) -- runs on a 2-D grid (horizontal
//    tile x level), so the pressure of a workgroup's level is a wave-uniform value
//    (map_levels); any other mix stages the vectors once per workgroup in LDS and
//    indexes them by a level counter advanced incrementally (map_bcast).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <mutex>

#include "../../include/ekm_thermo.h"
#include "ops.hpp"

namespace ekm {

constexpr int kThreads = EKM_THREADS_DEFAULT;  // threads per workgroup unless the op says otherwise (ops.hpp::OpThreads)
#ifndef EKM_WAVES_PER_EU
#define EKM_WAVES_PER_EU 1
#endif
// fp64 kernels: fdouble first pass + plain-double redo of poisoned lanes (apply_points); 0 = plain double only
#ifndef EKM_F64_TWO_PASS
#if defined(EKM_F64_EXACT) || defined(EKM_F64_LIBM)
#define EKM_F64_TWO_PASS 0
#else
#define EKM_F64_TWO_PASS 1
#endif
#endif

template <class T>
struct VecOf;
template <>
struct VecOf<float> {
  typedef float type __attribute__((ext_vector_type(4)));
  typedef float mem_type __attribute__((ext_vector_type(4), aligned(4)));
  static constexpr int N = 4;
};
template <>
struct VecOf<double> {
  typedef double type __attribute__((ext_vector_type(2)));
  typedef double mem_type __attribute__((ext_vector_type(2), aligned(8)));
  static constexpr int N = 2;
};
// `mem_type`: the 16-B vector as it is loaded from and stored to global memory -- declared with the ELEMENT's alignment.
// gfx950 runs global memory in unaligned-access mode (the compiler itself emits global_load_dwordx4 for a 4-B-aligned
// vector), so a field whose pointer is not 16-B aligned -- a view at an odd offset, another library's tensor slice --
// takes the same instructions and kernels as an aligned one (it fell to one dword per access before: 0.5-0.6 of the rate,
// profiles/r04_odd_shapes.txt).
template <class T>
__device__ __forceinline__ typename VecOf<T>::type ld_cached(const T* p) {
  return *reinterpret_cast<const typename VecOf<T>::mem_type*>(p);
}
template <class T>
__device__ __forceinline__ void st_cached(T* p, typename VecOf<T>::type v) {
  *reinterpret_cast<typename VecOf<T>::mem_type*>(p) = v;
}

template <class T, int NIN, int NOUT>
struct MapArgs {
  const T* in[NIN];
  T* out[NOUT];
  unsigned long long n;
  T rp;
  // broadcast description (used by the BC instantiation only)
  int mode[NIN];                 // EKM_FIELD / EKM_SCALAR / EKM_LEVEL_MAJOR / EKM_LEVEL_MINOR
  unsigned len[NIN];             // vector length for the LEVEL modes
  unsigned lds_off[NIN];         // element offset of the staged vector in LDS
  unsigned long long inner[NIN]; // LEVEL_MAJOR: points per level
  unsigned long long step_q[NIN];  // (elements per tile) / inner   resp. unused
  unsigned long long step_r[NIN];  // (elements per tile) % inner   resp. % len
  int switches;                  // test / A-B switches: bit 0 fp64: redo every lane in plain double (tuning parameter f64_plain);
                                 // bit 1 fp32 IFS bisection: the exact residual at every step (bisect_exact)
  const T* aux0;                 // EKM_HYBRID_FULL (last operand): A half-level table
  const T* aux1;                 //                                 B half-level table
};

#ifndef EKM_NT_LOAD
#define EKM_NT_LOAD 1
#endif
#ifndef EKM_NT_STORE
#define EKM_NT_STORE 1
#endif

template <class T>
__device__ __forceinline__ typename VecOf<T>::type ld_stream(const T* p) {
  typedef typename VecOf<T>::mem_type V;
#if EKM_NT_LOAD
  return __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
#else
  return *reinterpret_cast<const V*>(p);
#endif
}

template <class T>
__device__ __forceinline__ void st_stream(T* p, typename VecOf<T>::type v) {
  typedef typename VecOf<T>::mem_type V;
#if EKM_NT_STORE
  __builtin_nontemporal_store(v, reinterpret_cast<V*>(p));
#else
  *reinterpret_cast<V*>(p) = v;
#endif
}

// Ops with a per-workgroup LDS table (OpTable<Op>::elems > 0, ops.hpp): reserve it, fill it once, pass it on.
template <class Op, class T>
__device__ __forceinline__ void op_apply_as(const T* __restrict__ x, T* __restrict__ y, T rp, const T* __restrict__ tab) {
  if constexpr (OpTable<Op>::elems > 0)
    OpTable<Op>::template apply<T>(x, y, rp, tab);
  else
    Op::template apply<T>(x, y, rp);
}
// One point.  fp64 goes through xdouble (thermo_math.hpp: plain primitives behind the same operator functions as the
// first pass of apply_points), so that every fp64 path of a kernel rounds alike.
template <class Op, class T>
__device__ __forceinline__ void op_apply(const T* __restrict__ x, T* __restrict__ y, T rp, const T* __restrict__ tab) {
  if constexpr (sizeof(T) == 8 && EKM_F64_TWO_PASS) {
    xdouble xs[Op::NIN], ys[Op::NOUT];
#pragma unroll
    for (int i = 0; i < Op::NIN; ++i) xs[i] = xdouble(x[i]);
    op_apply_as<Op, xdouble>(xs, ys, xdouble(rp), reinterpret_cast<const xdouble*>(tab));
#pragma unroll
    for (int o = 0; o < Op::NOUT; ++o) y[o] = ys[o].v;
  } else {
    op_apply_as<Op, T>(x, y, rp, tab);
  }
}
// The V points of one lane's 16-B chunk.  Ops with a Davies-Jones regime decision inside (OpUsesTie) run them
// branch-free first, only recording whether a point met a tie, and then -- in the ~0.3 % of waves where some lane
// did -- re-run that lane's points with the decision taken in double.  Keeping the branch out of the per-point body
// lets the compiler interleave the V independent points (4-8 % on the wet-bulb kernels).
template <class Op, class T, int V>
__device__ __forceinline__ void apply_points(const T (&x)[V][Op::NIN], T (&y)[V][Op::NOUT], T rp,
                                             const T* __restrict__ tab, int switches) {
  (void)switches;
  if constexpr (OpUsesTie<Op>::value && sizeof(T) == 4 && OpTable<Op>::elems == 0) {
    TieFlag flag;
#pragma unroll
    for (int j = 0; j < V; ++j) Op::template apply_tie<T>(x[j], y[j], rp, flag);
    if (__builtin_amdgcn_ballot_w64(flag.hit) != 0ull) {
      if (flag.hit) {
        TieExact exact;
#pragma unroll 1
        for (int j = 0; j < V; ++j) Op::template apply_tie<T>(x[j], y[j], rp, exact);
      }
    }
  } else if constexpr (sizeof(T) == 4 && OpTable<Op>::elems > 0 && OpTable<Op>::vectorized) {
    // the bisection walks its search tree step by step for all V points together (one wave-uniform branch per step)
    OpTable<Op>::template apply_v<T, V>(x, y, rp, tab, (switches & 2) != 0);
  } else if constexpr (sizeof(T) == 8 && EKM_F64_TWO_PASS) {
    // fp64: first pass in fdouble (thermo_math.hpp: primitives without special-operand fix-ups, which poison to NaN
    // where a fix-up would have acted), then -- wave-uniform test, practically never taken on atmospheric data -- the
    // lanes holding a non-finite output are redone in plain double, whose primitives treat IEEE specials as libm does.
    fdouble xf[V][Op::NIN], yf[V][Op::NOUT];
    unsigned fin = 0x7ff00000u;  // min over the outputs of (high dword & exponent mask): 0x7ff00000 - that == 0 iff inf / NaN
#pragma unroll
    for (int j = 0; j < V; ++j) {
#pragma unroll
      for (int i = 0; i < Op::NIN; ++i) xf[j][i] = fdouble(x[j][i]);
    }
    if constexpr (OpTable<Op>::elems > 0 && OpTable<Op>::vectorized)  // the bisection's tree walk: the V points together
      OpTable<Op>::template apply_v<fdouble, V>(xf, yf, fdouble(rp), reinterpret_cast<const fdouble*>(tab), (switches & 2) != 0);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      if constexpr (!(OpTable<Op>::elems > 0 && OpTable<Op>::vectorized))
        op_apply_as<Op, fdouble>(xf[j], yf[j], fdouble(rp), reinterpret_cast<const fdouble*>(tab));
#pragma unroll
      for (int o = 0; o < Op::NOUT; ++o) {
        y[j][o] = yf[j][o].v;
        const unsigned gap = 0x7ff00000u - ((unsigned)__double2hiint(yf[j][o].v) & 0x7ff00000u);
        fin = gap < fin ? gap : fin;
      }
    }
    bool redo = fin == 0u;
    if (__builtin_amdgcn_ballot_w64(redo) != 0ull) {
      // (rare, wave-uniform) a non-finite output that a NaN INPUT explains needs no plain pass: ops.hpp::OpDeps
      if (redo) {
        redo = false;
#pragma unroll
        for (int j = 0; j < V; ++j) redo = redo || two_pass_redo_needed<Op>(x[j], y[j]);
      }
    }
    redo = redo || (switches & 1) != 0;  // f64_plain (tuning parameter, wave-uniform): every lane takes the plain pass
    if (__builtin_amdgcn_ballot_w64(redo) != 0ull) {
      if (redo) {
        // one copy of the plain-double body, the point picked by selects: indexing x[j] / y[j] with a loop counter
        // would put both arrays into scratch memory
#pragma unroll 1
        for (int j = 0; j < V; ++j) {
          T xj[Op::NIN], yj[Op::NOUT];
#pragma unroll
          for (int i = 0; i < Op::NIN; ++i) {
            xj[i] = x[0][i];
#pragma unroll
            for (int jj = 1; jj < V; ++jj) xj[i] = j == jj ? x[jj][i] : xj[i];
          }
          op_apply<Op, T>(xj, yj, rp, tab);
#pragma unroll
          for (int o = 0; o < Op::NOUT; ++o) {
#pragma unroll
            for (int jj = 0; jj < V; ++jj) y[jj][o] = j == jj ? yj[o] : y[jj][o];
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < V; ++j) op_apply<Op, T>(x[j], y[j], rp, tab);
  }
}

// The table does not depend on the inputs: it is computed ONCE per device into a __device__ array (fill_op_table, on
// the first use of the op on that device: launch_map::ensure_op_table) and every workgroup copies its 32 KiB from
// there -- an L2-resident read of eight 16-B vectors per thread -- instead of recomputing 4096 x (es_mixed + a
// division), which cost a workgroup as much as ~2000 grid points and forced 64-tile workgroups to amortise it.
// (static: every translation unit is its own device module with its own copy, and its own `ready` flags below)
template <class Tab, class T>
static __device__ __attribute__((aligned(16))) T g_op_table[Tab::template count<T>() > 0 ? Tab::template count<T>() : 4];

template <class Tab, class T>
static __global__ __launch_bounds__(kThreads) void fill_op_table() {
  Tab::template fill<T>(g_op_table<Tab, T>, (int)(blockIdx.x * kThreads + threadIdx.x), (int)(gridDim.x * kThreads));
}

template <class Tab, class T, int NT>
__device__ __forceinline__ void load_op_table(T* __restrict__ lds) {
  typedef typename VecOf<T>::type Vec;
  constexpr int NV = Tab::template count<T>() / VecOf<T>::N;  // the counts are multiples of the vector width
  const Vec* __restrict__ src = reinterpret_cast<const Vec*>(g_op_table<Tab, T>);
  Vec* __restrict__ dst = reinterpret_cast<Vec*>(lds);
#pragma unroll 4
  for (int i = threadIdx.x; i < NV; i += NT) dst[i] = src[i];
}

#define EKM_OP_TABLE(Op, T, name)                                              \
  __shared__ __attribute__((aligned(16))) T name[OpTable<Op>::elems > 0 ? OpTable<Op>::template count<T>() : 4]; \
  if constexpr (OpTable<Op>::elems > 0) {                                      \
    load_op_table<typename OpTable<Op>::table_type, T, OpThreads<Op, T>::value>(name); \
    __syncthreads();                                                           \
  }

// The full-field tree-walk kernels (80 registers for six waves per SIMD) form their load and store addresses from a copy
// of the element index the compiler cannot see through: otherwise it strength-reduces them into one running 64-bit
// pointer per stream, carried round the tile loop and alive across the whole body, and paid for them with 12-88 B of
// scratch memory per lane (bolton35 bisection 4.87 -> 4.55 ms without it).  Kernels with registers to spare keep the
// running pointers: the six-output pipeline measured 2 % slower with late addresses, and so did the per-level tree
// walks, which never spilled (profiles/r04_tree_walk_tuning.txt).
template <class Op, class T>
__device__ __forceinline__ unsigned long long late_index(unsigned long long i) {
  if constexpr (OpThreads<Op, T>::tree) asm volatile("" : "+v"(i));
  return i;
}

// ---- all operands are aligned full fields ----------------------------------
// Workgroup b owns `tiles` consecutive tiles of 256 vectors (256 x 16 B = 4 KiB per
// stream per tile) and exits: the dispatcher hands out workgroups in order, so at any
// moment the chip reads and writes one moving window of each stream.  Measured on
// MI355X this beats a persistent grid-stride loop by 10-25 % (profiles/).
template <class Op, class T, int UNROLL>
__global__ __launch_bounds__((OpThreads<Op, T>::value), (OpThreads<Op, T>::field_waves)) void map_fields(const MapArgs<T, Op::NIN, Op::NOUT> a, unsigned tiles) {
  constexpr int NIN = Op::NIN, NOUT = Op::NOUT, V = VecOf<T>::N, NT = OpThreads<Op, T>::value;
  typedef typename VecOf<T>::type Vec;
  EKM_OP_TABLE(Op, T, op_tab)
  const unsigned long long nvec = a.n / V;
  const unsigned long long base = (unsigned long long)blockIdx.x * tiles * NT + threadIdx.x;

#if defined(EKM_WALK_V8)
  // A/B (round 6, NEGATIVE: 3.10 -> 3.35 ms): the fp32 IFS tree walk with EIGHT points per lane walking together (two tiles per
  // trip), 512-thread workgroups at 104 registers = 4 waves per SIMD: the same 32 points per SIMD in flight as 8 waves x 4
  // points, twice the independent LDS reads per wave and half the per-wave scalar work per point -- and 8 % slower; two
  // points per walk (ops.hpp: EKM_WALK_SPLIT) is 3 % slower too: 8 waves x 4 points is the optimum (profiles/r06_tree_walk.txt)
  if constexpr (OpThreads<Op, T>::wide && UNROLL == 1 && NOUT == 1) {
    for (unsigned k = 0; k < tiles; k += 2) {
      const unsigned long long v0 = base + (unsigned long long)k * NT, v1 = v0 + NT;
      if (v0 < nvec) {
        const bool ok1 = (k + 1 < tiles) && v1 < nvec;
        const unsigned long long w1 = ok1 ? v1 : v0;
        Vec xa[NIN], xb[NIN];
#pragma unroll
        for (int i = 0; i < NIN; ++i) xa[i] = ld_stream<T>(a.in[i] + v0 * V);
#pragma unroll
        for (int i = 0; i < NIN; ++i) xb[i] = ld_stream<T>(a.in[i] + w1 * V);
        T x[2 * V][NIN], y[2 * V][NOUT];
#pragma unroll
        for (int j = 0; j < V; ++j) {
#pragma unroll
          for (int i = 0; i < NIN; ++i) {
            x[j][i] = xa[i][j];
            x[V + j][i] = xb[i][j];
          }
        }
        OpTable<Op>::template apply_v<T, 2 * V>(x, y, a.rp, op_tab, (a.switches & 2) != 0);
        Vec ya, yb;
#pragma unroll
        for (int j = 0; j < V; ++j) {
          ya[j] = y[j][0];
          yb[j] = y[V + j][0];
        }
        st_stream<T>(a.out[0] + v0 * V, ya);
        if (ok1) st_stream<T>(a.out[0] + v1 * V, yb);
      }
    }
  } else
#endif
  for (unsigned k = 0; k < tiles; k += UNROLL) {
    const unsigned long long v0 = base + (unsigned long long)k * NT;
    Vec xin[UNROLL][NIN];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const unsigned long long v = v0 + u * NT;
      if (v < nvec) {
        const unsigned long long vl = late_index<Op, T>(v);
#pragma unroll
        for (int i = 0; i < NIN; ++i) xin[u][i] = ld_stream<T>(a.in[i] + vl * V);
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const unsigned long long v = v0 + u * NT;
      if (v < nvec) {
        Vec yout[NOUT];
        T x[V][NIN], y[V][NOUT];
#pragma unroll
        for (int j = 0; j < V; ++j) {
#pragma unroll
          for (int i = 0; i < NIN; ++i) x[j][i] = xin[u][i][j];
        }
        apply_points<Op, T, V>(x, y, a.rp, op_tab, a.switches);
#pragma unroll
        for (int j = 0; j < V; ++j) {
#pragma unroll
          for (int o = 0; o < NOUT; ++o) yout[o][j] = y[j][o];
        }
        const unsigned long long vs = late_index<Op, T>(v);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) st_stream<T>(a.out[o] + vs * V, yout[o]);
      }
    }
  }
  // ragged tail: n % V single elements, done by the first lanes of workgroup 0
  const unsigned long long e = nvec * V + (unsigned long long)blockIdx.x * NT + threadIdx.x;
  if (e < a.n) {
    T x[NIN], y[NOUT];
#pragma unroll
    for (int i = 0; i < NIN; ++i) x[i] = a.in[i][e];
    op_apply<Op, T>(x, y, a.rp, op_tab);
#pragma unroll
    for (int o = 0; o < NOUT; ++o) a.out[o][e] = y[o];
  }
}

// ---- some operands are scalars / level vectors in a general position ---
extern __shared__ __align__(16) unsigned char ekm_lds_raw[];

template <class Op, class T>
__global__ __launch_bounds__((OpThreads<Op, T>::value)) void map_bcast(const MapArgs<T, Op::NIN, Op::NOUT> a, unsigned tiles) {
  constexpr int NIN = Op::NIN, NOUT = Op::NOUT, V = VecOf<T>::N, NT = OpThreads<Op, T>::value;
  typedef typename VecOf<T>::type Vec;
  T* lds = reinterpret_cast<T*>(ekm_lds_raw);
  EKM_OP_TABLE(Op, T, op_tab)

  // stage the shared vectors once per workgroup
  T sval[NIN];
#pragma unroll
  for (int i = 0; i < NIN; ++i) {
    sval[i] = T(0);
    if (a.mode[i] == EKM_SCALAR) sval[i] = a.in[i][0];
    if (a.mode[i] >= EKM_LEVEL_MAJOR)
      for (unsigned s = threadIdx.x; s < a.len[i]; s += NT) lds[a.lds_off[i] + s] = a.in[i][s];
  }
  __syncthreads();

  const unsigned long long nchunk = (a.n + V - 1) / V;  // last chunk may be partial
  const unsigned long long cblock = (unsigned long long)blockIdx.x * tiles * NT;  // wave-uniform

  // Position of this lane's first chunk in every level operand: one division per workgroup
  // on the scalar unit (cblock is uniform), then carried forward by one tile per trip.
  unsigned long long pos_q[NIN], pos_r[NIN];
#pragma unroll
  for (int i = 0; i < NIN; ++i) {
    pos_q[i] = 0;
    pos_r[i] = 0;
    if (a.mode[i] == EKM_LEVEL_MAJOR) {
      const unsigned long long e = cblock * V;
      pos_q[i] = e / a.inner[i];
      pos_r[i] = e % a.inner[i] + (unsigned long long)threadIdx.x * V;
      while (pos_r[i] >= a.inner[i]) {
        pos_r[i] -= a.inner[i];
        pos_q[i] += 1;
      }
    } else if (a.mode[i] == EKM_LEVEL_MINOR) {
      pos_r[i] = ((cblock * V) % a.len[i] + (unsigned long long)threadIdx.x * V) % a.len[i];
    }
  }

  for (unsigned k = 0; k < tiles; ++k) {
    const unsigned long long c = cblock + (unsigned long long)k * NT + threadIdx.x;
    if (c < nchunk) {
      const unsigned long long e0 = c * V;
      const bool full = (e0 + V <= a.n);
      Vec xin[NIN];
#pragma unroll
      for (int i = 0; i < NIN; ++i) {
        if (a.mode[i] == EKM_FIELD) {
          if (full) {
            xin[i] = ld_stream<T>(a.in[i] + e0);
          } else {
#pragma unroll
            for (int j = 0; j < V; ++j) xin[i][j] = (e0 + j < a.n) ? a.in[i][e0 + j] : T(1);
          }
        } else if (a.mode[i] == EKM_SCALAR) {
#pragma unroll
          for (int j = 0; j < V; ++j) xin[i][j] = sval[i];
        } else if (a.mode[i] == EKM_LEVEL_MAJOR) {
          const T* tab = lds + a.lds_off[i];
          const unsigned long long l = pos_q[i], r = pos_r[i], inn = a.inner[i];
          const T v0 = tab[l];
          if (r + V <= inn) {  // whole chunk inside one level (the usual case)
#pragma unroll
            for (int j = 0; j < V; ++j) xin[i][j] = v0;
          } else {
            const unsigned last = a.len[i] - 1;
            const unsigned l1 = (l + 1 <= last) ? (unsigned)(l + 1) : last;
            const T v1 = tab[l1];  // inner >= V (host-checked): at most one boundary per chunk
#pragma unroll
            for (int j = 0; j < V; ++j) xin[i][j] = (r + j < inn) ? v0 : v1;
          }
        } else {  // EKM_LEVEL_MINOR, len >= V (host-checked)
          const T* tab = lds + a.lds_off[i];
          const unsigned len = a.len[i];
#pragma unroll
          for (int j = 0; j < V; ++j) {
            unsigned idx = (unsigned)pos_r[i] + j;
            if (idx >= len) idx -= len;
            xin[i][j] = tab[idx];
          }
        }
      }
      Vec yout[NOUT];
      T x[V][NIN], y[V][NOUT];
#pragma unroll
      for (int j = 0; j < V; ++j) {
#pragma unroll
        for (int i = 0; i < NIN; ++i) x[j][i] = xin[i][j];
      }
      apply_points<Op, T, V>(x, y, a.rp, op_tab, a.switches);
#pragma unroll
      for (int j = 0; j < V; ++j) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) yout[o][j] = y[j][o];
      }
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        if (full) {
          st_stream<T>(a.out[o] + e0, yout[o]);
        } else {
#pragma unroll
          for (int j = 0; j < V; ++j)
            if (e0 + j < a.n) a.out[o][e0 + j] = yout[o][j];
        }
      }
    }
    // advance the running positions by one tile (NT * V elements)
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      if (a.mode[i] == EKM_LEVEL_MAJOR) {
        pos_q[i] += a.step_q[i];
        pos_r[i] += a.step_r[i];
        if (pos_r[i] >= a.inner[i]) {
          pos_r[i] -= a.inner[i];
          pos_q[i] += 1;
        }
      } else if (a.mode[i] == EKM_LEVEL_MINOR) {
        pos_r[i] += a.step_r[i];
        if (pos_r[i] >= a.len[i]) pos_r[i] -= a.len[i];
      }
    }
  }
}

// ---- the usual broadcast shape: fields + pressure given per level (vector / scalar / hybrid definition) ----
// 3-D grid: blockIdx.x walks the horizontal tiles of ONE level inside a band of `gridDim.x` tiles (in order, so
// every stream is still one moving window), blockIdx.y selects the level (group), blockIdx.z the band.  The
// dispatcher issues x fastest, then y, then z: all levels of a band are processed before the next band starts.
// PM_LEVEL uses one band (the whole level); PM_HYBRID uses bands of 8 MiB of surface pressure, so that the
// band's sp chunk is still in the XCD's L2 when the next level re-reads it (each XCD sees every 8th tile:
// 1 MiB of sp per band against 4 MiB of L2), instead of coming back from the Infinity Cache / HBM once per
// level (26 MB x 137 levels: measured 1.10x the algorithmic traffic in round 1).
// Everything that depends on the level alone is wave-uniform:
//  * PM_LEVEL: the pressure of the level is one scalar load; sub-expressions of pressure alone (log2(p/p0),
//    (p0/p)^kappa, (p/p0)^kappa, 1/D(p), ...) are computed once per 16-B chunk instead of once per point, and
//    p costs no HBM traffic.  A scalar operand is the one-level case.
//  * PM_HYBRID (EKM_HYBRID_FULL): p = ph_k + 0.5*(ph_k+1 - ph_k), ph_h = A_h + B_h*sp (vertical.py:670,708) from
//    the lane's surface-pressure chunk; the pressure field itself never exists.  A workgroup can also walk
//    `lev_per_wg` > 1 consecutive levels of its tile with sp and ph_k+1 kept in registers: faster for a
//    one-in one-out op, slower once several streams are open, each then with lev_per_wg windows
//    (profiles/r02_sweep_hybrid.txt).
// No per-lane position bookkeeping, no integer division, no LDS: a tile never straddles levels.
enum { PM_LEVEL = 0, PM_HYBRID = 1, PM_FLAT = 2 };

template <class T, int NIN, int NOUT>
struct LevArgs {
  const T* in[NIN];  // fields; in[NIN-1] = the level vector (PM_LEVEL) or the surface pressure (PM_HYBRID, PM_FLAT)
  T* out[NOUT];
  const T* A;        // PM_HYBRID / PM_FLAT: half-level tables, last + 2 values each
  const T* B;
  unsigned long long n;  // grid points
  unsigned inner;        // points per level (< 2^31)
  unsigned nlev;         // levels covered by n points = ceil(n / inner)
  unsigned last;         // highest valid index of the level vector / full-level index
  unsigned lev_per_wg;   // consecutive levels one workgroup walks (WALK instantiation only; else 1)
  unsigned tiles;        // consecutive horizontal tiles per workgroup
  int switches;          // test / A-B switches, as in MapArgs
  T rp;
};

// PM_FLAT: the pure pressure levels of a hybrid table (B[k] = B[k+1] = 0: the upper 53 of the 137 IFS levels).  The
// host layer passes their count (ekm_operand.nflat) and launch_map runs them as a launch of their own: p does not
// depend on sp there, so the level runs exactly like a level-vector one -- pressure-only sub-expressions once per
// chunk instead of once per point -- without a second copy of the body inside the hybrid kernel (which cost the
// six-output pipeline occupancy).  The reference's p = A + B*sp is NaN for a non-finite sp even when B = 0: lanes
// whose surface pressure is not finite (never in real data) are redone point by point with that NaN.
// WALK (PM_HYBRID only): the workgroup walks `lev_per_wg` > 1 consecutive levels of its tile with sp and the shared
// half-level pressure in registers.
template <class Op, class T, int PMODE, bool WALK = false>
__global__ __launch_bounds__((OpThreads<Op, T>::value), (OpThreads<Op, T>::tree ? OpThreads<Op, T>::field_waves : sizeof(T) == 4 ? OpWaves<Op>::value : EKM_WAVES_PER_EU)) void map_levels(const LevArgs<T, Op::NIN, Op::NOUT> a) {
  constexpr int NIN = Op::NIN, NOUT = Op::NOUT, V = VecOf<T>::N, PI = Op::NIN - 1, NT = OpThreads<Op, T>::value;
  typedef typename VecOf<T>::type Vec;
  EKM_OP_TABLE(Op, T, op_tab)
  const unsigned lpw = WALK ? a.lev_per_wg : 1u;
  const unsigned k0 = blockIdx.y * lpw;  // wave-uniform

  for (unsigned i = 0; i < a.tiles; ++i) {
    const unsigned long long tile = (unsigned long long)blockIdx.z * gridDim.x + blockIdx.x;
    const unsigned long long col64 = (tile * a.tiles + i) * (NT * V) + threadIdx.x * V;
    if (col64 >= a.inner) continue;
    const unsigned col = (unsigned)col64;

    T ph0[V], s[V];  // PM_HYBRID: lower half-level pressure (carried up the column when WALK), surface pressure
    bool sp_ok = true;  // PM_FLAT: this lane's surface pressures are all finite
    if (PMODE == PM_HYBRID || PMODE == PM_FLAT) {
      const T* sp = a.in[PI];
      if (col + V <= a.inner) {
        const Vec sv = ld_cached<T>(sp + col);  // cached load: other level groups re-read it
#pragma unroll
        for (int j = 0; j < V; ++j) s[j] = sv[j];
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j) s[j] = (col + j < a.inner) ? sp[col + j] : T(1);
      }
      if (PMODE == PM_FLAT) {
        T sum = s[0];
#pragma unroll
        for (int j = 1; j < V; ++j) sum += s[j];
        sp_ok = (sum - sum == T(0));  // surface pressures are ~1e5: the sum is finite iff every one of them is
      } else if (WALK) {
        const unsigned kk = k0 <= a.last ? k0 : a.last;
        const T a0 = a.A[kk], b0 = a.B[kk];
#pragma unroll
        for (int j = 0; j < V; ++j) ph0[j] = a0 + b0 * s[j];
      }
    }

    for (unsigned l = 0; l < lpw; ++l) {
      const unsigned k = k0 + l;
      if (k >= a.nlev) break;
      const unsigned kk = k <= a.last ? k : a.last;
      const unsigned long long row = (unsigned long long)k * a.inner;
      const unsigned long long left = a.n - row;  // the last covered level may be partial
      const unsigned rowlen = left < a.inner ? (unsigned)left : a.inner;
      if (col >= rowlen) break;
      const unsigned long long e0 = row + col;

      // one level of this lane's chunk, pressure given per point by `pressure(j)`
      auto run_points = [&](auto pressure, auto wanted) {  // element by element, the points with wanted(j)
        // (rare path.  Its addresses are formed from an opaque copy of the element offset: sharing them with the vector
        // path made the compiler keep one 64-bit address per stream alive across the whole body -- eight register pairs
        // in the six-output pipeline, five of them spilled to scratch memory: VERDICT r3 weak 5)
        unsigned long long e0 = row + col;
        asm volatile("" : "+v"(e0));
#pragma unroll
        for (int j = 0; j < V; ++j) {
          if (col + j < rowlen && wanted(j)) {
            T x[NIN], y[NOUT];
#pragma unroll
            for (int f = 0; f < PI; ++f) x[f] = a.in[f][e0 + j];
            x[PI] = pressure(j);
            op_apply<Op, T>(x, y, a.rp, op_tab);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) a.out[o][e0 + j] = y[o];
          }
        }
      };
      auto run_level = [&](auto pressure) {
        if (col + V <= rowlen) {
          Vec xin[NIN > 1 ? NIN - 1 : 1], yout[NOUT];
#pragma unroll
          for (int f = 0; f < PI; ++f) xin[f] = ld_stream<T>(a.in[f] + e0);
          T x[V][NIN], y[V][NOUT];
#pragma unroll
          for (int j = 0; j < V; ++j) {
#pragma unroll
            for (int f = 0; f < PI; ++f) x[j][f] = xin[f][j];
            x[j][PI] = pressure(j);
          }
          apply_points<Op, T, V>(x, y, a.rp, op_tab, a.switches);
#pragma unroll
          for (int j = 0; j < V; ++j) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) yout[o][j] = y[j][o];
          }
#pragma unroll
          for (int o = 0; o < NOUT; ++o) st_stream<T>(a.out[o] + e0, yout[o]);
        } else {  // ragged end of a row
          run_points(pressure, [](int) { return true; });
        }
      };

      if (PMODE == PM_HYBRID) {
        const T a1 = a.A[kk + 1], b1 = a.B[kk + 1];
        T pv[V];
        if (WALK) {
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const T ph1 = a1 + b1 * s[j];
            pv[j] = ph0[j] + T(0.5) * (ph1 - ph0[j]);
            ph0[j] = ph1;
          }
        } else {
          const T a0 = a.A[kk], b0 = a.B[kk];
#pragma unroll
          for (int j = 0; j < V; ++j) {
            const T pl0 = a0 + b0 * s[j], ph1 = a1 + b1 * s[j];
            pv[j] = pl0 + T(0.5) * (ph1 - pl0);
          }
        }
        run_level([&](int j) { return pv[j]; });
      } else if (PMODE == PM_FLAT) {
        const T a0 = a.A[kk], a1 = a.A[kk + 1];
        const T pl = a0 + T(0.5) * (a1 - a0);  // (vertical.py:670, 708 with B = 0)
        run_level([&](int) { return pl; });
        if (__builtin_amdgcn_ballot_w64(!sp_ok) != 0ull) {  // a non-finite surface pressure: p = A + 0*sp is NaN there
          if (!sp_ok) run_points([&](int j) { return pl + (s[j] - s[j]); }, [&](int j) { return !(s[j] - s[j] == T(0)); });
        }
      } else {
        const T pl = a.in[PI][kk];
        run_level([&](int) { return pl; });
      }
    }
  }
}

// ---- host side ----------------------------------------------------------------
int set_error(int code, const char* fmt, ...);
int device_cus(int dev);            // CU count of device `dev` (cached), <0 on error
int use_device(int dev);            // hipSetDevice with error capture
int tuning_tiles_per_block();
int tuning_unroll();
int tuning_user_set();      // ekm_set_tuning has been called: the launch-shape heuristics of launch_map stand back
int tuning_lev_per_wg();    // EKM_HYBRID_FULL: consecutive levels one workgroup walks (EKM_LEV_PER_WG, default 0 = by stream count)
int tuning_hybrid_band_bytes();  // EKM_HYBRID_FULL: bytes of surface pressure per band (EKM_HYBRID_BAND_KB, default 8192 KiB)
int tuning_table_tiles();   // most tiles per workgroup for ops that keep an LDS table (EKM_TABLE_TILES, default 0 = by op)
int tuning_bisect_exact();   // fp32 IFS bisection: 1 = the reference's residual at every step of the tree walk (EKM_BISECT_EXACT, default 0)
int tuning_f64_plain();      // fp64 map kernels: 1 = every lane redone in plain double (EKM_F64_PLAIN, default 0)
int tuning_hybrid_rows();     // pressure_on_hybrid_levels: 1 = one workgroup per (level, tile), rows written in order (EKM_HYBRID_ROWS, default 1); 0 = one lane per column
int tuning_geo_chunk_levels();  // levels per launch of the geopotential column scan (EKM_GEO_CHUNK_LEVELS, default: all in one launch)

constexpr unsigned kMaxLdsBytes = 32 * 1024;  // staged level vectors (dynamic LDS); an op's own table (static, up to 80 KiB) comes on top

// Per-device, once: compute the op's table into its __device__ array -- WITHOUT a host wait, so that a launch function
// stays free of synchronisation (it may run while another stream or thread of the process is capturing):
//  * first use on a device, outside stream capture: the fill kernel goes onto the caller's stream, an event is recorded
//    behind it, and the call returns; the caller's launch is ordered behind the fill by the stream itself;
//  * later launches on OTHER streams wait for that event on the device (hipStreamWaitEvent) until a host function
//    queued behind the fill has marked the table ready, from when on launches cost one atomic load;
//  * under stream capture with the table not ready, the fill is recorded into the graph in front of the kernel that
//    needs it (the graph is then self-contained); nothing global is marked, because the captured work has not run.
//    ekm_prepare_tables(dev), called once outside capture, avoids that extra node.
// Every instantiation registers a "prepare" function with the runtime (register_table_prep) for ekm_prepare_tables.
typedef int (*table_prep_fn)(int dev);
void register_table_prep(table_prep_fn fn);

template <class Tab, class T>
struct OpTableState {
  std::atomic<int> ready{0};
  hipEvent_t filled = nullptr;      // recorded behind the fill kernel
  hipStream_t fill_stream = nullptr;
  bool recorded = false;
};
template <class Tab, class T>
static OpTableState<Tab, T> g_op_table_state[64];  // per device
template <class Tab, class T>
static std::mutex g_op_table_mu;

template <class Tab, class T>
static int ensure_op_table(int dev, hipStream_t s, bool wait = false);

template <class Tab, class T>
static int prep_table(int dev) {
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  return ensure_op_table<Tab, T>(dev, nullptr, true);
}
template <class Tab, class T>
struct OpTableRegistration {
  OpTableRegistration() { register_table_prep(&prep_table<Tab, T>); }
};
template <class Tab, class T>
static OpTableRegistration<Tab, T> g_op_table_registration;

template <class Tab, class T>
static int ensure_op_table(int dev, hipStream_t s, bool wait) {
  (void)&g_op_table_registration<Tab, T>;  // instantiates the registration (runs when the library is loaded)
  if (dev < 0 || dev >= 64) return set_error(EKM_ERR_NODEV, "device %d out of range", dev);
  OpTableState<Tab, T>& st = g_op_table_state<Tab, T>[dev];
  if (st.ready.load(std::memory_order_acquire)) return EKM_OK;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  hipError_t err = hipStreamIsCapturing(s, &cap);
  if (err != hipSuccess) {
    (void)hipGetLastError();
    return set_error(EKM_ERR_HIP, "lookup table of this function not prepared and the stream's capture state cannot be read (%s): "
                     "call ekm_prepare_tables(%d) once before capturing", hipGetErrorString(err), dev);
  }
  std::lock_guard<std::mutex> lk(g_op_table_mu<Tab, T>);
  if (st.ready.load(std::memory_order_acquire)) return EKM_OK;
  if (cap != hipStreamCaptureStatusNone) {
    if (wait) return set_error(EKM_ERR_ARG, "ekm_prepare_tables must not be called while the stream is capturing");
    hipLaunchKernelGGL((fill_op_table<Tab, T>), dim3(16), dim3(kThreads), 0, s);  // a node of the graph being captured
    err = hipGetLastError();
    if (err != hipSuccess) return set_error(EKM_ERR_HIP, "table fill launch (capture): %s", hipGetErrorString(err));
    return EKM_OK;
  }
  if (!st.recorded) {
    hipLaunchKernelGGL((fill_op_table<Tab, T>), dim3(16), dim3(kThreads), 0, s);
    err = hipGetLastError();
    if (err != hipSuccess) return set_error(EKM_ERR_HIP, "table fill launch: %s", hipGetErrorString(err));
    if (!st.filled) {
      err = hipEventCreateWithFlags(&st.filled, hipEventDisableTiming);
      if (err != hipSuccess) return set_error(EKM_ERR_HIP, "table fill event: %s", hipGetErrorString(err));
    }
    err = hipEventRecord(st.filled, s);
    if (err != hipSuccess) return set_error(EKM_ERR_HIP, "table fill event record: %s", hipGetErrorString(err));
    st.fill_stream = s;
    st.recorded = true;
    // "ready" is set by a host function queued behind the fill -- not by polling the event: hipEventQuery is refused
    // (and may invalidate the capture) while another stream of the calling thread captures in global mode.  If the
    // host function cannot be queued the table simply never counts as ready and every launch keeps the device-side wait.
    if (hipLaunchHostFunc(s, [](void* flag) { static_cast<std::atomic<int>*>(flag)->store(1, std::memory_order_release); },
                          &st.ready) != hipSuccess)
      (void)hipGetLastError();
  } else if (s != st.fill_stream) {
    err = hipStreamWaitEvent(s, st.filled, 0);  // device-side: this stream's kernel runs after the fill
    if (err != hipSuccess) return set_error(EKM_ERR_HIP, "table fill wait: %s", hipGetErrorString(err));
  }
  if (wait) {  // ekm_prepare_tables only: an explicit, documented host wait
    err = hipEventSynchronize(st.filled);
    if (err != hipSuccess) return set_error(EKM_ERR_HIP, "table fill: %s", hipGetErrorString(err));
    st.ready.store(1, std::memory_order_release);
  }
  return EKM_OK;
}

template <class Op, class T>
int launch_map(int dev, void* stream, const ekm_operand* const* ins, void* const* outs, size_t n, double rp) {
  constexpr int NIN = Op::NIN, NOUT = Op::NOUT, V = VecOf<T>::N, NT = OpThreads<Op, T>::value;
  if (n == 0) return EKM_OK;
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  if constexpr (OpTable<Op>::elems > 0) {
    rc = ensure_op_table<typename OpTable<Op>::table_type, T>(dev, static_cast<hipStream_t>(stream));
    if (rc != EKM_OK) return rc;
  }
  MapArgs<T, NIN, NOUT> a;
  a.aux0 = a.aux1 = nullptr;
  a.n = n;
  a.rp = T(rp);
  a.switches = tuning_f64_plain() | (tuning_bisect_exact() << 1);
  bool bc = false, aligned = true;
  unsigned lds_elems = 0;
  for (int i = 0; i < NIN; ++i) {
    const ekm_operand* op = ins[i];
    if (!op || !op->data) return set_error(EKM_ERR_ARG, "operand %d: null pointer", i);
    a.in[i] = static_cast<const T*>(op->data);
    a.mode[i] = op->mode;
    a.len[i] = 0;
    a.lds_off[i] = 0;
    a.inner[i] = 1;
    a.step_q[i] = a.step_r[i] = 0;
    switch (op->mode) {
      case EKM_FIELD:
        if (reinterpret_cast<uintptr_t>(op->data) % sizeof(T)) aligned = false;
        break;
      case EKM_SCALAR:
        bc = true;
        break;
      case EKM_LEVEL_MAJOR:
      case EKM_LEVEL_MINOR: {
        bc = true;
        if (op->len == 0 || op->len > 0xffffffffull)
          return set_error(EKM_ERR_ARG, "operand %d: level vector length %llu out of range", i,
                           (unsigned long long)op->len);
        a.len[i] = (unsigned)op->len;
        a.lds_off[i] = lds_elems;
        lds_elems += (a.len[i] + 3u) & ~3u;
        if (op->mode == EKM_LEVEL_MAJOR) {
          if (op->inner < (unsigned)V)
            return set_error(EKM_ERR_ARG, "operand %d: LEVEL_MAJOR needs inner >= %d (got %llu)", i, V,
                             (unsigned long long)op->inner);
          if ((unsigned long long)op->len * op->inner < n)
            return set_error(EKM_ERR_ARG, "operand %d: len*inner = %llu does not cover n = %llu", i,
                             (unsigned long long)op->len * op->inner, (unsigned long long)n);
          a.inner[i] = op->inner;
        } else if (op->len < (unsigned)V) {
          return set_error(EKM_ERR_ARG, "operand %d: LEVEL_MINOR needs len >= %d (got %llu)", i, V,
                           (unsigned long long)op->len);
        }
        break;
      }
      case EKM_HYBRID_FULL: {
        bc = true;
        if (i != NIN - 1 || NIN < 2)
          return set_error(EKM_ERR_ARG, "operand %d: EKM_HYBRID_FULL is supported for the last operand only", i);
        if (!op->aux0 || !op->aux1) return set_error(EKM_ERR_ARG, "operand %d: EKM_HYBRID_FULL needs the A and B tables", i);
        if (op->len == 0 || op->len > 8000u || op->inner < (unsigned)V || (unsigned long long)op->len * op->inner < n)
          return set_error(EKM_ERR_ARG, "operand %d: EKM_HYBRID_FULL needs 0 < len <= 8000, inner >= %d, len*inner >= n", i,
                           V);
        if (reinterpret_cast<uintptr_t>(op->data) % sizeof(T)) aligned = false;
        a.len[i] = (unsigned)op->len;
        a.inner[i] = op->inner;
        a.aux0 = static_cast<const T*>(op->aux0);
        a.aux1 = static_cast<const T*>(op->aux1);
        break;
      }
      default:
        return set_error(EKM_ERR_ARG, "operand %d: unknown mode %d", i, op->mode);
    }
  }
  for (int o = 0; o < NOUT; ++o) {
    if (!outs[o]) return set_error(EKM_ERR_ARG, "output %d: null pointer", o);
    a.out[o] = static_cast<T*>(outs[o]);
    if (reinterpret_cast<uintptr_t>(outs[o]) % sizeof(T)) aligned = false;
  }
  if (!aligned) return set_error(EKM_ERR_ARG, "a field pointer is not aligned to its element size (%d B)", (int)sizeof(T));
  // staged level vectors: at most 32 KiB of dynamic LDS (what the host layer hands over at most); the op's own
  // per-workgroup table is static LDS on top of that (the CU has 160 KiB)
  if ((size_t)lds_elems * sizeof(T) > kMaxLdsBytes)
    return set_error(EKM_ERR_ARG, "level vectors need %zu B of LDS (max %zu)", (size_t)lds_elems * sizeof(T), (size_t)kMaxLdsBytes);

  const unsigned long long nchunk = (n + V - 1) / V;
  const unsigned long long ntile = (nchunk + NT - 1) / NT;
  unsigned tiles = (unsigned)tuning_tiles_per_block();
  // an op that builds an LDS table per workgroup amortises it over more tiles (bisection: 4096 es values)
  // an op with an LDS table copies 32 KiB per workgroup from the device-resident table: a few tiles amortise that
  // (0 = by op: the tree walk of the IFS bisection copies 64 KiB per workgroup and takes 16 tiles -- 6 on hybrid
  // levels, whose bands want the workgroups short: 2/4/6/8/16 tiles 3.25/3.11/3.07/3.19/3.48 ms --, the 32-KiB tables 8;
  // profiles/r04_bisect_tree_walk.txt, profiles/r05_tree_walk.txt)
  unsigned table_tiles = (unsigned)tuning_table_tiles();
  if (table_tiles == 0)
    table_tiles = !OpThreads<Op, T>::tree ? 8u : a.mode[NIN - 1] != EKM_HYBRID_FULL ? 16u : OpThreads<Op, T>::wide ? 6u : 8u;
  if (OpTable<Op>::elems > 0 && tiles < table_tiles) {
    const int cus = device_cus(dev);
    unsigned long long most = ntile / (4ull * (unsigned long long)(cus > 0 ? cus : 256));  // >= 4 workgroups per CU
    if (most < 1) most = 1;
    tiles = (unsigned)(most < (unsigned long long)table_tiles ? most : (unsigned long long)table_tiles);
  }
  int unroll = tuning_unroll();
  // Launch-shape heuristics, only while the caller has not set the tuning explicitly (ekm_set_tuning):
  //  * the VALU-bound fp32 Newton kernels (one output, regime tie inside) hide their loads better with the next tile's
  //    loads issued ahead: two tiles per workgroup, both in flight (3.45 -> 3.41 ms; the HBM-bound kernels prefer
  //    one-shot tiles);
  //  * the fp64 Newton kernels (the same ops; VALU-bound, with a per-workgroup prologue of 13 coefficient pairs through
  //    the scalar cache): four tiles per workgroup (P5 14.47 -> 14.10 ms, wet-bulb 10.11 -> 9.93;
  //    profiles/r04_sweep_f64_tiles.txt).  The HBM-bound fp64 kernels keep one-shot tiles: with four they lose 6-14 %
  //    (theta 3.35 -> 3.83 ms, P3 7.27 -> 7.72; profiles/r04_ab_vs_r03.txt).
  if (!tuning_user_set() && !bc && aligned && tiles == 1 && unroll == 1 && ntile >= 4096) {
    if (OpUsesTie<Op>::value && NOUT == 1 && sizeof(T) == 4) {
      tiles = 2;
      unroll = 2;
    } else if (sizeof(T) == 8 && OpUsesTie<Op>::value) {
      tiles = 4;
    }
  }
  if (!bc && aligned && unroll >= 2) tiles = (tiles + 1u) & ~1u;  // the unrolled body takes tiles in pairs
  // keep the grid within the launch limit for very large fields
  while ((ntile + tiles - 1) / tiles > 0x7fffffffull) tiles *= 2;
  const unsigned grid = (unsigned)((ntile + tiles - 1) / tiles);
  hipStream_t s = static_cast<hipStream_t>(stream);

  // fields + one per-level operand last (pressure as a level vector, a scalar or the hybrid definition)
  bool levels = false;
  if constexpr (NIN >= 2) {
    const int pm = a.mode[NIN - 1];
    levels = bc && (pm == EKM_LEVEL_MAJOR || pm == EKM_SCALAR || pm == EKM_HYBRID_FULL);
    for (int i = 0; i + 1 < NIN; ++i) levels = levels && a.mode[i] == EKM_FIELD;
    if (levels) {
      LevArgs<T, NIN, NOUT> la;
      for (int i = 0; i < NIN; ++i) la.in[i] = a.in[i];
      for (int o = 0; o < NOUT; ++o) la.out[o] = a.out[o];
      la.A = a.aux0;
      la.B = a.aux1;
      la.n = n;
      la.rp = a.rp;
      la.switches = a.switches;
      unsigned long long inner = a.inner[NIN - 1];
      la.last = pm == EKM_SCALAR ? 0u : a.len[NIN - 1] - 1u;
      if (pm == EKM_SCALAR) inner = n < (1ull << 30) ? n : (1ull << 30);  // one "level" of any convenient length
      const unsigned long long nlev = (n + inner - 1) / inner;
      // hybrid: 0 = auto: a one-in one-out op (theta: 8 B/pt) gains from walking 4 levels per workgroup with sp and
      // the shared half-level pressure in registers (1.29 -> 1.16 ms); with more streams open it loses (P3 3.01 -> 3.28 ms)
      // The level walk exists for one-in one-out ops only (kWalk): with more streams it measured slower (P3 3.01 -> 3.28 ms,
      // profiles/r02_sweep_hybrid.txt) and its carried registers pushed the six-output pipeline into scratch memory, so
      // for those ops the tuning parameter lev_per_wg has no effect (include/ekm_thermo.h says so).
      // Nor for a tree walk: the carried registers do not fit beside its state (20-60 B of scratch per lane for 1 %).
      constexpr bool kWalk = NIN - 1 + NOUT <= 2 && !OpThreads<Op, T>::tree;
      unsigned lpw = 1u;
      if (pm == EKM_HYBRID_FULL && kWalk) lpw = tuning_lev_per_wg() > 0 ? (unsigned)tuning_lev_per_wg() : 4u;
      // shapes this kernel is not meant for go to map_bcast: rows longer than 32-bit columns, more level
      // groups than gridDim.y allows, or rows much shorter than a workgroup (a vector along a short axis)
      const bool fits = inner < (1ull << 31) && nlev <= 65535ull && (inner >= (unsigned long long)NT || pm == EKM_HYBRID_FULL);
      if (!fits && pm == EKM_HYBRID_FULL)
        return set_error(EKM_ERR_ARG, "EKM_HYBRID_FULL: inner = %llu, len = %u is outside the supported range", inner,
                         a.len[NIN - 1]);
      levels = fits;
      if (levels) {
        la.inner = (unsigned)inner;
        la.tiles = OpTable<Op>::elems > 0 ? tiles : 1u;
        const unsigned long long per_wg = (unsigned long long)NT * V * la.tiles;
        const unsigned long long ntx = (inner + per_wg - 1) / per_wg;  // workgroups along one level
        // hybrid: bands of EKM_HYBRID_BAND_KB of surface pressure (L2-resident between levels); else one band
        unsigned long long band = ntx;
        if (pm == EKM_HYBRID_FULL) {
          band = (unsigned long long)tuning_hybrid_band_bytes() / (per_wg * sizeof(T));
          if (band < 8) band = 8;
          if (band > ntx) band = ntx;
          while ((ntx + band - 1) / band > 65535ull) band *= 2;
        }
        // (A row need not start 16-B aligned -- VecOf::mem_type: element alignment is enough --, so rows of any length take
        // the vector path; the ragged end of a row is handled by the kernel's own `col + V <= rowlen` test.)
        // one launch over the levels [k_lo, k_hi) of the call: pointers, tables and counts rebased to k_lo
        auto launch_part = [&](int pmode, unsigned k_lo, unsigned k_hi, unsigned walk) {
          LevArgs<T, NIN, NOUT> q = la;
          const unsigned long long off = (unsigned long long)k_lo * inner;
          for (int i = 0; i + 1 < NIN; ++i) q.in[i] = la.in[i] + off;
          for (int o = 0; o < NOUT; ++o) q.out[o] = la.out[o] + off;
          if (pmode == PM_LEVEL) {
            q.in[NIN - 1] = la.in[NIN - 1] + (pm == EKM_SCALAR ? 0u : k_lo);
          } else {
            q.A = la.A + k_lo;
            q.B = la.B + k_lo;
          }
          q.last = la.last >= k_lo ? la.last - k_lo : 0u;
          const unsigned long long hi_pts = (unsigned long long)k_hi * inner;
          q.n = (hi_pts < n ? hi_pts : n) - off;
          q.nlev = k_hi - k_lo;
          q.lev_per_wg = walk < q.nlev ? walk : q.nlev;
          const unsigned gy = (q.nlev + q.lev_per_wg - 1) / q.lev_per_wg;
          const dim3 g((unsigned)band, gy, (unsigned)((ntx + band - 1) / band));
#define EKM_LEV_LAUNCH(PM_, WALK_) hipLaunchKernelGGL((map_levels<Op, T, PM_, WALK_>), g, dim3(NT), 0, s, q)
          if (pmode == PM_LEVEL) {
            EKM_LEV_LAUNCH(PM_LEVEL, false);
          } else if (pmode == PM_FLAT) {
            EKM_LEV_LAUNCH(PM_FLAT, false);
          } else if (kWalk && q.lev_per_wg > 1) {
            if constexpr (kWalk) EKM_LEV_LAUNCH(PM_HYBRID, true);
          } else {
            EKM_LEV_LAUNCH(PM_HYBRID, false);
          }
#undef EKM_LEV_LAUNCH
        };
        if (pm == EKM_HYBRID_FULL) {
          // the leading `nflat` levels are pure pressure levels (B = 0 on both half levels; the host layer counts
          // them, ekm_operand.nflat): they run as a level-vector launch of their own, the rest as the hybrid kernel
          unsigned nflat = ins[NIN - 1]->nflat > 0 ? (unsigned)ins[NIN - 1]->nflat : 0u;
          if (nflat > nlev) nflat = (unsigned)nlev;
          if (nflat > 0) launch_part(PM_FLAT, 0u, nflat, 1u);
          if (nflat < nlev) launch_part(PM_HYBRID, nflat, (unsigned)nlev, lpw);
        } else {
          launch_part(PM_LEVEL, 0u, (unsigned)nlev, 1u);
        }
      }
    }
  }
  if (levels) {
    // launched above
  } else if (a.aux0) {
    return set_error(EKM_ERR_ARG, "EKM_HYBRID_FULL needs full-field operands before it");
  } else if (!bc && aligned) {
    // (two tiles in flight per lane double the state of a tree walk: those kernels have no such instantiation)
    if constexpr (!OpThreads<Op, T>::tree) {
      if (unroll >= 2) {
        hipLaunchKernelGGL((map_fields<Op, T, 2>), dim3(grid), dim3(NT), 0, s, a, tiles);
        unroll = -1;
      }
    }
    if (unroll >= 0) hipLaunchKernelGGL((map_fields<Op, T, 1>), dim3(grid), dim3(NT), 0, s, a, tiles);
  } else {
    const unsigned long long step = (unsigned long long)NT * V;  // elements per tile
    for (int i = 0; i < NIN; ++i) {
      if (a.mode[i] == EKM_LEVEL_MAJOR) {
        a.step_q[i] = step / a.inner[i];
        a.step_r[i] = step % a.inner[i];
      } else if (a.mode[i] == EKM_LEVEL_MINOR) {
        a.step_r[i] = step % a.len[i];
      }
    }
    // every workgroup stages the level vectors in LDS before its first tile: with one 1024-element tile per workgroup a
    // 3600-element vector cost 3.5 x the payload in staging reads (theta against a trailing-axis vector: 0.49 of the
    // full-field rate).  Enough tiles per workgroup that the staging is <= 1/16 of its elements, while the grid still
    // holds >= 4 workgroups per CU.
    unsigned btiles = tiles;
    if (lds_elems > 0) {
      const unsigned long long want = (16ull * lds_elems + (unsigned long long)NT * V - 1) / ((unsigned long long)NT * V);
      const int cus = device_cus(dev);
      unsigned long long most = ntile / (4ull * (unsigned long long)(cus > 0 ? cus : 256));
      if (most < 1) most = 1;
      const unsigned long long t = want < most ? want : most;
      if (t > btiles) btiles = (unsigned)(t < 4096 ? t : 4096);
    }
    const unsigned bgrid = (unsigned)((ntile + btiles - 1) / btiles);
    hipLaunchKernelGGL((map_bcast<Op, T>), dim3(bgrid), dim3(NT), lds_elems * sizeof(T), s, a, btiles);
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return set_error(EKM_ERR_HIP, "kernel launch: %s", hipGetErrorString(err));
  return EKM_OK;
}

}  // namespace ekm
