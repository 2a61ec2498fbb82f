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
//    137 model-level pressures) never touch HBM per point: the vector is staged
//    once per workgroup into LDS and indexed by level, with the level index
//    advanced incrementally (no per-point integer division).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/ekm_thermo.h"
#include "ops.hpp"

namespace ekm {

#ifndef EKM_THREADS
#define EKM_THREADS 256
#endif
constexpr int kThreads = EKM_THREADS;
#ifndef EKM_WAVES_PER_EU
#define EKM_WAVES_PER_EU 1
#endif

template <class T>
struct VecOf;
template <>
struct VecOf<float> {
  typedef float type __attribute__((ext_vector_type(4)));
  static constexpr int N = 4;
};
template <>
struct VecOf<double> {
  typedef double type __attribute__((ext_vector_type(2)));
  static constexpr int N = 2;
};

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
  int vec_ok;                    // all field pointers 16-B aligned
  const T* aux0;                 // EKM_HYBRID_FULL (last operand): A half-level table
  const T* aux1;                 //                                 B half-level table
};

#ifndef EKM_SCHED_SPLIT
#define EKM_SCHED_SPLIT 0
#endif
#ifndef EKM_NT_LOAD
#define EKM_NT_LOAD 1
#endif
#ifndef EKM_NT_STORE
#define EKM_NT_STORE 1
#endif

template <class T>
__device__ __forceinline__ typename VecOf<T>::type ld_stream(const T* p) {
  typedef typename VecOf<T>::type V;
#if EKM_NT_LOAD
  return __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
#else
  return *reinterpret_cast<const V*>(p);
#endif
}

template <class T>
__device__ __forceinline__ void st_stream(T* p, typename VecOf<T>::type v) {
  typedef typename VecOf<T>::type V;
#if EKM_NT_STORE
  __builtin_nontemporal_store(v, reinterpret_cast<V*>(p));
#else
  *reinterpret_cast<V*>(p) = v;
#endif
}

// ---- all operands are aligned full fields ----------------------------------
// Workgroup b owns `tiles` consecutive tiles of 256 vectors (256 x 16 B = 4 KiB per
// stream per tile) and exits: the dispatcher hands out workgroups in order, so at any
// moment the chip reads and writes one moving window of each stream.  Measured on
// MI355X this beats a persistent grid-stride loop by 10-25 % (profiles/).
template <class Op, class T, int UNROLL>
__global__ __launch_bounds__(kThreads, EKM_WAVES_PER_EU) void map_fields(const MapArgs<T, Op::NIN, Op::NOUT> a, unsigned tiles) {
  constexpr int NIN = Op::NIN, NOUT = Op::NOUT, V = VecOf<T>::N;
  typedef typename VecOf<T>::type Vec;
  const unsigned long long nvec = a.n / V;
  const unsigned long long base = (unsigned long long)blockIdx.x * tiles * kThreads + threadIdx.x;

  for (unsigned k = 0; k < tiles; k += UNROLL) {
    const unsigned long long v0 = base + (unsigned long long)k * kThreads;
    Vec xin[UNROLL][NIN];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const unsigned long long v = v0 + u * kThreads;
      if (v < nvec) {
#pragma unroll
        for (int i = 0; i < NIN; ++i) xin[u][i] = ld_stream<T>(a.in[i] + v * V);
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const unsigned long long v = v0 + u * kThreads;
      if (v < nvec) {
        Vec yout[NOUT];
#pragma unroll
        for (int j = 0; j < V; ++j) {
          T x[NIN], y[NOUT];
#pragma unroll
          for (int i = 0; i < NIN; ++i) x[i] = xin[u][i][j];
          Op::template apply<T>(x, y, a.rp);
#pragma unroll
          for (int o = 0; o < NOUT; ++o) yout[o][j] = y[o];
#if EKM_SCHED_SPLIT
          __builtin_amdgcn_sched_barrier(0);  // keep the per-point bodies apart: shorter live ranges
#endif
        }
#pragma unroll
        for (int o = 0; o < NOUT; ++o) st_stream<T>(a.out[o] + v * V, yout[o]);
      }
    }
  }
  // ragged tail: n % V single elements, done by the first lanes of workgroup 0
  const unsigned long long e = nvec * V + (unsigned long long)blockIdx.x * kThreads + threadIdx.x;
  if (e < a.n) {
    T x[NIN], y[NOUT];
#pragma unroll
    for (int i = 0; i < NIN; ++i) x[i] = a.in[i][e];
    Op::template apply<T>(x, y, a.rp);
#pragma unroll
    for (int o = 0; o < NOUT; ++o) a.out[o][e] = y[o];
  }
}

// ---- some operands are scalars / level vectors (or pointers are unaligned) ---
extern __shared__ __align__(16) unsigned char ekm_lds_raw[];

template <class Op, class T>
__global__ __launch_bounds__(kThreads) void map_bcast(const MapArgs<T, Op::NIN, Op::NOUT> a, unsigned tiles) {
  constexpr int NIN = Op::NIN, NOUT = Op::NOUT, V = VecOf<T>::N;
  typedef typename VecOf<T>::type Vec;
  T* lds = reinterpret_cast<T*>(ekm_lds_raw);

  // stage the shared vectors once per workgroup
  T sval[NIN];
#pragma unroll
  for (int i = 0; i < NIN; ++i) {
    sval[i] = T(0);
    if (a.mode[i] == EKM_SCALAR) sval[i] = a.in[i][0];
    if (a.mode[i] >= EKM_LEVEL_MAJOR)
      for (unsigned s = threadIdx.x; s < a.len[i]; s += kThreads) lds[a.lds_off[i] + s] = a.in[i][s];
  }
  __syncthreads();

  const unsigned long long nchunk = (a.n + V - 1) / V;  // last chunk may be partial
  const unsigned long long cblock = (unsigned long long)blockIdx.x * tiles * kThreads;  // wave-uniform

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
    const unsigned long long c = cblock + (unsigned long long)k * kThreads + threadIdx.x;
    if (c < nchunk) {
      const unsigned long long e0 = c * V;
      const bool full = (e0 + V <= a.n);
      Vec xin[NIN];
#pragma unroll
      for (int i = 0; i < NIN; ++i) {
        if (a.mode[i] == EKM_FIELD) {
          if (full && a.vec_ok) {
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
#pragma unroll
      for (int j = 0; j < V; ++j) {
        T x[NIN], y[NOUT];
#pragma unroll
        for (int i = 0; i < NIN; ++i) x[i] = xin[i][j];
        Op::template apply<T>(x, y, a.rp);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) yout[o][j] = y[o];
      }
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        if (full && a.vec_ok) {
          st_stream<T>(a.out[o] + e0, yout[o]);
        } else {
#pragma unroll
          for (int j = 0; j < V; ++j)
            if (e0 + j < a.n) a.out[o][e0 + j] = yout[o][j];
        }
      }
    }
    // advance the running positions by one tile (kThreads * V elements)
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

// ---- the usual broadcast shape: last operand (pressure) is a level vector or a scalar ----
// Fields are streamed exactly as in map_fields; the level vector sits in LDS and a lane's
// 16-B chunk almost always lies inside one level, so the pressure of the chunk is ONE
// value: every sub-expression that depends on pressure alone (log2(p/p0), (p0/p)^kappa,
// (p/p0)^kappa, 1/D(p), ...) is then computed once per chunk instead of once per point
// (the compiler hoists it out of the unrolled per-point bodies), and p costs no HBM traffic.
#ifndef EKM_LEVEL_LDS
#define EKM_LEVEL_LDS 1
#endif

template <class Op, class T>
__global__ __launch_bounds__(kThreads, EKM_WAVES_PER_EU) void map_plast(const MapArgs<T, Op::NIN, Op::NOUT> a,
                                                                      unsigned tiles) {
  constexpr int NIN = Op::NIN, NOUT = Op::NOUT, V = VecOf<T>::N, PI = Op::NIN - 1;
  typedef typename VecOf<T>::type Vec;
  const int mode = a.mode[PI];
  const bool lev = mode == EKM_LEVEL_MAJOR, hyb = mode == EKM_HYBRID_FULL;
  const unsigned long long inner = a.inner[PI];
  T* lds = reinterpret_cast<T*>(ekm_lds_raw);
#if EKM_LEVEL_LDS
  const T* tab = lds;
  if (lev) {
    for (unsigned s = threadIdx.x; s < a.len[PI]; s += kThreads) lds[s] = a.in[PI][s];
    __syncthreads();
  }
#else
  const T* tab = a.in[PI];
#endif
  const unsigned nhalf = a.len[PI] + 1;
  if (hyb) {  // A then B half-level tables in LDS
    for (unsigned s = threadIdx.x; s < nhalf; s += kThreads) {
      lds[s] = a.aux0[s];
      lds[nhalf + s] = a.aux1[s];
    }
    __syncthreads();
  }
  const T sval = (lev || hyb) ? T(0) : a.in[PI][0];
  const unsigned last = (lev || hyb) ? a.len[PI] - 1 : 0;
  // pressure of full level l at surface pressure s (vertical.py:670, 708)
  auto hybrid_p = [&](unsigned long long l, T s) {
    const T ph0 = lds[l] + lds[nhalf + l] * s;
    const T ph1 = lds[l + 1] + lds[nhalf + l + 1] * s;
    return ph0 + T(0.5) * (ph1 - ph0);
  };

  const unsigned long long nvec = a.n / V;
  const unsigned long long cblock = (unsigned long long)blockIdx.x * tiles * kThreads;  // wave-uniform
  unsigned long long q = 0, r = 0;
  if (lev || hyb) {  // one division per workgroup on the scalar unit, then carried forward per tile
    const unsigned long long e = cblock * V;
    q = e / inner;
    r = e % inner + (unsigned long long)threadIdx.x * V;
    while (r >= inner) {
      r -= inner;
      q += 1;
    }
  }

  for (unsigned k = 0; k < tiles; ++k) {
    const unsigned long long v = cblock + (unsigned long long)k * kThreads + threadIdx.x;
    if (v < nvec) {
      Vec xin[NIN], yout[NOUT];
#pragma unroll
      for (int i = 0; i < PI; ++i) xin[i] = ld_stream<T>(a.in[i] + v * V);
      if (hyb) {
        const T* sp = a.in[PI];
        if (r + V <= inner && (r % V) == 0 && a.vec_ok) {  // aligned chunk inside one level
          const Vec s = *reinterpret_cast<const Vec*>(sp + r);  // cached load: sp is re-read per level
#pragma unroll
          for (int j = 0; j < V; ++j) xin[PI][j] = hybrid_p(q, s[j]);
        } else {
#pragma unroll
          for (int j = 0; j < V; ++j) {
            unsigned long long l = q, rr = r + j;
            if (rr >= inner) {
              rr -= inner;
              l = l + 1 <= last ? l + 1 : last;
            }
            xin[PI][j] = hybrid_p(l, sp[rr]);
          }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
          T x[NIN], y[NOUT];
#pragma unroll
          for (int i = 0; i < NIN; ++i) x[i] = xin[i][j];
          Op::template apply<T>(x, y, a.rp);
#pragma unroll
          for (int o = 0; o < NOUT; ++o) yout[o][j] = y[o];
        }
      } else if (!lev || r + V <= inner) {  // the whole chunk has one pressure
        const T pv = lev ? tab[q] : sval;
#pragma unroll
        for (int j = 0; j < V; ++j) {
          T x[NIN], y[NOUT];
#pragma unroll
          for (int i = 0; i < PI; ++i) x[i] = xin[i][j];
          x[PI] = pv;
          Op::template apply<T>(x, y, a.rp);
#pragma unroll
          for (int o = 0; o < NOUT; ++o) yout[o][j] = y[o];
        }
      } else {  // the chunk straddles a level boundary (inner >= V: at most one)
        const T pv0 = tab[q], pv1 = tab[q + 1 <= last ? q + 1 : last];
#pragma unroll
        for (int j = 0; j < V; ++j) {
          T x[NIN], y[NOUT];
#pragma unroll
          for (int i = 0; i < PI; ++i) x[i] = xin[i][j];
          x[PI] = (r + j < inner) ? pv0 : pv1;
          Op::template apply<T>(x, y, a.rp);
#pragma unroll
          for (int o = 0; o < NOUT; ++o) yout[o][j] = y[o];
        }
      }
#pragma unroll
      for (int o = 0; o < NOUT; ++o) st_stream<T>(a.out[o] + v * V, yout[o]);
    }
    if (lev || hyb) {
      q += a.step_q[PI];
      r += a.step_r[PI];
      if (r >= inner) {
        r -= inner;
        q += 1;
      }
    }
  }
  // ragged tail: n % V single elements, done by the first lanes of workgroup 0
  const unsigned long long e = nvec * V + (unsigned long long)blockIdx.x * kThreads + threadIdx.x;
  if (e < a.n) {
    T x[NIN], y[NOUT];
#pragma unroll
    for (int i = 0; i < PI; ++i) x[i] = a.in[i][e];
    unsigned long long l = (lev || hyb) ? e / inner : 0;
    if (l > last) l = last;
    x[PI] = hyb ? hybrid_p(l, a.in[PI][e % inner]) : (lev ? tab[l] : sval);
    Op::template apply<T>(x, y, a.rp);
#pragma unroll
    for (int o = 0; o < NOUT; ++o) a.out[o][e] = y[o];
  }
}

// ---- host side ----------------------------------------------------------------
int set_error(int code, const char* fmt, ...);
int device_cus(int dev);            // CU count of device `dev` (cached), <0 on error
int use_device(int dev);            // hipSetDevice with error capture
int tuning_tiles_per_block();
int tuning_unroll();

constexpr unsigned kMaxLdsBytes = 64 * 1024;

template <class Op, class T>
int launch_map(int dev, void* stream, const ekm_operand* const* ins, void* const* outs, size_t n, double rp) {
  constexpr int NIN = Op::NIN, NOUT = Op::NOUT, V = VecOf<T>::N;
  if (n == 0) return EKM_OK;
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  MapArgs<T, NIN, NOUT> a;
  a.aux0 = a.aux1 = nullptr;
  a.n = n;
  a.rp = T(rp);
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
        if (reinterpret_cast<uintptr_t>(op->data) % 16) aligned = false;
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
        if (reinterpret_cast<uintptr_t>(op->data) % 16) aligned = false;
        a.len[i] = (unsigned)op->len;
        a.inner[i] = op->inner;
        a.aux0 = static_cast<const T*>(op->aux0);
        a.aux1 = static_cast<const T*>(op->aux1);
        lds_elems += 2 * ((a.len[i] + 1 + 3u) & ~3u);
        break;
      }
      default:
        return set_error(EKM_ERR_ARG, "operand %d: unknown mode %d", i, op->mode);
    }
  }
  for (int o = 0; o < NOUT; ++o) {
    if (!outs[o]) return set_error(EKM_ERR_ARG, "output %d: null pointer", o);
    a.out[o] = static_cast<T*>(outs[o]);
    if (reinterpret_cast<uintptr_t>(outs[o]) % 16) aligned = false;
  }
  a.vec_ok = aligned ? 1 : 0;
  if ((size_t)lds_elems * sizeof(T) > kMaxLdsBytes)
    return set_error(EKM_ERR_ARG, "level vectors need %zu B of LDS (max %u)", (size_t)lds_elems * sizeof(T),
                     kMaxLdsBytes);

  const unsigned long long nchunk = (n + V - 1) / V;
  const unsigned long long ntile = (nchunk + kThreads - 1) / kThreads;
  unsigned tiles = (unsigned)tuning_tiles_per_block();
  const int unroll = tuning_unroll();
  if (!bc && aligned && unroll >= 2) tiles = (tiles + 1u) & ~1u;  // the unrolled body takes tiles in pairs
  // keep the grid within the launch limit for very large fields
  while ((ntile + tiles - 1) / tiles > 0x7fffffffull) tiles *= 2;
  const unsigned grid = (unsigned)((ntile + tiles - 1) / tiles);
  hipStream_t s = static_cast<hipStream_t>(stream);

  bool plast = false;  // all fields aligned, only the last operand a level vector / scalar
  if constexpr (NIN >= 2) {
    plast = bc && aligned && (a.mode[NIN - 1] == EKM_LEVEL_MAJOR || a.mode[NIN - 1] == EKM_SCALAR ||
                              a.mode[NIN - 1] == EKM_HYBRID_FULL);
    for (int i = 0; i + 1 < NIN; ++i) plast = plast && a.mode[i] == EKM_FIELD;
  }
  if (plast) {
    const unsigned long long step = (unsigned long long)kThreads * V;
    if (a.mode[NIN - 1] == EKM_LEVEL_MAJOR || a.mode[NIN - 1] == EKM_HYBRID_FULL) {
      a.step_q[NIN - 1] = step / a.inner[NIN - 1];
      a.step_r[NIN - 1] = step % a.inner[NIN - 1];
    }
    hipLaunchKernelGGL((map_plast<Op, T>), dim3(grid), dim3(kThreads), lds_elems * sizeof(T), s, a, tiles);
  } else if (a.aux0) {
    return set_error(EKM_ERR_ARG, "EKM_HYBRID_FULL needs 16-B aligned full-field operands before it");
  } else if (!bc && aligned) {
    if (unroll >= 2)
      hipLaunchKernelGGL((map_fields<Op, T, 2>), dim3(grid), dim3(kThreads), 0, s, a, tiles);
    else
      hipLaunchKernelGGL((map_fields<Op, T, 1>), dim3(grid), dim3(kThreads), 0, s, a, tiles);
  } else {
    const unsigned long long step = (unsigned long long)kThreads * V;  // elements per tile
    for (int i = 0; i < NIN; ++i) {
      if (a.mode[i] == EKM_LEVEL_MAJOR) {
        a.step_q[i] = step / a.inner[i];
        a.step_r[i] = step % a.inner[i];
      } else if (a.mode[i] == EKM_LEVEL_MINOR) {
        a.step_r[i] = step % a.len[i];
      }
    }
    hipLaunchKernelGGL((map_bcast<Op, T>), dim3(grid), dim3(kThreads), lds_elems * sizeof(T), s, a, tiles);
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return set_error(EKM_ERR_HIP, "kernel launch: %s", hipGetErrorString(err));
  return EKM_OK;
}

}  // namespace ekm
