// pressure_on_hybrid_levels on gfx950: the producer of the model-level pressure the thermo
// kernels consume.  Reference: vertical/array/vertical.py:505-740
//   p_half[h] = A[h] + B[h]*sp                      (:670)
//   p_full[k] = p_half[k] + 0.5*(p_half[k+1]-p_half[k])   (:708)
//   delta[k]  = log(p_half[k+1]/p_half[k]),  alpha[k] = 1 - p_half[k]/(p_half[k+1]-p_half[k])*delta[k]
//   top layer: delta = log(p_half[1]/0.1), alpha = alpha_top when any p_half[0] <= 0.1   (:672-701)
//
// One lane owns 4 (fp32) / 2 (fp64) consecutive columns and walks down the levels with the
// previous half-level pressure in registers: sp is read from HBM once, every output element
// is written once with 16-B non-temporal stores (a workgroup writes 4 KiB contiguous per level
// and output), the A/B tables are wave-uniform scalar loads.  Store-bandwidth bound.
// delta/alpha use the accurate libm log and IEEE division: alpha cancels to ~1e-2 of its terms.
#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/ekm_thermo.h"
#include "map_kernel.hpp"

namespace ekm {

// delta = log(p_lo/p_hi) and alpha = 1 - p_hi/(p_lo - p_hi)*delta of one layer (vertical.py:679, 690).
// alpha cancels to ~1e-2 of its terms, so the logarithm must be relatively accurate for ratios near 1:
// fp32 uses log(r) = 2 atanh(s), s = (p_lo - p_hi)/(p_lo + p_hi) <= 0.25, as an odd series to s^15
// (<= 3e-8 relative, no cancellation, one v_rcp_f32 + 9 FMAs instead of libm log + two IEEE divisions);
// fp64 uses the device libm.
__device__ __forceinline__ void layer_delta_alpha(float p_hi, float p_lo, float& d, float& a) {
  const float dif = p_lo - p_hi;
  const float s = dif * __builtin_amdgcn_rcpf(p_lo + p_hi);
  const float z = s * s;
  float q = 1.0f / 15.0f;
  q = __builtin_fmaf(q, z, 1.0f / 13.0f);
  q = __builtin_fmaf(q, z, 1.0f / 11.0f);
  q = __builtin_fmaf(q, z, 1.0f / 9.0f);
  q = __builtin_fmaf(q, z, 1.0f / 7.0f);
  q = __builtin_fmaf(q, z, 1.0f / 5.0f);
  q = __builtin_fmaf(q, z, 1.0f / 3.0f);
  q = __builtin_fmaf(q, z, 1.0f);
  d = 2.0f * s * q;
  // thick layers, and a pressure that does not grow downwards or is not a positive number (a surface pressure of zero,
  // a negative one, inf, NaN: s <= -0.25, |s| > 1 for opposite signs, NaN): libm, as the reference has it
  if (!(__builtin_fabsf(s) < 0.25f)) d = log(p_lo / p_hi);
  a = 1.0f - p_hi * __builtin_amdgcn_rcpf(dif) * d;
}

__device__ __forceinline__ void layer_delta_alpha(double p_hi, double p_lo, double& d, double& a) {
  d = log(p_lo / p_hi);
  a = 1.0 - p_hi / (p_lo - p_hi) * d;
}

template <class T>
__global__ __launch_bounds__(kThreads) void hybrid_levels(const T* __restrict__ A, const T* __restrict__ B,
                                                         const T* __restrict__ sp, unsigned long long npts,
                                                         unsigned nfull, const int* __restrict__ row_full,
                                                         const int* __restrict__ row_half, int top_is_zero,
                                                         T alpha_top, T* __restrict__ full, T* __restrict__ half,
                                                         T* __restrict__ delta, T* __restrict__ alpha, int vec_ok) {
  constexpr int V = VecOf<T>::N;
  typedef typename VecOf<T>::type Vec;
  const unsigned long long i0 = ((unsigned long long)blockIdx.x * kThreads + threadIdx.x) * V;
  if (i0 >= npts) return;
  const bool whole = vec_ok && (i0 + V <= npts);
  Vec s;
  if (whole) {
    s = ld_cached<T>(sp + i0);
  } else {
#pragma unroll
    for (int j = 0; j < V; ++j) s[j] = (i0 + j < npts) ? sp[i0 + j] : T(1);
  }
  auto put = [&](T* base, int row, const Vec& v) {
    T* dst = base + (unsigned long long)row * npts + i0;
    if (whole) {
      st_stream<T>(dst, v);
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j)
        if (i0 + j < npts) dst[j] = v[j];
    }
  };

  Vec ph = A[0] + B[0] * s;
  if (half) {
    const int r = row_half ? row_half[0] : 0;
    if (r >= 0) put(half, r, ph);
  }
  for (unsigned k = 0; k < nfull; ++k) {
    const Vec phn = A[k + 1] + B[k + 1] * s;
    const int rf = row_full ? row_full[k] : (int)k;
    if (rf >= 0) {
      if (full) put(full, rf, ph + T(0.5) * (phn - ph));
      if (delta || alpha) {
        Vec d, a;
#pragma unroll
        for (int j = 0; j < V; ++j) {
          if (k == 0 && top_is_zero) {
            d[j] = log(phn[j] / T(0.1));
            a[j] = alpha_top;
          } else {
            T dj, aj;
            layer_delta_alpha(ph[j], phn[j], dj, aj);
            d[j] = dj;
            a[j] = aj;
          }
        }
        if (delta) put(delta, rf, d);
        if (alpha) put(alpha, rf, a);
      }
    }
    if (half) {
      const int r = row_half ? row_half[k + 1] : (int)(k + 1);
      if (r >= 0) put(half, r, phn);
    }
    ph = phn;
  }
}

// The same outputs with one workgroup per (level, tile): every output row is written by workgroups dispatched in order,
// like a map kernel's stream, instead of each workgroup visiting all rows of its column strip 26 MB apart -- the walk that
// cost the column kernel its address translations (58 x the UTCL1 misses of a map kernel, the level-2 translation cache
// busy 80 % of the time: profiles/r04_columns_pmc.txt).  Nothing is carried from level to level here: both half-level
// pressures of the layer are formed from sp (one fma more per element), and sp is re-read once per level from L2 --
// the grid runs bands of EKM_HYBRID_BAND_KB of surface pressure, all levels of a band before the next band, as the
// thermo kernels do in EKM_HYBRID_FULL mode.  blockIdx.y = nfull is the extra row of `half`.
// Same expressions as hybrid_levels, element for element: the two kernels agree bit for bit (tests/test_gpu_vertical.py).
template <class T>
__global__ __launch_bounds__(kThreads) void hybrid_rows(const T* __restrict__ A, const T* __restrict__ B,
                                                       const T* __restrict__ sp, unsigned long long npts, unsigned nfull,
                                                       const int* __restrict__ row_full, const int* __restrict__ row_half,
                                                       int top_is_zero, T alpha_top, T* __restrict__ full,
                                                       T* __restrict__ half, T* __restrict__ delta, T* __restrict__ alpha,
                                                       int vec_ok) {
  constexpr int V = VecOf<T>::N;
  typedef typename VecOf<T>::type Vec;
  const unsigned k = blockIdx.y;  // wave-uniform
  const int rf = k < nfull ? (row_full ? row_full[k] : (int)k) : -1;
  const int rh = half ? (row_half ? row_half[k] : (int)k) : -1;
  if (rf < 0 && rh < 0) return;  // a level the caller did not select
  const unsigned long long tile = (unsigned long long)blockIdx.z * gridDim.x + blockIdx.x;
  const unsigned long long i0 = (tile * kThreads + threadIdx.x) * V;
  if (i0 >= npts) return;
  const bool whole = vec_ok && (i0 + V <= npts);
  Vec s;
  if (whole) {
    s = ld_cached<T>(sp + i0);  // cached: the other levels of the band re-read it
  } else {
#pragma unroll
    for (int j = 0; j < V; ++j) s[j] = (i0 + j < npts) ? sp[i0 + j] : T(1);
  }
  auto put = [&](T* base, int row, const Vec& v) {
    T* dst = base + (unsigned long long)row * npts + i0;
    if (whole) {
      st_stream<T>(dst, v);
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j)
        if (i0 + j < npts) dst[j] = v[j];
    }
  };
  const Vec ph = A[k] + B[k] * s;
  if (rh >= 0) put(half, rh, ph);
  if (rf < 0) return;
  const Vec phn = A[k + 1] + B[k + 1] * s;
  if (full) put(full, rf, ph + T(0.5) * (phn - ph));
  if (delta || alpha) {
    Vec d, a;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      if (k == 0 && top_is_zero) {
        d[j] = log(phn[j] / T(0.1));
        a[j] = alpha_top;
      } else {
        T dj, aj;
        layer_delta_alpha(ph[j], phn[j], dj, aj);
        d[j] = dj;
        a[j] = aj;
      }
    }
    if (delta) put(delta, rf, d);
    if (alpha) put(alpha, rf, a);
  }
}

// any(a0 + b0*sp <= thresh): the reference's global test for a zero-pressure model top
template <class T>
__global__ __launch_bounds__(kThreads) void any_le(const T* __restrict__ sp, unsigned long long n, T a0, T b0, T thresh,
                                                  int* flag) {
  const unsigned long long stride = (unsigned long long)gridDim.x * kThreads;
  bool hit = false;
  for (unsigned long long i = (unsigned long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride)
    hit |= (a0 + b0 * sp[i]) <= thresh;
  if (__builtin_amdgcn_ballot_w64(hit) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// Geopotential chain on hybrid levels, reference vertical.py:741-1190:
//   d_k = (Rd + (Rv-Rd) q_k) t_k;  dphi_k = sum_{j>k} d_j delta_j + d_k alpha_k   (bottom-up scan)
// fused with the producer of alpha/delta (above) so that neither they nor the pressures ever
// touch HBM: per grid point 8 B read (t, q) + 4 B written.  One lane owns 4 (2) consecutive
// columns and walks from the surface to the model top with the running sum and the lower
// half-level pressure in registers; a workgroup reads/writes 4 KiB contiguous per level and field.
// mode: 0 thickness, 1 + zs (geopotential), 2 geometric height above sea, 3 geopotential height
// above sea, 4 geometric height above ground, 5 geopotential height above ground.
enum { GEO_THICKNESS = 0, GEO_GEOPOTENTIAL = 1, GEO_H_GEOM_SEA = 2, GEO_H_GP_SEA = 3, GEO_H_GEOM_GROUND = 4,
       GEO_H_GP_GROUND = 5 };

// VEC: every pointer is 16-B aligned and npts is a multiple of the vector width, so each active lane moves whole
// 16-B chunks and the level loop is straight-line code (no per-lane branch around any load or store).
#ifndef EKM_GEO_THREADS
#define EKM_GEO_THREADS 256
#endif
constexpr int kGeoThreads = EKM_GEO_THREADS;

// AD: alpha and delta are given as fields (outputs of pressure_on_hybrid_levels the caller already holds,
// vertical.py:815-893) and streamed like t and q instead of being formed from A, B, sp (which are then unused).
template <class T, bool VEC, bool AD>
__global__ __launch_bounds__(kGeoThreads) void geopotential_columns(const T* __restrict__ A, const T* __restrict__ B,
                                                                const T* __restrict__ sp, const T* __restrict__ zs,
                                                                const T* __restrict__ t, const T* __restrict__ q,
                                                                const T* __restrict__ alpha_in,
                                                                const T* __restrict__ delta_in,
                                                                unsigned long long npts, unsigned nfull,
                                                                unsigned k_lo, unsigned k_hi, int top_is_zero,
                                                                T alpha_top, int mode, T* __restrict__ out) {
  constexpr int V = VecOf<T>::N;
  typedef typename VecOf<T>::type Vec;
  const unsigned long long i0 = ((unsigned long long)blockIdx.x * kGeoThreads + threadIdx.x) * V;
  if (i0 >= npts) return;
  auto get = [&](const T* base, unsigned row) {
    const T* src = base + (unsigned long long)row * npts + i0;
    Vec v;
    if (VEC) {
      v = ld_stream<T>(src);
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = (i0 + j < npts) ? src[j] : T(1);
    }
    return v;
  };
  Vec s;
#pragma unroll
  for (int j = 0; j < V; ++j) s[j] = T(1);
  if (!AD) s = get(sp, 0);
  Vec z0, hs;
#pragma unroll
  for (int j = 0; j < V; ++j) z0[j] = hs[j] = T(0);
  if (mode != GEO_THICKNESS && mode != GEO_H_GP_GROUND) z0 = get(zs, 0);
  constexpr double g0 = 9.80665, re = 6371229.0;  // constants/constants.py
  if (mode == GEO_H_GEOM_GROUND) {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const T zz = z0[j] / T(g0);
      hs[j] = T(re) * zz / (T(re) - zz);  // vertical.py:500-501
    }
  }

  // This launch walks the levels [k_lo, k_hi) bottom-up.  The column can be cut into chunks of `geo_chunk_levels`
  // levels, one launch each from the surface upward (fewer of the 3 x 137 row streams open at once).  Measured in
  // one process (profiles/r02_sweep_geopotential.txt): no gain -- 1.80 ms whole, 1.95 ms in chunks of 35 -- and
  // neither from a register ring prefetching the next 2-5 levels nor from 128/512/1024-thread workgroups; what moves
  // this kernel is where its 3.55-GB fields land in physical memory (1.80 vs 2.05-2.15 ms between processes on the
  // same device).  So the default is one launch; the chunked path stays (and is tested) for very tall columns.
  // The only state that crosses a chunk boundary is the running sum `acc` (the half-level pressure is recomputed
  // from A, B, sp): it travels in the output row just above the chunk (row k_lo - 1), which the next launch reads
  // before it overwrites it -- same lane, program order -- so no workspace is needed and nothing is allocated.
  Vec acc;
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = T(0);
  if (k_hi < nfull) {  // not the bottom chunk: pick up the running sum left in our first output row
    const T* src = out + (unsigned long long)(k_hi - 1) * npts + i0;
    if (VEC) {
      acc = ld_cached<T>(src);
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] = (i0 + j < npts) ? src[j] : T(0);
    }
  }
  Vec phn = s, ph = s;
  if (!AD) phn = A[k_hi] + B[k_hi] * s;  // lower half level of the current layer
  for (unsigned kk = k_hi; kk-- > k_lo;) {
    if (!AD) ph = A[kk] + B[kk] * s;
    const Vec tk = get(t, kk), qk = get(q, kk);
    Vec al = s, de = s;
    if (AD) {
      al = get(alpha_in, kk);
      de = get(delta_in, kk);
    }
    Vec o;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      T d, a;
      if (AD) {
        d = de[j];
        a = al[j];
      } else if (kk == 0 && top_is_zero) {
        d = log(phn[j] / T(0.1));
        a = alpha_top;
      } else {
        layer_delta_alpha(ph[j], phn[j], d, a);
      }
      const T rt = (T(k::Rd) + T(k::Rv - k::Rd) * qk[j]) * tk[j];  // thermo.py:1706, vertical.py:800-801
      const T dphi = acc[j] + rt * a;
      acc[j] += rt * d;
      T r = dphi;
      if (mode == GEO_GEOPOTENTIAL) r = dphi + z0[j];
      if (mode == GEO_H_GEOM_SEA || mode == GEO_H_GEOM_GROUND) {
        const T zz = (dphi + z0[j]) / T(g0);
        r = T(re) * zz / (T(re) - zz) - hs[j];
      }
      if (mode == GEO_H_GP_SEA) r = (dphi + z0[j]) / T(g0);
      if (mode == GEO_H_GP_GROUND) r = dphi / T(g0);
      o[j] = r;
    }
    T* dst = out + (unsigned long long)kk * npts + i0;
    if (VEC) {
      st_stream<T>(dst, o);
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j)
        if (i0 + j < npts) dst[j] = o[j];
    }
    phn = ph;
  }
  if (k_lo > 0) {  // hand the running sum to the chunk above
    T* dst = out + (unsigned long long)(k_lo - 1) * npts + i0;
    if (VEC) {
      st_cached<T>(dst, acc);
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j)
        if (i0 + j < npts) dst[j] = acc[j];
    }
  }
}

template <class T>
static int launch_geopotential(int dev, void* stream, const T* A, const T* B, const T* sp, const T* zs, const T* t,
                               const T* q, size_t npts, uint32_t nfull, int top_is_zero, T alpha_top, int mode,
                               T* out, const T* alpha_in = nullptr, const T* delta_in = nullptr) {
  if (npts == 0 || nfull == 0) return EKM_OK;
  const bool ad = alpha_in || delta_in;
  if (ad && (!alpha_in || !delta_in || !t || !q || !out))
    return set_error(EKM_ERR_ARG, "geopotential_thickness_from_alpha_delta: null pointer");
  if (!ad && (!A || !B || !sp || !t || !q || !out))
    return set_error(EKM_ERR_ARG, "geopotential_on_hybrid_levels: null pointer");
  if (mode < 0 || mode > 5) return set_error(EKM_ERR_ENUM, "geopotential_on_hybrid_levels: mode=%d", mode);
  if (mode != GEO_THICKNESS && mode != GEO_H_GP_GROUND && !zs)
    return set_error(EKM_ERR_ARG, "geopotential_on_hybrid_levels: this mode needs the surface geopotential zs");
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  constexpr int V = VecOf<T>::N;
  int vec_ok = (npts % V == 0);
  for (const void* ptr : {(const void*)sp, (const void*)zs, (const void*)t, (const void*)q, (const void*)out,
                          (const void*)alpha_in, (const void*)delta_in})
    if (ptr && reinterpret_cast<uintptr_t>(ptr) % sizeof(T)) vec_ok = 0;
  const unsigned long long nchunk = (npts + V - 1) / V;
  const unsigned long long grid = (nchunk + kGeoThreads - 1) / kGeoThreads;
  if (grid > 0x7fffffffull) return set_error(EKM_ERR_ARG, "geopotential_on_hybrid_levels: too many columns");
  // chunks of at most `geo_chunk_levels` levels, balanced, launched from the surface upward on the same stream
  const unsigned maxc = (unsigned)tuning_geo_chunk_levels();
  const unsigned nchunks = (nfull + maxc - 1) / maxc;
  for (unsigned c = 0; c < nchunks; ++c) {
    const unsigned k_hi = nfull - (unsigned)((unsigned long long)nfull * c / nchunks);
    const unsigned k_lo = nfull - (unsigned)((unsigned long long)nfull * (c + 1) / nchunks);
    const dim3 g((unsigned)grid), b(kGeoThreads);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned long long np_ = npts;
#define EKM_GEO_LAUNCH(VEC_, AD_)                                                                                   \
  hipLaunchKernelGGL((geopotential_columns<T, VEC_, AD_>), g, b, 0, st, A, B, sp, zs, t, q, alpha_in, delta_in, np_, \
                     nfull, k_lo, k_hi, top_is_zero, alpha_top, mode, out)
    if (ad) {
      if (vec_ok) EKM_GEO_LAUNCH(true, true); else EKM_GEO_LAUNCH(false, true);
    } else {
      if (vec_ok) EKM_GEO_LAUNCH(true, false); else EKM_GEO_LAUNCH(false, false);
    }
#undef EKM_GEO_LAUNCH
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(EKM_ERR_HIP, "geopotential_columns launch: %s", hipGetErrorString(e));
  return EKM_OK;
}

template <class T>
static int launch_hybrid(int dev, void* stream, const T* A, const T* B, const T* sp, size_t npts, uint32_t nfull,
                         const int32_t* row_full, const int32_t* row_half, int top_is_zero, T alpha_top, T* full,
                         T* half, T* delta, T* alpha) {
  if (npts == 0 || nfull == 0) return EKM_OK;
  if (!A || !B || !sp) return set_error(EKM_ERR_ARG, "pressure_on_hybrid_levels: null A/B/sp");
  if (!full && !half && !delta && !alpha) return set_error(EKM_ERR_ARG, "pressure_on_hybrid_levels: no output");
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  constexpr int V = VecOf<T>::N;
  int vec_ok = (npts % V == 0) && reinterpret_cast<uintptr_t>(sp) % sizeof(T) == 0;
  for (T* o : {full, half, delta, alpha})
    if (o && reinterpret_cast<uintptr_t>(o) % sizeof(T)) vec_ok = 0;
  const unsigned long long nchunk = (npts + V - 1) / V;
  const unsigned long long grid = (nchunk + kThreads - 1) / kThreads;
  if (grid > 0x7fffffffull) return set_error(EKM_ERR_ARG, "pressure_on_hybrid_levels: too many columns");
  const unsigned rows = nfull + (half ? 1u : 0u);
  if (tuning_hybrid_rows() && rows <= 65535u) {
    unsigned long long band = (unsigned long long)tuning_hybrid_band_bytes() / ((unsigned long long)kThreads * V * sizeof(T));
    if (band < 8) band = 8;
    if (band > grid) band = grid;
    while ((grid + band - 1) / band > 65535ull) band *= 2;
    hipLaunchKernelGGL((hybrid_rows<T>), dim3((unsigned)band, rows, (unsigned)((grid + band - 1) / band)), dim3(kThreads), 0,
                       static_cast<hipStream_t>(stream), A, B, sp, (unsigned long long)npts, nfull, row_full, row_half,
                       top_is_zero, alpha_top, full, half, delta, alpha, vec_ok);
  } else {
    hipLaunchKernelGGL((hybrid_levels<T>), dim3((unsigned)grid), dim3(kThreads), 0, static_cast<hipStream_t>(stream), A,
                       B, sp, (unsigned long long)npts, nfull, row_full, row_half, top_is_zero, alpha_top, full, half,
                       delta, alpha, vec_ok);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(EKM_ERR_HIP, "hybrid_levels launch: %s", hipGetErrorString(e));
  return EKM_OK;
}

template <class T>
static int launch_any_le(int dev, void* stream, const T* sp, size_t n, T a0, T b0, T thresh, int32_t* flag) {
  if (!sp || !flag) return set_error(EKM_ERR_ARG, "any_le: null pointer");
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  const int cus = device_cus(dev);
  if (cus <= 0) return cus;
  unsigned long long want = (n + kThreads - 1) / kThreads, cap = (unsigned long long)cus * 8;
  if (want == 0) want = 1;
  hipLaunchKernelGGL((any_le<T>), dim3((unsigned)(want < cap ? want : cap)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), sp, (unsigned long long)n, a0, b0, thresh, flag);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(EKM_ERR_HIP, "any_le launch: %s", hipGetErrorString(e));
  return EKM_OK;
}

}  // namespace ekm

extern "C" {

int ekm_pressure_on_hybrid_levels_f32(int dev, void* stream, const float* A, const float* B, const float* sp,
                                      size_t npts, uint32_t nfull, const int32_t* row_full, const int32_t* row_half,
                                      int top_is_zero, float alpha_top, float* full, float* half, float* delta,
                                      float* alpha) {
  return ekm::launch_hybrid<float>(dev, stream, A, B, sp, npts, nfull, row_full, row_half, top_is_zero, alpha_top, full,
                                   half, delta, alpha);
}

int ekm_pressure_on_hybrid_levels_f64(int dev, void* stream, const double* A, const double* B, const double* sp,
                                      size_t npts, uint32_t nfull, const int32_t* row_full, const int32_t* row_half,
                                      int top_is_zero, double alpha_top, double* full, double* half, double* delta,
                                      double* alpha) {
  return ekm::launch_hybrid<double>(dev, stream, A, B, sp, npts, nfull, row_full, row_half, top_is_zero, alpha_top,
                                    full, half, delta, alpha);
}

int ekm_geopotential_on_hybrid_levels_f32(int dev, void* stream, const float* A, const float* B, const float* sp,
                                          const float* zs, const float* t, const float* q, size_t npts, uint32_t nfull,
                                          int top_is_zero, float alpha_top, int mode, float* out) {
  return ekm::launch_geopotential<float>(dev, stream, A, B, sp, zs, t, q, npts, nfull, top_is_zero, alpha_top, mode, out);
}

int ekm_geopotential_on_hybrid_levels_f64(int dev, void* stream, const double* A, const double* B, const double* sp,
                                          const double* zs, const double* t, const double* q, size_t npts,
                                          uint32_t nfull, int top_is_zero, double alpha_top, int mode, double* out) {
  return ekm::launch_geopotential<double>(dev, stream, A, B, sp, zs, t, q, npts, nfull, top_is_zero, alpha_top, mode, out);
}

int ekm_geopotential_thickness_from_alpha_delta_f32(int dev, void* stream, const float* t, const float* q,
                                                    const float* alpha, const float* delta, size_t npts, uint32_t nfull,
                                                    float* out) {
  return ekm::launch_geopotential<float>(dev, stream, nullptr, nullptr, nullptr, nullptr, t, q, npts, nfull, 0, 0.0f,
                                         ekm::GEO_THICKNESS, out, alpha, delta);
}

int ekm_geopotential_thickness_from_alpha_delta_f64(int dev, void* stream, const double* t, const double* q,
                                                    const double* alpha, const double* delta, size_t npts,
                                                    uint32_t nfull, double* out) {
  return ekm::launch_geopotential<double>(dev, stream, nullptr, nullptr, nullptr, nullptr, t, q, npts, nfull, 0, 0.0,
                                          ekm::GEO_THICKNESS, out, alpha, delta);
}

int ekm_any_le_f32(int dev, void* stream, const float* sp, size_t n, float a0, float b0, float thresh, int32_t* flag) {
  return ekm::launch_any_le<float>(dev, stream, sp, n, a0, b0, thresh, flag);
}

int ekm_any_le_f64(int dev, void* stream, const double* sp, size_t n, double a0, double b0, double thresh,
                   int32_t* flag) {
  return ekm::launch_any_le<double>(dev, stream, sp, n, a0, b0, thresh, flag);
}

}  // extern "C"
