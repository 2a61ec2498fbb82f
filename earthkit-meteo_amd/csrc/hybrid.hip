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
    s = *reinterpret_cast<const Vec*>(sp + i0);
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
            d[j] = log(phn[j] / ph[j]);
            a[j] = T(1.0) - ph[j] / (phn[j] - ph[j]) * d[j];
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
  int vec_ok = (npts % V == 0) && reinterpret_cast<uintptr_t>(sp) % 16 == 0;
  for (T* o : {full, half, delta, alpha})
    if (o && reinterpret_cast<uintptr_t>(o) % 16) vec_ok = 0;
  const unsigned long long nchunk = (npts + V - 1) / V;
  const unsigned long long grid = (nchunk + kThreads - 1) / kThreads;
  if (grid > 0x7fffffffull) return set_error(EKM_ERR_ARG, "pressure_on_hybrid_levels: too many columns");
  hipLaunchKernelGGL((hybrid_levels<T>), dim3((unsigned)grid), dim3(kThreads), 0, static_cast<hipStream_t>(stream), A,
                     B, sp, (unsigned long long)npts, nfull, row_full, row_half, top_is_zero, alpha_top, full, half,
                     delta, alpha, vec_ok);
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

int ekm_any_le_f32(int dev, void* stream, const float* sp, size_t n, float a0, float b0, float thresh, int32_t* flag) {
  return ekm::launch_any_le<float>(dev, stream, sp, n, a0, b0, thresh, flag);
}

int ekm_any_le_f64(int dev, void* stream, const double* sp, size_t n, double a0, double b0, double thresh,
                   int32_t* flag) {
  return ekm::launch_any_le<double>(dev, stream, sp, n, a0, b0, thresh, flag);
}

}  // extern "C"
