// Issue-rate microbenchmark for the fp64 building blocks of thermo_math.hpp on gfx950: v_fma_f64, v_mul_f64,
// v_rcp_f64, v_ldexp_f64, v_frexp_mant/exp_f64, v_rndne_f64, conversions, and an fp32 transcendental used
// as a seed (rcp_f32 + cvt).  8 waves per SIMD, 8 independent chains per lane.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rates_f64.hip -o valu_rates_f64 ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHK(x)                                                                  \
  do {                                                                          \
    hipError_t e = (x);                                                         \
    if (e != hipSuccess) {                                                      \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

constexpr int ITERS = 2048;
constexpr int CH = 8;

template <int KIND>
__global__ __launch_bounds__(256) void k(double* out, double seed) {
  double v[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) v[c] = seed + threadIdx.x * 1e-3 + c;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (KIND == 0) v[c] = __builtin_fma(v[c], 1.0000001, 1e-7);
      if (KIND == 1) v[c] = v[c] * 1.0000001;
      if (KIND == 2) v[c] = v[c] + 1.0000001;
      if (KIND == 3) v[c] = __builtin_amdgcn_rcp(v[c]);
      if (KIND == 4) v[c] = __builtin_amdgcn_ldexp(v[c], 1);
      if (KIND == 5) v[c] = __builtin_amdgcn_frexp_mant(v[c]) + 1.0;            // frexp_mant + add
      if (KIND == 6) v[c] = __builtin_rint(v[c]) + 0.25;                         // rndne + add
      if (KIND == 7) v[c] = (double)__builtin_amdgcn_rcpf((float)v[c]);          // cvt_f32_f64 + rcp_f32 + cvt_f64_f32
      if (KIND == 8) v[c] = (double)((float)v[c]);                               // two conversions
      if (KIND == 9) v[c] = v[c] > 0.5 ? v[c] + 1.0 : v[(c + 1) % CH];           // cmp + cndmask x2 + add
      if (KIND == 10) v[c] = (double)__builtin_amdgcn_frexp_exp(v[c]) + v[c];    // frexp_exp + cvt + add
    }
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += v[c];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
int run(const char* name, double* out, int blocks, double per_iter_instr) {
  hipEvent_t a, b;
  CHK(hipEventCreate(&a));
  CHK(hipEventCreate(&b));
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.5);
  CHK(hipEventRecord(a));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.5);
  CHK(hipEventRecord(b));
  CHK(hipDeviceSynchronize());
  float ms;
  CHK(hipEventElapsedTime(&ms, a, b));
  ms /= 5;
  const double winstr = blocks * 4.0 * ITERS * per_iter_instr;  // wave-instructions
  printf("%-22s %8.3f ms  %6.3f wave-instr/clk/SIMD @2.4GHz (1024 SIMDs)  [%g instr/iter/chain assumed]\n", name, ms,
         winstr / (ms * 1e-3) / 2.4e9 / 1024, per_iter_instr / CH);
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CHK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount * 8;
  double* out;
  CHK(hipMalloc(&out, blocks * 256 * sizeof(double)));
  printf("%s, %d CUs\n", p.name, p.multiProcessorCount);
  run<0>("fma_f64", out, blocks, CH);
  run<1>("mul_f64", out, blocks, CH);
  run<2>("add_f64", out, blocks, CH);
  run<3>("rcp_f64", out, blocks, CH);
  run<4>("ldexp_f64", out, blocks, CH);
  run<5>("frexp_mant_f64+add", out, blocks, 2 * CH);
  run<6>("rndne_f64+add", out, blocks, 2 * CH);
  run<7>("cvt+rcp_f32+cvt", out, blocks, 3 * CH);
  run<8>("cvt+cvt", out, blocks, 2 * CH);
  run<9>("cmp+cnd+cnd+add_f64", out, blocks, 4 * CH);
  run<10>("frexp_exp+cvt+add", out, blocks, 3 * CH);
  return 0;
}
