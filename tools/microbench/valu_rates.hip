// Issue-rate microbenchmark for the VALU instructions the thermo kernels are made of
// (gfx950): how many wave-instructions per cycle per SIMD for fma / mul / exp2 / log2 / rcp /
// packed fma / cndmask, with 8 waves per SIMD and 8 independent chains per lane.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHK(x)                                                                  \
  do {                                                                          \
    hipError_t e = (x);                                                         \
    if (e != hipSuccess) {                                                      \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

constexpr int ITERS = 4096;
constexpr int CH = 8;

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  float v[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) v[c] = seed + threadIdx.x * 1e-3f + c;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 w[CH / 2];
#pragma unroll
  for (int c = 0; c < CH / 2; ++c) w[c] = f2{v[2 * c], v[2 * c + 1]};
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (KIND == 0) v[c] = __builtin_fmaf(v[c], 1.0000001f, 1e-7f);
      if (KIND == 1) v[c] = v[c] * 1.0000001f;
      if (KIND == 2) v[c] = __builtin_amdgcn_exp2f(v[c]) * 0.0f + v[c];  // exp + fma
      if (KIND == 3) v[c] = __builtin_amdgcn_exp2f(v[c]);
      if (KIND == 4) v[c] = __builtin_amdgcn_logf(v[c]);
      if (KIND == 5) v[c] = __builtin_amdgcn_rcpf(v[c]);
      if (KIND == 7) v[c] = v[c] > 0.5f ? v[c] + 1.0f : v[(c + 1) % CH];
    }
    if (KIND == 6) {
#pragma unroll
      for (int c = 0; c < CH / 2; ++c) w[c] = __builtin_elementwise_fma(w[c], f2{1.0000001f, 1.0000001f}, f2{1e-7f, 1e-7f});
    }
  }
  float s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += v[c];
#pragma unroll
  for (int c = 0; c < CH / 2; ++c) s += w[c][0] + w[c][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
int run(const char* name, float* out, int blocks, double per_iter_instr) {
  hipEvent_t a, b;
  CHK(hipEventCreate(&a));
  CHK(hipEventCreate(&b));
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.5f);
  CHK(hipEventRecord(a));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.5f);
  CHK(hipEventRecord(b));
  CHK(hipDeviceSynchronize());
  float ms;
  CHK(hipEventElapsedTime(&ms, a, b));
  ms /= 5;
  const double waves = blocks * 4.0;
  const double winstr = waves * ITERS * per_iter_instr;  // wave-instructions
  printf("%-10s %8.3f ms  %8.2f G wave-instr/s  = %6.3f wave-instr/clk/SIMD @2.4GHz (1024 SIMDs)\n", name, ms,
         winstr / ms / 1e6, winstr / (ms * 1e-3) / 2.4e9 / 1024);
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CHK(hipGetDeviceProperties(&p, 0));
  const int blocks = p.multiProcessorCount * 8;
  float* out;
  CHK(hipMalloc(&out, blocks * 256 * sizeof(float)));
  printf("%s, %d CUs, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  run<0>("fma", out, blocks, CH);
  run<1>("mul", out, blocks, CH);
  run<2>("exp2+fma", out, blocks, 2 * CH);
  run<3>("exp2", out, blocks, CH);
  run<4>("log2", out, blocks, CH);
  run<5>("rcp", out, blocks, CH);
  run<6>("pk_fma", out, blocks, CH / 2);
  run<7>("cmp+cnd+add", out, blocks, 3 * CH);
  return 0;
}
