// Accuracy of the gfx950 fp64 seed instructions the fp64 primitives of thermo_math.hpp start from: v_rcp_f64 as it comes,
// v_rcp_f32 widened, each after zero / one Newton step.  Build: hipcc -O3 --offload-arch=gfx950 f64_seed_accuracy.hip -o
// f64_seed_accuracy ; run on the GPU box (prints the largest relative error over 2^24 arguments in [2^-20, 2^20]).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k(const double* x, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = x[i];
  const double r0 = __builtin_amdgcn_rcp(a);
  const double r1 = __builtin_fma(r0, __builtin_fma(-a, r0, 1.0), r0);
  const double s0 = (double)__builtin_amdgcn_rcpf((float)a);
  const double s1 = __builtin_fma(s0, __builtin_fma(-a, s0, 1.0), s0);
  const double q0 = __builtin_amdgcn_rsq(a);
  out[i] = r0;
  out[n + i] = r1;
  out[2 * n + i] = s0;
  out[3 * n + i] = s1;
  out[4 * n + i] = q0;
}

int main() {
  const int n = 1 << 24;
  std::vector<double> x(n), out(5 * (size_t)n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = (s >> 11) * (1.0 / 9007199254740992.0);
    x[i] = std::ldexp(1.0 + u, (int)(s % 41) - 20);
  }
  double *dx, *dout;
  hipMalloc(&dx, n * sizeof(double));
  hipMalloc(&dout, 5 * (size_t)n * sizeof(double));
  hipMemcpy(dx, x.data(), n * sizeof(double), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(out.data(), dout, 5 * (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
  const char* names[5] = {"v_rcp_f64", "v_rcp_f64 + 1 Newton", "v_rcp_f32 widened", "v_rcp_f32 + 1 Newton (fp64)", "v_rsq_f64"};
  for (int kx = 0; kx < 5; ++kx) {
    double worst = 0;
    for (int i = 0; i < n; ++i) {
      const double want = kx == 4 ? 1.0 / std::sqrt(x[i]) : 1.0 / x[i];
      const double e = std::fabs(out[(size_t)kx * n + i] - want) / want;
      if (e > worst) worst = e;
    }
    printf("%-30s max rel err %.3e\n", names[kx], worst);
  }
  return 0;
}
