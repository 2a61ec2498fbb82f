// What the memory system of an MI355X gives a streaming kernel, by the MIX of its streams: R input streams read and W
// output streams written, one 16-B vector per lane per stream, 256 lanes x 4 KiB tiles handed out in order -- the launch
// shape, the non-temporal loads and stores and the field size (3600 x 1800 x 137 fp32 points per stream) of the thermo map
// kernels, with no arithmetic beyond one add per stream.  The thermo kernels range from 2 reads + 1 write (theta) to
// 2 reads + 6 writes (the six-output pipeline with the pressure given per level): this table is the ceiling each of them
// is to be judged against.
// Build: hipcc -O3 --offload-arch=gfx950 stream_mix.hip -o stream_mix ; run on the GPU box: ./stream_mix [reps]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x)                                                                     \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kMaxStreams = 9;

struct Ptrs {
  f4* p[kMaxStreams];
};

template <int R, int W>
__global__ __launch_bounds__(256) void mix(Ptrs in, Ptrs out, unsigned long long nvec, float* sink) {
  const unsigned long long v = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  if (v >= nvec) return;
  f4 acc = {1.0f, 2.0f, 3.0f, 4.0f};
#pragma unroll
  for (int i = 0; i < R; ++i) acc += __builtin_nontemporal_load(in.p[i] + v);
#pragma unroll
  for (int o = 0; o < W; ++o) __builtin_nontemporal_store(acc + (float)o, out.p[o] + v);
  if (W == 0 && acc[0] + acc[1] + acc[2] + acc[3] == -1.2345f) *sink = acc[0];  // never true: keeps the loads alive
}

// Launch-shape variants of one mix, timed in ONE process on the SAME buffers (placement differences between processes
// are as large as the effects looked for): NT = 256 / 512 / 1024 lanes per workgroup, TILES consecutive tiles per
// workgroup, LD_NT / ST_NT non-temporal or plain accesses, XCD: workgroup b takes tile (b % 8) * (tiles / 8) + b / 8, so
// that each of the eight XCDs (workgroups are dealt to them round-robin) walks one contiguous eighth of every stream.
template <int R, int W, int NT, int TILES, bool LD_NT, bool ST_NT, bool XCD>
__global__ __launch_bounds__(NT) void mixv(Ptrs in, Ptrs out, unsigned long long nvec) {
  unsigned long long b = blockIdx.x;
  if (XCD) {
    const unsigned long long per = gridDim.x / 8;
    if (b < per * 8) b = (b % 8) * per + b / 8;
  }
#pragma unroll
  for (int k = 0; k < TILES; ++k) {
    const unsigned long long v = (b * TILES + k) * NT + threadIdx.x;
    if (v >= nvec) return;
    f4 acc = {1.0f, 2.0f, 3.0f, 4.0f};
#pragma unroll
    for (int i = 0; i < R; ++i) acc += LD_NT ? __builtin_nontemporal_load(in.p[i] + v) : in.p[i][v];
#pragma unroll
    for (int o = 0; o < W; ++o) {
      if (ST_NT)
        __builtin_nontemporal_store(acc + (float)o, out.p[o] + v);
      else
        out.p[o][v] = acc + (float)o;
    }
  }
}

template <int R, int W, int NT, int TILES, bool LD_NT, bool ST_NT, bool XCD>
int runv(const char* what, const Ptrs& in, const Ptrs& out, unsigned long long nvec, int reps) {
  hipEvent_t a, b;
  CHK(hipEventCreate(&a));
  CHK(hipEventCreate(&b));
  const unsigned grid = (unsigned)((nvec + (unsigned long long)NT * TILES - 1) / ((unsigned long long)NT * TILES));
  std::vector<float> ms;
  for (int r = 0; r < reps + 2; ++r) {
    CHK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((mixv<R, W, NT, TILES, LD_NT, ST_NT, XCD>), dim3(grid), dim3(NT), 0, 0, in, out, nvec);
    CHK(hipEventRecord(b, 0));
    CHK(hipEventSynchronize(b));
    float t;
    CHK(hipEventElapsedTime(&t, a, b));
    if (r >= 2) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double med = ms[ms.size() / 2], bytes = 16.0 * nvec * (R + W);
  printf("  %d + %d  %-44s median %7.3f ms  min %7.3f ms  %7.1f GB/s\n", R, W, what, med, ms[0], bytes / med * 1e-6);
  fflush(stdout);
  return 0;
}

template <int R, int W>
int variants(const Ptrs& in, const Ptrs& out, unsigned long long nvec, int reps) {
  for (int pass = 0; pass < 2; ++pass) {  // twice: drift within the process shows as a difference between the passes
    if (runv<R, W, 256, 1, true, true, false>("256 lanes, 1 tile, nt loads, nt stores (the map kernels)", in, out, nvec, reps)) return 1;
    if (runv<R, W, 256, 2, true, true, false>("2 tiles per workgroup", in, out, nvec, reps)) return 1;
    if (runv<R, W, 256, 4, true, true, false>("4 tiles per workgroup", in, out, nvec, reps)) return 1;
    if (runv<R, W, 512, 1, true, true, false>("512 lanes", in, out, nvec, reps)) return 1;
    if (runv<R, W, 1024, 1, true, true, false>("1024 lanes", in, out, nvec, reps)) return 1;
    if (runv<R, W, 256, 1, false, true, false>("plain loads", in, out, nvec, reps)) return 1;
    if (runv<R, W, 256, 1, true, false, false>("plain stores", in, out, nvec, reps)) return 1;
    if (runv<R, W, 256, 1, false, false, false>("plain loads and stores", in, out, nvec, reps)) return 1;
    if (runv<R, W, 256, 1, true, true, true>("XCD-contiguous tiles", in, out, nvec, reps)) return 1;
    if (runv<R, W, 256, 4, true, true, true>("XCD-contiguous, 4 tiles per workgroup", in, out, nvec, reps)) return 1;
  }
  return 0;
}

template <int R, int W>
int run(const Ptrs& in, const Ptrs& out, unsigned long long nvec, float* sink, int reps) {
  hipEvent_t a, b;
  CHK(hipEventCreate(&a));
  CHK(hipEventCreate(&b));
  const unsigned grid = (unsigned)((nvec + 255) / 256);
  std::vector<float> ms;
  for (int r = 0; r < reps + 2; ++r) {
    CHK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((mix<R, W>), dim3(grid), dim3(256), 0, 0, in, out, nvec, sink);
    CHK(hipEventRecord(b, 0));
    CHK(hipEventSynchronize(b));
    float t;
    CHK(hipEventElapsedTime(&t, a, b));
    if (r >= 2) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double med = ms[ms.size() / 2], bytes = 16.0 * nvec * (R + W);
  printf("%d reads + %d writes  (%4.0f %% writes)  %2d B/pt  median %7.3f ms  min %7.3f ms  %7.1f GB/s  %.3f of 8 TB/s\n", R, W,
         100.0 * W / (R + W), 4 * (R + W), med, ms[0], bytes / med * 1e-6, bytes / med * 1e-6 / 8000.0);
  fflush(stdout);
  return 0;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 9;
  const unsigned long long n = 3600ull * 1800ull * 137ull, nvec = n / 4;
  Ptrs in{}, out{};
  for (int i = 0; i < 3; ++i) {
    CHK(hipMalloc((void**)&in.p[i], n * 4));
    CHK(hipMemset(in.p[i], 0, n * 4));
  }
  for (int o = 0; o < 6; ++o) CHK(hipMalloc((void**)&out.p[o], n * 4));
  float* sink;
  CHK(hipMalloc((void**)&sink, 4));
  CHK(hipDeviceSynchronize());
  printf("stream mix, %llu fp32 points per stream, float4 per lane, non-temporal, in-order 4-KiB tiles\n", n);
  if (run<1, 0>(in, out, nvec, sink, reps)) return 1;
  if (run<3, 0>(in, out, nvec, sink, reps)) return 1;
  if (run<2, 1>(in, out, nvec, sink, reps)) return 1;   // theta
  if (run<1, 1>(in, out, nvec, sink, reps)) return 1;   // copy (celsius_to_kelvin, es)
  if (run<3, 1>(in, out, nvec, sink, reps)) return 1;   // rh, theta_e, wet-bulb
  if (run<3, 3>(in, out, nvec, sink, reps)) return 1;   // P3, p a field
  if (run<2, 3>(in, out, nvec, sink, reps)) return 1;   // P3, p per level
  if (run<3, 6>(in, out, nvec, sink, reps)) return 1;   // P5, p a field
  if (run<2, 6>(in, out, nvec, sink, reps)) return 1;   // P5, p per level / hybrid
  if (run<0, 1>(in, out, nvec, sink, reps)) return 1;   // pressure_on_hybrid_levels
  if (run<0, 6>(in, out, nvec, sink, reps)) return 1;
  if (argc > 2) {  // ./stream_mix reps variants
    printf("launch-shape variants, one process, the same buffers\n");
    if (variants<3, 6>(in, out, nvec, reps)) return 1;
    if (variants<2, 6>(in, out, nvec, reps)) return 1;
    if (variants<3, 3>(in, out, nvec, reps)) return 1;
  }
  return 0;
}
