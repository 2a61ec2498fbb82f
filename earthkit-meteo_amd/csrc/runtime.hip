// Runtime half of libekm_thermo.so: device bookkeeping, memory / stream / event
// wrappers, error reporting, launch tuning and the on-device synthetic input
// generator.  Everything here is thin HIP plumbing behind the C ABI declared
// in include/ekm_thermo.h; nothing throws across the boundary.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/ekm_thermo.h"
#include "thermo_math.hpp"

namespace ekm {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

static int hip_fail(hipError_t e, const char* what) {
  return set_error(EKM_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

#define EKM_HIP(call)                                 \
  do {                                                \
    hipError_t e_ = (call);                           \
    if (e_ != hipSuccess) return hip_fail(e_, #call); \
  } while (0)

constexpr int kMaxDev = 64;
static int g_ndev = -1;
static int g_cus[kMaxDev];
static std::mutex g_mu;
static std::atomic<int> g_tiles_per_block{1};
static std::atomic<int> g_unroll{1};

static int probe() {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_ndev >= 0) return EKM_OK;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return set_error(EKM_ERR_NODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  if (n > kMaxDev) n = kMaxDev;
  for (int d = 0; d < n; ++d) {
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, d);
    if (e != hipSuccess) return hip_fail(e, "hipGetDeviceProperties");
    g_cus[d] = prop.multiProcessorCount;
  }
  g_ndev = n;
  return EKM_OK;
}

int use_device(int dev) {
  int rc = probe();
  if (rc != EKM_OK) return rc;
  if (dev < 0 || dev >= g_ndev) return set_error(EKM_ERR_NODEV, "device %d of %d does not exist", dev, g_ndev);
  EKM_HIP(hipSetDevice(dev));
  return EKM_OK;
}

int device_cus(int dev) {
  int rc = probe();
  if (rc != EKM_OK) return rc;
  if (dev < 0 || dev >= g_ndev) return set_error(EKM_ERR_NODEV, "device %d of %d does not exist", dev, g_ndev);
  return g_cus[dev];
}

static std::atomic<int> g_tuning_user_set{0};  // ekm_set_tuning was called (or EKM_TILES / EKM_UNROLL set): launch heuristics stand back
int tuning_user_set() { return g_tuning_user_set.load(std::memory_order_relaxed); }
int tuning_tiles_per_block() { return g_tiles_per_block.load(std::memory_order_relaxed); }
int tuning_unroll() { return g_unroll.load(std::memory_order_relaxed); }

static int env_int(const char* name, int dflt, int lo, int hi) {
  const char* v = getenv(name);
  if (!v || !*v) return dflt;
  const long x = strtol(v, nullptr, 10);
  return x < lo ? lo : (x > hi ? hi : (int)x);
}
// secondary launch parameters: environment at first use, ekm_set_tuning_param afterwards
struct Param {
  const char* name;
  const char* env;
  int dflt, lo, hi;
  std::atomic<int> value{-1};
};
static Param g_params[] = {
    {"lev_per_wg", "EKM_LEV_PER_WG", 0, 0, 1024},
    {"hybrid_band_kb", "EKM_HYBRID_BAND_KB", 8192, 4, 1 << 20},
    {"table_tiles", "EKM_TABLE_TILES", 0, 0, 4096},
    {"geo_chunk_levels", "EKM_GEO_CHUNK_LEVELS", 1 << 20, 1, 1 << 20},
    {"f64_plain", "EKM_F64_PLAIN", 0, 0, 1},
    {"bisect_exact", "EKM_BISECT_EXACT", 0, 0, 1},
    {"hybrid_rows", "EKM_HYBRID_ROWS", 1, 0, 1},
};
static int param(int i) {
  int v = g_params[i].value.load(std::memory_order_relaxed);
  if (v < 0) {
    v = env_int(g_params[i].env, g_params[i].dflt, g_params[i].lo, g_params[i].hi);
    g_params[i].value.store(v, std::memory_order_relaxed);
  }
  return v;
}
int tuning_lev_per_wg() { return param(0); }
int tuning_hybrid_band_bytes() { return param(1) * 1024; }
int tuning_table_tiles() { return param(2); }
int tuning_geo_chunk_levels() { return param(3); }
int tuning_f64_plain() { return param(4); }
int tuning_bisect_exact() { return param(5); }
int tuning_hybrid_rows() { return param(6); }


// ---- synthetic atmosphere on the device (SURVEY.md 8d distribution) ---------
// Counter-based: every value is a pure function of (seed, global point index),
// so shards of one global field can be filled independently on different GPUs.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

__device__ __forceinline__ double u01(uint64_t bits) {  // (0, 1)
  return ((bits >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

__host__ __device__ inline double synth_level_p(unsigned k, unsigned nlev) {
  return nlev > 1 ? 1000.0 + (101325.0 - 1000.0) * (double)k / (double)(nlev - 1) : 101325.0;
}

template <class T>
__global__ __launch_bounds__(256) void synth_fill(T* t, T* q, T* p, uint64_t first, size_t n, uint64_t inner,
                                                 uint32_t nlev, uint64_t seed, int p_given) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint64_t g = first + i;
    uint64_t lev = g / inner;
    if (lev >= nlev) lev = nlev - 1;
    const uint64_t h0 = mix64(seed ^ (g * 0xd1342543de82ef95ull));
    const uint64_t h1 = mix64(h0 + 1), h2 = mix64(h0 + 2), h3 = mix64(h0 + 3);
    const double pl = synth_level_p((unsigned)lev, nlev);
    // with a level-vector p (p == NULL) the point sits exactly on its level
    const double pv = p_given ? (double)p[i] : (p ? pl * (1.0 + 0.05 * (2.0 * u01(h0) - 1.0)) : pl);
    // Box-Muller normal, sigma = 8 K around the standard atmosphere
    const double nrm = sqrt(-2.0 * log(u01(h1))) * cos(6.283185307179586 * u01(h2));
    double tv = fmax(288.15 * pow(pv / 101325.0, 0.190263), 216.65) + 8.0 * nrm;
    tv = fmin(fmax(tv, 180.0), 330.0);
    const double rh = 1.0 + 99.0 * u01(h3);
    // q = specific_humidity_from_relative_humidity(t, rh, p), capped (thermo.py:629-663)
    const double e = rh * es_mixed<double>(tv) / 100.0;
    double qv = q_from_e<double>(e, pv, 1e-4);
    double qcap = 0.04;
    // around a GIVEN pressure (hybrid model levels, which reach 1 Pa) the cap follows the pressure: the moisture
    // the real atmosphere holds falls off roughly as p^3 (6e-3 at 500 hPa, 1e-3 at 300 hPa, 4e-5 at 100 hPa) down to
    // the stratospheric 3e-6 -- a q of 0.04 at 38 Pa is not a state any model level is ever in
    if (p_given) qcap = fmax(3e-6, 0.04 * (pv / 101325.0) * (pv / 101325.0) * (pv / 101325.0));
    qv = fmin(qv, qcap);
    if (!(qv == qv)) qv = 3e-6;
    t[i] = (T)tv;
    q[i] = (T)qv;
    if (p && !p_given) p[i] = (T)pv;
  }
}

template <class T>
__global__ void synth_levels(T* pl, uint32_t nlev) {
  const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nlev) pl[k] = (T)synth_level_p(k, nlev);
}

template <class T>
static int synth_fill_launch(int dev, void* stream, T* t, T* q, T* p, uint64_t first, size_t n, uint64_t inner,
                             uint32_t nlev, uint64_t seed, int p_given = 0) {
  if (n == 0) return EKM_OK;
  if (!t || !q) return set_error(EKM_ERR_ARG, "synth_fill: null t/q pointer");
  if (inner == 0 || nlev == 0) return set_error(EKM_ERR_ARG, "synth_fill: inner and nlev must be > 0");
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  const int cus = device_cus(dev);
  if (cus <= 0) return cus;
  size_t want = (n + 255) / 256, cap = (size_t)cus * 8;
  const unsigned grid = (unsigned)(want < cap ? want : cap);
  hipLaunchKernelGGL((synth_fill<T>), dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), t, q, p, first, n,
                     inner, nlev, seed, p_given);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "synth_fill launch");
  return EKM_OK;
}

template <class T>
static int synth_levels_launch(int dev, void* stream, T* pl, uint32_t nlev) {
  if (!pl || nlev == 0) return set_error(EKM_ERR_ARG, "synth_levels: bad arguments");
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  hipLaunchKernelGGL((synth_levels<T>), dim3((nlev + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), pl,
                     nlev);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "synth_levels launch");
  return EKM_OK;
}

}  // namespace ekm

// ---- lookup tables of the table-driven ops (map_kernel.hpp::ensure_op_table) ----
namespace ekm {
typedef int (*table_prep_fn)(int dev);
static std::mutex g_prep_mu;
static std::vector<table_prep_fn>& prep_list() {
  static std::vector<table_prep_fn> v;  // constructed on first use: registrations run during static initialisation
  return v;
}
void register_table_prep(table_prep_fn fn) {
  std::lock_guard<std::mutex> lk(g_prep_mu);
  prep_list().push_back(fn);
}
}  // namespace ekm

namespace ekm {
// ---- the streaming reference of a launch: R input streams read, W output streams written, nothing computed -------------
// The map kernels' launch shape (256 lanes, 4-KiB tiles handed out in order, one 16-B non-temporal access per lane per
// stream) with one add per stream in place of the thermodynamics: what the memory system gives THIS set of buffers --
// bench.py times it on the arrays of the launch it has just measured, so that the kernel is compared with a ceiling that
// shares its placement (the same kernel moves by 5-10 % between processes: profiles/r04_stream_mix.txt).
struct MixPtrs {
  const void* in[4];
  void* out[8];
};
typedef float mix_vec __attribute__((ext_vector_type(4), aligned(4)));

template <int R, int W>
__global__ __launch_bounds__(256) void stream_mix(MixPtrs p, unsigned long long nvec) {
  const unsigned long long v = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  if (v >= nvec) return;
  mix_vec acc = {1.0f, 2.0f, 3.0f, 4.0f};
#pragma unroll
  for (int i = 0; i < R; ++i) acc += __builtin_nontemporal_load(static_cast<const mix_vec*>(p.in[i]) + v);
#pragma unroll
  for (int o = 0; o < W; ++o) __builtin_nontemporal_store(acc + (float)o, static_cast<mix_vec*>(p.out[o]) + v);
}

template <int R>
static int stream_mix_launch(int nout, const MixPtrs& p, unsigned long long nvec, hipStream_t s) {
  const dim3 g((unsigned)((nvec + 255) / 256)), b(256);
  switch (nout) {
    case 1: hipLaunchKernelGGL((stream_mix<R, 1>), g, b, 0, s, p, nvec); return EKM_OK;
    case 2: hipLaunchKernelGGL((stream_mix<R, 2>), g, b, 0, s, p, nvec); return EKM_OK;
    case 3: hipLaunchKernelGGL((stream_mix<R, 3>), g, b, 0, s, p, nvec); return EKM_OK;
    case 6: hipLaunchKernelGGL((stream_mix<R, 6>), g, b, 0, s, p, nvec); return EKM_OK;
    default: return set_error(EKM_ERR_ARG, "stream_mix: %d output streams (1, 2, 3 or 6)", nout);
  }
}
}  // namespace ekm

using namespace ekm;

extern "C" {

int ekm_init(void) { return probe(); }

int ekm_device_count(void) {
  int rc = probe();
  return rc != EKM_OK ? rc : g_ndev;
}

const char* ekm_last_error(void) { return g_err; }

const char* ekm_version(void) { return "ekm_thermo 0.1.0 (gfx950)"; }
int ekm_abi_version(void) { return EKM_ABI_VERSION; }

int ekm_device_name(int dev, char* buf, size_t buflen) {
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  if (!buf || buflen == 0) return set_error(EKM_ERR_ARG, "device_name: no buffer");
  hipDeviceProp_t prop;
  EKM_HIP(hipGetDeviceProperties(&prop, dev));
  snprintf(buf, buflen, "%s (%s)", prop.name, prop.gcnArchName);
  return EKM_OK;
}

int ekm_device_cus(int dev) { return device_cus(dev); }

int ekm_mem_info(int dev, size_t* free_bytes, size_t* total_bytes) {
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  size_t f = 0, t = 0;
  EKM_HIP(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return EKM_OK;
}

int ekm_malloc(int dev, size_t bytes, void** out) {
  if (!out) return set_error(EKM_ERR_ARG, "malloc: null out pointer");
  *out = nullptr;
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  EKM_HIP(hipMalloc(out, bytes ? bytes : 16));
  return EKM_OK;
}

int ekm_free(int dev, void* ptr) {
  if (!ptr) return EKM_OK;
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  EKM_HIP(hipFree(ptr));
  return EKM_OK;
}

int ekm_host_alloc(size_t bytes, void** out) {
  if (!out) return set_error(EKM_ERR_ARG, "host_alloc: null out pointer");
  *out = nullptr;
  int rc = probe();
  if (rc != EKM_OK) return rc;
  EKM_HIP(hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault));
  return EKM_OK;
}

int ekm_host_prefault(void* ptr, size_t bytes, int nthreads) {
  // Make the pages of a freshly allocated host buffer exist (writable) before a device-to-host copy lands
  // in them: first-touch faults otherwise throttle the copy 56 -> 16 GB/s.  NON-DESTRUCTIVE: the contents
  // of the buffer are never changed, so it is safe even while data is arriving in it.
  // madvise(MADV_POPULATE_WRITE) (Linux >= 5.14) where available (94 GB/s with 4 threads on the GPU box's host);
  // else a compare-and-swap of one byte per page with itself.
  if (!ptr || bytes == 0) return EKM_OK;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 16) nthreads = 16;
  char* base = static_cast<char*>(ptr);
  auto work = [base, bytes](size_t lo, size_t hi) {
#if defined(MADV_POPULATE_WRITE)
    static const bool use_madvise = env_int("EKM_PREFAULT_MADVISE", 1, 0, 1) != 0;
    if (use_madvise) {
      const uintptr_t a0 = (reinterpret_cast<uintptr_t>(base) + lo) & ~uintptr_t(4095);
      const uintptr_t a1 = (reinterpret_cast<uintptr_t>(base) + hi + 4095) & ~uintptr_t(4095);
      if (madvise(reinterpret_cast<void*>(a0), a1 - a0, MADV_POPULATE_WRITE) == 0) return;
    }
#endif
    // compare-and-swap of a byte with itself: always a locked write of the value just read (an atomic `or 0` is
    // folded into a plain load by the compiler and would not take the write fault)
    auto touch = [](char* p) {
      char v = __atomic_load_n(p, __ATOMIC_RELAXED);
      while (!__atomic_compare_exchange_n(p, &v, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
      }
    };
    for (size_t off = lo; off < hi; off += 4096) touch(base + off);
    if (hi == bytes) touch(base + bytes - 1);
  };
  const size_t pages = (bytes + 4095) / 4096, per = (pages + nthreads - 1) / nthreads * 4096;
  std::vector<std::thread> pool;
  for (int i = 1; i < nthreads; ++i) {
    const size_t lo = (size_t)i * per, hi = lo + per < bytes ? lo + per : bytes;
    if (lo < bytes) pool.emplace_back(work, lo, hi);
  }
  work(0, per < bytes ? per : bytes);
  for (auto& th : pool) th.join();
  return EKM_OK;
}

int ekm_host_free(void* ptr) {
  if (!ptr) return EKM_OK;
  EKM_HIP(hipHostFree(ptr));
  return EKM_OK;
}

static int copy(int dev, void* dst, const void* src, size_t bytes, void* stream, hipMemcpyKind kind) {
  if (bytes == 0) return EKM_OK;
  if (!dst || !src) return set_error(EKM_ERR_ARG, "memcpy: null pointer");
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  EKM_HIP(hipMemcpyAsync(dst, src, bytes, kind, static_cast<hipStream_t>(stream)));
  return EKM_OK;
}

int ekm_h2d(int dev, void* dst, const void* src, size_t bytes, void* stream) {
  return copy(dev, dst, src, bytes, stream, hipMemcpyHostToDevice);
}
int ekm_d2h(int dev, void* dst, const void* src, size_t bytes, void* stream) {
  return copy(dev, dst, src, bytes, stream, hipMemcpyDeviceToHost);
}
int ekm_d2d(int dev, void* dst, const void* src, size_t bytes, void* stream) {
  return copy(dev, dst, src, bytes, stream, hipMemcpyDeviceToDevice);
}

int ekm_memset(int dev, void* dst, int value, size_t bytes, void* stream) {
  if (bytes == 0) return EKM_OK;
  if (!dst) return set_error(EKM_ERR_ARG, "memset: null pointer");
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  EKM_HIP(hipMemsetAsync(dst, value, bytes, static_cast<hipStream_t>(stream)));
  return EKM_OK;
}

// `count` 32-bit words set to `value`, asynchronously on `stream`: how a host-side scalar (its bit pattern) reaches device
// memory without a host buffer -- nothing to keep alive, no wait, and recordable into a graph.
int ekm_fill_u32(int dev, void* dst, uint32_t value, size_t count, void* stream) {
  if (count == 0) return EKM_OK;
  if (!dst || reinterpret_cast<uintptr_t>(dst) % 4) return set_error(EKM_ERR_ARG, "fill_u32: null or unaligned pointer");
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  EKM_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(dst), (int)value, count, static_cast<hipStream_t>(stream)));
  return EKM_OK;
}

int ekm_sync(int dev) {
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  EKM_HIP(hipDeviceSynchronize());
  return EKM_OK;
}

int ekm_stream_create(int dev, void** out) {
  if (!out) return set_error(EKM_ERR_ARG, "stream_create: null out pointer");
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  hipStream_t s;
  EKM_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *out = s;
  return EKM_OK;
}

int ekm_stream_destroy(int dev, void* stream) {
  if (!stream) return EKM_OK;
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  EKM_HIP(hipStreamDestroy(static_cast<hipStream_t>(stream)));
  return EKM_OK;
}

int ekm_stream_sync(int dev, void* stream) {
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  EKM_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
  return EKM_OK;
}

int ekm_event_create(int dev, void** out) {
  if (!out) return set_error(EKM_ERR_ARG, "event_create: null out pointer");
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  hipEvent_t ev;
  EKM_HIP(hipEventCreate(&ev));
  *out = ev;
  return EKM_OK;
}

int ekm_event_destroy(int dev, void* event) {
  if (!event) return EKM_OK;
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  EKM_HIP(hipEventDestroy(static_cast<hipEvent_t>(event)));
  return EKM_OK;
}

int ekm_event_record(int dev, void* event, void* stream) {
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  EKM_HIP(hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)));
  return EKM_OK;
}

int ekm_stream_wait_event(int dev, void* stream, void* event) {
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  EKM_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0));
  return EKM_OK;
}

// ---- HIP graphs: record the launches of a stream once, replay them with one call -----------------------------------
// Every compute entry point only enqueues work on the caller's stream (no allocation, no host wait: ensure_op_table), so a
// sequence of them between ekm_graph_begin and ekm_graph_end is recorded instead of run.  Relaxed capture mode: the caller
// may allocate (ekm_malloc) while recording; copies from pageable host memory and waits cannot be recorded and fail.
int ekm_graph_begin(int dev, void* stream) {
  if (!stream) return set_error(EKM_ERR_ARG, "graph_begin: the default stream cannot be captured; pass a stream from ekm_stream_create");
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  EKM_HIP(hipStreamBeginCapture(static_cast<hipStream_t>(stream), hipStreamCaptureModeRelaxed));
  return EKM_OK;
}

int ekm_graph_end(int dev, void* stream, void** graph_exec) {
  if (graph_exec) *graph_exec = nullptr;
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  hipGraph_t g = nullptr;
  hipError_t err = hipStreamEndCapture(static_cast<hipStream_t>(stream), &g);  // always ends the capture, valid or not
  if (err != hipSuccess || !g) {
    (void)hipGetLastError();
    if (g) (void)hipGraphDestroy(g);
    return set_error(EKM_ERR_HIP, "graph_end: the capture is invalid (%s): something between begin and end could not be recorded",
                     hipGetErrorString(err));
  }
  if (!graph_exec) {  // capture abandoned by the caller
    (void)hipGraphDestroy(g);
    return EKM_OK;
  }
  hipGraphExec_t x = nullptr;
  err = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (err != hipSuccess) return set_error(EKM_ERR_HIP, "graph_end: hipGraphInstantiate: %s", hipGetErrorString(err));
  *graph_exec = x;
  return EKM_OK;
}

int ekm_graph_launch(int dev, void* graph_exec, void* stream) {
  if (!graph_exec) return set_error(EKM_ERR_ARG, "graph_launch: null graph");
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  EKM_HIP(hipGraphLaunch(static_cast<hipGraphExec_t>(graph_exec), static_cast<hipStream_t>(stream)));
  return EKM_OK;
}

int ekm_graph_destroy(int dev, void* graph_exec) {
  if (!graph_exec) return EKM_OK;
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  EKM_HIP(hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph_exec)));
  return EKM_OK;
}

int ekm_event_sync(int dev, void* event) {
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  EKM_HIP(hipEventSynchronize(static_cast<hipEvent_t>(event)));
  return EKM_OK;
}

int ekm_event_elapsed_ms(int dev, void* start, void* stop, float* ms) {
  if (!ms) return set_error(EKM_ERR_ARG, "event_elapsed_ms: null out pointer");
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  EKM_HIP(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
  return EKM_OK;
}

// ---- lookup tables of the table-driven ops (map_kernel.hpp::ensure_op_table) ----

int ekm_prepare_tables(int dev) {
  std::vector<ekm::table_prep_fn> fns;
  {
    std::lock_guard<std::mutex> lk(ekm::g_prep_mu);
    fns = ekm::prep_list();
  }
  for (ekm::table_prep_fn fn : fns) {
    const int rc = fn(dev);
    if (rc != EKM_OK) return rc;
  }
  return EKM_OK;
}

int ekm_set_tuning(int tiles_per_block, int unroll) {
  if (tiles_per_block < 0 || tiles_per_block > 65536 || unroll < 0 || unroll > 2)
    return set_error(EKM_ERR_ARG, "set_tuning: tiles_per_block in [1,65536], unroll in {1,2} (0 keeps)");
  if (tiles_per_block) g_tiles_per_block.store(tiles_per_block);
  if (unroll) g_unroll.store(unroll);
  if (tiles_per_block || unroll) g_tuning_user_set.store(1);
  return EKM_OK;
}

int ekm_get_tuning(int* tiles_per_block, int* unroll) {
  if (tiles_per_block) *tiles_per_block = g_tiles_per_block.load();
  if (unroll) *unroll = g_unroll.load();
  return EKM_OK;
}

int ekm_set_tuning_param(const char* name, int value) {
  if (!name) return set_error(EKM_ERR_ARG, "set_tuning_param: null name");
  for (Param& p : g_params) {
    if (strcmp(p.name, name) == 0) {
      if (value < p.lo || value > p.hi)
        return set_error(EKM_ERR_ARG, "set_tuning_param: %s must be in [%d, %d]", name, p.lo, p.hi);
      p.value.store(value);
      return EKM_OK;
    }
  }
  return set_error(EKM_ERR_ARG, "set_tuning_param: unknown parameter '%s' (lev_per_wg, hybrid_band_kb, table_tiles, geo_chunk_levels, f64_plain, bisect_exact, hybrid_rows)", name);
}

int ekm_synth_fill_f32(int dev, void* stream, float* t, float* q, float* p, uint64_t first, size_t n, uint64_t inner,
                       uint32_t nlev, uint64_t seed) {
  return synth_fill_launch<float>(dev, stream, t, q, p, first, n, inner, nlev, seed);
}
int ekm_synth_fill_f64(int dev, void* stream, double* t, double* q, double* p, uint64_t first, size_t n,
                       uint64_t inner, uint32_t nlev, uint64_t seed) {
  return synth_fill_launch<double>(dev, stream, t, q, p, first, n, inner, nlev, seed);
}
int ekm_synth_fill_given_p_f32(int dev, void* stream, float* t, float* q, const float* p, uint64_t first, size_t n,
                               uint64_t seed) {
  if (!p) return set_error(EKM_ERR_ARG, "synth_fill_given_p: null p");
  return synth_fill_launch<float>(dev, stream, t, q, const_cast<float*>(p), first, n, 1, 1, seed, 1);
}
int ekm_synth_fill_given_p_f64(int dev, void* stream, double* t, double* q, const double* p, uint64_t first, size_t n,
                               uint64_t seed) {
  if (!p) return set_error(EKM_ERR_ARG, "synth_fill_given_p: null p");
  return synth_fill_launch<double>(dev, stream, t, q, const_cast<double*>(p), first, n, 1, 1, seed, 1);
}
int ekm_stream_mix(int dev, void* stream, const void* const* ins, int nin, void* const* outs, int nout, size_t bytes) {
  if (nin < 0 || nin > 3 || !outs || (nin > 0 && !ins)) return set_error(EKM_ERR_ARG, "stream_mix: 0..3 input streams");
  if (bytes % 16 != 0 || bytes / 16 / 256 > 0x7fffffffull) return set_error(EKM_ERR_ARG, "stream_mix: bytes must be a multiple of 16 (and < 2^43)");
  MixPtrs p{};
  for (int i = 0; i < nin; ++i) {
    if (!ins[i] || reinterpret_cast<uintptr_t>(ins[i]) % 4) return set_error(EKM_ERR_ARG, "stream_mix: null or unaligned input %d", i);
    p.in[i] = ins[i];
  }
  for (int o = 0; o < nout && o < 8; ++o) {
    if (!outs[o] || reinterpret_cast<uintptr_t>(outs[o]) % 4) return set_error(EKM_ERR_ARG, "stream_mix: null or unaligned output %d", o);
    p.out[o] = outs[o];
  }
  if (bytes == 0) return EKM_OK;
  int rc = use_device(dev);
  if (rc != EKM_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned long long nvec = bytes / 16;
  switch (nin) {
    case 0: rc = stream_mix_launch<0>(nout, p, nvec, s); break;
    case 1: rc = stream_mix_launch<1>(nout, p, nvec, s); break;
    case 2: rc = stream_mix_launch<2>(nout, p, nvec, s); break;
    default: rc = stream_mix_launch<3>(nout, p, nvec, s); break;
  }
  if (rc != EKM_OK) return rc;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(EKM_ERR_HIP, "stream_mix launch: %s", hipGetErrorString(e));
  return EKM_OK;
}

int ekm_synth_levels_f32(int dev, void* stream, float* p_levels, uint32_t nlev) {
  return synth_levels_launch<float>(dev, stream, p_levels, nlev);
}
int ekm_synth_levels_f64(int dev, void* stream, double* p_levels, uint32_t nlev) {
  return synth_levels_launch<double>(dev, stream, p_levels, nlev);
}

}  // extern "C"
