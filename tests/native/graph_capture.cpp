// The compute entry points of libekm_thermo.so allocate nothing, copy nothing and do not synchronise:
// they can be recorded into a hipGraph and replayed (include/ekm_thermo.h "Conventions").
// Captures ekm_pipeline_svp_td_rh_f32 + ekm_wet_bulb_temperature_from_specific_humidity_f32 (Newton and bisection) on a stream,
// replays the graph on fresh inputs and compares with direct launches.  Exit code 0 = identical.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/ekm_thermo.h"

#define CHK(x)                                                                       \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);          \
      return 2;                                                                      \
    }                                                                                \
  } while (0)
#define EKM(x)                                                              \
  do {                                                                      \
    if ((x) < 0) {                                                          \
      printf("ekm error at line %d: %s\n", __LINE__, ekm_last_error());     \
      return 3;                                                             \
    }                                                                       \
  } while (0)

int main() {
  const size_t n = 1 << 20;
  if (ekm_init() < 0) {
    printf("no device: %s\n", ekm_last_error());
    return 4;
  }
  float *t, *q, *p, *o[5], *r[5];
  for (float** x : {&t, &q, &p}) EKM(ekm_malloc(0, n * 4, (void**)x));
  for (int i = 0; i < 5; ++i) {
    EKM(ekm_malloc(0, n * 4, (void**)&o[i]));
    EKM(ekm_malloc(0, n * 4, (void**)&r[i]));
  }
  EKM(ekm_synth_fill_f32(0, nullptr, t, q, p, 0, n, n / 8, 8, 42));
  CHK(hipDeviceSynchronize());
  ekm_operand ot = {t, EKM_FIELD, 0, 0, 0, nullptr, nullptr}, oq = {q, EKM_FIELD, 0, 0, 0, nullptr, nullptr},
              op = {p, EKM_FIELD, 0, 0, 0, nullptr, nullptr};

  hipStream_t s;
  CHK(hipStreamCreate(&s));
  hipGraph_t graph;
  CHK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  EKM(ekm_pipeline_svp_td_rh_f32(0, s, &ot, &oq, &op, o[0], o[1], o[2], n));
  EKM(ekm_wet_bulb_temperature_from_specific_humidity_f32(0, s, &ot, &oq, &op, EKM_EPT_IFS, EKM_T_NEWTON, o[3], n));
  // the bisection keeps a device-resident lattice table, computed on first use: this IS the first use, so the fill is
  // recorded into the graph in front of the kernel (and not marked done: the direct launch below fills it again)
  EKM(ekm_wet_bulb_temperature_from_specific_humidity_f32(0, s, &ot, &oq, &op, EKM_EPT_IFS, EKM_T_BISECT, o[4], n));
  CHK(hipStreamEndCapture(s, &graph));
  hipGraphExec_t exec;
  CHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));

  for (int round = 0; round < 3; ++round) {
    EKM(ekm_synth_fill_f32(0, s, t, q, p, 0, n, n / 8, 8, 100 + round));  // new inputs, same buffers
    for (int i = 0; i < 5; ++i) CHK(hipMemsetAsync(o[i], 0xff, n * 4, s));
    CHK(hipGraphLaunch(exec, s));
    EKM(ekm_pipeline_svp_td_rh_f32(0, s, &ot, &oq, &op, r[0], r[1], r[2], n));
    EKM(ekm_wet_bulb_temperature_from_specific_humidity_f32(0, s, &ot, &oq, &op, EKM_EPT_IFS, EKM_T_NEWTON, r[3], n));
    EKM(ekm_wet_bulb_temperature_from_specific_humidity_f32(0, s, &ot, &oq, &op, EKM_EPT_IFS, EKM_T_BISECT, r[4], n));
    CHK(hipStreamSynchronize(s));
    std::vector<float> a(n), b(n);
    for (int i = 0; i < 5; ++i) {
      CHK(hipMemcpy(a.data(), o[i], n * 4, hipMemcpyDeviceToHost));
      CHK(hipMemcpy(b.data(), r[i], n * 4, hipMemcpyDeviceToHost));
      if (std::memcmp(a.data(), b.data(), n * 4) != 0) {
        printf("round %d output %d: graph replay differs from the direct launch\n", round, i);
        return 1;
      }
      if (!(a[12345] > 0.0f) || std::isnan(a[n - 1])) {
        printf("round %d output %d: implausible value %g\n", round, i, a[12345]);
        return 1;
      }
    }
  }
  printf("graph capture + 3 replays: identical to direct launches\n");

  // ---- no entry point waits on the host: the FIRST use of a table-driven function (bolton35 bisection: its own table)
  // on stream s3 while stream s is capturing in global mode.  A host synchronisation here would be illegal and would
  // invalidate the capture; the table fill must go onto s3 with an event behind it instead.
  hipStream_t s3;
  CHK(hipStreamCreate(&s3));
  hipGraph_t g2;
  CHK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  EKM(ekm_pipeline_svp_td_rh_f32(0, s, &ot, &oq, &op, o[0], o[1], o[2], n));
  EKM(ekm_wet_bulb_temperature_from_specific_humidity_f32(0, s3, &ot, &oq, &op, EKM_EPT_BOLTON35, EKM_T_BISECT, o[3], n));
  // ... and a second stream right behind it, before the fill can have been seen complete: it must wait on the device
  hipStream_t s4;
  CHK(hipStreamCreate(&s4));
  EKM(ekm_wet_bulb_temperature_from_specific_humidity_f32(0, s4, &ot, &oq, &op, EKM_EPT_BOLTON35, EKM_T_BISECT, o[4], n));
  hipError_t endrc = hipStreamEndCapture(s, &g2);
  if (endrc != hipSuccess) {
    printf("the capture on s was invalidated by a launch on another stream: %s\n", hipGetErrorString(endrc));
    return 1;
  }
  CHK(hipStreamSynchronize(s3));
  CHK(hipStreamSynchronize(s4));
  EKM(ekm_wet_bulb_temperature_from_specific_humidity_f32(0, s3, &ot, &oq, &op, EKM_EPT_BOLTON35, EKM_T_BISECT, r[3], n));
  CHK(hipStreamSynchronize(s3));
  {
    std::vector<float> a(n), b(n), c(n);
    CHK(hipMemcpy(a.data(), o[3], n * 4, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(c.data(), o[4], n * 4, hipMemcpyDeviceToHost));
    CHK(hipMemcpy(b.data(), r[3], n * 4, hipMemcpyDeviceToHost));
    if (std::memcmp(a.data(), b.data(), n * 4) != 0 || std::memcmp(c.data(), b.data(), n * 4) != 0) {
      printf("first use on a side stream during a capture: results differ from a later launch\n");
      return 1;
    }
  }
  printf("first use of a table op on side streams during a global capture: capture intact, results identical\n");

  // ---- ekm_prepare_tables: after it a capture of a table-driven function holds the kernel alone (no fill node)
  EKM(ekm_prepare_tables(0));
  hipGraph_t g3;
  CHK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  EKM(ekm_wet_bulb_temperature_from_specific_humidity_f32(0, s, &ot, &oq, &op, EKM_EPT_BOLTON39, EKM_T_BISECT, o[4], n));
  CHK(hipStreamEndCapture(s, &g3));
  size_t nodes = 0;
  CHK(hipGraphGetNodes(g3, nullptr, &nodes));
  if (nodes != 1) {
    printf("after ekm_prepare_tables a captured bisection launch has %zu nodes (expected 1)\n", nodes);
    return 1;
  }
  printf("ekm_prepare_tables: captured bisection launch is one node\n");

  // ---- the ABI's own graph handles (a client without HIP headers): record, allocate meanwhile, replay twice, destroy;
  // a second recording that is dropped; a default-stream recording refused
  void* es = nullptr;
  EKM(ekm_stream_create(0, &es));
  EKM(ekm_graph_begin(0, es));
  float* late = nullptr;
  EKM(ekm_malloc(0, n * 4, (void**)&late));  // relaxed capture: allocation is allowed while recording
  EKM(ekm_pipeline_svp_td_rh_f32(0, es, &ot, &oq, &op, o[0], o[1], late, n));
  EKM(ekm_wet_bulb_temperature_from_specific_humidity_f32(0, es, &ot, &oq, &op, EKM_EPT_IFS, EKM_T_BISECT, o[4], n));
  void* gx = nullptr;
  EKM(ekm_graph_end(0, es, &gx));
  if (!gx) {
    printf("ekm_graph_end returned no executable graph\n");
    return 1;
  }
  for (int round = 0; round < 2; ++round) {
    EKM(ekm_synth_fill_f32(0, es, t, q, p, 0, n, n / 8, 8, 200 + round));
    CHK(hipMemsetAsync(late, 0xff, n * 4, static_cast<hipStream_t>(es)));
    CHK(hipMemsetAsync(o[4], 0xff, n * 4, static_cast<hipStream_t>(es)));
    EKM(ekm_graph_launch(0, gx, es));
    EKM(ekm_pipeline_svp_td_rh_f32(0, es, &ot, &oq, &op, r[0], r[1], r[2], n));
    EKM(ekm_wet_bulb_temperature_from_specific_humidity_f32(0, es, &ot, &oq, &op, EKM_EPT_IFS, EKM_T_BISECT, r[4], n));
    EKM(ekm_stream_sync(0, es));
    std::vector<float> a(n), b(n);
    const float* pairs[2][2] = {{late, r[2]}, {o[4], r[4]}};
    for (auto& pr : pairs) {
      CHK(hipMemcpy(a.data(), pr[0], n * 4, hipMemcpyDeviceToHost));
      CHK(hipMemcpy(b.data(), pr[1], n * 4, hipMemcpyDeviceToHost));
      if (std::memcmp(a.data(), b.data(), n * 4) != 0 || !(a[777] > 0.0f)) {
        printf("ekm_graph_launch round %d: replay differs from the direct launch\n", round);
        return 1;
      }
    }
  }
  EKM(ekm_graph_destroy(0, gx));
  EKM(ekm_graph_begin(0, es));
  EKM(ekm_pipeline_svp_td_rh_f32(0, es, &ot, &oq, &op, o[0], o[1], o[2], n));
  EKM(ekm_graph_end(0, es, nullptr));  // recording dropped; the stream is usable again
  EKM(ekm_pipeline_svp_td_rh_f32(0, es, &ot, &oq, &op, o[0], o[1], o[2], n));
  EKM(ekm_stream_sync(0, es));
  if (ekm_graph_begin(0, nullptr) >= 0) {
    printf("ekm_graph_begin accepted the default stream\n");
    return 1;
  }
  EKM(ekm_free(0, late));
  EKM(ekm_stream_destroy(0, es));
  printf("ekm_graph_begin/end/launch/destroy: replays identical to direct launches\n");
  return 0;
}
