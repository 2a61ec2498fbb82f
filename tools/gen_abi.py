#!/usr/bin/env python3
"""Single source of truth for the C ABI of libekm_thermo.so.

Emits (all committed):
  include/ekm_thermo.h                          the public header
  earthkit-meteo_amd/csrc/gen/entries_<g>_<dtype>.hip   extern "C" definitions, one file per group and dtype
  earthkit-meteo_amd/csrc/gen/host_entries.inc  the host-twin definitions (test infrastructure)
  earthkit-meteo_amd/ekm_hip/_optable.py        the ctypes signature table

Run `python tools/gen_abi.py` after editing OPS.
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PHASE = ("phase", "EKM_PHASE_*", 3)
EPT = ("ept_method", "EKM_EPT_*", 3)
LCL = ("method", "EKM_LCL_*", 2)
EPTM = ("method", "EKM_EPT_*", 3)
TM2 = ("t_method", "EKM_T_BISECT | EKM_T_NEWTON", 2)
TM3 = ("t_method", "EKM_T_BISECT | EKM_T_NEWTON | EKM_T_DIRECT", 3)

# name, functor, inputs, outputs, int params, has eps, reference lines (thermo.py unless noted), group
OPS = [
    ("celsius_to_kelvin", "OpCelsiusToKelvin", ["t"], ["out"], [], False, "21-35", "basic"),
    ("kelvin_to_celsius", "OpKelvinToCelsius", ["t"], ["out"], [], False, "38-52", "basic"),
    ("specific_humidity_from_mixing_ratio", "OpQFromW", ["w"], ["out"], [], False, "55-77", "basic"),
    ("mixing_ratio_from_specific_humidity", "OpWFromQ", ["q"], ["out"], [], False, "80-102", "basic"),
    ("vapour_pressure_from_specific_humidity", "OpEFromQ", ["q", "p"], ["out"], [], False, "105-131", "basic"),
    ("vapour_pressure_from_mixing_ratio", "OpEFromW", ["w", "p"], ["out"], [], False, "134-159", "basic"),
    ("specific_humidity_from_vapour_pressure", "OpQFromE", ["e", "p"], ["out"], [], True, "162-196", "basic"),
    ("mixing_ratio_from_vapour_pressure", "OpWFromE", ["e", "p"], ["out"], [], True, "199-232", "basic"),
    ("saturation_vapour_pressure", "OpSvp", ["t"], ["out"], [PHASE], False, "235-279; es_comp.py:31-79,133-166", "svp"),
    ("saturation_mixing_ratio", "OpSatW", ["t", "p"], ["out"], [PHASE], False, "282-310", "svp"),
    ("saturation_specific_humidity", "OpSatQ", ["t", "p"], ["out"], [PHASE], False, "313-341", "svp"),
    ("saturation_vapour_pressure_slope", "OpSvpSlope", ["t"], ["out"], [PHASE], False, "344-364; es_comp.py:82-106,169-200", "svp"),
    ("saturation_mixing_ratio_slope", "OpSatWSlope", ["t", "p"], ["out"], [PHASE], True, "367-415", "svp"),
    ("saturation_specific_humidity_slope", "OpSatQSlope", ["t", "p"], ["out"], [PHASE], True, "418-467", "svp"),
    ("saturation_mixing_ratio_slope_from_es", "OpSatWSlopeFromEs", ["p", "es", "es_slope"], ["out"], [], True,
     "407-415 (caller-supplied es, es_slope)", "svp"),
    ("saturation_specific_humidity_slope_from_es", "OpSatQSlopeFromEs", ["p", "es", "es_slope"], ["out"], [], True,
     "459-467 (caller-supplied es, es_slope)", "svp"),
    ("temperature_from_saturation_vapour_pressure", "OpTFromEs", ["es"], ["out"], [], False, "470-491; es_comp.py:109-130", "basic"),
    ("relative_humidity_from_dewpoint", "OpRhFromTd", ["t", "td"], ["out"], [], False, "494-521", "basic"),
    ("relative_humidity_from_specific_humidity", "OpRhFromQ", ["t", "q", "p"], ["out"], [], False, "524-556", "basic"),
    ("specific_humidity_from_dewpoint", "OpQFromTd", ["td", "p"], ["out"], [], False, "559-591", "basic"),
    ("mixing_ratio_from_dewpoint", "OpWFromTd", ["td", "p"], ["out"], [], False, "594-626", "basic"),
    ("specific_humidity_from_relative_humidity", "OpQFromRh", ["t", "r", "p"], ["out"], [], False, "629-663", "basic"),
    ("dewpoint_from_relative_humidity", "OpTdFromRh", ["t", "r"], ["out"], [], False, "666-699", "basic"),
    ("dewpoint_from_specific_humidity", "OpTdFromQ", ["q", "p"], ["out"], [], False, "702-735", "basic"),
    ("virtual_temperature", "OpVirtualT", ["t", "q"], ["out"], [], False, "738-764", "basic"),
    ("virtual_potential_temperature", "OpVirtualTheta", ["t", "q", "p"], ["out"], [], False, "767-798", "basic"),
    ("potential_temperature", "OpTheta", ["t", "p"], ["out"], [], False, "801-829", "basic"),
    ("temperature_from_potential_temperature", "OpTFromTheta", ["th", "p"], ["out"], [], False, "832-858", "basic"),
    ("pressure_on_dry_adiabat", "OpPOnDryAdiabat", ["t", "t_def", "p_def"], ["out"], [], False, "861-889", "basic"),
    ("temperature_on_dry_adiabat", "OpTOnDryAdiabat", ["p", "t_def", "p_def"], ["out"], [], False, "892-920", "basic"),
    ("lcl_temperature", "OpLclT", ["t", "td"], ["out"], [LCL], False, "923-968", "basic"),
    ("lcl", "OpLcl", ["t", "td", "p"], ["t_lcl", "p_lcl"], [LCL], False, "971-1000", "basic"),
    ("ept_from_dewpoint", "OpEptFromTd", ["t", "td", "p"], ["out"], [EPTM], False, "1326-1387,1031-1040,1169-1175,1205-1213,1268-1278", "ept"),
    ("ept_from_specific_humidity", "OpEptFromQ", ["t", "q", "p"], ["out"], [EPTM], False, "1390-1415,1031-1040", "ept"),
    ("saturation_ept", "OpSatEpt", ["t", "p"], ["out"], [EPTM], False, "1418-1469,1042-1045,1177-1182,1215-1224,1280-1295", "ept"),
    ("temperature_on_moist_adiabat", "OpTOnMa", ["ept", "p"], ["out"], [EPT, TM2], False, "1472-1509,1055-1159", "moist"),
    ("wet_bulb_temperature_from_dewpoint", "OpWetBulbFromTd", ["t", "td", "p"], ["out"], [EPT, TM2], False, "1512-1549", "moist"),
    ("wet_bulb_temperature_from_specific_humidity", "OpWetBulbFromQ", ["t", "q", "p"], ["out"], [EPT, TM2], False, "1552-1590", "moist"),
    ("wet_bulb_potential_temperature_from_dewpoint", "OpWbptFromTd", ["t", "td", "p"], ["out"], [EPT, TM3], False, "1593-1634,1047-1053", "wbpt"),
    ("wet_bulb_potential_temperature_from_specific_humidity", "OpWbptFromQ", ["t", "q", "p"], ["out"], [EPT, TM3], False, "1637-1675,1047-1053", "wbpt"),
    ("specific_gas_constant", "OpGasConstant", ["q"], ["out"], [], False, "1678-1707", "basic"),
    # SURVEY.md 8f rank 4 names this free rider on the map skeleton (the only non-thermo entry point)
    ("w_from_omega", "OpWFromOmega", ["omega", "t", "p"], ["out"], [], False, "wind/array/wind.py:192-222", "wind"),
    ("pipeline_svp_td_rh", "OpPipelineSvpTdRh", ["t", "q", "p"], ["es", "td", "rh"], [], False,
     "235-279 + 702-735 + 524-556 fused (SURVEY.md 8a row a13, P3)", "pipeline"),
    ("pipeline_full", "OpPipelineFull", ["t", "q", "p"], ["theta", "es", "rh", "td", "theta_e", "tw"], [], False,
     "801-829 + 235-279 + 524-556 + 702-735 + 1390-1415 + 1552-1590(ifs,newton) fused (SURVEY.md 8a row a13, P5)", "pipeline"),
]

DTYPES = (("f32", "float"), ("f64", "double"))

HEADER_TOP = r'''/* ekm_thermo.h -- C ABI of libekm_thermo.so: MI355X (gfx950) kernels for the
 * earthkit-meteo `thermo` elementwise hot path.
 *
 * GENERATED by tools/gen_abi.py -- edit the table there.
 *
 * Reference interface this replaces: the Python call layer
 * `earthkit.meteo.thermo.<name>(*args, **kwargs)`
 * (/root/reference/src/earthkit/meteo/thermo/thermo.py:13-166, forwarding to
 * thermo/array/thermo.py).  The reference has no FFI of its own: its "backend"
 * is whatever array namespace `array_namespace(*inputs)` returns
 * (thermo/array/thermo.py:826, es_comp.py:73).  Each compute entry point below
 * is what a native backend for one of those functions binds to; the comment on
 * each names the reference lines it implements.
 *
 * Conventions
 *  - plain C types only; every function returns EKM_OK (0) or a negative
 *    EKM_ERR_* code and never throws; ekm_last_error() gives the message of the
 *    calling thread's last failure;
 *  - `dev` is a HIP device ordinal; `stream` is a hipStream_t passed as void*
 *    (NULL = the device's default stream);
 *  - data pointers of compute entry points are DEVICE pointers owned by the
 *    caller; launches are asynchronous on `stream`, allocate nothing, copy
 *    nothing and do not synchronise (graph-capturable);
 *  - an input is described by an ekm_operand: a full field of n values, one
 *    scalar, or a level vector broadcast over the field (see EKM_LEVEL_*), so a
 *    137-level pressure vector is never materialised per grid point;
 *  - outputs are full fields of n values and must not alias inputs of a
 *    different index range;
 *  - `_f32` computes in float with the CDNA4 transcendental unit, `_f64` in
 *    double; NaN/inf results follow the reference (in-band, never an error).
 */
#ifndef EKM_THERMO_H
#define EKM_THERMO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EKM_API __attribute__((visibility("default")))

/* ---- status codes ---- */
#define EKM_OK 0
#define EKM_ERR_HIP (-1)     /* a HIP runtime call failed */
#define EKM_ERR_ARG (-2)     /* bad pointer / size / operand description */
#define EKM_ERR_ENUM (-3)    /* unknown phase / method value */
#define EKM_ERR_NODEV (-4)   /* no such device */

/* ---- enums (fixed values) ---- */
/* phase of saturation_vapour_pressure & co (es_comp.py:22) */
#define EKM_PHASE_MIXED 0
#define EKM_PHASE_WATER 1
#define EKM_PHASE_ICE 2
/* equivalent-potential-temperature method (thermo/array/thermo.py:1319-1323) */
#define EKM_EPT_IFS 0
#define EKM_EPT_BOLTON35 1
#define EKM_EPT_BOLTON39 2
/* moist-adiabat inversion (thermo.py:1504-1509, 1631) */
#define EKM_T_BISECT 0
#define EKM_T_NEWTON 1
#define EKM_T_DIRECT 2
/* lcl_temperature method (thermo.py:960-968) */
#define EKM_LCL_DAVIES 0
#define EKM_LCL_BOLTON 1

/* ---- operand description ---- */
#define EKM_FIELD 0        /* data[i], i < n                                          */
#define EKM_SCALAR 1       /* data[0] for every point                                 */
#define EKM_LEVEL_MAJOR 2  /* data[i / inner]: `len` levels of `inner` points each    */
#define EKM_LEVEL_MINOR 3  /* data[i % len]: the vector runs along the fastest axis   */
#define EKM_HYBRID_FULL 4  /* pressure on hybrid full levels formed in the kernel:
                              data = surface pressure sp (`inner` values), `len` full levels,
                              aux0/aux1 = A/B half-level tables (len+1 values each);
                              value = ph(k) + 0.5*(ph(k+1)-ph(k)), ph(h) = A[h] + B[h]*sp[i % inner],
                              k = i / inner  (vertical/array/vertical.py:670,708).  Last operand only. */

typedef struct ekm_operand {
  const void* data; /* device pointer (float* for _f32, double* for _f64)         */
  int32_t mode;     /* one of the five: EKM_FIELD / _SCALAR / _LEVEL_MAJOR / _LEVEL_MINOR / _HYBRID_FULL */
  int32_t nflat;    /* EKM_HYBRID_FULL: number of LEADING levels k with B[k] = B[k+1] = 0 (pure pressure
                       levels: the upper 53 of the 137 IFS levels), which then run at level-vector
                       speed in a launch of their own; 0 is always valid.  Else 0.            */
  uint64_t len;     /* vector length for the LEVEL modes, else ignored            */
  uint64_t inner;   /* points per level for EKM_LEVEL_MAJOR / EKM_HYBRID_FULL      */
  const void* aux0; /* EKM_HYBRID_FULL: A table (device), else NULL                */
  const void* aux1; /* EKM_HYBRID_FULL: B table (device), else NULL                */
} ekm_operand;

/* ---- ABI version ----
 * Bumped whenever an exported symbol is removed or changes its signature or meaning (history: INTEGRATION.md).  A client
 * built against this header compares EKM_ABI_VERSION with ekm_abi_version() of the library it loaded before any
 * other call (ekm_hip/_ffi.py does); a library older than version 5 does not export the function at all.
 *   5  (round 5) ekm_abi_version added.  Since round 3 (unnumbered "4"): ekm_copy_staged, ekm_host_memcpy,
 *      ekm_host_register, ekm_host_unregister removed; "table_tiles" default 8 -> 0 (= by op); field pointers must be
 *      aligned to their element size; at most 32 KiB (was 64) of staged level vectors per launch. */
#define EKM_ABI_VERSION 5
EKM_API int ekm_abi_version(void);              /* EKM_ABI_VERSION of the library as built */

/* ---- lifecycle ---- */
EKM_API int ekm_init(void);                     /* probes the HIP runtime; EKM_ERR_NODEV without a GPU */
EKM_API int ekm_device_count(void);             /* >= 0, or a negative error code */
EKM_API const char* ekm_last_error(void);       /* thread-local, never NULL */
EKM_API const char* ekm_version(void);
EKM_API int ekm_device_name(int dev, char* buf, size_t buflen);
EKM_API int ekm_device_cus(int dev);            /* compute units, or a negative error code */
EKM_API int ekm_mem_info(int dev, size_t* free_bytes, size_t* total_bytes);

/* ---- device memory, streams, events ---- */
EKM_API int ekm_malloc(int dev, size_t bytes, void** out);
EKM_API int ekm_free(int dev, void* ptr);
EKM_API int ekm_host_alloc(size_t bytes, void** out);  /* pinned host memory for fast transfers */
EKM_API int ekm_host_free(void* ptr);
EKM_API int ekm_host_prefault(void* ptr, size_t bytes, int nthreads); /* touch the pages of a fresh host buffer on `nthreads` threads */
EKM_API int ekm_h2d(int dev, void* dst, const void* src, size_t bytes, void* stream);
EKM_API int ekm_d2h(int dev, void* dst, const void* src, size_t bytes, void* stream);
EKM_API int ekm_d2d(int dev, void* dst, const void* src, size_t bytes, void* stream);
EKM_API int ekm_memset(int dev, void* dst, int value, size_t bytes, void* stream);
EKM_API int ekm_fill_u32(int dev, void* dst, uint32_t value, size_t count, void* stream); /* `count` 32-bit words = value (async; a scalar operand's bit pattern without a host buffer) */
EKM_API int ekm_sync(int dev);                  /* hipDeviceSynchronize */
EKM_API int ekm_stream_create(int dev, void** out);
EKM_API int ekm_stream_destroy(int dev, void* stream);
EKM_API int ekm_stream_sync(int dev, void* stream);
EKM_API int ekm_event_create(int dev, void** out);
EKM_API int ekm_event_destroy(int dev, void* event);
EKM_API int ekm_event_record(int dev, void* event, void* stream);
EKM_API int ekm_stream_wait_event(int dev, void* stream, void* event); /* later work on `stream` waits for `event` (no host wait) */
EKM_API int ekm_event_sync(int dev, void* event);
EKM_API int ekm_event_elapsed_ms(int dev, void* start, void* stop, float* ms);
/* HIP graphs: between ekm_graph_begin and ekm_graph_end the launches made on `stream` (from ekm_stream_create; not the
 * default stream) are RECORDED, not run -- every compute entry point only enqueues kernels, so any sequence of them
 * can be recorded; ekm_malloc is allowed meanwhile, copies from pageable host memory and waits are not.
 * ekm_graph_end always ends the recording; it returns the executable graph in *graph_exec (NULL pointer: drop the
 * recording).  ekm_graph_launch replays it on a stream: same kernels, same pointers, one call. */
EKM_API int ekm_graph_begin(int dev, void* stream);
EKM_API int ekm_graph_end(int dev, void* stream, void** graph_exec);
EKM_API int ekm_graph_launch(int dev, void* graph_exec, void* stream);
EKM_API int ekm_graph_destroy(int dev, void* graph_exec);

/* ---- launch tuning (process-wide; defaults are the measured best) ---- */
/* tiles_per_block: consecutive 4-KiB tiles (256 lanes x 16 B) one workgroup streams per field;
 * unroll: tiles in flight per lane per trip (1 or 2; the bisection functions' tree-walk kernels always take 1).  0 keeps a value. */
EKM_API int ekm_set_tuning(int tiles_per_block, int unroll);
EKM_API int ekm_get_tuning(int* tiles_per_block, int* unroll);
/* secondary parameters by name (defaults from the environment, in brackets):
 *   "hybrid_band_kb" [EKM_HYBRID_BAND_KB, 8192]  EKM_HYBRID_FULL: KiB of surface pressure per L2-resident band;
 *   "lev_per_wg"     [EKM_LEV_PER_WG, 0]         EKM_HYBRID_FULL, one-in one-out functions only (theta ...): consecutive levels one workgroup walks (0 = 4); no effect on functions with more streams or on the bisection functions;
 *   "table_tiles"    [EKM_TABLE_TILES, 0]        most tiles per workgroup for ops that keep an LDS table (bisection); 0 = by op (8 or 16);
 *   "geo_chunk_levels" [EKM_GEO_CHUNK_LEVELS, all] levels per launch of the geopotential column scan;
 *   "hybrid_rows"      [EKM_HYBRID_ROWS, 1]        pressure_on_hybrid_levels: 1 = one workgroup per (level, tile), every output row
 *                      written in order like a map kernel's stream; 0 = one lane per column walking down the levels; same results;
 *   "f64_plain"        [EKM_F64_PLAIN, 0]          1: the fp64 map kernels redo EVERY lane with the plain-double primitives
 *                      (IEEE special operands fixed up as libm does) instead of only the lanes whose fast first pass
 *                      produced a non-finite output; same results, for tests and A/B timing;
 *   "bisect_exact"     [EKM_BISECT_EXACT, 0]       1: the bisection functions (every method, fp32 and fp64) evaluate the reference's residual (rcp + exp2) at
 *                      EVERY step of their tree walk instead of only where the transcendental-free sign test is within
 *                      rounding of zero; same results, for tests and A/B timing. */
EKM_API int ekm_set_tuning_param(const char* name, int value);
/* The table-driven functions (wet-bulb / moist-adiabat inversion by bisection) read a lookup table that is computed on
 * the device at their first launch there -- asynchronously: no entry point of this library waits on the host.  A
 * launch captured into a HIP graph before the table exists records the fill into that graph.  ekm_prepare_tables(dev)
 * computes every table on `dev` now and WAITS for it (the one synchronising call; not while capturing), so that later
 * launches and captures find them ready. */
EKM_API int ekm_prepare_tables(int dev);

/* ---- synthetic benchmark input, generated on the device (SURVEY.md 8d) ----
 * Fills t, q (and p unless NULL) for points [first, first+n) of a level-major
 * [nlev, inner] grid with the benchmark distribution; counter-based, so any
 * shard of the global index range can be generated independently. */
EKM_API int ekm_synth_fill_f32(int dev, void* stream, float* t, float* q, float* p, uint64_t first, size_t n,
                               uint64_t inner, uint32_t nlev, uint64_t seed);
EKM_API int ekm_synth_fill_f64(int dev, void* stream, double* t, double* q, double* p, uint64_t first, size_t n,
                               uint64_t inner, uint32_t nlev, uint64_t seed);
/* same distribution of t, q around a pressure field that is already on the device (e.g. hybrid levels) */
EKM_API int ekm_synth_fill_given_p_f32(int dev, void* stream, float* t, float* q, const float* p, uint64_t first,
                                       size_t n, uint64_t seed);
EKM_API int ekm_synth_fill_given_p_f64(int dev, void* stream, double* t, double* q, const double* p, uint64_t first,
                                       size_t n, uint64_t seed);
/* The streaming reference of a launch (a benchmark helper like the two above): reads `nin` (0..3) and writes `nout` (1, 2, 3
 * or 6) streams of `bytes` bytes each with the map kernels' launch shape and one add per stream -- what the memory system
 * gives this set of buffers; bench.py times it on the arrays of the launch it has just measured (roofline.stream_ceiling).
 * The outputs are OVERWRITTEN. */
EKM_API int ekm_stream_mix(int dev, void* stream, const void* const* ins, int nin, void* const* outs, int nout, size_t bytes);
EKM_API int ekm_synth_levels_f32(int dev, void* stream, float* p_levels, uint32_t nlev);
EKM_API int ekm_synth_levels_f64(int dev, void* stream, double* p_levels, uint32_t nlev);

/* ---- hybrid model levels: the producer of the pressure field (SURVEY.md 8f rank 1) ----
 * pressure_on_hybrid_levels: reference vertical/array/vertical.py:505-740.
 * A, B: nfull+1 half-level coefficients of the (contiguous) level range, on the device;
 * sp: npts surface pressures; outputs are [rows, npts] level-major, any may be NULL.
 * row_full[k] / row_half[h] give the output row of layer k / half level h, or -1 to skip it
 * (NULL = identity): this is how the reference's `levels=` selection and ordering is expressed.
 * top_is_zero: the reference's any(p_half[0] <= 0.1) (see ekm_any_le_*); alpha_top: log(2) or 1. */
EKM_API int ekm_pressure_on_hybrid_levels_f32(int dev, void* stream, const float* A, const float* B, const float* sp,
                                              size_t npts, uint32_t nfull, const int32_t* row_full,
                                              const int32_t* row_half, int top_is_zero, float alpha_top, float* full,
                                              float* half, float* delta, float* alpha);
EKM_API int ekm_pressure_on_hybrid_levels_f64(int dev, void* stream, const double* A, const double* B, const double* sp,
                                              size_t npts, uint32_t nfull, const int32_t* row_full,
                                              const int32_t* row_half, int top_is_zero, double alpha_top, double* full,
                                              double* half, double* delta, double* alpha);
/* Geopotential chain on hybrid levels, fused with the producer of alpha/delta: reference
 * vertical/array/vertical.py:741-1190 (relative_geopotential_thickness_on_hybrid_levels,
 * geopotential_on_hybrid_levels, height_on_hybrid_levels).  t, q, out: [nfull, npts] level-major;
 * A, B: the nfull+1 half-level coefficients of those levels; zs: surface geopotential (may be NULL
 * for modes 0 and 5).  mode: 0 thickness, 1 geopotential (+zs), 2 geometric height above sea,
 * 3 geopotential height above sea, 4 geometric height above ground, 5 geopotential height above ground. */
EKM_API int ekm_geopotential_on_hybrid_levels_f32(int dev, void* stream, const float* A, const float* B,
                                                  const float* sp, const float* zs, const float* t, const float* q,
                                                  size_t npts, uint32_t nfull, int top_is_zero, float alpha_top,
                                                  int mode, float* out);
EKM_API int ekm_geopotential_on_hybrid_levels_f64(int dev, void* stream, const double* A, const double* B,
                                                  const double* sp, const double* zs, const double* t, const double* q,
                                                  size_t npts, uint32_t nfull, int top_is_zero, double alpha_top,
                                                  int mode, double* out);
/* The same bottom-up scan for callers who already hold alpha and delta (outputs of pressure_on_hybrid_levels):
 * reference vertical/array/vertical.py:741-893 (relative_geopotential_thickness_on_hybrid_levels_from_alpha_delta).
 * t, q, alpha, delta, out: [nfull, npts] level-major; 16 B read + 4 B written per point (fp32). */
EKM_API int ekm_geopotential_thickness_from_alpha_delta_f32(int dev, void* stream, const float* t, const float* q,
                                                            const float* alpha, const float* delta, size_t npts,
                                                            uint32_t nfull, float* out);
EKM_API int ekm_geopotential_thickness_from_alpha_delta_f64(int dev, void* stream, const double* t, const double* q,
                                                            const double* alpha, const double* delta, size_t npts,
                                                            uint32_t nfull, double* out);
/* *flag |= any(a0 + b0*sp[i] <= thresh); *flag must be zeroed by the caller (ekm_memset) */
EKM_API int ekm_any_le_f32(int dev, void* stream, const float* sp, size_t n, float a0, float b0, float thresh,
                           int32_t* flag);
EKM_API int ekm_any_le_f64(int dev, void* stream, const double* sp, size_t n, double a0, double b0, double thresh,
                           int32_t* flag);

/* ---- thermo entry points ----
 * Argument order: dev, stream, inputs..., enum parameters..., [eps], outputs..., n. */
'''

HEADER_BOTTOM = r'''
#ifdef __cplusplus
}
#endif
#endif /* EKM_THERMO_H */
'''


def proto(name, ins, outs, ints, has_eps, tag, ctype, prefix="ekm_"):
    args = ["int dev", "void* stream"]
    args += [f"const ekm_operand* {i}" for i in ins]
    args += [f"int {p[0]}" for p in ints]
    if has_eps:
        args.append(f"{ctype} eps")
    args += [f"{ctype}* {o}" for o in outs]
    args.append("size_t n")
    return f"int {prefix}{name}_{tag}({', '.join(args)})"


def cite(ref):
    return ref if ("SURVEY" in ref or ".py:" in ref.split(";")[0].split(",")[0] and not ref[0].isdigit()) else f"thermo/array/thermo.py:{ref}"


def gen_header():
    out = [HEADER_TOP]
    for name, functor, ins, outs, ints, has_eps, ref, group in OPS:
        enums = "; ".join(f"{p[0]}: {p[1]}" for p in ints)
        out.append(f"/* {name}: reference {cite(ref)}" + (f" [{enums}]" if enums else "") + " */")
        for tag, ctype in DTYPES:
            out.append("EKM_API " + proto(name, ins, outs, ints, has_eps, tag, ctype) + ";")
        out.append("")
    out.append(HEADER_BOTTOM)
    return "\n".join(out)


def dispatch(functor, ints, call):
    """Nested switch over the enum parameters -> template arguments."""
    if not ints:
        return f"  return {call(functor)};\n"
    if len(ints) == 1:
        (pn, _, cnt), = ints
        s = f"  switch ({pn}) {{\n"
        for v in range(cnt):
            s += f"    case {v}: return {call(f'{functor}<{v}>')};\n"
        s += f'    default: return ekm::set_error(EKM_ERR_ENUM, "{pn}=%d is not a valid value", {pn});\n  }}\n'
        return s
    (p1, _, c1), (p2, _, c2) = ints
    s = f"  switch ({p1} * 8 + {p2}) {{\n"
    for a in range(c1):
        for b in range(c2):
            s += f"    case {a * 8 + b}: return {call(f'{functor}<{a}, {b}>')};\n"
    s += (f'    default: return ekm::set_error(EKM_ERR_ENUM, "{p1}=%d / {p2}=%d is not a valid combination", '
          f"{p1}, {p2});\n  }}\n")
    return s


def gen_entries(group, dtype):
    out = ["// GENERATED by tools/gen_abi.py -- do not edit.\n", '#include "../map_kernel.hpp"\n\n', 'extern "C" {\n\n']
    for name, functor, ins, outs, ints, has_eps, ref, g in OPS:
        if g != group:
            continue
        for tag, ctype in (dtype,):
            out.append(proto(name, ins, outs, ints, has_eps, tag, ctype) + " {\n")
            out.append(f"  const ekm_operand* ins_[] = {{{', '.join(ins)}}};\n")
            out.append(f"  void* outs_[] = {{{', '.join(outs)}}};\n")
            rp = "eps" if has_eps else "0.0"
            if has_eps:
                out.append(f'  if (!(eps > 0)) return ekm::set_error(EKM_ERR_ARG, "{name}(): eps=%g must be > 0", (double)eps);\n')
            out.append(dispatch(functor, ints, lambda f: f"ekm::launch_map<ekm::{f}, {ctype}>(dev, stream, ins_, outs_, n, {rp})"))
            out.append("}\n\n")
    out.append('}  // extern "C"\n')
    return "".join(out)


def gen_host_entries():
    out = ["// GENERATED by tools/gen_abi.py -- do not edit.  Host twin (test infrastructure).\n\n"]
    for name, functor, ins, outs, ints, has_eps, ref, g in OPS:
        for tag, ctype in DTYPES:
            args = [f"const {ctype}* {i}" for i in ins] + [f"int {p[0]}" for p in ints]
            if has_eps:
                args.append(f"{ctype} eps")
            args += [f"{ctype}* {o}" for o in outs] + ["size_t n"]
            out.append(f'extern "C" int ekm_host_{name}_{tag}({", ".join(args)}) {{\n')
            out.append(f"  const {ctype}* ins_[] = {{{', '.join(ins)}}};\n")
            out.append(f"  {ctype}* outs_[] = {{{', '.join(outs)}}};\n")
            rp = "eps" if has_eps else "0.0"
            out.append(dispatch(functor, ints, lambda f: f"host_map<ekm::{f}, {ctype}>(ins_, outs_, n, {rp})").replace(
                "ekm::set_error(EKM_ERR_ENUM, ", "host_bad_enum(").replace("EKM_ERR_ENUM", "-3"))
            out.append("}\n\n")
    return "".join(out)


def gen_optable():
    out = ['"""GENERATED by tools/gen_abi.py -- ctypes signature table of libekm_thermo.so."""\n\n',
           "# name: (inputs, outputs, int parameters, has_eps)\nOPS = {\n"]
    for name, functor, ins, outs, ints, has_eps, ref, g in OPS:
        out.append(f"    {name!r}: ({tuple(ins)!r}, {tuple(outs)!r}, {tuple(p[0] for p in ints)!r}, {has_eps}),\n")
    out.append("}\n")
    return "".join(out)


def write(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(text)


def main():
    write(os.path.join(ROOT, "include", "ekm_thermo.h"), gen_header())
    gen = os.path.join(ROOT, "earthkit-meteo_amd", "csrc", "gen")
    for group in sorted({o[7] for o in OPS}):
        for dtype in DTYPES:
            write(os.path.join(gen, f"entries_{group}_{dtype[0]}.hip"), gen_entries(group, dtype))
    write(os.path.join(gen, "host_entries.inc"), gen_host_entries())
    write(os.path.join(ROOT, "earthkit-meteo_amd", "ekm_hip", "_optable.py"), gen_optable())
    print("generated", len(OPS), "ops x", len(DTYPES), "dtypes")


if __name__ == "__main__":
    main()
