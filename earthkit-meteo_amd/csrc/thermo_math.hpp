// Per-grid-point thermodynamics, templated on the real type (float / double).
//
// One statement of the arithmetic for both the gfx950 kernels (gen/entries_*.hip)
// and the host test twin (host_twin.cpp, test infrastructure only).  Every
// function cites the reference lines it restates:
//   thermo.py:N  = /root/reference/src/earthkit/meteo/thermo/array/thermo.py:N
//   es_comp.py:N = /root/reference/src/earthkit/meteo/thermo/array/es_comp.py:N
//
// fp32 on the device goes through the CDNA4 transcendental unit directly
// (v_exp_f32 / v_log_f32 / v_rcp_f32 via __builtin_amdgcn_*): no libm calls,
// no IEEE division expansion.  NaN / inf semantics of the reference are kept
// (no -ffast-math): rcp(0)=inf, log2(0)=-inf, log2(<0)=NaN, 0*inf=NaN.
// fp64 has no hardware transcendentals: rcp / exp2 / log2 are built here from v_rcp_f64, v_ldexp_f64, v_frexp_* and
// short polynomials (below); the device libm (ocml) is used only by the -DEKM_F64_LIBM diagnostic build.
#pragma once

#include <cmath>
#include <limits>

#if defined(__HIPCC__)
#define EKM_HD __host__ __device__ __forceinline__
#else
#define EKM_HD inline
#endif

namespace ekm {

// ---- constants (constants/constants.py:22-50, es_comp.py:14-20) -----------
// Formed in double exactly as the Python literals/expressions are, then
// rounded to the array dtype where they meet the data (NEP-50 weak scalars).
namespace k {
constexpr double Rd = 287.0597;
constexpr double Rv = 461.51;
constexpr double g = 9.80665;  // constants/constants.py:53
constexpr double c_pd = 1004.79;
constexpr double Lv = 2.5008e6;
constexpr double kappa = 0.285691;
constexpr double p0 = 1e5;
constexpr double eps = 0.621981;
constexpr double T0 = 273.16;
constexpr double C1 = 611.21;
constexpr double C3W = 17.502;
constexpr double C4W = 32.19;
constexpr double C3I = 22.587;
constexpr double C4I = -0.7;
constexpr double TI = T0 - 23;
constexpr double lambda = 1.0 / kappa;            // thermo.py:1022
constexpr double K0_ifs = Lv / c_pd;              // thermo.py:1164
constexpr double q_c = eps * (1.0 / eps - 1.0);   // thermo.py:130
constexpr double tv_c1 = (1.0 - eps) / eps;       // thermo.py:763
constexpr double sw = C3W * (T0 - C4W);           // es_comp.py:170
constexpr double si = C3I * (T0 - C4I);           // es_comp.py:174
constexpr double dalpha_c = 2.0 / ((T0 - TI) * (T0 - TI));  // es_comp.py:193
constexpr double eps_default = 1e-4;
}  // namespace k

enum Phase { PHASE_MIXED = 0, PHASE_WATER = 1, PHASE_ICE = 2 };
enum EptMethod { EPT_IFS = 0, EPT_BOLTON35 = 1, EPT_BOLTON39 = 2 };
enum TMethod { T_BISECT = 0, T_NEWTON = 1, T_DIRECT = 2 };
enum LclMethod { LCL_DAVIES = 0, LCL_BOLTON = 1 };

// ---- math primitives -------------------------------------------------------
template <class T>
EKM_HD T nan_v() {
  return std::numeric_limits<T>::quiet_NaN();
}

// ---- fdouble: the fp64 kernels' FAST arithmetic type ----------------------------------------------------------
// A double whose three primitives (m_rcp / m_exp2 / m_log2 below) carry NO special-operand fix-ups: for an operand
// where the fixed-up primitive returns one of the IEEE specials (rcp(0) = inf, rcp(inf) = 0, exp2(+-inf), log2(0),
// log2(inf)) they return NaN instead ("poison"), for every other operand exactly what the plain-double primitive
// returns.  Everything downstream of a poisoned value is NaN (no formula of this file takes fmin/fmax of a
// primitive's result or selects on a comparison alone without carrying the compared value), so a point whose fast
// outputs are all finite got exactly the plain-double result, and the map kernels redo the lanes with a non-finite
// output in plain double (map_kernel.hpp::apply_points).  What that buys: the static instruction stream of the
// six-output fp64 pipeline held 121 v_cndmask + 65 v_cmp + 30 v_min/v_max per point for the fix-ups, a quarter of
// its issue time, for operands atmospheric data never has.  The host twin runs the same two passes with libm-based
// stand-ins of the poisoning primitives, so the golden edge cases check the re-run logic without a GPU.
// xdouble is the same wrapper around the PLAIN primitives: the kernels' second pass and their element-wise paths run it
// instead of a bare double, so that both passes are the same template instantiated over the same operator functions and
// the compiler contracts a*b+c into fma at the same places -- the redo must reproduce the first pass bit for bit
// wherever that was valid (tests/test_gpu_two_pass.py).
EKM_HD double nc_add(double a, double b) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  return a + b;
}
EKM_HD double nc_mul(double a, double b) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  return a * b;
}
template <bool FAST>
struct fd64 {
  double v;
  fd64() = default;
  EKM_HD constexpr fd64(double x) : v(x) {}
  EKM_HD explicit operator double() const { return v; }
  EKM_HD explicit operator float() const { return (float)v; }
  EKM_HD explicit operator int() const { return (int)v; }
  // hidden friends: found by argument-dependent lookup, so a bare constant on one side converts.
  // The arithmetic operators do NOT contract (a product and a sum written with them stay two roundings): which pairs
  // the compiler fuses otherwise differs between two instantiations of the same formula, and the two passes must round
  // alike.  Where a formula wants a fused multiply-add it says so: m_fma / m_fms / m_fnma below.
  EKM_HD friend fd64 operator+(fd64 a, fd64 b) { return fd64(nc_add(a.v, b.v)); }
  EKM_HD friend fd64 operator-(fd64 a, fd64 b) { return fd64(nc_add(a.v, -b.v)); }
  EKM_HD friend fd64 operator*(fd64 a, fd64 b) { return fd64(nc_mul(a.v, b.v)); }
  EKM_HD friend fd64 operator-(fd64 a) { return fd64(-a.v); }
  EKM_HD friend fd64& operator+=(fd64& a, fd64 b) { a.v = nc_add(a.v, b.v); return a; }
  EKM_HD friend fd64& operator-=(fd64& a, fd64 b) { a.v = nc_add(a.v, -b.v); return a; }
  EKM_HD friend fd64& operator*=(fd64& a, fd64 b) { a.v = nc_mul(a.v, b.v); return a; }
  EKM_HD friend bool operator<(fd64 a, fd64 b) { return a.v < b.v; }
  EKM_HD friend bool operator<=(fd64 a, fd64 b) { return a.v <= b.v; }
  EKM_HD friend bool operator>(fd64 a, fd64 b) { return a.v > b.v; }
  EKM_HD friend bool operator>=(fd64 a, fd64 b) { return a.v >= b.v; }
  EKM_HD friend bool operator==(fd64 a, fd64 b) { return a.v == b.v; }
  EKM_HD friend bool operator!=(fd64 a, fd64 b) { return a.v != b.v; }
};
typedef fd64<true> fdouble;
typedef fd64<false> xdouble;
#define EKM_FD template <bool F> EKM_HD
template <>
EKM_HD fdouble nan_v<fdouble>() {
  return fdouble(std::numeric_limits<double>::quiet_NaN());
}
template <>
EKM_HD xdouble nan_v<xdouble>() {
  return xdouble(std::numeric_limits<double>::quiet_NaN());
}

#if defined(__HIP_DEVICE_COMPILE__)
EKM_HD float m_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
EKM_HD float m_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
EKM_HD float m_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
EKM_HD float m_log2(float x) { return __builtin_amdgcn_logf(x); }  // v_log_f32 is log2
EKM_HD float m_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
EKM_HD float m_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
EKM_HD float m_pow(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }
#else
EKM_HD float m_rcp(float x) { return 1.0f / x; }
EKM_HD float m_div(float a, float b) { return a / b; }
EKM_HD float m_exp2(float x) { return std::exp2(x); }
EKM_HD float m_log2(float x) { return std::log2(x); }
EKM_HD float m_exp(float x) { return std::exp(x); }
EKM_HD float m_log(float x) { return std::log(x); }
EKM_HD float m_pow(float x, float y) { return std::pow(x, y); }
#endif
#if defined(__HIP_DEVICE_COMPILE__) && !defined(EKM_F64_LIBM)
// fp64 on gfx950: there is no fp64 transcendental unit and the device libm pays for correctly rounded
// results (pow alone is ~200 instructions).  Measured issue cost (tools/microbench/valu_rates_f64.hip):
// v_fma_f64 5.2 clk per wave, v_rcp_f64 17 clk.  The parity bar for fp64 is 1e-6 relative, so by default
// the three primitives are built to ~1e-10 (four orders inside the bar; the GPU tests assert <= 1e-9 against the
// reference's fp64 goldens):
//   rcp  = v_rcp_f64 seed + ONE Newton step (<= 1e-14);
//   exp2 = round-to-nearest split + degree-7 near-minimax polynomial of 2^f, ln 2 folded in + v_ldexp_f64 (4.0e-11);
//   log2 = v_frexp + 2*atanh(s), s = (m-1)/(m+1), as s*q(s^2) with a degree-4 near-minimax q (4.2e-12);
//   pow  = exp2(y*log2(x)).
// -DEKM_F64_EXACT selects the <= 3e-16 versions (two Newton steps, degree-12 Taylor, atanh series to s^21)
// for A/B comparison.  inf / 0 / NaN behave as in libm in both.
#if defined(EKM_F64_EXACT)
EKM_HD double m_rcp(double x) {
  const double r0 = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r0, 1.0);
  double r = __builtin_fma(r0, e, r0);
  e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  return __builtin_isfinite(r) ? r : r0;  // x = 0, inf, NaN: keep the hardware answer (inf, 0, NaN)
}
EKM_HD double m_div(double a, double b) { return a * m_rcp(b); }
EKM_HD double m_exp2(double x) {
  const double xc = __builtin_fmin(__builtin_fmax(x, -1100.0), 1100.0);
  const double n = __builtin_rint(xc);
  const double y = (xc - n) * 0.69314718055994530942;  // |y| <= 0.3466
  double p = 1.0 / 479001600.0;                          // Taylor of e^y, degree 12
  p = __builtin_fma(p, y, 1.0 / 39916800.0);
  p = __builtin_fma(p, y, 1.0 / 3628800.0);
  p = __builtin_fma(p, y, 1.0 / 362880.0);
  p = __builtin_fma(p, y, 1.0 / 40320.0);
  p = __builtin_fma(p, y, 1.0 / 5040.0);
  p = __builtin_fma(p, y, 1.0 / 720.0);
  p = __builtin_fma(p, y, 1.0 / 120.0);
  p = __builtin_fma(p, y, 1.0 / 24.0);
  p = __builtin_fma(p, y, 1.0 / 6.0);
  p = __builtin_fma(p, y, 0.5);
  p = __builtin_fma(p, y, 1.0);
  p = __builtin_fma(p, y, 1.0);
  const double r = __builtin_amdgcn_ldexp(p, (int)n);
  return x != x ? x : r;
}
EKM_HD double m_log2(double x) {
  int e = __builtin_amdgcn_frexp_exp(x);
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  if (m < 0.70710678118654752440) {
    m *= 2.0;
    e -= 1;
  }
  const double s = (m - 1.0) * m_rcp(m + 1.0);  // |s| <= 0.1716
  const double z = s * s;
  double p = 1.0 / 21.0;                           // atanh series: ln(m) = 2s(1 + z/3 + z^2/5 + ...)
  p = __builtin_fma(p, z, 1.0 / 19.0);
  p = __builtin_fma(p, z, 1.0 / 17.0);
  p = __builtin_fma(p, z, 1.0 / 15.0);
  p = __builtin_fma(p, z, 1.0 / 13.0);
  p = __builtin_fma(p, z, 1.0 / 11.0);
  p = __builtin_fma(p, z, 1.0 / 9.0);
  p = __builtin_fma(p, z, 1.0 / 7.0);
  p = __builtin_fma(p, z, 1.0 / 5.0);
  p = __builtin_fma(p, z, 1.0 / 3.0);
  p = __builtin_fma(p, z, 1.0);
  double r = __builtin_fma(p * s, 2.0 * 1.44269504088896340736, (double)e);
  if (x == 0.0) r = -__builtin_inf();
  if (x == __builtin_inf()) r = x;
  if (x < 0.0 || x != x) r = __builtin_nan("");
  return r;
}
#else
// Shared pieces of the default (~1e-10) set.  The polynomial coefficients live in constant memory, NOT in the
// instruction stream: v_fma_f64 cannot take a 64-bit literal, so with literal coefficients the compiler materialises
// each one with two v_mov_b32 in front of a v_fmac_f64 (122 v_mov per point in the six-output pipeline, ~10 % of its
// issue time; profiles/r03).  Read through the scalar cache they are SGPR pairs, loaded once per wave, that
// v_fma_f64 takes directly as its addend.  (static: one copy per translation unit / device module.)
// Round 5 measured one degree less in each polynomial (-DEKM_F64_R5_POLY: exp2 degree 6, 1.9e-9; atanh degree 3, 6.9e-10):
// the six-output pipeline 14.5 -> 14.2 ms, the Newton wet-bulb 10.2 -> 10.0 -- and row 470 of the reference's own 480-row
// table beyond the bar (bolton35's one Newton step, whose dlnf cancels to 1e-5 of its terms there, amplifies a primitive's
// error a thousandfold: 1.8e-6 against 1e-6).  Two per cent are not worth a miss on the reference's own test data: the
// default keeps round 4's degree 7 / 4 (4.0e-11 / 4.2e-12).  Nor can the reciprocal drop its Newton step: raw v_rcp_f64 is
// 4.6e-8 (tools/microbench/f64_seed_accuracy.hip) and feeds exponents of up to 40 in es.
#if !defined(EKM_F64_R5_POLY)
static __constant__ double kF64Coef[16] = {
    // 2^f on |f| <= 0.5, degree 7, near-minimax in relative error (4.0e-11), ln 2 folded in; ascending
    0.9999999999616818, 0.693147180728452, 0.24022651198156714, 0.05550410353429554, 0.009618027253757476,
    0.0013333922578355431, 0.00015469291117256424, 1.5201918192496034e-05,
    // atanh(s)/s on z = s^2 in [0, 0.0295], degree 4 (4.2e-12); ascending
    1.0000000000041798, 0.3333333262373743, 0.20000192337193154, 0.14267525468490147, 0.1180818033212343,
    0.0, 0.0, 0.0};
#else
static __constant__ double kF64Coef[16] = {
    // 2^f on |f| <= 0.5, degree 6, minimax in relative error (1.86e-9), ln 2 folded in; ascending
    1.0000000005541665, 0.6931472057372673, 0.2402264689063404, 0.05550328776965614, 0.00961848895713086,
    0.0013399931219141663, 0.0001534581199765244, 0.0,
    // atanh(s)/s on z = s^2 in [0, 0.0295], degree 3, minimax in relative error (6.9e-10); ascending
    0.9999999993106649, 0.3333340797542721, 0.19987397462507944, 0.14962825347475195, 0.0,
    0.0, 0.0, 0.0};
#endif
#if defined(EKM_F64_COEF_LITERAL)
#error "EKM_F64_COEF_LITERAL was a round-3 A/B switch; the coefficients live in constant memory"
#endif
#define EKM_F64C(i) (kF64Coef[i])
EKM_HD double exp2_poly(double f) {  // 2^f, |f| <= 0.5
#if !defined(EKM_F64_R5_POLY)
  double p = EKM_F64C(7);
  p = __builtin_fma(p, f, EKM_F64C(6));
#else
  double p = EKM_F64C(6);
#endif
  p = __builtin_fma(p, f, EKM_F64C(5));
  p = __builtin_fma(p, f, EKM_F64C(4));
  p = __builtin_fma(p, f, EKM_F64C(3));
  p = __builtin_fma(p, f, EKM_F64C(2));
  p = __builtin_fma(p, f, EKM_F64C(1));
  p = __builtin_fma(p, f, EKM_F64C(0));
  return p;
}
EKM_HD double atanh_poly(double z) {  // atanh(s)/s, z = s^2
#if !defined(EKM_F64_R5_POLY)
  double p = EKM_F64C(12);
  p = __builtin_fma(p, z, EKM_F64C(11));
#else
  double p = EKM_F64C(11);
#endif
  p = __builtin_fma(p, z, EKM_F64C(10));
  p = __builtin_fma(p, z, EKM_F64C(9));
  p = __builtin_fma(p, z, EKM_F64C(8));
  return p;
}
// v_cvt_i32_f64 saturates (+-2^31, NaN -> 0); the C cast is undefined out of range
EKM_HD int cvt_sat_i32(double n) {
  int i;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(i) : "v"(n));
  return i;
}
// 1/a for a in [1, 4] (the log2 argument): v_rcp_f32 seed + one Newton step (<= 3e-14), ~21 clocks instead of ~28
EKM_HD double rcp_small(double a) {
  const double r0 = (double)__builtin_amdgcn_rcpf((float)a);
  return __builtin_fma(r0, __builtin_fma(-a, r0, 1.0), r0);
}
// log2 of a positive finite x: x = m * 2^e with m in [0.7071, 1.4142) (the exponent of x*sqrt(2) puts the split at
// sqrt(1/2)), 2*atanh(s), s = (m - 1)/(m + 1), |s| <= 0.1716
EKM_HD double log2_core(double x) {
  const int e = __builtin_amdgcn_frexp_exp(x * 1.41421356237309504880) - 1;
  const double m = __builtin_amdgcn_ldexp(x, -e);
  const double s = (m - 1.0) * rcp_small(m + 1.0);
  return __builtin_fma(atanh_poly(s * s) * s, 2.0 * 1.44269504088896340736, (double)e);
}

// plain double: IEEE special operands handled as libm does.  Off the hot path since round 4 (the map kernels run
// fdouble and come here only for the lanes it poisoned, for ragged ends and for unaligned input).
EKM_HD double m_rcp(double x) {
  const double r0 = __builtin_amdgcn_rcp(x);
  const double e = __builtin_fma(-x, r0, 1.0);
  const double r = __builtin_fma(r0, e, r0);
  return __builtin_isfinite(r) ? r : r0;  // x = 0, inf, NaN: keep the hardware answer (inf, 0, NaN)
}
EKM_HD double m_div(double a, double b) { return a * m_rcp(b); }
// x - rint(x), exact for every finite x, |f| <= 0.5.  Contraction is switched off for this one subtraction: where x is a
// product the compiler may otherwise fuse it into fma(a, b, -n) in one instantiation and not in another, and the two
// passes of the kernels must round alike (tests/test_gpu_two_pass.py).
EKM_HD double exp2_frac(double x, double n) {
#pragma clang fp contract(off)
  return x - n;
}
EKM_HD double m_exp2(double x) {
  const double n = __builtin_rint(x);
  double f = exp2_frac(x, n);         // NaN for +-inf
  if (__builtin_isinf(x)) f = 0.0;    // 2^(+-inf) = inf / 0 through the saturated exponent
  return __builtin_amdgcn_ldexp(exp2_poly(f), cvt_sat_i32(n));  // NaN stays NaN
}
EKM_HD double m_log2(double x) {
  double r = log2_core(x);
  if (x == 0.0) r = -__builtin_inf();
  if (x == __builtin_inf()) r = x;
  if (x < 0.0 || x != x) r = __builtin_nan("");
  return r;
}
#define EKM_HAVE_FDOUBLE_FAST 1
// fdouble: the same arithmetic, poison instead of fix-ups (see the type's comment above)
EKM_HD fdouble m_rcp(fdouble x) {
  const double r0 = __builtin_amdgcn_rcp(x.v);
  return fdouble(__builtin_fma(r0, __builtin_fma(-x.v, r0, 1.0), r0));  // 0, inf, NaN, overflow: NaN
}
EKM_HD fdouble m_exp2(fdouble x) {
  const double n = __builtin_rint(x.v);
  return fdouble(__builtin_amdgcn_ldexp(exp2_poly(exp2_frac(x.v, n)), cvt_sat_i32(n)));  // +-inf: NaN
}
EKM_HD fdouble m_log2(fdouble x) {
  const double r = log2_core(x.v);
  // x <= 0, inf, NaN: poison through the high dword (one v_cmp_class_f64 + one v_cndmask_b32)
  const bool good = __builtin_amdgcn_class(x.v, 0x080 | 0x100);  // +denormal | +normal
  return fdouble(__hiloint2double(good ? __double2hiint(r) : 0x7ff80000, __double2loint(r)));
}
#endif
EKM_HD double m_exp(double x) { return m_exp2(x * 1.44269504088896340736); }
EKM_HD double m_log(double x) { return m_log2(x) * 0.69314718055994530942; }
EKM_HD double m_pow(double x, double y) { return m_exp2(y * m_log2(x)); }
#else
EKM_HD double m_rcp(double x) { return 1.0 / x; }
EKM_HD double m_div(double a, double b) { return a / b; }
EKM_HD double m_exp2(double x) { return exp2(x); }
EKM_HD double m_log2(double x) { return log2(x); }
EKM_HD double m_exp(double x) { return exp(x); }
EKM_HD double m_log(double x) { return log(x); }
EKM_HD double m_pow(double x, double y) { return pow(x, y); }
#endif

#if defined(EKM_HAVE_FDOUBLE_FAST)
EKM_HD fdouble m_div(fdouble a, fdouble b) { return a * m_rcp(b); }
EKM_HD fdouble m_exp(fdouble x) { return m_exp2(x * fdouble(1.44269504088896340736)); }
EKM_HD fdouble m_log(fdouble x) { return m_log2(x) * fdouble(0.69314718055994530942); }
EKM_HD fdouble m_pow(fdouble x, fdouble y) { return m_exp2(y * m_log2(x)); }
#else
// fdouble where the fast primitives do not exist (host twin; -DEKM_F64_EXACT / -DEKM_F64_LIBM device builds): the plain
// function of this build, poisoned under exactly the conditions under which the gfx950 composition above poisons, so
// that the two-pass logic is what the CPU tests exercise (tests/test_hosttwin_two_pass.py).
EKM_HD bool fd_rcp_ok(double x) { return x != 0.0 && __builtin_isfinite(x) && __builtin_isfinite(1.0 / x); }
EKM_HD bool fd_log_ok(double x) { return x > 0.0 && x < __builtin_inf(); }
EKM_HD fdouble fd_nan() { return fdouble(__builtin_nan("")); }
EKM_HD fdouble m_rcp(fdouble x) { return fd_rcp_ok(x.v) ? fdouble(m_rcp(x.v)) : fd_nan(); }
EKM_HD fdouble m_div(fdouble a, fdouble b) { return fd_rcp_ok(b.v) ? fdouble(m_div(a.v, b.v)) : fd_nan(); }
EKM_HD fdouble m_exp2(fdouble x) { return __builtin_isinf(x.v) ? fd_nan() : fdouble(m_exp2(x.v)); }
EKM_HD fdouble m_log2(fdouble x) { return fd_log_ok(x.v) ? fdouble(m_log2(x.v)) : fd_nan(); }
EKM_HD fdouble m_exp(fdouble x) { return __builtin_isinf(x.v * 1.44269504088896340736) ? fd_nan() : fdouble(m_exp(x.v)); }
EKM_HD fdouble m_log(fdouble x) { return fd_log_ok(x.v) ? fdouble(m_log(x.v)) : fd_nan(); }
EKM_HD fdouble m_pow(fdouble x, fdouble y) {
  return (fd_log_ok(x.v) && !__builtin_isinf(y.v * m_log2(x.v))) ? fdouble(m_pow(x.v, y.v)) : fd_nan();
}
#endif

// xdouble: the plain primitives behind the wrapper's operators
EKM_HD xdouble m_rcp(xdouble x) { return xdouble(m_rcp(x.v)); }
EKM_HD xdouble m_exp2(xdouble x) { return xdouble(m_exp2(x.v)); }
EKM_HD xdouble m_log2(xdouble x) { return xdouble(m_log2(x.v)); }
#if defined(EKM_HAVE_FDOUBLE_FAST)
EKM_HD xdouble m_div(xdouble a, xdouble b) { return a * m_rcp(b); }
EKM_HD xdouble m_exp(xdouble x) { return m_exp2(x * xdouble(1.44269504088896340736)); }
EKM_HD xdouble m_log(xdouble x) { return m_log2(x) * xdouble(0.69314718055994530942); }
EKM_HD xdouble m_pow(xdouble x, xdouble y) { return m_exp2(y * m_log2(x)); }
#else
EKM_HD xdouble m_div(xdouble a, xdouble b) { return xdouble(m_div(a.v, b.v)); }
EKM_HD xdouble m_exp(xdouble x) { return xdouble(m_exp(x.v)); }
EKM_HD xdouble m_log(xdouble x) { return xdouble(m_log(x.v)); }
EKM_HD xdouble m_pow(xdouble x, xdouble y) { return xdouble(m_pow(x.v, y.v)); }
#endif

template <class T>
EKM_HD T m_sq(T x) {
  return x * x;
}

// a*b + c, a*b - c, c - a*b.  float / double: the plain expression, which the compiler contracts where it sees fit (the
// fp32 kernels are what they were); fd64: ONE fused operation on the device, since its operators never contract.
EKM_HD float m_fma(float a, float b, float c) { return a * b + c; }
EKM_HD float m_fms(float a, float b, float c) { return a * b - c; }
EKM_HD float m_fnma(float a, float b, float c) { return c - a * b; }
EKM_HD double m_fma(double a, double b, double c) { return a * b + c; }
EKM_HD double m_fms(double a, double b, double c) { return a * b - c; }
EKM_HD double m_fnma(double a, double b, double c) { return c - a * b; }
#if defined(__HIP_DEVICE_COMPILE__)
EKM_FD fd64<F> m_fma(fd64<F> a, fd64<F> b, fd64<F> c) { return fd64<F>(__builtin_fma(a.v, b.v, c.v)); }
EKM_FD fd64<F> m_fms(fd64<F> a, fd64<F> b, fd64<F> c) { return fd64<F>(__builtin_fma(a.v, b.v, -c.v)); }
EKM_FD fd64<F> m_fnma(fd64<F> a, fd64<F> b, fd64<F> c) { return fd64<F>(__builtin_fma(-a.v, b.v, c.v)); }
#else  // host twin: built with -ffp-contract=off, two roundings like its plain double
EKM_FD fd64<F> m_fma(fd64<F> a, fd64<F> b, fd64<F> c) { return fd64<F>(a.v * b.v + c.v); }
EKM_FD fd64<F> m_fms(fd64<F> a, fd64<F> b, fd64<F> c) { return fd64<F>(a.v * b.v - c.v); }
EKM_FD fd64<F> m_fnma(fd64<F> a, fd64<F> b, fd64<F> c) { return fd64<F>(c.v - a.v * b.v); }
#endif

// min(|a|, |b|, |c|), NaN operands ignored (one v_min3_f32 with abs modifiers on the device)
EKM_HD float m_min3abs(float a, float b, float c) {
  return __builtin_fminf(__builtin_fminf(__builtin_fabsf(a), __builtin_fabsf(b)), __builtin_fabsf(c));
}
EKM_HD double m_min3abs(double a, double b, double c) {
  return __builtin_fmin(__builtin_fmin(__builtin_fabs(a), __builtin_fabs(b)), __builtin_fabs(c));
}

// max of two values neither of which is NaN (one v_max_f32 on the device)
EKM_HD float m_max(float a, float b) { return __builtin_fmaxf(a, b); }
EKM_HD double m_max(double a, double b) { return __builtin_fmax(a, b); }
EKM_FD fd64<F> m_max(fd64<F> a, fd64<F> b) { return fd64<F>(__builtin_fmax(a.v, b.v)); }

// numpy.sign: -1 / 0 / +1, NaN stays NaN
template <class T>
EKM_HD T m_sign(T x) {
  return x > T(0) ? T(1) : (x < T(0) ? T(-1) : (x == T(0) ? T(0) : x));
}

// Wave-uniform "does any lane need this?" test.  On the device it is a ballot, i.e. a
// scalar branch the whole wave takes or skips together (no divergence cost, and whole
// waves of warm-only / cold-only / non-regime points skip work); on the host it is the
// plain per-point condition.  Purely an execution shortcut: lanes that need a value
// always get it computed.
// EKM_WAVE_MASK(cond): the wave's lanes where cond holds, as a 64-bit mask (host: 0 / 1).  Masks of several fresh
// comparisons OR-ed together and tested against zero stay on the scalar unit (s_or_b64 + s_cmp), where OR-ing the
// conditions first and balloting the result costs a v_cndmask + v_cmp per test.
#if defined(__HIP_DEVICE_COMPILE__)
#define EKM_WAVE_MASK(cond) (__builtin_amdgcn_ballot_w64(cond))
#else
#define EKM_WAVE_MASK(cond) ((cond) ? 1ull : 0ull)
#endif
#if defined(EKM_NO_WAVE_SKIP)
#define EKM_ANY(cond) (true)
#elif defined(__HIP_DEVICE_COMPILE__)
#define EKM_ANY(cond) (__builtin_amdgcn_ballot_w64(cond) != 0ull)
#else
#define EKM_ANY(cond) (cond)
#endif

// ---- saturation vapour pressure (es_comp.py) -------------------------------
// es = C1*exp(C3*(t-T0)/(t-C4)) evaluated as exp2(fma((t-T0)*rcp(t-C4), C3*log2(e), log2(C1))):
// one rcp + one exp2 on the transcendental unit, constants folded in double.
namespace k {
constexpr double LOG2E = 1.4426950408889634074;
constexpr double LN2 = 0.69314718055994530942;
constexpr double LOG2_C1 = 9.25552433725897;  // log2(611.21)
}  // namespace k

template <class T>
EKM_HD T es_water(T t) {  // es_comp.py:133-134
  return m_exp2(m_fma((t - T(k::T0)) * m_rcp(t - T(k::C4W)), T(k::C3W * k::LOG2E), T(k::LOG2_C1)));
}

template <class T>
EKM_HD T es_ice(T t) {  // es_comp.py:137-138
  return m_exp2(m_fma((t - T(k::T0)) * m_rcp(t - T(k::C4I)), T(k::C3I * k::LOG2E), T(k::LOG2_C1)));
}

// es and d(es)/dT for one phase from one reciprocal (es_comp.py:169-174).
// ONE_FMA (the IFS Newton step and its regime-1 guess only -- the VALU-bound kernels of BASELINE configs 4 and 5):
// (t - T0)/(t - C4) = 1 - (T0 - C4)*r, so the exponent is ONE fma of r = 1/(t - C4) instead of a subtraction, a product and
// an fma.  The cancellation costs the exponent 1.5e-6 absolute in fp32 -- es to 6e-6 of the reference instead of 2.5e-6
// over 180-330 K, against a bar of 1e-4; nothing in fp64.  Everything else keeps the reference's operator order: the es
// OUTPUTS, the slope functions, the bisection's tables (a wider error band would move rounding-level sign decisions) and
// the Bolton Newton steps (whose cancelling dlnf amplifies an es error a thousandfold on unphysical input: with the
// one-fma form there the wide-domain fuzz counted 5,500 points beyond 1e-4 where the reference's own fp32 run has 2,300).
template <bool ONE_FMA = false, class T>
EKM_HD void es_slope_water(T t, T& es, T& des) {
  const T r = m_rcp(t - T(k::C4W));
  if (ONE_FMA)
    es = m_exp2(m_fma(r, T(-k::C3W * k::LOG2E * (k::T0 - k::C4W)), T(k::C3W * k::LOG2E + k::LOG2_C1)));
  else
    es = m_exp2(m_fma((t - T(k::T0)) * r, T(k::C3W * k::LOG2E), T(k::LOG2_C1)));
  des = es * T(k::sw) * (r * r);
}

template <bool ONE_FMA = false, class T>
EKM_HD void es_slope_ice(T t, T& es, T& des) {
  const T r = m_rcp(t - T(k::C4I));
  if (ONE_FMA)
    es = m_exp2(m_fma(r, T(-k::C3I * k::LOG2E * (k::T0 - k::C4I)), T(k::C3I * k::LOG2E + k::LOG2_C1)));
  else
    es = m_exp2(m_fma((t - T(k::T0)) * r, T(k::C3I * k::LOG2E), T(k::LOG2_C1)));
  des = es * T(k::si) * (r * r);
}

// Where the reference's es is exactly ZERO: its exp underflows (argument below ln 2^-150 resp. ln 2^-1075) for
// t <= 48.175793 K in fp32 and t < 7.35720061305125 K in fp64 (largest / smallest such values found on the NumPy
// restatement; one transition).  Only the Davies-Jones regime-1 guess cares: it divides des by es (thermo.py:1119), 0/0
// there.  Above that temperature the reference's es is a denormal down to which v_exp_f32 does not go (it returns 0
// below 2^-126, t < 52.5 K): such an es is nothing beside any pressure, and the formulas that divide by it take the
// limit instead of the quotient.
template <class T>
EKM_HD T es_zero_below() {
  return sizeof(T) == 4 ? T(48.175797f) : T(7.35720061305125);
}
template <class T>
EKM_HD T es_negligible() {
  return sizeof(T) == 4 ? T(2e-38f) : T(1e-300);
}

// Mixed phase (es_comp.py:141-166).  The reference gathers three masks; per point that
// is: ice at t <= TI, water at t >= T0, alpha-blend in between.  NaN fails both tests and
// lands in the blend, giving NaN.  A wave whose lanes are all at or below TI (all at or
// above T0) takes the one-phase formula alone: atmospheric fields are coherent along a level.
template <class T>
EKM_HD T es_mixed(T t) {
  const bool ice = t <= T(k::TI), wat = t >= T(k::T0);
  if (!EKM_ANY(!ice)) return es_ice(t);    // the whole wave is at or below TI: one rcp + one exp2
  if (!EKM_ANY(!wat)) return es_water(t);  // the whole wave is at or above T0
  const T ew = es_water(t), ei = es_ice(t);
  const T a = m_sq((t - T(k::TI)) * T(1.0 / (k::T0 - k::TI)));
  const T mid = m_fma(a, ew - ei, ei);  // = a*ew + (1-a)*ei
  return ice ? ei : (wat ? ew : mid);
}

// es and slope of the mixed phase together (es_comp.py:177-200)
template <bool ONE_FMA = false, class T>
EKM_HD void es_slope_mixed(T t, T& es, T& des) {
  const bool ice = t <= T(k::TI), wat = t >= T(k::T0);
  if (!EKM_ANY(!ice)) {
    es_slope_ice<ONE_FMA>(t, es, des);
    return;
  }
  if (!EKM_ANY(!wat)) {
    es_slope_water<ONE_FMA>(t, es, des);
    return;
  }
  T ew, dw, ei, di;
  es_slope_water<ONE_FMA>(t, ew, dw);
  es_slope_ice<ONE_FMA>(t, ei, di);
  const T x = t - T(k::TI);
  const T a = m_sq(x * T(1.0 / (k::T0 - k::TI)));
  const T da = T(k::dalpha_c) * x;
  const T dif = ew - ei;
  const T mid = m_fma(a, dif, ei);                      // a*ew + (1-a)*ei
  const T dmid = m_fma(da, dif, m_fma(a, dw - di, di));  // da*ew + a*dw - da*ei + (1-a)*di
  es = ice ? ei : (wat ? ew : mid);
  des = ice ? di : (wat ? dw : dmid);
}

template <int PHASE, class T>
EKM_HD T es_phase(T t) {  // es_comp.py:31-79
  if (PHASE == PHASE_WATER) return es_water(t);
  if (PHASE == PHASE_ICE) return es_ice(t);
  return es_mixed(t);
}

template <int PHASE, class T>
EKM_HD void es_slope_phase(T t, T& es, T& des) {  // es_comp.py:82-106
  if (PHASE == PHASE_WATER)
    es_slope_water(t, es, des);
  else if (PHASE == PHASE_ICE)
    es_slope_ice(t, es, des);
  else
    es_slope_mixed(t, es, des);
}

template <class T>
EKM_HD T t_from_es(T es) {  // es_comp.py:109-130 (always the water formula)
  const T l = m_log2(es * T(1.0 / k::C1));  // v = ln(es/C1) = l*ln2, folded into the constants
  return m_div(m_fms(l, T(k::LN2 * k::C4W), T(k::C3W * k::T0)), m_fms(l, T(k::LN2), T(k::C3W)));
}

// ---- humidity conversions ---------------------------------------------------
template <class T>
EKM_HD T e_from_q(T q, T p) {  // thermo.py:105-131
  return m_div(p * q, m_fma(T(k::q_c), q, T(k::eps)));
}

template <class T>
EKM_HD T e_from_w(T w, T p) {  // thermo.py:134-159
  return m_div(p * w, T(k::eps) + w);
}

template <class T>
EKM_HD T q_from_e(T e, T p, T epsv) {  // thermo.py:162-196
  T v = m_fma(T(k::eps - 1), e, p);
  if ((p - e) < epsv) v = nan_v<T>();
  return m_div(T(k::eps) * e, v);
}

template <class T>
EKM_HD T w_from_e(T e, T p, T epsv) {  // thermo.py:199-232
  T v = p - e;
  if (v < epsv) v = nan_v<T>();
  return m_div(T(k::eps) * e, v);
}

template <class T>
EKM_HD T w_from_q(T q) {  // thermo.py:80-102
  return m_div(q, T(1) - q);
}

template <class T>
EKM_HD T q_from_w(T w) {  // thermo.py:55-77
  return m_div(w, T(1) + w);
}

// d(ws)/dT = eps*des*p/(p-es)^2, NaN where p-es < eps (thermo.py:367-415)
template <class T>
EKM_HD T ws_slope(T p, T es, T des, T epsv) {
  T v = p - es;
  if (v < epsv) v = nan_v<T>();
  const T r = m_rcp(v);
  return T(k::eps) * des * p * (r * r);
}

// d(qs)/dT = eps*des*p/(p+es*(eps-1))^2, NaN where p-es < eps (thermo.py:418-467)
template <class T>
EKM_HD T qs_slope(T p, T es, T des, T epsv) {
  T v = m_sq(m_fma(es, T(k::eps - 1.0), p));
  if ((p - es) < epsv) v = nan_v<T>();
  return m_div(T(k::eps) * des * p, v);
}

// ---- dry thermodynamics ------------------------------------------------------
// power with a NON-INTEGER exponent as the reference's namespace (libm) has it for a base of -inf: +inf for y > 0, +0 for
// y < 0, as for +inf -- exp2(y*log2(-inf)) is NaN.  (A pressure of -0.0 makes p0/p = -inf; every other special base --
// negative, zero of either sign, NaN -- comes out of exp2(y*log2(x)) as libm's pow has it.)
template <class T>
EKM_HD T m_pow_ni(T x, T y) {
  return m_pow(x == T(-std::numeric_limits<double>::infinity()) ? -x : x, y);
}

template <class T>
EKM_HD T theta(T t, T p) {  // thermo.py:801-829
  return t * m_pow_ni(m_div(T(k::p0), p), T(k::kappa));
}

template <class T>
EKM_HD T t_from_theta(T th, T p) {  // thermo.py:832-858
  return th * m_pow_ni(p * T(1.0 / k::p0), T(k::kappa));
}

template <class T>
EKM_HD T p_on_dry_adiabat(T t, T t_def, T p_def) {  // thermo.py:861-889
  return p_def * m_pow_ni(m_div(t, t_def), T(1 / k::kappa));
}

template <class T>
EKM_HD T t_on_dry_adiabat(T p, T t_def, T p_def) {  // thermo.py:892-920
  return t_def * m_pow_ni(m_div(p, p_def), T(k::kappa));
}

template <class T>
EKM_HD T virtual_t(T t, T q) {  // thermo.py:738-764
  return t * m_fma(T(k::tv_c1), q, T(1));
}

template <int METHOD, class T>
EKM_HD T lcl_t(T t, T td) {  // thermo.py:923-968
  if (METHOD == LCL_DAVIES)  // the two "- T0" of the reference's bracket folded into its constant: two fma
    return m_fnma(m_fnma(T(4.36e-4), t, m_fma(T(1.571e-3), td, T(0.212 - (1.571e-3 - 4.36e-4) * k::T0))), t - td, td);
  // a saturated parcel (t == td): the reference's quotient is exactly 1 and its logarithm exactly 0; t*rcp(td) is one ulp off
  // (finite and nonzero: inf/inf and 0/0 are NaN in the reference)
  const T lr = (t - td == T(0) && t != T(0)) ? T(0) : m_log(m_div(t, td));
  return T(56.0) + m_rcp(m_fma(lr, T(1.0 / 800), m_rcp(td - T(56))));
}

// exp2 whose result may be a denormal (v_exp_f32 flushes those to zero; the reference's exp / pow round them).  Needed
// where the reference's own arithmetic meets such a value again: the exact step of the bolton35 search (its two terms
// are compared where both are that small) and bolton35's th_sat and theta_e (a denormal (p0/p)^(kappa*(1 - 0.28*w)) times an
// overflowed exponential is inf in the reference, 0*inf = NaN only once the power has gone to zero altogether).
EKM_HD float m_exp2_denorm(float x) { return x < -100.0f ? m_exp2(x + 64.0f) * 0x1p-64f : m_exp2(x); }
EKM_HD double m_exp2_denorm(double x) { return m_exp2(x); }
EKM_FD fd64<F> m_exp2_denorm(fd64<F> x) { return m_exp2(x); }

// ---- equivalent potential temperature (thermo.py:1020-1323) ------------------
// HAVE_Q: humidity given as specific humidity q (td derived from it,
// thermo.py:1036-1037); otherwise as dewpoint td.
template <int METHOD, bool HAVE_Q, class T>
EKM_HD T ept(T t, T hum, T p) {
  T td, q = T(0);
  if (HAVE_Q) {
    q = hum;
    td = t_from_es(e_from_q(q, p));  // thermo.py:702-735
  } else {
    td = hum;
  }
  if (METHOD == EPT_IFS) {  // thermo.py:1169-1175
    const T th = theta(t, p);
    const T tl = lcl_t<LCL_DAVIES>(t, td);
    if (!HAVE_Q) q = q_from_e(es_water(td), p, T(k::eps_default));
    return th * m_exp(m_div(T(k::K0_ifs) * q, tl));
  }
  const T tl = lcl_t<LCL_BOLTON>(t, td);
  const T w = HAVE_Q ? w_from_q(q) : w_from_e(es_water(td), p, T(k::eps_default));
  if (METHOD == EPT_BOLTON35) {  // thermo.py:1205-1213
    const T th = t * m_exp2_denorm((T(k::kappa) * (T(1) - T(0.28) * w)) * m_log2(m_div(T(k::p0), p)));  // (a power that keeps its denormals)
    return th * m_exp(m_div(T(2675.0) * w, tl));
  }
  // bolton39, thermo.py:1268-1278
  const T e = e_from_w(w, p);
  const T th = theta(t, p - e) * m_pow_ni(m_div(t, tl), T(0.28) * w);
  return th * m_exp((m_div(T(3036.0), tl) - T(1.78)) * w * (T(1) + T(0.448) * w));
}

// th_sat and G_sat(scale) of the saturated parcel (thermo.py:1177-1182,
// 1215-1224, 1280-1295).  bolton39 masks es where p - es < 1e-4.
template <int METHOD, class T>
EKM_HD void sat_terms(T t, T p, T scale, T& th_sat, T& g_sat) {
  T es = es_mixed(t);
  if (METHOD == EPT_IFS) {
    th_sat = theta(t, p);
    const T qs = q_from_e(es, p, T(k::eps_default));
    g_sat = m_div((scale * T(k::K0_ifs)) * qs, t);
  } else if (METHOD == EPT_BOLTON35) {
    const T ws = w_from_e(es, p, T(k::eps_default));
    th_sat = t * m_exp2_denorm((T(k::kappa) * (T(1) - T(0.28) * ws)) * m_log2(m_div(T(k::p0), p)));
    g_sat = m_div((scale * T(2675.0)) * ws, t);
  } else {
    if ((p - es) < T(1e-4)) es = nan_v<T>();
    th_sat = theta(t, p - es);
    const T ws = w_from_e(es, p, T(k::eps_default));
    g_sat = (m_div(scale * T(3036.0), t) - scale * T(1.78)) * ws * (T(1) + T(0.448) * ws);
  }
}

template <int METHOD, class T>
EKM_HD T ept_sat(T t, T p) {  // thermo.py:1042-1045
  T th, g;
  sat_terms<METHOD>(t, p, T(1), th, g);
  return th * m_exp(g);
}

// Horner with ascending coefficients (the namespace's polyval(x, c))
template <class T>
EKM_HD T poly2(T x, double c0, double c1, double c2) {
  return m_fma(m_fma(T(c2), x, T(c1)), x, T(c0));
}

template <class T>
EKM_HD T wbpt_direct_f64(T e) {  // thermo.py:1047-1053
  const T x = e * T(1.0 / 273.16);
  const T a = m_fma(m_fma(m_fma(m_fma(T(-5.205688), x, T(2.574631)), x, T(16.11182)), x, T(-20.68208)), x, T(7.101574));
  const T b = m_fma(m_fma(m_fma(m_fma(T(-0.5929340), x, T(-0.6899655)), x, T(3.781782)), x, T(-3.552497)), x, T(1.0));
  return e - m_exp(m_div(a, b));
}
EKM_HD double wbpt_direct(double e) { return wbpt_direct_f64(e); }
EKM_FD fd64<F> wbpt_direct(fd64<F> e) { return wbpt_direct_f64(e); }

EKM_HD float wbpt_direct(float e) {
  const float x = e * float(1.0 / 273.16);
  // The reference's polyval runs in fp64 whatever the array dtype (fp64 coefficient
  // array); x^4 overflows fp32 for absurd theta_e where fp64 stays finite.  Only
  // there (never for atmospheric input) follow it in double.
  if (!(x <= 1e4f && x >= -1e4f) && x == x) return float(wbpt_direct(double(e)));
  const float a = 7.101574f + (-20.68208f + (16.11182f + (2.574631f + -5.205688f * x) * x) * x) * x;
  const float b = 1.0f + (-3.552497f + (3.781782f + (-0.6899655f + -0.5929340f * x) * x) * x) * x;
  return e - m_exp(m_div(a, b));
}

// sign(r)*dt of the reference's `t += sign(r)*dt` without branches: +-dt by sign-bit transfer, r itself
// (0 -> no move, NaN -> NaN) when r is zero or unordered
EKM_HD float bisect_step(float r, float dt) { return (r < 0.0f || r > 0.0f) ? __builtin_copysignf(dt, r) : r; }
EKM_HD double bisect_step(double r, double dt) { return (r < 0.0 || r > 0.0) ? __builtin_copysign(dt, r) : r; }
EKM_FD fd64<F> bisect_step(fd64<F> r, fd64<F> dt) { return fd64<F>(bisect_step(r.v, dt.v)); }

// Moist-adiabat inversion by 12 fixed halvings (thermo.py:1055-1079), IFS variant, table-free statement (host twin with
// EKM_TWIN_TABLE_FREE=1; the gfx950 kernels walk the search tree below, which takes the same decisions from an LDS table).
//  * The reference's residual ept*exp(G_sat) - t*(p0/p)^kappa (thermo.py:1075) is divided by the positive per-point
//    constant (p0/p)^kappa: r = te*exp2(g) - t with te = ept*(p/p0)^kappa has the same sign and costs one fma.
//  * The two divisions of G_sat = -K0*qs/t (thermo.py:1177-1182) are merged into one reciprocal.
//  * The reference's mask `p - es < eps -> NaN` (thermo.py:192-196; it makes t NaN from that step on) is applied ONCE
//    after the search: fl(p - es) is non-increasing in es, so "some visited lattice point had p - es < eps" is exactly
//    "p - max(es visited) < eps".  Outside the mask v = p - 0.378*es > 0, so the unmasked steps stay finite.
template <class T>
EKM_HD T t_on_ma_bisect_ifs_te(T te, T p) {
  T t = T(k::T0 - 20);
  T dt = T(120.0);
  T esmax = T(0);
#pragma unroll 1
  for (int it = 0; it < 12; ++it) {
    const T es = es_mixed(t);
    esmax = m_max(esmax, es);
    const T g = T(-k::K0_ifs * k::eps * k::LOG2E) * es * m_rcp(m_fma(T(k::eps - 1), es, p) * t);  // log2 of exp(-K0*qs/t)
    dt *= T(0.5);
    t += bisect_step(m_fms(te, m_exp2(g), t), dt);
  }
  if ((p - esmax) < T(k::eps_default)) t = nan_v<T>();
  return t;
}

template <class T>
EKM_HD T t_on_ma_bisect_ifs(T e, T p) {
  return t_on_ma_bisect_ifs_te(e * m_exp2(T(k::kappa) * m_log2(p * T(1.0 / k::p0))), p);
}

// The search only ever evaluates the saturated parcel at the lattice temperatures t_m = 253.16 + (m - 2048)*120/2048,
// m in [1, 4095]: float32(253.16) is a multiple of 2^-15 and every step is a multiple of 15/512, so the reference's
// accumulated fp32 `t` IS the lattice value, bit for bit (checked exhaustively in tests/test_engine_host.py), and
// everything that depends on t_m alone can be tabulated once per device and kept in LDS.
constexpr int kBisectLattice = 4096;

template <class T>
EKM_HD T bisect_lattice_t(int m) {
  return T(k::T0 - 20) + T(m - kBisectLattice / 2) * T(120.0 / 2048);
}

// Beside es_m = es_mixed(t_m) the tables hold the factor of the step's exponent that depends on t_m alone:
//   ifs       a_m = -K0*eps*log2(e)*es_m/t_m      exponent g = a_m * rcp(p + (eps-1)*es_m)
//   bolton35  a_m = -2675*log2(e)/t_m             exponent g = kl + ws*(a_m - 0.28*kl),  kl = kappa*log2(p/p0)
//   bolton39  a_m = (-3036/t_m + 1.78)*log2(e)    exponent g = a_m*ws*(1 + 0.448*ws) + kappa*log2((p - es_m)/p0)
// with ws = eps*es_m/(p - es_m).  A step of the search is r = e*exp2(g) - t_m: the reference's residual
// ept*exp(G_sat) - th_sat (thermo.py:1075) divided by the positive th_sat/t -- for ifs the per-point constant
// (p0/p)^kappa, folded into e = te (the wet-bulb from q then needs no power of the pressure at all); for bolton35 /
// bolton39 (thermo.py:1215-1224, 1280-1295) the step's own (p0/p)^(kappa*(1-0.28*ws)) resp. (p0/(p-es))^kappa, whose
// exponent joins g.  The reference's mask `p - es < eps -> NaN` (thermo.py:192-196, 229-232, 1283-1284: it makes t NaN
// from that step on) is applied ONCE after the search: fl(p - es) is non-increasing in es, so "some visited lattice
// point had p - es < eps" is exactly "p - max(es visited) < eps".  Outside the mask p - es >= eps > 0: the unmasked
// steps stay finite.
template <int METHOD, class T>
EKM_HD T bisect_second(T es, T rt) {  // rt = 1/t_m
  if (METHOD == EPT_IFS) return T(-k::K0_ifs * k::eps * k::LOG2E) * es * rt;
  if (METHOD == EPT_BOLTON35) return T(-2675.0 * k::LOG2E) * rt;
  return m_fma(T(-3036.0), rt, T(1.78)) * T(k::LOG2E);
}

// fp64: es_m alone, in lattice order (the fp64 step forms a_m from the ONE reciprocal it takes, 1/(v*t_m))
EKM_HD void bisect_es_fill(double* __restrict__ tab, int m) { tab[m] = es_mixed(bisect_lattice_t<double>(m)); }

// ---- the IFS search in fp32 as a walk down the search TREE, most steps decided without a transcendental ---------
// The 12 halvings visit the nodes of a complete binary tree over the lattice: depth d holds the lattice points
// m = (2j + 1) * 2^(11-d), j = 0 .. 2^d - 1, and the next node is the left or right child by the sign of the residual.
// With the nodes stored in heap order (node i, children 2i and 2i + 1) the whole search state is ONE integer and a step
// is i = i + i + (r > 0): ONE v_alignbit_b32 on the sign of -r (bisect_heap_child), no temperature arithmetic, no index conversion.
//
// The reference decides on r = theta_e*exp(G_sat) - th_sat (thermo.py:1075), i.e. after division by the positive
// (p0/p)^kappa on  r_m = te*2^(g_m) - t_m,  g_m = a_m/w_m,  w_m = p + (eps - 1)*es_m  (one rcp and one exp2 per step;
// the stepwise search).  In logarithms:  r_m > 0  <=>  log2(te) + g_m > log2(t_m)  <=>  a_m > u_m*w_m  with
// u_m = L_m - lte,  L_m = log2(t_m/273.16) tabulated beside (es_m, a_m)  and  lte = log2(te/273.16), which the callers
// have anyway (the logarithm of te is formed before te is).  D_m = a_m - u_m*w_m is three fused operations.  Its sign
// is the sign of the reference's fp32 residual whenever |D_m| exceeds what rounding can move either of the two
// evaluations: in units of the exponent (D/w = g - u) the fp32 residual carries <= |g|*1.5e-7 + 1.7e-7 (a_m*rcp(w),
// exp2, the product), the logarithmic form <= 4e-7 + |u|*1.2e-7 (lte, L_m, u*w), and |u| = |g| at the root.  A step
// with |D_m| <= kHeapTau0*p + kHeapTau1*|a_m| (four times those bounds, p >= w) is AMBIGUOUS and is decided by the
// reference's own arithmetic, evaluated for the wave -- a wave-uniform branch, taken on about two of the twelve steps
// of a wave (the last ones, where some lane of the wave stands within a millikelvin of its root).  So every decision is
// the stepwise search takes, bit for bit, at 7 plain instructions per step (round 4: 10; the stepwise search: 11 + 2
// transcendentals).
// `all_exact` (tuning parameter bisect_exact) makes every step ambiguous: the stepwise search itself, against which
// tests/test_gpu_census.py compares the default on every point of the benchmark field.
//
// Exactly zero or NaN residuals (the reference then stops moving, resp. turns NaN: `t += sign(r)*dt`) are always
// ambiguous: the exact branch records the node's temperature (or the NaN) as the lane's final answer, together with the
// largest es visited so far -- the reference keeps re-evaluating the same point from then on -- for the NaN rule
// `p - max(es visited) < eps` (thermo.py:192-196, applied once at the end).
// Table layout: heap order, node 0 unused; 4096 (es_i, a_i) pairs then the 4096 L_i (48 KiB) or 4096 16-byte records (es_i,
// a_i, t_i, L_i) read by one ds_read_b128 (64 KiB: the fp32 IFS walk) -- heap_rec / heap_read below.
constexpr int kHeapNodes = kBisectLattice;
constexpr double kHeapTau0 = 2.5e-6, kHeapTau1 = 1.2e-6;

// Two layouts of the tree (REC = floats of table per node):
//   3  4096 (es, a) pairs, then the 4096 L: 48 KiB, one ds_read_b64 + one ds_read_b32 per node (two address shifts);
//   4  4096 records (es, a, L, t_m) of 16 B: 64 KiB, ONE ds_read_b128 per node (one shift, one LDS instruction less per
//      step and point; t_m spares the exact branch the lattice arithmetic).  Two 1024-thread workgroups with 64 KiB each
//      fill a CU with 8 waves per SIMD at <= 64 registers: the fp32 IFS walk (ops.hpp::OpThreads).  The Bolton walks need
//      more registers than that and the fp64 walks carry the fp64 lattice beside the tree: both keep layout 3.
#ifndef EKM_HEAP_REC_IFS
#define EKM_HEAP_REC_IFS 4
#endif
#ifndef EKM_HEAP_B128
#define EKM_HEAP_B128 1    // 1: the record in one ds_read_b128; 0: ds_read_b64 + ds_read_b32 from the one address (A/B:
#endif                     // 3.32 against 3.21 ms; merged by the compiler into ONE ds_read_b96 when L sits right behind a: 3.73)
#ifndef EKM_HEAP_L_SLOT
#define EKM_HEAP_L_SLOT 3  // record = (es, a, b, L)
#endif
// A/B (round 6, NEGATIVE: 3.07 against 3.02 ms): the fourth float of the 16-byte record as b_m = kHeapTau1*|a_m|, the node's own
// share of the tolerance band, so that the band of a step is ONE fma on values the step has anyway, b_m + (kHeapTau0/eps)*
// w_m  (w_m = p + (eps - 1)*es_m >= eps*p wherever es_m <= p), instead of the product kHeapTau0*p -- recomputed at every
// step for want of a register at the 64-VGPR cap -- and an fma: 8 -> 7 vector instructions per step and point in the
// static code, but the band is 1/eps = 1.6 x wider in its pressure part, the exact branch is entered that much more
// often, and the EXECUTED count went up, 130.9 -> 139.1 per point (profiles/r06_tree_walk.txt).  0 (default): round 5's
// band, kHeapTau1*|a_m| + kHeapTau0*p, and the slot holds t_m.
#ifndef EKM_HEAP_BAND_SLOT
#define EKM_HEAP_BAND_SLOT 0
#endif
// A/B only (per-depth attribution of LDS conflict cycles, tools/pmc_bisect_depth.sh): the walk stops after this many steps
#ifndef EKM_WALK_DEPTH
#define EKM_WALK_DEPTH 12
#endif
template <int METHOD, class T>
constexpr int heap_rec() {
  return (METHOD == EPT_IFS && sizeof(T) == 4) ? EKM_HEAP_REC_IFS : 3;
}
struct HeapNode {
  float es, a, L, b;  // b: the record's fourth float (16-byte records only), else 0
};
// A/B only (profiles/r06_tree_walk.txt): the records of depths >= 8 -- where the LDS bank conflicts
// are, 96 % of them in the last four steps -- stored at idx ^ ((idx >> 4) & 15), so that nodes 16 apart stop sharing the
// banks of a 16-byte slot.  The collisions there are random (a wave's lanes scatter over hundreds of leaves of the noisy
// benchmark field), a permutation of the slots cannot change their statistics, and the three extra instructions per deep
// step cost more than nothing: 0 (default) = plain heap order.
#ifndef EKM_HEAP_SWIZZLE
#define EKM_HEAP_SWIZZLE 0
#endif
EKM_HD unsigned heap_slot(unsigned node) {
#if EKM_HEAP_SWIZZLE
  return node >= 256u ? node ^ ((node >> 4) & 15u) : node;
#else
  return node;
#endif
}
template <int REC>
EKM_HD HeapNode heap_read(const float* __restrict__ tab, unsigned node) {
  HeapNode r;
  if (REC == 4) node = heap_slot(node);
#if defined(__HIP_DEVICE_COMPILE__)
  const char* __restrict__ base = reinterpret_cast<const char*>(tab);
  if (REC == 4) {
#if EKM_HEAP_B128
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = *reinterpret_cast<const f4*>(base + (node << 4));  // one ds_read_b128 (a register tuple of four per point)
    r.es = v[0];
    r.a = v[1];
    r.L = v[EKM_HEAP_L_SLOT];
    r.b = v[5 - EKM_HEAP_L_SLOT];
#else
    typedef float f2 __attribute__((ext_vector_type(2)));
    const char* __restrict__ rec = base + (node << 4);  // ONE address: ds_read_b64 + ds_read_b32 offset:12
    const f2 ea = *reinterpret_cast<const f2*>(rec);
    r.es = ea[0];
    r.a = ea[1];
    r.L = *reinterpret_cast<const float*>(rec + 4 * EKM_HEAP_L_SLOT);
    r.b = EKM_HEAP_BAND_SLOT ? *reinterpret_cast<const float*>(rec + 4 * (5 - EKM_HEAP_L_SLOT)) : 0.0f;
#endif
  } else {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 ea = *reinterpret_cast<const f2*>(base + (node << 3));  // one ds_read_b64
    r.es = ea[0];
    r.a = ea[1];
    r.L = *reinterpret_cast<const float*>(base + 8 * kHeapNodes + (node << 2));
    r.b = 0.0f;
  }
#else
  if (REC == 4) {
    r.es = tab[4 * node];
    r.a = tab[4 * node + 1];
    r.L = tab[4 * node + EKM_HEAP_L_SLOT];
    r.b = tab[4 * node + (5 - EKM_HEAP_L_SLOT)];
  } else {
    r.es = tab[2 * node];
    r.a = tab[2 * node + 1];
    r.L = tab[2 * kHeapNodes + node];
    r.b = 0.0f;
  }
#endif
  return r;
}
template <int REC>
EKM_HD float heap_es(const float* __restrict__ tab, unsigned node) {
  if (REC == 4) node = heap_slot(node);
  return tab[(REC == 4 ? 4 : 2) * node];
}

// lattice index of heap node i at depth d (2^d <= i < 2^(d+1)), d <= 11
EKM_HD int bisect_heap_lattice(int i, int d) { return (2 * (i - (1 << d)) + 1) << (11 - d); }

// F64: the tree the fp64 walk tests on.  Its exact steps take es from the fp64 lattice, so the tree holds THOSE values
// rounded to float (6e-8) -- not the fp32 evaluation of es, which is 1e-6 .. 2.5e-6 off them: where p - es cancels (a parcel
// near boiling: es = 0.9 p at 369 K and p0) that difference moved bolton39's test by more than its band (found by the
// fuzz of theta_w, 5 of 1 M points two quanta off the stepwise search).
template <int METHOD = EPT_IFS, int REC = 3, bool F64 = false>
EKM_HD void bisect_heap_fill(float* __restrict__ tab, int i) {
  float es = 0.0f, a = 0.0f, L = 0.0f, t = 0.0f;
  if (i >= 1) {
    int d = 0;
    while ((2 << d) <= i) ++d;
    if (F64) {
      const double td = bisect_lattice_t<double>(bisect_heap_lattice(i, d)), esd = es_mixed(td);
      t = (float)td;
      es = (float)esd;
      a = (float)bisect_second<METHOD>(esd, 1.0 / td);
      L = (float)m_log2(td * (1.0 / 273.16));
    } else {
      t = bisect_lattice_t<float>(bisect_heap_lattice(i, d));
      es = es_mixed(t);
      a = bisect_second<METHOD>(es, 1.0f / t);  // IEEE division: once per device
      L = (float)m_log2(double(t) * (1.0 / 273.16));
    }
  }
  if (REC == 4) {
    const int s = (int)heap_slot((unsigned)i);  // (a permutation within each aligned group of 16 records)
    tab[4 * s] = es;
    tab[4 * s + 1] = a;
    tab[4 * s + EKM_HEAP_L_SLOT] = L;
    tab[4 * s + (5 - EKM_HEAP_L_SLOT)] = EKM_HEAP_BAND_SLOT ? float(kHeapTau1) * __builtin_fabsf(a) : t;
  } else {
    tab[2 * i] = es;
    tab[2 * i + 1] = a;
    tab[2 * kHeapNodes + i] = L;
  }
}

// heap child of `node` by the sign of the residual r: 2*node + (r > 0), from MINUS r (`nr`): ONE v_alignbit_b32 on the device,
// the sign bit of -r shifted in behind the node index ({node, -r} >> 31); the sign tests below form -D directly.  (-r has
// its sign bit set for r = +0 as well, where `r > 0` is false: a zero residual is always ambiguous, the exact branch then
// records the node as the lane's answer and where the walk goes afterwards no longer matters; NaN likewise.)
EKM_HD unsigned bisect_heap_child(unsigned node, float nr) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(node, __builtin_bit_cast(unsigned, nr), 31u);
#else
  return (node << 1) | (__builtin_bit_cast(unsigned, nr) >> 31);
#endif
}

// The Bolton methods walk the same tree with their own second table value (bisect_second) and their own form of the test,
// each the reference's residual in logarithms, multiplied through by the positive denominators (v = p - es_m > 0 outside
// the NaN mask; inside it the result is NaN whatever the walk does):
//   bolton35  g = kl + ws*(a_m - 0.28*kl), ws = eps*es_m/v, kl = kappa*log2(p/p0)   (thermo.py:1215-1224)
//             g > u  <=>  D = (kl - u)*v + eps*es_m*(a_m - 0.28*kl) > 0                        no transcendental
//   bolton39  g = a_m*ws*(1 + 0.448*ws) + kappa*log2(v/p0)                           (thermo.py:1280-1295)
//             g > u  <=>  D = a_m*eps*es_m*(v + 0.448*eps*es_m) + v^2*(kappa*log2(v/p0) - u) > 0   ONE log2 (of three)
// with u = L_m - le, le = log2(theta_e/273.16) (`te` is theta_e itself for these methods; `kl` is read for bolton35 only).
// D of one node (fp32): positive <=> the residual is positive, unless |D| <= the band (then `amb`); returned NEGATED.  w = the positive
// denominator the exact step divides by (ifs: p + (eps-1)*es; Bolton: p - es); thr0 = the part of the band that goes with p.
constexpr double kB35WsExact = 2.0;
#ifndef EKM_B35_FOLD
#define EKM_B35_FOLD 1  // kl folded into the logarithm the walk carries (A/B: 0 subtracts it at every step)
#endif


// The reference's bolton35 residual as it stands, theta_e*exp(G_sat(scale=-1)) - th_sat (thermo.py:1075, 1215-1224), in
// base 2: theta_e*2^(a_m*ws) - t_m*2^(kl*(0.28*ws - 1)), a_m = -2675*log2(e)/t_m, kl = kappa*log2(p/p0).  NOT divided by
// th_sat/t_m: where ws is large both terms underflow and the reference's 0 - 0 = 0 keeps the search on the node.
template <class T>
EKM_HD T bisect_b35_residual(T te, T tm, T a, T ws, T kl) {
  return m_fms(te, m_exp2_denorm(a * ws), tm * m_exp2_denorm(kl * m_fms(T(0.28), ws, T(1))));
}

// The reference's residual at one lattice node, theta_e*exp(G_sat(scale=-1)) - th_sat (thermo.py:1075), as the stepwise search
// evaluates it (operation for operation the same in the tree walk's exact branch and in bisect_exact_walk): for ifs and
// bolton39 divided by the positive th_sat/t_m -- r = te*2^g - t_m --, for bolton35 as it stands (bisect_b35_residual).
// w = the denominator of the step (ifs: p + (eps-1)*es; Bolton: p - es).
template <int METHOD>
EKM_HD float bisect_exact_residual(float es, float a, float w, float te, float tm, float kl) {
  if (METHOD == EPT_BOLTON35) return bisect_b35_residual(te, tm, a, float(k::eps) * es * m_rcp(w), kl);
  float g;
  if (METHOD == EPT_IFS) {
    g = a * m_rcp(w);
  } else {
    const float ws = float(k::eps) * es * m_rcp(w);
    const float gs = a * ws;
    g = m_fma(gs, m_fma(0.448f, ws, 1.0f), float(k::kappa) * m_log2(w * float(1.0 / k::p0)));
    // an INFINITE theta_e (fp32 overflow of its exponential: a parcel near boiling): the reference's theta_e*exp(G_sat(-1))
    // is +inf while that exponential is a nonzero (denormal) number and NaN (inf*0) once it underflows -- decided by the
    // exponential ALONE, not by the one above that carries kappa*log2(v/p0) as well (found by the fuzz of the wet-bulb from
    // the dewpoint: t 378.8 K, td 374.8 K, p 1162 hPa -> reference 373.13 K, the walk NaN)
    if (EKM_ANY(!(te < std::numeric_limits<float>::infinity()))) {
      if (!(te < std::numeric_limits<float>::infinity())) return te * m_exp2_denorm(gs * m_fma(0.448f, ws, 1.0f));
    }
  }
  return m_fms(te, METHOD == EPT_IFS ? m_exp2(g) : m_exp2_denorm(g), tm);
}

// WS: test ws >= kB35WsExact at every node (A/B only: both walks ask once, of the hottest node they visited)
template <int METHOD, bool WS = false, bool F64 = false, bool BAND_SLOT = false>
EKM_HD float bisect_fast_test(float es, float a, float u, float p, float kl, float& w, float thr0, bool& amb, float b = 0.0f) {
  float D, scale;  // D: MINUS the quantity of the comments above (bisect_heap_child takes it so); scale: |a_m| resp. its
  bool big_ws = false;  // counterpart -- the part of the band that goes with the size of the exponent
  if (METHOD == EPT_IFS) {
    w = m_fma(float(k::eps - 1), es, p);
    D = m_fms(u, w, a);
    scale = a;
  } else if (METHOD == EPT_BOLTON35) {
    w = p - es;
    const float ees = float(k::eps) * es;
    scale = ees * m_fnma(0.28f, kl, a);
    D = m_fms(EKM_B35_FOLD ? u : u - kl, w, scale);  // u = L_m - (le + kl): the walks fold kl into the logarithm they carry
    // ws = eps*es/(p - es) >= kB35WsExact: the reference's two terms, theta_e*exp(-2675*ws/t) and th_sat =
    // t*(p0/p)^(kappa*(1 - 0.28*ws)), can BOTH leave the normal range there (ws of several hundred where p - es is a
    // fraction of a pascal: 0 - 0, sign 0, the reference stays on this node for good), and D -- the residual divided
    // by th_sat/t -- no longer says what the reference's own subtraction gives.  Below the limit both terms are
    // normal numbers (|exponents| <= 29*2 resp. 47*1.56) and the division changes nothing.  Decided by the exact step.
    if (WS) big_ws = !(w > float(kB35WsExact) * ees);
  } else {
    w = p - es;
    const float ees = float(k::eps) * es;
    const float aees = a * ees;
    scale = aees * m_fma(0.448f, ees, w);
    const float v2 = w * w;
    const float X = m_fnma(float(k::kappa), m_log2(w * float(1.0 / k::p0)), u);
    D = m_fms(v2, X, scale);
    thr0 = float(2.0 * kHeapTau0) * v2;  // the band in units of v^2 here (log2 of a small v adds its own rounding)
    // the fp64 walk tests on p and es ROUNDED to float: w = p - es is off by up to 1.2e-7*p, which where it cancels is
    // more than the band above knows of -- dD = dw*(2*w*|X| + kappa/ln2*w + |a|*eps*es)
    if (F64) thr0 = m_fma(1.5e-7f * p, m_fma(w, m_fma(2.0f, __builtin_fabsf(X), 0.5f), __builtin_fabsf(aees)), thr0);
  }
  if (METHOD == EPT_IFS && BAND_SLOT)  // b = kHeapTau1*|a_m| from the record; (kHeapTau0/eps)*w >= kHeapTau0*p (EKM_HEAP_BAND_SLOT)
    amb = !(__builtin_fabsf(D) > m_fma(w, float(kHeapTau0 / k::eps), b));
  else
    amb = !(__builtin_fabsf(D) > m_fma(__builtin_fabsf(scale), float(METHOD == EPT_IFS ? kHeapTau1 : 2.0 * kHeapTau1), thr0));  // NaN: ambiguous
  if (METHOD == EPT_BOLTON35 && WS) amb = amb || big_ws;
  return D;
}

// The stepwise search itself on the fp32 tree, one point, every step the reference's own residual -- rolled up (it runs for
// the rare points the tree walk hands back, below), the same operations as the walk's exact branch: with `all_exact` the
// walk returns these bits.
template <int METHOD>
EKM_HD float bisect_exact_walk(float te, float p, float kl, const float* __restrict__ tab) {
  constexpr int REC = heap_rec<METHOD, float>();
  unsigned node = 1u;
  float tfix = 0.0f, esmax = 0.0f;
#pragma unroll 1
  for (int d = 0; d < 12; ++d) {
    const HeapNode nd = heap_read<REC>(tab, node);
    const float es = nd.es, a = nd.a;
    const float w = METHOD == EPT_IFS ? m_fma(float(k::eps - 1), es, p) : p - es;
    const float tm = bisect_lattice_t<float>(bisect_heap_lattice((int)node, d));
    const float r = bisect_exact_residual<METHOD>(es, a, w, te, tm, kl);
    if (tfix == 0.0f) {
      esmax = m_max(esmax, es);
      if (!(r < 0.0f || r > 0.0f)) tfix = r == 0.0f ? tm : r;
    }
    node = bisect_heap_child(node, -r);
  }
  float t = float(k::T0 - 20) + float(2 * (int)node - (3 * kHeapNodes - 1)) * float(120.0 / 4096);
  if (tfix != 0.0f) t = tfix;
  if ((p - esmax) < float(k::eps_default)) t = nan_v<float>();
  return t;
}

template <int METHOD, int V>
EKM_HD void t_on_ma_bisect_heap(const float (&lte)[V], const float (&te)[V], const float (&p)[V], const float (&kl)[V],
                                const float* __restrict__ tab, float (&out)[V], bool all_exact = false) {
  // The largest es visited (for the NaN rule) is not tracked step by step: es grows with t, so it is es of the HOTTEST node
  // visited -- the first node the walk left downwards, or the deepest one if it never did -- and the heap index of every
  // visited node is a prefix of the leaf's (node at depth d = leaf >> (12 - d)): one count-leading-zeros on the inverted
  // path and ONE table read at the end instead of a running maximum in every step (two registers per point less, too).
  unsigned node[V];
  float thr0[V], tfix[V], lq[V];  // tfix: the final answer of a lane whose residual came out exactly zero or NaN; 0 = none yet
#pragma unroll
  for (int j = 0; j < V; ++j) {
    node[j] = 1u;
    thr0[j] = float(kHeapTau0) * p[j];
    // an infinite theta_e (fp32 overflow of its exponential for q of 0.8) times an exp(G_sat) that underflowed is NaN in the
    // reference where the logarithmic test reads +inf: with a NaN logarithm every test of such a point is NaN, i.e.
    // ambiguous, and every step takes the reference's own residual
    lq[j] = (METHOD != EPT_IFS && !(lte[j] < std::numeric_limits<float>::infinity())) ? nan_v<float>() : lte[j];
    // a NEGATIVE te (a temperature handed over in Celsius) has no logarithm; the reference's residual te*2^g - t_m is
    // negative at every node then, as for te = 0: the same walk (all the way down, 133.19 K), not NaN
    if (METHOD == EPT_IFS && te[j] < 0.0f) lq[j] = -std::numeric_limits<float>::infinity();
    if (METHOD == EPT_BOLTON35 && EKM_B35_FOLD) lq[j] += kl[j];  // bolton35's test takes u - kl (bisect_fast_test)
    tfix[j] = 0.0f;
  }
  constexpr int REC = heap_rec<METHOD, float>();
  constexpr bool BAND_SLOT = METHOD == EPT_IFS && REC == 4 && EKM_HEAP_BAND_SLOT;
#pragma unroll
  for (int d = 0; d < EKM_WALK_DEPTH; ++d) {
    float es[V], a[V], w[V], D[V], L[V], b[V];
    bool amb[V];
    unsigned long long any = 0ull;  // lanes of the wave with an ambiguous test at this depth, over the V points
    // (forcing all table reads of a step out before the first is waited for -- the compiler pairs the points, two LDS
    // round trips per step -- changes nothing: profiles/r05_tree_walk.txt)
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const HeapNode nd = heap_read<REC>(tab, node[j]);
      es[j] = nd.es;
      a[j] = nd.a;
      L[j] = nd.L;
      b[j] = nd.b;
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float u = L[j] - lq[j];
      D[j] = bisect_fast_test<METHOD, false, false, BAND_SLOT>(es[j], a[j], u, p[j], kl[j], w[j], thr0[j], amb[j], b[j]);
      any |= EKM_WAVE_MASK(amb[j]);
      amb[j] = amb[j] || all_exact;
    }
    if (any != 0ull || all_exact) {  // (wave-uniform) the reference's own residual for the lanes that need it
#pragma unroll
      for (int j = 0; j < V; ++j) {
        if (amb[j]) {
          const float tm = bisect_lattice_t<float>(bisect_heap_lattice((int)node[j], d));
          // ifs: te itself is needed here only -- it is formed from its logarithm, which the sign tests carry anyway (one
          // exp2 per point less on the common path, and no register for it)
          const float tej = METHOD == EPT_IFS ? 273.16f * m_exp2(lq[j]) : te[j];
          const float r = bisect_exact_residual<METHOD>(es[j], a[j], w[j], tej, tm, kl[j]);  // the stepwise search's step
          D[j] = -r;
          if (!(r < 0.0f || r > 0.0f) && tfix[j] == 0.0f) {  // zero: the reference stays on this point; NaN: it turns NaN
            tfix[j] = r == 0.0f ? tm : r;  // (lattice temperatures are >= 133 K: never the "none" value)
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) node[j] = bisect_heap_child(node[j], D[j]);
  }
#pragma unroll
  for (int j = 0; j < V; ++j) {
    // the leaf (4096 <= node < 8192): t0 + (M - 4096)*120/4096 with M = 2*(node - 4096) + 1, i.e. 2*node - 12287 half
    // steps from t0 -- exactly the reference's accumulated fp32 sum
    float t = float(k::T0 - 20) + float(2 * (int)node[j] - (3 * kHeapNodes - 1)) * float(120.0 / 4096);
    // hottest node visited: decisions of depths 0 .. 10 are bits 11 .. 1 of the leaf index; the first 0 among them
    const unsigned inv = ~node[j] & 0xFFEu;
    unsigned dmax = (inv ? (unsigned)__builtin_clz(inv) : 32u) - 20u;
    dmax = dmax < 11u ? dmax : 11u;
    unsigned nmax = node[j] >> (12u - dmax);
    if (EKM_ANY(tfix[j] != 0.0f)) {  // (rare, wave-uniform) a search that got stuck at depth k visited depths 0 .. k only
      if (tfix[j] != 0.0f) {
        t = tfix[j];
        if (t == t) {  // the stuck temperature is a lattice point: its depth is 11 - (trailing zeros of its lattice index)
          const unsigned m = (unsigned)(int)__builtin_rintf((t - float(k::T0 - 20)) * float(2048.0 / 120.0)) + 2048u;
          const unsigned kf = 11u - (unsigned)__builtin_ctz(m | 2048u);
          const unsigned invk = inv & ~((1u << (12u - kf)) - 1u);
          dmax = invk ? (unsigned)__builtin_clz(invk) - 20u : kf;
          nmax = node[j] >> (12u - dmax);
        }
      }
    }
    const float esmax = heap_es<REC>(tab, nmax);
    if ((p[j] - esmax) < float(k::eps_default)) t = nan_v<float>();
    if (METHOD == EPT_BOLTON35) {
      // ws = eps*es/(p - es) >= kB35WsExact at the hottest node visited (ws grows with es): the reference's two terms, theta_e*
      // exp(-2675*ws/t) and th_sat = t*(p0/p)^(kappa*(1 - 0.28*ws)), can BOTH leave the normal range there (ws of several
      // hundred where p - es is a fraction of a pascal: 0 - 0, sign 0, the reference stays on that node for good), and the
      // sign test -- the residual divided by th_sat/t -- no longer says what the reference's own subtraction gives.  Below
      // the limit both terms are normal numbers (|exponents| <= 29*2 resp. 47*1.56) and the division changes nothing.  The
      // walk agrees with the stepwise search up to the first such node, which is therefore ON its path: asked once per
      // point, here, and such a point (never an atmospheric one) is searched again step by step.
      const bool suspect = !((p[j] - esmax) > float(k::eps / kB35WsExact) * esmax) && !all_exact;
      if (EKM_ANY(suspect)) {
        if (suspect) t = bisect_exact_walk<METHOD>(te[j], p[j], kl[j], tab);
      }
    }
    out[j] = t;
  }
}

// The same walk for the fp64 kernels (T = double or fd64): the sign tests run in fp32 on the SAME fp32 tree (the inputs
// rounded to float: their rounding, <= 6e-8 relative, is far inside the tolerance band), and a step whose test is within
// the band is decided by the fp64 residual of the stepwise search -- es from the fp64 lattice table, ONE software
// reciprocal, one software exp2, ~60 fp64 operations -- which the stepwise fp64 search paid at every one of its twelve
// steps and this walk pays on the few ambiguous ones.  The band is the fp32 one (the fp64 residual's own rounding is
// nine orders below it), so outside it the fp32 sign IS the sign of the fp64 residual.
// Largest es visited (the NaN rule): es grows with t, so it is es of the hottest node visited -- the first node the
// walk left DOWNWARDS, or the deepest node if it never did -- read once at the end from the fp64 table.
// the stepwise search's step in T at lattice point m (es from the fp64 lattice table), operation for operation the same in
// the walk's exact branch and in bisect_exact_walk64: 1/(v*t) gives both 1/v and 1/t
template <int METHOD, class T>
EKM_HD T bisect_exact_residual64(T es, T tm, T te, T p, T kl) {
  const T v = METHOD == EPT_IFS ? m_fma(T(k::eps - 1), es, p) : p - es;
  const T r1 = m_rcp(v * tm);
  const T rv = r1 * tm, a = bisect_second<METHOD>(es, r1 * v);
  if (METHOD == EPT_BOLTON35) return bisect_b35_residual(te, tm, a, T(k::eps) * es * rv, kl);
  T g;
  if (METHOD == EPT_IFS) {
    g = a * rv;
  } else {
    const T ws = T(k::eps) * es * rv;
    g = m_fma(a * ws, m_fma(T(0.448), ws, T(1)), T(k::kappa) * m_log2(v * T(1.0 / k::p0)));
    if (!(te < T(std::numeric_limits<double>::infinity())))  // an infinite theta_e: see bisect_exact_residual
      return te * m_exp2_denorm((a * ws) * m_fma(T(0.448), ws, T(1)));
  }
  return m_fms(te, METHOD == EPT_IFS ? m_exp2(g) : m_exp2_denorm(g), tm);
}

// the stepwise search in T, one point, rolled up (for the rare points the fp64 walk hands back; bisect_exact_walk's twin)
template <int METHOD, class T>
EKM_HD T bisect_exact_walk64(T te, T p, T kl, const T* __restrict__ es_tab) {
  unsigned node = 1u;
  bool fixed = false;
  T tfix = T(0.0), esmax = T(0.0);
#pragma unroll 1
  for (int d = 0; d < 12; ++d) {
    const int m = bisect_heap_lattice((int)node, d);
    const T tm = bisect_lattice_t<T>(m), es = es_tab[m];
    const T r = bisect_exact_residual64<METHOD>(es, tm, te, p, kl);
    if (!fixed) {
      esmax = m_max(esmax, es);
      if (!(r < T(0) || r > T(0))) {
        fixed = true;
        tfix = r == T(0) ? tm : r;
      }
    }
    node = bisect_heap_child(node, r > T(0) ? -1.0f : 1.0f);
  }
  T t = T(k::T0 - 20) + T(2 * (int)node - (3 * kHeapNodes - 1)) * T(120.0 / 4096);
  if (fixed) t = tfix;
  if ((p - esmax) < T(k::eps_default)) t = nan_v<T>();
  return t;
}

template <int METHOD, class T, int V>
EKM_HD void t_on_ma_bisect_heap64(const T (&lte)[V], const T (&te)[V], const T (&p)[V], const T (&kl)[V],
                                  const float* __restrict__ heap, const T* __restrict__ es_tab, T (&out)[V],
                                  bool all_exact = false) {
  unsigned node[V];
  int kfix[V];  // step at which the residual came out exactly zero / NaN (the reference then stays / turns NaN); -1 = never
  float ltef[V], pf[V], klf[V], thr0[V];
  T tfix[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    node[j] = 1u;
    kfix[j] = -1;
    ltef[j] = (float)lte[j];
    pf[j] = (float)p[j];
    klf[j] = (float)kl[j];
    thr0[j] = float(1.5 * kHeapTau0) * pf[j];
    if (METHOD != EPT_IFS && !(ltef[j] < std::numeric_limits<float>::infinity())) ltef[j] = nan_v<float>();  // as in the fp32 walk
    if (METHOD == EPT_IFS && te[j] < T(0.0)) ltef[j] = -std::numeric_limits<float>::infinity();              // likewise
    if (METHOD == EPT_BOLTON35 && EKM_B35_FOLD) ltef[j] += klf[j];
    tfix[j] = T(0.0);
  }
  constexpr int REC = heap_rec<METHOD, double>();
#pragma unroll
  for (int d = 0; d < 12; ++d) {
    float D[V];
    bool amb[V];
    unsigned long long any = 0ull;  // lanes of the wave with an ambiguous test at this depth, over the V points
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const HeapNode nd = heap_read<REC>(heap, node[j]);
      const float es = nd.es, a = nd.a;
      const float u = nd.L - ltef[j];
      float w;
      D[j] = bisect_fast_test<METHOD, false, true>(es, a, u, pf[j], klf[j], w, thr0[j], amb[j]);  // (the band is wide enough for
      any |= EKM_WAVE_MASK(amb[j]);                                                         //  the inputs' rounding to float)
      amb[j] = amb[j] || all_exact;
    }
    if (any != 0ull || all_exact) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        if (amb[j]) {
          const int m = bisect_heap_lattice((int)node[j], d);
          const T tm = bisect_lattice_t<T>(m);
          const T r = bisect_exact_residual64<METHOD>(es_tab[m], tm, te[j], p[j], kl[j]);
          D[j] = r > T(0) ? -1.0f : 1.0f;
          if (!(r < T(0) || r > T(0)) && kfix[j] < 0) {
            kfix[j] = d;
            tfix[j] = r == T(0) ? tm : r;
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) node[j] = bisect_heap_child(node[j], D[j]);
  }
#pragma unroll
  for (int j = 0; j < V; ++j) {
    T t = T(k::T0 - 20) + T(2 * (int)node[j] - (3 * kHeapNodes - 1)) * T(120.0 / 4096);
    // hottest node visited (as in the fp32 walk): among the nodes of depth 0 .. last (last = 11, or the step the search
    // got stuck at), the first one left downwards -- decisions of depths 0 .. 10 are bits 11 .. 1 of the leaf index --,
    // else the node of depth `last`; a visited node's heap index is a prefix of the leaf's
    const unsigned last = kfix[j] >= 0 ? (unsigned)kfix[j] : 11u;
    const unsigned inv = ~node[j] & 0xFFEu & ~((1u << (12u - last)) - 1u);
    const unsigned dmax = inv ? (unsigned)__builtin_clz(inv) - 20u : last;
    const unsigned nmax = node[j] >> (12u - dmax);
    const T esmax = es_tab[bisect_heap_lattice((int)nmax, (int)dmax)];
    if (kfix[j] >= 0) t = tfix[j];
    if ((p[j] - esmax) < T(k::eps_default)) t = nan_v<T>();
    if (METHOD == EPT_BOLTON35) {  // ws >= kB35WsExact at the hottest node visited: searched again step by step (see the fp32 walk)
      const bool suspect = !((p[j] - esmax) > T(k::eps / kB35WsExact) * esmax) && !all_exact;
      if (EKM_ANY(suspect)) {
        if (suspect) t = bisect_exact_walk64<METHOD, T>(te[j], p[j], kl[j], es_tab);
      }
    }
    out[j] = t;
  }
}

template <int METHOD, class T>
EKM_HD T t_on_ma_bisect(T e, T p) {
  if (METHOD == EPT_IFS) return t_on_ma_bisect_ifs(e, p);
  T t = T(k::T0 - 20);
  T dt = T(120.0);
#pragma unroll 1
  for (int it = 0; it < 12; ++it) {
    T th, g;
    sat_terms<METHOD>(t, p, T(-1.0), th, g);
    dt *= T(0.5);
    t += m_sign(e * m_exp(g) - th) * dt;
  }
  return t;
}

// ---- Davies-Jones regime selection (thermo.py:1114-1128) -------------------------------------------
// The initial guess switches formula where c_te crosses D(p), 1 and 0.4, and the guesses on the two sides
// of a threshold differ (by ~2 K at 10 hPa), so a point whose c_te is within rounding noise of a threshold
// must take the reference's decision, not the one our differently rounded fp32 c_te happens to give.  The
// fp32 kernels therefore re-derive the four predicates in DOUBLE on the rare lanes whose fast c_te lies
// within kTieBand of a threshold (~1.3e-5 of the points of the benchmark field; ~0.3 % of its waves hold
// one): te is recomputed from the op's own inputs by `te_exact()`, then c_te and D in the reference's operator
// order with the reference's fp32-rounded constants -- power(t0/te, lambda), 1/(0.1859e-5*p + 0.6512),
// c_te > D, 1 <= c_te <= D, 0.4 <= c_te < 1, c_te < 0.4 -- i.e. what the reference's fp32 sequence computes
// up to its own rounding noise.  The guess FORMULAS stay in fp32 (they are continuous in c_te).
struct Regime {
  bool r1, r2, r3, r4;  // c_te > D | 1 <= c_te <= D | 0.4 <= c_te < 1 | c_te < 0.4  (a NaN c_te: r2 alone, whose guess is NaN then)
};

constexpr double kTieBand = 4e-6;  // > 3x the worst fp32 error of the fast c_te (te ~3e-7, x lambda, + exp2/log2)

struct TeNone {  // no exact te available (fp64 kernels never need one)
  static constexpr bool have = false;
  static constexpr int kind = 0;
  EKM_HD double operator()() const { return 0.0; }
};

// te = ept*(p/p0)^kappa (thermo.py:1109-1110) from a given theta_e
template <class T>
struct TeFromEpt {
  static constexpr bool have = true;
  static constexpr int kind = 1;
  T e, p;
  EKM_HD double operator()() const {
    return double(e) * m_exp2(double(float(k::kappa)) * m_log2(double(p) * (1.0 / k::p0)));  // p0 = 1e5 is exact in fp32
  }
};

// The double-precision decision itself: stand-ins for c_te/D ("cd", only ever compared with 1) and for c_te ("c",
// compared with 1 and 0.4) on the side of each threshold that the double evaluation found.  On the device it is a
// NOINLINE function: inlined, its fp64 code and 64-bit constants cost the fast path 20-30 VGPRs and pushed the
// kernel to the SGPR limit (spills through v_writelane), although it runs on ~1e-5 of the points.
struct TieDecision {
  float cd, c;
};

EKM_HD TieDecision tie_decision_from_te(double te, double p) {
  const double c = m_exp2(double(float(k::lambda)) * m_log2(double(273.16f) * m_rcp(te)));
  const double D = m_rcp(double(0.1859e-5f) * p + double(0.6512f));
  TieDecision d;
  d.cd = c > D ? 2.0f : 0.5f;
  float cf = float(c);  // keep the rounded value on the side of 1 and of 0.4 that c is on
  if (c >= 1.0 && cf < 1.0f) cf = 1.0f;
  if (c < 1.0 && cf >= 1.0f) cf = 0.99999994f;
  if (c >= double(0.4f) && cf < 0.4f) cf = 0.4f;
  if (c < double(0.4f) && cf >= 0.4f) cf = 0.39999998f;
  d.c = cf;
  return d;
}

// How a tie is handled is a policy of the caller:
//   TieInline  decide in double on the spot (host twin; a wave-uniform branch inside the per-point body);
//   TieFlag    the gfx950 kernels' first pass: decide in fp32, only RECORD that this lane met a tie -- the body stays
//              branch-free, so the compiler can interleave the four points of a lane (with the branch inside, the
//              wet-bulb kernel lost 4-8 %: profiles/r02_sweep_tie.txt);
//   TieExact   the kernels' second pass, run after the four points for the (rare) lanes that recorded a tie
//              (map_kernel.hpp::apply_points): the same arithmetic with the decision taken in double.
struct TieInline {
  static constexpr int mode = 0;
};
struct TieFlag {
  static constexpr int mode = 1;
  bool hit = false;
};
struct TieExact {
  static constexpr int mode = 2;
};

#if defined(__HIP_DEVICE_COMPILE__)
#define EKM_TIE_NOINLINE __device__ __attribute__((noinline))
#else
#define EKM_TIE_NOINLINE inline
#endif
EKM_TIE_NOINLINE TieDecision tie_decide_ept(float e, float p);           // te = ept*(p/p0)^kappa
EKM_TIE_NOINLINE TieDecision tie_decide_tqp(float t, float q, float p);  // te = t*exp(K0*q/t_lcl(t, td(q, p)))

template <class TeExact, class T>
EKM_HD TieDecision tie_decide(const TeExact& x, T p) {
  if constexpr (TeExact::kind == 1)
    return tie_decide_ept(float(x.e), float(x.p));
  else if constexpr (TeExact::kind == 2)
    return tie_decide_tqp(float(x.t), float(x.q), float(x.p));
  else
    return tie_decision_from_te(x(), double(p));
}

template <class T, class TeExact, class Tie>
EKM_HD Regime davies_regime(T c_te, T cd, T p, const TeExact& te_exact, Tie& tie) {
  // c_dec / cd_dec: the values the four predicates are taken from.  On a tie lane they are replaced by stand-ins
  // that fall on the side of each threshold the double evaluation found.
  T c_dec = c_te, cd_dec = cd;
#ifndef EKM_NO_TIE
  if constexpr (TeExact::have && sizeof(T) == 4) {
    // one v_min3 over |c_te/D - 1|, |c_te - 1|, |c_te/0.4 - 1| and one compare (NaN compares false: no tie)
    const T d1 = cd - T(1), d2 = c_te - T(1), d3 = c_te * T(2.5) - T(1);
    const bool is_tie = m_min3abs(d1, d2, d3) < T(kTieBand);
    if constexpr (Tie::mode == 1) {
      tie.hit = tie.hit || is_tie;
    } else {
      bool enter = is_tie;
      if constexpr (Tie::mode == 0) enter = EKM_ANY(is_tie);
      if (enter) {
        if (is_tie) {
          const TieDecision dec = tie_decide(te_exact, p);
          cd_dec = T(dec.cd);
          c_dec = T(dec.c);
        }
      }
    }
  }
#endif
  (void)tie;
  // Three comparisons, the rest is mask logic on the scalar unit: r3 = c < 1 and not c < 0.4; r2 = not c < 1 and not
  // c/D > 1.  For an ordered c_te these ARE the reference's masks (1 <= c_te <= D, 0.4 <= c_te < 1); a NaN c_te, for
  // which the reference selects nothing, lands in r2 here -- and every guess formed from a NaN c_te is NaN, as is what
  // the reference then carries through its Newton step (tw = ept with a NaN c_te in the residual).
  Regime r;
  const bool lt1 = c_dec < T(1);
  r.r1 = cd_dec > T(1);
  r.r4 = c_dec < T(0.4);
  r.r3 = lt1 && !r.r4;
  r.r2 = !lt1 && !r.r1;
  return r;
}

// Pressure-only terms of the moist-adiabat inversion.  With p a level vector they are
// level constants; with p a field they cost one log2 and two exp2 per point.
template <class T>
struct PTerms {
  T p;     // pressure (Pa)
  T l;     // log2(p/p0): (p/p0)^kappa = exp2(kappa*l) is formed only where a regime 2-4 guess needs it (thermo.py:1109)
  T thf;   // (p0/p)^kappa   (thermo.py:829)
  T dinv;  // 0.1859e-5*p + 0.6512 = 1/D(p)   (thermo.py:1100-1102)
};

template <class T>
EKM_HD PTerms<T> pterms(T p) {
  PTerms<T> r;
  r.l = m_log2(p * T(1.0 / k::p0));
  r.p = p;
  r.thf = m_exp2(T(-k::kappa) * r.l);
  r.dinv = m_fma(T(0.1859e-5), p, T(0.6512));
  return r;
}

// Davies-Jones inversion for the IFS theta_e (the path BASELINE.json configs 4/5 name):
// same regimes, same single Newton step as thermo.py:1081-1159 + 1184-1197, with the
// transcendental count cut to the minimum: powers as exp2 of one shared log2, 1/c_te as exp2 of
// the negated exponent, the Newton residual 1 - c_te/f from ONE exp2 (the exponents of c_te, c_tw and
// exp(G) add), reciprocals shared between qs and its slope, and regime / phase work skipped by whole
// waves that do not need it.  `lte` = log2(te/273.16); `te` itself is only needed by the regime-1 guess.
// `te_fn()` = te and `pp_fn()` = (p/p0)^kappa are evaluated lazily, inside the wave-uniform branches that need them:
// te only feeds the regime-1 guess, (p/p0)^kappa only the guesses of regimes 2-4 (thermo.py:1114-1128), and regimes
// are coherent along a level -- a wave of cold upper-level points never forms the pressure power, a wave of warm
// points never forms te (one log2 + one exp2 resp. one exp2 per point, and the k1/k2 polynomials).
template <class T, class TeFn, class PpFn, class TeExact, class Tie>
EKM_HD T t_on_ma_newton_ifs_core(T lte, T p, T dinv, const TeFn& te_fn, const PpFn& pp_fn, const TeExact& te_exact,
                                 Tie& tie) {
  const T lam = T(k::lambda);
  const T c_te = m_exp2(-lam * lte);  // (t0/te)^lambda
  const T cd = c_te * dinv;           // c_te / D
  const Regime R = davies_regime(c_te, cd, p, te_exact, tie);

  // initial guess in deg C; later regimes overwrite earlier ones (thermo.py:1114-1128).  No regime holds only for a
  // NaN c_te, i.e. a NaN lte: the reference then carries ept through one Newton step against a NaN c_te -- NaN.
  T tw = lte;
  if (EKM_ANY(R.r1)) {
    // (lanes outside regime 1 do not use this guess: they take a temperature far below TI, so that a wave whose regime-1
    // lanes are all at or below TI -- regime 1 is te < 244-260 K, ice for most of them -- evaluates ONE phase of es instead of
    // both phases and the blend because of the warmer lanes beside them; the regime-1 lanes' own values are unchanged)
    const T te = R.r1 ? te_fn() : T(k::TI - 50.0);
    T es, des;
    es_slope_mixed<true>(te, es, des);
    // te - t0 - A*ws/(1 + A*ws*des/es) with ws = eps*es/v, v = p - es (thermo.py:1116-1119): the two divisions are one,
    // A*eps*es/(v + A*eps*des).  v is NaN where p - es < eps, and where the reference's es underflows to zero its
    // des/es is 0/0 (es_zero_below; an es of ours that is zero above that temperature leaves te - t0, as it should).
    T v = p - es;
    if (v < T(k::eps_default) || te < es_zero_below<T>()) v = nan_v<T>();
    const T g1 = m_fnma(T(2675 * k::eps) * es, m_rcp(m_fma(T(2675 * k::eps), des, v)), te - T(273.16));
    if (R.r1) tw = g1;
  }
  if (EKM_ANY(R.r2 || R.r3 || R.r4)) {
    const T pp = pp_fn();
    const T k1 = poly2(pp, -53.737, 137.81, -38.5);
    const T k2 = poly2(pp, -0.384, 56.831, -4.392);
    const T k2m = k2 - T(1.21);
    if (R.r2) tw = m_fnma(k2, c_te, k1);
    if (R.r3) tw = m_fnma(k2m, c_te, k1 - T(1.21));
    if (EKM_ANY(R.r4)) {
      const T g4 = m_fma(T(0.58), m_exp2(lam * lte), m_fnma(k2m, c_te, k1 - T(2.66)));  // 0.58/c_te
      if (R.r4) tw = g4;
    }
  }
  tw = tw + T(k::T0);

  // one Newton step (thermo.py:1132-1149, 1184-1197): tw -= (f - c_te)/(f*dlnf) = (1 - c_te/f)/dlnf with
  // c_te/f = exp2(lambda*(log2(tw/t0) - log2(te/t0)) + lambda*K0*log2(e)*qs/tw)
  // With qr = qs/tw and dqs the mixed-phase slope of qs (thermo.py:418-467): dlnf = -lambda*(1/tw + K0*(dqs - qr)/tw), so
  // the step is tw*(1 - ratio)/(lambda*(1 + K0*(dqs - qr))): two reciprocals -- 1/(v*tw), shared by qr and dqs, and the
  // step's denominator -- instead of three.
  const T ltw = m_log2(tw * T(1.0 / 273.16));
  T es, des;
  es_slope_mixed<true>(tw, es, des);
  T v = m_fma(T(k::eps - 1), es, p);
  if ((p - es) < T(k::eps_default)) v = nan_v<T>();
  const T r2 = m_rcp(v * tw);
  const T rv = r2 * tw;
  const T qr = T(k::eps) * es * r2;
  const T ratio = m_exp2(m_fma(lam, ltw - lte, T(k::lambda * k::K0_ifs * k::LOG2E) * qr));
  const T dqs = (T(k::eps) * p) * des * (rv * rv);
  const T den = m_fma(T(k::K0_ifs), dqs - qr, T(1));
  // f == 0 (tw -> inf) or f == inf make the reference's (f - c_te)/(f*dlnf) NaN; so does ratio = inf/NaN here
  tw = m_fma(((T(1) - ratio) * T(1.0 / k::lambda)) * m_rcp(den), tw, tw);
  if (tw <= T(0)) tw = nan_v<T>();  // thermo.py:1155
  return tw;
}

// theta_e given (temperature_on_moist_adiabat, wet-bulb from dewpoint, theta_w by Newton)
template <class T, class Tie>
EKM_HD T t_on_ma_newton_ifs(T e, const PTerms<T>& P, Tie& tie) {
  // te = ept*(p/p0)^kappa (thermo.py:1110); its logarithm needs no power of the pressure
  const TeFromEpt<T> exact{e, P.p};
  const T kl = T(k::kappa) * P.l;
  return t_on_ma_newton_ifs_core(m_log2(e * T(1.0 / 273.16)) + kl, P.p, P.dinv, [&] { return e * m_exp2(kl); },
                                 [&] { return m_exp2(kl); }, exact, tie);
}

// te = theta_e*(p/p0)^kappa for the IFS theta_e from specific humidity (thermo.py:1169-1175 with td from
// q, thermo.py:702-735): the two pressure powers cancel, te = t*exp(K0*q/t_lcl).  The fp32 kernels use the
// fast chain below (`x` = log2 of the exponential factor); this functor is the same quantity in double with
// the reference's fp32-rounded constants, for the regime tie-break only.
template <class T>
struct TeFromTQP {
  static constexpr bool have = true;
  static constexpr int kind = 2;
  T t, q, p;
  EKM_HD double operator()() const {
    const double td_ = double(t), qd = double(q), pd = double(p);
    const double e = pd * qd * m_rcp(double(float(k::eps)) + double(float(k::q_c)) * qd);
    const double v = m_log2(e * m_rcp(double(float(k::C1)))) * k::LN2;
    const double td = (v * double(float(k::C4W)) - double(float(k::C3W * k::T0))) * m_rcp(v - double(float(k::C3W)));
    const double tl = td - (double(0.212f) + double(1.571e-3f) * (td - double(float(k::T0))) -
                            double(4.36e-4f) * (td_ - double(float(k::T0)))) * (td_ - td);
    return td_ * m_exp2(double(float(k::K0_ifs)) * k::LOG2E * qd * m_rcp(tl));
  }
};

EKM_TIE_NOINLINE TieDecision tie_decide_ept(float e, float p) {
  const TeFromEpt<float> x{e, p};
  return tie_decision_from_te(x(), double(p));
}
EKM_TIE_NOINLINE TieDecision tie_decide_tqp(float t, float q, float p) {
  const TeFromTQP<float> x{t, q, p};
  return tie_decision_from_te(x(), double(p));
}

// Davies-Jones (2008): regime initial guess + exactly one Newton step
// (max_iter = 1, thermo.py:1104), tw <= 0 -> NaN (thermo.py:1081-1159).
template <int METHOD, class T, class Tie>
EKM_HD T t_on_ma_newton(T e, T p, Tie& tie) {
  const T t0 = T(273.16);
  const T A = T(2675);
  const T lam = T(k::lambda);
  const T pr = p * T(1.0 / k::p0);
  const T pp = m_pow(pr, T(k::kappa));
  const T te = e * pp;
  const T c_te = m_pow(m_div(t0, te), lam);

  // initial guess (deg C); later regimes overwrite earlier ones (thermo.py:1114-1128)
  T tw = e;
  {
    const TeFromEpt<T> exact{e, p};
    const Regime R = davies_regime(c_te, c_te * (T(0.1859e-5) * p + T(0.6512)), p, exact, tie);
    T g1 = te;  // (only ever used by regime-1 lanes)
    if (EKM_ANY(R.r1)) {
      // lanes outside regime 1 do not use this guess: they take a temperature far below TI, so that they do not make a wave
      // of cold regime-1 lanes evaluate both phases of es (as in t_on_ma_newton_ifs_core; the regime-1 lanes' values are unchanged)
      const T te1 = R.r1 ? te : T(k::TI - 50.0);
      T es, des;
      es_slope_mixed(te1, es, des);
      const T ws = w_from_e(es, p, T(k::eps_default));
      const T aw = A * ws;
      T dr = m_div(aw * des, es);  // A*ws*des/es (thermo.py:1119): 0/0 where the reference's es is zero, else nothing beside 1
      if (!(es > es_negligible<T>()) && es == es) dr = te1 < es_zero_below<T>() ? nan_v<T>() : T(0);
      g1 = te1 - t0 - m_div(aw, T(1) + dr);
    }
    const T k1 = poly2(pp, -53.737, 137.81, -38.5);
    const T k2 = poly2(pp, -0.384, 56.831, -4.392);
    if (R.r1) tw = g1;
    if (R.r2) tw = k1 - k2 * c_te;
    if (R.r3) tw = (k1 - T(1.21)) - (k2 - T(1.21)) * c_te;
    if (R.r4) tw = (k1 - T(2.66)) - (k2 - T(1.21)) * c_te + m_div(T(0.58), c_te);
  }
  tw = tw + T(k::T0);

  // one Newton step (thermo.py:1132-1149)
  {
    const T rtw = m_rcp(tw);
    const T c_tw = m_pow(t0 * rtw, lam);
    T es, des;
    es_slope_mixed(tw, es, des);
    T f, dlnf;
    if (METHOD == EPT_IFS) {  // thermo.py:1184-1197
      const T qs = q_from_e(es, p, T(k::eps_default));
      f = c_tw * m_exp((-lam * T(k::K0_ifs)) * qs * rtw);
      const T dqs = qs_slope(p, es, des, T(k::eps_default));
      const T dg = -T(k::K0_ifs) * qs * (rtw * rtw) + T(k::K0_ifs) * dqs * rtw;
      dlnf = -lam * (rtw + dg);
    } else if (METHOD == EPT_BOLTON35) {  // thermo.py:1226-1250
      const T ws = w_from_e(es, p, T(k::eps_default));
      f = c_tw * m_pow(pr, T(0.28) * ws) * m_exp((-lam * T(2675.0)) * ws * rtw);
      const T dws = ws_slope(p, es, des, T(k::eps_default));
      const T dg = -T(2675.0) * ws * (rtw * rtw) + T(2675.0) * dws * rtw;
      // the middle term multiplies the *es* slope, as the reference does
      dlnf = -lam * (rtw + T(0.28) * m_log(pr) * des + dg);
    } else {  // bolton39, thermo.py:1297-1316 (es un-masked here)
      const T ws = w_from_e(es, p, T(k::eps_default));
      const T g = ((-lam * T(3036.0)) * rtw - (-lam * T(1.78))) * ws * (T(1) + T(0.448) * ws);
      f = c_tw * (T(1) - m_div(es, p)) * m_exp(g);
      const T dws = ws_slope(p, es, des, T(k::eps_default));
      const T dg = -T(3036.0) * (ws + T(0.448) * (ws * ws)) * (rtw * rtw) +
                   (T(3036.0) * rtw - T(1.78)) * (T(1) + T(2 * 0.448) * ws) * dws;
      dlnf = -lam * (rtw + m_div(T(k::kappa) * des, p - es) + dg);
    }
    tw -= m_div(f - c_te, f * dlnf);
  }
  if (tw <= T(0)) tw = nan_v<T>();  // thermo.py:1155
  return tw;
}

template <int METHOD, int TM, class T, class Tie>
EKM_HD T t_on_ma(T e, T p, Tie& tie) {  // thermo.py:1472-1509
  if (TM == T_BISECT) return t_on_ma_bisect<METHOD>(e, p);
  if (METHOD == EPT_IFS) {
    PTerms<T> P;  // (p0/p)^kappa is not needed here
    P.p = p;
    P.l = m_log2(p * T(1.0 / k::p0));
    P.thf = T(0);
    P.dinv = m_fma(T(0.1859e-5), p, T(0.6512));
    return t_on_ma_newton_ifs(e, P, tie);
  }
  return t_on_ma_newton<METHOD>(e, p, tie);
}

}  // namespace ekm
