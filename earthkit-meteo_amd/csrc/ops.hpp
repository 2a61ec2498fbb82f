// Point operators: one functor per C-ABI entry point.  Each maps NIN input
// values of one grid point to NOUT output values; the map kernel
// (map_kernel.hpp) and the host twin (host_twin.cpp) instantiate them.
// `rp` is the op's real-valued parameter (eps of the reference signatures).
#pragma once

#include "thermo_math.hpp"

#ifndef EKM_WAVES_PER_EU
#define EKM_WAVES_PER_EU 1  // one macro for every kernel's launch bounds (a -DEKM_WAVES_PER_EU=N sweep reaches all of them)
#endif
#ifndef EKM_THREADS_DEFAULT
#define EKM_THREADS_DEFAULT 256
#endif
#ifndef EKM_TREE_WAVES
#define EKM_TREE_WAVES 6
#endif
#ifndef EKM_TREE_THREADS
#define EKM_TREE_THREADS 512
#endif
#ifndef EKM_WAVES_PER_EU_DEFAULT
#define EKM_WAVES_PER_EU_DEFAULT EKM_WAVES_PER_EU
#endif
#ifndef EKM_P5_WAVES
#define EKM_P5_WAVES 5
#endif

namespace ekm {

#define EKM_OP(NAME, NIN_, NOUT_, ...)                                      \
  struct NAME {                                                              \
    static constexpr int NIN = NIN_;                                         \
    static constexpr int NOUT = NOUT_;                                       \
    template <class T, class Tie>                                            \
    EKM_HD static void apply_tie(const T* __restrict__ x, T* __restrict__ y, T rp, Tie& tie) { \
      (void)rp;                                                              \
      (void)tie;                                                             \
      __VA_ARGS__                                                            \
    }                                                                        \
    template <class T>                                                       \
    EKM_HD static void apply(const T* __restrict__ x, T* __restrict__ y, T rp) { \
      TieInline tie;                                                         \
      apply_tie(x, y, rp, tie);                                              \
    }                                                                        \
  };

#define EKM_OP_T1(NAME, PARAM, NIN_, NOUT_, ...)                            \
  template <int PARAM>                                                       \
  struct NAME {                                                              \
    static constexpr int NIN = NIN_;                                         \
    static constexpr int NOUT = NOUT_;                                       \
    template <class T, class Tie>                                            \
    EKM_HD static void apply_tie(const T* __restrict__ x, T* __restrict__ y, T rp, Tie& tie) { \
      (void)rp;                                                              \
      (void)tie;                                                             \
      __VA_ARGS__                                                            \
    }                                                                        \
    template <class T>                                                       \
    EKM_HD static void apply(const T* __restrict__ x, T* __restrict__ y, T rp) { \
      TieInline tie;                                                         \
      apply_tie(x, y, rp, tie);                                              \
    }                                                                        \
  };

#define EKM_OP_T2(NAME, P1, P2, NIN_, NOUT_, ...)                           \
  template <int P1, int P2>                                                  \
  struct NAME {                                                              \
    static constexpr int NIN = NIN_;                                         \
    static constexpr int NOUT = NOUT_;                                       \
    template <class T, class Tie>                                            \
    EKM_HD static void apply_tie(const T* __restrict__ x, T* __restrict__ y, T rp, Tie& tie) { \
      (void)rp;                                                              \
      (void)tie;                                                             \
      __VA_ARGS__                                                            \
    }                                                                        \
    template <class T>                                                       \
    EKM_HD static void apply(const T* __restrict__ x, T* __restrict__ y, T rp) { \
      TieInline tie;                                                         \
      apply_tie(x, y, rp, tie);                                              \
    }                                                                        \
  };

// thermo.py:21-52
EKM_OP(OpCelsiusToKelvin, 1, 1, y[0] = x[0] + T(k::T0);)
EKM_OP(OpKelvinToCelsius, 1, 1, y[0] = x[0] - T(k::T0);)
// thermo.py:55-102
EKM_OP(OpQFromW, 1, 1, y[0] = q_from_w(x[0]);)
EKM_OP(OpWFromQ, 1, 1, y[0] = w_from_q(x[0]);)
// thermo.py:105-159
EKM_OP(OpEFromQ, 2, 1, y[0] = e_from_q(x[0], x[1]);)
EKM_OP(OpEFromW, 2, 1, y[0] = e_from_w(x[0], x[1]);)
// thermo.py:162-232
EKM_OP(OpQFromE, 2, 1, y[0] = q_from_e(x[0], x[1], rp);)
EKM_OP(OpWFromE, 2, 1, y[0] = w_from_e(x[0], x[1], rp);)
// thermo.py:235-341 (the inner q/w conversion runs with the default eps)
EKM_OP_T1(OpSvp, PHASE, 1, 1, y[0] = es_phase<PHASE>(x[0]);)
EKM_OP_T1(OpSatW, PHASE, 2, 1, y[0] = w_from_e(es_phase<PHASE>(x[0]), x[1], T(k::eps_default));)
EKM_OP_T1(OpSatQ, PHASE, 2, 1, y[0] = q_from_e(es_phase<PHASE>(x[0]), x[1], T(k::eps_default));)
// thermo.py:344-467
EKM_OP_T1(OpSvpSlope, PHASE, 1, 1, T es; T des; es_slope_phase<PHASE>(x[0], es, des); y[0] = des;)
EKM_OP_T1(OpSatWSlope, PHASE, 2, 1, T es; T des; es_slope_phase<PHASE>(x[0], es, des);
          y[0] = ws_slope(x[1], es, des, rp);)
EKM_OP_T1(OpSatQSlope, PHASE, 2, 1, T es; T des; es_slope_phase<PHASE>(x[0], es, des);
          y[0] = qs_slope(x[1], es, des, rp);)
// same with caller-supplied es / es_slope: inputs (p, es, es_slope)
EKM_OP(OpSatWSlopeFromEs, 3, 1, y[0] = ws_slope(x[0], x[1], x[2], rp);)
EKM_OP(OpSatQSlopeFromEs, 3, 1, y[0] = qs_slope(x[0], x[1], x[2], rp);)
// thermo.py:470-491
EKM_OP(OpTFromEs, 1, 1, y[0] = t_from_es(x[0]);)
// thermo.py:494-556
EKM_OP(OpRhFromTd, 2, 1, y[0] = m_div(T(100.0) * es_water(x[1]), es_water(x[0]));)
EKM_OP(OpRhFromQ, 3, 1, y[0] = m_div(T(100.0) * e_from_q(x[1], x[2]), es_mixed(x[0]));)
// thermo.py:559-663
EKM_OP(OpQFromTd, 2, 1, y[0] = q_from_e(es_water(x[0]), x[1], T(k::eps_default));)
EKM_OP(OpWFromTd, 2, 1, y[0] = w_from_e(es_water(x[0]), x[1], T(k::eps_default));)
EKM_OP(OpQFromRh, 3, 1, y[0] = q_from_e(x[1] * es_mixed(x[0]) * T(1.0 / 100.0), x[2], T(k::eps_default));)
// thermo.py:666-735
EKM_OP(OpTdFromRh, 2, 1, y[0] = t_from_es(es_water(x[0]) * x[1] * T(1.0 / 100.0));)
EKM_OP(OpTdFromQ, 2, 1, y[0] = t_from_es(e_from_q(x[0], x[1]));)
// thermo.py:738-920
EKM_OP(OpVirtualT, 2, 1, y[0] = virtual_t(x[0], x[1]);)
EKM_OP(OpVirtualTheta, 3, 1, y[0] = theta(x[0], x[2]) * m_fma(T(k::tv_c1), x[1], T(1));)
EKM_OP(OpTheta, 2, 1, y[0] = theta(x[0], x[1]);)
EKM_OP(OpTFromTheta, 2, 1, y[0] = t_from_theta(x[0], x[1]);)
EKM_OP(OpPOnDryAdiabat, 3, 1, y[0] = p_on_dry_adiabat(x[0], x[1], x[2]);)
EKM_OP(OpTOnDryAdiabat, 3, 1, y[0] = t_on_dry_adiabat(x[0], x[1], x[2]);)
// thermo.py:923-1000
EKM_OP_T1(OpLclT, METHOD, 2, 1, y[0] = lcl_t<METHOD>(x[0], x[1]);)
EKM_OP_T1(OpLcl, METHOD, 3, 2, const T tl = lcl_t<METHOD>(x[0], x[1]); y[0] = tl;
          y[1] = p_on_dry_adiabat(tl, x[0], x[2]);)
// thermo.py:1326-1469
EKM_OP_T1(OpEptFromTd, METHOD, 3, 1, y[0] = (ept<METHOD, false>(x[0], x[1], x[2]));)
EKM_OP_T1(OpEptFromQ, METHOD, 3, 1, y[0] = (ept<METHOD, true>(x[0], x[1], x[2]));)
EKM_OP_T1(OpSatEpt, METHOD, 2, 1, y[0] = ept_sat<METHOD>(x[0], x[1]);)
// thermo.py:1472-1590
EKM_OP_T2(OpTOnMa, METHOD, TM, 2, 1, y[0] = (t_on_ma<METHOD, TM>(x[0], x[1], tie));)
EKM_OP_T2(OpWetBulbFromTd, METHOD, TM, 3, 1,
          y[0] = (t_on_ma<METHOD, TM>(ept<METHOD, false>(x[0], x[1], x[2]), x[2], tie));)
EKM_OP_T2(OpWetBulbFromQ, METHOD, TM, 3, 1,
          y[0] = (t_on_ma<METHOD, TM>(ept<METHOD, true>(x[0], x[1], x[2]), x[2], tie));)
// The configuration BASELINE.json names (wet-bulb from q, IFS theta_e, Newton): the chain
// thermo.py:1589-1590 with theta_e never formed -- te = theta_e*(p/p0)^kappa = t*exp(K0*q/t_lcl)
// because the two pressure powers cancel, which drops one exp2 and the theta products.
template <>
struct OpWetBulbFromQ<EPT_IFS, T_NEWTON> {
  static constexpr int NIN = 3;
  static constexpr int NOUT = 1;
  template <class T>
  EKM_HD static void apply(const T* __restrict__ x, T* __restrict__ y, T rp) {
    TieInline tie;
    apply_tie(x, y, rp, tie);
  }
  template <class T, class Tie>
  EKM_HD static void apply_tie(const T* __restrict__ x, T* __restrict__ y, T, Tie& tie) {
    const T t = x[0], q = x[1], p = x[2];
    const T td = t_from_es(e_from_q(q, p));
    const T tl = lcl_t<LCL_DAVIES>(t, td);
    const T xe = T(k::K0_ifs * k::LOG2E) * q * m_rcp(tl);  // log2 of exp(K0*q/t_lcl)
    const T lte = m_log2(t * T(1.0 / 273.16)) + xe;        // log2(te/273.16) without forming te
    const TeFromTQP<T> exact{t, q, p};
    y[0] = t_on_ma_newton_ifs_core(
        lte, p, m_fma(T(0.1859e-5), p, T(0.6512)), [&] { return t * m_exp2(xe); },
        [&] { return m_exp2(T(k::kappa) * m_log2(p * T(1.0 / k::p0))); }, exact, tie);
  }
};

// ---- ops whose result depends on a Davies-Jones regime decision (thermo_math.hpp::davies_regime) ----------
// For these the fp32 map kernels run the points with TieFlag (branch-free) and re-run, with TieExact, the points
// of the rare lanes that recorded a tie.
template <class Op>
struct OpUsesTie {
  static constexpr bool value = false;
};
template <int M>
struct OpUsesTie<OpTOnMa<M, T_NEWTON>> {
  static constexpr bool value = true;
};
template <int M>
struct OpUsesTie<OpWetBulbFromTd<M, T_NEWTON>> {
  static constexpr bool value = true;
};
template <int M>
struct OpUsesTie<OpWetBulbFromQ<M, T_NEWTON>> {
  static constexpr bool value = true;
};

// ---- ops that keep a per-workgroup table in LDS -------------------------------------------------------
// OpTable<Op>::elems > 0: the map kernels reserve count<T>() values of LDS, call fill() once per workgroup
// (all threads, followed by a barrier) and then OpTable<Op>::apply(x, y, rp, table) instead of Op::apply.
// Op::apply itself stays the table-free statement of the same arithmetic (host twin, reference for tests).
template <class Op>
struct OpTable {
  static constexpr int elems = 0;
  static constexpr bool vectorized = false;
  template <class T>
  static constexpr int count() {
    return 0;
  }
};

// Bisection (the reference's DEFAULT t_method), all three theta_e methods.  fp32 walks the search tree in heap order ((es,
// a) pairs + log2 t, 48 KiB; thermo_math.hpp::t_on_ma_bisect_heap); fp64 walks the same fp32 tree for its sign tests with
// the lattice table of es in double behind it (80 KiB; t_on_ma_bisect_heap64).  An op supplies `prep`: the quantity the search
// inverts -- for ifs te = theta_e*(p/p0)^kappa, for the Bolton methods theta_e itself --, its logarithm
// lte = log2(./273.16) (LOG), the pressure and, for bolton35, kl = kappa*log2(p/p0).
template <int METHOD>
struct BisectTable {
  typedef BisectTable<METHOD> table_type;  // all ops of one theta_e method share one device-resident table
  static constexpr int elems = kBisectLattice;
  static constexpr int method = METHOD;
  template <class T>
  static constexpr int count() {  // fp32: the tree (thermo_math.hpp::heap_rec floats per node); fp64: es lattice, then the fp32 tree
    return sizeof(T) == 4 ? heap_rec<METHOD, float>() * kHeapNodes : kBisectLattice + heap_rec<METHOD, double>() * kHeapNodes / 2;
  }
  template <class T>
  EKM_HD static void fill(T* __restrict__ tab, int tid, int nthreads) {
    for (int m = tid; m < kBisectLattice; m += nthreads) {
      if constexpr (sizeof(T) == 4) {
        bisect_heap_fill<METHOD, heap_rec<METHOD, float>()>(tab, m);
      } else {
        bisect_es_fill(tab, m);
        bisect_heap_fill<METHOD, heap_rec<METHOD, double>(), true>(reinterpret_cast<float*>(tab + kBisectLattice), m);
      }
    }
  }
};
template <class Derived, int NIN_, int METHOD>
struct BisectOp : BisectTable<METHOD> {
  static constexpr bool vectorized = true;  // map_kernel.hpp::apply_points hands over the V points of a lane together
  template <class T, int V>
  EKM_HD static void apply_v(const T (&x)[V][NIN_], T (&y)[V][1], T, const T* __restrict__ tab, bool all_exact = false) {
    T te[V], lte[V], p[V], kl[V], out[V];
#pragma unroll
    for (int j = 0; j < V; ++j) Derived::template prep<T, true>(x[j], te[j], lte[j], p[j], kl[j]);
    if constexpr (sizeof(T) == 4) {
#if defined(EKM_WALK_SPLIT)  // A/B (round 6, NEGATIVE: 3.08 -> 3.17 ms): the four points of a lane's chunk as two walks of two -- 57
      // registers instead of 64 and no recomputed thr0, but two LDS reads in flight per wave instead of four (profiles/r06_tree_walk.txt)
      if constexpr (V == 4) {
        T l2[2], t2[2], p2[2], k2[2], o2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            l2[j] = lte[2 * h + j];
            t2[j] = te[2 * h + j];
            p2[j] = p[2 * h + j];
            k2[j] = kl[2 * h + j];
          }
          t_on_ma_bisect_heap<METHOD, 2>(l2, t2, p2, k2, tab, o2, all_exact);
          out[2 * h] = o2[0];
          out[2 * h + 1] = o2[1];
        }
      } else
#endif
      t_on_ma_bisect_heap<METHOD, V>(lte, te, p, kl, tab, out, all_exact);
    }
    else  // double / fd64: sign tests in fp32 on the tree behind the fp64 lattice table, ambiguous steps in T
      t_on_ma_bisect_heap64<METHOD, T, V>(lte, te, p, kl, reinterpret_cast<const float*>(tab + kBisectLattice), tab, out, all_exact);
#pragma unroll
    for (int j = 0; j < V; ++j) y[j][0] = out[j];
  }
  template <class T>
  EKM_HD static void apply(const T* __restrict__ x, T* __restrict__ y, T rp, const T* __restrict__ tab, bool all_exact = false) {
    T xx[1][NIN_], yy[1][1];
#pragma unroll
    for (int i = 0; i < NIN_; ++i) xx[0][i] = x[i];
    apply_v<T, 1>(xx, yy, rp, tab, all_exact);
    y[0] = yy[0][0];
  }
};
// bolton35 / bolton39: theta_e as the reference forms it
template <int M, class T, bool LOG>
EKM_HD void bolton_terms(T e, T p, T& lte, T& kl) {
  lte = LOG ? m_log2(e * T(1.0 / 273.16)) : T(0);
  kl = (LOG && M == EPT_BOLTON35) ? T(k::kappa) * m_log2(p * T(1.0 / k::p0)) : T(0);
}
template <int M>
struct OpTable<OpTOnMa<M, T_BISECT>> : BisectOp<OpTable<OpTOnMa<M, T_BISECT>>, 2, M> {
  template <class T, bool LOG>
  EKM_HD static void prep(const T* __restrict__ x, T& e, T& lte, T& p, T& kl) {
    e = x[0];
    p = x[1];
    bolton_terms<M, T, LOG>(e, p, lte, kl);
  }
};
template <int M>
struct OpTable<OpWetBulbFromTd<M, T_BISECT>> : BisectOp<OpTable<OpWetBulbFromTd<M, T_BISECT>>, 3, M> {
  template <class T, bool LOG>
  EKM_HD static void prep(const T* __restrict__ x, T& e, T& lte, T& p, T& kl) {
    e = ept<M, false>(x[0], x[1], x[2]);
    p = x[2];
    bolton_terms<M, T, LOG>(e, p, lte, kl);
  }
};
template <int M>
struct OpTable<OpWetBulbFromQ<M, T_BISECT>> : BisectOp<OpTable<OpWetBulbFromQ<M, T_BISECT>>, 3, M> {
  template <class T, bool LOG>
  EKM_HD static void prep(const T* __restrict__ x, T& e, T& lte, T& p, T& kl) {
    e = ept<M, true>(x[0], x[1], x[2]);
    p = x[2];
    bolton_terms<M, T, LOG>(e, p, lte, kl);
  }
};
// te from theta_e and p: te = theta_e*(p/p0)^kappa (thermo.py:1109-1110)
template <class T, bool LOG>
EKM_HD void ifs_te_from_ept(T e, T p, T& te, T& lte) {
  const T kl = T(k::kappa) * m_log2(p * T(1.0 / k::p0));
  te = e * m_exp2(kl);
  lte = LOG ? m_log2(e * T(1.0 / 273.16)) + kl : T(0);
}
// te from t, q and the LCL temperature: the pressure powers of theta_e (thermo.py:1169-1175) cancel, te = t*exp(K0*q/t_lcl)
template <class T, bool LOG>
EKM_HD void ifs_te_from_tq(T t, T q, T tl, T& te, T& lte) {
  const T xe = T(k::K0_ifs * k::LOG2E) * q * m_rcp(tl);
  te = t * m_exp2(xe);
  lte = LOG ? m_log2(t * T(1.0 / 273.16)) + xe : T(0);
}
template <>
struct OpTable<OpTOnMa<EPT_IFS, T_BISECT>> : BisectOp<OpTable<OpTOnMa<EPT_IFS, T_BISECT>>, 2, EPT_IFS> {
  template <class T, bool LOG>
  EKM_HD static void prep(const T* __restrict__ x, T& te, T& lte, T& p, T& kl) {
    kl = T(0);
    p = x[1];
    ifs_te_from_ept<T, LOG>(x[0], p, te, lte);
  }
};
template <>
struct OpTable<OpWetBulbFromTd<EPT_IFS, T_BISECT>> : BisectOp<OpTable<OpWetBulbFromTd<EPT_IFS, T_BISECT>>, 3, EPT_IFS> {
  template <class T, bool LOG>
  EKM_HD static void prep(const T* __restrict__ x, T& te, T& lte, T& p, T& kl) {
    kl = T(0);
    const T t = x[0], td = x[1];
    p = x[2];
    ifs_te_from_tq<T, LOG>(t, q_from_e(es_water(td), p, T(k::eps_default)), lcl_t<LCL_DAVIES>(t, td), te, lte);
  }
};
template <>
struct OpTable<OpWetBulbFromQ<EPT_IFS, T_BISECT>> : BisectOp<OpTable<OpWetBulbFromQ<EPT_IFS, T_BISECT>>, 3, EPT_IFS> {
  template <class T, bool LOG>
  EKM_HD static void prep(const T* __restrict__ x, T& te, T& lte, T& p, T& kl) {
    kl = T(0);
    // the search only needs te = theta_e*(p/p0)^kappa -- no log2 / exp2 of the pressure at all
    const T t = x[0], q = x[1];
    p = x[2];
    ifs_te_from_tq<T, LOG>(t, q, lcl_t<LCL_DAVIES>(t, t_from_es(e_from_q(q, p))), te, lte);
  }
};

// thermo.py:1593-1675: "direct" closed form, else the moist adiabat followed to p0
template <int METHOD, int TM, class T, class Tie>
EKM_HD T wbpt_from_ept(T e, Tie& tie) {
  if (TM == T_DIRECT) return wbpt_direct(e);
  return t_on_ma<METHOD, TM == T_DIRECT ? T_NEWTON : TM>(e, T(k::p0), tie);
}
EKM_OP_T2(OpWbptFromTd, METHOD, TM, 3, 1, y[0] = wbpt_from_ept<METHOD, TM>(ept<METHOD, false>(x[0], x[1], x[2]), tie);)
EKM_OP_T2(OpWbptFromQ, METHOD, TM, 3, 1, y[0] = wbpt_from_ept<METHOD, TM>(ept<METHOD, true>(x[0], x[1], x[2]), tie);)
template <int M>
struct OpTable<OpWbptFromTd<M, T_BISECT>> : BisectOp<OpTable<OpWbptFromTd<M, T_BISECT>>, 3, M> {
  template <class T, bool LOG>
  EKM_HD static void prep(const T* __restrict__ x, T& e, T& lte, T& p, T& kl) {
    e = ept<M, false>(x[0], x[1], x[2]);
    p = T(k::p0);
    bolton_terms<M, T, LOG>(e, p, lte, kl);
  }
};
template <int M>
struct OpTable<OpWbptFromQ<M, T_BISECT>> : BisectOp<OpTable<OpWbptFromQ<M, T_BISECT>>, 3, M> {
  template <class T, bool LOG>
  EKM_HD static void prep(const T* __restrict__ x, T& e, T& lte, T& p, T& kl) {
    e = ept<M, true>(x[0], x[1], x[2]);
    p = T(k::p0);
    bolton_terms<M, T, LOG>(e, p, lte, kl);
  }
};
template <>
struct OpTable<OpWbptFromTd<EPT_IFS, T_BISECT>> : BisectOp<OpTable<OpWbptFromTd<EPT_IFS, T_BISECT>>, 3, EPT_IFS> {
  template <class T, bool LOG>
  EKM_HD static void prep(const T* __restrict__ x, T& te, T& lte, T& p, T& kl) {
    kl = T(0);
    p = T(k::p0);  // (p/p0)^kappa = 1 at p0: te = theta_e
    te = ept<EPT_IFS, false>(x[0], x[1], x[2]);
    lte = LOG ? m_log2(te * T(1.0 / 273.16)) : T(0);
  }
};
template <>
struct OpTable<OpWbptFromQ<EPT_IFS, T_BISECT>> : BisectOp<OpTable<OpWbptFromQ<EPT_IFS, T_BISECT>>, 3, EPT_IFS> {
  template <class T, bool LOG>
  EKM_HD static void prep(const T* __restrict__ x, T& te, T& lte, T& p, T& kl) {
    kl = T(0);
    p = T(k::p0);
    te = ept<EPT_IFS, true>(x[0], x[1], x[2]);
    lte = LOG ? m_log2(te * T(1.0 / 273.16)) : T(0);
  }
};
template <int M>
struct OpUsesTie<OpWbptFromTd<M, T_NEWTON>> {
  static constexpr bool value = true;
};
template <int M>
struct OpUsesTie<OpWbptFromQ<M, T_NEWTON>> {
  static constexpr bool value = true;
};
// every bisection entry point runs on an LDS table (without its OpTable an op silently takes the table-free statement,
// 2x slower and otherwise indistinguishable)
static_assert(OpTable<OpTOnMa<EPT_IFS, T_BISECT>>::elems > 0 && OpTable<OpTOnMa<EPT_BOLTON35, T_BISECT>>::elems > 0 &&
                  OpTable<OpTOnMa<EPT_BOLTON39, T_BISECT>>::elems > 0 && OpTable<OpWetBulbFromTd<EPT_IFS, T_BISECT>>::elems > 0 &&
                  OpTable<OpWetBulbFromTd<EPT_BOLTON35, T_BISECT>>::elems > 0 && OpTable<OpWetBulbFromTd<EPT_BOLTON39, T_BISECT>>::elems > 0 &&
                  OpTable<OpWetBulbFromQ<EPT_IFS, T_BISECT>>::elems > 0 && OpTable<OpWetBulbFromQ<EPT_BOLTON35, T_BISECT>>::elems > 0 &&
                  OpTable<OpWetBulbFromQ<EPT_BOLTON39, T_BISECT>>::elems > 0 && OpTable<OpWbptFromTd<EPT_IFS, T_BISECT>>::elems > 0 &&
                  OpTable<OpWbptFromTd<EPT_BOLTON35, T_BISECT>>::elems > 0 && OpTable<OpWbptFromTd<EPT_BOLTON39, T_BISECT>>::elems > 0 &&
                  OpTable<OpWbptFromQ<EPT_IFS, T_BISECT>>::elems > 0 && OpTable<OpWbptFromQ<EPT_BOLTON35, T_BISECT>>::elems > 0 &&
                  OpTable<OpWbptFromQ<EPT_BOLTON39, T_BISECT>>::elems > 0,
              "a bisection op lost its lookup table");
static_assert(OpTable<OpWetBulbFromQ<EPT_IFS, T_BISECT>>::vectorized && OpTable<OpWetBulbFromQ<EPT_BOLTON35, T_BISECT>>::vectorized &&
                  OpTable<OpWetBulbFromQ<EPT_IFS, T_NEWTON>>::elems == 0,
              "table traits");

// thermo.py:1678-1707
EKM_OP(OpGasConstant, 1, 1, y[0] = m_fma(T(k::Rv - k::Rd), x[0], T(k::Rd));)

// wind/array/wind.py:192-222: hydrostatic vertical velocity w = (-Rd/g) * (omega*t/p)
EKM_OP(OpWFromOmega, 3, 1, y[0] = T(-k::Rd / k::g) * m_div(x[0] * x[1], x[2]);)

// Fused compositions: one read of (t, q, p),
// one write per output field.
// P3: es = saturation_vapour_pressure(t); td = dewpoint_from_specific_humidity(q, p);
//     rh = relative_humidity_from_specific_humidity(t, q, p)
EKM_OP(OpPipelineSvpTdRh, 3, 3, const T es = es_mixed(x[0]); const T e = e_from_q(x[1], x[2]); y[0] = es;
       y[1] = t_from_es(e); y[2] = m_div(T(100.0) * e, es);)

// P5: theta, es, rh, td, theta_e(ifs), tw(ifs, newton); sub-expressions shared:
// e(q,p) feeds td and rh, td feeds the LCL, theta feeds theta_e, theta_e feeds tw.
struct OpPipelineFull {
  static constexpr int NIN = 3;
  static constexpr int NOUT = 6;
  template <class T>
  EKM_HD static void apply(const T* __restrict__ x, T* __restrict__ y, T rp) {
    TieInline tie;
    apply_tie(x, y, rp, tie);
  }
  template <class T, class Tie>
  EKM_HD static void apply_tie(const T* __restrict__ x, T* __restrict__ y, T, Tie& tie) {
    const T t = x[0], q = x[1], p = x[2];
    const PTerms<T> P = pterms(p);
    const T th = t * P.thf;                           // thermo.py:829
    const T es = es_mixed(t);                         // es_comp.py:141-166
    const T e = e_from_q(q, p);                       // thermo.py:130-131
    const T td = t_from_es(e);                        // es_comp.py:128-130
    const T tl = lcl_t<LCL_DAVIES>(t, td);            // thermo.py:961
    const T xe = T(k::K0_ifs * k::LOG2E) * q * m_rcp(tl);
    const T ex = m_exp2(xe);
    const T the = th * ex;                            // thermo.py:1175
    y[0] = th;
    y[1] = es;
    y[2] = T(100.0) * e * m_rcp(es);                  // thermo.py:556
    y[3] = td;
    y[4] = the;
#ifdef EKM_P5_NOWB  // diagnostic build: how long do the nine streams take with the wet-bulb arithmetic removed?
    y[5] = the + P.l;
#else
    // thermo.py:1081-1159; te = theta_e*(p/p0)^kappa = t*exp(K0*q/t_lcl): the pressure powers cancel
    const TeFromTQP<T> exact{t, q, p};
    y[5] = t_on_ma_newton_ifs_core(m_log2(t * T(1.0 / 273.16)) + xe, p, P.dinv, [&] { return t * ex; },
                                   [&] { return m_exp2(T(k::kappa) * P.l); }, exact, tie);
#endif
  }
};

template <>
struct OpUsesTie<OpPipelineFull> {
  static constexpr bool value = true;
};

// ---- fp64 two-pass kernels: which non-finite outputs of the fast pass need the plain pass ---------------------------
// The fast pass (fdouble) poisons to NaN where a plain primitive would have fixed up an IEEE special operand, so a point
// with a non-finite fast output is redone in plain double (map_kernel.hpp::apply_points).  NOT needed where the output
// is non-finite because an INPUT it depends on is NaN -- a missing value, a masked land / sea point: NaN goes through
// every formula of thermo_math.hpp (none drops it in a select or a min / max), so the plain pass returns NaN there too
// and the fast pass's NaN stands.  Without this rule every masked point sent its whole wave through the serial plain
// pass (fp64 kernels on a masked field cost more than twice a clean one; tools/f64_nan_rate.py).
// OpDeps<Op>::of(o): bit i set = output o depends on input i.  Default: every output on every input.
template <class Op>
struct OpDeps {
  static constexpr unsigned of(int) { return (1u << Op::NIN) - 1u; }
};
template <>
struct OpDeps<OpPipelineSvpTdRh> {  // inputs (t, q, p): es(t), td(q, p), rh(t, q, p)
  static constexpr unsigned of(int o) { return o == 0 ? 1u : o == 1 ? 6u : 7u; }
};
template <>
struct OpDeps<OpPipelineFull> {  // theta(t, p), es(t), rh(t, q, p), td(q, p), theta_e(t, q, p), tw(t, q, p)
  static constexpr unsigned of(int o) { return o == 0 ? 5u : o == 1 ? 1u : o == 3 ? 6u : 7u; }
};
template <int M>
struct OpDeps<OpLcl<M>> {  // inputs (t, td, p): t_lcl(t, td), p_lcl(t, td, p)
  static constexpr unsigned of(int o) { return o == 0 ? 3u : 7u; }
};

// one point: is there a non-finite output that no NaN input explains?  (x, y: the point's inputs and fast-pass outputs)
template <class Op>
EKM_HD bool two_pass_redo_needed(const double* __restrict__ x, const double* __restrict__ y) {
  unsigned nan_in = 0u;
#pragma unroll
  for (int i = 0; i < Op::NIN; ++i) nan_in |= (x[i] != x[i] ? 1u : 0u) << i;
  bool need = false;
#pragma unroll
  for (int o = 0; o < Op::NOUT; ++o) need = need || (!__builtin_isfinite(y[o]) && !(nan_in & OpDeps<Op>::of(o)));
  return need;
}

// Threads per workgroup of an op's map kernels, and the waves per SIMD its kernels are compiled for.  The fp32 IFS
// bisection keeps a 48-KiB search tree in LDS: 512 threads share one copy, so that three workgroups = 24 waves fit a CU
// (with 256 threads three workgroups were 12 waves, and the search -- a chain of dependent LDS reads -- ran
// latency-bound at 4.1 ms where the same instruction stream at five workgroups of a 32-KiB table took 3.4;
// profiles/r04_bisect_tree_walk.txt), and its kernels are held to the 80 registers six waves per SIMD allow.
template <class Op, bool TABLE = (OpTable<Op>::elems > 0)>
struct OpTreeMethod {
  static constexpr int value = -1;
};
template <class Op>
struct OpTreeMethod<Op, true> {
  static constexpr int value = OpTable<Op>::method;
};
template <class Op, class T>
struct OpThreads {
  static constexpr bool tree = OpTable<Op>::elems > 0 && OpTable<Op>::vectorized;
  // the fp32 IFS walk with 16-B records: 64 KiB, two 1024-thread workgroups = eight waves per SIMD (<= 64 registers)
  static constexpr bool wide = tree && sizeof(T) == 4 && OpTreeMethod<Op>::value == EPT_IFS && heap_rec<EPT_IFS, float>() == 4;
#if defined(EKM_WALK_V8)
  static constexpr int value = wide ? 512 : tree ? EKM_TREE_THREADS : EKM_THREADS_DEFAULT;  // A/B: map_kernel.hpp::map_fields
#else
  static constexpr int value = wide ? 1024 : tree ? EKM_TREE_THREADS : EKM_THREADS_DEFAULT;
#endif
  // else fp32: 48 KiB, three workgroups = six waves per SIMD (<= 80 registers); fp64: 80 KiB, two workgroups = four waves
#if defined(EKM_WALK_V8)
  static constexpr int field_waves = wide ? 4 : tree ? (sizeof(T) == 4 ? EKM_TREE_WAVES : 4) : EKM_WAVES_PER_EU;
#else
  static constexpr int field_waves = wide ? 8 : tree ? (sizeof(T) == 4 ? EKM_TREE_WAVES : 4) : EKM_WAVES_PER_EU;
#endif
};

// Waves per SIMD a kernel of this op should be compiled for (launch bounds: caps the register allocation).
// The six-output pipeline is held to the 96 VGPRs that allow a fifth wave per SIMD in the per-level kernels: it is
// HBM-bound there, so the extra wave in flight is worth more than the registers (hybrid 5.35 -> 5.07 ms).  Left alone
// the compiler takes 97-108 -- round 3's builds met the cap with 16-48 B of scratch per lane; since round 4 the
// per-level kernels fit (87-94 VGPRs, no scratch: map_kernel.hpp::map_levels::run_points, tests/test_isa_resources.py).
template <class Op>
struct OpWaves {
  static constexpr int value = EKM_WAVES_PER_EU_DEFAULT;
};
template <>
struct OpWaves<OpPipelineFull> {
  static constexpr int value = EKM_P5_WAVES;
};

}  // namespace ekm
