"""Where the reference's own algorithm takes rounding-determined decisions -- TEST INFRASTRUCTURE.

Two selectors of the moist-adiabat inversion are discontinuous in their inputs:

* bisection (thermo.py:1055-1079): ``sign(residual)`` at each of 12 lattice points;
* Davies-Jones Newton (thermo.py:1114-1128): the initial-guess regime chosen by comparing
  ``c_te`` with ``D(p)``, 1 and 0.4.  At 10 hPa the regime-1 and regime-2 guesses differ by ~2 K and
  one Newton step leaves ~0.17 K (7e-4) of that, so a point whose ``c_te`` is within rounding of a
  threshold legitimately lands on either side in any fp32 implementation, the reference's included.

The masks below identify such points from the oracle evaluated in fp64 -- never from the output
under test -- so that parity checks can exclude and count them.
"""
import numpy as np

from . import thermo_oracle as orc


def _ept_and_p(func, inputs, kwargs):
    m = kwargs.get("ept_method", "ifs")
    a = [np.asarray(x, dtype=np.float64) for x in inputs]
    if func in ("temperature_on_moist_adiabat",):
        ept, p = a
    elif func in ("pipeline_full",):
        ept, p = orc.ept_from_specific_humidity(a[0], a[1], a[2], method="ifs"), a[2]
    else:
        f = orc.ept_from_dewpoint if "dewpoint" in func else orc.ept_from_specific_humidity
        ept = f(a[0], a[1], a[2], method=m)
        p = a[2] if "potential" not in func else np.full_like(ept, orc.p0)
    ept, p = np.broadcast_arrays(ept, p)
    return ept.ravel().copy(), p.ravel().copy(), m


def bisect_sign_noise(func, inputs, kwargs, thresh, return_points=False):
    """Points whose bisection meets a residual |ept*exp(-G) - th_sat| <= thresh*th_sat on the
    oracle's own fp64 path (e.g. exactly saturated input, where the residual at the second lattice
    point is mathematically zero): ``sign()`` is noise there and one early flip can end in the
    ``p - es < eps`` NaN region.  With `return_points` also the lattice temperatures at which that
    happens, shape (n, 12), NaN where the step is decided cleanly: a search whose sign flips at such
    a point ends within one quantum of it (it converges back onto it from the other side)."""
    with np.errstate(all="ignore"):
        # the arithmetic under test: where one of the two terms' exponential factors -- exp(G_sat(-1)), th_sat/t --
        # lands within a few quanta of the dtype's smallest subnormal while the other term is subnormal too, whether
        # that factor rounds to zero or to one quantum (a 1e-6 difference in es moves bolton35's exponent by 0.3 where
        # p - es is a fraction of a pascal) decides between "0 - 0: stay on this node" and "move on": the reference's
        # own rounding (t 316.68 K, q 1.083e-4, p 7393.3145 Pa, bolton35: th_sat = 313.16 * 2^-149.9)
        in_dt = np.result_type(*[np.asarray(x).dtype for x in inputs])
        fin = np.finfo(in_dt if in_dt.kind == "f" else np.float64)
        sub, tiny = float(fin.smallest_subnormal), float(fin.tiny)
        ept, p, m = _ept_and_p(func, inputs, kwargs)
        meth = orc._EPT[m]
        t = np.full(ept.size, orc.T0 - 20.0)
        dt = 120.0
        noisy = np.zeros(ept.size, dtype=bool)
        pts = np.full((ept.size, 12), np.nan) if return_points else None
        for it in range(12):
            st = orc._state(t=t, p=p)
            dt /= 2.0
            g = meth["gsat"](st, scale=-1.0)
            th = meth["thsat"](st)
            fa = np.exp(g)
            r = ept * fa - th
            here = np.abs(r) <= thresh * np.abs(th)
            # ... or by es itself: the residual's sign changes when es moves by `thresh` (the fp32 evaluations of es differ
            # by a few 1e-6; through exp(-G) that is G * d ln qs/d ln es times as much -- 2.5e-5 of the residual's terms at
            # 371 K and p0, where a theta_w search with theta_e = 2e5 K has r = +3.5e-6 * th_sat in the reference)
            dtn = thresh * orc.saturation_vapour_pressure(t) / orc.saturation_vapour_pressure_slope(t)
            for sg in (1.0, -1.0):
                s2 = orc._state(t=t + sg * dtn, p=p)
                r2 = ept * np.exp(meth["gsat"](s2, scale=-1.0)) - meth["thsat"](s2)
                here |= np.isfinite(r) & np.isfinite(r2) & (np.sign(r2) != np.sign(r))
            fb = th / t
            here |= ((fa > sub / 8) & (fa < 16 * sub) & (np.abs(th) < tiny)) | ((fb > sub / 8) & (fb < 16 * sub) & (np.abs(ept * fa) < tiny))
            noisy |= here
            if return_points:
                pts[here, it] = t[here]
            t = t + np.sign(r) * dt
    return (noisy, pts) if return_points else noisy


def bisect_nan_rule_noise(func, inputs, kwargs, thresh):
    """Points whose bisection visits a lattice temperature where the reference's NaN rule, `p - es < 1e-4 -> NaN`
    (thermo.py:192-196, 229-232, 1283-1284), is decided by the rounding of es itself: |(p - es) - 1e-4| <= thresh * es on
    the oracle's fp64 path (es at 313.16 K is 7384.176 Pa with an fp32 spacing of 4.9e-4 and an fp32 error of up to 0.02:
    a pressure within that of it turns the search NaN or not by the last bits of es).  Either outcome is the
    reference's there."""
    with np.errstate(all="ignore"):
        ept, p, m = _ept_and_p(func, inputs, kwargs)
        meth = orc._EPT[m]
        t = np.full(ept.size, orc.T0 - 20.0)
        dt = 120.0
        noisy = np.zeros(ept.size, dtype=bool)
        for _ in range(12):
            es = orc.saturation_vapour_pressure(t)
            noisy |= np.abs((p - es) - 1e-4) <= thresh * es
            st = orc._state(t=t, p=p)
            dt /= 2.0
            r = ept * np.exp(meth["gsat"](st, scale=-1.0)) - meth["thsat"](st)
            t = t + np.sign(r) * dt
    return noisy


def newton_regime_boundary(func, inputs, kwargs, thresh):
    """Points whose c_te lies within relative `thresh` of a regime threshold (D(p), 1, 0.4); a tuple of
    thresholds gives a tuple of masks (c_te is formed once)."""
    with np.errstate(all="ignore"):
        ept, p, _ = _ept_and_p(func, inputs, kwargs)
        pp = np.power(p / orc.p0, orc.kappa)
        c_te = np.power(273.16 / (ept * pp), orc.LAMBDA)
        d = 1.0 / (0.1859e-5 * p + 0.6512)
        dist = np.minimum(np.minimum(np.abs(c_te - d) / d, np.abs(c_te - 1.0)), np.abs(c_te - 0.4) / 0.4)
        if isinstance(thresh, (tuple, list)):
            return tuple(dist <= th for th in thresh)
        return dist <= thresh


def amplification(f, xs, h=1e-6, return_candidates=False):
    """How strongly `f` (a function of the oracle, evaluated in fp64) amplifies a relative perturbation of its inputs
    `xs`: kappa = max over the inputs of |f(x(1+h)) - f(x(1-h))| / (2 h |f(x)|); inf where a perturbed evaluation
    changes NaN-ness (the point sits on the edge of a NaN region or on a regime threshold).  Identified from the oracle
    alone.  `return_candidates`: also the 1 + 2*len(xs) fp64 values themselves: the unperturbed one and the perturbed."""
    with np.errstate(all="ignore"):
        x = [np.asarray(a, dtype=np.float64) for a in xs]
        base = f(*x)
        kappa = np.zeros(base.shape)
        cands = [base]
        for i in range(len(x)):
            lo, hi = list(x), list(x)
            lo[i], hi[i] = x[i] * (1.0 - h), x[i] * (1.0 + h)
            a = f(*lo)
            b = f(*hi)
            k = np.abs(b - a) / (2.0 * h * np.abs(base))
            k = np.where(np.isnan(a) | np.isnan(b) | np.isnan(base), np.inf, k)
            kappa = np.maximum(kappa, k)
            cands += [a, b]
    return (kappa, np.stack(cands)) if return_candidates else kappa


def _wb_ifs_newton(t, q, p):
    return orc.wet_bulb_temperature_from_specific_humidity(t, q, p, "ifs", "newton")


def newton_amplification(t, q, p, h=1e-6, return_candidates=False):
    """`amplification` of the reference's own one-step Newton wet-bulb (ifs) over (t, q, p).  At hPa-level pressures
    (p < ~60 Pa: es(tw) ~ p, qs ~ 1) the single Newton step is far from converged and kappa reaches 1e3-1e6: an fp32
    rounding of an intermediate (6e-8) then shows up at 1e-4 and beyond in ANY fp32 evaluation, the reference's own
    included."""
    return amplification(_wb_ifs_newton, (t, q, p), h, return_candidates)


def misses_explained(f, xs, got, want, tol, h=1e-6, unit=2.0 ** -24, factor=8.0):
    """`_misses_explained_at` at the perturbation scales h, h/3, h/10 and h/30: a POLE of the reference inside +-h (bolton35's
    one Newton step at p within 1e-7 of where its dlnf crosses zero: theta_e 436.42 K, p 99975.96 Pa gives 518.6 K, 751.0 K at
    p*(1 - 1e-7), NaN at p*(1 - 3e-7), 260.0 K at p*(1 - 1e-6)) hides from the +-h secant and from the span of its outcomes;
    the smaller scales -- still of the size of an fp32 rounding of the input -- see it.  A point is explained if any
    scale explains it."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    xs = [np.asarray(a, np.float64) for a in xs]
    finite = np.zeros(got.shape, bool)
    on_edge = np.zeros(got.shape, bool)
    for scale in (1.0, 1.0 / 3.0, 0.1, 1.0 / 30.0):
        todo = np.flatnonzero(~(finite | on_edge))
        if not todo.size:
            break
        fi, ed = _misses_explained_at(f, [a[todo] if a.ndim else a for a in xs], got[todo], want[todo], tol, h * scale, unit, factor)
        finite[todo] |= fi
        on_edge[todo] |= ed
    return finite, on_edge


def _misses_explained_at(f, xs, got, want, tol, h=1e-6, unit=2.0 ** -24, factor=8.0):
    """Which deviations of `got` from the reference's `want` (same dtype run) the reference's own conditioning explains
    -- from the fp64 oracle `f` alone, never from `got`.  Returns two masks over the points:
      * finite kappa: the deviation is within factor x kappa x unit (kappa: `amplification`; unit: the rounding unit of
        the arithmetic under test, 2^-24 for fp32; factor: how many such roundings may add up, 8 on the benchmark field);
      * on a NaN / regime edge (kappa = inf: a 1e-6 perturbation of an input changes the NaN-ness of the fp64 result): the
        value under test must BE one of the outcomes that edge offers -- NaN where the fp64 oracle or one of its
        perturbed evaluations is NaN, or a finite value inside the span [lo, hi] of the finite ones (the function is
        continuous on its finite side: every value in the span is the outcome of some perturbation within +-h), the
        span extended by its own width on both sides where the other side of the edge is NaN, and by `tol`.
    A point outside both masks is a real miss."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    with np.errstate(all="ignore"):
        nanmm = np.isnan(got) != np.isnan(want)
        r = np.abs(got - want) / np.abs(want)
        r = np.where(np.isfinite(r), r, 0.0)
        kap, cands = amplification(f, xs, h, return_candidates=True)
        edge = np.isinf(kap)
        finite = ~edge & ~nanmm & (r <= factor * np.where(edge, 0.0, kap) * unit)
        cn = np.isnan(cands)
        lo = np.min(np.where(cn, np.inf, cands), axis=0)
        hi = np.max(np.where(cn, -np.inf, cands), axis=0)
        some = np.isfinite(lo)
        # the finite outcomes of the edge span [lo, hi]; the function is continuous on its finite side, so every value in
        # between is the outcome of SOME perturbation within +-h -- and where the other side of the edge is NaN (the
        # tw <= 0 and p - es < eps masks cut the function off) the span is extended by its own width on both sides
        width = np.where(some & cn.any(axis=0), hi - lo, 0.0)
        lo_e, hi_e = lo - width, hi + width
        ok_range = some & (got >= lo_e - tol * np.abs(lo_e)) & (got <= hi_e + tol * np.abs(hi_e))
        on_edge = edge & np.where(np.isnan(got), cn.any(axis=0), ok_range)
        # ... and a DISCONTINUITY of the reference that is not a NaN edge: the Davies-Jones regime thresholds (c_te against
        # D(p), 1, 0.4; thermo.py:1114-1128), across which the guess jumps -- by 25 K at 70 Pa.  kappa is finite but meaningless
        # there (a secant through the jump); the value under test is explained if it lies inside the span of the outcomes
        # that the +-h perturbations produce (no extension: both sides are finite), e.g. it IS the other regime's result.
        in_span = ~nanmm & some & ~cn.any(axis=0) & (got >= lo - tol * np.abs(lo)) & (got <= hi + tol * np.abs(hi))
        on_edge = on_edge | in_span
    return finite, on_edge


def newton_misses_explained(t, q, p, got, want, tol, h=1e-6):
    """`misses_explained` for the one-step Newton wet-bulb (ifs) from (t, q, p)."""
    return misses_explained(_wb_ifs_newton, (t, q, p), got, want, tol, h)
