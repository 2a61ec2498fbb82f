"""Parity bars shared by the CPU and GPU tests.

north_star: results must match the reference NumPy path within 1e-6 relative in
fp64 and 1e-4 relative in fp32 on the same inputs, with the same NaN pattern.
The fp32 comparison is always against the fp32 reference/oracle output.

Bisection (the reference's default `t_method`) is a 12-step sign search whose
result is quantised to 120/4096 K; a rounding-level difference in the residual
flips one `sign()` and moves the answer by up to two quanta (0.0586 K ~ 2e-4
relative).  The reference's own tests use rtol=1e-3 for wet-bulb for this
reason (tests/thermo/test_thermo.py:802 there).  For bisect outputs the bar is
therefore: every point within 2 quanta, and at most `BISECT_FLIP_LIMIT[dtype]`
of the points the reference decides stably off the reference's lattice value at
all (budgets re-based in round 6 on what the MI355X runs actually used, see
`bisect_flip_allowed`).

At exactly saturated points (t == tw) the residual of the very first steps is
~0 and its sign is pure rounding noise; one early flip can send the search into
the p - es < eps region where it turns NaN for good.  The reference's own fp32
and fp64 outputs disagree there (NaN vs 313.16 K on row 431 of its 480-row
table).  Such reference-unstable points -- identified from the reference's own
fp32-vs-fp64 disagreement and from its own residual being rounding noise at a
visited lattice point, never from our output -- are NOT excluded (round 2 did):
they are checked against a wider, reference-defined anchor set (`_assert_bisect`:
a NaN only where the fp32 or the fp64 reference has one; a finite value within 2
quanta of one of them or of a noise point of the reference's own residual) and
counted in the ledger.
"""
import numpy as np

RTOL = {"f64": 1e-6, "f32": 1e-4}
# fp64: the bar is 1e-6; the primitives are built to ~2e-9 (csrc/thermo_math.hpp) and every fp64 assertion of the GPU suite is this
F64_ASSERT = 1e-7
BISECT_QUANTUM = 120.0 / 4096.0
# Stable bisect points that may sit on another lattice value than the reference's.  Measured (round 5, 178 checks on the
# MI355X, 32,344,340 stable points): ONE point off, in fp64 (bolton35 fuzz, 1 of 1,048,576); the host twin: 0 of
# 14,193,996.  Round 5's limit was a flat 2 % -- ten thousand times that use; VERDICT r5 showed a run with 1 % of the
# points moved by two quanta passing.  Now: a fraction of the points per dtype, nothing at all below 100,000 points.
BISECT_FLIP_LIMIT = {"f32": 1e-5, "f64": 2e-6}
BISECT_FLIP_MIN_POINTS = 100_000
# the share of ALL points (stable or not) that must carry the reference's very bits on a non-adversarial draw
# (measured: 99.987-99.994 % in fp32, 100 % in fp64 on 400 k points; the suites: >= 99.98 %)
BISECT_MIN_IDENTICAL = 0.999


def bisect_flip_allowed(tag, n):
    """How many of n stable points may differ from the reference's lattice value (by one or two quanta)."""
    return int(np.ceil(BISECT_FLIP_LIMIT[tag] * n)) if n >= BISECT_FLIP_MIN_POINTS else 0


def rel_err(got, want):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    with np.errstate(all="ignore"):
        d = np.abs(got - want)
        r = np.where(d == 0, 0.0, d / np.abs(want))
    r = np.where(np.isfinite(want) & np.isfinite(got), r, 0.0)
    return r


def assert_same_nonfinite(got, want, what):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} != {want.shape}"
    nan_g, nan_w = np.isnan(got), np.isnan(want)
    assert np.array_equal(nan_g, nan_w), (
        f"{what}: NaN pattern differs at {np.flatnonzero(nan_g != nan_w)[:8]} "
        f"got={got.ravel()[np.flatnonzero(nan_g != nan_w)[:4]]} want={want.ravel()[np.flatnonzero(nan_g != nan_w)[:4]]}")
    inf_w = np.isinf(want)
    assert np.array_equal(np.isinf(got), inf_w), f"{what}: inf pattern differs"
    assert np.array_equal(got[inf_w], want[inf_w]), f"{what}: inf signs differ"


def bisect_unstable(ref32, ref64):
    """Points where the reference disagrees with itself across precisions: NaN-ness, or a
    different sign sequence (results half a quantum or more apart)."""
    a = np.asarray(ref32, dtype=np.float64)
    b = np.asarray(ref64, dtype=np.float64)
    with np.errstate(all="ignore"):
        return (np.isnan(a) != np.isnan(b)) | (np.abs(a - b) > 0.5 * BISECT_QUANTUM)


def bisect_sign_noise(orc, func, inputs, kwargs, thresh, return_points=False):
    """See oracle/conditioning.py (shared with bench.py)."""
    from oracle import conditioning

    return conditioning.bisect_sign_noise(func, inputs, kwargs, thresh, return_points)


def newton_regime_boundary(func, inputs, kwargs, thresh):
    """See oracle/conditioning.py: points whose Davies-Jones regime is decided by rounding."""
    from oracle import conditioning

    return conditioning.newton_regime_boundary(func, inputs, kwargs, thresh)


# ---- budget ledger: every relaxation a test uses is recorded and printed at the end of the run --------
LEDGER = []  # (what, kind, used, allowed, n)
CENSUS = []  # one line per whole-field census (tests/test_gpu_census.py), printed at the end of the run

# Measured on MI355X (round 2, profiles/r02_parity_budgets.txt), limits set to ~2x the largest use seen:
BISECT_UNSTABLE_FRACTION = 0.08   # reference-unstable bisect points; largest seen 5.4 % (the reference's 480-row table: 1/6 of its rows are exactly saturated)
BOUNDARY_VIOLATION_FRACTION = 0.02  # of the regime-boundary points, how many may miss the bar (the fp32 reference's own flips); measured: 0 of 28,807 on the benchmark field
ILL_CONDITIONED_FRACTION = 1e-4   # Bolton Newton points needing max(rtol, 4*delta)


def _record(what, kind, used, allowed, n):
    LEDGER.append((what, kind, int(used), float(allowed), int(n)))


def assert_parity(got, want, tag, what, bisect=False, rtol=None, unstable=None, ref64=None, max_relaxed=None, noise_t=None):
    """`unstable`: points where the REFERENCE's own algorithm takes a rounding-determined decision, always
    identified from the reference (the oracle in fp64), never from the output under test.
      * bisect: excluded and counted (<= BISECT_UNSTABLE_FRACTION of the points, at least 3);
      * Newton (Davies-Jones regime boundaries): still COMPARED -- the kernels settle near-threshold lanes in
        double in the reference's operator order -- but up to BOUNDARY_VIOLATION_FRACTION of them (at least 1)
        may miss the bar (none when there are fewer than 50), which is what the fp32 reference's own rounding flips
        could amount to; measured use: 0 (profiles/r02_parity_budgets.txt).
    `ref64` (fp32 comparisons only): the reference's fp64 result on the same fp32 inputs.  Where the
    reference's own fp32 output sits delta away from it (an ill-conditioned point of the reference's
    algorithm, e.g. the Bolton-35 Newton step near p = p0), the bar there is max(rtol, 4*delta), and at
    most ILL_CONDITIONED_FRACTION of the points (at least 3) may need it."""
    rtol = RTOL[tag] if rtol is None else rtol
    boundary = None
    if bisect:
        return _assert_bisect(got, want, what, rtol, unstable, ref64, noise_t, tag)
    if unstable is not None and np.any(unstable):
        unstable = np.asarray(unstable).ravel()
        lim = max(3, 2e-4 * unstable.size)
        assert unstable.sum() <= lim, f"{what}: {unstable.sum()} regime-boundary points (limit {lim:.0f})"
        boundary = unstable
    assert_same_nonfinite(got, want, what)
    r = rel_err(got, want).ravel()
    if boundary is not None:  # compared, with a small allowance for the reference's own flips
        over = int((r[boundary] > rtol).sum())
        allowed = int(BOUNDARY_VIOLATION_FRACTION * boundary.sum())  # none for fewer than 50 boundary points
        _record(what, "newton: regime-boundary points beyond the bar", over, allowed, int(boundary.sum()))
        assert over <= allowed, (f"{what}: {over} of {int(boundary.sum())} regime-boundary points beyond {rtol:g} "
                                 f"(allowed {allowed}); worst {r[boundary].max():.3e}")
        r = r[~boundary]
        if ref64 is not None:
            ref64 = np.asarray(ref64).ravel()[~boundary]
            want = np.asarray(want).ravel()[~boundary]
    if ref64 is not None:
        bar = np.maximum(rtol, 4.0 * rel_err(want, ref64).ravel())
        relaxed = int((bar > rtol).sum())
        allowed = max(3, ILL_CONDITIONED_FRACTION * r.size) if max_relaxed is None else max_relaxed
        if relaxed:
            _record(what, "newton: ill-conditioned points at max(rtol, 4*delta)", relaxed, allowed, r.size)
        assert relaxed <= allowed, f"{what}: {relaxed} ill-conditioned points in the reference"
        bad = r > bar
        assert not bad.any(), f"{what}: rel err {r[bad].max():.3e} beyond max({rtol:g}, 4*delta) at {np.flatnonzero(bad)[:4]}"
        return float(r[bar <= rtol].max()) if (bar <= rtol).any() else 0.0
    worst = float(r.max()) if r.size else 0.0
    assert worst <= rtol, f"{what}: max rel err {worst:.3e} > {rtol:g} at {int(np.argmax(r))}"
    return worst


def _assert_bisect(got, want, what, rtol, unstable, ref64, noise_t=None, tag="f32"):
    """The 12-step sign search (the reference's default t_method), result quantised to 120/4096 K.

    Every point -- reference-unstable or not -- is CHECKED:
      * points the reference decides stably: identical NaN / inf pattern, within 2 quanta, at most
        bisect_flip_allowed(dtype, n) of them one or two quanta off (any difference from the reference's lattice value counts);
      * reference-unstable points (`unstable`: the reference's own residual is below rounding noise at some step, or
        its fp32 and fp64 runs disagree -- identified from the oracle, never from the output under test; counted,
        limit BISECT_UNSTABLE_FRACTION):
          - a NaN is accepted only where the fp32 OR the fp64 reference (`ref64`) has one;
          - a finite value only within 2 quanta of the fp32 reference, of the fp64 reference, or of a lattice
            temperature at which the reference's own residual is noise (`noise_t`, oracle/conditioning.py: a search
            whose sign flips there converges back onto that point from the other side).
        Where the second reference / the noise points are not supplied, what cannot be decided is counted in the
        ledger against its own limit instead of passing silently."""
    got = np.asarray(got, dtype=np.float64).ravel()
    want = np.asarray(want, dtype=np.float64).ravel()
    assert got.shape == want.shape, f"{what}: shape {got.shape} != {want.shape}"
    uns = np.zeros(got.size, bool) if unstable is None else np.asarray(unstable).ravel().copy()
    r64 = None if ref64 is None else np.asarray(ref64, dtype=np.float64).ravel()
    two = 2 * BISECT_QUANTUM * (1 + 1e-6)
    if uns.any():
        lim = max(3, (BISECT_UNSTABLE_FRACTION if uns.size <= 1000 else 0.5 * BISECT_UNSTABLE_FRACTION) * uns.size)
        _record(what, "bisect: reference-unstable points (still checked: 2 quanta / NaN of either reference)", uns.sum(), lim, uns.size)
        assert uns.sum() <= lim, f"{what}: {uns.sum()} reference-unstable points (limit {lim:.0f})"
    st = ~uns
    assert_same_nonfinite(got[st], want[st], what)
    with np.errstate(all="ignore"):
        d = np.abs(got - want)
        near = np.isfinite(got) & np.isfinite(want) & (d <= two)           # finite, beside the fp32 reference
        if r64 is not None:
            near |= np.isfinite(got) & np.isfinite(r64) & (np.abs(got - r64) <= two)
        if noise_t is not None:
            nt = np.asarray(noise_t, dtype=np.float64).reshape(got.size, -1)
            near |= np.isfinite(got) & (np.nanmin(np.where(np.isnan(nt), np.inf, np.abs(nt - got[:, None])), axis=1) <= two)
    bad = st & np.isfinite(got) & ~near
    assert not bad.any(), f"{what}: bisect off by {d[bad].max():.4f} K (> 2 quanta) at {np.flatnonzero(bad)[:4]}"
    # unstable points: NaN needs a reference NaN, a finite value needs one of the anchors above
    nan_ok = np.isnan(want) | (np.isnan(r64) if r64 is not None else False)
    bad_nan = uns & np.isnan(got) & ~nan_ok
    bad_fin = uns & np.isfinite(got) & ~near
    undecidable = 0
    if r64 is None:          # fp64 tests: no second reference for the NaN question
        undecidable += int(bad_nan.sum())
        bad_nan[:] = False
    if noise_t is None:      # no noise points supplied: a finite value where the reference(s) are NaN cannot be anchored
        loose = bad_fin & ~(np.isfinite(want) | (np.isfinite(r64) if r64 is not None else False))
        undecidable += int(loose.sum())
        bad_fin &= ~loose
    assert not bad_nan.any(), f"{what}: NaN at unstable points {np.flatnonzero(bad_nan)[:4]} where neither reference has one"
    assert not bad_fin.any(), (f"{what}: finite values at unstable points {np.flatnonzero(bad_fin)[:4]} "
                               f"({got[bad_fin][:4]}) more than 2 quanta from both references and from every noise point")
    if undecidable:
        lim = max(3, 0.5 * uns.sum())
        _record(what, "bisect: unstable points accepted without an anchor (no fp64 reference / noise points given)",
                undecidable, lim, int(uns.sum()))
        assert undecidable <= lim, f"{what}: {undecidable} unanchored differences on {uns.sum()} unstable points"
    r = rel_err(got, want)
    # a flip = a stable point on ANOTHER lattice value (results are lattice temperatures: half a quantum tells them apart;
    # the relative bar alone would miss a one-quantum flip above 293 K in fp32, 0.0293/300 < 1e-4)
    with np.errstate(all="ignore"):
        flips = int((st & np.isfinite(got) & np.isfinite(want) & (d > 0.5 * BISECT_QUANTUM)).sum())
    allowed = bisect_flip_allowed(tag, int(st.sum()))
    _record(what, "bisect: stable points one or two quanta off", flips, allowed, int(st.sum()))
    assert flips <= allowed, f"{what}: {flips}/{int(st.sum())} stable bisect points off the reference's lattice value (allowed {allowed})"
    return float(r[st].max()) if st.any() else 0.0
