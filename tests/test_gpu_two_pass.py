"""fp64 on the GPU: the kernels' two passes (fdouble first pass without IEEE special-operand fix-ups, plain double for the
lanes it poisoned; csrc/map_kernel.hpp::apply_points) against the SAME kernels with every lane forced through the plain
pass (tuning parameter f64_plain) -- bit for bit, on operands chosen to hit every fix-up (zeros, infinities, NaN,
negative, denormal, huge, the formulas' thresholds), for every function and variant.  The CPU twin of this test is
tests/test_hosttwin_two_pass.py."""
import numpy as np
import pytest

from test_hosttwin_two_pass import CASES, IDS, adversarial_args, assert_bit_equal

pytestmark = pytest.mark.gpu
np.seterr(all="ignore")


@pytest.fixture()
def plain_switch(ek):
    from ekm_hip import _ffi

    lib = _ffi.lib()

    def set_plain(on):
        _ffi.check(lib.ekm_set_tuning_param(b"f64_plain", 1 if on else 0))

    yield set_plain
    set_plain(False)


@pytest.mark.parametrize("func,argnames,kwargs", CASES, ids=IDS)
def test_two_pass_equals_plain_double_on_the_gpu(ek, plain_switch, func, argnames, kwargs):
    args = adversarial_args(func, argnames)
    fn = getattr(ek.thermo, func)
    plain_switch(True)
    plain = fn(*[a.copy() for a in args], **kwargs)
    plain_switch(False)
    two = fn(*[a.copy() for a in args], **kwargs)
    assert_bit_equal(func, kwargs, args, plain, two)


def test_two_pass_equals_plain_double_on_level_and_hybrid_kernels(ek, plain_switch):
    """The per-level kernels (pressure as a level vector / hybrid definition) call the same apply_points."""
    from ekm_hip import thermo
    from ekm_hip.vertical import hybrid_level_parameters

    rng = np.random.default_rng(5)
    nlev, inner = 137, 2048
    A, B = hybrid_level_parameters(137, model="ifs")
    sp = 101325.0 * (1.0 - 0.3 * rng.random(inner) ** 3)
    sp[::97] = np.nan
    sp[5::211] = 0.0
    t = rng.uniform(190.0, 315.0, (nlev, inner))
    q = 10.0 ** rng.uniform(-6.5, -1.7, (nlev, inner))
    for arr, vals in ((t, (0.0, np.inf, np.nan, -5.0, 32.19)), (q, (0.0, -1e-5, np.nan, np.inf, 1e300))):
        idx = rng.random(arr.shape) < 0.02
        arr[idx] = rng.choice(vals, int(idx.sum()))
    plev = np.linspace(1.0, 101325.0, nlev)[:, None]
    for p in (plev, ek.HybridPressure(A, B, sp)):
        plain_switch(True)
        plain = thermo.pipeline_full(t, q, p)
        plain_switch(False)
        two = thermo.pipeline_full(t, q, p)
        assert_bit_equal("pipeline_full", {}, (t.ravel(), q.ravel()), tuple(x.ravel() for x in plain), tuple(x.ravel() for x in two))
