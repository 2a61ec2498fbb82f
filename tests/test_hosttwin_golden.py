"""The per-point math the gfx950 kernels execute (csrc/thermo_math.hpp, ops.hpp),
compiled for the host, against the vectors recorded from the reference.
fp64: 1e-6 relative; fp32: 1e-4 relative against the reference's fp32 output."""
import os

import numpy as np
import pytest

import _hosttwin
from oracle import thermo_oracle as orc
from _compare import assert_parity, bisect_sign_noise, bisect_unstable, newton_regime_boundary
from _golden import case_inputs, case_outputs, golden, manifest

CASES = manifest()
pytestmark = pytest.mark.skipif(not os.path.exists(_hosttwin.PATH), reason="host twin not built (run make)")


@pytest.mark.parametrize("case", CASES, ids=[c["id"] for c in CASES])
def test_kernel_math_vs_reference(case):
    dtype = np.float32 if case["dtype"] == "f32" else np.float64
    out = _hosttwin.by_reference_name(case["func"], case_inputs(case), case["kwargs"], dtype)
    outs = out if isinstance(out, tuple) else (out,)
    bisect = case["kwargs"].get("t_method") == "bisect"
    cid = case["id"].split(".")
    for i, (o, g) in enumerate(zip(outs, case_outputs(case))):
        both = [golden()[".".join([cid[0], tag] + cid[2:]) + f".out{i}"] for tag in ("f32", "f64")]
        unstable = ref64 = noise_t = None
        if bisect:
            noisy, noise_t = bisect_sign_noise(orc, case["func"], case_inputs(case), case["kwargs"],
                                               3e-6 if case["dtype"] == "f32" else 1e-14, return_points=True)
            unstable = bisect_unstable(*both) | noisy
            if case["dtype"] == "f32":
                ref64 = both[1]  # unstable points may sit with the fp64 reference (NaN-ness, 2 quanta)
        elif "newton" in case["id"]:
            unstable = newton_regime_boundary(case["func"], case_inputs(case), case["kwargs"],
                                              1e-5 if case["dtype"] == "f32" else 1e-13)
            if case["dtype"] == "f32":
                ref64 = both[1]  # the reference's own fp64 answer: conditioning yardstick for the Newton step
        assert_parity(o, g, case["dtype"], case["id"], bisect=bisect, unstable=unstable, ref64=ref64, noise_t=noise_t)
