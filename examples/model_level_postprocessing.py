#!/usr/bin/env python3
"""Model-level post-processing on an MI355X with the earthkit-meteo signatures.

Temperature and specific humidity on IFS hybrid levels + surface pressure and surface geopotential
-> pressure, relative humidity, dewpoint, theta_e, wet-bulb temperature and geopotential height,
everything resident in HBM between calls.  Run from the repository root:

    python examples/model_level_postprocessing.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "earthkit-meteo_amd"))

import ekm_hip  # noqa: E402
from ekm_hip import thermo, vertical  # noqa: E402


def main(nlat=181, nlon=360):
    # IFS L137 half-level coefficients (the same call as earthkit.meteo.vertical.array.hybrid_level_parameters)
    A, B = vertical.hybrid_level_parameters(137, model="ifs")
    rng = np.random.default_rng(0)
    sp = (101325.0 * (1.0 - 0.3 * rng.random((nlat, nlon)) ** 3)).astype(np.float32)
    zs = ((101325.0 - sp) / 1.2).astype(np.float32)   # g*z ~ dp / rho

    d_sp, d_zs = ekm_hip.to_device(sp), ekm_hip.to_device(zs)
    p = vertical.pressure_on_hybrid_levels(A.astype(np.float32), B.astype(np.float32), d_sp)   # [137, nlat, nlon] on the GPU
    ph = p.to_host()
    t = (np.maximum(288.15 * (ph / 101325.0) ** 0.190263, 216.65) + rng.normal(0, 5, ph.shape)).astype(np.float32)
    q = np.clip(0.006 * (ph / 101325.0) ** 3, 2e-6, None).astype(np.float32)
    d_t, d_q = ekm_hip.to_device(t), ekm_hip.to_device(q)

    pressure = ekm_hip.HybridPressure(A, B, d_sp)          # the definition of p: never read from HBM as a field
    es, td, rh = thermo.pipeline_svp_td_rh(d_t, d_q, pressure)
    theta_e = thermo.ept_from_specific_humidity(d_t, d_q, pressure)
    tw = thermo.wet_bulb_temperature_from_specific_humidity(d_t, d_q, pressure, t_method="newton")
    h = vertical.height_on_hybrid_levels(d_t, d_q, d_zs, A.astype(np.float32), B.astype(np.float32), d_sp,
                                         h_type="geopotential", h_reference="sea")
    ekm_hip.synchronize()

    k = 120  # a level near 900 hPa
    for name, arr in (("p [Pa]", p), ("rh [%]", rh), ("td [K]", td), ("theta_e [K]", theta_e), ("tw [K]", tw),
                      ("height [m]", h)):
        a = arr.to_host()[k]
        print(f"level {k + 1:3d}  {name:12s} min {a.min():10.3f}  mean {a.mean():10.3f}  max {a.max():10.3f}")
    return rh.to_host(), h.to_host()


if __name__ == "__main__":
    main()
