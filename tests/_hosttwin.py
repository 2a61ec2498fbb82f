"""ctypes access to the host twin of the kernel math (tests/_build/libekm_hosttwin.so).

TEST INFRASTRUCTURE: exposes the same per-point functors the gfx950 kernels run,
compiled by g++, under the reference's function names, so the formulas can be
checked against the golden vectors without a GPU.  Never used by ekm_hip.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.environ.get("EKM_HOSTTWIN_LIB") or os.path.join(ROOT, "tests", "_build", "libekm_hosttwin.so")
ASAN_PATH = os.path.join(ROOT, "tests", "_build", "libekm_hosttwin_asan.so")

import sys  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "earthkit-meteo_amd"))
from ekm_hip._ffi import EPT_METHOD, LCL_METHOD, PHASE, T_METHOD  # noqa: E402
from ekm_hip._optable import OPS  # noqa: E402

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(PATH)
    return _lib


def call(name, args, ints=(), eps=None, dtype=np.float64):
    ins, outs, int_names, has_eps = OPS[name]
    dtype = np.dtype(dtype)
    tag, real = ("f32", C.c_float) if dtype == np.float32 else ("f64", C.c_double)
    arrs = np.broadcast_arrays(*[np.asarray(a, dtype=dtype) for a in args])
    shape = arrs[0].shape
    arrs = [np.ascontiguousarray(a).ravel() for a in arrs]
    n = arrs[0].size
    res = [np.empty(n, dtype=dtype) for _ in outs]
    fn = getattr(lib(), f"ekm_host_{name}_{tag}")
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p] * len(ins) + [C.c_int] * len(ints) + ([real] if has_eps else []) + \
        [C.c_void_p] * len(outs) + [C.c_size_t]
    cargs = [a.ctypes.data for a in arrs] + list(ints) + ([eps] if has_eps else []) + [r.ctypes.data for r in res] + [n]
    rc = fn(*cargs)
    assert rc == 0, rc
    return tuple(r.reshape(shape) for r in res)


def by_reference_name(func, args, kwargs, dtype):
    """Map a reference-style call (function name + kwargs) onto the twin's entry points."""
    kw = dict(kwargs)
    ints, eps = [], None
    name = func
    if func in ("saturation_mixing_ratio_slope", "saturation_specific_humidity_slope",
                "specific_humidity_from_vapour_pressure", "mixing_ratio_from_vapour_pressure"):
        eps = kw.pop("eps", 1e-4)
    if "phase" in kw:
        ints.append(PHASE[kw.pop("phase")])
    elif func.startswith("saturation_") and func != "saturation_ept":
        ints.append(PHASE["mixed"])
    if func in ("lcl_temperature", "lcl"):
        ints.append(LCL_METHOD[kw.pop("method", "davies")])
    if func in ("ept_from_dewpoint", "ept_from_specific_humidity", "saturation_ept"):
        ints.append(EPT_METHOD[kw.pop("method", "ifs")])
    if func.startswith(("temperature_on_moist", "wet_bulb_")):
        ints.append(EPT_METHOD[kw.pop("ept_method", "ifs")])
        ints.append(T_METHOD[kw.pop("t_method", "direct" if "potential" in func else "bisect")])
    assert not kw, kw
    out = call(name, args, ints, eps, dtype)
    return out if len(out) > 1 else out[0]
