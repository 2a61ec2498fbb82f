"""Minimal stand-in for `earthkit.utils.array` (earthkit-utils >= 0.2, an
un-vendored dependency of the reference; pyproject.toml:29-34 there).

Used ONLY by tests/golden/gen_golden.py in the build container to import the
reference and record golden vectors.  For NumPy inputs the real package hands
back NumPy plus a few helpers; the ones the thermo path touches are below.
The reference's own 92 thermo tests pass on top of this stand-in.
"""
import numpy as _np


class _NumpyNamespace:
    def __getattr__(self, name):
        return getattr(_np, name)

    @staticmethod
    def polyval(x, c):
        return _np.polynomial.polynomial.polyval(x, c)

    @staticmethod
    def device(x):
        return getattr(x, "device", "cpu")

    @staticmethod
    def asarray(x, *args, device=None, **kw):
        return _np.asarray(x, *args, **kw)

    @staticmethod
    def full(shape, value, dtype=None, device=None):
        return _np.full(shape, value, dtype=dtype)


_NS = _NumpyNamespace()


def array_namespace(*arrays):
    return _NS
