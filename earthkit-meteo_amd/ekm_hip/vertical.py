"""`earthkit.meteo.vertical.pressure_on_hybrid_levels` on MI355X -- the producer of the
model-level pressure field the thermo kernels consume.

Same signature, outputs, level selection/ordering and error messages as the reference
(/root/reference/src/earthkit/meteo/vertical/array/vertical.py:505-740).  NumPy `sp` in ->
NumPy out; `DeviceArray` `sp` in -> DeviceArrays out (then `vertical_axis` must be 0).
`HybridPressure(A, B, sp)` hands the same definition to the thermo kernels instead of a
materialised pressure field (ekm_hip.thermo functions accept it in place of `p`).
"""
import ctypes as C
import math

import numpy as np

from . import _ffi
from .device import DeviceArray, _Allocation, current_device, current_stream

_F32, _F64 = np.dtype(np.float32), np.dtype(np.float64)
PRESSURE_TOA = 0.1  # vertical.py:674


def _foreign_aware(*names):
    """Arguments `names` of the wrapped function may be arrays of another ROCm library (torch / cupy device tensors:
    `__dlpack_device__()` says kDLROCM): they are taken over through DLPack exactly as in ekm_hip.thermo
    (`_engine._adopt_foreign`), and when every such argument came from one library the DeviceArray results go back
    through that library's `from_dlpack`."""
    import functools
    import inspect

    def deco(fn):
        sig = inspect.signature(fn)

        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            from . import _engine

            bound = sig.bind(*args, **kwargs)
            present = [n for n in names if n in bound.arguments and bound.arguments[n] is not None]
            adopted, module = _engine._adopt_foreign([bound.arguments[n] for n in present])
            for n, a in zip(present, adopted):
                bound.arguments[n] = a
            res = fn(*bound.args, **bound.kwargs)
            if module is None:
                return res
            out = _engine._hand_back(res if isinstance(res, tuple) else (res,), module)
            return out if isinstance(res, tuple) else out[0]

        wrapper.__signature__ = sig
        return wrapper

    return deco


def _dev_bytes(host_array, device):
    a = np.ascontiguousarray(host_array)
    alloc = _Allocation(max(a.nbytes, 16), device)
    lib = _ffi.lib()
    _ffi.check(lib.ekm_h2d(device, alloc.ptr, a.ctypes.data, a.nbytes, current_stream()))
    _ffi.check(lib.ekm_stream_sync(device, current_stream()))
    return alloc


def _compute_dtype(A, B, sp):
    parts = [x.dtype for x in (A, B, sp)]
    rd = np.result_type(*parts, 0.0)
    return _F32 if rd == _F32 else _F64


def _select_levels(A, B, levels):
    """vertical.py:641-661: contiguous half-level range covering the request + row maps."""
    nlev = A.shape[0] - 1
    levels = np.asarray(levels)
    lmax, lmin = int(levels.max()), int(levels.min())
    if lmax > nlev:
        raise ValueError(f"Requested level {lmax} exceeds the maximum number of levels {nlev}.")
    if lmin < 1:
        raise ValueError(f"Level numbering starts at 1. Found level={lmin} < 1.")
    half_idx = np.arange(lmin - 1, lmax + 1)
    out_half_idx = np.nonzero(levels[:, None] == half_idx[None, :])[1]  # local half index per requested level
    return A[half_idx], B[half_idx], out_half_idx


@_foreign_aware("sp")
def pressure_on_hybrid_levels(A, B, sp, levels=None, alpha_top="ifs", output="full", vertical_axis=0):
    """Pressure on hybrid full/half levels and the delta/alpha layer parameters (vertical.py:505-740)."""
    if isinstance(output, str):
        output = (output,)
    if not output:
        raise ValueError("At least one output type must be specified.")
    for out in output:
        if out not in ["full", "half", "alpha", "delta"]:
            raise ValueError(f"Unknown output type '{out}'. Allowed values are 'full', 'half', 'alpha' or 'delta'.")
    if alpha_top not in ["ifs", "arpege"]:
        raise ValueError(f"Unknown method '{alpha_top}' for pressure calculation. Use 'ifs' or 'arpege'.")

    A = np.asarray(A)
    B = np.asarray(B)
    on_device = isinstance(sp, DeviceArray)
    if not on_device:
        sp = np.asarray(sp)
    if on_device and vertical_axis != 0 and sp.ndim > 0:
        raise ValueError("DeviceArray input: outputs are level-major, vertical_axis must be 0")
    dtype = _compute_dtype(A, B, sp)
    device = sp.device if on_device else current_device()

    sel = None
    if levels is not None:
        A, B, sel = _select_levels(A, B, levels)
    nfull = A.shape[0] - 1
    sp_shape = tuple(sp.shape)
    npts = int(math.prod(sp_shape))

    # output rows: identity, or the requested levels in the requested order
    if sel is None:
        nrow_full, nrow_half, row_full, row_half = nfull, nfull + 1, None, None
        dup = None
    else:
        uniq, first = np.unique(sel, return_index=True)
        dup = None if len(uniq) == len(sel) else sel  # repeated levels: compute unique rows, then gather
        order = uniq if dup is not None else sel
        row_half = np.full(nfull + 1, -1, np.int32)
        row_half[order] = np.arange(len(order), dtype=np.int32)
        row_full = np.full(nfull, -1, np.int32)
        row_full[order - 1] = np.arange(len(order), dtype=np.int32)
        nrow_full = nrow_half = len(order)

    lib = _ffi.lib()
    stream = current_stream()
    tag, real = ("f32", C.c_float) if dtype == _F32 else ("f64", C.c_double)
    d_sp = sp if (on_device and sp.dtype == dtype) else DeviceArray.from_host(np.asarray(sp, dtype=dtype), device)
    d_a = DeviceArray.from_host(A.astype(dtype), device)
    d_b = DeviceArray.from_host(B.astype(dtype), device)
    d_rf = _dev_bytes(row_full, device) if row_full is not None else None
    d_rh = _dev_bytes(row_half, device) if row_half is not None else None

    d_sp.on(stream)  # an `sp` last used on another stream: order this stream after that work
    top_is_zero = _top_is_zero(A[0], B[0], sp, d_sp, dtype, npts, device, on_device)
    a_top = float(np.log(2)) if alpha_top == "ifs" else 1.0

    bufs = {}
    for name in set(output):
        rows = nrow_half if name == "half" else nrow_full
        bufs[name] = DeviceArray.empty((rows,) + sp_shape, dtype, device)
    ptr = lambda n: bufs[n].ptr if n in bufs else None  # noqa: E731
    _ffi.check(getattr(lib, f"ekm_pressure_on_hybrid_levels_{tag}")(
        device, stream, d_a.ptr, d_b.ptr, d_sp.ptr, npts, nfull, d_rf.ptr if d_rf else None,
        d_rh.ptr if d_rh else None, int(top_is_zero), real(a_top), ptr("full"), ptr("half"), ptr("delta"),
        ptr("alpha")))

    res = []
    if on_device:
        _ffi.check(lib.ekm_stream_sync(device, stream))
        for name in output:
            arr = bufs[name]
            if dup is not None:  # repeated levels: gather rows device-to-device
                uniq = np.unique(dup)
                out = DeviceArray.empty((len(dup),) + sp_shape, dtype, device)
                row_bytes = npts * dtype.itemsize
                for r, h in enumerate(dup):
                    src = int(np.searchsorted(uniq, h))
                    _ffi.check(lib.ekm_d2d(device, out.ptr + r * row_bytes, arr.ptr + src * row_bytes, row_bytes, stream))
                _ffi.check(lib.ekm_stream_sync(device, stream))
                arr = out
            res.append(arr)
    else:
        for name in output:
            h = bufs[name].to_host()
            if dup is not None:
                h = h[np.searchsorted(np.unique(dup), dup)]
            if name in ("delta", "alpha") and h.dtype != _F64:
                h = h.astype(np.float64)  # the reference allocates these with the default dtype (vertical.py:678, 687)
            res.append(h)
        for b in bufs.values():
            b.free()
        if vertical_axis != 0 and res[0].ndim > 1:
            res = [np.moveaxis(r, 0, vertical_axis) for r in res]
    return res[0] if len(res) == 1 else tuple(res)


def _top_is_zero(A0, B0, sp, d_sp, dtype, npts, device, on_device):
    """The reference's global any(p_half[0] <= 0.1) (vertical.py:680, 694)."""
    lib, stream = _ffi.lib(), current_stream()
    a0, b0 = float(A0), float(B0)
    if b0 == 0.0 or npts == 0:
        return bool(a0 <= PRESSURE_TOA)
    if not on_device:
        return bool(np.any(np.asarray(A0, dtype) + np.asarray(B0, dtype) * np.asarray(sp, dtype=dtype) <= PRESSURE_TOA))
    tag, real = ("f32", C.c_float) if dtype == _F32 else ("f64", C.c_double)
    flag = _dev_bytes(np.zeros(1, np.int32), device)
    _ffi.check(getattr(lib, f"ekm_any_le_{tag}")(device, stream, d_sp.ptr, npts, real(a0), real(b0),
                                                 real(PRESSURE_TOA), flag.ptr))
    host = np.zeros(1, np.int32)
    _ffi.check(lib.ekm_d2h(device, host.ctypes.data, flag.ptr, 4, stream))
    _ffi.check(lib.ekm_stream_sync(device, stream))
    return bool(host[0])


_GEO_MODE = {"thickness": 0, "geopotential": 1, ("geometric", "sea"): 2, ("geopotential", "sea"): 3,
             ("geometric", "ground"): 4, ("geopotential", "ground"): 5}


def _chain(t, q, zs, A, B, sp, alpha_top, mode, vertical_axis):
    """t, q on hybrid full levels + surface pressure -> geopotential thickness / geopotential / height in
    ONE bottom-up pass per column (vertical.py:741-1190); alpha, delta and the half-level pressures are
    formed on the fly and never stored."""
    if alpha_top not in ["ifs", "arpege"]:
        raise ValueError(f"Unknown method '{alpha_top}' for pressure calculation. Use 'ifs' or 'arpege'.")
    A, B = np.asarray(A), np.asarray(B)
    arrs = dict(t=t, q=q, sp=sp, zs=zs)
    on_device = any(isinstance(v, DeviceArray) for v in arrs.values())
    if on_device and vertical_axis != 0:
        raise ValueError("DeviceArray input: fields are level-major, vertical_axis must be 0")
    host = {k: (v if isinstance(v, DeviceArray) else (None if v is None else np.asarray(v))) for k, v in arrs.items()}
    if not on_device and vertical_axis != 0:
        host["t"], host["q"] = (np.moveaxis(host[k], vertical_axis, 0) for k in ("t", "q"))
    parts = [v.dtype for v in host.values() if v is not None] + [A.dtype, B.dtype]
    dtype = _F32 if np.result_type(*parts, 0.0) == _F32 else _F64
    out_parts = [host["t"].dtype, host["q"].dtype] + ([host["zs"].dtype] if zs is not None and mode not in (0, 5) else [])
    out_dtype = np.result_type(*out_parts, np.float32 if all(p == _F32 for p in out_parts) else 0.0)
    if not on_device and host["t"].ndim >= 1:
        # NumPy input broadcasts as it does in the reference (t of [levels, 1] against q and sp of [.., n]): the level count is
        # t's (vertical.py:1191-1203), everything is brought to [levels, *columns] on the host
        try:
            full = np.broadcast_shapes(host["t"].shape, host["q"].shape, (host["t"].shape[0],) + tuple(host["sp"].shape))
            if full[0] == host["t"].shape[0] and (host["zs"] is None or np.broadcast_shapes(full[1:], host["zs"].shape) == full[1:]):
                host["t"], host["q"] = np.broadcast_to(host["t"], full), np.broadcast_to(host["q"], full)
                host["sp"] = np.broadcast_to(host["sp"], full[1:])
                if host["zs"] is not None:
                    host["zs"] = np.broadcast_to(host["zs"], full[1:])
        except ValueError:
            pass  # not broadcastable: the shape error below
    shape = tuple(host["t"].shape)
    if tuple(host["q"].shape) != shape or shape[1:] != tuple(host["sp"].shape):
        raise ValueError(f"t {shape}, q {tuple(host['q'].shape)} must be [levels, *sp.shape] with sp {tuple(host['sp'].shape)}")
    nlev_t, nlev = shape[0], A.shape[0] - 1
    if nlev_t > nlev:
        raise ValueError(f"data have {nlev_t} levels, A/B have {nlev} levels")
    A, B = A[nlev - nlev_t:], B[nlev - nlev_t:]  # the bottom-most nlev_t layers (vertical.py:1191-1203)
    npts = int(math.prod(shape[1:]))
    device = next((v.device for v in host.values() if isinstance(v, DeviceArray)), current_device())

    def dev(v):
        if isinstance(v, DeviceArray) and v.dtype == dtype:
            return v
        return DeviceArray.from_host(np.ascontiguousarray(np.asarray(v), dtype=dtype), device)

    d = {k: (dev(v) if v is not None else None) for k, v in host.items()}
    for v in d.values():
        if v is not None:
            v.on(current_stream())  # inputs last used on another stream: order this stream after that work
    d_a, d_b = dev(A), dev(B)
    top = _top_is_zero(A[0], B[0], host["sp"], d["sp"], dtype, npts, device, isinstance(host["sp"], DeviceArray))
    a_top = float(np.log(2)) if alpha_top == "ifs" else 1.0
    out = DeviceArray.empty(shape, dtype, device)
    lib = _ffi.lib()
    tag, real = ("f32", C.c_float) if dtype == _F32 else ("f64", C.c_double)
    _ffi.check(getattr(lib, f"ekm_geopotential_on_hybrid_levels_{tag}")(
        device, current_stream(), d_a.ptr, d_b.ptr, d["sp"].ptr, d["zs"].ptr if d["zs"] is not None else None,
        d["t"].ptr, d["q"].ptr, npts, nlev_t, int(top), real(a_top), mode, out.ptr))
    if on_device:
        _ffi.check(lib.ekm_stream_sync(device, current_stream()))
        return out
    res = out.to_host().astype(out_dtype, copy=False)
    out.free()
    if vertical_axis != 0:
        res = np.moveaxis(res, 0, vertical_axis)
    return res


@_foreign_aware("t", "q", "alpha", "delta")
def relative_geopotential_thickness_on_hybrid_levels_from_alpha_delta(t, q, alpha, delta, vertical_axis=0):
    """Geopotential thickness between the surface and the hybrid full levels from alpha and delta the caller
    already holds (outputs of `pressure_on_hybrid_levels`): vertical.py:741-893.  The same bottom-up column scan
    as `relative_geopotential_thickness_on_hybrid_levels`, with alpha and delta STREAMED (16 B read + 4 B written
    per point in fp32) instead of formed from A, B, sp.  NumPy result dtype = that of t and q, as in the reference."""
    arrs = dict(t=t, q=q, alpha=alpha, delta=delta)
    on_device = any(isinstance(v, DeviceArray) for v in arrs.values())
    if on_device and vertical_axis != 0:
        raise ValueError("DeviceArray input: fields are level-major, vertical_axis must be 0")
    host = {k: (v if isinstance(v, DeviceArray) else np.asarray(v)) for k, v in arrs.items()}
    if vertical_axis != 0:
        host = {k: np.moveaxis(v, vertical_axis, 0) for k, v in host.items()}
    # the reference writes into zeros_like(R(q) * t) (vertical.py:762): the RESULT has the dtype of t and q, while the
    # arithmetic runs in the promotion of all four (its own alpha / delta are fp64 whatever the input dtype)
    out_dtype = np.result_type(host["t"].dtype, host["q"].dtype)
    if out_dtype.kind != "f":
        out_dtype = _F64
    cd = np.result_type(*[v.dtype for v in host.values()])
    dtype = _F32 if cd in (_F32, np.dtype(np.float16)) else _F64
    shape = tuple(host["t"].shape)
    if len(shape) == 0 or any(tuple(v.shape) != shape for v in host.values()):
        raise ValueError("t, q, alpha and delta must have the same shape [levels, ...]: "
                         + ", ".join(f"{k} {tuple(v.shape)}" for k, v in host.items()))
    nlev, npts = shape[0], int(math.prod(shape[1:]))
    device = next((v.device for v in host.values() if isinstance(v, DeviceArray)), current_device())
    stream = current_stream()
    d = {}
    for k, v in host.items():
        d[k] = v if isinstance(v, DeviceArray) and v.dtype == dtype else DeviceArray.from_host(
            np.ascontiguousarray(np.asarray(v), dtype=dtype), device)
        d[k].on(stream)
    out = DeviceArray.empty(shape, dtype, device)
    lib = _ffi.lib()
    tag = "f32" if dtype == _F32 else "f64"
    _ffi.check(getattr(lib, f"ekm_geopotential_thickness_from_alpha_delta_{tag}")(
        device, stream, d["t"].ptr, d["q"].ptr, d["alpha"].ptr, d["delta"].ptr, npts, nlev, out.on(stream)))
    if on_device:
        return out  # device-resident: in the arithmetic dtype
    res = out.to_host().astype(out_dtype, copy=False)
    out.free()
    if vertical_axis != 0:
        res = np.moveaxis(res, 0, vertical_axis)
    return res


@_foreign_aware("t", "q", "sp")
def relative_geopotential_thickness_on_hybrid_levels(t, q, A, B, sp, alpha_top="ifs", vertical_axis=0):
    """Geopotential thickness between the surface and the hybrid full levels (vertical.py:894-994)."""
    return _chain(t, q, None, A, B, sp, alpha_top, _GEO_MODE["thickness"], vertical_axis)


@_foreign_aware("t", "q", "zs", "sp")
def geopotential_on_hybrid_levels(t, q, zs, A, B, sp, alpha_top="ifs", vertical_axis=0):
    """Geopotential on hybrid full levels (vertical.py:997-1069)."""
    return _chain(t, q, zs, A, B, sp, alpha_top, _GEO_MODE["geopotential"], vertical_axis)


@_foreign_aware("t", "q", "zs", "sp")
def height_on_hybrid_levels(t, q, zs, A, B, sp, alpha_top="ifs", h_type="geometric", h_reference="ground",
                            vertical_axis=0):
    """Geometric / geopotential height above sea level / ground on hybrid full levels (vertical.py:1072-1188)."""
    if h_reference not in ["sea", "ground"]:
        raise ValueError(f"Unknown '{h_reference=}'. Use 'sea' or 'ground'.")
    key = ("geometric" if h_type == "geometric" else "geopotential", h_reference)
    return _chain(t, q, zs, A, B, sp, alpha_top, _GEO_MODE[key], vertical_axis)


_LEVEL_TABLES = None


def hybrid_level_parameters(n_levels, model="ifs"):
    """The A and B half-level coefficients of a hybrid-level configuration (vertical/array/hybrid.py:40-102):
    two float64 arrays of length n_levels + 1; `model="ifs"` with 91 or 137 levels, same errors as the
    reference.  The tables are the constant data of the reference's conf/ifs_levels_conf.json, recorded into
    ekm_hip/data/ifs_levels.npz by tests/golden/gen_golden_vertical.py and shipped inside the package."""
    global _LEVEL_TABLES
    import os

    if _LEVEL_TABLES is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "ifs_levels.npz")
        with np.load(path) as f:
            tabs = {}
            for key in f.files:
                mdl, n, which = key.split(".")
                tabs.setdefault(mdl, {}).setdefault(n, {})[which] = f[key]
        _LEVEL_TABLES = tabs
    model = model.lower()
    n_levels = str(n_levels)
    if model in _LEVEL_TABLES:
        if n_levels in _LEVEL_TABLES[model]:
            c = _LEVEL_TABLES[model][n_levels]
            return c["A"], c["B"]
        raise ValueError(f"Hybrid level parameters not available for {n_levels} levels in model '{model}'.")
    raise ValueError(f"Model '{model}' not recognized for hybrid level parameters.")


class HybridPressure:
    """Pressure on hybrid full levels given by its definition, p_k(x) = p_half[k] + 0.5*(p_half[k+1]-p_half[k])
    with p_half[h] = A[h] + B[h]*sp(x), for use in place of a pressure field in ekm_hip.thermo calls on
    `[level, ...sp.shape]` fields: the kernels form p from `sp` and the LDS-resident A/B tables, so the
    pressure field is never read from (or written to) HBM."""

    def __init__(self, A, B, sp):
        self.A = np.array(A, dtype=np.float64, order="C")  # copies: the device tables made from them are kept (device_tables)
        self.B = np.array(B, dtype=np.float64, order="C")
        if self.A.ndim != 1 or self.A.shape != self.B.shape or self.A.size < 2:
            raise ValueError("A and B must be 1-D half-level tables of the same length >= 2")
        self.sp = sp
        self.nlev = self.A.size - 1
        # leading pure pressure levels (B = 0 on both half levels: the upper 53 of the 137 IFS levels): the library runs
        # them at level-vector speed in a launch of their own (ekm_operand.nflat)
        nz = np.flatnonzero(self.B != 0.0)
        self.nflat = int(max(0, (nz[0] if nz.size else self.B.size) - 1))
        self._tables = {}  # (device, dtype) -> the two tables in device memory, uploaded once

    def device_tables(self, device, dtype):
        """A and B in `dtype` on `device`, uploaded at the first use there (synchronously: any stream may read them
        afterwards) and kept with this object.  Inside an `ekm_hip.graph()` block nothing can be uploaded: use the object
        once, or call this, before the block."""
        key = (int(device), np.dtype(dtype).char)
        tabs = self._tables.get(key)
        if tabs is None:
            tabs = self._tables[key] = tuple(DeviceArray.from_host(x.astype(dtype), device=device) for x in (self.A, self.B))
        return tabs

    @property
    def shape(self):
        return (self.nlev,) + tuple(np.shape(self.sp))


# `earthkit.meteo.vertical.array.<name>` is how the reference reaches the array-level functions
# (vertical/__init__.py:20, vertical/array/__init__.py:14-15): same module here.
import sys as _sys  # noqa: E402

array = _sys.modules[__name__]
