"""Argument normalisation and launch: the host logic between the reference's
Python signatures (ekm_hip.thermo) and the C ABI (libekm_thermo.so).

What it reproduces from the reference:
 * inputs may be Python scalars, lists, 0-d / N-d arrays, mutually broadcastable;
 * output dtype is NumPy's promotion of the inputs with Python scalars weak
   (fp32 arrays stay fp32), integers and lists promote to float64;
 * inputs are never mutated, outputs are fresh arrays;
 * all-scalar calls return a NumPy scalar.
What it adds: `DeviceArray` inputs keep the whole call on the GPU and return
DeviceArrays; broadcast operands (a scalar, a level vector along the leading or
trailing axes) are passed to the kernels as such instead of being materialised.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._optable import OPS
from .device import DeviceArray, current_device, current_stream
from .vertical import HybridPressure

_F32, _F64 = np.dtype(np.float32), np.dtype(np.float64)
_MAX_LDS_BYTES = 64 * 1024
_MIN_VEC = 4  # a LEVEL operand must span at least one 16-B chunk per level
_PRETOUCH_BYTES = 8 << 20
_STREAM_BYTES = 256 << 20  # total input bytes from which a single-GPU NumPy call is streamed in slices


def _pretouch(arrays):
    """Fault in the pages of fresh result arrays from a helper thread while the inputs upload
    (ekm_host_prefault runs outside the GIL, on a few threads of its own)."""
    import threading

    lib = _ffi.lib()

    def work():
        for a in arrays:
            lib.ekm_host_prefault(a.ctypes.data, a.nbytes, 4)

    th = threading.Thread(target=work, daemon=True)
    th.start()
    return th


def _result_dtype(args):
    """NumPy promotion with weak Python scalars; anything non-float computes in float64."""
    parts = []
    for a in args:
        if isinstance(a, HybridPressure):
            a = a.sp
        if isinstance(a, DeviceArray):
            parts.append(a.dtype)
        elif isinstance(a, (bool, int, float)):
            parts.append(a)
        else:
            parts.append(np.asarray(a).dtype if not isinstance(a, (np.ndarray, np.generic)) else a.dtype)
    rd = np.result_type(*parts, 0.0) if parts else _F64
    if rd == np.float16:
        return np.dtype(np.float16), _F32
    if rd == _F32:
        return _F32, _F32
    if rd.kind in "fiub":
        return _F64, _F64
    raise TypeError(f"unsupported dtype for thermo computation: {rd}")


def _padded(shape, ndim):
    return (1,) * (ndim - len(shape)) + tuple(shape)


def classify(shape, out_shape):
    """How an operand of `shape` broadcasts to `out_shape`.

    Returns (mode, len, inner) with mode one of FIELD / SCALAR / LEVEL_MAJOR /
    LEVEL_MINOR, or None when the pattern needs materialising.
    """
    size = int(np.prod(shape, dtype=np.int64))
    n = int(np.prod(out_shape, dtype=np.int64))
    if size == 1 and n != 1:
        return _ffi.SCALAR, 0, 0
    s = _padded(shape, len(out_shape))
    if s == tuple(out_shape):
        return _ffi.FIELD, 0, 0
    nz = [i for i, d in enumerate(s) if d != 1]
    lo, hi = nz[0], nz[-1]
    if any(s[i] != out_shape[i] for i in range(lo, hi + 1)):
        return None  # a 1 in the middle of the block: not a plain vector
    if lo == 0 or all(out_shape[i] == 1 for i in range(lo)):
        # varies along the leading axes only: value index = flat_index // inner
        inner = int(np.prod(out_shape[hi + 1:], dtype=np.int64))
        if inner >= _MIN_VEC:
            return _ffi.LEVEL_MAJOR, size, inner
    if hi == len(out_shape) - 1:
        # varies along the trailing axes only: value index = flat_index % len
        if size >= _MIN_VEC:
            return _ffi.LEVEL_MINOR, size, 0
    return None


class _Plan:
    """Normalised operands of one call."""

    def __init__(self, args, dtype_override=None):
        self.hybrid = [a for a in args if isinstance(a, HybridPressure)]
        if self.hybrid and args[-1] is not self.hybrid[0] or len(self.hybrid) > 1:
            raise ValueError("HybridPressure can only stand in for the last (pressure) argument")
        args = [a.sp if isinstance(a, HybridPressure) else a for a in args]
        self.on_device = any(isinstance(a, DeviceArray) for a in args)
        self.out_dtype, self.dtype = _result_dtype(args)
        if dtype_override is not None:
            self.out_dtype = self.dtype = np.dtype(dtype_override)
        devs = {a.device for a in args if isinstance(a, DeviceArray)}
        if len(devs) > 1:
            raise ValueError(f"inputs live on different devices: {sorted(devs)}")
        self.device = devs.pop() if devs else current_device()
        self.all_scalar = all(np.ndim(a) == 0 for a in args if not isinstance(a, DeviceArray)) and not self.on_device
        self.host = []
        shapes = []
        for a in args:
            if isinstance(a, DeviceArray):
                self.host.append(a)
                shapes.append(a.shape)
            else:
                h = np.asarray(a)
                self.host.append(h)
                shapes.append(h.shape)
        if self.hybrid:  # p has the shape [level, ...sp.shape]; the fields must have exactly that shape
            shapes[-1] = self.hybrid[0].shape
        self.shape = tuple(np.broadcast_shapes(*shapes))
        if self.hybrid and (self.shape != self.hybrid[0].shape or any(tuple(s) != self.shape for s in shapes[:-1])):
            raise ValueError(f"HybridPressure of shape {self.hybrid[0].shape} needs fields of exactly that shape")
        self.n = int(np.prod(self.shape, dtype=np.int64))


def run(name, args, ints=(), eps=None, dtype=None):
    """Launch entry point `name` on `args`; returns a tuple of outputs.  Inside `ekm_hip.multi_gpu()`
    NumPy inputs are split along their leading axis across the selected GPUs (one host thread each)."""
    from .device import current_devices

    devs = current_devices()
    numpy_only = not any(isinstance(a, (DeviceArray, HybridPressure)) for a in args)
    if numpy_only and not (devs and len(devs) > 1):
        # one GPU, large host arrays: eight slices on eight streams, so that the upload of one slice
        # overlaps the download of another (PCIe is full duplex: 19 ms instead of 23 ms for 3 x 207 MB
        # in, 3 x 207 MB out)
        nbytes = sum(np.asarray(a).nbytes for a in args if np.ndim(a) > 0)
        if nbytes >= _STREAM_BYTES:
            devs = [current_device()] * 8
    if devs and len(devs) > 1 and numpy_only:
        sharded = _run_sharded(name, args, ints, eps, devs)
        if sharded is not None:
            return sharded
    return _run_single(name, args, ints, eps, dtype)


def leading_axis_bounds(n0, nshards):
    """Contiguous [lo, hi) ranges of a leading axis of length n0 for nshards GPUs (sizes differ by <= 1)."""
    base, extra = divmod(n0, nshards)
    out, lo = [], 0
    for r in range(nshards):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


_streams = {}


def _shard_stream(dev, key):
    """One long-lived stream per (device, shard slot)."""
    from .device import stream_create

    k = (dev, key)
    if k not in _streams:
        _streams[k] = stream_create(dev)
    return _streams[k]


def _run_sharded(name, args, ints, eps, devs):
    """Grid points are independent: cut the broadcast result along its leading axis into one contiguous
    block per GPU (for [level, lat, lon] fields: ~17 whole levels each on 8 GPUs), give every operand that
    spans that axis the matching slice and every other operand (scalars, trailing-axis vectors) whole,
    and run the blocks concurrently, one host thread per device, straight into slices of the result."""
    import threading

    from .device import set_device, set_stream

    host = [np.asarray(a) for a in args]
    shape = tuple(np.broadcast_shapes(*[h.shape for h in host]))
    if len(shape) == 0 or shape[0] < len(devs):
        return None
    out_dtype, _ = _result_dtype(args)
    nout = len(OPS[name][1])
    outs = [np.empty(shape, out_dtype) for _ in range(nout)]
    toucher = _pretouch(outs) if outs[0].nbytes >= _PRETOUCH_BYTES else None
    bounds = [b for b in leading_axis_bounds(shape[0], len(devs)) if b[1] > b[0]]
    errors = []

    def work(dev, lo, hi):
        try:
            set_device(dev)
            set_stream(_shard_stream(dev, lo))  # own stream: shards on one GPU overlap upload / kernel / download
            # operands that span the leading axis get the matching slice; everything else is passed as
            # the caller gave it (a Python scalar must stay a weak scalar for the dtype promotion)
            part = [h[lo:hi] if (h.ndim == len(shape) and h.shape[0] == shape[0]) else a for h, a in zip(host, args)]
            _run_single(name, part, ints, eps, None, host_out=[o[lo:hi] for o in outs])
        except BaseException as exc:  # surfaced in the calling thread
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(d, lo, hi)) for d, (lo, hi) in zip(devs, bounds)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    if toucher is not None:
        toucher.join()
    if errors:
        raise errors[0]
    return tuple(outs)


def _run_single(name, args, ints=(), eps=None, dtype=None, host_out=None):
    ins, outs, int_names, has_eps = OPS[name]
    assert len(args) == len(ins) and len(ints) == len(int_names)
    plan = _Plan(args, dtype)
    lib = _ffi.lib()
    dev, stream = plan.device, current_stream()
    tag = "f32" if plan.dtype == _F32 else "f64"
    fn = getattr(lib, f"ekm_{name}_{tag}")

    temps = []  # device buffers owned by this call
    operands = []
    lds_bytes = 0
    toucher = None
    if not plan.on_device and host_out is None and plan.n * plan.dtype.itemsize >= _PRETOUCH_BYTES:
        # Large NumPy result: the pages of a fresh array fault on first touch, which slows the download
        # from 56 to ~24 GB/s.  Fault them in from a helper thread while the inputs are uploading.
        host_out = [np.empty(plan.shape, plan.dtype) for _ in outs]
        toucher = _pretouch(host_out)
    for k, a in enumerate(plan.host):
        if plan.hybrid and k == len(plan.host) - 1:
            hp = plan.hybrid[0]
            sp = a if isinstance(a, DeviceArray) and a.dtype == plan.dtype else DeviceArray.from_host(
                np.ascontiguousarray(np.asarray(a), dtype=plan.dtype), device=dev)
            tabs = [DeviceArray.from_host(x.astype(plan.dtype), device=dev) for x in (hp.A, hp.B)]
            temps.extend(tabs + ([sp] if sp is not a else []))
            inner = max(1, sp.size)
            operands.append(_ffi.Operand(sp.ptr, _ffi.HYBRID_FULL, 0, hp.nlev, inner, tabs[0].ptr, tabs[1].ptr))
            continue
        shape = a.shape
        cls = classify(shape, plan.shape) if plan.n else (_ffi.FIELD, 0, 0)
        if cls is not None and cls[0] >= _ffi.LEVEL_MAJOR:
            nb = cls[1] * plan.dtype.itemsize
            if lds_bytes + nb > _MAX_LDS_BYTES:
                cls = None
            else:
                lds_bytes += (nb + 15) & ~15
        if isinstance(a, DeviceArray) and cls is not None and a.dtype == plan.dtype:
            darr = a
        else:
            h = np.asarray(a)  # DeviceArray -> host copy only on the slow path
            if cls is None:
                h = np.broadcast_to(h.reshape(_padded(h.shape, len(plan.shape))), plan.shape)
                cls = (_ffi.FIELD, 0, 0)
            darr = DeviceArray.from_host(np.ascontiguousarray(h, dtype=plan.dtype), device=dev)
            temps.append(darr)
        operands.append(_ffi.Operand(darr.ptr, cls[0], 0, cls[1], cls[2]))

    results = [DeviceArray.empty(plan.shape, plan.dtype, dev) for _ in outs]
    cargs = [dev, stream] + [C.byref(o) for o in operands] + [int(v) for v in ints]
    if has_eps:
        cargs.append(float(eps))
    cargs += [r.ptr for r in results] + [plan.n]
    _ffi.check(fn(*cargs))

    if plan.on_device:
        # temporaries are freed by HIP in stream order after the kernel has run
        for t in temps:
            t.free()
        return tuple(results)

    if toucher is not None:
        toucher.join()
    host = []
    for k, r in enumerate(results):
        if host_out is not None and host_out[k].dtype == plan.dtype and host_out[k].flags.c_contiguous:
            h = r.to_host(out=host_out[k])  # straight into the caller's slice
        else:
            h = r.to_host()  # synchronises the stream
            if host_out is not None:
                host_out[k][...] = h
                h = host_out[k]
        r.free()
        if plan.out_dtype != plan.dtype and host_out is None:
            h = h.astype(plan.out_dtype)
        host.append(h[()] if plan.all_scalar else h)
    for t in temps:
        t.free()
    return tuple(host)
