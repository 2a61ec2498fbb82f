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
import math
import os
import threading

import numpy as np

from . import _ffi
from ._optable import OPS
from .device import DeviceArray, _capturing, _no_capture, current_device, current_stream
from .vertical import HybridPressure

_F32, _F64 = np.dtype(np.float32), np.dtype(np.float64)
_MAX_LDS_BYTES = 32 * 1024  # staged level vectors per workgroup (csrc/map_kernel.hpp::kMaxLdsBytes); an op's own table comes on top
_MIN_VEC = 4  # a LEVEL operand must span at least one 16-B chunk per level
_PRETOUCH_BYTES = 8 << 20
_STREAM_BYTES = 256 << 20  # total input bytes from which a single-GPU NumPy call is streamed in slices


def _pretouch(arrays):
    """Fault in the pages of fresh result arrays from a helper thread while the inputs upload
    (ekm_host_prefault runs outside the GIL, on a few threads of its own)."""
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
    size = int(math.prod(shape))
    n = int(math.prod(out_shape))
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
        inner = int(math.prod(out_shape[hi + 1:]))
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
        self.n = int(math.prod(self.shape))


_KDLCPU, _KDLROCM, _KDLROCMHOST = 1, 10, 11


def _adopt_foreign(args):
    """Arrays of another array library are taken as the reference takes them: it selects its backend from the input type
    (`array_namespace(*inputs)`, thermo/array/thermo.py:826, es_comp.py:73) and returns that type.
      * a ROCm array (`__dlpack_device__()` says kDLROCM: a torch / cupy device tensor) is taken over through DLPack --
        zero copy, the producer ordered before our stream -- exactly as if the caller had written
        `ekm_hip.from_dlpack(x)`;
      * a HOST array of a foreign library (kDLCPU / kDLROCMHost: a torch CPU tensor) is viewed as a NumPy array through
        DLPack (zero copy; `np.asarray` where the producer refuses the export, e.g. a tensor that requires grad).
    Returns the arguments with such arrays replaced, and (library name, "device" | "host") when EVERY array argument came
    from that one library and one side (else None): the results then go back through that library."""
    mods, sides, plain, out = set(), set(), False, []
    for a in args:
        dd = None
        if not isinstance(a, (DeviceArray, HybridPressure, np.ndarray, np.generic, bool, int, float, list, tuple)):
            dd = getattr(a, "__dlpack_device__", None)
        kind = dd()[0] if dd is not None and hasattr(a, "__dlpack__") else None
        if kind == _KDLROCM:
            from .dlpack import from_dlpack

            if _capturing() is not None:
                raise _ffi.EkmError("inside an ekm_hip.graph() block the operands must be ekm_hip DeviceArrays; take another "
                                    "library's array over with ekm_hip.from_dlpack before the block")
            try:
                out.append(from_dlpack(a))
            except ValueError:  # not C-contiguous: the producer's own compaction, then zero copy
                if not hasattr(a, "contiguous"):
                    raise
                out.append(from_dlpack(a.contiguous()))
            mods.add(type(a).__module__.split(".")[0])
            sides.add("device")
        elif kind in (_KDLCPU, _KDLROCMHOST):
            try:
                out.append(np.from_dlpack(a))
            except Exception:  # the producer refuses (requires grad, exotic dtype ...): its own conversion
                out.append(np.asarray(a.detach() if hasattr(a, "detach") else a))
            mods.add(type(a).__module__.split(".")[0])
            sides.add("host")
        else:
            out.append(a)
            plain = plain or isinstance(a, (DeviceArray, HybridPressure)) or np.ndim(a) > 0
    one = len(mods) == 1 and len(sides) == 1 and not plain
    return out, ((mods.pop(), sides.pop()) if one else None)


def _hand_back(results, foreign):
    """Results for a caller whose inputs all came from one foreign library: through that library's own `from_dlpack`
    (zero copy; for device arrays it passes its current stream, which is then ordered after our kernel), NumPy scalars
    through its `asarray`.  The library is looked up among the modules the CALLER has imported -- the product never
    imports it; a library without these entry points gets our own types back."""
    import sys

    module, side = foreign
    mod = sys.modules.get(module)
    fn = getattr(mod, "from_dlpack", None)
    if fn is None:
        return results
    if side == "device":
        return tuple(fn(r) if isinstance(r, DeviceArray) else r for r in results)
    out = []
    for r in results:
        if isinstance(r, np.ndarray) and r.ndim > 0:
            if not r.flags.writeable:
                r = r.copy()
            out.append(fn(r))
        else:
            conv = getattr(mod, "asarray", None) or getattr(mod, "as_tensor", None)
            out.append(conv(r) if conv is not None else r)
    return tuple(out)


def run(name, args, ints=(), eps=None, dtype=None):
    """Launch entry point `name` on `args`; returns a tuple of outputs.  NumPy inputs that are large
    (>= 256 MB), or any NumPy inputs inside `ekm_hip.multi_gpu()`, are streamed through the GPU(s) in
    slices of their leading axis with a bounded device working set (`_streamed._run_streamed`)."""
    if dtype is None:
        recipe = _recipes.get(_recipe_key(name, args, ints))  # every operand a DeviceArray and the call seen before
        if recipe is not None and not _sharding():
            return _run_remembered(recipe, args, eps)
        if recipe is None:
            swapped = _scalars_on_device(args)
            if swapped is not None:
                args = swapped
                recipe = _recipes.get(_recipe_key(name, args, ints))
                if recipe is not None and not _sharding():
                    return _run_remembered(recipe, args, eps)
    args, foreign = _adopt_foreign(args)
    if foreign is not None:
        return _hand_back(_run(name, args, ints, eps, dtype), foreign)
    outs = _run(name, args, ints, eps, dtype)
    if dtype is None:
        outs = _as_the_reference_types_them(name, ints, args, outs)
    return outs


# ---- the reference's own result types on the NumPy path ---------------------------------------------------------------------
# Plain promotion with weak Python scalars is what the reference does where a function uses an operand as it comes; where
# it passes the operand through `xp.asarray` first the scalar is a float64 array and the result float64
# (potential_temperature(280.0, p_float32)); theta_w "direct" is float64 even from float32 arrays; temperature_on_moist_
# adiabat takes its type from theta_e alone; lcl's t_lcl depends on (t, td) only -- in type AND shape.  The deviations from
# plain promotion were recorded from the reference (tests/golden/gen_dtype_rules.py -> _dtype_rules.RULES, data only) and
# are applied to what a NumPy call returns; the arithmetic keeps the promotion's type (a float64 result of float32
# arithmetic is what the reference has there, too).  DeviceArrays and other libraries' arrays keep their own types.
from ._dtype_rules import RULES as _DTYPE_RULES  # noqa: E402  (data only)

# a result that is a scalar: np.float32 / np.float64, or -- calls on Python scalars alone -- a Python float where the reference's
# function is plain arithmetic on them ('p') and a 0-d array where it fills a result buffer ('a')
_SCALAR_AS = {"f": np.float32, "d": np.float64, "p": float, "a": np.asarray}
_SHAPED = frozenset(("lcl", "temperature_on_moist_adiabat", "wet_bulb_temperature_from_dewpoint", "wet_bulb_temperature_from_specific_humidity",
                     "wet_bulb_potential_temperature_from_dewpoint", "wet_bulb_potential_temperature_from_specific_humidity"))


def _kind_of(a):
    ta = type(a)
    if ta is np.ndarray:
        return "f" if a.dtype == _F32 else "d"
    if ta in (float, int, bool):
        return "s"
    if isinstance(a, (np.ndarray, np.generic, list, tuple)):
        return "f" if np.asarray(a).dtype == _F32 else "d"
    return None


def _as_the_reference_types_them(name, ints, args, outs):
    if type(ints) is not tuple or (ints and type(ints[0]) is not int):
        ints = tuple(int(i) for i in ints)
    rules = _DTYPE_RULES.get((name, ints))
    lcl = one_d = like_ept = False
    if name in _SHAPED:
        lcl = name == "lcl"
        # the reference's bisection works on atleast_1d(theta_e): operands that are all 0-d come back with shape (1,), not ()
        one_d = not lcl and len(ints) == 2 and ints[1] == 0 and len(outs) == 1 and np.ndim(outs[0]) == 0
        # ... and temperature_on_moist_adiabat writes into an array of theta_e's shape: a one-element p of more dimensions ([1, 1]
        # beside theta_e of [n]) does not add them (where p has more elements than that the reference raises)
        like_ept = (name == "temperature_on_moist_adiabat" and len(outs) == 1 and np.ndim(args[1]) > np.ndim(args[0]) >= 1
                    and np.size(args[1]) == 1 and isinstance(outs[0], np.ndarray))
    if rules is None and not (lcl or one_d or like_ept):
        return outs
    kinds = ""
    for a in args:
        k = _kind_of(a)
        if k is None:
            return outs
        kinds += k
    for o in outs:
        if not isinstance(o, (np.ndarray, np.generic)):
            return outs
    chars = rules.get(kinds) if rules is not None else None
    if chars is None and rules is not None and np.ndim(outs[0]) == 0:
        # a function that fills a result buffer returns a 0-d ARRAY for 0-d operands of any kind (recorded for Python scalars)
        alone = rules.get("s" * len(args))
        if alone is not None and "a" in alone:
            chars = "".join("a" if c == "a" else ("f" if np.asarray(o).dtype == _F32 else "d") for c, o in zip(alone, outs))
    if chars is None and not (lcl or one_d or like_ept):
        return outs
    outs = list(outs)
    if one_d:
        outs[0] = np.asarray(outs[0]).reshape(1)
    if like_ept:
        outs[0] = outs[0].reshape(np.shape(args[0]))
    if chars is not None:
        outs = [np.asarray(o).astype(_F32 if c == "f" else _F64, copy=False) if np.ndim(o) else _SCALAR_AS[c](o) for o, c in zip(outs, chars)]
    if lcl and np.ndim(outs[0]):  # t_lcl has the shape of broadcast(t, td); the values do not vary along p's own axes
        full = np.shape(outs[0])
        want = np.broadcast_shapes(np.shape(args[0]), np.shape(args[1]))
        if tuple(want) != tuple(full):
            padded = (1,) * (len(full) - len(want)) + tuple(want)
            t_lcl = np.asarray(outs[0])[tuple(slice(None) if w == f else slice(0, 1) for w, f in zip(padded, full))].reshape(want)
            outs[0] = t_lcl.copy() if t_lcl.ndim else t_lcl.dtype.type(t_lcl)
    if lcl and kinds[:2] == "ss" and np.ndim(outs[0]) == 0 and rules is not None and rules.get("sss", "d")[0] == "p":
        outs[0] = float(outs[0])  # davies' t_lcl of two Python floats is plain arithmetic on them: a Python float, whatever p is
    return tuple(outs)


# ---- Python scalars beside DeviceArrays ---------------------------------------------------------------------------------
# `potential_temperature(t_dev, 85000.0)`: the scalar used to reach the kernel through the general path at every call (a
# block, an asynchronous fill of its bit pattern, the whole plan: 37 us against 11 for the same call on two DeviceArrays).
# Its value on the device is remembered instead -- one 0-d DeviceArray per (device, dtype, bit pattern), filled once -- and
# the call is then a call on DeviceArrays, which has a recipe.  Only where that changes nothing: every array operand a
# DeviceArray of ONE dtype on ONE device (a Python scalar is weak: the arrays' dtype wins, as in the reference), not while
# recording (there the fill is a node of the graph and the value a constant of it), not inside multi_gpu().
# One 0-d device array per (device, STREAM, dtype, bit pattern): a cached scalar is only ever read by kernels of the stream
# that wrote it, so when an entry is dropped its block goes back to that stream's block cache and any reuse is ordered
# behind those kernels (shared across streams -- rounds 4-5 -- a block re-filed under the last toucher's stream could be
# overwritten while a kernel on another stream still read it: ADVICE r5).  Least recently used entries go first.
import collections as _collections

_scalar_cache = _collections.OrderedDict()
_SCALARS_MAX = 256


def _scalars_on_device(args):
    dt = dev = None
    has_scalar = False
    for a in args:
        if type(a) is DeviceArray:
            if dt is None:
                dt, dev = a.dtype, a.device
            elif a.dtype != dt or a.device != dev:
                return None
        elif type(a) in (float, int):
            has_scalar = True
        else:
            return None
    if not has_scalar or dt is None or _capturing() is not None or _sharding():
        return None
    out = []
    for a in args:
        if type(a) is DeviceArray:
            out.append(a)
            continue
        bits = np.asarray(a, dtype=dt).tobytes()
        stream = current_stream()
        key = (dev, stream, dt.char, bits)
        d = _scalar_cache.get(key)
        if d is not None:
            _scalar_cache.move_to_end(key)
        else:
            while len(_scalar_cache) >= _SCALARS_MAX:
                _scalar_cache.popitem(last=False)
            d = DeviceArray.empty((), dt, dev)
            words = np.frombuffer(bits, dtype=np.uint32)
            base = d.on(stream)
            for w in range(words.size):
                _ffi.check(_ffi.lib().ekm_fill_u32(dev, base + 4 * w, int(words[w]), 1, stream))
            _scalar_cache[key] = d
        out.append(d)
    return tuple(out)


# ---- the device-resident call, remembered ---------------------------------------------------------------------------
# A thermo call on DeviceArrays is one kernel launch of ~10 us on a field of a million points; planning it (dtype
# promotion, broadcasting, how each operand reaches the kernel) costs more than that in Python.  The plan depends only on
# the entry point, its enum arguments and each operand's (shape, dtype, device): the first such call goes the general way
# and leaves a recipe behind (`_submit`), later ones with the same key launch from it.
_recipes = {}
_RECIPES_MAX = 512


class _Recipe:
    __slots__ = ("fn", "device", "dtype", "shape", "n", "nbytes", "classes", "ints", "nout", "has_eps")


def _recipe_key(name, args, ints):
    try:
        return (name, tuple(ints)) + tuple([(a.shape, a.dtype.char, a.device) if type(a) is DeviceArray else _NOT_DEVICE for a in args])
    except TypeError:  # an unhashable enum argument: the general path reports it
        return None


_NOT_DEVICE = object()  # never part of a stored key: a call with any other operand type misses


def _sharding():
    """Inside `multi_gpu()` with more than one device (the general path explains why device-resident inputs cannot be sharded)."""
    from .device import current_devices

    devs = current_devices()
    return bool(devs) and len(devs) > 1


def _run_remembered(rec, args, eps):
    stream = current_stream()
    operands = [_ffi.Operand(a.on(stream), c[0], 0, c[1], c[2]) for a, c in zip(args, rec.classes)]
    results = [DeviceArray._new(rec.shape, rec.dtype, rec.device, rec.nbytes) for _ in range(rec.nout)]
    cargs = [rec.device, stream]
    cargs += [C.byref(o) for o in operands]
    cargs += rec.ints
    if rec.has_eps:
        cargs.append(float(eps))
    cargs += [r.ptr for r in results]
    cargs.append(rec.n)
    _ffi.check(rec.fn(*cargs))
    return tuple(results)


# ---- tiny NumPy calls -------------------------------------------------------------------------------------------------
# `potential_temperature(np.array([264.12, 261.45]), np.array([85000., 85000.]))` -- the reference's quick start, and the
# shape of its whole test suite: a few elements in, a few out.  The general path costs such a call two uploads, a launch,
# a download and a wait plus a device block per array (100 us).  Here the operands are written into ONE pinned
# host block per thread and device, the kernel reads them and writes its results THERE (pinned host memory is mapped into
# the device's address space: for a few KiB the link's latency is nothing beside two copies), and the call is a launch and
# a wait.  What the plan of such a call depends on -- entry point, enum arguments, each operand's shape and dtype -- is
# remembered like a device-resident call's.
_TINY_BYTES = 64 << 10     # inputs + outputs in the compute dtype
_tiny_recipes = {}
_tiny_tls = threading.local()


class _TinyRecipe:
    __slots__ = ("fn", "cdtype", "out_dtype", "shape", "n", "classes", "sizes", "offsets", "out_offsets", "nout", "has_eps",
                 "ints", "all_scalar", "nbytes")


def _tiny_key(name, args, ints):
    parts = []
    for a in args:
        ta = type(a)
        if ta is np.ndarray:
            if a.size > _TINY_BYTES // 4:
                return None
            parts.append((a.shape, a.dtype.str))
        elif ta is float or ta is int:
            parts.append(ta)
        elif isinstance(a, np.generic):
            parts.append(((), a.dtype.str))
        else:
            return None  # lists, DeviceArrays, another library's arrays, HybridPressure: the general path
    try:
        return (name, tuple(ints)) + tuple(parts)
    except TypeError:
        return None


def _tiny_plan(name, args, ints):
    ins, outs, int_names, has_eps = OPS[name]
    plan = _Plan(args, None)
    if plan.on_device or plan.hybrid or plan.n == 0:
        return None
    item = plan.dtype.itemsize
    classes, sizes, offsets, off = [], [], [], 0
    for h in plan.host:
        cls = classify(h.shape, plan.shape) if plan.n else (_ffi.FIELD, 0, 0)
        if cls is None:
            return None  # a broadcast pattern the general path materialises
        classes.append(cls)
        sizes.append(int(h.size))
        offsets.append(off)
        off += (h.size * item + 255) & ~255
    out_offsets = []
    for _ in outs:
        out_offsets.append(off)
        off += (plan.n * item + 255) & ~255
    if off > _TINY_BYTES:
        return None
    rec = _TinyRecipe()
    rec.fn = getattr(_ffi.lib(), f"ekm_{name}_{'f32' if plan.dtype == _F32 else 'f64'}")
    rec.cdtype, rec.out_dtype, rec.shape, rec.n = plan.dtype, plan.out_dtype, plan.shape, plan.n
    rec.classes, rec.sizes, rec.offsets, rec.out_offsets = classes, sizes, offsets, out_offsets
    rec.nout, rec.has_eps, rec.ints, rec.all_scalar, rec.nbytes = len(outs), has_eps, [int(v) for v in ints], plan.all_scalar, off
    return rec


class _TinyBlock:
    """64 KiB of pinned host memory, owned by one thread's state for one device; freed with it."""

    def __init__(self):
        ptr = C.c_void_p()
        _ffi.check(_ffi.lib().ekm_host_alloc(_TINY_BYTES, C.byref(ptr)))
        self.ptr = ptr.value
        self.buf = (C.c_char * _TINY_BYTES).from_address(self.ptr)

    def __del__(self):
        try:
            if self.ptr:
                self.buf = None
                _ffi.lib().ekm_host_free(self.ptr)
                self.ptr = None
        except Exception:  # interpreter shutdown
            pass


def _tiny_block(dev):
    blocks = getattr(_tiny_tls, "blocks", None)
    if blocks is None:
        blocks = _tiny_tls.blocks = {}
    b = blocks.get(dev)
    if b is None:
        b = blocks[dev] = _TinyBlock()
    return b


def _run_tiny(rec, args, eps):
    dev, stream = current_device(), current_stream()
    blk = _tiny_block(dev)
    base, buf = blk.ptr, blk.buf
    cargs = [dev, stream]
    for a, cls, size, off in zip(args, rec.classes, rec.sizes, rec.offsets):
        np.frombuffer(buf, dtype=rec.cdtype, count=size, offset=off)[...] = np.reshape(a, -1) if type(a) is np.ndarray else a
        cargs.append(C.byref(_ffi.Operand(base + off, cls[0], 0, cls[1], cls[2])))
    cargs += rec.ints
    if rec.has_eps:
        cargs.append(float(eps))
    cargs += [base + off for off in rec.out_offsets]
    cargs.append(rec.n)
    lib = _ffi.lib()
    _ffi.check(rec.fn(*cargs))
    _ffi.check(lib.ekm_stream_sync(dev, stream))
    out = []
    for off in rec.out_offsets:
        h = np.frombuffer(buf, dtype=rec.cdtype, count=rec.n, offset=off).reshape(rec.shape).astype(rec.out_dtype)  # a copy: fresh arrays
        out.append(h[()] if rec.all_scalar else h)
    return tuple(out)


def _run(name, args, ints, eps, dtype):
    from .device import current_devices

    devs = current_devices()
    if dtype is None and _capturing() is None and not (devs and len(devs) > 1):
        key = _tiny_key(name, args, ints)
        if key is not None:
            rec = _tiny_recipes.get(key)
            if rec is None and key not in _tiny_recipes:
                if len(_tiny_recipes) >= _RECIPES_MAX:
                    _tiny_recipes.clear()
                rec = _tiny_recipes[key] = _tiny_plan(name, args, ints)  # None: not a tiny call (remembered as such)
            if rec is not None:
                return _run_tiny(rec, args, eps)
    numpy_only = not any(isinstance(a, (DeviceArray, HybridPressure)) for a in args)
    if devs and len(devs) > 1 and not numpy_only:
        # a DeviceArray lives on ONE GPU: computing on it inside multi_gpu() would silently use that single device
        where = sorted({a.device for a in args if isinstance(a, DeviceArray)}
                       | {a.sp.device for a in args if isinstance(a, HybridPressure) and isinstance(a.sp, DeviceArray)})
        if where:
            raise _ffi.EkmError(f"{name}: inside ekm_hip.multi_gpu(devices={list(devs)}) the inputs must be host (NumPy) arrays, which "
                                f"are sharded across the GPUs; these are device-resident on GPU {where} -- call it outside the "
                                f"block, or shard by hand with ekm_hip.shard_bounds and one DeviceArray per device")
    if numpy_only:
        multi = bool(devs) and len(devs) > 1
        if multi or sum(np.asarray(a).nbytes for a in args if np.ndim(a) > 0) >= _STREAM_BYTES:
            from ._streamed import _run_streamed  # (imports this module: late)

            streamed = _run_streamed(name, args, ints, eps, dtype, devs if multi else [current_device()])
            if streamed is not None:
                return streamed
    return _run_single(name, args, ints, eps, dtype)


# Streamed path: plain hipMemcpyAsync per operand, and the RESULTS of big calls in pinned host memory from a recycling pool
# (device.pinned_empty) so that the downloads are plain DMAs: P3 on 8 levels 61 -> 80 GB/s, on 32 levels 74 -> 85 GB/s both
# directions together (profiles/r03_host_path_rate.txt; link capacity 57 GB/s per direction, 96 both ways:
# profiles/r03_host_link_probe.txt).  Three other routes were built, tested bit-equal and measured in round 3 -- transfers
# staged through a ring of pinned buffers (44-57 GB/s in the pipeline), the caller's memory pinned in place slice by slice
# (31-44), both together (43-54) -- and removed in round 4: none came near the default.
# Pinned memory is page-locked: it cannot be swapped and counts against container and memlock limits.  So the pool is
# bounded -- blocks held by callers plus blocks cached never exceed EKM_PINNED_CACHE_BYTES in total (default min(25 % of
# MemAvailable, 16 GiB): device._PinnedPool) --, and a call's results go there only if the pool would keep them
# afterwards (EKM_PINNED_RESULTS_BYTES, default = that limit): blocks the pool cannot keep are pinned anew by every call,
# and pinning is slow -- with a 1-GiB cache the 2.5 GB of results of a 32-level P3 call took 401 ms per call from the pool
# against 82 ms as ordinary arrays (profiles/r04_host_path_rate.txt).  Beyond the limit results are ordinary arrays,
# prefaulted from a helper thread.  A result in pooled memory does not own its data (`.base` is a ctypes buffer;
# `ndarray.resize` refuses): EKM_PINNED_RESULTS=0 turns the pool off.
_PINNED_OUT = os.environ.get("EKM_PINNED_RESULTS", "1") != "0"
# a call's results go to the pinned pool only if the pool would KEEP them afterwards: the per-call limit is the cache limit
from .device import _pinned as _pinned_pool  # noqa: E402

_PINNED_OUT_BYTES = int(os.environ.get("EKM_PINNED_RESULTS_BYTES", str(_pinned_pool.limit)))
# Round 6, OPT-IN (EKM_DIRECT_RESULTS=1): results that live in pooled pinned memory are WRITTEN THERE BY THE KERNEL (pinned host
# memory is mapped into the device's address space: the output pointers of the launch are the host blocks) instead of into
# device blocks that a DMA engine then copies down; the kernel's stores cross PCIe as they are issued while the next slice's
# uploads run on their DMA engine.  Same bits (tests/test_gpu_streaming.py).  Measured (profiles/r06_host_path_rate.txt, one
# box, 8 / 32 levels, both directions together): the six-output pipeline 69.9 / 72.8 GB/s against 68.2 / 72.1 with downloads,
# es-td-rh 74.8 / 80.9 against 80.3 / 88.5, theta 62.2 / 68.4 against 67.1 / 76.0 -- shader stores to host memory reach a
# little less than a DMA engine does, so where as much goes up as comes down the downloads win: off by default.  One device
# only (a block pinned through one device's context), compute dtype = result dtype.
_DIRECT_OUT = os.environ.get("EKM_DIRECT_RESULTS", "0") == "1"


class _Pending:
    """One submitted launch: device results (and the temporaries its operands live in) not yet collected."""

    __slots__ = ("plan", "results", "temps", "host_out", "internal_out", "toucher", "stream", "keep", "direct")


def _reserved(nbytes, reserve_rows):
    if not reserve_rows or reserve_rows[0] <= 0:
        return 0
    rows, most = reserve_rows
    return -(-nbytes // rows) * most if nbytes % rows == 0 else 0  # only blocks that scale with the slice's rows


def _run_single(name, args, ints=(), eps=None, dtype=None, host_out=None, toucher=None):
    """One launch.  `host_out` (caller-owned NumPy destinations, e.g. slices of a result array) receives the
    outputs directly; `toucher` is a thread prefaulting them, joined before anything is downloaded."""
    return _collect(_submit(name, args, ints, eps, dtype, host_out, toucher))


def _submit(name, args, ints=(), eps=None, dtype=None, host_out=None, toucher=None, reserve_rows=None, uploaded=None,
            pinned_out=False):
    """Upload what lives on the host, launch, return without waiting for the kernel (the uploads themselves are
    synchronous copies on the current stream).  `reserve_rows` = (rows of this slice, rows of the longest slice):
    field-sized device blocks are reserved at the longest slice's size (streamed path).  `uploaded`: an event handle
    recorded on the stream behind the last upload and BEFORE the launch (the streamed path waits for a slice's uploads,
    not for its kernel).  `pinned_out`: the caller's `host_out` arrays are pooled pinned memory -- the kernel may write
    its results there directly (_DIRECT_OUT)."""
    ins, outs, int_names, has_eps = OPS[name]
    assert len(args) == len(ins) and len(ints) == len(int_names)
    plan = _Plan(args, dtype)
    lib = _ffi.lib()
    dev, stream = plan.device, current_stream()
    tag = "f32" if plan.dtype == _F32 else "f64"
    fn = getattr(lib, f"ekm_{name}_{tag}")

    temps = []  # device buffers owned by this call
    keep = []   # host arrays an upload still reads from (kept until the results are collected)
    operands = []
    lds_bytes = 0
    internal_out = False  # host_out allocated here (compute dtype) rather than supplied by the caller
    if not plan.on_device and host_out is None and plan.n * plan.dtype.itemsize >= _PRETOUCH_BYTES:
        # Large NumPy result: the pages of a fresh array fault on first touch, which slows the download
        # from 56 to ~24 GB/s.  Fault them in from a helper thread while the inputs are uploading
        # (ekm_host_prefault never changes the contents, and it is joined before the download anyway).
        # ... or, better, take the result arrays from the pool of pinned host blocks (device.pinned_empty): no faults at
        # all and a plain DMA (see _streamed._run_streamed).
        host_out = None
        if _PINNED_OUT and len(outs) * plan.n * plan.dtype.itemsize <= _PINNED_OUT_BYTES:
            from .device import pinned_empty

            host_out = [pinned_empty(plan.shape, plan.dtype) for _ in outs]
            if any(h is None for h in host_out):
                host_out = None
            else:
                pinned_out = True
        if host_out is None:
            host_out = [np.empty(plan.shape, plan.dtype) for _ in outs]
            toucher = _pretouch(host_out)
        internal_out = True
    for k, a in enumerate(plan.host):
        if plan.hybrid and k == len(plan.host) - 1:
            hp = plan.hybrid[0]
            sp = a if isinstance(a, DeviceArray) and a.dtype == plan.dtype else DeviceArray.from_host(
                np.ascontiguousarray(np.asarray(a), dtype=plan.dtype), device=dev)
            sp.on(stream)
            tabs = hp.device_tables(dev, plan.dtype)  # uploaded once per (device, dtype), owned by the HybridPressure
            if sp is not a:
                temps.append(sp)
            inner = max(1, sp.size)
            # (a B that is zero in double stays zero in the compute dtype, so the count holds for the uploaded table)
            operands.append(_ffi.Operand(sp.ptr, _ffi.HYBRID_FULL, hp.nflat, hp.nlev, inner, tabs[0].on(stream), tabs[1].on(stream)))
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
        elif cls is not None and cls[0] == _ffi.SCALAR and not isinstance(a, DeviceArray):
            # a host scalar: its bit pattern written by an asynchronous fill (one 32-bit word, two for fp64) -- no host
            # buffer to keep alive, no synchronous copy (which cost more than the kernel of a small call), and
            # recordable inside ekm_hip.graph()
            words = np.asarray(a, dtype=plan.dtype).reshape(1).view(np.uint32)
            darr = DeviceArray.empty((), plan.dtype, dev)
            base = darr.on(stream)
            for w in range(words.size):
                _ffi.check(lib.ekm_fill_u32(dev, base + 4 * w, int(words[w]), 1, stream))
            temps.append(darr)
        else:
            h = np.asarray(a)  # DeviceArray -> host copy only on the slow path
            if cls is None:
                h = np.broadcast_to(h.reshape(_padded(h.shape, len(plan.shape))), plan.shape)
                cls = (_ffi.FIELD, 0, 0)
            hc = np.ascontiguousarray(h, dtype=plan.dtype)
            cap = _reserved(hc.size * plan.dtype.itemsize, reserve_rows)
            # the upload is queued, not waited for: from pageable memory the call returns once the data is staged, from
            # pinned memory (ekm_hip.pinned_empty) at once -- the operands of a call go up back to back, the kernel follows
            # on the same stream, and the one host wait is the download's (`hc` stays alive until then)
            _no_capture("an upload (a NumPy operand)")
            darr = DeviceArray.empty(hc.shape, hc.dtype, dev, capacity=cap)
            if hc.nbytes:
                _ffi.check(lib.ekm_h2d(dev, darr.on(stream), hc.ctypes.data, hc.nbytes, stream))
            keep.append(hc)
            temps.append(darr)
        # .on(stream): an input last used on another stream makes this stream wait for that work (device-side)
        operands.append(_ffi.Operand(darr.on(stream), cls[0], 0, cls[1], cls[2]))

    # results straight into the caller's pinned host blocks (see _DIRECT_OUT), else into device blocks that are downloaded
    direct = bool(_DIRECT_OUT and pinned_out and host_out is not None and not plan.on_device and plan.n and _capturing() is None
                  and not _sharding() and all(h.dtype == plan.dtype and h.flags.c_contiguous and h.size == plan.n for h in host_out))
    results = [] if direct else [
        DeviceArray.empty(plan.shape, plan.dtype, dev, capacity=_reserved(plan.n * plan.dtype.itemsize, reserve_rows)) for _ in outs]
    cargs = [dev, stream] + [C.byref(o) for o in operands] + [int(v) for v in ints]
    if has_eps:
        cargs.append(float(eps))
    cargs += ([h.ctypes.data for h in host_out] if direct else [r.ptr for r in results]) + [plan.n]
    if uploaded is not None:
        _ffi.check(lib.ekm_event_record(dev, uploaded, stream))
    _ffi.check(fn(*cargs))
    if (not temps and plan.on_device and not plan.hybrid and dtype is None and host_out is None and reserve_rows is None
            and plan.n and len(_recipes) < _RECIPES_MAX and all(type(a) is DeviceArray for a in args)):
        rec = _Recipe()
        rec.fn, rec.device, rec.dtype, rec.shape, rec.n = fn, dev, plan.dtype, plan.shape, plan.n
        rec.nbytes, rec.nout, rec.has_eps = plan.n * plan.dtype.itemsize, len(outs), has_eps
        rec.classes, rec.ints = [(o.mode, o.len, o.inner) for o in operands], [int(v) for v in ints]
        key = _recipe_key(name, args, ints)
        if key is not None:
            _recipes[key] = rec
    pend = _Pending()
    pend.plan, pend.results, pend.temps, pend.host_out = plan, results, temps, host_out
    pend.internal_out, pend.toucher, pend.stream, pend.keep, pend.direct = internal_out, toucher, stream, keep, direct
    return pend


def _collect(pend, refile=None):
    """Results of a submitted launch: DeviceArrays as they are, NumPy results downloaded on the CURRENT stream -- the
    stream the launch was submitted on, or another one (`DeviceArray.on` orders it behind the kernel on the device).
    `refile`: the stream whose block cache the result blocks go back to (the streamed path downloads on a stream of its
    own and hands the blocks back to the lane that will take them again; the download has completed on the host by then)."""
    plan, results, temps, host_out = pend.plan, pend.results, pend.temps, pend.host_out
    internal_out, toucher = pend.internal_out, pend.toucher
    if plan.on_device:
        if pend.keep:  # a NumPy operand among DeviceArrays: its upload must have left the host array before we let go of it
            _ffi.check(_ffi.lib().ekm_stream_sync(plan.device, pend.stream))
            pend.keep = None
        # temporaries are freed by HIP in stream order after the kernel has run
        for t in temps:
            t.free()
        return tuple(results)

    if toucher is not None:
        toucher.join()
    if pend.direct:  # the kernel wrote into the pinned host blocks: wait for IT (on the stream it runs on), nothing to copy
        _ffi.check(_ffi.lib().ekm_stream_sync(plan.device, pend.stream))
        pend.keep = None
        for t in temps:
            t.free()
        host = []
        for h in host_out:
            if plan.out_dtype != plan.dtype and internal_out:
                h = h.astype(plan.out_dtype)
            host.append(h[()] if plan.all_scalar else h)
        return tuple(host)
    host = []
    # the downloads of a call are queued back to back and waited for ONCE
    direct = [host_out is not None and host_out[k].dtype == plan.dtype and host_out[k].flags.c_contiguous for k in range(len(results))]
    staged = [r.to_host(out=host_out[k] if direct[k] else None, sync=False) for k, r in enumerate(results)]
    if results:
        _ffi.check(_ffi.lib().ekm_stream_sync(plan.device, current_stream()))
    pend.keep = None
    for k, r in enumerate(results):
        h = staged[k]
        if not direct[k] and host_out is not None:
            host_out[k][...] = h
            h = host_out[k]
        if refile is not None:
            r._alloc.stream = refile  # (the copies have been waited for: nothing is pending on the block)
        r.free()
        if plan.out_dtype != plan.dtype and (host_out is None or internal_out):
            h = h.astype(plan.out_dtype)
        host.append(h[()] if plan.all_scalar else h)
    for t in temps:
        t.free()
    return tuple(host)
