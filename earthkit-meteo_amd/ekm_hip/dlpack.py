"""DLPack exchange for DeviceArray (zero-copy interop with other ROCm array libraries).

`DeviceArray.__dlpack__()` / `__dlpack_device__()` export a C-contiguous float32/float64 array as a
`kDLROCM` tensor; `from_dlpack(obj)` wraps any `kDLROCM` producer's memory as a DeviceArray without
copying.  Pure ctypes: the DLPack structs (v0.x "dltensor" capsules) are laid out below.
"""
import ctypes as C

import numpy as np

import threading

from .device import DeviceArray, _capturing, _no_capture, _register_owner, current_device, current_stream, order_streams

kDLROCM = 10
kDLFloat = 2


class DLDevice(C.Structure):
    _fields_ = [("device_type", C.c_int32), ("device_id", C.c_int32)]


class DLDataType(C.Structure):
    _fields_ = [("code", C.c_uint8), ("bits", C.c_uint8), ("lanes", C.c_uint16)]


class DLTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("device", DLDevice), ("ndim", C.c_int32), ("dtype", DLDataType),
                ("shape", C.POINTER(C.c_int64)), ("strides", C.POINTER(C.c_int64)), ("byte_offset", C.c_uint64)]


class DLManagedTensor(C.Structure):
    pass


_DELETER = C.CFUNCTYPE(None, C.POINTER(DLManagedTensor))
DLManagedTensor._fields_ = [("dl_tensor", DLTensor), ("manager_ctx", C.c_void_p), ("deleter", _DELETER)]

_api = C.pythonapi
_api.PyCapsule_New.restype = C.py_object
_api.PyCapsule_New.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
_api.PyCapsule_IsValid.restype = C.c_int
_api.PyCapsule_IsValid.argtypes = [C.py_object, C.c_char_p]
_api.PyCapsule_GetPointer.restype = C.c_void_p
_api.PyCapsule_GetPointer.argtypes = [C.py_object, C.c_char_p]
_api.PyCapsule_SetName.restype = C.c_int
_api.PyCapsule_SetName.argtypes = [C.py_object, C.c_char_p]

# the capsule destructor runs while the capsule is being torn down: use prototypes that take the raw
# PyObject* and never touch its reference count
_raw_is_valid = C.PYFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p)(("PyCapsule_IsValid", C.pythonapi))
_raw_get_pointer = C.PYFUNCTYPE(C.c_void_p, C.c_void_p, C.c_char_p)(("PyCapsule_GetPointer", C.pythonapi))

_NAME, _USED = b"dltensor", b"used_dltensor"
_exports = {}  # address of an exported DLManagedTensor -> everything that must outlive the consumer


@_DELETER
def _export_deleter(managed_ptr):
    _exports.pop(C.addressof(managed_ptr.contents), None)


@C.CFUNCTYPE(None, C.c_void_p)
def _capsule_destructor(capsule_ptr):
    # a capsule nobody consumed still owns the tensor
    if _raw_is_valid(capsule_ptr, _NAME):
        _exports.pop(_raw_get_pointer(capsule_ptr, _NAME), None)


def to_dlpack(arr):
    """PyCapsule("dltensor") for a DeviceArray (the array stays alive until the consumer's deleter runs)."""
    shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
    m = DLManagedTensor()
    m.dl_tensor.data = arr.ptr
    m.dl_tensor.device = DLDevice(kDLROCM, arr.device)
    m.dl_tensor.ndim = arr.ndim
    m.dl_tensor.dtype = DLDataType(kDLFloat, arr.dtype.itemsize * 8, 1)
    m.dl_tensor.shape = C.cast(shape, C.POINTER(C.c_int64))
    m.dl_tensor.strides = None  # NULL = compact row-major
    m.dl_tensor.byte_offset = 0
    m.manager_ctx = None
    m.deleter = _export_deleter
    addr = C.addressof(m)
    _exports[addr] = (m, shape, arr)
    return _api.PyCapsule_New(addr, _NAME, C.cast(_capsule_destructor, C.c_void_p))


class _Borrowed:
    """Keeps a consumed DLManagedTensor alive; runs the producer's deleter when the last view goes away.
    Tracks the stream the memory was last used on here, like an allocation of our own (device._Allocation)."""

    def __init__(self, managed, device, stream):
        self.managed, self.device, self.stream = managed, device, stream
        self.pins, self.free_pending = 0, False  # recorded graphs that hold this memory's address (ekm_hip.graph)
        self._lock = threading.Lock()
        _register_owner(self)

    def touch(self, stream):
        cap = _capturing()
        if cap is not None:
            # recording (as device._Allocation.touch): the graph pins the tensor -- the producer's deleter runs only after
            # Graph.close() -- and orders itself after our other uses of it at every launch; no event is recorded on a
            # capturing stream.  What the PRODUCER does to the tensor between launches is on streams we do not know:
            # synchronise the producer before Graph.launch() (documented there).
            cap._adopt(self)
            with self._lock:
                self.stream = stream
            return
        with self._lock:
            if stream != self.stream:
                order_streams(self.device, self.stream, stream)
                self.stream = stream

    def free(self):
        if self.pins:  # a recorded graph still launches kernels on this address: released when the graph is closed
            self.free_pending = True
            return
        m, self.managed = self.managed, None
        if m is not None and m.deleter:
            # the producer may reuse the memory as soon as its deleter has run: our work on it must be complete
            from . import _ffi

            _ffi.check(_ffi.lib().ekm_stream_sync(self.device, self.stream))
            m.deleter(C.pointer(m))

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def from_dlpack(obj):
    """Wrap the memory of any object with `__dlpack__` (or a "dltensor" capsule) as a DeviceArray, zero-copy.

    This is the way in for arrays of other ROCm libraries (the reference selects its backend from the
    input type, `array_namespace(*inputs)`, thermo/array/thermo.py:826): `from_dlpack(torch_tensor)`.
    The producer is handed OUR current stream (array-API `__dlpack__(stream=...)`: 0 = default stream on
    ROCm), so it orders its pending work before anything we launch on that stream -- no host wait."""
    _no_capture("ekm_hip.from_dlpack")  # (the producer would be handed a stream that is recording)
    stream = current_stream()
    if hasattr(obj, "__dlpack__"):
        if hasattr(obj, "__dlpack_device__"):
            dl_type, dl_dev = obj.__dlpack_device__()
            if dl_type != kDLROCM:
                raise TypeError(f"from_dlpack: device type {dl_type} is not ROCm (kDLROCM = {kDLROCM})")
            if int(dl_dev) != current_device():
                stream = None  # our current stream belongs to another GPU: hand over on the tensor's default stream
        try:
            cap = obj.__dlpack__(stream=stream or 0)
        except TypeError:  # a producer without the stream argument
            cap = obj.__dlpack__()
    else:
        cap = obj
    if not _api.PyCapsule_IsValid(cap, _NAME):
        raise TypeError("from_dlpack: expected an object with __dlpack__ or an unused 'dltensor' capsule")
    m = DLManagedTensor.from_address(_api.PyCapsule_GetPointer(cap, _NAME))
    t = m.dl_tensor
    if t.device.device_type != kDLROCM:
        raise TypeError(f"from_dlpack: device type {t.device.device_type} is not ROCm (kDLROCM = {kDLROCM})")
    if t.dtype.code != kDLFloat or t.dtype.lanes != 1 or t.dtype.bits not in (32, 64):
        raise TypeError("from_dlpack: only float32 / float64 tensors are supported")
    shape = tuple(int(t.shape[i]) for i in range(t.ndim))
    if t.strides:
        expect = 1
        for i in range(t.ndim - 1, -1, -1):
            if shape[i] != 1 and int(t.strides[i]) != expect:
                raise ValueError("from_dlpack: only C-contiguous tensors are supported")
            expect *= shape[i]
    _api.PyCapsule_SetName(cap, _USED)  # we own the tensor now
    dtype = np.dtype(np.float32 if t.dtype.bits == 32 else np.float64)
    dev = int(t.device.device_id)
    return DeviceArray(_Borrowed(m, dev, stream), (t.data or 0) + int(t.byte_offset), shape, dtype, dev)
