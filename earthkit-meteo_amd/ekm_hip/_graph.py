"""HIP graphs for sequences of small calls: record once, replay with one launch.

A thermo call on a field of a million points keeps the GPU busy for ~10 us and the Python layer for ~30 us (argument
checks, ctypes, one allocation per result): a pipeline of such calls is bound by the host.  Inside `ekm_hip.graph()`
the calls are RECORDED -- their kernels, on their device pointers -- and `launch()` replays the whole sequence with one
HIP call:

    t, q, p = (ekm_hip.to_device(x) for x in (t0, q0, p0))
    with ekm_hip.graph() as g:                       # nothing runs in here
        rh = ekm_hip.thermo.relative_humidity_from_specific_humidity(t, q, p)
        td = ekm_hip.thermo.dewpoint_from_specific_humidity(q, p)
    for t_new in fields:
        t.copy_from_host(t_new)                      # same arrays, new contents
        g.launch()                                   # rh, td recomputed (asynchronously, stream-ordered)
        use(rh.to_host(), td.to_host())
    g.close()

Rules: every array operand inside the block is a DeviceArray, because uploads cannot be recorded (a Python scalar is fine:
its value is written by a recorded fill and is a constant of the graph; a 0-d DeviceArray is a scalar that can change
between launches); results hold no data until the first `launch()`; the graph keeps every array it touches alive
(and their addresses fixed) until `close()` -- a tensor of another library taken over with `ekm_hip.from_dlpack` BEFORE
the block included: its deleter runs after `close()`.  What that library does to such a tensor between launches runs on
streams this package does not know: synchronise it (`torch.cuda.synchronize()`) before `launch()`.  One thread records
and launches a graph: work another thread submits on another stream, on an array of the block, after the block began
is ordered before the first `launch()` only.  The reference has no counterpart: its functions run eagerly on the host."""
import ctypes as C

from . import _ffi
from . import device as _device


class Graph:
    """A recorded sequence of launches (see the module docstring).  Use as a context manager, then `launch()`."""

    def __init__(self, device=None):
        self.device = _device.current_device() if device is None else int(device)
        self.stream = None
        self._exec = None
        self._allocs = []      # every allocation the recorded kernels read or write: pinned until close()
        self._seen = set()
        self._prev_stream = None
        self._recording = False
        self.launches = 0

    # ---- recording ----
    def __enter__(self):
        if _device._capturing() is not None:
            raise _ffi.EkmError("ekm_hip.graph() blocks do not nest")
        if self._exec is not None or self.stream is not None:
            raise _ffi.EkmError("this graph has been recorded already; make a new one")
        if self.device != _device.current_device():
            raise _ffi.EkmError(f"ekm_hip.graph(device={self.device}) while the current device is {_device.current_device()}: "
                                "call ekm_hip.set_device first (the block's calls run on the current device)")
        lib = _ffi.lib()
        _ffi.check(lib.ekm_prepare_tables(self.device))   # lookup tables exist before the recording: no fill is recorded
        _ffi.check(lib.ekm_sync(self.device))             # every array enters the recording with its work complete
        self.stream = _device.stream_create(self.device)
        self._prev_stream = _device.current_stream()
        _device.set_stream(self.stream)
        try:
            _ffi.check(lib.ekm_graph_begin(self.device, self.stream))
        except Exception:
            _device.set_stream(self._prev_stream)
            _device.stream_destroy(self.stream, self.device)
            self.stream = None
            raise
        self._recording = True
        _device._tls.capture = self
        return self

    def __exit__(self, etype, evalue, tb):
        _device._tls.capture = None
        self._recording = False
        _device.set_stream(self._prev_stream)
        lib = _ffi.lib()
        out = C.c_void_p()
        rc = lib.ekm_graph_end(self.device, self.stream, C.byref(out) if etype is None else None)
        if etype is not None:
            self.close()
            return False
        if rc < 0:
            msg = lib.ekm_last_error().decode()
            self.close()
            raise _ffi.EkmError(msg)
        self._exec = out.value
        return False

    def _adopt(self, alloc):
        """Called by the allocator for every block created or used while recording."""
        if id(alloc) not in self._seen:
            self._seen.add(id(alloc))
            alloc.pins += 1
            self._allocs.append(alloc)

    # ---- replay ----
    def launch(self):
        """Replay the recorded launches (asynchronous).  The graph's stream is first ordered after the latest use of
        every array it reads or writes -- an upload into an input, a download of a result -- on whatever stream that was."""
        if self._exec is None:
            raise _ffi.EkmError("graph.launch(): nothing recorded (use `with ekm_hip.graph() as g:` first, launch after the block)")
        if _device._capturing() is not None:
            raise _ffi.EkmError("graph.launch() inside an ekm_hip.graph() block")
        for a in self._allocs:
            a.touch(self.stream)
        _ffi.check(_ffi.lib().ekm_graph_launch(self.device, self._exec, self.stream))
        self.launches += 1
        return self

    def synchronize(self):
        """Wait until every launch so far has finished."""
        if self.stream is not None:
            _ffi.check(_ffi.lib().ekm_stream_sync(self.device, self.stream))
        return self

    def close(self):
        """Destroy the graph and let go of its arrays (those still referenced elsewhere stay valid)."""
        if self.stream is None:
            return
        lib = _ffi.lib()
        stream, self.stream = self.stream, None
        try:
            _ffi.check(lib.ekm_stream_sync(self.device, stream))
            if self._exec is not None:
                ex, self._exec = self._exec, None
                _ffi.check(lib.ekm_graph_destroy(self.device, ex))
        finally:
            allocs, self._allocs, self._seen = self._allocs, [], set()
            for a in allocs:
                a.pins -= 1
                if a.pins == 0 and a.free_pending:
                    a.free_pending = False
                    a.free()
            del allocs
            _device.stream_destroy(stream, self.device)

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass


def graph(device=None):
    """`with ekm_hip.graph() as g: ...` -- record the thermo calls of the block into a HIP graph (see `ekm_hip.Graph`)."""
    return Graph(device)
