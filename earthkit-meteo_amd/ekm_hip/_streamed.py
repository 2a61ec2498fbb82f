"""The NumPy-in / NumPy-out path for big calls and for several GPUs: the broadcast result is cut along its leading axis and
the slices are pipelined through the GPU(s) -- an uploader and a downloader thread per device, `lanes` slices resident
between them on their own streams, results written straight into (pinned, pooled) host arrays.  `_engine._run` comes here
from 256 MB of NumPy input or inside `ekm_hip.multi_gpu()`; one slice is one `_engine._submit` / `_collect`."""
import ctypes as C
import math
import os
import threading

import numpy as np

from . import _engine, _ffi
from ._engine import _PRETOUCH_BYTES, _collect, _result_dtype, _submit
from ._optable import OPS


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
_streams_lock = threading.Lock()
_MAX_LANES = 8            # slices in flight per GPU when memory allows (upload of one overlaps download of another)
_MIN_SLICE_BYTES = 16 << 20  # do not cut finer than this: small copies waste PCIe bandwidth
_BLOCK_OVERHEAD = 1 << 20  # worst-case rounding of one device block (the allocator's buckets are multiples of 1 MiB)


def _lane_stream(dev, slot):
    """One long-lived stream per (device, lane slot 0.._MAX_LANES-1) plus one per device for the downloads (slot
    "download"); `release_streams()` destroys them."""
    from .device import stream_create

    with _streams_lock:
        k = (dev, slot)
        if k not in _streams:
            _streams[k] = stream_create(dev)
        return _streams[k]


_events = {}  # (device, lane slot) -> event recorded behind a slice's uploads (one per lane: the uploader waits for it each time)


def _uploaded_event(dev, slot):
    import ctypes as C

    with _streams_lock:
        k = (dev, slot)
        if k not in _events:
            ev = C.c_void_p()
            _ffi.check(_ffi.lib().ekm_event_create(dev, C.byref(ev)))
            _events[k] = ev
        return _events[k]


def release_streams():
    """Destroy the lane streams of the streamed NumPy path (and return their cached blocks to HIP)."""
    from .device import stream_destroy

    with _streams_lock:
        items = list(_streams.items())
        _streams.clear()
        events = list(_events.items())
        _events.clear()
    for (dev, _slot), ev in events:
        _ffi.lib().ekm_event_destroy(dev, ev)
    for (dev, _slot), st in items:
        stream_destroy(st, dev)


def stream_budget_bytes(dev):
    """Device bytes the streamed path may hold in flight on `dev`: 80 % of what is free now plus what our own
    block cache holds, capped by EKM_STREAM_BUDGET_BYTES when set."""
    from .device import _cache

    free, total = C.c_size_t(), C.c_size_t()
    _ffi.check(_ffi.lib().ekm_mem_info(dev, C.byref(free), C.byref(total)))
    budget = int(0.8 * (free.value + _cache.cached_bytes(dev)))
    cap = os.environ.get("EKM_STREAM_BUDGET_BYTES")
    return min(budget, int(cap)) if cap else budget


def plan_slices(rows, row_bytes, budget, max_lanes=_MAX_LANES, min_slice=_MIN_SLICE_BYTES, overhead=0,
                pref_slice=256 << 20):
    """How to stream `rows` leading-axis rows of `row_bytes` device bytes each (inputs + outputs) through a
    device working set of at most `budget` bytes: returns (lanes, nslices) -- `lanes` slices are in flight at
    a time, each lane recycling its device blocks from slice to slice, and
    lanes * ceil(rows / nslices) * row_bytes <= budget.
      * everything fits: up to `max_lanes` lanes, slices of at least `min_slice` bytes (small copies waste PCIe
        bandwidth; a small call is one slice) and, for big calls, of about `pref_slice` bytes (more slices than lanes:
        a short pipeline ramp; the lanes recycle their device blocks);
      * it does not fit: as many lanes as the budget allows with slices of at least `min_slice` (8, 4), at
        least two (double buffering) whatever the slice size; None if two single-row slices do not fit.
    `overhead`: device bytes every in-flight slice costs on top of its rows (allocator rounding)."""
    total = rows * row_bytes
    if total == 0:
        return 1, 1
    lanes = int(max(1, min(max_lanes, rows, total // min_slice)))
    if total + lanes * overhead <= budget:
        # everything fits.  More slices than lanes still pay: the transfers of the first slice up and of the last
        # slice down overlap with nothing, so slices of about `pref_slice` bytes keep that ramp short for big calls
        nslices = int(max(lanes, min(rows, total // pref_slice)))
        return lanes, nslices
    for lanes in (8, 4, 2):
        if lanes > max(max_lanes, 2) or lanes > rows:
            continue
        rows_per = max(0, budget // lanes - overhead) // row_bytes
        if rows_per < 1 or (lanes > 2 and rows_per * row_bytes < min_slice):
            continue
        return lanes, -(-rows // rows_per)
    return None




def _run_streamed(name, args, ints, eps, dtype, devs):
    """Grid points are independent: cut the broadcast result along its leading axis, give every operand that
    spans that axis the matching slice and every other operand (scalars, trailing-axis vectors) whole, and
    stream the slices through the GPU(s) straight into slices of the result arrays.

    * several GPUs (`multi_gpu()`): one contiguous block of rows per GPU (~17 whole levels each for
      [137, lat, lon] fields on 8 GPUs), no exchange of any kind;
    * per GPU: an uploader thread (slice k: upload its operands, launch, hand over) and a downloader thread (slice
      k: download its results, free its device blocks) with `lanes` slices resident between them, uploads and
      kernels on the stream (device, k mod lanes), downloads on ONE stream of their own: exactly one upload and one
      download are in flight at a time -- PCIe is full duplex, but concurrent pageable uploads collapse -- and the
      kernels run under both.  (The download stream matters: HIP binds a stream to a DMA engine at its first copy, every
      lane stream's first copy is an upload, and downloads issued on the lane streams shared that one engine with the
      next slice's upload -- 29 + 29 GB/s where the link does 54 + 47 in both directions at once; round 5,
      profiles/r05_host_path_rate.txt); a collected slice's
      device blocks go back to the block cache of its stream and are taken again `lanes` slices later, so the
      device working set is lanes x slice, chosen to fit `stream_budget_bytes` (fields larger than HBM stream
      through; with a tight budget this degrades to two resident slices, i.e. double buffering)."""
    import queue

    from .device import set_device, set_stream

    host = [np.asarray(a) for a in args]
    shape = tuple(np.broadcast_shapes(*[h.shape for h in host]))
    if len(shape) == 0 or shape[0] < len(devs):
        return None
    out_dtype, cdtype = _result_dtype(args)
    if dtype is not None:
        out_dtype = cdtype = np.dtype(dtype)
    nout = len(OPS[name][1])
    spans = [h.ndim == len(shape) and h.shape[0] == shape[0] for h in host]
    row_pts = int(math.prod(shape[1:]))
    # device bytes per leading-axis row: sliced inputs (a broadcast row still costs its own size) + outputs
    row_bytes = (sum(int(math.prod(h.shape[1:])) for h, sp in zip(host, spans) if sp) + nout * row_pts) * cdtype.itemsize
    blocks = [b for b in leading_axis_bounds(shape[0], len(devs)) if b[1] > b[0]]
    plans = []
    for dev, (lo, hi) in zip(devs, blocks):
        # every device block of a slice is rounded up to 1 MiB (device._Allocation)
        nblocks = sum(spans) + nout
        pl = plan_slices(hi - lo, max(row_bytes, 1), stream_budget_bytes(dev), overhead=nblocks * _BLOCK_OVERHEAD)
        if pl is None:
            raise _ffi.EkmError(f"{name}: one leading-axis row needs {row_bytes} B on the device, two do not fit the "
                                f"streaming budget of {stream_budget_bytes(dev)} B on device {dev}")
        plans.append(pl)
    # Results of calls up to _PINNED_OUT_BYTES land in pinned host memory from a recycling pool (device.pinned_empty):
    # the downloads are then plain DMAs -- no page faults, no pin / unpin around every copy.  Beyond that (or when
    # pinned memory cannot be had) ordinary arrays, prefaulted slice by slice.
    outs, pinned_outs = None, False
    if _engine._PINNED_OUT and nout * int(math.prod(shape)) * out_dtype.itemsize <= _engine._PINNED_OUT_BYTES:
        from .device import pinned_empty

        outs = [pinned_empty(shape, out_dtype) for _ in range(nout)]
        pinned_outs = all(o is not None for o in outs)
    if not pinned_outs:
        outs = [np.empty(shape, out_dtype) for _ in range(nout)]
    errors = []
    slices, ready, max_rows = [], {}, []  # per device: its slices in order (and the longest); per slice: "its result pages exist" event
    for (lo, hi), (lanes, nslices) in zip(blocks, plans):
        mine = [(lo + a, lo + b) for a, b in leading_axis_bounds(hi - lo, nslices) if b > a]
        slices.append(mine)
        max_rows.append(max(b - a for a, b in mine))
        for sl in mine:
            ready[sl] = threading.Event()

    def toucher():
        # make the result pages of ordinary (non-pooled) result arrays exist slice by slice, in the order the transfers
        # will need them, all GPUs interleaved
        try:
            lib = _ffi.lib()
            for k in range(max(len(m) for m in slices)):
                for mine in slices:
                    if k < len(mine):
                        lo, hi = mine[k]
                        if not pinned_outs and outs[0][lo:hi].nbytes >= _PRETOUCH_BYTES // 8:
                            for o in outs:
                                lib.ekm_host_prefault(o[lo:hi].ctypes.data, o[lo:hi].nbytes, 4)
                        ready[mine[k]].set()
        finally:
            for ev in ready.values():
                ev.set()

    trace = [] if os.environ.get("EKM_TRACE_STREAM") else None
    import time as _time

    def uploader(dev, mine, depth, slots, handoff, most, d):
        # ONE host-to-device copy in flight per GPU: concurrent pageable uploads collapse (207 MB in 1 / 2 / 4 / 8
        # threads: 48 / 53 / 19 / 16 GB/s), while an upload and a download run together at full rate (PCIe duplex)
        try:
            set_device(dev)
            for k, (lo, hi) in enumerate(mine):
                t0 = _time.perf_counter()
                slots.acquire()  # at most `depth` slices resident on the device
                if errors:
                    break
                t1 = _time.perf_counter()
                set_stream(_lane_stream(dev, k % depth))
                # operands that span the leading axis get the matching slice; everything else is passed as
                # the caller gave it (a Python scalar must stay a weak scalar for the dtype promotion)
                part = [h[lo:hi] if sp else a for h, a, sp in zip(host, args, spans)]
                # slices differ by one row: every slice's device blocks are sized for the LONGEST slice, so a lane takes
                # back exactly the blocks it released (same bucket) and the footprint stays lanes x slice, live + cached
                ev = _uploaded_event(dev, k % depth)
                pend = _submit(name, part, ints, eps, dtype, host_out=[o[lo:hi] for o in outs], reserve_rows=(hi - lo, most),
                               uploaded=ev, pinned_out=pinned_outs and len(devs) == 1)
                # ONE slice's uploads in flight: from pinned inputs they are asynchronous, and the uploads of several lanes
                # queued at once share the link badly (P3 from pinned inputs: 60 GB/s against 77 with this wait).  The wait
                # is for the UPLOADS (an event recorded behind the last one, before the launch), not for the slice's kernel:
                # upload k+1 runs under kernel k (rounds 4-5 waited for the whole stream, which put a compute-heavy kernel --
                # the bisection wet-bulb, fp64 P5 -- in series with the uploads: ADVICE r5)
                _ffi.check(_ffi.lib().ekm_event_sync(dev, ev))
                handoff.put(((lo, hi), pend))
                if trace is not None:
                    trace.append(("up", k, t0, t1, _time.perf_counter()))
        except BaseException as exc:  # surfaced in the calling thread
            errors.append(exc)
        finally:
            handoff.put(None)

    def downloader(dev, slots, handoff, d):
        # ... and ONE device-to-host copy; a collected slice's device blocks go back to the block cache of its
        # stream and are taken again when the uploader comes round to that stream
        try:
            set_device(dev)
            while True:
                item = handoff.get()
                if item is None:
                    return
                sl, pend = item
                try:
                    if not errors:
                        t0 = _time.perf_counter()
                        ready[sl].wait()
                        t1 = _time.perf_counter()
                        set_stream(_lane_stream(dev, "download"))
                        _collect(pend, refile=pend.stream)
                        if trace is not None:
                            trace.append(("down", sl[0], t0, t1, _time.perf_counter()))
                finally:
                    slots.release()
        except BaseException as exc:
            errors.append(exc)
            while handoff.get() is not None:  # drain, so that the uploader is never left blocked
                slots.release()

    threads = [threading.Thread(target=toucher)]
    for d, (dev, mine, (depth, _n), most) in enumerate(zip(devs, slices, plans, max_rows)):
        slots, handoff = threading.Semaphore(depth), queue.Queue()
        threads.append(threading.Thread(target=uploader, args=(dev, mine, depth, slots, handoff, most, d)))
        threads.append(threading.Thread(target=downloader, args=(dev, slots, handoff, d)))
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    if errors:
        raise errors[0]
    if trace is not None:  # EKM_TRACE_STREAM=1: when each slice waited / moved (ms since the first event)
        z = min(e[2] for e in trace)
        for kind, k, t0, t1, t2 in sorted(trace, key=lambda e: e[2]):
            print(f"[stream] {kind:4s} slice {k:4d}: wait {1e3 * (t0 - z):7.2f} -> {1e3 * (t1 - z):7.2f}, work -> {1e3 * (t2 - z):7.2f} ms")
    return tuple(outs)
