#!/usr/bin/env python3
"""What the host link does in both directions at once, from two Python threads through the C ABI (the shape of the streamed
NumPy path, ekm_hip/_streamed.py): uploads on one stream, downloads on another, 8 / 26 / 78 MB per copy, pageable or
pinned source, every copy waited for or only the last.  On the GPU box of round 5: 54 + 47 GB/s together (91-96 GB/s) in
every combination -- so the 29 + 29 GB/s the pipeline saw when a slice's download was issued on the LANE stream (whose
first copy had been an upload) is the streams' binding to one DMA engine, not the link: profiles/r05_host_path_rate.txt.

    python tools/duplex_probe.py"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
import numpy as np
import ekm_hip
from ekm_hip import _ffi
lib = _ffi.lib()
CH = 26 << 20
N = 24  # chunks per direction
def run(src_pinned, chunk=CH, nstream=1, sync_each=True):
    src = [ekm_hip.pinned_empty(chunk, np.uint8) if src_pinned else np.ones(chunk, np.uint8) for _ in range(3)]
    dst = [ekm_hip.pinned_empty(chunk, np.uint8) for _ in range(3)]
    for s in src: s[:] = 1
    dA = [ekm_hip.DeviceArray.empty(chunk // 4, np.float32) for _ in range(3)]
    dB = [ekm_hip.DeviceArray.empty(chunk // 4, np.float32) for _ in range(3)]
    sa = [ekm_hip.stream_create() for _ in range(nstream)]; sb = [ekm_hip.stream_create() for _ in range(nstream)]
    res = {}
    def up():
        t0 = time.perf_counter()
        for k in range(N):
            st = sa[k % nstream]
            _ffi.check(lib.ekm_h2d(0, dA[k % 3].ptr, src[k % 3].ctypes.data, chunk, st))
            if sync_each: _ffi.check(lib.ekm_stream_sync(0, st))
        for st in sa: _ffi.check(lib.ekm_stream_sync(0, st))
        res["up"] = N * chunk / (time.perf_counter() - t0) / 1e9
    def down():
        t0 = time.perf_counter()
        for k in range(N):
            st = sb[k % nstream]
            _ffi.check(lib.ekm_d2h(0, dst[k % 3].ctypes.data, dB[k % 3].ptr, chunk, st))
            if sync_each: _ffi.check(lib.ekm_stream_sync(0, st))
        for st in sb: _ffi.check(lib.ekm_stream_sync(0, st))
        res["down"] = N * chunk / (time.perf_counter() - t0) / 1e9
    up(); down()
    alone = dict(res)
    ta, tb = threading.Thread(target=up), threading.Thread(target=down)
    t0 = time.perf_counter(); ta.start(); tb.start(); ta.join(); tb.join(); wall = time.perf_counter() - t0
    print(f"src {'pinned' if src_pinned else 'pageable'} chunk {chunk >> 20} MB sync_each={sync_each}: alone up {alone['up']:.1f} down {alone['down']:.1f} GB/s; together up {res['up']:.1f} down {res['down']:.1f}, sum over wall {2 * N * chunk / wall / 1e9:.1f} GB/s")
for pinned in (False, True):
    for chunk in (26 << 20, 78 << 20, 8 << 20):
        run(pinned, chunk)
    run(pinned, 26 << 20, sync_each=False)
