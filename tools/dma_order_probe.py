#!/usr/bin/env python3
"""Round 6: the streamed NumPy-in / NumPy-out call runs at 68 GB/s in some processes and 37 in others on the same box.
Suspect: which DMA engine HIP binds to which stream depends on the FIRST copies of the process.  Each sequence below runs in
a fresh child process, then times the 8-level six-output call (best of 5).

    python tools/dma_order_probe.py"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEQS = ["none", "big_vram_first", "h2d207_then_big_vram", "big_vram_used_first", "h2d207_then_big_vram_used"]


def child(seq):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
    import numpy as np

    import ekm_hip
    from ekm_hip import thermo
    from oracle import synthetic

    t, q, p, _ = synthetic.make_fields(8, 1800 * 3600, dtype=np.float32, seed=3)
    small = np.ones(1024, np.float32)
    keep = []
    if seq.startswith("h2d207"):
        for _ in range(3):
            ekm_hip.to_device(np.ones(8 * 1800 * 3600, np.float32)).free()
    if "big_vram" in seq:
        keep = [ekm_hip.DeviceArray.empty((137 * 1800 * 3600,), np.float32) for _ in range(9)]
        if "used" in seq:  # touch them: kernels over all nine fields, a download of a window
            for _ in range(5):
                o = thermo.pipeline_full(keep[0], keep[1], keep[2])
                ekm_hip.synchronize()
                for x in o:
                    x.free()
            keep[3].flat_slice(0, 256).to_host()
    if seq in ("d2h_first", "d2h_then_h2d"):
        d = ekm_hip.DeviceArray.empty((1024,), np.float32)
        d.to_host()
    if seq == "d2h_first_big":
        d = ekm_hip.DeviceArray.empty((1800 * 3600,), np.float32)
        for _ in range(8):
            d.to_host()
    if seq == "kernel_then_d2h":
        d = ekm_hip.DeviceArray.empty((1800 * 3600,), np.float32)
        o = thermo.celsius_to_kelvin(d)
        o.to_host()
    if seq in ("h2d_first", "d2h_then_h2d"):
        ekm_hip.to_device(small)
    ekm_hip.synchronize()
    best = 1e9
    for _ in range(6):
        res = None
        t0 = time.perf_counter()
        res = thermo.pipeline_full(t, q, p)
        best = min(best, time.perf_counter() - t0)
    print(f"{seq:16s} pipeline_full 8 levels: {best * 1e3:6.1f} ms = {9 * t.nbytes / best / 1e9:5.1f} GB/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for rep in range(2):
            for s in SEQS:
                subprocess.call([sys.executable, __file__, s])
