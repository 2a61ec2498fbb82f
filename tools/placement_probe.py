#!/usr/bin/env python3
"""Does the time of one kernel depend on WHERE its fields land in device memory?  One process, several sets of
buffers allocated one after the other (all kept), the same kernels timed on each set, interleaved.

    python tools/placement_probe.py [--sets 4] [--workloads p3,full,geopotential]
"""
import argparse
import ctypes as C
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
from ekm_hip import _ffi  # noqa: E402

INNER, NLEV = 1800 * 3600, 137
N = INNER * NLEV


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--pad-mb", type=int, default=0, help="extra MiB allocated between sets (shifts later sets)")
    ap.add_argument("--arena", action="store_true", help="carve the nine fields of a set out of ONE allocation")
    ap.add_argument("--slack-mb", type=int, default=16, help="room after every field for the skews (k-th array: k*skew bytes)")
    ap.add_argument("--skews", default="0", help="comma list of byte skews: the k-th array of a set is used at base + k*skew")
    a = ap.parse_args()
    lib = _ffi.lib()
    chk = _ffi.check
    dev = 0

    def dmalloc(nbytes):
        p = C.c_void_p()
        chk(lib.ekm_malloc(dev, nbytes, C.byref(p)))
        return p.value

    from ekm_hip.vertical import hybrid_level_parameters

    A, B = (x.astype(np.float32) for x in hybrid_level_parameters(137))
    sp = (101325.0 * (1.0 - 0.35 * np.random.default_rng(1).random(INNER) ** 3)).astype(np.float32)
    sets = []
    for s in range(a.sets):
        if a.arena:
            per = (4 * N + (a.slack_mb << 20) + (2 << 20) - 1) // (2 << 20) * (2 << 20)
            base = dmalloc(9 * per)
            bufs = [base + k * per for k in range(9)]
        else:
            bufs = [dmalloc(4 * N + (a.slack_mb << 20)) for _ in range(9)]
        small = []
        for arr in (A, B, sp, sp):
            ptr = dmalloc(arr.nbytes)
            chk(lib.ekm_h2d(dev, ptr, arr.ctypes.data, arr.nbytes, None))
            small.append(ptr)
        chk(lib.ekm_synth_fill_f32(dev, None, bufs[0], bufs[1], bufs[2], 0, N, INNER, NLEV, 20260313))
        chk(lib.ekm_sync(dev))
        if a.pad_mb:
            dmalloc(a.pad_mb << 20)
        sets.append((bufs, small))
        print(f"set {s}: first buffer at 0x{bufs[0]:x}", flush=True)
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    chk(lib.ekm_event_create(dev, C.byref(ev0)))
    chk(lib.ekm_event_create(dev, C.byref(ev1)))
    F = _ffi.Operand

    skews = [int(x) for x in a.skews.split(",")]

    def launch(w, bufs, small, skew=0):
        bufs = [b + k * skew for k, b in enumerate(bufs)]
        t, q, p = bufs[:3]
        outs = bufs[3:]
        ops = [C.byref(F(x, 0, 0, 0, 0)) for x in (t, q, p)]
        if w == "p3":
            chk(lib.ekm_pipeline_svp_td_rh_f32(dev, None, *ops, outs[0], outs[1], outs[2], N))
        elif w == "full":
            chk(lib.ekm_pipeline_full_f32(dev, None, *ops, *outs[:6], N))
        elif w == "theta":
            chk(lib.ekm_potential_temperature_f32(dev, None, ops[0], ops[2], outs[0], N))
        else:
            chk(lib.ekm_geopotential_on_hybrid_levels_f32(dev, None, small[0], small[1], small[2], small[3], t, q, INNER,
                                                          NLEV, 1, 0.6931471805599453, 1, outs[0]))

    times = {}
    for rnd in range(a.rounds + 1):
        for w in ("theta", "p3", "full", "geopotential"):
            for s, (bufs, small) in enumerate(sets):
                for sk in skews:
                    if sk:  # the inputs must be valid at the skewed addresses too
                        if rnd == 0 and w == "theta":
                            b = [x + k * sk for k, x in enumerate(bufs)]
                            chk(lib.ekm_synth_fill_f32(dev, None, b[0], b[1], b[2], 0, N, INNER, NLEV, 20260313))
                    launch(w, bufs, small, sk)
                    chk(lib.ekm_event_record(dev, ev0, None))
                    for _ in range(5):
                        launch(w, bufs, small, sk)
                    chk(lib.ekm_event_record(dev, ev1, None))
                    chk(lib.ekm_sync(dev))
                    ms = C.c_float()
                    chk(lib.ekm_event_elapsed_ms(dev, ev0, ev1, C.byref(ms)))
                    if rnd:
                        times.setdefault((w, sk, s), []).append(ms.value / 5)
    for w in ("theta", "p3", "full", "geopotential"):
        for sk in skews:
            row = [statistics.median(times[(w, sk, s)]) for s in range(len(sets))]
            print(f"{w:13s} skew {sk:8d}: " + "  ".join(f"{x:.3f}" for x in row) + f"   worst {max(row):.3f}  best {min(row):.3f} ms")


if __name__ == "__main__":
    main()
