#!/usr/bin/env python3
"""P3 (5 streams) on three arenas of one process, six launches each -- to be run under rocprofv3 --pmc: which hardware
counter separates a slow arena from a fast one?  (tools/placement_probe.py --arena: the time of a multi-stream kernel is
a property of the allocation its streams live in, not of the distances between them.)

    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d OUT -- python3 tools/arena_pmc.py
    python3 tools/arena_pmc.py --summarize OUT        # per arena: mean of every counter over its launches
"""
import collections
import csv
import ctypes as C
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
INNER, NLEV = 1800 * 3600, 137
N = INNER * NLEV
SETS, LAUNCHES = 3, 6


def summarize(out):
    rows = []
    for path in glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True):
        rows += [r for r in csv.DictReader(open(path)) if "OpPipelineSvpTdRh" in r["Kernel_Name"]]
    by_disp = collections.defaultdict(dict)
    for r in rows:
        by_disp[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(by_disp)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for k, d in enumerate(ids):
        for c, v in by_disp[d].items():
            agg[(k // LAUNCHES) % SETS][c].append(v)
    for s in sorted(agg):
        print(f"arena {s}: " + "  ".join(f"{c} {sum(v) / len(v):.4g}" for c, v in sorted(agg[s].items())))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
        return summarize(sys.argv[2])
    from ekm_hip import _ffi

    lib, chk = _ffi.lib(), _ffi.check
    F = _ffi.Operand
    arenas = []
    per = (4 * N + (16 << 20) + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    for s in range(SETS):
        p = C.c_void_p()
        chk(lib.ekm_malloc(0, 6 * per, C.byref(p)))
        bufs = [p.value + k * per for k in range(6)]
        chk(lib.ekm_synth_fill_f32(0, None, bufs[0], bufs[1], bufs[2], 0, N, INNER, NLEV, 20260313))
        arenas.append(bufs)
    chk(lib.ekm_sync(0))
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    chk(lib.ekm_event_create(0, C.byref(ev0)))
    chk(lib.ekm_event_create(0, C.byref(ev1)))
    for rnd in range(2):
        for s, b in enumerate(arenas):
            ops = [C.byref(F(x, 0, 0, 0, 0)) for x in b[:3]]
            chk(lib.ekm_event_record(0, ev0, None))
            for _ in range(LAUNCHES):
                chk(lib.ekm_pipeline_svp_td_rh_f32(0, None, *ops, b[3], b[4], b[5], N))
            chk(lib.ekm_event_record(0, ev1, None))
            chk(lib.ekm_sync(0))
            ms = C.c_float()
            chk(lib.ekm_event_elapsed_ms(0, ev0, ev1, C.byref(ms)))
            print(f"round {rnd} arena {s} at 0x{b[0]:x}: {ms.value / LAUNCHES:.3f} ms per launch", flush=True)


if __name__ == "__main__":
    main()
