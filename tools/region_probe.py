#!/usr/bin/env python3
"""Is streaming speed a property of WHERE in device memory a buffer lies?  Allocate most of the HBM in chunks, run the
same one-read-one-write kernel (celsius_to_kelvin, in place) on every chunk, and print each chunk's rate in allocation
order.  (tools/placement_probe.py --arena showed that the time of a multi-stream kernel does not depend on the distances
between its streams -- 256 B ... 1 MiB of skew change nothing -- but on the allocation the streams live in.)"""
import argparse
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
from ekm_hip import _ffi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunk-mb", type=int, default=2048)
    ap.add_argument("--fraction", type=float, default=0.85)
    ap.add_argument("--reps", type=int, default=6)
    a = ap.parse_args()
    lib, chk = _ffi.lib(), _ffi.check
    free, total = C.c_size_t(), C.c_size_t()
    chk(lib.ekm_mem_info(0, C.byref(free), C.byref(total)))
    chunk = a.chunk_mb << 20
    n = int(free.value * a.fraction) // chunk
    bufs = []
    for _ in range(n):
        p = C.c_void_p()
        chk(lib.ekm_malloc(0, chunk, C.byref(p)))
        bufs.append(p.value)
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    chk(lib.ekm_event_create(0, C.byref(ev0)))
    chk(lib.ekm_event_create(0, C.byref(ev1)))
    npts = chunk // 4
    F = _ffi.Operand
    for b in bufs:
        chk(lib.ekm_memset(0, b, 0x41, chunk, None))  # ~12.1 as float: plain data, not zeros
    chk(lib.ekm_sync(0))
    rates = []
    for k, b in enumerate(bufs):
        op = F(b, 0, 0, 0, 0)
        ts = []
        for r in range(a.reps + 1):
            chk(lib.ekm_event_record(0, ev0, None))
            chk(lib.ekm_kelvin_to_celsius_f32(0, None, C.byref(op), b, npts) if r % 2 else lib.ekm_celsius_to_kelvin_f32(0, None, C.byref(op), b, npts))
            chk(lib.ekm_event_record(0, ev1, None))
            chk(lib.ekm_sync(0))
            ms = C.c_float()
            chk(lib.ekm_event_elapsed_ms(0, ev0, ev1, C.byref(ms)))
            if r:
                ts.append(ms.value)
        gbs = 2 * chunk / statistics.median(ts) / 1e6
        rates.append(gbs)
        print(f"chunk {k:3d} at 0x{b:x}: {statistics.median(ts):7.4f} ms  {gbs:7.1f} GB/s", flush=True)
    rs = sorted(rates)
    print(f"{n} chunks of {a.chunk_mb} MiB: min {rs[0]:.0f}  p10 {rs[len(rs) // 10]:.0f}  median {statistics.median(rs):.0f}  "
          f"p90 {rs[-1 - len(rs) // 10]:.0f}  max {rs[-1]:.0f} GB/s")


if __name__ == "__main__":
    main()
