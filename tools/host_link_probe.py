#!/usr/bin/env python3
"""What the host <-> GPU link of this box can do, by the routes the NumPy path could take:
pinned and pageable copies in each direction alone and in both at once, the threaded host-to-host copy that staging
through pinned buffers needs, and the price of pinning caller memory in place (hipHostRegister)."""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
from ekm_hip import _ffi  # noqa: E402

lib = _ffi.lib()
chk = _ffi.check
MB = 1 << 20
N = 512 * MB


def dmalloc(n):
    p = C.c_void_p()
    chk(lib.ekm_malloc(0, n, C.byref(p)))
    return p.value


def pinned(n):
    p = C.c_void_p()
    chk(lib.ekm_host_alloc(n, C.byref(p)))
    return p.value


def stream():
    s = C.c_void_p()
    chk(lib.ekm_stream_create(0, C.byref(s)))
    return s.value


def rate(fn, nbytes, reps=4):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return nbytes / best / 1e9


def both(f1, f2):
    def run():
        th = threading.Thread(target=f2)
        th.start()
        f1()
        th.join()
    return run


def main():
    d1, d2 = dmalloc(N), dmalloc(N)
    h1, h2 = pinned(N), pinned(N)
    a1, a2 = np.ones(N // 4, np.float32), np.ones(N // 4, np.float32)  # pageable, touched
    s1, s2 = stream(), stream()

    def h2d(src, s):
        return lambda: (chk(lib.ekm_h2d(0, d1, src, N, s)), chk(lib.ekm_stream_sync(0, s)))

    def d2h(dst, s):
        return lambda: (chk(lib.ekm_d2h(0, dst, d2, N, s)), chk(lib.ekm_stream_sync(0, s)))

    print(f"{N // MB} MiB per copy, best of 4")
    print(f"pinned   H2D alone {rate(h2d(h1, s1), N):6.1f} GB/s   D2H alone {rate(d2h(h2, s2), N):6.1f} GB/s   "
          f"both at once {rate(both(h2d(h1, s1), d2h(h2, s2)), 2 * N):6.1f} GB/s (sum)")
    p1, p2 = a1.ctypes.data, a2.ctypes.data
    print(f"pageable H2D alone {rate(h2d(p1, s1), N):6.1f} GB/s   D2H alone {rate(d2h(p2, s2), N):6.1f} GB/s   "
          f"both at once {rate(both(h2d(p1, s1), d2h(p2, s2)), 2 * N):6.1f} GB/s (sum)")
    for nt in (1, 2, 4, 8):
        r1 = rate(lambda: chk(lib.ekm_host_memcpy(h1, p1, N, nt)), N)
        r2 = rate(lambda: chk(lib.ekm_host_memcpy(p2, h2, N, nt)), N)
        r3 = rate(both(lambda: chk(lib.ekm_host_memcpy(h1, p1, N, nt)), lambda: chk(lib.ekm_host_memcpy(p2, h2, N, nt))), 2 * N)
        print(f"host memcpy {nt} thread(s): pageable->pinned {r1:6.1f} GB/s   pinned->pageable {r2:6.1f} GB/s   both at once {r3:6.1f} GB/s (sum)")
    # staged pipeline in chunks: memcpy to pinned chunk k+1 while chunk k is in flight, both directions at once
    for chunk_mb, nt in ((16, 4), (32, 4), (64, 4), (32, 8), (32, 2)):
        ch = chunk_mb * MB
        nch = N // ch
        ring = 3
        evs_up = [C.c_void_p() for _ in range(ring)]
        evs_dn = [C.c_void_p() for _ in range(ring)]
        for e in evs_up + evs_dn:
            chk(lib.ekm_event_create(0, C.byref(e)))

        def up():
            for k in range(nch):
                slot = k % ring
                if k >= ring:
                    chk(lib.ekm_event_sync(0, evs_up[slot]))
                chk(lib.ekm_host_memcpy(h1 + slot * ch, p1 + k * ch, ch, nt))
                chk(lib.ekm_h2d(0, d1 + k * ch, h1 + slot * ch, ch, s1))
                chk(lib.ekm_event_record(0, evs_up[slot], s1))
            chk(lib.ekm_stream_sync(0, s1))

        def dn():
            for k in range(nch + ring - 1):
                if k < nch:
                    slot = k % ring
                    chk(lib.ekm_d2h(0, h2 + slot * ch, d2 + k * ch, ch, s2))
                    chk(lib.ekm_event_record(0, evs_dn[slot], s2))
                j = k - (ring - 1)
                if j >= 0:
                    chk(lib.ekm_event_sync(0, evs_dn[j % ring]))
                    chk(lib.ekm_host_memcpy(p2 + j * ch, h2 + (j % ring) * ch, ch, nt))

        print(f"staged through {ring} pinned chunks of {chunk_mb} MiB, {nt} copy threads per direction: upload alone {rate(up, N):6.1f} GB/s   "
              f"download alone {rate(dn, N):6.1f} GB/s   both at once {rate(both(up, dn), 2 * N):6.1f} GB/s (sum)")
    # the library's own staged transfer (ekm_copy_staged): one 512-MiB segment, and the same bytes as 20 segments
    def staged(to_dev, nseg, nt, s):
        seg = N // nseg
        if to_dev:
            dst = (C.c_void_p * nseg)(*[d1 + k * seg for k in range(nseg)])
            src = (C.c_void_p * nseg)(*[p1 + k * seg for k in range(nseg)])
        else:
            dst = (C.c_void_p * nseg)(*[p2 + k * seg for k in range(nseg)])
            src = (C.c_void_p * nseg)(*[d2 + k * seg for k in range(nseg)])
        nb = (C.c_size_t * nseg)(*[seg] * nseg)
        return lambda: (chk(lib.ekm_copy_staged(0, int(to_dev), nseg, dst, src, nb, s, nt)), chk(lib.ekm_stream_sync(0, s)))

    for nseg, nt in ((1, 4), (20, 4), (1, 2), (1, 8)):
        print(f"ekm_copy_staged {nseg:2d} segment(s), {nt} threads: upload alone {rate(staged(True, nseg, nt, s1), N):6.1f} GB/s   "
              f"download alone {rate(staged(False, nseg, nt, s2), N):6.1f} GB/s   both at once "
              f"{rate(both(staged(True, nseg, nt, s1), staged(False, nseg, nt, s2)), 2 * N):6.1f} GB/s (sum)", flush=True)
    t0 = time.perf_counter()
    chk(lib.ekm_host_register(p1, N))
    t1 = time.perf_counter()
    print(f"hipHostRegister of {N // MB} MiB of touched pageable memory: {(t1 - t0) * 1e3:.1f} ms = {N / (t1 - t0) / 1e9:.1f} GB/s; "
          f"registered H2D {rate(h2d(p1, s1), N):6.1f} GB/s")
    t0 = time.perf_counter()
    chk(lib.ekm_host_unregister(p1))
    print(f"hipHostUnregister: {(time.perf_counter() - t0) * 1e3:.1f} ms")
    print("usable cores:", len(os.sched_getaffinity(0)))


if __name__ == "__main__":
    main()
