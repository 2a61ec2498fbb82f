"""Does P5's run-to-run spread come from where hipMalloc places the nine 3.55 GB arrays?"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
import numpy as np
from ekm_hip import _ffi
lib = _ffi.lib()
chk = _ffi.check
nlev, inner = 137, 1800 * 3600
n = nlev * inner
F = _ffi.Operand
ev0, ev1 = C.c_void_p(), C.c_void_p()
chk(lib.ekm_event_create(0, C.byref(ev0))); chk(lib.ekm_event_create(0, C.byref(ev1)))

def malloc(nbytes):
    p = C.c_void_p(); chk(lib.ekm_malloc(0, nbytes, C.byref(p))); return p.value

def time_p5(ptrs, reps=10):
    t, q, p = ptrs[:3]; outs = ptrs[3:]
    chk(lib.ekm_synth_fill_f32(0, None, t, q, p, 0, n, inner, nlev, 20260313))
    ops = [C.byref(F(x, 0, 0, 0, 0)) for x in (t, q, p)]
    args = [0, None] + ops + outs + [n]
    for _ in range(3): chk(lib.ekm_pipeline_full_f32(*args))
    chk(lib.ekm_event_record(0, ev0, None))
    for _ in range(reps): chk(lib.ekm_pipeline_full_f32(*args))
    chk(lib.ekm_event_record(0, ev1, None)); chk(lib.ekm_sync(0))
    ms = C.c_float(); chk(lib.ekm_event_elapsed_ms(0, ev0, ev1, C.byref(ms)))
    return ms.value / reps

print("separate hipMalloc per array, re-allocated each round:")
for r in range(6):
    ptrs = [malloc(4 * n) for _ in range(9)]
    ms = time_p5(ptrs)
    print("  round", r, "%.3f ms" % ms, "base addrs mod 2^21:", [hex(p % (1 << 21)) for p in ptrs[:3]], "gaps GiB", [round((ptrs[i+1]-ptrs[i])/2**30,3) for i in range(3)])
    for p in ptrs: chk(lib.ekm_free(0, p))
print("one slab, arrays at k*(4n+pad):")
for pad in (0, 4096, 65536, (1 << 20) + 4096, (2 << 20), (2 << 20) + 8192, 33 << 20):
    stride = 4 * n + pad
    stride = (stride + 255) // 256 * 256
    base = malloc(9 * stride)
    ptrs = [base + k * stride for k in range(9)]
    res = [time_p5(ptrs) for _ in range(2)]
    print("  pad", pad, ["%.3f" % x for x in res])
    chk(lib.ekm_free(0, base))
