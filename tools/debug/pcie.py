import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
import numpy as np
from ekm_hip import _ffi
lib = _ffi.lib(); chk = _ffi.check
N = 1 << 30
d0, d1 = C.c_void_p(), C.c_void_p()
chk(lib.ekm_malloc(0, N, C.byref(d0))); chk(lib.ekm_malloc(0, N, C.byref(d1)))
h0, h1 = C.c_void_p(), C.c_void_p()
chk(lib.ekm_host_alloc(N, C.byref(h0))); chk(lib.ekm_host_alloc(N, C.byref(h1)))
pg = np.ones(N // 4, np.float32); pg2 = np.empty(N // 4, np.float32)
s0, s1 = C.c_void_p(), C.c_void_p()
chk(lib.ekm_stream_create(0, C.byref(s0))); chk(lib.ekm_stream_create(0, C.byref(s1)))
def t(f, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); chk(lib.ekm_sync(0)); best = min(best, time.perf_counter() - t0)
    return N / best / 1e9
print("pinned   H2D %.1f GB/s" % t(lambda: chk(lib.ekm_h2d(0, d0, h0, N, s0))))
print("pinned   D2H %.1f GB/s" % t(lambda: chk(lib.ekm_d2h(0, h1, d1, N, s1))))
def both():
    chk(lib.ekm_h2d(0, d0, h0, N, s0)); chk(lib.ekm_d2h(0, h1, d1, N, s1))
print("pinned   H2D+D2H concurrently: %.1f GB/s each way" % t(both))
print("pageable H2D %.1f GB/s" % t(lambda: chk(lib.ekm_h2d(0, d0, pg.ctypes.data, N, None))))
print("pageable D2H %.1f GB/s" % t(lambda: chk(lib.ekm_d2h(0, pg2.ctypes.data, d1, N, None))))
hv = np.ctypeslib.as_array(C.cast(h0, C.POINTER(C.c_float)), shape=(N // 4,))
t0 = time.perf_counter(); np.copyto(hv, pg); dt = time.perf_counter() - t0
print("host memcpy pageable->pinned (1 thread) %.1f GB/s" % (N / dt / 1e9))
