"""the host restatement of the kernel math under AddressSanitizer + UndefinedBehaviorSanitizer.

`make -C earthkit-meteo_amd twin-asan` builds csrc/host_twin.cpp (the same thermo_math.hpp / ops.hpp templates the
gfx950 kernels instantiate) with -fsanitize=address,undefined -fno-sanitize-recover.  This test runs every golden
case of tests/test_hosttwin_golden.py and tests/test_wind_oracle.py through that library in a child interpreter
with libasan preloaded: any out-of-bounds access, use of an uninitialised bool/enum, signed overflow, invalid
shift or float-to-int conversion out of range in the math aborts the child.  (GPU sanitizers are not available
on the pool; this is the CPU-side check the survey asked for.)"""
import os
import subprocess
import sys

import pytest

import _hosttwin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    out = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


@pytest.mark.skipif(not os.path.exists(_hosttwin.ASAN_PATH), reason="ASan host twin not built (make twin-asan)")
def test_goldens_through_the_sanitized_host_twin():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan:
        pytest.skip("libasan.so not found next to gcc")
    env = dict(os.environ, EKM_HOSTTWIN_LIB=_hosttwin.ASAN_PATH, LD_PRELOAD=":".join(x for x in (asan, ubsan) if x),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="1", PYTHONMALLOC="malloc")
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_hosttwin_golden.py"), os.path.join(ROOT, "tests", "test_wind_oracle.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = (r.stdout[-3000:] + r.stderr[-3000:])
    assert r.returncode == 0, tail
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, tail
    assert " passed" in r.stdout, tail
