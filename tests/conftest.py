import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "earthkit-meteo_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ek():
    """The product package on a real GPU.  A machine without any AMD GPU device node skips the `-m gpu`
    tests; a machine WITH one (the GPU box) must load the HIP library and see the device -- anything else
    is a failure, never a silent skip or fallback."""
    if not os.path.exists("/dev/kfd"):
        pytest.skip("no AMD GPU on this machine (/dev/kfd missing)")
    import ekm_hip
    import ekm_hip.vertical  # noqa: F401

    assert ekm_hip.device_count() >= 1, "a GPU device node exists but HIP sees no device"
    return ekm_hip


def pytest_terminal_summary(terminalreporter):
    """Print how much of every parity relaxation was actually used (tests/_compare.py)."""
    try:
        from _compare import CENSUS, LEDGER
    except Exception:
        return
    if CENSUS:
        terminalreporter.write_sep("-", "whole-field parity censuses (tests/test_gpu_census.py)")
        for line in CENSUS:
            terminalreporter.write_line(line)
    if not LEDGER:
        return
    agg = {}
    for what, kind, used, allowed, n in LEDGER:
        a = agg.setdefault(kind, dict(tests=0, used=0, n=0, worst=(0.0, "", 0, 0.0)))
        a["tests"] += 1
        a["used"] += used
        a["n"] += n
        frac = used / allowed if allowed else 0.0
        if frac >= a["worst"][0]:
            a["worst"] = (frac, what, used, allowed)
    tr = terminalreporter
    tr.write_sep("-", "parity budgets used (tests/_compare.py)")
    for kind, a in sorted(agg.items()):
        w = a["worst"]
        tr.write_line(f"{kind}: {a['used']} of {a['n']} points in {a['tests']} checks; closest to its limit: "
                      f"{w[2]} of {w[3]:.0f} allowed ({w[1][:70]})")
    if os.environ.get("EKM_LEDGER_DETAIL"):
        for what, kind, used, allowed, n in sorted(LEDGER, key=lambda x: -x[2] / max(x[4], 1))[:40]:
            tr.write_line(f"  {used:6d} / {n:8d} = {used / max(n, 1):.4f} (limit {allowed:.0f})  {kind[:28]:28s} {what[:90]}")
