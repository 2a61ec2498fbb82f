#!/usr/bin/env python3
"""Does the NumPy-in / NumPy-out rate depend on which NUMA node the process runs on?  (Round 6: bench.py's `end_to_end` came out
at 27 ms in some processes and 50 ms in others on the same box, every transfer twice as slow.)  Prints the topology the
process can see, then runs the 8-level six-output call in CHILD processes pinned (before they touch the GPU) to the usable
CPUs of each NUMA node in turn, and once unpinned.

    python tools/numa_probe.py"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpulist(txt):
    out = set()
    for part in txt.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def child(cpus):
    if cpus:
        os.sched_setaffinity(0, cpus)
    sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]
    import time

    import numpy as np

    import ekm_hip
    from ekm_hip import thermo
    from oracle import synthetic

    t, q, p, _ = synthetic.make_fields(8, 1800 * 3600, dtype=np.float32, seed=3)
    best = {}
    for name, fn, nio in (("pipeline_svp_td_rh", thermo.pipeline_svp_td_rh, 6), ("pipeline_full", thermo.pipeline_full, 9)):
        b = 1e9
        for _ in range(6):
            res = None
            t0 = time.perf_counter()
            res = fn(t, q, p)
            b = min(b, time.perf_counter() - t0)
        res = None
        best[name] = (b * 1e3, nio * t.nbytes / b / 1e9)
    print(f"   cpus {sorted(os.sched_getaffinity(0))[:4]}...({len(os.sched_getaffinity(0))}): " +
          ", ".join(f"{k} {v[0]:.1f} ms = {v[1]:.1f} GB/s" for k, v in best.items()), flush=True)


def main():
    if len(sys.argv) > 1:
        return child(cpulist(sys.argv[1]) if sys.argv[1] != "-" else None)
    allowed = os.sched_getaffinity(0)
    print(f"usable CPUs: {len(allowed)} of {os.cpu_count()}: {sorted(allowed)}")
    nodes = {}
    for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
        nodes[int(d.rsplit("node", 1)[1])] = cpulist(open(d + "/cpulist").read())
    for n, c in nodes.items():
        print(f"NUMA node {n}: {len(c)} CPUs, usable here: {sorted(c & allowed)}")
    for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        try:
            print(d, "numa_node", open(d + "/numa_node").read().strip(), "local_cpulist", open(d + "/local_cpulist").read().strip())
        except OSError as e:
            print(d, e)
    print("unpinned:")
    subprocess.call([sys.executable, __file__, "-"])
    for n, c in nodes.items():
        use = c & allowed
        if use:
            print(f"pinned to the usable CPUs of node {n}:")
            subprocess.call([sys.executable, __file__, ",".join(map(str, sorted(use)))])


if __name__ == "__main__":
    main()
