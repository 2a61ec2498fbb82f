#!/usr/bin/env python3
"""Instruction census of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).

    python tools/isa_census.py listing.s 'map_fields.*OpWetBulbFromQILi0ELi1E.*Li1E'

Counts static instructions by class; the per-point figures divide by the 4 points a lane's 16-B chunk
holds (the per-point body is fully unrolled four times in map_fields<.., float, 1>)."""
import collections
import re
import sys


def census(path, pattern):
    rx = re.compile(pattern)
    lines = open(path).read().splitlines()
    start = next(i for i, ln in enumerate(lines) if ln.endswith(":") or ": ;" in ln if rx.search(ln.split(":")[0]) and ln.startswith("_Z"))
    body = []
    for ln in lines[start + 1:]:
        if ln.startswith("\ts_endpgm"):
            break
        body.append(ln)
    ops = [ln.split()[0] for ln in body if ln.startswith("\t") and not ln.startswith("\t.") and not ln.startswith("\t;")]
    c = collections.Counter(ops)
    trans = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32")
    cls = collections.Counter()
    for op, n in c.items():
        if op.rsplit("_e", 1)[0] in trans:
            cls["trans:" + op] += n
        elif op.startswith("v_cndmask"):
            cls["v_cndmask"] += n
        elif op.startswith("v_cmp"):
            cls["v_cmp*"] += n
        elif op.startswith("v_mov") or op.startswith("v_accvgpr"):
            cls["v_mov/accvgpr"] += n
        elif op.startswith("v_fma") or op.startswith("v_mul") or op.startswith("v_add") or op.startswith("v_sub") or op.startswith("v_mac") or op.startswith("v_fmac") or op.startswith("v_pk_"):
            cls["v_arith(fma/mul/add/sub)"] += n
        elif op.startswith("v_"):
            cls["v_other:" + op] += n
        elif op.startswith("s_"):
            cls["scalar"] += n
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("ds_") or op.startswith("scratch_"):
            cls["mem:" + op] += n
        else:
            cls["other:" + op] += n
    return c, cls


if __name__ == "__main__":
    c, cls = census(sys.argv[1], sys.argv[2])
    valu = sum(n for k, n in cls.items() if k.startswith(("trans", "v_")))
    tr = sum(n for k, n in cls.items() if k.startswith("trans"))
    for k, n in sorted(cls.items(), key=lambda kv: -kv[1]):
        print(f"{n:6d}  {k}")
    print(f"static VALU {valu} (of which transcendental {tr}); issue units = VALU + 3*trans = {valu + 3 * tr}; per point (/4): "
          f"{valu / 4:.1f} VALU, {tr / 4:.1f} trans, {(valu + 3 * tr) / 4:.1f} units")
