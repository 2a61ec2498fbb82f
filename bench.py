#!/usr/bin/env python3
"""Benchmark of the fused thermo pipeline on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload full|p3|wetbulb|...]

A "step" is one pass of the hot path over one 0.1-degree global field
(3600 x 1800 x 137 fp32 grid points per variable) resident in HBM: one launch of
the fused kernel, inputs generated on the device beforehand (synthetic, seeded).
N > 1 is one process per GPU (the driver launches it through
torch.distributed.run); grid points are independent, so ranks share nothing and
the only communication is the timing barrier / max-reduce, done over gloo on the
CPU so that PyTorch never touches the GPUs the HIP library is using.

Rank 0 prints ONE JSON line: the contract fields plus `roofline` (algorithmic
bytes / HIP-event kernel time against the 8 TB/s HBM peak), `cpu_baseline` (the
NumPy oracle timed on this box's host cores on a bounded sample, N = 1 only) and
`parity` (GPU output vs the oracle: the whole 8-level slab SURVEY.md 8d names, 51.84 M points of the timed arrays,
through a pool of host processes, with the wet-bulb's regime-flip count; plus 256-point windows of 32 levels),
`end_to_end` (H2D + kernel + D2H of that slab through the NumPy-in / NumPy-out path: PCIe-inclusive, never `value`) and,
in `roofline`, the spread of the kernel time over three independently allocated buffer sets (`frac_min/median/max`).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "earthkit-meteo_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

NLEV, NLAT, NLON = 137, 1800, 3600
INNER = NLAT * NLON
N3 = NLEV * INNER
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0  # same guide: float4 copy, measured
# VALU issue roof.  Peak by the guide: a wave64 VALU instruction issues over 2 clocks on a SIMD-32, 4 SIMDs x 256 CUs,
# 2.4 GHz: 0.5 x 1024 x 2.4e9 x 64 lanes.  Achievable: what tools/microbench/valu_rates.hip sustains (v_fma_f32: 0.381
# wave-instr/clk/SIMD, profiles/r03_valu_microbench.txt).  One issue unit = one plain fp32 VALU wave-instruction; the
# weights of the other classes are their issue time relative to it: fp32 transcendentals issue at a quarter of the
# plain rate (4), fp64 arithmetic at half (2; spec 78.6 vs 157.3 TFLOP/s), v_rcp_f64 3.3 x an fp64 fma (6.6; measured).
VALU_PEAK_UNITS = 0.5 * 1024 * 2.4e9 * 64
VALU_ACHIEVABLE_UNITS = 5.99e13
VALU_WEIGHTS = {"SQ_INSTS_VALU_TRANS_F32": 4.0, "SQ_INSTS_VALU_ADD_F64": 2.0, "SQ_INSTS_VALU_MUL_F64": 2.0,
                "SQ_INSTS_VALU_FMA_F64": 2.0, "SQ_INSTS_VALU_TRANS_F64": 6.6}

# workload -> (entry point, inputs, outputs, algorithmic bytes/point fp32 with p a full field, oracle call)
WORKLOADS = {
    "full": ("pipeline_full", 3, 6, 36, "fused theta+es+rh+td+theta_e+tw(ifs,newton)"),
    "p3": ("pipeline_svp_td_rh", 3, 3, 24, "fused es+td+rh"),
    "wetbulb": ("wet_bulb_temperature_from_specific_humidity", 3, 1, 16, "wet-bulb (ifs, newton)"),
    "wetbulb_bisect": ("wet_bulb_temperature_from_specific_humidity", 3, 1, 16, "wet-bulb (ifs, bisect)"),
    "wetbulb_bisect_bolton35": ("wet_bulb_temperature_from_specific_humidity", 3, 1, 16, "wet-bulb (bolton35, bisect)"),
    "wetbulb_bisect_bolton39": ("wet_bulb_temperature_from_specific_humidity", 3, 1, 16, "wet-bulb (bolton39, bisect)"),
    "rh": ("relative_humidity_from_specific_humidity", 3, 1, 16, "rh from q"),
    "theta": ("potential_temperature", 2, 1, 12, "potential temperature"),
    "svp": ("saturation_vapour_pressure", 1, 1, 8, "saturation vapour pressure (mixed)"),
    "ept": ("ept_from_specific_humidity", 3, 1, 16, "theta_e (ifs)"),
    # the producer of p; reads sp once (1/137 of 4 B/pt), writes p_full
    "hybrid_levels": ("pressure_on_hybrid_levels", 0, 1, 4, "p_full on hybrid levels from sp (A, B tables)"),
    # t, q, sp, zs -> geopotential on model levels in one fused column scan
    "geopotential": ("geopotential_on_hybrid_levels", 2, 1, 12, "geopotential on hybrid levels (fused alpha/delta + scan)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="full", choices=sorted(WORKLOADS))
    ap.add_argument("--pmode", default="field", choices=["field", "level", "hybrid"],
                    help="pressure as a full field, as the 137-level vector staged in LDS, or formed in the kernel "
                         "from surface pressure and the IFS L137 A/B tables (hybrid model levels)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="strong: ONE global field split across the GPUs (BASELINE.json config 5; the default for "
                         "N > 1); weak: one full global field per GPU (the only reading at N = 1, and its label there)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--levels", type=int, default=NLEV, help="levels per field (137 = the named config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-slab-levels", type=int, default=8,
                    help="levels (1800 x 3600 points each) of the slab the single-core CPU baseline runs on")
    ap.add_argument("--allow-shared-device", action="store_true",
                    help="more ranks than visible GPUs: still report a value (ranks share devices, so it is NOT a scaling "
                         "measurement); without it such a run reports value null and oversubscribed true")
    ap.add_argument("--dry-run", action="store_true",
                    help="exercise rendezvous/sharding/reporting without touching a GPU (CI on CPU); value is null")
    ap.add_argument("--traffic", default="measure", choices=["measure", "file", "none"],
                    help="roofline.traffic: 'measure' = two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of a short run of this "
                         "same command as child processes after the timed region (N = 1 only; falls back to 'file' if the "
                         "profiler is unavailable); 'file' = the committed profiles/traffic_latest.json; 'none' = null")
    ap.add_argument("--valu", default="measure", choices=["measure", "file", "none"],
                    help="roofline VALU side (executed issue units per point): 'measure' = a third rocprofv3 child pass of this "
                         "command (--pmc SQ_INSTS_VALU ..., N = 1 only; falls back to 'file' with the reason stated); 'file' = the "
                         "committed profiles/valu_latest.json; 'none' = HBM roof only")
    ap.add_argument("--no-stream-ceiling", dest="stream_ceiling", action="store_false",
                    help="skip roofline.stream_ceiling (the no-arithmetic streaming reference timed on the launch's own arrays)")
    ap.add_argument("--sustain", type=float, default=2.0,
                    help="after the timed region, keep launching the same step back to back for at least this many seconds and "
                         "report it as `sustained` (0 = skip); K timed steps stay what --steps asked for")
    ap.add_argument("--tiles", type=int, default=0)
    ap.add_argument("--unroll", type=int, default=0)
    ap.add_argument("--parity-slab-levels", type=int, default=8,
                    help="SURVEY.md 8d: levels (1800 x 3600 points each) of the timed arrays checked point by point against the "
                         "oracle on a pool of host processes (N = 1; 0 = the 256-point windows only)")
    ap.add_argument("--no-end-to-end", dest="end_to_end", action="store_false",
                    help="skip `end_to_end` (H2D + kernel + D2H of the 8-level slab through the NumPy path, N = 1)")
    ap.add_argument("--buffer-sets", type=int, default=3,
                    help="independently allocated copies of the input/output fields the kernel is timed on (the first is the timed "
                         "region's; the others add roofline.frac_min/median/max: where a 3.5-GB field lands in HBM moves a "
                         "streaming kernel by 5-10 %%); N = 1, 1 = skip")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------
def oracle_call(workload, t, q, p):
    from oracle import thermo_oracle as orc

    with np.errstate(all="ignore"):
        if workload == "full":
            return orc.pipeline_full(t, q, p)
        if workload == "p3":
            return orc.pipeline_svp_td_rh(t, q, p)
        if workload == "wetbulb":
            return (orc.wet_bulb_temperature_from_specific_humidity(t, q, p, "ifs", "newton"),)
        if workload == "wetbulb_bisect":
            return (orc.wet_bulb_temperature_from_specific_humidity(t, q, p, "ifs", "bisect"),)
        if workload.startswith("wetbulb_bisect_"):
            return (orc.wet_bulb_temperature_from_specific_humidity(t, q, p, workload.rsplit("_", 1)[1], "bisect"),)
        if workload == "rh":
            return (orc.relative_humidity_from_specific_humidity(t, q, p),)
        if workload == "theta":
            return (orc.potential_temperature(t, p),)
        if workload == "svp":
            return (orc.saturation_vapour_pressure(t),)
        if workload == "ept":
            return (orc.ept_from_specific_humidity(t, q, p),)
    raise KeyError(workload)


CPU_BUDGET_S = 30.0  # the CPU baseline stops repeating once a leg has used this much time


def _usable_cores():
    """Host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _cpu_worker(job):
    workload, nlev, npts, seed, dtype, reps = job
    from oracle import synthetic

    t, q, p, _ = synthetic.make_fields(nlev, npts, dtype=np.dtype(dtype), seed=seed)
    t, q, p = t.ravel(), q.ravel(), p.ravel()
    best, done, start = float("inf"), 0, time.perf_counter()
    while done < reps and (done == 0 or time.perf_counter() - start < CPU_BUDGET_S):  # bounded: stop early on a slow host
        t0 = time.perf_counter()
        oracle_call(workload, t, q, p)
        best = min(best, time.perf_counter() - t0)
        done += 1
    return t.size, best, done


def cpu_baseline(workload, slab_levels, dtype, reps=3):
    """The NumPy oracle (same operator sequence as the reference; kind "port") on the host cores, on the
    sample BASELINE.md section 3 names: an 8-level slab of the benchmark atmosphere (8 x 1800 x 3600 =
    51.84 M points).  Two figures, best of `reps` each: one process on the whole slab ("NumPy, 1 core":
    NumPy ufuncs are single-threaded) and one level (6.48 M points) per worker on every usable core at
    once.  Must run before this process initialises HIP (fork)."""
    import multiprocessing as mp

    cores = _usable_cores()
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        # the single-core run has the machine to itself first, then all workers run concurrently
        n1, t1, reps1 = pool.apply(_cpu_worker, ((workload, slab_levels, INNER, 1, dtype, reps),))
        res = pool.map(_cpu_worker, [(workload, 1, INNER, 100 + i, dtype, reps) for i in range(cores)], chunksize=1)
    wall = time.perf_counter() - t0
    pts = sum(r[0] for r in res)
    slowest = max(r[1] for r in res)
    return {
        "value": pts / slowest, "unit": "grid-points/s", "cores": cores, "kind": "port",
        "value_1core": n1 / t1, "os_cpu_count": os.cpu_count(),
        "repetitions": {"1core": reps1, "all_cores": min(r[2] for r in res)},
        "sample": f"NumPy oracle oracle/thermo_oracle.py ({workload}) on the synthetic atmosphere: 1 core = one process "
                  f"on a {slab_levels}-level slab ({n1} points); all cores = {cores} concurrent workers x one level "
                  f"({INNER} points) each; best of up to {reps} (a leg stops repeating after {CPU_BUDGET_S:.0f} s); os.cpu_count() = {os.cpu_count()}, usable = {cores}; "
                  f"wall {wall:.1f} s incl. input generation",
    }


# ---------------------------------------------------------------------------------------------
class Dist:
    """Timing barrier / max-reduce across ranks (gloo on the CPU; no data-path collective exists)."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self.td = None
        if self.world > 1:
            import datetime

            import torch.distributed as td

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")  # (never reached under a launcher: torch.distributed.run always sets it)
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # single node: never depend on hostname resolution
            td.init_process_group("gloo", rank=self.rank, world_size=self.world,
                                  timeout=datetime.timedelta(seconds=600))
            self.td = td

    def barrier(self):
        if self.td:
            self.td.barrier()

    def reduce(self, value, op="max"):
        if not self.td:
            return value
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64)
        self.td.all_reduce(t, op=getattr(self.td.ReduceOp, op.upper()))
        return float(t[0])

    def gather(self, values):
        """Every rank's list of floats, as a list of lists (rank order)."""
        if not self.td:
            return [list(map(float, values))]
        import torch

        mine = torch.tensor(list(map(float, values)), dtype=torch.float64)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self.td.all_gather(out, mine)
        return [o.tolist() for o in out]

    def gather_object(self, obj):
        """Every rank's (picklable) object on every rank, rank order."""
        if not self.td:
            return [obj]
        out = [None] * self.world
        self.td.all_gather_object(out, obj)
        return out

    def close(self):
        if self.td:
            self.td.destroy_process_group()


COLUMN_WORKLOADS = ("hybrid_levels", "geopotential")


def plan_shard(workload, pmode, scaling, rank, world, nlev=NLEV, inner=INNER):
    """Which part of the global [nlev, inner] field rank `rank` of `world` owns.

    weak:   every rank owns one whole field (n_total = world fields).
    strong: ONE field, no halo, no exchange (BASELINE.json config 5):
      * p a full field: the flat grid-point range cut by `shard_bounds` (64-B aligned starts);
      * p a level vector / hybrid levels: cut on level boundaries (137 levels -> 18/17/17/... per GPU at 8),
        so a shard's pressure operand is the contiguous sub-vector (sub-table) of its levels;
      * the column workloads (hybrid_levels, geopotential: every column needs all its levels): cut along
        the horizontal axis, each rank holding [nlev, its columns].
    Returns first (flat index of the shard's first point in the global field; column shards: first column),
    n_local, n_total, lev0/lev1 (level range), col0/col1 (column range)."""
    from ekm_hip._streamed import leading_axis_bounds
    from ekm_hip.device import shard_bounds

    n_field = nlev * inner
    d = dict(lev0=0, lev1=nlev, col0=0, col1=inner, first=0, n_local=n_field, n_total=n_field * world, cut="none")
    if scaling == "weak" or world == 1:
        if scaling != "weak":
            d["n_total"] = n_field
        return d
    d["n_total"] = n_field
    if workload in COLUMN_WORKLOADS:
        lo, hi = shard_bounds(inner, world)[rank]
        d.update(col0=lo, col1=hi, first=lo, n_local=nlev * (hi - lo), cut="columns")
    elif pmode in ("level", "hybrid"):
        lo, hi = leading_axis_bounds(nlev, world)[rank]
        d.update(lev0=lo, lev1=hi, first=lo * inner, n_local=(hi - lo) * inner, cut="levels")
    else:
        lo, hi = shard_bounds(n_field, world)[rank]
        d.update(first=lo, n_local=hi - lo, lev0=lo // inner, lev1=-(-hi // inner), cut="grid points")
    return d


def build_inputs(args, sh, dev, nlev, np_dtype, seed):
    """The synthetic inputs of one shard (`sh` from plan_shard) on device `dev`, generated there: t, q, p (None unless p is a
    full field), the level-pressure vector and, for hybrid levels, the shard's A/B tables + surface pressure.  The generator
    is counter-based on the GLOBAL point index, so a grid-point or level shard holds exactly the values the whole field has
    at its points (tests/test_gpu_streaming.py regenerates the whole field with world = 1 and compares; a column shard
    of the column workloads is its own [level, column] field)."""
    from ekm_hip import _ffi
    from ekm_hip.device import DeviceArray

    lib = _ffi.lib()
    first, n_local = sh["first"], sh["n_local"]
    shape = (n_local,)
    lev0, lev1, col0, col1 = sh["lev0"], sh["lev1"], sh["col0"], sh["col1"]
    nlev_loc, ncol = lev1 - lev0, col1 - col0   # this shard's levels / columns
    t = DeviceArray.empty(shape, np_dtype, dev)
    q = DeviceArray.empty(shape, np_dtype, dev)
    p = DeviceArray.empty(shape, np_dtype, dev) if args.pmode == "field" else None
    plev = DeviceArray.empty((nlev,), np_dtype, dev)
    _ffi.check(getattr(lib, f"ekm_synth_levels_{args.dtype}")(dev, None, plev.ptr, nlev))
    hyb = None
    if args.pmode == "hybrid" or args.workload == "hybrid_levels":
        # IFS L137 half-level tables (ekm_hip.vertical.hybrid_level_parameters, shipped inside the package); a level
        # shard takes the half levels lev0 .. lev1 of the table, a column shard its columns of sp
        assert nlev <= 137, "hybrid mode: at most the 137 IFS levels"
        from ekm_hip.vertical import hybrid_level_parameters

        A137, B137 = hybrid_level_parameters(137, model="ifs")
        A = A137[137 - nlev:][lev0:lev1 + 1]
        B = B137[137 - nlev:][lev0:lev1 + 1]
        rng = np.random.default_rng(20260313 if args.scaling == "strong" else seed)
        sp_host = (101325.0 * (1.0 - 0.35 * rng.random(INNER) ** 3)).astype(np_dtype)  # mostly near sea level, some orography
        sp_host = np.ascontiguousarray(sp_host[col0:col1])
        hyb = dict(A=DeviceArray.from_host(A.astype(np_dtype), dev), B=DeviceArray.from_host(B.astype(np_dtype), dev),
                   sp=DeviceArray.from_host(sp_host, dev), Ah=A, Bh=B, sph=sp_host,
                   top0=int(A[0] == 0.0 and B[0] == 0.0))  # the table's first half level is the model top (p = 0): alpha = ln 2 there
    if args.pmode == "hybrid":
        # t, q drawn around the hybrid-level pressure (materialised once, then dropped)
        ptmp = DeviceArray.empty(shape, np_dtype, dev)
        _ffi.check(getattr(lib, f"ekm_pressure_on_hybrid_levels_{args.dtype}")(
            dev, None, hyb["A"].ptr, hyb["B"].ptr, hyb["sp"].ptr, ncol, nlev_loc, None, None, hyb["top0"],
            float(np.log(2)), ptmp.ptr, None, None, None))
        _ffi.check(getattr(lib, f"ekm_synth_fill_given_p_{args.dtype}")(dev, None, t.ptr, q.ptr, ptmp.ptr, first,
                                                                         n_local, seed))
        _ffi.check(lib.ekm_sync(dev))
        ptmp.free()
    else:
        fill = getattr(lib, f"ekm_synth_fill_{args.dtype}")
        _ffi.check(fill(dev, None, t.ptr, q.ptr, p.ptr if p else None, first, n_local, INNER, nlev, seed))
    return t, q, p, plev, hyb


def free_port():
    """A TCP port nobody listens on right now (two benchmark runs on one node must not share a rendezvous port)."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # convenience: relaunch under torch.distributed.run as a child (nothing has touched the GPU yet)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT") or str(free_port()),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    dist = Dist()
    entry, nin, nout, bpp, desc = WORKLOADS[args.workload]
    itemsize = 4 if args.dtype == "f32" else 8
    np_dtype = np.float32 if args.dtype == "f32" else np.float64
    bpp = bpp * itemsize // 4
    if args.pmode in ("level", "hybrid") and nin >= 2:
        bpp -= itemsize  # p is not read per point (hybrid: sp is 1/137 of a field and served from cache)
    nlev = args.levels
    n_field = nlev * INNER
    if args.scaling == "auto":
        args.scaling = "strong" if dist.world > 1 else "weak"
    if args.workload == "geopotential":
        args.pmode = "hybrid"
    sh = plan_shard(args.workload, args.pmode, args.scaling, dist.rank, dist.world, nlev)
    first, n_local, n_total = sh["first"], sh["n_local"], sh["n_total"]

    cpu = None
    if (dist.rank == 0 and dist.world == 1 and not args.no_cpu_baseline and not args.dry_run
            and args.workload not in COLUMN_WORKLOADS):
        cpu = cpu_baseline(args.workload, args.cpu_slab_levels, np_dtype)  # before HIP is initialised (fork)

    kernel_ms, parity = None, None
    if args.dry_run:
        step = lambda: None  # noqa: E731
        sync = lambda: None  # noqa: E731
    else:
        import ekm_hip
        from ekm_hip import _ffi
        from ekm_hip.device import DeviceArray

        lib = _ffi.lib()
        ndev = ekm_hip.device_count()
        dev = dist.local_rank % ndev  # more ranks than devices share them: flagged below, value null unless allowed
        ekm_hip.set_device(dev)
        if args.tiles or args.unroll:
            _ffi.check(lib.ekm_set_tuning(args.tiles, args.unroll))
        shape = (n_local,)
        lev0, lev1, col0, col1 = sh["lev0"], sh["lev1"], sh["col0"], sh["col1"]
        nlev_loc, ncol = lev1 - lev0, col1 - col0   # this shard's levels / columns
        seed = 20260313 + (dist.rank if args.scaling == "weak" else 0)
        link_at_start = None
        if dist.world == 1 and args.end_to_end and not os.environ.get("EKM_BENCH_NO_LINK_PROBE"):
            # what a plain 207-MB host-to-device copy gets in THIS process before anything else has run (diagnostic for
            # `end_to_end`: on the same box some processes get half the link's rate for every copy, from their first one on)
            probe = np.ones(8 * INNER, np.float32)
            rates = []
            for _ in range(3):
                t0 = time.perf_counter()
                dprobe = DeviceArray.from_host(probe, dev)
                _ffi.check(lib.ekm_sync(dev))
                rates.append(probe.nbytes / (time.perf_counter() - t0) / 1e9)
                dprobe.free()
            link_at_start = round(max(rates), 1)
            del probe
        t, q, p, plev, hyb = build_inputs(args, sh, dev, nlev, np_dtype, seed)
        outs = [DeviceArray.empty(shape, np_dtype, dev) for _ in range(nout)]

        fn = getattr(lib, f"ekm_{entry}_{args.dtype}")
        F = _ffi.Operand
        op_t, op_q = F(t.ptr, _ffi.FIELD, 0, 0, 0), F(q.ptr, _ffi.FIELD, 0, 0, 0)
        if p is not None:
            op_p = F(p.ptr, _ffi.FIELD, 0, 0, 0)
        elif args.pmode == "hybrid":
            nz = np.flatnonzero(hyb["Bh"] != 0.0)  # leading pure pressure levels of this shard's table (ekm_operand.nflat)
            nflat = int(max(0, (nz[0] if nz.size else hyb["Bh"].size) - 1))
            op_p = F(hyb["sp"].ptr, _ffi.HYBRID_FULL, nflat, nlev_loc, ncol, hyb["A"].ptr, hyb["B"].ptr)
        else:  # level vector: a shard starts on a level boundary (plan_shard) and passes the sub-vector of its levels
            assert first % INNER == 0, "level mode needs level-aligned shards"
            op_p = F(plev.ptr + lev0 * itemsize, _ffi.LEVEL_MAJOR, 0, nlev_loc, INNER)
        operands = {"pipeline_full": (op_t, op_q, op_p), "pipeline_svp_td_rh": (op_t, op_q, op_p),
                    "wet_bulb_temperature_from_specific_humidity": (op_t, op_q, op_p),
                    "relative_humidity_from_specific_humidity": (op_t, op_q, op_p),
                    "ept_from_specific_humidity": (op_t, op_q, op_p),
                    "potential_temperature": (op_t, op_p), "saturation_vapour_pressure": (op_t,),
                    "pressure_on_hybrid_levels": (), "geopotential_on_hybrid_levels": ()}[entry]
        ints = {"wetbulb": (0, 1), "wetbulb_bisect": (0, 0), "wetbulb_bisect_bolton35": (1, 0), "wetbulb_bisect_bolton39": (2, 0), "svp": (0,), "ept": (0,)}.get(args.workload, ())
        cargs = [dev, None] + [C.byref(o) for o in operands] + list(ints) + [o.ptr for o in outs] + [n_local]
        if args.workload == "geopotential":
            zs_host = np.maximum(0.0, (101325.0 - hyb["sph"].astype(np.float64)) / 1.2).astype(np_dtype)  # g*z ~ dp / rho
            hyb["zs"], hyb["zsh"] = DeviceArray.from_host(zs_host, dev), zs_host
            cargs = [dev, None, hyb["A"].ptr, hyb["B"].ptr, hyb["sp"].ptr, hyb["zs"].ptr, t.ptr, q.ptr, ncol, nlev,
                     hyb["top0"], float(np.log(2)), 1, outs[0].ptr]
        if args.workload == "hybrid_levels":
            cargs = [dev, None, hyb["A"].ptr, hyb["B"].ptr, hyb["sp"].ptr, ncol, nlev, None, None, hyb["top0"],
                     float(np.log(2)), outs[0].ptr, None, None, None]

        def step():
            _ffi.check(fn(*cargs))

        def sync():
            _ffi.check(lib.ekm_sync(dev))

        evs = [C.c_void_p() for _ in range(args.steps + 1)]  # one event before each launch + one at the end
        for e in evs:
            _ffi.check(lib.ekm_event_create(dev, C.byref(e)))

    for _ in range(args.warmup):
        step()
    sync()
    dist.barrier()
    sync()
    # The ranks leave a gloo barrier hundreds of microseconds apart -- of the order of a strong-scaling step at N = 8 (0.65 ms),
    # in a timed region of ~13 ms.  CLOCK_MONOTONIC is one clock for every process of the node: inside the bracket the ranks
    # agree on an instant 10 ms ahead (the latest rank's clock, all-reduced) and spin until then, so that they START within
    # microseconds of one another; a rank that arrives late starts at once (barrier_skew_ms shows it).
    if dist.world > 1:
        t_go = dist.reduce(time.perf_counter(), "max") + 10e-3
        while time.perf_counter() < t_go:
            pass
    t0 = time.perf_counter()
    for k in range(args.steps):
        if not args.dry_run:
            _ffi.check(lib.ekm_event_record(dev, evs[k], None))  # same (default) stream the kernels are launched on
        step()
    if not args.dry_run:
        _ffi.check(lib.ekm_event_record(dev, evs[-1], None))
    sync()
    t1 = time.perf_counter()
    elapsed = t1 - t0                   # this rank's K steps, from the common start to its own device sync
    dist.barrier()                      # the closing bracket: nobody reports before everybody has finished
    sync()
    elapsed = dist.reduce(elapsed, "max")  # the job took as long as its slowest rank
    # how much of the timed region is the barrier's exit skew (at N = 8 a strong-scaling step is ~0.7 ms and the whole
    # region ~13 ms): the spread of the ranks' own start stamps, and of their end stamps
    stamps = dist.gather([t0, t1])
    barrier_skew_ms = (max(v[0] for v in stamps) - min(v[0] for v in stamps)) * 1e3
    end_skew_ms = (max(v[1] for v in stamps) - min(v[1] for v in stamps)) * 1e3

    ndev_seen, dev_used, my_ms = -1, -1, float("nan")
    sustained = None
    if not args.dry_run:
        ms = C.c_float()
        _ffi.check(lib.ekm_event_elapsed_ms(dev, evs[0], evs[-1], C.byref(ms)))
        my_ms = ms.value / args.steps
        kernel_ms = dist.reduce(my_ms, "max")  # average launch duration, slowest rank
        if args.sustain > 0:
            # the same launch, back to back, for >= --sustain seconds (outside the K timed steps): what the chip holds once
            # clocks and temperature have settled, and long enough for an external activity sampler to see a busy GPU
            batch = max(1, int(250.0 / max(my_ms, 1e-3)))  # ~0.25 s of launches between two host syncs
            e0, e1 = C.c_void_p(), C.c_void_p()
            _ffi.check(lib.ekm_event_create(dev, C.byref(e0)))
            _ffi.check(lib.ekm_event_create(dev, C.byref(e1)))
            dist.barrier()
            ts, n_launch = time.perf_counter(), 0
            _ffi.check(lib.ekm_event_record(dev, e0, None))
            while n_launch < args.steps or time.perf_counter() - ts < args.sustain:
                for _ in range(batch):
                    step()
                n_launch += batch
                _ffi.check(lib.ekm_event_record(dev, e1, None))
                sync()
            wall = time.perf_counter() - ts
            _ffi.check(lib.ekm_event_elapsed_ms(dev, e0, e1, C.byref(ms)))
            # the slowest rank's figures (every rank ran for its own >= --sustain seconds)
            sus_ms = dist.reduce(ms.value / n_launch, "max")
            wall_per = dist.reduce(wall / n_launch, "max")
            sustained = {"launches": n_launch, "seconds": round(wall, 3), "ms_per_step": round(wall_per * 1e3, 4),
                         "kernel_ms": round(sus_ms, 4)}
        per_launch = []
        for k in range(args.steps):
            _ffi.check(lib.ekm_event_elapsed_ms(dev, evs[k], evs[k + 1], C.byref(ms)))
            per_launch.append(ms.value)
        ndev_seen, dev_used = ndev, dev
    # parity: rank 0 judges a sample of its own shard AND, at N > 1, a sample of the LAST rank's shard (whose first point is
    # not point 0 of the field), which travels through the gather together with the sample's place in the global field
    smp = None if args.dry_run else sample_shard(args, t, q, p, plev, outs, sh, nlev, np_dtype, hyb)
    slab, e2e, sets = None, None, None
    if not args.dry_run and dist.world == 1:
        # the additional legs must never cost the run its line: a failure is reported in its place (and makes parity.ok false
        # where it is the slab census that failed)
        def leg(what, f, *a):
            try:
                return f(*a)
            except Exception as exc:  # noqa: BLE001
                import traceback

                traceback.print_exc(file=sys.stderr)
                return {"failed": f"{what}: {type(exc).__name__}: {exc}"}

        if args.parity_slab_levels > 0:
            slab = leg("slab_parity", slab_parity, args, t, q, p, plev, outs, nlev, np_dtype)
            if "failed" in slab:
                slab["ok"] = False
                slab["points"] = 0
        # (end_to_end BEFORE the buffer sets: returning tens of GB of device memory to the driver -- what the sets do when they are
        # freed, beyond what the block cache keeps -- is followed by a period in which every host<->device copy runs at half
        # rate, in this process and in the next one on the device: profiles/r06_host_path_rate.txt)
        if args.end_to_end:
            e2e = leg("end_to_end", end_to_end, args, t, q, p, plev, nlev, np_dtype)
            if e2e is not None:
                e2e["h2d_gbs_at_process_start"] = link_at_start
        if args.buffer_sets > 1 and args.workload not in COLUMN_WORKLOADS:
            sets = leg("time_buffer_sets", time_buffer_sets, args, sh, dev, nlev, np_dtype, seed, nout, entry, ints, my_ms)
            if "failed" in sets:
                print("bench.py: " + sets["failed"], file=sys.stderr)
                sets = None
    # The streaming reference of THIS launch on THESE buffers (the sample above is on the host by now: the outputs are
    # overwritten): the same input fields read and the same output fields written by a kernel that computes nothing
    # (ekm_stream_mix: the map kernels' launch shape, one add per stream).  The same kernel is 5-10 % faster or slower from
    # process to process (where the 3.5-GB fields land in physical memory); this ceiling shares the placement.
    ceiling_ms = None
    if not args.dry_run and args.stream_ceiling:
        with_p = {"pipeline_full": [t, q, p], "pipeline_svp_td_rh": [t, q, p], "wet_bulb_temperature_from_specific_humidity": [t, q, p],
                  "relative_humidity_from_specific_humidity": [t, q, p], "ept_from_specific_humidity": [t, q, p],
                  "potential_temperature": [t, p]}
        if entry in with_p:  # the pressure is a stream only when it is a field
            fields_in = with_p[entry] if args.pmode == "field" else with_p[entry][:-1]
        else:
            fields_in = {"saturation_vapour_pressure": [t], "pressure_on_hybrid_levels": [], "geopotential_on_hybrid_levels": [t, q]}[entry]
        nbytes = (n_local * itemsize) // 16 * 16
        ins_arr = (C.c_void_p * max(1, len(fields_in)))(*[x.ptr for x in fields_in])
        outs_arr = (C.c_void_p * len(outs))(*[o.ptr for o in outs])
        if len(outs) in (1, 2, 3, 6):
            mix = lambda: _ffi.check(lib.ekm_stream_mix(dev, None, ins_arr, len(fields_in), outs_arr, len(outs), nbytes))  # noqa: E731
            e0, e1 = C.c_void_p(), C.c_void_p()
            _ffi.check(lib.ekm_event_create(dev, C.byref(e0)))
            _ffi.check(lib.ekm_event_create(dev, C.byref(e1)))
            for _ in range(3):
                mix()
            _ffi.check(lib.ekm_event_record(dev, e0, None))
            for _ in range(10):
                mix()
            _ffi.check(lib.ekm_event_record(dev, e1, None))
            sync()
            _ffi.check(lib.ekm_event_elapsed_ms(dev, e0, e1, C.byref(ms)))
            ceiling_ms = dist.reduce(ms.value / 10, "max")
            ceiling_streams = (len(fields_in), len(outs))
    last_smp = dist.gather_object(smp if dist.rank == dist.world - 1 else None)
    parity_last, shard_window = None, None
    if not args.dry_run and dist.rank == 0:
        parity = judge_sample(args, smp)
        if slab is not None:
            # the slab is THE parity figure of the line; the windows (32 levels of the whole column) stay beside it
            parity = dict(slab, windows=parity, ok=bool(slab["ok"] and parity["ok"])) if "points" in slab else dict(parity, slab=slab)
        if dist.world > 1:
            ls = last_smp[-1]
            parity_last = dict(judge_sample(args, ls), rank=dist.world - 1)
            parity["ok"] = bool(parity["ok"] and parity_last["ok"])
            if ls["kind"] == "windows":  # lets a reader regenerate the field and check the shard's offset (tests do)
                shard_window = {"rank": dist.world - 1, "global_index": ls["global_index"][0], "level": ls["levels"][0],
                                "t": [float(x) for x in ls["t"][:8]], "q": [float(x) for x in ls["q"][:8]],
                                "out0": [float(x) for x in ls["outs"][0][:8]]}
            else:
                shard_window = {"rank": dist.world - 1, "global_col0": ls["global_col0"], "t": [float(x) for x in ls["t"][0, :8]],
                                "out": [float(x) for x in ls["out"][0, :8]]}
    # what every rank did: [kernel ms per launch, points it owns, hipGetDeviceCount() it saw, device it used]
    per_rank = [dict(rank=r, kernel_ms=None if v[0] != v[0] else round(v[0], 4), points=int(v[1]),
                     hip_device_count=int(v[2]), device=int(v[3]))
                for r, v in enumerate(dist.gather([my_ms, n_local, ndev_seen, dev_used]))]

    if dist.rank == 0:
        value = None if args.dry_run else n_total * args.steps / elapsed
        devices = sorted({r["device"] for r in per_rank if r["device"] >= 0})
        ndev_min = min((r["hip_device_count"] for r in per_rank if r["hip_device_count"] >= 0), default=-1)
        oversub = not args.dry_run and len(devices) < dist.world
        if oversub and not args.allow_shared_device:
            value = None  # ranks shared a GPU: whatever this is, it is not the N-GPU rate
        roof = None
        traffic, traffic_source = None, "not collected (--traffic none)"
        if kernel_ms and args.traffic == "measure" and dist.world == 1:
            traffic, traffic_source = measure_traffic(args, entry)
        if kernel_ms and traffic is None and args.traffic != "none":
            why = "" if args.traffic == "file" else f" ({traffic_source})"
            traffic = traffic_from_profiles(args, n_local)
            traffic_source = ("profiles/traffic_latest.json: FETCH_SIZE x 2 (gfx950) + WRITE_SIZE from separate rocprofv3 --pmc passes "
                              "of this command (tools/profile_gpu.sh), committed; not re-measured in this run" + why)
        if kernel_ms:
            achieved = bpp * n_local / (kernel_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                    "kernel": entry, "bytes_per_point": bpp, "points_per_launch": n_local,
                    "kernel_ms": round(kernel_ms, 4),
                    "kernel_ms_median": round(float(np.median(per_launch)), 4),
                    "kernel_ms_min": round(float(np.min(per_launch)), 4),
                    "hbm_achievable_gbs": HBM_ACHIEVABLE_GBS}
            if sets:
                # the same kernel on independently allocated copies of its fields, one process: `frac` above stays the timed
                # region's (set 0); the spread is the placement lottery DESIGN.md section 3 describes, now in the record
                ms_all = [kernel_ms] + sets["kernel_ms"]
                fr = [bpp * n_local / (m * 1e-3) / 1e9 / HBM_PEAK_GBS for m in ms_all]
                roof.update(buffer_sets=len(ms_all), kernel_ms_sets=[round(m, 4) for m in ms_all], frac_sets=[round(f, 4) for f in fr],
                            frac_min=round(min(fr), 4), frac_median=round(float(np.median(fr)), 4), frac_max=round(max(fr), 4),
                            buffer_sets_what=sets["what"])
            valu, why_not = None, "not collected (--valu none)"
            if args.valu == "measure" and dist.world == 1 and args.workload not in COLUMN_WORKLOADS:
                valu, why_not = measure_valu(args, n_local)
            elif args.valu == "measure":
                why_not = "counter passes run at N = 1 only" if dist.world > 1 else "no VALU side for the column workloads"
            if valu is None and args.valu != "none":
                valu = valu_from_profiles(args, "" if args.valu == "file" else why_not)
            if valu:
                # The VALU side, quoted the same way as the HBM side: executed issue units per point (SQ counters) x points /
                # kernel time against the PART's peak -- MI355X_MICROARCH.md: one plain fp32 VALU wave-instruction per 2 clocks
                # per SIMD = 0.5 x 1024 SIMDs x 2.4 GHz x 64 lanes = 7.86e13 lane-ops/s -- with what the issue-rate
                # microbenchmark sustains (5.99e13) beside it as "achievable", like 8.0 vs 6.3 TB/s.  The roof with the larger
                # fraction is the one that binds.
                units = valu["units_per_point"] * n_local / (kernel_ms * 1e-3)
                vfrac = units / VALU_PEAK_UNITS
                roof.update(valu_issue_units_per_point=valu["units_per_point"], valu_achieved_units_per_s=float(f"{units:.4g}"),
                            valu_peak_units_per_s=VALU_PEAK_UNITS, valu_achievable_units_per_s=VALU_ACHIEVABLE_UNITS,
                            valu_frac=round(vfrac, 4), valu_frac_of_achievable=round(units / VALU_ACHIEVABLE_UNITS, 4),
                            valu_source=valu["source"], valu_counters=valu.get("counters"))
                if vfrac > roof["frac"]:
                    roof.update(bound="valu", achieved=float(f"{units:.4g}"), peak=VALU_PEAK_UNITS, unit="issue-units/s",
                                frac=round(vfrac, 4), hbm_achieved_gbs=round(achieved, 1),
                                hbm_frac=round(achieved / HBM_PEAK_GBS, 4))
                    if roof.get("kernel_ms_sets"):  # the spread over the buffer sets in the units of the roof that binds
                        vf = [valu["units_per_point"] * n_local / (m * 1e-3) / VALU_PEAK_UNITS for m in roof["kernel_ms_sets"]]
                        roof.update(hbm_frac_sets=roof["frac_sets"], frac_sets=[round(f, 4) for f in vf], frac_min=round(min(vf), 4),
                                    frac_median=round(float(np.median(vf)), 4), frac_max=round(max(vf), 4))
            else:
                roof["valu_source"] = why_not
            if ceiling_ms:
                # bytes the reference moves: its streams x the field size (the kernel's algorithmic bytes may be a little less
                # or more: a per-level pressure costs nothing, hybrid levels re-read sp from L2)
                c_bytes = sum(ceiling_streams) * itemsize * n_local
                roof["stream_ceiling"] = {
                    "what": "ekm_stream_mix on the arrays of this launch: the same %d input and %d output fields moved with the "
                            "map kernels' launch shape and no arithmetic, 10 launches after the timed region" % ceiling_streams,
                    "kernel_ms": round(ceiling_ms, 4), "gbs": round(c_bytes / (ceiling_ms * 1e-3) / 1e9, 1),
                    "frac": round(c_bytes / (ceiling_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "kernel_ms_over_ceiling_ms": round(kernel_ms / ceiling_ms, 4)}
            if sustained:
                s_ach = bpp * n_local / (sustained["kernel_ms"] * 1e-3) / 1e9
                sustained["frac"] = round(s_ach / HBM_PEAK_GBS, 4) if roof["bound"] == "hbm" else round(
                    roof["valu_issue_units_per_point"] * n_local / (sustained["kernel_ms"] * 1e-3) / VALU_PEAK_UNITS, 4)
                sustained["hbm_frac"] = round(s_ach / HBM_PEAK_GBS, 4)
                sustained["value"] = None if value is None else n_total / (sustained["ms_per_step"] * 1e-3)
        line = {
            "metric": "grid-points/sec for fused thermo pipeline; achieved HBM GB/s vs peak",
            "value": value, "unit": "grid-points/s", "n_gpus": dist.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": None if args.dry_run else round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{desc} on {nlev}x{NLAT}x{NLON} {args.dtype} "
                                   f"({'one field per GPU' if args.scaling == 'weak' else f'ONE field split across {dist.world} GPUs by ' + sh['cut']}),"
                                   f" p as {dict(field='full field', level='137-level vector in LDS', hybrid='hybrid levels formed in-kernel from sp + A/B tables')[args.pmode]}",
                       "entry_point": f"ekm_{entry}_{args.dtype}", "points_per_gpu": n_total // dist.world, "p_mode": args.pmode,
                       "points_total": n_total, "shard_cut": sh["cut"], "per_rank": per_rank},
            "roofline": roof, "sustained": sustained, "cpu_baseline": cpu, "parity": parity, "end_to_end": e2e,
            "parity_last_rank": parity_last, "shard_window_last_rank": shard_window,
            "hip_device_count": ndev_min, "devices_used": devices, "oversubscribed": oversub,
            # `value` = points x steps / timed_region; barrier_skew_ms = latest minus earliest start stamp of the ranks
            # (one monotonic clock per node): the part of the region that is the barrier's exit skew, not GPU time
            "timed_region_ms": round(elapsed * 1e3, 4), "barrier_skew_ms": round(barrier_skew_ms, 4), "end_skew_ms": round(end_skew_ms, 4),
        }
        if kernel_ms:
            # `value` is the metric: points x steps / wall time between the barriers (slowest rank).  Beside it the same
            # quantity from the HIP events of the slowest rank alone -- at N > 1 a strong-scaling step is under a millisecond
            # and the barriers' exit skew is of that order, so the two differ; the event figure is a cross-check, not the metric.
            line["value_from_kernel_ms"] = None if value is None else n_total / (kernel_ms * 1e-3)
            line["value_is"] = "points x steps / wall time of the slowest rank (barrier to device sync); value_from_kernel_ms = points / slowest rank's HIP-event time per launch"
        if args.dry_run:
            line["dry_run"] = True
        print(json.dumps(line), flush=True)
    dist.close()


CENSUS_KIND = {"full": ("full", 5), "p3": ("p3", None), "wetbulb": ("wetbulb", 0), "wetbulb_bisect": ("bisect", None),
               "wetbulb_bisect_bolton35": ("bisect:bolton35", None), "wetbulb_bisect_bolton39": ("bisect:bolton39", None)}
NP_CALL = {"full": ("pipeline_full", {}), "p3": ("pipeline_svp_td_rh", {}),
           "wetbulb": ("wet_bulb_temperature_from_specific_humidity", dict(ept_method="ifs", t_method="newton")),
           "wetbulb_bisect": ("wet_bulb_temperature_from_specific_humidity", dict(ept_method="ifs", t_method="bisect")),
           "wetbulb_bisect_bolton35": ("wet_bulb_temperature_from_specific_humidity", dict(ept_method="bolton35", t_method="bisect")),
           "wetbulb_bisect_bolton39": ("wet_bulb_temperature_from_specific_humidity", dict(ept_method="bolton39", t_method="bisect")),
           "rh": ("relative_humidity_from_specific_humidity", {}), "ept": ("ept_from_specific_humidity", dict(method="ifs"))}


def slab_parity(args, t, q, p, plev, outs, nlev, np_dtype):
    """every point of the first `--parity-slab-levels` levels of the TIMED arrays (8 x 1800 x 3600 = 51.84 M
    points) against the oracle in the same dtype, on a pool of spawned host processes (oracle/census.py: they never touch
    HIP; a level travels through one shared-memory file, nothing is pickled).  Per output: max relative error, points
    beyond the bar, NaN-pattern mismatches; for the Newton wet-bulb the census of Davies-Jones regime ties (`regime_flips`:
    points where the fp32 reference's own rounding took the other regime and the output under test sides with the fp64
    reference); for the bisection the census in quanta.  Bounded like cpu_baseline: ~1 s of oracle time per level on 16 cores."""
    from oracle import census

    if args.workload not in CENSUS_KIND or args.pmode == "hybrid":
        return {"skipped": f"no slab census for workload {args.workload} with p as {args.pmode} (windows only)"}
    kind, tw_index = CENSUS_KIND[args.workload]
    levels = list(range(min(args.parity_slab_levels, nlev)))
    tol = 1e-4 if args.dtype == "f32" else 1e-6
    pl = plev.to_host()

    def fetch(lev, rows):
        lo, hi = lev * INNER, (lev + 1) * INNER
        rows[0] = t.flat_slice(lo, hi).to_host()
        rows[1] = q.flat_slice(lo, hi).to_host()
        rows[2] = p.flat_slice(lo, hi).to_host() if p is not None else pl[lev]
        for k, o in enumerate(outs):
            rows[3 + k] = o.flat_slice(lo, hi).to_host()

    t0 = time.perf_counter()
    total, _ = census.run_levels(fetch, levels, INNER, np_dtype, kind, len(outs), tw_index=tw_index, tol=tol)
    res = {"points": len(levels) * INNER, "levels": [levels[0], levels[-1]], "tolerance": tol, "excluded_points": 0,
           "seconds": round(time.perf_counter() - t0, 2),
           "what": "every point of these levels of the timed arrays vs oracle/thermo_oracle.py in the same dtype (oracle/census.py)"}
    if kind.startswith("bisect"):
        b = total[0]
        res.update(identical=b["identical"], one_quantum=b["one_quantum"], two_quanta=b["two_quanta"], more_than_two_quanta=b["more"],
                   nan_mismatch=b["nan_mismatch"], differing_points_unanchored=b["differ_unanchored"],
                   max_rel_err=b["max_quanta"] * 120.0 / 4096.0 / 250.0,
                   ok=bool(b["more"] == 0 and b["differ_unanchored"] == 0 and b["identical"] >= 0.999 * b["n"]))
        return res
    res.update(max_rel_err=max(e["max_rel"] for e in total), max_rel_err_per_output=[float(f"{e['max_rel']:.3g}") for e in total],
               over=sum(e["over"] for e in total), nan_mismatch=sum(e["nan_mismatch"] for e in total))
    ok = res["nan_mismatch"] == 0
    if tw_index is not None:
        e = total[tw_index]
        res.update(regime_flips=e["over_regime_flip_of_the_fp32_reference"], regime_boundary_points_1e5=e["band_1e5"],
                   tw_over=e["over"], tw_over_outside_the_1e6_band=e["over_outside_band_1e6"], tw_over_unexplained=e["over_unexplained"],
                   tw_over_vs_fp64_oracle=e["over_vs_fp64_oracle"], reference_fp32_vs_fp64_over=e["reference_fp32_vs_fp64_over"])
        others = sum(x["over"] for k, x in enumerate(total) if k != tw_index)
        # the bar of tests/test_gpu_configs.py: nothing outside the regime band misses, nothing unexplained, and inside the
        # band no more misses than twice the reference's own fp32-vs-fp64 disagreements
        ok = ok and others == 0 and e["over_outside_band_1e6"] == 0 and e["over_unexplained"] == 0 and \
            e["over"] <= 2 * e["reference_fp32_vs_fp64_over"]
    else:
        ok = ok and res["over"] == 0
    res["ok"] = bool(ok)
    return res


def end_to_end(args, t, q, p, plev, nlev, np_dtype):
    """the 8-level slab through the NumPy-in / NumPy-out path -- pageable host arrays in, host arrays out --,
    PCIe-inclusive; reported beside `value`, never as it.  `call_ms` is the call a user makes (best of 3; the library streams it
    in slices, so the three phases overlap); h2d_ms / kernel_ms / d2h_ms are the same work done one phase at a time."""
    import ekm_hip
    from ekm_hip import thermo

    if args.workload not in NP_CALL or args.pmode == "hybrid":
        return None
    name, kw = NP_CALL[args.workload]
    fn = getattr(thermo, name)
    L = min(8, nlev)
    n = L * INNER
    ht, hq = t.flat_slice(0, n).to_host().reshape(L, INNER), q.flat_slice(0, n).to_host().reshape(L, INNER)
    hp = p.flat_slice(0, n).to_host().reshape(L, INNER) if p is not None else plev.to_host()[:L].reshape(L, 1)
    ins = (ht, hq, hp)

    def best(f, reps=3):
        b, r = float("inf"), None
        for _ in range(reps):
            r = None  # the previous result is released outside the timed region
            t0 = time.perf_counter()
            r = f()
            ekm_hip.synchronize()
            b = min(b, time.perf_counter() - t0)
        return b * 1e3, r

    for _ in range(2):
        fn(*ins, **kw)  # warm: the pinned result pool (hipHostMalloc pins page by page), lane streams, first-use tables
    call_ms, res = best(lambda: fn(*ins, **kw), reps=5)
    nout = len(res) if isinstance(res, tuple) else 1
    res = None
    h2d_ms, dins = best(lambda: [ekm_hip.to_device(a) for a in ins])
    kernel_ms, douts = best(lambda: fn(*dins, **kw), reps=5)
    douts = list(douts) if isinstance(douts, tuple) else [douts]
    # downloads into host arrays that already exist (a fresh pageable array faults page by page under the DMA: 22 GB/s
    # instead of the link's 50; the call itself downloads into pooled pinned blocks)
    houts = [np.empty(o.shape, o.dtype) for o in douts]
    for h in houts:
        h.fill(0)
    d2h_ms, _ = best(lambda: [o.to_host(out=h) for o, h in zip(douts, houts)])
    for a in dins + douts:
        a.free()
    nbytes = sum(a.nbytes for a in ins) + nout * n * np.dtype(np_dtype).itemsize
    return {"what": f"thermo.{name} on a {L}-level slab of the timed field, NumPy arrays in (pageable) and out, over PCIe",
            "points": n, "arrays_in": len(ins), "arrays_out": nout, "bytes": int(nbytes),
            "h2d_ms": round(h2d_ms, 3), "kernel_ms": round(kernel_ms, 3), "d2h_ms": round(d2h_ms, 3),
            "phases_sum_ms": round(h2d_ms + kernel_ms + d2h_ms, 3), "call_ms": round(call_ms, 3),
            "gbs": round(nbytes / (call_ms * 1e-3) / 1e9, 1), "points_per_s": n / (call_ms * 1e-3),
            "note": "PCIe-inclusive; `value` is the HBM-resident rate (inputs already on the device when the timed region starts).  "
                    "h2d_ms / d2h_ms: pageable host arrays, one synchronous copy after the other; call_ms: the library's streamed "
                    "call (slices, uploads and downloads on two DMA engines at once, results in pooled pinned memory)"}


def time_buffer_sets(args, sh, dev, nlev, np_dtype, seed, nout, entry, ints, my_ms):
    """The timed kernel on `--buffer-sets` - 1 further, independently allocated copies of its input and output fields (same
    generator, same seed: the same values at other addresses), 3 warm-up + min(steps, 20) launches each, HIP events."""
    from ekm_hip import _ffi
    from ekm_hip.device import DeviceArray

    lib = _ffi.lib()
    fn = getattr(lib, f"ekm_{entry}_{args.dtype}")
    F = _ffi.Operand
    itemsize = np.dtype(np_dtype).itemsize
    out_ms = []
    k_steps = max(1, min(args.steps, 20))
    e0, e1, ms = C.c_void_p(), C.c_void_p(), C.c_float()
    _ffi.check(lib.ekm_event_create(dev, C.byref(e0)))
    _ffi.check(lib.ekm_event_create(dev, C.byref(e1)))
    # the sets are allocated one after the other and ALL kept until the end, so that no set reuses the blocks of another
    keep = []
    for _ in range(args.buffer_sets - 1):
        t, q, p, plev, hyb = build_inputs(args, sh, dev, nlev, np_dtype, seed)
        outs = [DeviceArray.empty((sh["n_local"],), np_dtype, dev) for _ in range(nout)]
        keep.append((t, q, p, plev, hyb, outs))
        op_t, op_q = F(t.ptr, _ffi.FIELD, 0, 0, 0), F(q.ptr, _ffi.FIELD, 0, 0, 0)
        if p is not None:
            op_p = F(p.ptr, _ffi.FIELD, 0, 0, 0)
        elif args.pmode == "hybrid":
            nz = np.flatnonzero(hyb["Bh"] != 0.0)
            nflat = int(max(0, (nz[0] if nz.size else hyb["Bh"].size) - 1))
            op_p = F(hyb["sp"].ptr, _ffi.HYBRID_FULL, nflat, sh["lev1"] - sh["lev0"], sh["col1"] - sh["col0"], hyb["A"].ptr, hyb["B"].ptr)
        else:
            op_p = F(plev.ptr + sh["lev0"] * itemsize, _ffi.LEVEL_MAJOR, 0, sh["lev1"] - sh["lev0"], INNER)
        operands = {"potential_temperature": (op_t, op_p), "saturation_vapour_pressure": (op_t,)}.get(entry, (op_t, op_q, op_p))
        cargs = [dev, None] + [C.byref(o) for o in operands] + list(ints) + [o.ptr for o in outs] + [sh["n_local"]]
        for _ in range(3):
            _ffi.check(fn(*cargs))
        _ffi.check(lib.ekm_event_record(dev, e0, None))
        for _ in range(k_steps):
            _ffi.check(fn(*cargs))
        _ffi.check(lib.ekm_event_record(dev, e1, None))
        _ffi.check(lib.ekm_sync(dev))
        _ffi.check(lib.ekm_event_elapsed_ms(dev, e0, e1, C.byref(ms)))
        out_ms.append(ms.value / k_steps)
    for t, q, p, plev, hyb, outs in keep:
        for a in [t, q, p, plev] + outs + ([hyb["A"], hyb["B"], hyb["sp"]] if hyb else []):
            if a is not None:
                a.free()
    return {"kernel_ms": out_ms,
            "what": f"set 0 = the timed region's arrays ({args.steps} launches); sets 1.. = fresh allocations of every field, same "
                    f"values, 3 warm-up + {k_steps} launches each, HIP events, same process"}


def sample_shard(args, t, q, p, plev, outs, sh, nlev, np_dtype, hyb=None):
    """Host copies of what the parity check needs from THIS rank's shard of the timed arrays: 256-point windows of up to 32
    levels spread over the whole column (the column workloads: 64 whole columns), inputs and every output, plus where
    the windows sit in the GLOBAL field.  Small enough to travel to rank 0 through the gather."""
    first, n_local, lev0 = sh["first"], sh["n_local"], sh["lev0"]
    by_columns = sh["cut"] == "columns"
    ncol = sh["col1"] - sh["col0"]  # row length of this shard's [level, column] layout
    smp = {"workload": args.workload, "dtype": args.dtype, "first": int(first), "lev0": int(lev0), "col0": int(sh["col0"]),
           "cut": sh["cut"]}
    if args.workload == "geopotential":  # whole columns: 64 columns x all levels
        c0, nc = min(4321, max(0, ncol - 64)), min(64, ncol)
        col = lambda a: np.stack([a.flat_slice(k * ncol + c0, k * ncol + c0 + nc).to_host() for k in range(nlev)])  # noqa: E731
        smp.update(kind="columns", t=col(t), q=col(q), zs=hyb["zsh"][c0:c0 + nc], sp=hyb["sph"][c0:c0 + nc], A=hyb["Ah"], B=hyb["Bh"],
                   out=col(outs[0]), global_col0=int(sh["col0"] + c0))
        return smp
    off = min(4321, max(0, (ncol if by_columns else INNER) - 256))
    wins = []
    for lev in np.linspace(0, nlev - 1, 32).round().astype(int):  # the whole column, hybrid top levels (1 Pa ...) included
        lo = int(lev) * ncol + off if by_columns else int(lev) * INNER - first + off
        if 0 <= lo and lo + 256 <= n_local:
            wins.append((int(lev), lo))
    wins = sorted(set(wins))
    if not wins:
        wins = [((first + 0) // INNER, 0)]
    nwin = min(256, n_local)
    grab = lambda a: np.concatenate([a.flat_slice(lo, lo + nwin).to_host() for _, lo in wins])  # noqa: E731
    smp.update(kind="windows", levels=[w[0] for w in wins], t=grab(t), q=grab(q), outs=[grab(o) for o in outs],
               global_index=[int(lo + (sh["col0"] if by_columns else first)) for _, lo in wins])
    if hyb is not None:
        from oracle import vertical_oracle as vo

        pf = vo.pressure_on_hybrid_levels(hyb["Ah"].astype(np_dtype), hyb["Bh"].astype(np_dtype), hyb["sph"][off:off + nwin])
        smp["p"] = np.concatenate([pf[lev - lev0] for lev, _ in wins]).astype(np_dtype)
    elif p is not None:
        smp["p"] = grab(p)
    else:
        pl = plev.to_host()
        smp["p"] = np.concatenate([np.full(nwin, pl[lev], np_dtype) for lev, _ in wins])
    return smp


def judge_sample(args, smp):
    """GPU outputs of a shard sample (sample_shard) vs the oracle on the same inputs."""
    tol = 1e-4 if args.dtype == "f32" else 1e-6
    np_dtype = np.float32 if args.dtype == "f32" else np.float64
    if smp["kind"] == "columns":
        from oracle import vertical_oracle as vo

        got = smp["out"].astype(np.float64)
        # bar: the reference's own fp32 tolerance for this chain (atol 10 m2/s2, rtol 1e-6), against the oracle
        # evaluated in fp64 on the same inputs (the fp32 reference is itself 2 % off in alpha for thin layers)
        want = vo.geopotential_on_hybrid_levels(*(np.asarray(x, np.float64) for x in (
            smp["t"], smp["q"], smp["zs"], smp["A"], smp["B"], smp["sp"])))
        with np.errstate(all="ignore"):
            aerr = np.abs(got - want)
            r = aerr / np.abs(want)
        nanmm = int((np.isnan(got) != np.isnan(want)).sum())
        ok = bool(nanmm == 0 and np.all(aerr <= 10.0 + 1e-6 * np.abs(want))) if args.dtype == "f32" else bool(
            nanmm == 0 and np.nanmax(r) <= tol)
        return {"points": int(got.size), "max_rel_err": float(np.nanmax(r)), "max_abs_err": float(np.nanmax(aerr)),
                "nan_mismatch": nanmm, "tolerance": "atol 10 m2/s2 + rtol 1e-6 (reference's fp32 bar)" if args.dtype == "f32" else tol,
                "excluded_regime_boundary_points": 0, "ok": ok}
    ht, hq, hp, gouts = smp["t"], smp["q"], smp["p"], smp["outs"]
    want = (hp,) if args.workload == "hybrid_levels" else oracle_call(args.workload, ht, hq, hp)
    if args.workload.startswith("wetbulb_bisect"):
        # the 12-step sign search is quantised to 120/4096 K.  No point is excluded: every differing point must be a NaN
        # one of the reference's two precisions also has, or lie within 2 quanta of the fp32 / fp64 reference or of a
        # lattice temperature where the reference's own residual is rounding noise (oracle/census.py::bisect_job)
        from oracle import census

        method = args.workload.rsplit("_", 1)[1] if args.workload.count("_") > 1 else "ifs"
        b = census.bisect_job(dict(t=ht, q=hq, p=hp, got=gouts[0], method=method))[0]
        return {"points": int(ht.size), "identical": b["identical"], "one_quantum": b["one_quantum"],
                "two_quanta": b["two_quanta"], "more_than_two_quanta": b["more"], "nan_mismatch": b["nan_mismatch"],
                "differing_points_unanchored": b["differ_unanchored"], "max_rel_err": b["max_quanta"] * 120.0 / 4096.0 / 250.0,
                "tolerance": "2 quanta of the 12-step search = 0.0586 K absolute; differing points anchored to the fp32 / fp64 "
                             "reference or to a noise point of the reference's own residual",
                "excluded_points": 0, "ok": bool(b["more"] == 0 and b["differ_unanchored"] == 0)}
    # No point is excluded: the kernels settle Davies-Jones regime ties in double (csrc/thermo_math.hpp::davies_regime)
    worst, nan_mismatch, explained = 0.0, 0, 0
    for k, (o, w) in enumerate(zip(gouts, want)):
        g = o.astype(np.float64)
        w = np.asarray(w, dtype=np.float64)
        nanmm = np.isnan(g) != np.isnan(w)
        with np.errstate(all="ignore"):
            r = np.abs(g - w) / np.abs(w)
        r = np.where(np.isfinite(r), r, 0.0)
        is_tw = args.workload == "wetbulb" or (args.workload == "full" and k == 5)
        if is_tw and ((r > tol) | nanmm).any():
            # hPa-level pressures (hybrid top levels): a miss counts unless the reference's OWN one-step Newton amplifies
            # input perturbations enough to explain it (oracle/conditioning.py::newton_misses_explained, fp64 oracle only)
            from oracle import conditioning

            miss = np.flatnonzero((r > tol) | nanmm)
            fin, edge = conditioning.newton_misses_explained(ht[miss], hq[miss], hp[miss], g[miss], w[miss], tol)
            ok = fin | edge
            explained += int(ok.sum())
            r[miss[ok]] = 0.0
            nanmm[miss[ok]] = False
        nan_mismatch += int(nanmm.sum())
        worst = max(worst, float(r.max()) if r.size else 0.0)
    return {"points": int(ht.size), "max_rel_err": worst, "nan_mismatch": nan_mismatch, "tolerance": tol,
            "excluded_points": 0, "tw_misses_explained_by_reference_amplification": explained,
            "levels_sampled": smp["levels"][:1] + smp["levels"][-1:],
            "ok": bool(nan_mismatch == 0 and worst <= tol)}


def valu_units(counts):
    """Issue units per point from per-point executed wave-instruction counts (x 64 lanes / points): every VALU instruction
    counts 1, the slower classes their extra issue time (VALU_WEIGHTS - 1)."""
    units = counts["SQ_INSTS_VALU"]
    for name, w in VALU_WEIGHTS.items():
        units += (w - 1.0) * counts.get(name, 0.0)
    return units


def valu_from_profiles(args, why=""):
    """Executed VALU instructions per point of this workload by class, from a rocprofv3 --pmc pass of this same command
    (tools/profile_valu.sh) committed in profiles/valu_latest.json; used when the in-run pass is unavailable."""
    try:
        with open(os.path.join(ROOT, "profiles", "valu_latest.json")) as f:
            d = json.load(f)
        counts = d["counts_per_point"].get(f"{args.workload}:{args.pmode}:{args.dtype}")
        if counts is None:
            return None
        return {"units_per_point": round(valu_units(counts), 2), "counters": {k: round(v, 2) for k, v in counts.items()},
                "source": "profiles/valu_latest.json (" + d["source"] + "), committed; not re-measured in this run" + (f" ({why})" if why else "")}
    except Exception:
        return None


def _child_cmd(args, steps, warm):
    return [sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", str(warm), "--no-cpu-baseline",
            "--traffic", "none", "--valu", "none", "--sustain", "0", "--no-stream-ceiling", "--parity-slab-levels", "0", "--no-end-to-end",
            "--buffer-sets", "1", "--workload", args.workload, "--pmode", args.pmode,
            "--dtype", args.dtype, "--levels", str(args.levels)]


def rocprof_pass(args, counters, steps=3, warm=1):
    """One child run of this script under `rocprofv3 --pmc <counters> --kernel-trace` (the program itself right after `--`,
    counters in a pass of their own, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  Returns ({kernel: {counter:
    [value per launch]}}, None) for the kernels of the timed step, or (None, why not)."""
    import collections
    import csv
    import glob
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="ekm_pmc_", dir="/tmp")
    try:
        cmd = [exe, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", tmp, "--"] + _child_cmd(args, steps, warm)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
        if r.returncode != 0:
            return None, f"rocprofv3 --pmc {' '.join(counters)} failed (rc {r.returncode})"
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for path in glob.glob(os.path.join(tmp, "**", "*_counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path)):
                if row["Counter_Name"] in counters:
                    agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        # the kernels of the timed step: launched warm-up + steps times (fill / table kernels run once)
        ks = {k: dict(v) for k, v in agg.items() if any(t in k for t in ("map_", "geopotential_columns", "hybrid_levels", "hybrid_rows"))
              and all(len(x) == steps + warm for x in v.values())}
        if not ks:
            return None, f"no kernel with {steps + warm} launches in the {' '.join(counters)} pass"
        return ks, None
    except Exception as exc:  # the profiler must never take the benchmark down
        return None, f"profiler pass failed: {type(exc).__name__}: {exc}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_traffic(args, entry):
    """HBM bytes per step of THIS workload, measured now: two child runs of this script (3 timed steps each) under
    `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` -- separate passes -- summed over
    the map kernels of one step.  gfx950 corrections (MI355X_MICROARCH.md): the counters are in KiB, and FETCH_SIZE reports
    half of a 16-B-per-lane streaming read.  Returns (bytes per step, description) or (None, why not)."""
    total = 0.0
    for counter, scale in (("FETCH_SIZE", 2.0 * 1024.0), ("WRITE_SIZE", 1024.0)):
        ks, why = rocprof_pass(args, [counter])
        if ks is None:
            return None, why
        total += scale * sum(sum(v[counter]) / len(v[counter]) for v in ks.values())
    return total, ("measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate child passes of this command, "
                   "3 steps each), KiB x 1024, FETCH_SIZE x 2 (gfx950: half of a 16-B/lane streaming read is reported), "
                   "summed over the kernels of one step")


def measure_valu(args, n_local):
    """Executed VALU wave-instructions per point of THIS workload by class, measured now: one more child run under
    `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 [fp64 classes] --kernel-trace`, summed over the kernels of one
    step, x 64 lanes / points.  Returns ({units_per_point, counters, source}, None) or (None, why not)."""
    names = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_TRANS_F32"]
    if args.dtype == "f64":
        names += ["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"]
    ks, why = rocprof_pass(args, names)
    if ks is None and args.dtype == "f64":
        return None, why + "; the fp64 instruction classes are needed to weigh an fp64 kernel"
    if ks is None:
        return None, why
    counts = {c: sum(sum(v.get(c, [0.0])) / max(1, len(v.get(c, [0.0]))) for v in ks.values()) * 64.0 / n_local for c in names}
    return {"units_per_point": round(valu_units(counts), 2), "counters": {k: round(v, 2) for k, v in counts.items()},
            "source": "measured in this run: rocprofv3 --pmc " + " ".join(names) + " (one child pass of this command, 3 steps), wave-"
                      "instructions x 64 lanes / points, summed over the kernels of one step; issue units = every VALU instruction 1 + "
                      "the extra issue time of the slower classes (fp32 transcendental 4, fp64 add/mul/fma 2, fp64 rcp/rsq/sqrt 6.6)"}, None


def traffic_from_profiles(args, n_local):
    """HBM bytes per launch measured with the PMC counters (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate
    rocprofv3 passes of this same command, committed as profiles/traffic_latest.json), scaled to this
    launch's share of the field; null when no PMC pass exists for the workload."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as f:
            d = json.load(f)
        whole = d.get(f"{args.workload}:{args.pmode}:{args.dtype}:{args.levels}")
        return None if whole is None else whole * n_local / (args.levels * INNER)
    except Exception:
        return None


if __name__ == "__main__":
    main()
