"""Child process of tests/test_gpu_streaming.py::test_dlpack_with_torch_as_foreign_producer_and_consumer.

torch (a FOREIGN ROCm array library; test infrastructure only, the product never imports it) is imported and
initialised first, then ekm_hip.  Exit code 77 = torch has no ROCm device here (skip)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "earthkit-meteo_amd")]

import numpy as np  # noqa: E402

try:
    import torch
except ImportError:
    print("torch is not installed")
    sys.exit(77)
if not torch.cuda.is_available():
    print("torch sees no ROCm device")
    sys.exit(77)
torch.zeros(1, device="cuda").cpu()  # initialise torch's HIP context before the other library loads

import ekm_hip as ek  # noqa: E402
from oracle import synthetic  # noqa: E402

np.seterr(all="ignore")
t, q, p, _ = synthetic.make_fields(4, 1 << 20, dtype=np.float32, seed=5)
d = [ek.to_device(a) for a in (t, q, p)]
want = ek.thermo.relative_humidity_from_specific_humidity(*d).to_host()  # our own path, our own memory

dev = torch.device("cuda", ek.current_device())
side = torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):  # producer work on a non-default torch stream, still in flight at hand-over
    tt = torch.from_numpy(t).to(dev) * 1.0
    tq = torch.from_numpy(q).to(dev) * 1.0
    tp = torch.from_numpy(p).to(dev) * 1.0
    dt_, dq_, dp_ = (ek.from_dlpack(x) for x in (tt, tq, tp))  # hands torch OUR stream: torch orders its work before it
assert dt_.ptr == tt.data_ptr() and dq_.ptr == tq.data_ptr() and dp_.ptr == tp.data_ptr(), "copy on the way in"
assert dt_.shape == t.shape and dt_.dtype == np.float32 and dt_.device == ek.current_device()
rh = ek.thermo.relative_humidity_from_specific_humidity(dt_, dq_, dp_)
back = torch.from_dlpack(rh)  # torch passes ITS current stream: ordered after our kernel on the device
assert back.data_ptr() == rh.ptr and tuple(back.shape) == t.shape and back.dtype == torch.float32, "copy on the way out"
doubled = (back * 2.0).cpu().numpy()
assert np.array_equal(back.cpu().numpy(), want, equal_nan=True), "kernel on torch memory differs from our own path"
assert np.array_equal(doubled, want * 2.0, equal_nan=True)

# IMPLICIT: torch tensors straight into the thermo functions (the reference's array_namespace(*inputs) dispatch,
# thermo/array/thermo.py:826): taken over through DLPack, and -- every array input being torch's -- handed back as a
# torch tensor on the same device: zero copy both ways, stream-ordered, the bits of the DeviceArray path
with torch.cuda.stream(side):
    ut, uq, up = tt * 1.0, tq * 1.0, tp * 1.0   # still in flight on torch's side stream at the call
    out = ek.thermo.relative_humidity_from_specific_humidity(ut, uq, up)
    assert isinstance(out, torch.Tensor) and out.device == ut.device and out.dtype == torch.float32 and tuple(out.shape) == t.shape
    assert np.array_equal(out.cpu().numpy(), want, equal_nan=True), "implicit torch path differs from the DeviceArray path"
    es_, td_, rh_ = ek.thermo.pipeline_svp_td_rh(ut, uq, up)
    assert all(isinstance(x, torch.Tensor) for x in (es_, td_, rh_)) and np.array_equal(rh_.cpu().numpy(), want, equal_nan=True)
    mixed = ek.thermo.potential_temperature(ut, 85000.0)           # Python scalars do not change whose call it is
    assert isinstance(mixed, torch.Tensor)
    plain = ek.thermo.potential_temperature(ut, d[2])               # a DeviceArray among the inputs: DeviceArray out
    assert isinstance(plain, ek.DeviceArray)
    strided = ek.thermo.saturation_vapour_pressure(ut.t())          # not C-contiguous: the producer compacts it
    assert isinstance(strided, torch.Tensor) and tuple(strided.shape) == tuple(ut.t().shape)
    assert np.array_equal(strided.cpu().numpy(), ek.thermo.saturation_vapour_pressure(np.ascontiguousarray(t.T)), equal_nan=True)
    # CPU tensors: computed through the NumPy path, and -- as the reference does for every backend -- handed back as torch
    host_t = ek.thermo.potential_temperature(torch.from_numpy(t), torch.from_numpy(p))
    assert isinstance(host_t, torch.Tensor) and host_t.device.type == "cpu" and host_t.dtype == torch.float32
    assert np.array_equal(host_t.numpy(), ek.thermo.potential_temperature(t, p), equal_nan=True)
    host_s = ek.thermo.potential_temperature(torch.tensor(264.12, dtype=torch.float64), 85000.0)   # 0-d tensor: 0-d tensor
    assert isinstance(host_s, torch.Tensor) and host_s.ndim == 0 and abs(float(host_s) - 276.672291) < 1e-5
    host_mix = ek.thermo.potential_temperature(torch.from_numpy(t), p)                             # NumPy among them: NumPy out
    assert isinstance(host_mix, np.ndarray)
    # the vertical functions take foreign arrays the same way (ekm_hip.vertical._foreign_aware)
    A_, B_ = ek.vertical.hybrid_level_parameters(137)
    sp_t = torch.full((64, 32), 101325.0, device=dev) * (1.0 - 0.3 * torch.rand(64, 32, device=dev))
    pf = ek.vertical.pressure_on_hybrid_levels(A_.astype(np.float32), B_.astype(np.float32), sp_t)
    assert isinstance(pf, torch.Tensor) and tuple(pf.shape) == (137, 64, 32) and pf.dtype == torch.float32
    pf_host = ek.vertical.pressure_on_hybrid_levels(A_.astype(np.float32), B_.astype(np.float32), sp_t.cpu().numpy())
    assert np.array_equal(pf.cpu().numpy(), pf_host)
    t_t = torch.full((137, 64, 32), 250.0, device=dev) + 30.0 * torch.rand(137, 64, 32, device=dev)
    q_t = torch.full((137, 64, 32), 1e-3, device=dev)
    z = ek.vertical.relative_geopotential_thickness_on_hybrid_levels(t_t, q_t, A_.astype(np.float32), B_.astype(np.float32), sp_t)
    z_host = ek.vertical.relative_geopotential_thickness_on_hybrid_levels(t_t.cpu().numpy(), q_t.cpu().numpy(), A_.astype(np.float32),
                                                                          B_.astype(np.float32), sp_t.cpu().numpy())
    assert isinstance(z, torch.Tensor) and np.array_equal(z.cpu().numpy(), z_host)
    del out, es_, td_, rh_, mixed, plain, strided, pf, z
torch.cuda.synchronize()

# the reference parametrises EVERY function over its array backends (tests/thermo/test_thermo.py:73 ...): every public
# function x variant of the case table with torch-ROCm inputs against the DeviceArray path -- the same bits, torch out
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from _case_table import case_table  # noqa: E402

rng = np.random.default_rng(11)
npts = 1 << 14
col = {"t": rng.uniform(215.0, 315.0, npts), "p": 10.0 ** rng.uniform(3.0, 5.02, npts)}
col["tc"] = col["t"] - 273.16
col["t2"] = col["t"] - rng.uniform(0.0, 15.0, npts)
col["td"] = col["t"] - rng.uniform(0.0, 25.0, npts)
col["q"] = 10.0 ** rng.uniform(-6.0, -1.8, npts)
col["w"] = col["q"] / (1.0 - col["q"])
col["p2"] = col["p"] * rng.uniform(0.5, 1.0, npts)
col["e"] = col["p"] * col["q"] / 0.622
col["es"] = 611.21 * np.exp(17.502 * (col["t"] - 273.16) / (col["t"] - 32.19))
col["r"] = rng.uniform(1.0, 100.0, npts)
col["th"] = col["t"] * (1e5 / col["p"]) ** 0.2857
col["ept"] = col["th"] + rng.uniform(0.0, 40.0, npts)
ncase = 0
for tag, npdt, tdt in (("f32", np.float32, torch.float32), ("f64", np.float64, torch.float64)):
    host = {k: v.astype(npdt) for k, v in col.items()}
    tens = {k: torch.from_numpy(v).to(dev) for k, v in host.items()}
    devs = {k: ek.to_device(v) for k, v in host.items()}
    for func, names, kwargs in case_table():
        fn = getattr(ek.thermo, func)
        got = fn(*[tens[n] for n in names], **kwargs)
        want_ = fn(*[devs[n] for n in names], **kwargs)
        got = got if isinstance(got, tuple) else (got,)
        want_ = want_ if isinstance(want_, tuple) else (want_,)
        for g, w in zip(got, want_):
            assert isinstance(g, torch.Tensor) and g.device == dev and g.dtype == tdt and tuple(g.shape) == (npts,), (func, kwargs, type(g))
            assert np.array_equal(g.cpu().numpy(), w.to_host(), equal_nan=True), f"{func} {kwargs} {tag}: torch path differs from the DeviceArray path"
        ncase += 1
    # ... and with torch CPU tensors (the NumPy path underneath): torch CPU tensors out, the same bits again
    for func, names, kwargs in case_table()[::7]:
        fn = getattr(ek.thermo, func)
        got = fn(*[torch.from_numpy(host[n]) for n in names], **kwargs)
        want_ = fn(*[devs[n] for n in names], **kwargs)
        for g, w in zip(got if isinstance(got, tuple) else (got,), want_ if isinstance(want_, tuple) else (want_,)):
            assert isinstance(g, torch.Tensor) and g.device.type == "cpu" and g.dtype == tdt
            assert np.array_equal(g.numpy(), w.to_host(), equal_nan=True), f"{func} {kwargs} {tag}: torch CPU path differs"
    del tens, devs
print(f"torch-ROCm inputs: {ncase} function x variant x dtype cases bit-equal to the DeviceArray path")

# a second round on our own non-default stream: producer is handed that stream
s1 = ek.stream_create()
ek.set_stream(s1)
big = torch.rand(1 << 24, device=dev) * 50.0 + 250.0   # asynchronous on torch's stream
es = ek.thermo.saturation_vapour_pressure(ek.from_dlpack(big))
got = torch.from_dlpack(es).cpu().numpy()
ek.set_stream(None)
ref = ek.thermo.saturation_vapour_pressure(ek.to_device(big.cpu().numpy())).to_host()
assert np.array_equal(got, ref), "stream hand-over lost ordering"

# a torch tensor taken over BEFORE an ekm_hip.graph() block, used as an operand inside it (the route _graph.py and the
# `_adopt_foreign` message recommend): the recording stream gets no event, the graph pins the borrowed tensor (its deleter
# waits for close()), and a replay after torch has changed the tensor in place computes on the new contents
gt = torch.rand(1 << 20, device=dev) * 60.0 + 240.0
gp = torch.full((1 << 20,), 85000.0, device=dev)
torch.cuda.synchronize()
bt, bp = ek.from_dlpack(gt), ek.from_dlpack(gp)
with ek.graph() as g:
    gth = ek.thermo.potential_temperature(bt, bp)
owner = bt._alloc
assert owner.pins == 1, "the graph did not adopt the borrowed tensor"
for rnd in range(3):
    gt.mul_(1.0 + 0.01 * rnd).add_(0.5)   # torch mutates the operand in place, on its own stream
    torch.cuda.synchronize()               # (documented: synchronise the producer before launch)
    g.launch().synchronize()
    eager = ek.thermo.potential_temperature(ek.to_device(gt.cpu().numpy()), ek.to_device(gp.cpu().numpy())).to_host()
    assert np.array_equal(gth.to_host(), eager), f"graph replay {rnd} on a DLPack operand differs from the eager call"
bt.free()                                  # released while the graph still holds the address: the deleter must wait
assert owner.free_pending and owner.managed is not None, "a pinned borrowed tensor was released under a live graph"
g.launch().synchronize()
g.close()
assert owner.managed is None, "the producer's deleter did not run after Graph.close()"
del bt, bp, gth, owner
print("graph with DLPack operands: 4 replays across in-place torch updates, deleter deferred until close()")

# rejections
for bad, exc in ((torch.ones(4), TypeError), (torch.ones(4, dtype=torch.int32, device=dev), TypeError),
                 (torch.ones(4, 4, device=dev).t(), ValueError)):
    try:
        ek.from_dlpack(bad)
    except exc:
        pass
    else:
        raise AssertionError(f"from_dlpack accepted {bad.dtype} {bad.device} contiguous={bad.is_contiguous()}")
del back, rh, dt_, dq_, dp_, es
ek.synchronize()
print("DLPACK_TORCH_OK: zero-copy in and out, stream-ordered, torch", torch.__version__)
