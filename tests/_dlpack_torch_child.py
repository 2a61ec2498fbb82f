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
    host_t = ek.thermo.potential_temperature(torch.from_numpy(t), torch.from_numpy(p))   # CPU tensors: NumPy semantics
    assert isinstance(host_t, np.ndarray)
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

# a second round on our own non-default stream: producer is handed that stream
s1 = ek.stream_create()
ek.set_stream(s1)
big = torch.rand(1 << 24, device=dev) * 50.0 + 250.0   # asynchronous on torch's stream
es = ek.thermo.saturation_vapour_pressure(ek.from_dlpack(big))
got = torch.from_dlpack(es).cpu().numpy()
ek.set_stream(None)
ref = ek.thermo.saturation_vapour_pressure(ek.to_device(big.cpu().numpy())).to_host()
assert np.array_equal(got, ref), "stream hand-over lost ordering"

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
