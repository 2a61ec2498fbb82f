"""ekm_hip -- MI355X-native backend for the `earthkit.meteo.thermo` hot path.

    from ekm_hip import thermo
    theta = thermo.potential_temperature(t, p)          # NumPy in -> NumPy out (computed on the GPU)
    d_t = ekm_hip.to_device(t); ...                      # DeviceArray in -> DeviceArray out (stays in HBM)

Hand-written HIP kernels for gfx950 behind a C ABI (include/ekm_thermo.h),
called through ctypes.  No PyTorch / CuPy / Triton on the product path and no
CPU fallback: a missing library or GPU raises.
"""
from . import thermo, vertical, wind  # noqa: F401
from ._ffi import EkmError, EkmLibraryError  # noqa: F401
from .device import (  # noqa: F401
    DeviceArray,
    current_device,
    current_stream,
    device_count,
    device_info,
    empty_cache,
    memory_stats,
    order_streams,
    pinned_empty,
    multi_gpu,
    set_device,
    set_stream,
    shard_bounds,
    stream_create,
    stream_destroy,
    synchronize,
    to_device,
)

__version__ = "0.1.0"
from .vertical import HybridPressure  # noqa: F401,E402
from .dlpack import from_dlpack  # noqa: F401,E402
from ._streamed import release_streams  # noqa: F401,E402
from ._graph import Graph, graph  # noqa: F401,E402
