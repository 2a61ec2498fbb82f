"""ctypes binding of libekm_thermo.so (the C ABI in include/ekm_thermo.h).

No torch, no cupy: the only native dependency is the in-tree shared library
built from earthkit-meteo_amd/csrc by `make` / `__graft_entry__.build()`.
There is no CPU fallback -- a missing library or a missing GPU is an error.
"""
import ctypes as C
import os
import threading

from ._optable import OPS

EKM_OK = 0
ABI_VERSION = 5  # include/ekm_thermo.h: EKM_ABI_VERSION (tests/test_abi.py keeps the two equal)
EKM_ERR_HIP, EKM_ERR_ARG, EKM_ERR_ENUM, EKM_ERR_NODEV = -1, -2, -3, -4
FIELD, SCALAR, LEVEL_MAJOR, LEVEL_MINOR, HYBRID_FULL = 0, 1, 2, 3, 4

PHASE = {"mixed": 0, "water": 1, "ice": 2}
EPT_METHOD = {"ifs": 0, "bolton35": 1, "bolton39": 2}
T_METHOD = {"bisect": 0, "newton": 1, "direct": 2}
LCL_METHOD = {"davies": 0, "bolton": 1}


class EkmError(RuntimeError):
    """A call into libekm_thermo.so failed (HIP error, bad argument, no device)."""


class EkmLibraryError(ImportError):
    """libekm_thermo.so is missing or cannot be loaded."""


class Operand(C.Structure):
    _fields_ = [("data", C.c_void_p), ("mode", C.c_int32), ("nflat", C.c_int32),
                ("len", C.c_uint64), ("inner", C.c_uint64), ("aux0", C.c_void_p), ("aux1", C.c_void_p)]


_lib = None
_lock = threading.Lock()


def library_path():
    env = os.environ.get("EKM_THERMO_LIB")
    if env:
        return env
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libekm_thermo.so")


def _signatures():
    vp, sz, i, u64, u32 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint64, C.c_uint32
    pvp = C.POINTER(C.c_void_p)
    sig = {
        "ekm_init": ([], i), "ekm_device_count": ([], i),
        "ekm_last_error": ([], C.c_char_p), "ekm_version": ([], C.c_char_p), "ekm_abi_version": ([], i),
        "ekm_device_name": ([i, C.c_char_p, sz], i), "ekm_device_cus": ([i], i),
        "ekm_mem_info": ([i, C.POINTER(sz), C.POINTER(sz)], i),
        "ekm_malloc": ([i, sz, pvp], i), "ekm_free": ([i, vp], i),
        "ekm_host_alloc": ([sz, pvp], i), "ekm_host_free": ([vp], i),
        "ekm_host_prefault": ([vp, sz, i], i),
        "ekm_h2d": ([i, vp, vp, sz, vp], i), "ekm_d2h": ([i, vp, vp, sz, vp], i),
        "ekm_d2d": ([i, vp, vp, sz, vp], i), "ekm_memset": ([i, vp, i, sz, vp], i),
        "ekm_fill_u32": ([i, vp, u32, sz, vp], i),
        "ekm_sync": ([i], i),
        "ekm_stream_create": ([i, pvp], i), "ekm_stream_destroy": ([i, vp], i), "ekm_stream_sync": ([i, vp], i),
        "ekm_event_create": ([i, pvp], i), "ekm_event_destroy": ([i, vp], i),
        "ekm_event_record": ([i, vp, vp], i), "ekm_event_sync": ([i, vp], i),
        "ekm_stream_wait_event": ([i, vp, vp], i),
        "ekm_event_elapsed_ms": ([i, vp, vp, C.POINTER(C.c_float)], i),
        "ekm_graph_begin": ([i, vp], i), "ekm_graph_end": ([i, vp, pvp], i),
        "ekm_graph_launch": ([i, vp, vp], i), "ekm_graph_destroy": ([i, vp], i),
        "ekm_set_tuning": ([i, i], i), "ekm_set_tuning_param": ([C.c_char_p, i], i), "ekm_prepare_tables": ([i], i), "ekm_get_tuning": ([C.POINTER(i), C.POINTER(i)], i),
        "ekm_synth_fill_f32": ([i, vp, vp, vp, vp, u64, sz, u64, u32, u64], i),
        "ekm_synth_fill_f64": ([i, vp, vp, vp, vp, u64, sz, u64, u32, u64], i),
        "ekm_synth_fill_given_p_f32": ([i, vp, vp, vp, vp, u64, sz, u64], i),
        "ekm_synth_fill_given_p_f64": ([i, vp, vp, vp, vp, u64, sz, u64], i),
        "ekm_synth_levels_f32": ([i, vp, vp, u32], i), "ekm_synth_levels_f64": ([i, vp, vp, u32], i),
        "ekm_stream_mix": ([i, vp, pvp, i, pvp, i, sz], i),
    }
    for tag, real in (("f32", C.c_float), ("f64", C.c_double)):
        sig[f"ekm_pressure_on_hybrid_levels_{tag}"] = (
            [i, vp, vp, vp, vp, sz, u32, vp, vp, i, real, vp, vp, vp, vp], i)
        sig[f"ekm_any_le_{tag}"] = ([i, vp, vp, sz, real, real, real, vp], i)
        sig[f"ekm_geopotential_on_hybrid_levels_{tag}"] = ([i, vp, vp, vp, vp, vp, vp, vp, sz, u32, i, real, i, vp], i)
        sig[f"ekm_geopotential_thickness_from_alpha_delta_{tag}"] = ([i, vp, vp, vp, vp, vp, sz, u32, vp], i)
    for name, (ins, outs, ints, has_eps) in OPS.items():
        for tag, real in (("f32", C.c_float), ("f64", C.c_double)):
            args = [i, vp] + [C.POINTER(Operand)] * len(ins) + [i] * len(ints)
            if has_eps:
                args.append(real)
            args += [vp] * len(outs) + [sz]
            sig[f"ekm_{name}_{tag}"] = (args, i)
    return sig


def declared_symbols():
    """Every symbol this binding expects the library (and the header) to provide."""
    return sorted(_signatures())


def lib():
    """The loaded library (loads and declares on first use)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                path = library_path()
                if not os.path.exists(path):
                    raise EkmLibraryError(
                        f"{path} not found: build it with `make -C earthkit-meteo_amd` "
                        "(or __graft_entry__.build()); ekm_hip has no CPU fallback")
                try:
                    handle = C.CDLL(path)
                except OSError as exc:
                    raise EkmLibraryError(f"cannot load {path}: {exc}") from exc
                # the ABI version first: a library of another round fails here with a sentence, not with an
                # AttributeError on whichever symbol happens to be missing
                try:
                    handle.ekm_abi_version.restype = C.c_int
                    have = handle.ekm_abi_version()
                except AttributeError:
                    have = None
                # EKM_THERMO_LIB_ANY_ABI=1 (A/B tools only, together with EKM_THERMO_LIB): load a library of another round
                # anyway; a symbol it lacks then fails at its first use
                any_abi = os.environ.get("EKM_THERMO_LIB_ANY_ABI") == "1" and os.environ.get("EKM_THERMO_LIB")
                if have != ABI_VERSION and not any_abi:
                    raise EkmLibraryError(f"{path} has ABI version {have if have is not None else '< 5 (no ekm_abi_version)'}, this "
                                          f"binding needs {ABI_VERSION} (include/ekm_thermo.h: EKM_ABI_VERSION): rebuild with "
                                          "`make -C earthkit-meteo_amd`")
                for name, (args, res) in _signatures().items():
                    try:
                        fn = getattr(handle, name)  # AttributeError here = header/library mismatch
                    except AttributeError:
                        if any_abi:
                            continue
                        raise
                    fn.argtypes = args
                    fn.restype = res
                _lib = handle
    return _lib


def check(rc):
    if rc < 0:
        msg = lib().ekm_last_error().decode(errors="replace")
        raise EkmError(f"libekm_thermo error {rc}: {msg}")
    return rc
