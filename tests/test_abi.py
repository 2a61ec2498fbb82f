"""The C-ABI boundary without a GPU: the header, the library and the ctypes binding agree,
and the product fails loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ekm_thermo.h")


def header_symbols():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"EKM_API\s+[\w\s\*]+?\b(ekm_\w+)\s*\(", text)))


def test_header_declares_every_reference_function():
    from oracle import thermo_oracle as orc

    syms = set(header_symbols())
    for f in orc.ALL_FUNCTIONS:
        assert f"ekm_{f}_f32" in syms and f"ekm_{f}_f64" in syms, f
    for f in ("pipeline_svp_td_rh", "pipeline_full"):
        assert f"ekm_{f}_f32" in syms and f"ekm_{f}_f64" in syms
    text = open(HEADER).read()
    assert text.count("thermo/array/thermo.py:") >= 39  # every entry point cites the reference lines


def test_binding_matches_header():
    from ekm_hip import _ffi

    assert sorted(_ffi.declared_symbols()) == header_symbols()


def test_library_exports_every_declared_symbol():
    from ekm_hip import _ffi

    lib = C.CDLL(_ffi.library_path())  # loads here: libamdhip64 is present, only the GPU is not
    for s in header_symbols():
        assert hasattr(lib, s), f"{s} declared in include/ekm_thermo.h but not exported"
    assert _ffi.lib().ekm_version().decode().startswith("ekm_thermo")


def test_abi_version_of_header_binding_and_library_agree():
    """removed entry points and changed meanings must be visible as a version, checked at load time."""
    import re

    from ekm_hip import _ffi

    in_header = int(re.search(r"#define EKM_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert in_header == _ffi.ABI_VERSION == _ffi.lib().ekm_abi_version()


def test_a_library_of_another_abi_version_is_refused_with_a_sentence(tmp_path, monkeypatch):
    import subprocess

    from ekm_hip import _ffi

    src = tmp_path / "old.c"
    src.write_text("int ekm_abi_version(void) { return 4; }\n")
    old = tmp_path / "libold.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(old), str(src)], check=True)
    monkeypatch.setenv("EKM_THERMO_LIB", str(old))
    monkeypatch.setattr(_ffi, "_lib", None)
    with pytest.raises(_ffi.EkmLibraryError, match="ABI version 4.*needs 5"):
        _ffi.lib()
    none = tmp_path / "libnone.so"
    src.write_text("int something_else(void) { return 0; }\n")
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(none), str(src)], check=True)
    monkeypatch.setenv("EKM_THERMO_LIB", str(none))
    with pytest.raises(_ffi.EkmLibraryError, match="no ekm_abi_version"):
        _ffi.lib()


def test_no_torch_on_the_product_path():
    import subprocess
    import sys

    code = ("import sys; sys.path[:0]=[%r]; import ekm_hip; from ekm_hip import _ffi; _ffi.lib(); "
            "bad=[m for m in sys.modules if m.split('.')[0] in ('torch','cupy','triton')]; print(bad); "
            "sys.exit(1 if bad else 0)") % os.path.join(ROOT, "earthkit-meteo_amd")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def _gpu_present():
    from ekm_hip import _ffi

    return _ffi.lib().ekm_device_count() > 0


def test_fails_loudly_without_gpu():
    """No silent CPU path: without a device the product raises EkmError."""
    import ekm_hip

    if _gpu_present():
        pytest.skip("a GPU is present")
    from ekm_hip import _ffi

    assert _ffi.lib().ekm_init() == _ffi.EKM_ERR_NODEV
    assert _ffi.lib().ekm_last_error()
    with pytest.raises(ekm_hip.EkmError):
        ekm_hip.thermo.potential_temperature(np.array([280.0]), np.array([9e4]))


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    import subprocess
    import sys

    code = ("import sys, os; sys.path[:0]=[%r]; os.environ['EKM_THERMO_LIB']=%r; import ekm_hip, numpy as np\n"
            "try:\n    ekm_hip.thermo.potential_temperature(np.array([280.0]), np.array([9e4]))\n"
            "except ekm_hip.EkmLibraryError as e:\n    print('raised'); sys.exit(0)\nsys.exit(1)"
            ) % (os.path.join(ROOT, "earthkit-meteo_amd"), str(tmp_path / "nope.so"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and "raised" in r.stdout, r.stdout + r.stderr


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "earthkit-meteo_amd", "ekm_hip")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src and "_hosttwin" not in src, fn


def test_tuning_parameters_by_name_need_no_gpu():
    """ekm_set_tuning / ekm_set_tuning_param are host-side state: usable (and range-checked) without a device."""
    from ekm_hip import _ffi

    lib = _ffi.lib()
    assert lib.ekm_set_tuning_param(b"hybrid_band_kb", 4096) == 0 and lib.ekm_set_tuning_param(b"hybrid_band_kb", 8192) == 0
    assert lib.ekm_set_tuning_param(b"geo_chunk_levels", 1 << 20) == 0 and lib.ekm_set_tuning_param(b"lev_per_wg", 0) == 0
    assert lib.ekm_set_tuning_param(b"table_tiles", 16) == 0 and lib.ekm_set_tuning_param(b"table_tiles", 0) == 0  # 0 = by op
    assert lib.ekm_set_tuning_param(b"table_tiles", -1) == _ffi.EKM_ERR_ARG and b"table_tiles" in lib.ekm_last_error()
    assert lib.ekm_set_tuning_param(b"bisect_exact", 1) == 0 and lib.ekm_set_tuning_param(b"bisect_exact", 0) == 0
    assert lib.ekm_set_tuning_param(b"f64_plain", 2) == _ffi.EKM_ERR_ARG
    assert lib.ekm_set_tuning_param(b"no_such_parameter", 1) == _ffi.EKM_ERR_ARG
    assert b"unknown parameter" in lib.ekm_last_error()
    assert lib.ekm_set_tuning_param(None, 1) == _ffi.EKM_ERR_ARG
    t, u = C.c_int(), C.c_int()
    assert lib.ekm_get_tuning(C.byref(t), C.byref(u)) == 0 and t.value >= 1 and u.value in (1, 2)
    assert lib.ekm_set_tuning(0, 3) == _ffi.EKM_ERR_ARG


def test_wind_entry_point_cites_the_reference():
    text = open(HEADER).read()
    assert "ekm_w_from_omega_f32" in text and "wind/array/wind.py:192-222" in text
