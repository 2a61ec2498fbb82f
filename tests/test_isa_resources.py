"""Register budget of the six-output fp32 pipeline kernels, read from the gfx950 assembly (hipcc cross-compiles without a
GPU): no instantiation may use scratch memory (three per-level instantiations spilled 16-48 B per lane
at the 96-VGPR cap), and the kernels the benchmark launches must keep five waves per SIMD (<= 96 VGPRs)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / "pipeline_f32.s"
    src = os.path.join(ROOT, "earthkit-meteo_amd", "csrc", "gen", "entries_pipeline_f32.hip")
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden", "-fno-slp-vectorize", "-S",
           "--cuda-device-only", src, "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    txt = out.read_text()
    found = {}
    for m in re.finditer(r"^(_ZN3ekm\S+):\s*;(?:(?!^_ZN3ekm).)*?; NumVgprs: (\d+)(?:(?!^_ZN3ekm).)*?; ScratchSize: (\d+)", txt, re.S | re.M):
        found[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    assert found
    return found


def test_no_pipeline_kernel_uses_scratch_memory(kernels):
    spilled = {k: v for k, v in kernels.items() if v[1] != 0}
    assert not spilled, spilled


def test_the_benchmarked_pipeline_kernels_keep_five_waves_per_simd(kernels):
    # map_fields<OpPipelineFull, float, 1> and the per-level instantiations (level vector, flat, hybrid; no level walk)
    want = [k for k in kernels if "14OpPipelineFullEf" in k and ("map_fieldsINS_14OpPipelineFullEfLi1E" in k or
                                                                  ("map_levels" in k and "ELb0EEEv" in k))]
    assert len(want) == 4, want
    for k in want:
        assert kernels[k][0] <= 96, (k, kernels[k])


# ---- every kernel of the BUILT library, from the code objects' own metadata (seconds, no compilation) ----------------
LLVM_BIN = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "earthkit-meteo_amd", "ekm_hip", "lib", "libekm_thermo.so")


@pytest.fixture(scope="module")
def built_kernels(tmp_path_factory):
    objdump, readelf = os.path.join(LLVM_BIN, "llvm-objdump"), os.path.join(LLVM_BIN, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf) and os.path.exists(LIB)):
        pytest.skip("llvm-objdump / llvm-readelf or the built library not available")
    work = tmp_path_factory.mktemp("codeobj")
    shutil.copy(LIB, work / "lib.so")
    subprocess.run([objdump, "--offloading", "lib.so"], cwd=work, check=True, capture_output=True, timeout=300)
    found = {}
    for f in sorted(os.listdir(work)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([readelf, "--notes", f], cwd=work, check=True, capture_output=True, text=True, timeout=300).stdout
        for block in notes.split("\n  - .agpr_count:")[1:]:
            get = lambda key: re.search(r"^    \." + key + r":\s*(\S+)", block, re.M).group(1)  # noqa: E731
            found[get("name")] = {"vgpr": int(get("vgpr_count")), "scratch": int(get("private_segment_fixed_size")),
                                  "lds": int(get("group_segment_fixed_size")), "threads": int(get("max_flat_workgroup_size"))}
    assert len(found) > 1000, len(found)
    return found


# a tree walk's kernels: the bisection entry points (T_BISECT = 0: OpTOnMa<M, 0>, OpWetBulbFromTd/Q<M, 0>, OpWbptFromTd/Q<M, 0>)
TREE = re.compile(r"(7OpTOnMa|15OpWetBulbFromTd|14OpWetBulbFromQ|12OpWbptFromTd|11OpWbptFromQ)ILi\dELi0EEE")


def test_no_kernel_of_the_built_library_runs_out_of_registers(built_kernels):
    """Scratch memory = registers spilled: nowhere, except 12 B per lane in full-field tree-walk kernels from three inputs
    (bolton35; the fp32 IFS walk at its 64-register cap), spilled once at kernel entry (the lane's first element index,
    reloaded for the ragged tail), outside the tile loop; and 12 B in ONE fp64 kernel at its 128-register cap (the bolton39
    walk from (t, q) on hybrid levels: two 8-byte values parked around the walk, four scratch instructions in 15,900)."""
    allowed = re.compile(r"map_fieldsINS_1[1245]Op(WetBulb|Wbpt)From(Q|Td)ILi[01]ELi0EEEfLi1E|map_levelsINS_14OpWetBulbFromQILi2ELi0EEEdLi2ELb0E")
    bad = {k: v for k, v in built_kernels.items() if v["scratch"] > (12 if allowed.search(k) else 0)}
    assert not bad, bad


def test_tree_walk_kernels_keep_their_occupancy(built_kernels):
    """One copy of the tree per workgroup.  fp32 IFS walk (round 5): 1024 threads around 64 KiB of 16-B records and <= 64
    registers (two workgroups = eight waves per SIMD); fp32 Bolton walks: 512 threads, 48 KiB, <= 80 registers (three
    workgroups = six waves per SIMD); fp64: 512 threads, 80 KiB, <= 128 registers (two workgroups = four waves per SIMD);
    no two-tile or level-walk instantiation."""
    tree = {k: v for k, v in built_kernels.items() if TREE.search(k) and ("map_fields" in k or "map_levels" in k or "map_bcast" in k)}
    assert len(tree) == 2 * 15 * 5, len(tree)   # map_fields, map_bcast, map_levels x (level vector, flat, hybrid)
    wide = 0
    for k, v in tree.items():
        f64 = re.search(r"EEEd(Li|EE)", k) is not None
        ifs32 = not f64 and re.search(r"ILi0ELi0EEE", k) is not None
        wide += ifs32
        assert v["threads"] == (1024 if ifs32 else 512), (k, v)
        assert v["lds"] == ((80 if f64 else 64 if ifs32 else 48) << 10), (k, v)
        if "map_bcast" not in k:
            assert v["vgpr"] <= (128 if f64 else 64 if ifs32 else 80), (k, v)
        assert not re.search(r"map_fields.*Li2EEEv", k), k                      # UNROLL = 2
        assert not re.search(r"map_levels.*Li\dELb1EEEv", k), k                 # WALK
    assert wide == 5 * 5, wide


def test_six_output_pipeline_kernels_of_the_built_library(built_kernels):
    want = [k for k in built_kernels if "14OpPipelineFullEf" in k and ("map_fieldsINS_14OpPipelineFullEfLi1E" in k or
                                                                        ("map_levels" in k and "ELb0EEEv" in k))]
    assert len(want) == 4, want
    for k in want:
        assert built_kernels[k]["vgpr"] <= 96 and built_kernels[k]["scratch"] == 0 and built_kernels[k]["lds"] == 0, (k, built_kernels[k])
