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
    # map_fields<OpPipelineFull, float, 1> and the aligned per-level instantiations (level vector, flat, hybrid)
    want = [k for k in kernels if "14OpPipelineFullEf" in k and ("map_fieldsINS_14OpPipelineFullEfLi1E" in k or
                                                                  ("map_levels" in k and "ELb1ELb0E" in k))]
    assert len(want) == 4, want
    for k in want:
        assert kernels[k][0] <= 96, (k, kernels[k])
