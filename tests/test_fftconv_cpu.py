"""CPU check of the fast-convolution arithmetic the fused stage B filters the PSS signal with (csrc/fmx_fftconv.h): the header's
host half -- the very stage functions the device runs, all 256 threads in turn -- against a direct convolution in double
precision (tools/diag/fftconv_check.cpp, compiled for the host only)."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("ntaps", [295, 1, 64])
def test_fast_convolution_matches_direct(tmp_path, ntaps):
    cc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(cc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "fftconv_check")
    subprocess.check_call([cc, "-O2", "-std=c++17", "-w", "-o", exe, os.path.join(ROOT, "tools", "diag", "fftconv_check.cpp")])
    out = json.loads(subprocess.check_output([exe, str(ntaps)]).decode())
    # f32 transform of 2048 points: a few 1e-7 of the output scale
    assert out["conv_worst_abs"] <= 2e-6 * max(out["conv_scale"], 1.0), out
    assert out["roundtrip_worst_abs"] <= 3e-6, out
