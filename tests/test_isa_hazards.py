"""CPU check of the gfx950 listings of every kernel source: no matrix instruction whose destination PARTLY overlaps its addend.
A multi-pass v_mfma reads the addend (srcC) row by row while it writes the destination row by row: the two must be the same registers or
disjoint ones.  hipcc (ROCm 7.2) emits the partly overlapping form for v_mfma_f32_16x16x32_f16 when register pressure makes it re-base an
accumulator -- round 5 met it in stage B's second kernel (the PSS low-pass on the matrix pipe, now tools/experiments/fmx_mfmaconv.h), where channels
with identical input came apart.  Every kernel source of the product is scanned; front4_kernel is the one that must hold matrix instructions."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sdr-j-fm_amd", "csrc")
# (source, extra flags as sdr-j-fm_amd/build.py compiles it)
SOURCES = [("fmx_front4.hip", []), ("fmx_front.hip", []), ("fmx_audio.hip", []), ("fmx_stageb.hip", ["-ffp-contract=off"])]
PAT = re.compile(r"v_mfma_\w+\s+([av])\[(\d+):(\d+)\],\s*\S+\s*\S+\s*([av])\[(\d+):(\d+)\]")


def partial_overlaps(text):
    bad = []
    for m in PAT.finditer(text):
        if m.group(1) != m.group(4):
            continue
        d0, d1, c0, c1 = int(m.group(2)), int(m.group(3)), int(m.group(5)), int(m.group(6))
        if (d0, d1) != (c0, c1) and not (d1 < c0 or c1 < d0):
            bad.append(m.group(0))
    return bad


def test_the_scan_finds_what_it_looks_for():
    assert partial_overlaps("v_mfma_f32_16x16x32_f16 v[18:21], v[46:49], v[78:81], v[20:23]")
    assert not partial_overlaps("v_mfma_f32_16x16x32_f16 v[18:21], v[46:49], v[78:81], v[18:21]")
    assert not partial_overlaps("v_mfma_f32_16x16x32_f16 v[18:21], v[46:49], v[78:81], v[22:25]")
    assert not partial_overlaps("v_mfma_f32_16x16x32_f16 v[18:21], v[46:49], v[78:81], 0")


@pytest.mark.parametrize("src,extra", SOURCES)
def test_no_matrix_instruction_with_a_partly_overlapping_addend(src, extra, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    out = tmp_path / (src + ".s")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only"] + extra + [os.path.join(CSRC, src), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    assert ("v_mfma" in text) == (src == "fmx_front4.hip")
    bad = partial_overlaps(text)
    assert not bad, bad[:5]
