"""The C++ multi-GPU host (sdr-j-fm_amd/host/multi_gpu_host.cpp; VERDICT r3 missing #5): one process, one thread and one RCCL rank per GPU,
one fmx handle per rank through the C ABI, RCCL only for the fan-out / gather / clock.  CPU: it compiles and links against libfmx and
librccl and refuses to run without a device.  GPU box (one device): a one-rank run -- communicator set-up, all-reduce clock, gather and
broadcast legs on library-produced buffers, and the self check (the gathered shard equals the local one bit for bit, channels of one
programme are identical)."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "sdr-j-fm_amd", "host", "multi_gpu_host.cpp")
LIBDIR = os.path.join(ROOT, "sdr-j-fm_amd", "lib")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists("/opt/rocm/include/rccl/rccl.h")), reason="needs hipcc and the RCCL headers")


def build(outdir):
    exe = os.path.join(str(outdir), "multi_gpu_host")
    subprocess.check_call([HIPCC, "-O2", "-std=c++17", "-Wall", SRC, "-I" + os.path.join(ROOT, "include"), "-L" + LIBDIR, "-lfmx", "-lrccl",
                           "-lpthread", "-Wl,-rpath," + LIBDIR, "-o", exe])
    return exe


def test_multi_gpu_host_builds_and_needs_a_device(tmp_path, fmx_amd):
    exe = build(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, "--gpus", "1"], capture_output=True, text=True)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_multi_gpu_host_one_rank(tmp_path, fmx_amd):
    exe = build(tmp_path)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe, "--gpus", "1", "--channels", "66", "--steps", "4", "--warmup", "44"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    j = json.loads(r.stdout.strip().splitlines()[-1])
    print("\n[C++ multi-GPU host, one rank]", j)
    assert j["self_check"] is True and j["rccl_ranks"] == 1 and j["n_gpus"] == 1 and j["channels_per_rank"] == [66]
    assert j["value"] > 1000 and j["gather_ms"] > 0
    r = subprocess.run([exe, "--gpus", "1", "--channels", "22", "--steps", "2", "--shared-streams", "2"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["broadcast_ms"] > 0 and j["value"] > 100
    r = subprocess.run([exe, "--gpus", "2", "--channels", "8"], capture_output=True, text=True, env=env, timeout=60)
    assert r.returncode == 2 and "device(s) visible" in r.stderr        # (one GPU here: the N > 1 launch is refused with the reason)
