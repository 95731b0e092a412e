"""The N>1 path on CPU: two processes, gloo backend.  Each rank takes its block of channels
(sdr-j-fm_amd/shard.py), demodulates it (with the oracle standing in for the GPU on this GPU-less host),
rank 0 gathers the PCM and checks it against a single-process run over all channels, and the
max-over-ranks timing reduction used by bench.py is exercised."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import importlib, os, sys
    import numpy as np, torch
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import oracle_lib as ol
    shard = importlib.import_module("sdr-j-fm_amd").shard
    rank, world = shard.init("gloo")
    TOTAL, N = 5, 16384 * 8
    first, count = shard.shard_channels(TOTAL, world, rank)
    pcm = []
    for c in range(first, first + count):
        iq = ol.synth_iq(N, leftHz=300.0 + 37 * c, rightHz=500.0 + 53 * c)
        pcm.append(ol.OracleChain(inputFilterBw=0).process(iq))
    local = torch.from_numpy(np.stack(pcm)) if pcm else torch.zeros((0, N // 48 // 48 * 48, 2))
    full = shard.gather_pcm(local, TOTAL, dst=0)
    slow = shard.max_over_ranks(1.0 + rank)
    assert slow == float(world), slow
    if rank == 0:
        np.save(os.environ["FMX_TEST_OUT"], full.numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
''')


def test_shard_partition(fmx_amd):
    sh = fmx_amd.shard
    for total in (0, 1, 5, 4096, 16384):
        for world in (1, 2, 3, 8):
            spans = [sh.shard_channels(total, world, r) for r in range(world)]
            assert spans[0][0] == 0
            assert sum(c for _, c in spans) == total
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert sh.shard_channels(4096, 8, 3) == (1536, 512)
    with pytest.raises(ValueError):
        sh.shard_channels(4, 2, 2)


def test_two_rank_gloo_gather(tmp_path, ol):
    out = str(tmp_path / "pcm.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, FMX_TEST_OUT=out, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", OMP_NUM_THREADS="1")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e))
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = np.load(out)
    assert got.shape[0] == 5
    N = 16384 * 8
    for c in range(5):
        iq = ol.synth_iq(N, leftHz=300.0 + 37 * c, rightHz=500.0 + 53 * c)
        want = ol.OracleChain(inputFilterBw=0).process(iq)
        assert np.array_equal(got[c], want), c
