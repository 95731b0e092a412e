#!/usr/bin/env python3
"""Diagnostic (GPU box): tests/test_gpu_round3.py::test_creeping_pilots_at_batch_scale without the oracle, several times in one process with
other handles created and destroyed in between (allocations that reuse freed, non-zero memory): every channel c must equal channel c % 4
bit for bit in every call.  On a mismatch: which channels, and in which tap (fm-rate IQ / demodulator output / pilot phase / PCM) the
difference starts.  usage: python tools/diag/stress_creep_batch.py [runs] [channels]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import oracle_lib as ol
import test_gpu_round3 as T3
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
nst = 4
blocks = T3.BLOCKS * 8
n = sum(blocks)
iqs = np.stack([T3.creeping_pilot_iq(n, phase=-0.5 + 0.9 * k, period=1.6 - 0.13 * k) for k in range(nst)])
noise = ol.synth_iq(16384 * 40, carrierAmp=0.0, noiseSeed=9, noiseSigma=0.3)
total_bad = 0
for r in range(runs):
    # dirty the allocator: a handle with other sizes, filled with noise-driven state, then freed
    g = pkg.Fmx(777 + 131 * r, streams=1, stream_of_channel=[0] * (777 + 131 * r), max_block=16384 * 8)
    T3.gui_defaults(g); g.set_param(M.P_RDS_MODE, 2)
    for k in range(5):
        g.process_host(noise[k * 16384 * 8:(k + 1) * 16384 * 8])
    del g
    f = pkg.Fmx(nch, streams=nst, stream_of_channel=[c % nst for c in range(nch)], max_block=max(blocks))
    T3.gui_defaults(f)
    pos = 0
    for ci, b in enumerate(blocks):
        pg = f.process_host(iqs[:, pos:pos + b]); pos += b
        ref4 = pg[:nst]
        eq = (pg.reshape(nch // nst, nst, -1) == ref4.reshape(1, nst, -1)).all(axis=2)
        if not eq.all():
            total_bad += 1
            bad = np.argwhere(~eq)
            c = int(bad[0][0] * nst + bad[0][1])
            nt = f.last_fm_samples()
            msg = []
            for name, tap in (("fm IQ", M.TAP_FM_IQ), ("pre-resampler", M.TAP_PRE_RESAMPLER)):
                a, b2 = f.tap(tap, nt, c), f.tap(tap, nt, c % nst)
                w = np.flatnonzero((a != b2).reshape(nt, -1).any(axis=1))
                msg.append("%s: %s" % (name, "same" if len(w) == 0 else "first at %d (segment %d, thread %d), %d samples, max %.2e" % (w[0], w[0] // 1536, (w[0] % 1536) // 6, len(w), float(np.abs(a - b2).max()))))
            w = np.flatnonzero((pg[c] != pg[c % nst]).any(axis=1))
            print("run %d call %d (%d fm samples): %d channels differ, first %d vs %d; PCM first at frame %d (%d frames); %s" % (r, ci, nt, len(bad), c, c % nst, w[0], len(w), "; ".join(msg)))
            print("   differing channels:", [int(x[0] * nst + x[1]) for x in bad[:16]])
    print("run %d done: exact segments %d replays %d" % (r, f.pll_exact_segments(), f.pll_replays()))
    del f
print("calls with a mismatch:", total_bad)
