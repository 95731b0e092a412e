#!/usr/bin/env python3
"""Diagnostic (GPU box): many channels on ONE stream whose pilot PLL runs on the exact trajectory in every segment (noise only, or a weak
pilot): every channel must produce the same PCM and taps bit for bit, call after call -- a race in the workgroup-wide steps of the
exact solvers would show as a channel that differs.  usage: python tools/diag/stress_exact.py [channels] [calls] [form]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import oracle_lib as ol
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
nch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 30
form = int(sys.argv[3]) if len(sys.argv) > 3 else 0
block = 16384 * 9
iq = ol.synth_iq(block * calls, pilotLevel=0.03, noiseSeed=3, noiseSigma=0.02)
f = pkg.Fmx(nch, streams=1, stream_of_channel=[0] * nch, max_block=block)
for p, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FM_MODE, 0), (M.P_STAGEB_FORM, form)):
    f.set_param(p, v)
bad_calls = 0
for k in range(calls):
    pcm = f.process_host(iq[k * block:(k + 1) * block])
    nt = f.last_fm_samples()
    diff = [c for c in range(1, nch) if not np.array_equal(pcm[c], pcm[0])]
    ph0 = f.tap(M.TAP_PILOT_PHASE, nt, 0)
    dph = [c for c in (diff[:3] if diff else []) if not np.array_equal(f.tap(M.TAP_PILOT_PHASE, nt, c), ph0)]
    if diff:
        bad_calls += 1
        c = diff[0]
        phc = f.tap(M.TAP_PILOT_PHASE, nt, c)
        w = np.flatnonzero(phc != ph0)
        print("call %d: %d channels differ from channel 0 (first %s); pilot phase differs in %s; channel %d: first phase difference at fm sample %s (segment %s), %d samples" %
              (k, len(diff), diff[:5], dph, c, w[:1], (w[:1] // 1536), len(w)))
print("channels %d, calls %d, form %d: calls with a differing channel: %d; exact segments %d, replays %d" % (nch, calls, form, bad_calls, f.pll_exact_segments(), f.pll_replays()))
