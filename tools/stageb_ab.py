#!/usr/bin/env python3
"""Diagnostic (GPU box): stage B as ONE kernel per call against the two-kernel schedule (FMX_STAGE_B=split) on the same
input: PCM, taps and metaData call by call, uneven call lengths, lock acquisition, several channel settings.
usage: python tools/stageb_ab.py            (spawns itself once per schedule)"""
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(out):
    pkg = importlib.import_module("sdr-j-fm_amd")
    M = pkg.fmx
    import oracle_lib as ol
    blocks = ([16384 * 3, 16384 * 5 + 12 * 77, 16384 * 2, 230400, 16384 * 7, 1200, 16384 * 9, 230400, 230400] * 3)[:22]
    iq = ol.synth_iq(sum(blocks))
    C = 6
    f = pkg.Fmx(C, max_block=max(blocks), streams=1)
    for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FM_MODE, 0)):
        f.set_param(pid, v)
    f.set_param(M.P_FM_DECODER, 5, channel=1)
    f.set_param(M.P_FM_DECODER, 6, channel=2)
    f.set_param(M.P_FM_MODE, 2, channel=3)
    f.set_param(M.P_PSS, 0, channel=4)
    f.set_param(M.P_AUTO_MONO, 0, channel=5)
    res = {}
    pos = 0
    for k, n in enumerate(blocks):
        pcm = f.process_host(iq[pos:pos + n]); pos += n
        res["pcm%d" % k] = pcm
        nf = n // 12
        for c in range(C):
            m = f.meta(c)
            res["meta%d_%d" % (k, c)] = np.array([m.PilotPllLocked, m.PilotPllLockStrength, m.PssState, m.PssPhaseShiftDegree, m.PssPhaseChange, m.DcValIf], np.float64)
        res["dem%d" % k] = f.tap(M.TAP_DEMOD, nf, 0)
        res["lr%d" % k] = f.tap(M.TAP_LR_RAW, nf, 5)
        res["lrnopss%d" % k] = f.tap(M.TAP_LR_RAW, nf, 4)
    res["replays"] = np.array([f.pll_replays()], np.int64)
    np.savez(out, **res)


if len(sys.argv) > 1:
    child(sys.argv[1])
    sys.exit(0)
outs = {}
for mode in ("split", "one"):
    env = dict(os.environ)
    if mode == "split":
        env["FMX_STAGE_B"] = "split"
    else:
        env.pop("FMX_STAGE_B", None)
    path = "/tmp/stageb_ab_%s.npz" % mode
    subprocess.check_call([sys.executable, os.path.abspath(__file__), path], env=env)
    outs[mode] = np.load(path)
a, b = outs["split"], outs["one"]
worst = 0.0
nbad = 0
for key in a.files:
    x, y = a[key].astype(np.float64), b[key].astype(np.float64)
    same = np.array_equal(a[key], b[key])
    d = float(np.max(np.abs(x - y))) if x.size else 0.0
    worst = max(worst, d)
    if not same:
        nbad += 1
        if d > 2e-6 or key.startswith("meta"):
            print("%-10s differs: max |d| %.3e (rms of a %.3e)" % (key, d, float(np.sqrt(np.mean(x * x))) if x.size else 0.0))
for k in range(22):
    def r(x): return float(np.sqrt(np.mean(np.square(x.astype(np.float64)))))
    print("call %2d: rms diff  lr(pss) %.2e  lr(no pss) %.2e  pcm %.2e | pss deg split %.6f one %.6f | lock strength %.6f %.6f" % (k, r(a["lr%d" % k] - b["lr%d" % k]), r(a["lrnopss%d" % k] - b["lrnopss%d" % k]), r(a["pcm%d" % k] - b["pcm%d" % k]), a["meta%d_0" % k][3], b["meta%d_0" % k][3], a["meta%d_0" % k][1], b["meta%d_0" % k][1]))
print("arrays compared %d, not bit-identical %d, worst |d| %.3e; PLL replays %s / %s" % (len(a.files), nbad, worst, a["replays"], b["replays"]))
