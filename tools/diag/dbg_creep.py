#!/usr/bin/env python3
"""Diagnostic (GPU box): the creeping-pilot signal of tests/test_gpu_round3.py through both PLL solvers and the oracle; the
lock metric of pilot-recover.cpp:62-80 is rebuilt in f64 from each source's demodulator output and pilot phase taps, so that
the three can be compared sample by sample: where does the metric cross 0.07 for the last time before each lock, how far
apart are the sources there, and how large is the metric difference.  Writes gpurun_out/creep.npz for offline work.
usage: python tools/diag/dbg_creep.py [solver ...]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("sdr-j-fm_amd")
M = pkg.fmx
import oracle_lib as ol  # noqa: E402

BLOCKS = [16384 * 3, 16384 * 5, 16384 * 2, 16384 * 14, 16384 * 7, 16384, 16384 * 9] * 8
rate = 2304000
n = sum(BLOCKS)
t = np.arange(n) / rate
pil = 0.036 + 0.024 * np.sin(2 * np.pi * t / 1.6 - 0.5)
lft, rgt = 0.5 * np.sin(2 * np.pi * 1000 * t), 0.5 * np.sin(2 * np.pi * 400 * t)
p19 = 2 * np.pi * 19000 * t
mpx = 0.45 * (lft + rgt) + pil * np.sin(p19) + 0.45 * (lft - rgt) * np.sin(2 * p19)
ph = 2 * np.pi * 75000.0 / rate * np.cumsum(mpx)
iq = np.stack([0.5 * np.cos(ph), 0.5 * np.sin(ph)], axis=1).astype(np.float32)

N = 192000
OMEGA = float(np.float32(np.float32(np.float32(19000) / np.float32(192000)) * (2 * np.pi)))


def metric(dem, cur):
    """lock metric per sample from demod and currentPilotPhase (the value getPilotPhase returns), in f64"""
    phase = np.concatenate([[0.0], np.mod(cur[:-1].astype(np.float64) + OMEGA, 2 * np.pi)])
    idx = (phase * (N / (2 * np.pi))).astype(np.int64) % N
    osc = np.sin(2 * np.pi * idx / N)
    quad = np.diff(np.concatenate([[0.0], osc])) / OMEGA
    x = (1.0 / 3000.0) * (-quad * 5 * dem.astype(np.float64))
    from scipy.signal import lfilter
    return lfilter([1.0], [1.0, -(1.0 - 1.0 / 3000.0)], x)


def run(solver):
    f = pkg.Fmx(1, max_block=max(BLOCKS))
    for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FM_MODE, 0)):
        f.set_param(pid, v)
    f.set_param(M.P_PLL_SOLVER, solver)
    pcm, dem, cur, pos = [], [], [], 0
    for b in BLOCKS:
        pcm.append(f.process_host(iq[pos:pos + b])[0]); pos += b
        nt = f.last_fm_samples()
        dem.append(f.tap(M.TAP_DEMOD, nt)); cur.append(f.tap(M.TAP_PILOT_PHASE, nt))
    return np.concatenate(pcm), np.concatenate(dem), np.concatenate(cur), f.pll_replays()


o = ol.OracleChain(inputFilterBw=165000, taps=[ol.TAP_PILOT, ol.TAP_DEMOD], tap_seconds=n / rate + 0.2)
po, pos = [], 0
for b in BLOCKS:
    po.append(o.process(iq[pos:pos + b])); pos += b
po = np.concatenate(po)
nfm = n // 12
dem_o, cur_o = o.tap(ol.TAP_DEMOD)[:nfm], o.tap(ol.TAP_PILOT)[:nfm]
m_o = metric(dem_o, cur_o)
out = dict(po=po, dem_o=dem_o, cur_o=cur_o)
solvers = [int(a) for a in sys.argv[1:]] or [1, 2]
for s in solvers:
    pg, dem_g, cur_g, rep = run(s)
    m_g = metric(dem_g, cur_g)
    d = cur_g.astype(np.float64) - cur_o
    d -= np.round(d / (2 * np.pi)) * 2 * np.pi
    print(f"solver {s}: replays {rep}; pilot phase diff rms {np.sqrt(np.mean(d * d)):.2e} max {np.abs(d).max():.2e}; demod diff rms {np.sqrt(np.mean((dem_g - dem_o) ** 2)):.2e}; "
          f"metric diff rms {np.sqrt(np.mean((m_g - m_o) ** 2)):.2e} max {np.abs(m_g - m_o).max():.2e}")
    # the runs of the metric above the threshold, and the last sample below it in front of each run of > 96000
    for name, m in (("oracle", m_o), ("gpu", m_g)):
        below = np.flatnonzero(m <= 0.07)
        gaps = np.flatnonzero(np.diff(below) > 96000)
        print(f"  {name}: last samples below 0.07 in front of a lock:", [int(below[g]) for g in gaps], "first below behind it:", [int(below[g + 1]) for g in gaps])
    e = pg.astype(np.float64) - po
    fr = np.cumsum([0] + [b // 48 for b in BLOCKS])
    per = [float(np.sqrt(np.mean(e[a:b] ** 2))) for a, b in zip(fr[:-1], fr[1:])]
    print("  calls above 1e-5:", [(k, "%.1e" % v) for k, v in enumerate(per) if v > 1e-5])
    bad = np.flatnonzero(np.abs(e).max(axis=1) > 1e-3)
    if len(bad):
        print("  PCM frames off by > 1e-3: %d .. %d (%d frames) = fm samples %d .. %d" % (bad[0], bad[-1], len(bad), bad[0] * 4, bad[-1] * 4))
    out[f"pg{s}"] = pg; out[f"dem{s}"] = dem_g; out[f"cur{s}"] = cur_g
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "creep.npz"), **out)
