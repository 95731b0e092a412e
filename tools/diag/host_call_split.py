#!/usr/bin/env python3
"""Latency of the single-receiver call (1 channel, 16384 samples, fmx_process_host) by input and by solver: where the time of
bench.py's host_call goes.  usage: python tools/diag/host_call_split.py"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
fmx_amd = importlib.import_module("sdr-j-fm_amd")
m = fmx_amd.fmx


def fm_stereo_iq(n, seed=0, fs=2304000.0):
    """a stereo station: L / R tones, 19 kHz pilot at 9 %, 75 kHz deviation"""
    t = np.arange(n) / fs
    l, r = np.sin(2 * np.pi * 1000 * t), np.sin(2 * np.pi * 1700 * t + 0.3)
    p = 2 * np.pi * 19000 * t
    mpx = 0.4 * (l + r) + 0.4 * (l - r) * np.sin(2 * p) + 0.09 * np.sin(p)
    ph = 2 * np.pi * 75000 * np.cumsum(mpx) / fs
    return np.stack([np.cos(ph), np.sin(ph)], 1).astype(np.float32) * 0.5


def run(iq_blocks, solver, restarts, calls=300):
    f = fmx_amd.Fmx(1, device=0, max_block=16384)
    for pid, v in ((m.P_BANDWIDTH, 165000), (m.P_LF_CUTOFF, 15000), (m.P_DEEMPHASIS, 50), (m.P_VOLUME_DB, -6.0)):
        f.set_param(pid, v)
    if solver is not None:
        f.set_param(m.P_PLL_SOLVER, solver)
    if restarts is not None:
        f.set_param(m.P_FILTER_RESTARTS, restarts)
    nb = len(iq_blocks)
    for k in range(40):
        f.process_host(iq_blocks[k % nb])
    ts = []
    for k in range(calls):
        b = iq_blocks[k % nb]
        t0 = time.perf_counter(); f.process_host(b); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    return "median %.4f p10 %.4f p90 %.4f p99 %.4f mean %.4f" % (np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90), np.percentile(ts, 99), ts.mean())


rng = np.random.default_rng(1)
noise = [(rng.random((16384, 2), dtype=np.float32) - 0.5)]
sig = fm_stereo_iq(16384 * 40, seed=3)
sig = [np.ascontiguousarray(sig[i * 16384:(i + 1) * 16384]) for i in range(40)]
for name, blocks in (("noise", noise), ("stereo station", sig)):
    for solver in (None, 2):
        for restarts in (None, 2):
            print("%-15s solver %-4s restarts %-4s : %s" % (name, solver, restarts, run(blocks, solver, restarts)), flush=True)
