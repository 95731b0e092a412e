#!/usr/bin/env python3
"""Diagnostic (GPU box): the pilot phase (currentPilotPhase) of stage B against the oracle's, sample by sample over the last
call of a run, with the Newton rounds per segment.  usage: python tools/pll_vs_oracle.py [seconds] [noise sigma]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("sdr-j-fm_amd")
M = pkg.fmx
import oracle_lib as ol  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1.3
block = 16384 * 4
n = int(seconds * 2304000) // block * block
iq = ol.synth_iq(n)
if len(sys.argv) > 2:
    rng = np.random.default_rng(5)
    iq = (iq + float(sys.argv[2]) * rng.standard_normal(iq.shape)).astype(np.float32)
ch = ol.OracleChain(taps=[ol.TAP_PILOT, ol.TAP_LRRAW, ol.TAP_DEMOD], inputFilterBw=165000, fmMode=0, tap_seconds=seconds + 0.1)
ch.process(iq)
f = pkg.Fmx(1, max_block=block)
for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FM_MODE, 0)):
    f.set_param(pid, v)
if os.environ.get("PLL_SOLVER"):                           # 1 sequential, 2 Newton (a one-channel handle takes the sequential one by itself)
    f.set_param(M.P_PLL_SOLVER, int(os.environ["PLL_SOLVER"]))
f.L.fmx_debug_phase_cycles.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_ulonglong)]
dbg = (C.c_ulonglong * 96)()
f.L.fmx_debug_phase_cycles(f.h, 1, None)
for p in range(0, n, block):
    f.process_host(iq[p:p + block])
f.L.fmx_debug_phase_cycles(f.h, 1, dbg)
nt = block // 12
g = f.tap(M.TAP_PILOT_PHASE, nt).astype(np.float64)
o = ch.tap(ol.TAP_PILOT)[-nt:].astype(np.float64)
d = g - o
d = d - np.round(d / (2 * np.pi)) * 2 * np.pi
print("Newton rounds per segment %.2f (%d segments), integrator rounds %.2f, replays %d" % (dbg[8] / max(dbg[11], 1), dbg[11], dbg[9] / max(dbg[12], 1), f.pll_replays()))
print("pilot phase vs oracle over the last %d samples: mean %.3e  rms %.3e  max %.3e" % (nt, d.mean(), np.sqrt(np.mean(d * d)), np.abs(d).max()))
for k in range(0, nt, 1536):
    s = d[k:k + 1536]
    print("  samples %5d..: mean %+.2e rms %.2e" % (k, s.mean(), np.sqrt(np.mean(s * s))))
lr_g, lr_o = f.tap(M.TAP_LR_RAW, nt), ch.tap(ol.TAP_LRRAW)[-nt:]
print("lr rms diff %.3e" % np.sqrt(np.mean((lr_g.astype(np.float64) - lr_o) ** 2)))
