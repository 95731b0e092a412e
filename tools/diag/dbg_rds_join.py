#!/usr/bin/env python3
"""Diagnostic: tests/test_gpu_round4.py::test_rds_decoders_switched_on_channel_by_channel call by call -- the 24 kS/s RDS baseband of every
channel against its oracle chain."""
import importlib, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as ol
import test_gpu_round4 as T
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
block, calls = 16384 * 15, 15
join = {0: 0, 1: 3, 2: 5}
iq = ol.synth_iq(block * calls, rds=1, rdsLevel=0.05, rdsBitsSeed=4242)
f = T._handle(pkg, 3, 1, block, 0, 2, [0, 0, 0])
chains = [ol.OracleChain(inputFilterBw=165000, rdsMode=0, taps=[ol.TAP_RDS_IQ, ol.TAP_PILOT, ol.TAP_DEMOD], tap_seconds=2.0) for _ in range(3)]
pos = [0, 0, 0]
for k in range(calls):
    for c in range(3):
        if join[c] == k:
            f.set_param(M.P_RDS_MODE, 2, c); chains[c].configure(rdsMode=2)
    x = iq[k * block:(k + 1) * block]
    f.process_host(x[None])
    msg = []
    for c in range(3):
        chains[c].process(x)
        if k >= join[c]:
            n = f.last_rds_samples(c)
            g = f.tap(M.TAP_RDS_IQ, n, c)
            o = chains[c].tap(ol.TAP_RDS_IQ)[pos[c]:pos[c] + n]; pos[c] += n
            msg.append("ch%d err %.2e (sig %.2e)" % (c, np.sqrt(np.mean((g - o).astype(np.float64) ** 2)), np.sqrt(np.mean(o.astype(np.float64) ** 2))))
    nt = f.last_fm_samples()
    pg = f.tap(M.TAP_PILOT_PHASE, nt, 1); po = chains[1].tap(ol.TAP_PILOT)[k * nt:(k + 1) * nt]
    d = np.abs(np.angle(np.exp(1j * (pg.astype(np.float64) - po))))
    print("call %2d: %s; pilot phase ch1 max diff %.2e" % (k, "; ".join(msg), d.max()))
