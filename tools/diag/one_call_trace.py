#!/usr/bin/env python3
"""Diagnostic (GPU box, under rocprofv3 --kernel-trace --stats): the single-receiver call -- 1 channel, 16384 samples, fmx_process_host --
200 times, so that the per-kernel table shows what one drop-in block costs.  usage: rocprofv3 --kernel-trace --stats -d out -- python tools/diag/one_call_trace.py [restarts]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
pkg = importlib.import_module("sdr-j-fm_amd"); m = pkg.fmx
f = pkg.Fmx(1, max_block=16384)
if len(sys.argv) > 1:
    f.set_param(m.P_FILTER_RESTARTS, int(sys.argv[1]))
for pid, v in ((m.P_BANDWIDTH, 165000), (m.P_LF_CUTOFF, 15000), (m.P_DEEMPHASIS, 50), (m.P_VOLUME_DB, -6.0)):
    f.set_param(pid, v)
if len(sys.argv) > 2:
    f.set_param(m.P_DC_REMOVE, int(sys.argv[2]))
iq = (np.random.default_rng(1).random((16384, 2), dtype=np.float32) - 0.5)
for _ in range(200):
    f.process_host(iq)
