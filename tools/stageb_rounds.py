#!/usr/bin/env python3
"""Diagnostics of the fused stage-B kernel on the GPU box: rounds of the two fixed-point iterations per segment (pilot PLL,
PSS integrator) and the number of replayed segments, summed over channels, for a short bench-like run.
usage: python tools/stageb_rounds.py [channels] [calls]"""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

pkg = importlib.import_module("sdr-j-fm_amd")
m = pkg.fmx
ch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n = bench.BLOCK
f = pkg.Fmx(ch, max_block=n)
for pid, v in ((m.P_BANDWIDTH, 165000), (m.P_LF_CUTOFF, 15000), (m.P_DEEMPHASIS, 50), (m.P_VOLUME_DB, -6.0), (m.P_FM_MODE, 0)):
    f.set_param(pid, v)
dev = torch.device("cuda", 0)
iq = bench.synth_device(torch, ch, n, dev)
pcm = torch.zeros((ch, n // 48 + 96, 2), dtype=torch.float32, device=dev)
st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
L = f.L
L.fmx_debug_phase_cycles.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 96)()
for k in range(calls):
    if k % 8 == 0:
        L.fmx_debug_phase_cycles(f.h, 1, None)
    f.process_device(iq.data_ptr(), n, n, pcm.data_ptr(), pcm.shape[1], hip_stream=st.cuda_stream)
    if k % 8 == 7:
        L.fmx_debug_phase_cycles(f.h, 1, out)
        v = list(out)
        print("calls %2d-%2d: PLL rounds/segment %.2f  integrator rounds/segment %.2f (of %d steady segments)  replayed %d of %d segments"
              % (k - 7, k, v[8] / max(v[11], 1), v[9] / max(v[12], 1), v[12], v[10], v[11]))
        names = ["disc", "afc", "pll", "lock", "| fft", "pss", "mix", "deemph", "tail"]
        tot = sum(v[16:25]) or 1
        print("   thread-0 cycles per phase: " + "  ".join("%s %.1f%%" % (nm, 100.0 * v[16 + i] / tot) for i, nm in enumerate(names)) + "  | per segment %.0f cycles" % (tot / max(v[11], 1)))
