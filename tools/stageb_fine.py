#!/usr/bin/env python3
"""Fine-grained thread-0 cycle counts of the fused stage-B kernel (diagnostic build -DSB_FINE_TICKS, FMX_LIB=.../libfmx_ft.so):
cycles per segment between the SB_FT points.   usage: python tools/stageb_fine.py [channels] [calls]"""
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
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 24
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
names = {0: "top", 1: "wait zn", 2: "limiter", 3: "atan arm", 4: "atan gather(W)", 5: "atan finish", 6: "afc", 7: "pll seed", 8: "pll eval",
         9: "pll f64 scan", 10: "pll newton", 11: "pll (loop exit)", 12: "pll osc handoff", 13: "lock local", 14: "lock scan+run", 15: "lock flags scan",
         16: "taps store", 17: "sring addr+issue", 18: "sring wait(W)", 19: "fft convolve", 20: "er write+barrier", 21: "tags", 22: "integr scan",
         23: "mean scan", 24: "integr rounds", 25: "mean run", 26: "pss tail", 27: "mix sincos", 28: "matrix", 29: "fetch issue", 30: "deemph local",
         31: "deemph scan", 32: "deemph store", 33: "seq candidates", 34: "seq barrier", 35: "seq serial pass", 36: "seq fallback + read"}
if os.environ.get("PLL_SOLVER"):
    f.set_param(m.P_PLL_SOLVER, int(os.environ["PLL_SOLVER"]))
for k in range(calls):
    if k == calls - 8:
        L.fmx_debug_phase_cycles(f.h, 1, None)
    f.process_device(iq.data_ptr(), n, n, pcm.data_ptr(), pcm.shape[1], hip_stream=st.cuda_stream)
L.fmx_debug_phase_cycles(f.h, 1, out)
v = list(out)
segs = max((calls - (calls - 8)) * -(-(n // 12) // 1536) * ch, 1)
tot = sum(v[32:96])
print("channels %d: %d segments, %.0f cycles per segment (thread 0)" % (ch, segs, tot / segs))
for i in range(64):
    if v[32 + i]:
        print("  %2d %-18s %8.0f  %5.1f%%" % (i, names.get(i, "?"), v[32 + i] / segs, 100.0 * v[32 + i] / tot))
