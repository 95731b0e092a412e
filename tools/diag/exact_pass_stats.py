#!/usr/bin/env python3
"""Diagnostic (GPU box): how the sequential-trajectory segments of a mixed population are solved -- passes of the cycle-parallel solver per
segment, and how many segments fall through to the single-thread pass.  usage: python tools/diag/exact_pass_stats.py [channels] [calls]"""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
pkg = importlib.import_module("sdr-j-fm_amd"); m = pkg.fmx
ch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 60
n = bench.BLOCK
dev = torch.device("cuda", 0)
kinds = bench.population_kinds("mixed", ch)
nblk = bench.FLAP_BLOCKS if "flap" in kinds else 1
iq = bench.synth_population(torch, ch, n, nblk, dev, kinds, seed=0)
f = pkg.Fmx(ch, max_block=n)
for pid, v in ((m.P_BANDWIDTH, 165000), (m.P_LF_CUTOFF, 15000), (m.P_DEEMPHASIS, 50), (m.P_VOLUME_DB, -6.0), (m.P_FM_MODE, 0)):
    f.set_param(pid, v)
pcm = torch.zeros((ch, n // 48 + 96, 2), dtype=torch.float32, device=dev)
st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
L = f.L
L.fmx_debug_phase_cycles.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_ulonglong)]
out = (C.c_ulonglong * 96)()
for k in range(calls):
    if k == calls - 10:
        L.fmx_debug_phase_cycles(f.h, 1, None)
    b = k % nblk
    f.process_device(iq[:, b * n:(b + 1) * n].data_ptr(), nblk * n, n, pcm.data_ptr(), pcm.shape[1], hip_stream=st.cuda_stream)
L.fmx_debug_phase_cycles(f.h, 0, out)
v = list(out)
segs = 10 * -(-(n // 12) // 1536) * ch
print("PSS integrator: steady segments %d (rounds %.2f), replayed by one thread %d" % (v[12], v[9] / max(v[12], 1), v[10]))
print("segments %d; cycle solver ran on %d (%.1f %%): passes per segment %.2f, not settled by it %d; plain replays %d; Newton rounds per Newton segment %.2f"
      % (segs, v[29], 100.0 * v[29] / segs, v[28] / max(v[29], 1), v[30], v[15], v[8] / max(v[11], 1)))
