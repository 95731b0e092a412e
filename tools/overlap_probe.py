#!/usr/bin/env python3
"""Does overlapping one batch's stage A / C with another batch's stage B pay?  Two handles of `half` channels each on two
streams, calls issued alternately (each call is asynchronous), against one handle of 2 * half channels.
    python tools/overlap_probe.py [half=2048]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
pkg = importlib.import_module("sdr-j-fm_amd"); m = pkg.fmx
half = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = bench.BLOCK
dev = torch.device("cuda", 0)

def make(ch):
    f = pkg.Fmx(ch, device=0, max_block=n)
    for pid, v in ((m.P_BANDWIDTH, 165000), (m.P_LF_CUTOFF, 15000), (m.P_DEEMPHASIS, 50), (m.P_VOLUME_DB, -6.0), (m.P_FM_MODE, 0)):
        f.set_param(pid, v)
    return f

def run(handles, iqs, pcms, streams, steps, warm=12):
    cap = n // 48 + 96
    for it in range(warm + steps):
        if it == warm:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        for f, iq, pcm, s in zip(handles, iqs, pcms, streams):
            f.process_device(iq.data_ptr(), n, n, pcm.data_ptr(), cap, hip_stream=s.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

cap = n // 48 + 96
iqs = [bench.synth_device(torch, half, n, dev, seed=1 + k) for k in range(parts)]
pcms = [torch.zeros((half, cap, 2), device=dev) for _ in range(parts)]
streams = [torch.cuda.Stream() for _ in range(parts)]
hs = [make(half) for _ in range(parts)]
ms2 = run(hs, iqs, pcms, streams, 10)
del hs
big = torch.cat(iqs); pbig = torch.zeros((parts * half, cap, 2), device=dev)
del iqs
f1 = make(parts * half)
ms1 = run([f1], [big], [pbig], [streams[0]], 10)
print("%d handles x %d channels, interleaved: %.3f ms per round -> %.1f GS/s;  one handle x %d: %.3f ms -> %.1f GS/s" %
      (parts, half, ms2, parts * half * n / ms2 / 1e6, parts * half, ms1, parts * half * n / ms1 / 1e6))
