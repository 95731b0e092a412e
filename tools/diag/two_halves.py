#!/usr/bin/env python3
"""Experiment: the 4096-channel batch as TWO handles of 2048 channels whose calls are enqueued on two streams (the stages of one half overlapping the
other half's) against one handle of 4096 channels.  usage: two_halves.py [channels] [block] [steps]"""
import importlib, os, sys, time
import numpy as np
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as ol
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
block = int(sys.argv[2]) if len(sys.argv) > 2 else 230400
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
groups = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dev = torch.device("cuda", 0)
base = np.stack([ol.synth_iq(block, leftHz=300.0 + 370 * j, rightHz=500.0 + 530 * j) for j in range(4)])
d_base = torch.from_numpy(base).to(dev)
cap = block // 48 + 96


def mk(ch):
    f = pkg.Fmx(ch, max_block=block, device=0)
    for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FM_MODE, 0), (M.P_FM_DECODER, 3)): f.set_param(pid, v)
    iq = d_base.unsqueeze(0).expand(ch // 4, 4, block, 2).reshape(ch, block, 2).contiguous()
    pcm = torch.zeros((ch, cap, 2), dtype=torch.float32, device=dev)
    return f, iq, pcm


def run(handles, streams, n):
    for _ in range(n):
        for (f, iq, pcm), st in zip(handles, streams):
            f.process_device(iq.data_ptr(), block, block, pcm.data_ptr(), cap, hip_stream=st.cuda_stream)


for label, parts in (("one handle", 1), ("%d handles" % groups, groups)):
    handles = [mk(C // parts) for _ in range(parts)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    run(handles, streams, 48); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(handles, streams, steps); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s: %.4f ms per step of %d channels, %.1f GS/s" % (label, dt / steps * 1e3, C, C * block * steps / dt / 1e9), flush=True)
    del handles
