#!/usr/bin/env python3
"""One run of a plain batch (channel c on programme c % 4, device buffers) whose PCM and second-group size are saved: the GPU suite runs it with FMX_TAIL_SPLIT=0 and =1
(stages B / C as one channel group and as two, fmx_api.hip run_call_one) and compares bit for bit.  usage: groups_check.py OUT.npz CHANNELS CALLS BLOCK"""
import importlib, os, sys
import numpy as np
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as ol   # (signal generator only)
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
out, C, calls, block = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda", 0)
kw = [dict(), dict(noiseSeed=5, noiseSigma=0.05), dict(pilotLevel=0.02), dict(carrierAmp=0.2, noiseSeed=7, noiseSigma=0.02)]
base = np.stack([ol.synth_iq(block * calls, leftHz=300.0 + 370 * j, rightHz=500.0 + 530 * j, **kw[j]) for j in range(4)])
d_base = torch.from_numpy(base).to(dev)
cap = block // 48 + 96
d_pcm = torch.zeros((C, cap, 2), dtype=torch.float32, device=dev)
f = pkg.Fmx(C, max_block=block, device=0)
for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0)): f.set_param(pid, v)
pcm, groups = [], []
for i in range(calls):
    if i == 2: f.set_param(M.P_VOLUME_DB, -9.0)              # (a call with a gain correction is made as one group)
    reps = -(-C // 4)
    d_iq = d_base[:, i * block:(i + 1) * block].unsqueeze(0).expand(reps, 4, block, 2).reshape(reps * 4, block, 2)[:C].contiguous()
    torch.cuda.synchronize()
    frames = f.process_device(d_iq.data_ptr(), block, block, d_pcm.data_ptr(), cap)
    f.synchronize()
    pcm.append(d_pcm[:, :frames].cpu().numpy().copy()); groups.append(f.last_second_group())
    tw = d_pcm[:(C // 4) * 4, :frames].reshape(C // 4, 4, frames, 2)
    assert not bool((tw != tw[0:1]).any()), "twins differ in call %d" % i
np.savez(out, pcm=np.concatenate(pcm, axis=1), groups=np.array(groups))
print("second group per call:", groups)
