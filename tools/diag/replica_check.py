"""diagnostic: replicas of two programmes through fmx_process_device; which (channel, call) pairs differ, and by how much
usage: python tools/diag/replica_check.py CHANNELS RDS(0/1) [calls] [block]"""
import sys, os, importlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch
import oracle_lib as ol
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
C, rds = int(sys.argv[1]), int(sys.argv[2])
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 8
block = int(sys.argv[4]) if len(sys.argv) > 4 else 230400
sigrds = int(sys.argv[5]) if len(sys.argv) > 5 else 1
base = [ol.synth_iq(block * calls, rds=1, rdsLevel=0.05, rdsBitsSeed=sd) if sigrds else ol.synth_iq(block * calls, leftHz=300.0 + 370 * k, rightHz=500.0 + 530 * k) for k, sd in enumerate((12345, 777))]
dev = torch.device("cuda", 0)
d_base = torch.from_numpy(np.stack(base)).to(dev)
cap = block // 48 + 96
d_pcm = torch.zeros((C, cap, 2), dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
f = pkg.Fmx(C, max_block=block, device=0)
f.set_param(M.P_SCOPE_TAPS, 1)      # (the scope taps, which a batch does not keep by default, are among the taps compared)
for p, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FM_MODE, 0), (M.P_FM_DECODER, 3)):
    f.set_param(p, v, -1)
if rds: f.set_param(M.P_RDS_MODE, 2)
for i in range(calls):
    mode = os.environ.get("DIAG_MODE", "fresh")
    src = d_base[:, i * block:(i + 1) * block].unsqueeze(0).expand(C // 2, 2, block, 2)
    if mode == "fresh":
        d_iq = src.reshape(C, block, 2).contiguous()
    else:
        if i == 0: d_keep = torch.empty((C, block, 2), dtype=torch.float32, device=dev)
        d_keep.view(C // 2, 2, block, 2).copy_(src); d_iq = d_keep
    if mode == "presync": torch.cuda.synchronize()
    print("   iq ptr", hex(d_iq.data_ptr()), flush=True)
    frames = f.process_device(d_iq.data_ptr(), block, block, d_pcm.data_ptr(), cap, hip_stream=stream)
    f.synchronize(); torch.cuda.synchronize()
    v = d_iq.view(C // 2, 2, block, 2)
    print("   input replicas identical after the call:", bool((v == v[0:1]).all().item()))
    out = d_pcm[:, :frames].reshape(C // 2, 2, frames, 2)
    d = (out - out[0:1]).abs()
    bad = d.amax(dim=(2, 3)) > 0            # [C/2, 2]
    nb = int(bad.sum().item())
    idx = bad.nonzero()[:6].tolist()
    first_frame = None
    if nb:
        k, j = idx[0]
        fr = (d[k, j].amax(dim=1) > 0).nonzero()
        first_frame = (int(fr[0]), int(fr[-1]), int(fr.numel()))
    if nb:
        k, j = idx[0]; cb = 2 * k + j
        nt = block // 12
        for name, tid in (("fm_iq", M.TAP_FM_IQ), ("demod", M.TAP_DEMOD), ("lr", M.TAP_LR_RAW)):
            a, b = f.tap(tid, nt, channel=cb), f.tap(tid, nt, channel=j)
            dd = np.abs(a - b).reshape(nt, -1).max(axis=1); w = np.nonzero(dd)[0]
            print("   tap", name, "ch", cb, "vs", j, "max", float(dd.max()), "first/last/count", (int(w[0]), int(w[-1]), len(w)) if len(w) else None, "nan", int(np.isnan(a).sum()), int(np.isnan(b).sum()))
        mg, mr = f.meta(cb), f.meta(j)
        print("   meta lock", mg.PilotPllLocked, mr.PilotPllLocked, "pss", mg.PssState, mr.PssState)
    print("call", i, "frames", frames, "differing", nb, "max", float(d.max()), "first", idx, "frames(first,last,count) of first bad", first_frame, flush=True)
if rds:
    bits = [f.rds_bits(c, 8192) for c in range(C)]
    badb = [c for c in range(2, C) if not np.array_equal(bits[c], bits[c % 2])]
    print("rds bits: len", len(bits[0]), len(bits[1]), "channels differing", len(badb), badb[:10])
