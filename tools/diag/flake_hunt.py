#!/usr/bin/env python3
"""Hunt for the rare cross-channel mismatch at batch scale (GPU box): 4096 channels, channel c listens to programme c % 4 (device buffers,
fmx_process_device as tests/test_gpu_parity.py::test_full_size_config4_device_path), handles created and destroyed in a loop, each run
through pilot acquisition; every call every channel's PCM is compared with channel c % 4's on the device.  On a mismatch: channel, call,
and the first tap (fm IQ / demodulator / pilot phase / L-R raw / pre-resampler) in which the channel differs from its twin.
usage: python tools/diag/flake_hunt.py [runs] [calls per run] [block] [channels] [rds 0|1] [stage-B form 0 auto|1 one kernel|2 two] [signals 0|1]
signals 1: the four programmes are a pilot flapping across the lock threshold, one creeping through it, noise only, and a station with
a DC offset and a local oscillator -- lock transitions, PSS replays, the guard's sequential segments and the per-sample DC / mix pass in every call."""
import importlib, os, sys, time
import numpy as np
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as ol   # (signal generator only)
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 8
block = int(sys.argv[3]) if len(sys.argv) > 3 else 230400
C = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
rds = int(sys.argv[5]) if len(sys.argv) > 5 else 0
form = int(sys.argv[6]) if len(sys.argv) > 6 else 0
hard = int(sys.argv[7]) if len(sys.argv) > 7 else 0
dev = torch.device("cuda", 0)
base = np.stack([ol.synth_iq(block * calls, leftHz=300.0 + 370 * j, rightHz=500.0 + 530 * j, **(dict(rds=1, rdsLevel=0.05, rdsBitsSeed=12345 + j) if rds else {})) for j in range(4)])
if hard:
    import test_gpu_round3 as T3
    n = block * calls
    t = np.arange(n) / 2304000.0
    pil = np.where(np.mod(t, 0.35) < 0.22, 0.06, 0.01) * (1 + 0.2 * np.sin(2 * np.pi * t / 0.31))
    lft, rgt = 0.5 * np.sin(2 * np.pi * 1000 * t), 0.5 * np.sin(2 * np.pi * 400 * t)
    p19 = 2 * np.pi * 19000 * t
    base[0] = T3.fm_modulate(0.45 * (lft + rgt) + pil * np.sin(p19) + 0.45 * (lft - rgt) * np.sin(2 * p19))
    base[1] = T3.creeping_pilot_iq(n, phase=-0.5, period=0.5)
    base[2] = ol.synth_iq(n, carrierAmp=0.0, noiseSeed=9, noiseSigma=0.3)
    base[3] = ol.synth_iq(n, offsetHz=200000.0, dcI=0.004, dcQ=-0.02)
d_base = torch.from_numpy(base).to(dev)
cap = block // 48 + 96
d_pcm = torch.zeros((C, cap, 2), dtype=torch.float32, device=dev)
side = torch.cuda.Stream(device=dev)
bad_total = 0
t0 = time.time()
for r in range(runs):
    f = pkg.Fmx(C, max_block=block, device=0)
    for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FM_MODE, 0), (M.P_FM_DECODER, 3)):
        f.set_param(pid, v)
    if rds: f.set_param(M.P_RDS_MODE, 2)
    if form: f.set_param(M.P_STAGEB_FORM, form)
    if hard:
        for c in range(3, C, 4): f.set_param(M.P_LOCAL_OSCILLATOR, 200000, c)
    for i in range(calls):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            d_iq = d_base[:, i * block:(i + 1) * block].unsqueeze(0).expand(C // 4, 4, block, 2).reshape(C, block, 2).contiguous()
        frames = f.process_device(d_iq.data_ptr(), block, block, d_pcm.data_ptr(), cap, hip_stream=side.cuda_stream)
        f.synchronize()
        out = d_pcm[:, :frames].reshape(C // 4, 4, frames, 2)
        ne = (out != out[0:1]).any(dim=3).any(dim=2)
        nb = int(ne.sum().item())
        if nb:
            bad_total += nb
            idx = torch.nonzero(ne).cpu().numpy()
            chans = [int(a * 4 + b) for a, b in idx[:16]]
            c = chans[0]
            nt = f.last_fm_samples()
            msg = []
            for name, tap in (("fm IQ", M.TAP_FM_IQ), ("pre-resampler", M.TAP_PRE_RESAMPLER)):
                a, b2 = f.tap(tap, nt, c), f.tap(tap, nt, c % 4)
                w = np.flatnonzero((a != b2).reshape(nt, -1).any(axis=1))
                msg.append("%s: %s" % (name, "same" if len(w) == 0 else "first at %d (segment %d, thread %d), last %d, %d samples, max %.2e" % (w[0], w[0] // 1536, (w[0] % 1536) // 6, w[-1], len(w), float(np.abs(a - b2).max()))))
            pa, pb = out[c // 4, c % 4].cpu().numpy(), out[0, c % 4].cpu().numpy()
            w = np.flatnonzero((pa != pb).any(axis=1))
            print("run %d call %d: %d channels differ %s; channel %d: PCM frames %d..%d (%d), max %.2e; %s; exact segs %d replays %d"
                  % (r, i, nb, chans, c, w[0], w[-1], len(w), float(np.abs(pa - pb).max()), "; ".join(msg), f.pll_exact_segments(), f.pll_replays()), flush=True)
    if rds:
        bits = [f.rds_bits(c, 8192) for c in range(C)]
        badb = [c for c in range(4, C) if not np.array_equal(bits[c], bits[c % 4])]
        if badb:
            bad_total += len(badb)
            print("run %d: RDS bits of %d channels differ from their twins: %s (lengths %s vs %d)" % (r, len(badb), badb[:12], [len(bits[c]) for c in badb[:4]], len(bits[0])), flush=True)
    del f
print("runs %d, calls %d each, %.0f s: (channel, call) pairs with a mismatch: %d" % (runs, calls, time.time() - t0, bad_total))
