"""A volume change on a call that starts inside a resampler block: every channel against channel 0 and the oracle, frame by frame (the diagnostic that
located gain_fix_kernel's read beyond its LDS array).  usage: dbg_gainfix.py CHANNELS"""
import importlib, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as ol
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
block = 16384 * 14; nb = 3
iq = ol.synth_iq(nb * block, stereo=1, noiseSigma=0.002)
nch=int(sys.argv[1])
f = pkg.Fmx(nch, streams=1, stream_of_channel=[0] * nch, max_block=block)
for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FM_MODE, 0), (M.P_FM_DECODER, 3)): f.set_param(pid, v)
o = ol.OracleChain(inputFilterBw=165000)
for b in range(nb):
    if b == 2: f.set_param(M.P_VOLUME_DB, -10.5); o.configure(volumeDb=-10.5)
    pcm = f.process_host(iq[None, b * block:(b + 1) * block])
    po = o.process(iq[b * block:(b + 1) * block])
    bad = [c for c in range(1, nch) if not np.array_equal(pcm[c], pcm[0])]
    print("call", b, "differing from channel 0:", len(bad), bad, [float(np.abs(pcm[c] - pcm[0]).max()) for c in bad[:3]])
    if bad:
        c = bad[0]
        d = np.flatnonzero((pcm[c] != pcm[0]).any(axis=1))
        print(" frames", d[:12], "ch0", pcm[0][d[:4]], "ch", c, pcm[c][d[:4]], "oracle", po[d[:4]])
        for cc in bad:
            print("   ch", cc, "err vs oracle per frame (first 12):", np.round(1e6 * (pcm[cc][:12, 0] - po[:12, 0])).astype(int))
        print("   ch 0 err vs oracle per frame (first 12):", np.round(1e6 * (pcm[0][:12, 0] - po[:12, 0])).astype(int))
        print(" err ch0 vs oracle first 40 frames", np.abs(pcm[0][:40] - po[:40]).max(), " ch", c, np.abs(pcm[c][:40] - po[:40]).max())
