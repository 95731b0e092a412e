"""PCM checksum of a plain stereo batch (the headline's population, plus mono / unlocked / offset streams) for comparing two builds bit for bit:
FMX_LIB=... python tools/diag/batch_checksum.py"""
import hashlib, importlib, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as ol
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
nch, nst, n = 400, 5, 16384 * 3 * 4
iq = np.stack([ol.synth_iq(4 * n, stereo=0 if k == 1 else 1, leftHz=400.0 + 300 * k, rightHz=700.0 + 200 * k, noiseSeed=3 + k, noiseSigma=(0.0, 0.0, 0.01, 0.2, 0.0)[k],
                           offsetHz=(0.0, 0.0, 3000.0, 0.0, -20000.0)[k], pilotLevel=(0.1, 0.1, 0.1, 0.1, 0.05)[k]) for k in range(nst)])
f = pkg.Fmx(nch, streams=nst, stream_of_channel=[c % nst for c in range(nch)], max_block=n)
for p_, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0)): f.set_param(p_, v)
for c in range(nch):
    f.set_param(M.P_SOUND_MODE, (0, 1, 4, 6, 0)[(c // 5) % 5], c)
    f.set_param(M.P_FM_MODE, (0, 0, 1, 2)[(c // 25) % 4], c)
h = hashlib.md5()
for k in range(4):
    p = f.process_host(np.ascontiguousarray(iq[:, k * n:(k + 1) * n]))
    h.update(p.tobytes())
print("pcm md5", h.hexdigest(), "finite", bool(np.isfinite(p).all()), "rms", float(np.sqrt((p.astype(np.float64) ** 2).mean())))
