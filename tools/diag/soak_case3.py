"""seed 2 channel 58 of the oscillator population (DIFF decoder, first call 1.26e-5): where in the call do library and oracle part? (diagnostic)"""
import importlib, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as ol
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
seed, sidx = 2, 3
n = 16384 * 3 * 5
rng = np.random.default_rng(seed)
streams = []
for s_ in range(5):
    x = ol.synth_iq(n, stereo=1 if s_ != 1 else 0, noiseSeed=100 * seed + s_, noiseSigma=0.002 * s_, rds=1, rdsLevel=0.05, rdsBitsSeed=seed * 10 + s_,
                    pilotLevel=float(rng.choice([0.10, 0.10, 0.05])))
    x[:, 0] += float(rng.choice([0.0, 0.007, -0.02])); x[:, 1] += float(rng.choice([0.0, -0.004, 0.015]))
    streams.append(x)
x = streams[sidx]
kw = {'inputFilterBw': 200000, 'attL': 0.9, 'attR': 1.0, 'loFrequency': 11000, 'dcRemove': 1, 'decoder': int(os.environ.get("DEC", "6")), 'fmMode': 2, 'soundSelector': 0, 'panorama': 60, 'deemphasis': 75,
      'volumeDb': 0.0, 'lfCutoff': 15000, 'autoMono': 1, 'squelchMode': 0, 'squelchValue': 66}
pid = dict(inputFilterBw=M.P_BANDWIDTH, attL=M.P_ATTENUATION_L, attR=M.P_ATTENUATION_R, loFrequency=M.P_LOCAL_OSCILLATOR, dcRemove=M.P_DC_REMOVE,
           decoder=M.P_FM_DECODER, fmMode=M.P_FM_MODE, soundSelector=M.P_SOUND_MODE, panorama=M.P_STEREO_PANORAMA, deemphasis=M.P_DEEMPHASIS,
           volumeDb=M.P_VOLUME_DB, lfCutoff=M.P_LF_CUTOFF, autoMono=M.P_AUTO_MONO, squelchMode=M.P_SQUELCH_MODE, squelchValue=M.P_SQUELCH_VALUE)
f = pkg.Fmx(1, max_block=n)
for k, v in kw.items(): f.set_param(pid[k], v, 0)
o = ol.OracleChain(taps=[ol.TAP_FM_IQ, ol.TAP_DEMOD], tap_seconds=0.3, **kw)
pg, po = f.process_host(x)[0], o.process(x)
m = min(len(pg), len(po))
d = pg[:m].astype(np.float64) - po[:m]
print("PCM frames %d, rms diff %.3e; rms per 256 frames:" % (m, np.sqrt((d ** 2).mean())))
print("  ", " ".join("%.1e" % np.sqrt((d[i:i + 256] ** 2).mean()) for i in range(0, m, 256)))
print("   PCM scale per 256 frames:", " ".join("%.1e" % np.abs(po[i:i + 256]).max() for i in range(0, m, 256)))
zg, zo = f.tap(M.TAP_FM_IQ, n // 12, 0), o.tap(ol.TAP_FM_IQ)[:n // 12]
dg, do = f.tap(M.TAP_DEMOD, n // 12, 0), o.tap(ol.TAP_DEMOD)[:n // 12]
dd = np.abs(dg.astype(np.float64) - do)
idx = np.argsort(dd)[::-1][:12]
print("largest demodulator differences (fm sample, lib, oracle, |z| oracle):")
for j in sorted(idx):
    print("   %6d  % .6f  % .6f   |z| %.3e  z lib (% .3e % .3e) oracle (% .3e % .3e)" % (j, dg[j], do[j], np.hypot(*zo[j]), zg[j, 0], zg[j, 1], zo[j, 0], zo[j, 1]))
print("demod rms diff after fm sample 6000: %.2e" % np.sqrt((dd[6000:] ** 2).mean()))
