"""seed 22 stream 2, the failing channel's settings: fm-rate IQ and demodulator output around the input filter's first outputs, library against oracle (diagnostic)"""
import importlib, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as ol
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
seed = 22
n = 16384 * 3 * 5
rng = np.random.default_rng(seed)
streams = []
for sidx in range(5):
    x = ol.synth_iq(16384 * 3 * 28, stereo=1 if sidx != 1 else 0, noiseSeed=100 * seed + sidx, noiseSigma=0.002 * sidx, rds=1, rdsLevel=0.05, rdsBitsSeed=seed * 10 + sidx,
                    pilotLevel=float(rng.choice([0.10, 0.10, 0.05])))
    x[:, 0] += float(rng.choice([0.0, 0.007, -0.02])); x[:, 1] += float(rng.choice([0.0, -0.004, 0.015]))
    streams.append(x)
x = streams[2][:n]
print("stream 2: dc", x[:, 0].mean(), x[:, 1].mean())
for name, kw, restarts in (("att 0.9/1.1 machines", dict(attL=0.9, attR=1.1), 1), ("att 0.9/1.1 folded", dict(attL=0.9, attR=1.1), 2), ("att 1 machines", {}, 1), ("att 0.9/1.1 no DC removal, machines", dict(attL=0.9, attR=1.1, dcRemove=0), 1)):
    kw = dict(dict(inputFilterBw=130000, decoder=6, dcRemove=1), **kw)
    f = pkg.Fmx(1, max_block=n)
    f.set_param(M.P_FILTER_RESTARTS, restarts)
    f.set_param(M.P_BANDWIDTH, kw["inputFilterBw"]); f.set_param(M.P_FM_DECODER, 6); f.set_param(M.P_DC_REMOVE, kw["dcRemove"])
    f.set_param(M.P_ATTENUATION_L, kw.get("attL", 1.0)); f.set_param(M.P_ATTENUATION_R, kw.get("attR", 1.0))
    o = ol.OracleChain(taps=[ol.TAP_FM_IQ, ol.TAP_DEMOD], tap_seconds=0.3, **kw)
    f.process_host(x); o.process(x)
    zg, zo = f.tap(M.TAP_FM_IQ, n // 12, 0), o.tap(ol.TAP_FM_IQ)[:n // 12]
    dg, do = f.tap(M.TAP_DEMOD, n // 12, 0), o.tap(ol.TAP_DEMOD)[:n // 12]
    print(name)
    for j in range(5438, 5450):
        print("   %d  z lib (% .3e % .3e) oracle (% .3e % .3e) |diff| %.1e   demod lib % .5f oracle % .5f" % (j, zg[j, 0], zg[j, 1], zo[j, 0], zo[j, 1], np.abs(zg[j] - zo[j]).max(), dg[j], do[j]))
    print("   max |z diff| over fm samples 5440..20479: %.2e (scale %.2f)" % (np.abs(zg[5440:] - zo[5440:]).max(), np.abs(zo).max()))
    del f
