"""How large is the rounding noise of the reference's own FFT input filter at the fm rate -- measured as oracle against library (whose filter is an exact
convolution, f32 FMAs) -- next to what the oracle's test hook (fmo_config::testFilterNoise) injects?  Calibrates the soak's knife-edge test (diagnostic)."""
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
for bw, lo in ((200000, 11000), (165000, 0), (130000, 0)):
    kw = dict(inputFilterBw=bw, decoder=6, loFrequency=lo, attL=0.9)
    f = pkg.Fmx(1, max_block=n)
    f.set_param(M.P_BANDWIDTH, bw); f.set_param(M.P_FM_DECODER, 6); f.set_param(M.P_LOCAL_OSCILLATOR, lo); f.set_param(M.P_ATTENUATION_L, 0.9)
    o = ol.OracleChain(taps=[ol.TAP_FM_IQ, ol.TAP_DEMOD], tap_seconds=0.3, **kw)
    pg, po = f.process_host(x), o.process(x)
    zg, zo = f.tap(M.TAP_FM_IQ, n // 12, 0), o.tap(ol.TAP_FM_IQ)[:n // 12]
    print("bw %d lo %d: scale %.3f" % (bw, lo, np.abs(zo).max()))
    def show(name, za, zb):
        d = (za.astype(np.float64) - zb)
        print("   %-46s fm-rate IQ rms diff: start-up (5440..5540) %.2e, steady (8000..20000) %.2e, max %.2e" % (name, np.sqrt((d[5440:5540] ** 2).mean()), np.sqrt((d[8000:20000] ** 2).mean()), np.abs(d[5440:]).max()))
    show("library against oracle", zg, zo)
    m_ = min(pg.shape[1], po.shape[0])
    print("   PCM rms library against oracle, this call: %.2e" % np.sqrt(((pg[0][:m_].astype(np.float64) - po[:m_]) ** 2).mean()))
    for lvl in (3e-7, 1e-6, 3e-6):
        o2 = ol.OracleChain(taps=[ol.TAP_FM_IQ], tap_seconds=0.3, testFilterNoise=lvl, testNoiseSeed=1, **kw)
        p2 = o2.process(x)
        show("oracle against itself, hook %.0e" % lvl, o2.tap(ol.TAP_FM_IQ)[:n // 12], zo)
        print("   PCM rms: %.2e" % np.sqrt(((p2[:m_].astype(np.float64) - po[:m_]) ** 2).mean()))
    del f
