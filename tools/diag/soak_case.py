"""one channel of the soak's seed 22 (stream 2) with variations of the failing settings: per-call PCM rms against the oracle (diagnostic)"""
import importlib, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as ol
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
seed = 22
blocks = [16384 * 3 * k for k in (5, 4, 6, 5, 3, 5)]
n = sum(blocks)
rng = np.random.default_rng(seed)
streams = []
for sidx in range(5):
    x = ol.synth_iq(n, stereo=1 if sidx != 1 else 0, noiseSeed=100 * seed + sidx, noiseSigma=0.002 * sidx, rds=1, rdsLevel=0.05, rdsBitsSeed=seed * 10 + sidx,
                    pilotLevel=float(rng.choice([0.10, 0.10, 0.05])))
    x[:, 0] += float(rng.choice([0.0, 0.007, -0.02])); x[:, 1] += float(rng.choice([0.0, -0.004, 0.015]))
    streams.append(x)
x = streams[2]
base = {'inputFilterBw': 130000, 'attL': 0.9, 'attR': 1.1, 'loFrequency': 0, 'dcRemove': 1, 'decoder': 6, 'fmMode': 0, 'soundSelector': 1, 'panorama': 100, 'deemphasis': 75,
        'volumeDb': -6.0, 'lfCutoff': 0, 'autoMono': 1, 'squelchMode': 2, 'squelchValue': 67}
pid = dict(inputFilterBw=M.P_BANDWIDTH, attL=M.P_ATTENUATION_L, attR=M.P_ATTENUATION_R, loFrequency=M.P_LOCAL_OSCILLATOR, dcRemove=M.P_DC_REMOVE,
           decoder=M.P_FM_DECODER, fmMode=M.P_FM_MODE, soundSelector=M.P_SOUND_MODE, panorama=M.P_STEREO_PANORAMA, deemphasis=M.P_DEEMPHASIS,
           volumeDb=M.P_VOLUME_DB, lfCutoff=M.P_LF_CUTOFF, autoMono=M.P_AUTO_MONO, squelchMode=M.P_SQUELCH_MODE, squelchValue=M.P_SQUELCH_VALUE)
variants = [("as drawn", {}), ("balance 1", dict(attL=1.0, attR=1.0))]
for name, ov in variants:
    kw = dict(base, **ov)
    f = pkg.Fmx(1, max_block=max(blocks))
    for k, v in kw.items(): f.set_param(pid[k], v, 0)
    o = ol.OracleChain(rdsMode=0, taps=[ol.TAP_DEMOD], tap_seconds=1.0, **kw)
    pos, res = 0, []
    for b in blocks:
        pg = f.process_host(x[pos:pos + b])[0]; po = o.process(x[pos:pos + b]); pos += b
        m = min(len(pg), len(po))
        d = pg[:m].astype(np.float64) - po[:m]
        res.append("%.1e@%d[lock %.5f]" % (np.sqrt((d ** 2).mean()), int(np.abs(d).max(axis=1).argmax()), f.meta(0).live_lock_strength))
    print("%-18s %s  squelch %d/%d" % (name, " ".join(res), f.meta(0).squelch_active, o.meta().squelchActive))
    if name == "as drawn":
        f.set_param(M.P_SCOPE_TAPS, 1)
        dg = f.tap(M.TAP_DEMOD, blocks[-1] // 12, 0); do = o.tap(ol.TAP_DEMOD)[-(blocks[-1] // 12):]
        dd = np.abs(dg - do)
        i = int(dd.argmax())
        print("   demodulator output of the last call: max |diff| %.3e at %d of %d; there: %s against %s; count > 1e-4: %d" % (dd.max(), i, len(dd), dg[max(0, i - 2):i + 3], do[max(0, i - 2):i + 3], int((dd > 1e-4).sum())))
    del f

# the oracle against itself: the input moved by k ulps at random
kw = dict(base)
ref = None
for k_ulp in (0, 1, 4, 16):
    rng2 = np.random.default_rng(5)
    xp = x.copy()
    for _ in range(k_ulp):
        up = rng2.integers(0, 2, x.shape).astype(bool)
        xp = np.nextafter(xp, np.where(up, np.float32(np.inf), np.float32(-np.inf)).astype(np.float32))
    o = ol.OracleChain(rdsMode=0, **kw)
    pos, outs = 0, []
    for b in blocks:
        outs.append(o.process(xp[pos:pos + b])); pos += b
    if ref is None: ref = outs
    print("oracle, input moved by %2d random ulps: per-call rms against the unmoved oracle: %s; lock strength snapshot %.5f locked %d" % (k_ulp, " ".join("%.1e" % np.sqrt(((a.astype(np.float64) - b) ** 2).mean()) for a, b in zip(outs, ref)), o.meta().pilotLockStrength, o.meta().pilotLocked))

# where does the stereo decoder switch on (first sample with a non-zero L-R in front of the matrix)?  library tap per call against the oracle's
for name, ov in (("as drawn", {}), ("balance 1", dict(attL=1.0, attR=1.0))):
    kw = dict(base, **ov)
    f = pkg.Fmx(1, max_block=max(blocks))
    for k, v in kw.items(): f.set_param(pid[k], v, 0)
    o = ol.OracleChain(rdsMode=0, taps=[ol.TAP_LRRAW, ol.TAP_DEMOD], tap_seconds=1.0, **kw)
    pos, lg, dg = 0, [], []
    for b in blocks:
        f.process_host(x[pos:pos + b]); o.process(x[pos:pos + b]); pos += b
        lg.append(f.tap(M.TAP_LR_RAW, b // 12, 0)); dg.append(f.tap(M.TAP_DEMOD, b // 12, 0))
    lg = np.concatenate(lg); dg = np.concatenate(dg); lo_ = o.tap(ol.TAP_LRRAW); do = o.tap(ol.TAP_DEMOD)
    nzg = np.nonzero(lg[:, 1])[0]; nzo = np.nonzero(lo_[:, 1])[0]
    print("%s: first non-zero L-R at fm sample: library %s, oracle %s (of %d); demodulator output max |diff| over the stream %.2e at %d" % (name, nzg[:1], nzo[:1], len(lg), np.abs(dg - do[:len(dg)]).max(), int(np.abs(dg - do[:len(dg)]).argmax())))
    # the lock metric of both demodulator outputs through the oracle's own pilot PLL
    import ctypes as C
    L = ol.oracle()
    for nm, dem in (("library's demod", dg), ("oracle's demod", do[:len(dg)])):
        d = np.ascontiguousarray(dem, np.float32); ph = np.zeros(len(d), np.float32); lk = np.zeros(len(d), np.uint8); stg = np.zeros(len(d), np.float32)
        try:
            L.fmo_pilot_run.restype = None
            om = np.float32(np.float64(np.float32(19000) / np.float32(192000)) * (2 * np.pi)); gn = np.float32(10 * (2 * np.pi) / 192000)
            L.fmo_pilot_run.argtypes = [C.c_int32, C.c_float, C.c_float, C.POINTER(C.c_float), C.c_long, C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_float)]
            p5 = np.ascontiguousarray(5 * d, np.float32)
            L.fmo_pilot_run(192000, float(om), float(gn), ol.fptr(p5), len(d), ol.fptr(ph), ol.u8ptr(lk), ol.fptr(stg))
            below = np.nonzero(stg <= 0.07)[0]
            print("   %s through the oracle's pilot PLL: last sample with the metric <= 0.07: %d; locked from %s; metric minimum behind sample 20000: %.5f" % (nm, below[-1] if len(below) else -1, np.nonzero(lk)[0][:1], stg[20000:].min()))
        except Exception as e:
            print("   (fmo_pilot_run: %s)" % e)
    del f
