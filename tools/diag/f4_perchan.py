"""kernel 3 against kernel 1 on test_gpu_round5's handle: per-channel worst fm-rate IQ difference (diagnostic)"""
import importlib, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as ol
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
import test_gpu_round5 as t5
T = 1536
blocks = [T * 10, T * 150, T * 7]
iq = t5._streams(ol, sum(blocks))
outs = []
for kn in (1, 3):
    f = t5._handle(pkg, kn, max(blocks))
    outs.append(t5._run(f, iq, blocks))
    del f
a, b = outs
for c in range(t5.NCH):
    d = np.abs(a[1][c].astype(np.float64) - b[1][c]).max(axis=1)
    print("channel %d: fm IQ max diff %.2e at %d (scale %.2f), dc diff %.2e, pcm %.2e" % (c, d.max(), int(d.argmax()), np.abs(a[1][c]).max(), np.abs(a[2][:, c] - b[2][:, c]).max(), np.abs(a[0][c] - b[0][c]).max()))
# channels 0 and 2 of both kernels against the oracle's fm-rate IQ (stage B reads the ring 5440 fm samples back: compare what the taps hold)
n = sum(blocks)
for c, kw in ((0, {}), (2, dict(attL=0.9, attR=1.1))):
    o = ol.OracleChain(inputFilterBw=165000, taps=[ol.TAP_FM_IQ], tap_seconds=1.0, **kw)
    o.process(iq[c % t5.NST][: (n // 16384) * 16384])
    zo = o.tap(ol.TAP_FM_IQ)
    m = min(zo.shape[0], a[1][c].shape[0])
    for name, z in (("kernel 1", a[1][c]), ("kernel 3", b[1][c])):
        d = np.abs(z[:m].astype(np.float64) - zo[:m]).max(axis=1)
        print("channel %d %s against the oracle: max %.2e at %d, rms %.2e; first 5440+600: max %.2e" % (c, name, d.max(), int(d.argmax()), np.sqrt((d ** 2).mean()), d[:6040].max()))
