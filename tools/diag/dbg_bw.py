import importlib, sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import oracle_lib as ol
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
block = 16384 * 3
switches = {5: 165000, 10: 0, 14: 130000, 18: 200000, 22: 0}
bw0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nb = 27
iq = ol.synth_iq(nb * block); iq[:, 0] += 0.004; iq[:, 1] -= 0.003
o = ol.OracleChain(inputFilterBw=bw0, taps=[ol.TAP_FM_IQ], tap_seconds=4.0)
f = pkg.Fmx(1, max_block=block)
for p, v in ((M.P_BANDWIDTH, bw0), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FM_MODE, 0)): f.set_param(p, v)
nt = block // 12
for b in range(nb):
    if b in switches:
        o.configure(inputFilterBw=switches[b]); f.set_param(M.P_BANDWIDTH, switches[b])
    x = iq[b * block:(b + 1) * block]
    po = o.process(x); pg = f.process_host(x)[0]
    z_g, z_o = f.tap(M.TAP_FM_IQ, nt), o.tap(ol.TAP_FM_IQ)[b * nt:(b + 1) * nt]
    d = np.abs(z_g - z_o).max(axis=1)
    bad = np.nonzero(d > 1e-5)[0]
    print(b, switches.get(b, ""), "fm IQ max %.2e at %d (n>1e-5: %d, first %s last %s) | pcm rms %.2e | frames %d/%d" % (d.max(), d.argmax(), len(bad), bad[:1], bad[-1:], float(np.sqrt(np.mean((pg - po) ** 2))) if len(pg) == len(po) else -1, len(pg), len(po)))
