import importlib, sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import oracle_lib as ol
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
bw = 165000
block = 16384 * 3
switches = {4: dict(loFrequency=3000), 8: dict(loFrequency=0), 11: dict(dcRemove=0), 14: dict(dcRemove=1),
            17: dict(loFrequency=-2500), 19: dict(dcRemove=0), 21: dict(loFrequency=0), 23: dict(dcRemove=1)}
nb = 26
iq = ol.synth_iq(nb * block); iq[:, 0] += 0.007; iq[:, 1] -= 0.005
o = ol.OracleChain(inputFilterBw=bw, attL=0.9, attR=1.1, taps=[ol.TAP_FM_IQ], tap_seconds=4.0)
f = pkg.Fmx(1, max_block=block)
for p, v in ((M.P_BANDWIDTH, bw), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FM_MODE, 0), (M.P_ATTENUATION_L, 0.9), (M.P_ATTENUATION_R, 1.1)): f.set_param(p, v)
ids = dict(loFrequency=M.P_LOCAL_OSCILLATOR, dcRemove=M.P_DC_REMOVE)
nt = block // 12
for b in range(nb):
    for k, v in switches.get(b, {}).items():
        o.configure(**{k: v}); f.set_param(ids[k], v)
    x = iq[b * block:(b + 1) * block]
    po = o.process(x); pg = f.process_host(x)[0]
    z_g, z_o = f.tap(M.TAP_FM_IQ, nt), o.tap(ol.TAP_FM_IQ)[b * nt:(b + 1) * nt]
    d = np.abs(z_g - z_o).max(axis=1)
    if d.max() > 3e-6: print(b, switches.get(b), "first64 %.2e  all %.2e at %d | |z_g| max %.3f |z_o| max %.3f" % (d[:64].max(), d.max(), d.argmax(), np.abs(z_g).max(), np.abs(z_o).max()))
