import importlib, sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import oracle_lib as ol
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
rng = np.random.default_rng(7)
block = 16384; nb = 6; n = block * nb
steps = rng.choice([0, 1, 1, 1, 2, 3], size=n); k = np.cumsum(steps) % 4
unit = np.array([[1, 0], [0, 1], [-1, 0], [0, -1]], np.float32)
iq = 0.5 * unit[k]
smooth = ol.synth_iq(n); use_smooth = (np.arange(n) // 3000) % 3 == 0; iq[use_smooth] = smooth[use_smooth]
zero = (np.arange(n) // 1777) % 7 == 3; iq[zero] = 0.0
f = pkg.Fmx(1, max_block=block, inputRate=192000)
for p, v in ((M.P_BANDWIDTH, 0), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FM_MODE, 0), (M.P_DC_REMOVE, 0), (M.P_FM_DECODER, 3)): f.set_param(p, v)
o = ol.OracleChain(inputRate=192000, inputFilterBw=0, dcRemove=0, decoder=3, taps=[ol.TAP_DEMOD, ol.TAP_FM_IQ], tap_seconds=3.0)
for b in range(nb):
    x = iq[b * block:(b + 1) * block]
    f.process_host(x); o.process(x)
    d_g, d_o = f.tap(M.TAP_DEMOD, block), o.tap(ol.TAP_DEMOD)[b * block:(b + 1) * block]
    z_g, z_o = f.tap(M.TAP_FM_IQ, block), o.tap(ol.TAP_FM_IQ)[b * block:(b + 1) * block]
    d = np.abs(d_g - d_o); i = int(d.argmax())
    big = np.nonzero(d > 1e-5)[0]
    print(b, "max %.2e at %d; n>1e-5: %d; fmIQ max diff %.2e" % (d.max(), i, len(big), np.abs(z_g - z_o).max()))
    for j in big[:6]:
        g = b * block + j
        print("   j", j, "gpu %.7f oracle %.7f" % (d_g[j], d_o[j]), "iq[j-1], iq[j]:", iq[g - 1], iq[g], "kind smooth/zero:", use_smooth[g - 1], use_smooth[g], zero[g - 1], zero[g])
