import importlib, sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle_lib as ol
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
block = 16384 * 3
per_s = 2304000 / block
gap = int(1.6 * per_s)
order = [dict(inputFilterBw=0), dict(inputFilterBw=120000), dict(inputFilterBw=165000)]
switches = {(i + 1) * gap: d for i, d in enumerate(order)}
nb = (len(order) + 1) * gap
iq = ol.synth_iq(nb * block)
o = ol.OracleChain(inputFilterBw=165000)
f = pkg.Fmx(1, max_block=block)
for p, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FM_MODE, 0)): f.set_param(p, v)
for b in range(nb):
    for k, v in switches.get(b, {}).items():
        o.configure(**{k: v}); f.set_param(M.P_BANDWIDTH, v)
    x = iq[b * block:(b + 1) * block]
    po, pg = o.process(x), f.process_host(x)[0]
    a, m = f.meta(0), o.meta()
    e = float(np.sqrt(np.mean((pg - po) ** 2)))
    if b % 5 == 0 or b in switches or (b - 1) in switches or (b-2) in switches:
        print(b, switches.get(b, ""), "%.1e" % e, "gpu lock %d strength %.4f pss %d %.3f | oracle lock %d strength %.4f pss %d %.3f" % (a.live_pilot_locked, a.live_lock_strength, a.PssState, a.PssPhaseShiftDegree, m.pilotLocked, m.pilotLockStrength, m.pssState, m.pssPhaseShiftDegree))
