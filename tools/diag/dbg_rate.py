import sys, importlib, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import oracle_lib as ol
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
def rms(a): return float(np.sqrt(np.mean(np.square(np.asarray(a, np.float64)))))
def run(rate, lo, dc, sw):
    decim = 1 if rate // 192000 <= 1 else 6 * ((rate // 6) // 192000)
    block = 16384 * 5; nblocks = 22; n = block * nblocks
    iq = ol.synth_iq(n, offsetHz=float(lo), dcI=0.004 if dc else 0.0, dcQ=-0.003 if dc else 0.0)
    f = pkg.Fmx(1, max_block=block, inputRate=rate)
    for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FM_MODE, 0)): f.set_param(pid, v)
    f.set_param(M.P_LOCAL_OSCILLATOR, lo)
    o = ol.OracleChain(inputRate=rate, inputFilterBw=165000, loFrequency=lo)
    errs = []
    for k, i in enumerate(range(0, n, block)):
        if sw and k == nblocks - 6:
            f.set_param(M.P_BANDWIDTH, 0); o.configure(inputFilterBw=0)
        a = f.process_host(iq[i:i + block])[0]; b = o.process(iq[i:i + block])
        errs.append(rms(a - b))
    print(rate, "lo", lo, "dc", dc, "switch", sw, " per-block rms:", " ".join("%.1e" % e for e in errs[-8:]))
for args in [(2304000, 0, 0, 1), (2304000, 30000, 0, 1), (2304000, 0, 1, 1), (2304000, 30000, 1, 1), (2400000, 30000, 1, 0)]:
    run(*args)
