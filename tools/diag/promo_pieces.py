"""promotion beside the demodulator pre-pass: which combination breaks (diagnostic)"""
import importlib, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as ol
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
def rms(x): return float(np.sqrt(np.mean(np.asarray(x, np.float64) ** 2)))
def run(nch, block, nb, restarts, pieces, at=8):
    iq = ol.synth_iq(nb * block)
    f = pkg.Fmx(nch, streams=1, stream_of_channel=[0] * nch, max_block=block)
    for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FM_DECODER, 2)): f.set_param(pid, v)
    if restarts: f.set_param(M.P_FILTER_RESTARTS, restarts)
    if pieces is not None: f.set_param(M.P_CALL_PIECES, pieces)
    o = ol.OracleChain(inputFilterBw=165000, decoder=2)
    waiting, res = False, []
    for b in range(nb):
        if b == at and restarts != 1: f.set_param(M.P_BANDWIDTH, 130000); waiting = True
        if waiting and f.filter_change_due() <= 0: o.configure(inputFilterBw=130000); waiting = False; res.append("applied@%d" % b)
        x = iq[b * block:(b + 1) * block]
        pg, po = f.process_host(x[None]), o.process(x)
        res.append("%.0e/%d/k%d" % (rms(pg[nch - 1] - po), f.last_call_pieces(), f.last_front_kernel()))
    print("channels %d block %d restarts %s pieces %s: %s" % (nch, block, restarts, pieces, " ".join(res)), flush=True)
    del f
run(1100, 16384 * 6, 8, 1, None)          # the machines from the start, PLL decoder, 1100 channels
run(70, 16384 * 6, 16, 0, None)           # promotion, no pieces (70 channels)
run(1100, 16384 * 6, 16, 0, 0)            # promotion, 1100 channels, pieces switched off
run(1100, 16384 * 6, 16, 0, None)         # promotion, 1100 channels, pieces
