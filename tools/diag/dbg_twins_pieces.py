"""Twins of a 70-channel handle (PLL / AM decoder, squelches) with a frequency change (t), a volume change (v), a squelch slider (s) between calls, made whole or
in overlapping pieces.  usage: dbg_twins_pieces.py PIECES {t|v|s}+"""
import importlib, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as ol
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
pieces=int(sys.argv[1]); what=sys.argv[2]
block = 16384 * 14; nb = 6
iq = ol.synth_iq(nb * block, stereo=1, noiseSigma=0.002)
nch=70
f = pkg.Fmx(nch, streams=1, stream_of_channel=[0] * nch, max_block=block)
for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FM_MODE, 0), (M.P_FM_DECODER, 3)): f.set_param(pid, v)
f.set_param(M.P_CALL_PIECES, pieces)
for c in range(nch):
    k = c % 5
    if k == 0: f.set_param(M.P_FM_DECODER, 2, c)
    elif k == 1: f.set_param(M.P_FM_DECODER, 1, c); f.set_param(M.P_FM_MODE, 2, c)
    elif k == 2: f.set_param(M.P_SQUELCH_MODE, 2, c); f.set_param(M.P_SQUELCH_VALUE, 50, c)
    elif k == 3: f.set_param(M.P_SQUELCH_MODE, 1, c); f.set_param(M.P_SQUELCH_VALUE, 60, c)
for b in range(nb):
    if b == 3 and "t" in what: f.set_param(M.A_TRIGGER_FREQUENCY_CHANGE, 0)
    if b == 4 and "v" in what: f.set_param(M.P_VOLUME_DB, -10.5)
    if b == 4 and "s" in what:
        for c in range(3, nch, 5): f.set_param(M.P_SQUELCH_VALUE, 100, c)
    pcm = f.process_host(iq[None, b * block:(b + 1) * block])
    bad = [(c, int(np.flatnonzero((pcm[c] != pcm[c % 5]).any(axis=1))[0]), int((pcm[c] != pcm[c % 5]).any(axis=1).sum())) for c in range(5, nch) if not np.array_equal(pcm[c], pcm[c % 5])]
    print("call", b, "pieces", f.last_call_pieces(), "differing twins:", len(bad), bad[:6])
