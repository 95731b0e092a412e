#!/usr/bin/env python3
"""A batch call made in overlapping pieces (FMX_P_CALL_PIECES, fmx_api.hip run_call) against the same pieces one after the other on one stream
(FMX_CALL_PIECES_SERIAL=1: must be equal bit for bit -- the same kernels on the same data, unless a stage of one piece races with another's)
and against the call made whole (the chain's invariance to how a stream is cut: a tolerance).  Channel c listens to programme c % 4 (one of
them noisy, one mistuned), the decoders / squelches per MODE.
usage: pieces_check.py CHANNELS CALLS BLOCK MODE [ROWS]      MODE: pll | am | nsq | lsq | mix"""
import importlib, os, sys
import numpy as np
import torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as ol   # (signal generator only)
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx


def main():
    C, calls, block, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    rows = int(sys.argv[5]) if len(sys.argv) > 5 else -1        # (-1: the library's choice)
    dev = torch.device("cuda", 0)
    kw = [dict(), dict(noiseSeed=5, noiseSigma=0.05), dict(offsetHz=20000.0), dict(carrierAmp=0.2, noiseSeed=7, noiseSigma=0.02)]
    base = np.stack([ol.synth_iq(block * calls, leftHz=300.0 + 370 * j, rightHz=500.0 + 530 * j, **kw[j]) for j in range(4)])
    d_base = torch.from_numpy(base).to(dev)
    cap = block // 48 + 96
    d_pcm = torch.zeros((C, cap, 2), dtype=torch.float32, device=dev)

    def run(pieces, serial):
        os.environ["FMX_CALL_PIECES_SERIAL"] = "1" if serial else "0"
        f = pkg.Fmx(C, max_block=block, device=0)
        for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0)): f.set_param(pid, v)
        f.set_param(M.P_CALL_PIECES, pieces)
        for c in range(C):
            m = mode if mode != "mix" else ("pll", "am", "nsq", "lsq", "none")[(c // 4) % 5]
            if m == "pll": f.set_param(M.P_FM_DECODER, 2, c)
            elif m == "am": f.set_param(M.P_FM_DECODER, 1, c)
            elif m == "nsq": f.set_param(M.P_SQUELCH_MODE, 1, c); f.set_param(M.P_SQUELCH_VALUE, 40, c)
            elif m == "lsq": f.set_param(M.P_SQUELCH_MODE, 2, c); f.set_param(M.P_SQUELCH_VALUE, 40, c)
        out, metas, npieces = [], [], []
        for i in range(calls):
            d_iq = d_base[:, i * block:(i + 1) * block].unsqueeze(0).expand(C // 4, 4, block, 2).reshape(C, block, 2).contiguous()
            torch.cuda.synchronize()
            frames = f.process_device(d_iq.data_ptr(), block, block, d_pcm.data_ptr(), cap)
            f.synchronize()
            out.append(d_pcm[:, :frames].cpu().numpy().copy()); npieces.append(f.last_call_pieces())
            metas.append([(f.meta(c).DcValIf, f.meta(c).squelch_active) for c in range(0, min(C, 40))])
        del f
        return np.concatenate(out, axis=1), metas, npieces
    whole, m0, p0 = run(0, False)
    over, m1, p1 = run(rows, False)
    ser, m2, p2 = run(rows, True)
    print("pieces per call:", p0, p1, p2)
    bad = int((over != ser).any(axis=(1, 2)).sum())
    d = float(np.abs(over - whole).max()); ref = float(np.abs(whole).max())
    print("overlapping vs whole: max |d| %.3g of %.3g; metaData snapshots equal (overlapping / serial / whole): %s %s" % (d, ref, m1 == m2, m1 == m0))
    print("channels differing between overlapping and serial pieces, mismatch: %d" % bad)
    return 0 if bad == 0 and max(p1) > 1 and p2 == p1 and d <= 2e-5 and m1 == m2 else 1


if __name__ == "__main__":
    sys.exit(main())
