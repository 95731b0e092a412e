#!/usr/bin/env python3
"""Two builds of libfmx.so on the same inputs, bit for bit: the demodulator pre-pass configurations (PLL / AM decoder, level and noise squelch, alone
and mixed in one handle), PCM and the demodulator tap.   tools/diag/ab_equal.py <libA.so> <libB.so>      (on the GPU box)"""
import os, subprocess, sys, tempfile
import numpy as np

CASES = {
    "pll":      dict(nch=70, dec=lambda c: 2, sq=lambda c: 0),
    "am":       dict(nch=70, dec=lambda c: 1, sq=lambda c: 0),
    "pll+nsq":  dict(nch=70, dec=lambda c: 2, sq=lambda c: 1 if c % 3 == 0 else 0),
    "nsq only": dict(nch=70, dec=lambda c: 3, sq=lambda c: 1 if c % 2 == 0 else 0),
    "lsq only": dict(nch=70, dec=lambda c: 3 + c % 4, sq=lambda c: 2 if c % 2 == 0 else 0),
    "pll+lsq":  dict(nch=70, dec=lambda c: 2, sq=lambda c: 2 if c % 2 else 0),
    "mixed":    dict(nch=130, dec=lambda c: 1 + c % 6, sq=lambda c: c % 3),
}

def child(out):
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
    import importlib
    fmx_amd = importlib.import_module("sdr-j-fm_amd")
    M = fmx_amd.fmx
    rng = np.random.default_rng(3)
    n = 16384 * 12
    t = np.arange(3 * n) / 2304000.0
    res = {}
    for name, c in CASES.items():
        nch = c["nch"]
        f = fmx_amd.Fmx(nch, max_block=n)
        iq = np.empty((nch, 3 * n, 2), np.float32)
        for ch in range(nch):
            ph = 2 * np.pi * (75000 / 1000.0) * np.sin(2 * np.pi * (700 + 13 * ch) * t) * 0.5
            a = 0.3 + 0.2 * np.sin(2 * np.pi * (3 + ch % 5) * t)
            z = a * np.exp(1j * ph) + 0.01 * (rng.standard_normal(3 * n) + 1j * rng.standard_normal(3 * n))
            if ch % 7 == 3: z[n // 2:n // 2 + 4000] = 0          # silence: the limiter's floor, the arc-tangent's corner arguments
            iq[ch, :, 0] = z.real; iq[ch, :, 1] = z.imag
        for ch in range(nch):
            f.set_param(M.P_FM_DECODER, c["dec"](ch), ch)
            f.set_param(M.P_SQUELCH_MODE, c["sq"](ch), ch)
            f.set_param(M.P_SQUELCH_VALUE, 40 + ch % 30, ch)
        pcm = [f.process_host(iq[:, k * n:(k + 1) * n]) for k in range(3)]
        res[name + "/pcm"] = np.concatenate(pcm, axis=1)
        res[name + "/demod"] = np.stack([f.tap(M.TAP_DEMOD, n // 12, ch) for ch in range(0, nch, 9)])
        del f
    np.savez(out, **res)

if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2]); sys.exit(0)
    outs = []
    for lib in sys.argv[1:3]:
        o = tempfile.mktemp(suffix=".npz")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", o], env=dict(os.environ, FMX_LIB=os.path.abspath(lib)))
        outs.append(np.load(o))
    bad = 0
    for k in outs[0].files:
        a, b = outs[0][k], outs[1][k]
        same = a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        print("%-16s %s  (finite %s, rms %.3e)" % (k, "identical" if same else "DIFFERENT: max |d| %.3e in %d values" % (np.nanmax(np.abs(a - b)), int(np.sum(a.view(np.uint32) != b.view(np.uint32)))), bool(np.isfinite(a).all()), float(np.sqrt(np.mean(a.astype(np.float64) ** 2)))))
        bad += 0 if same else 1
    sys.exit(1 if bad else 0)
