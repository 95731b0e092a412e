#!/usr/bin/env python3
"""Diagnostic (GPU box): the fused stage-B layout against the chunked one on the same input, tap by tap and call by call.
usage: python tools/fused_vs_chunked.py [calls] [decoder 3..6]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("sdr-j-fm_amd")
M = pkg.fmx
import oracle_lib as ol  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 6
decoder = int(sys.argv[2]) if len(sys.argv) > 2 else 3
blocks = [16384 * 6] * calls if len(sys.argv) <= 3 else ([16384 * 3, 16384 * 5 + 12 * 77, 16384 * 2, 230400, 16384 * 7, 1200, 16384 * 9] * 8)[:calls]
block = max(blocks)
iq = ol.synth_iq(sum(blocks))
res = {}
for layout in ("chunked", "fused"):
    if layout == "chunked":
        os.environ["FMX_STAGE_B"] = "chunked"
    else:
        os.environ.pop("FMX_STAGE_B", None)
    f = pkg.Fmx(1, max_block=block)
    for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FM_MODE, 0)):
        f.set_param(pid, v)
    f.set_param(M.P_FM_DECODER, decoder)
    out = []
    import ctypes as C
    f.L.fmx_debug_phase_cycles.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_ulonglong)]
    dbg = (C.c_ulonglong * 96)()
    pos = 0
    for k in range(calls):
        f.L.fmx_debug_phase_cycles(f.h, 1, None)
        pcm = f.process_host(iq[pos:pos + blocks[k]]); pos += blocks[k]
        f.L.fmx_debug_phase_cycles(f.h, 1, dbg)
        if layout == 'fused': print('fused call %d: PLL rounds %d in %d segments, integrator rounds %d in %d steady segments, %d replayed; pilot phase %.7f lock metric %.6f' % (k, dbg[8], dbg[11], dbg[9], dbg[12], dbg[10], np.array([dbg[13]], np.uint32).view(np.float32)[0], np.array([dbg[14]], np.uint32).view(np.float32)[0]))
        nf = blocks[k] // 12
        m = f.meta(0)
        out.append((pcm[0], f.tap(M.TAP_DEMOD, nf, 0), f.tap(M.TAP_LR_RAW, nf, 0), m.PilotPllLocked, m.PilotPllLockStrength, m.PssState, m.PssPhaseShiftDegree))
    res[layout] = out
o = ol.OracleChain(taps=[ol.TAP_LRRAW], inputFilterBw=165000, fmMode=0, decoder=decoder, lfCutoff=15000, deemphasis=50, volumeDb=-6.0, tap_seconds=sum(blocks) / 2304000.0 + 0.1)
o.process(iq)
olr = o.tap(ol.TAP_LRRAW)
onz = np.nonzero(olr[:, 1])[0]
print("oracle: first fm sample with a nonzero L-R: %s" % onz[:1])
fmpos = np.cumsum([0] + [b for b in blocks])
rms = lambda x: float(np.sqrt(np.mean(np.square(x.astype(np.float64)))))
for k in range(calls):
    a, b = res["chunked"][k], res["fused"][k]
    print("call %d: pcm %.2e (sig %.2e)  demod %.2e  lr %.2e | chunked lock %d %.4f pss %d %.4f | fused lock %d %.4f pss %d %.4f"
          % (k, rms(a[0] - b[0]), rms(a[0]), rms(a[1] - b[1]), rms(a[2] - b[2]), a[3], a[4], a[5], a[6], b[3], b[4], b[5], b[6]))
    d = np.abs(a[2] - b[2]); bad = np.nonzero(d > 1e-3)[0]
    if len(bad):
        nz_a = np.nonzero(a[2][:, 1] if a[2].ndim == 2 else a[2])[0]; nz_b = np.nonzero(b[2][:, 1] if b[2].ndim == 2 else b[2])[0]
        j1 = fmpos[k + 1] // 12
        print("    absolute fm index of the first nonzero L-R: chunked %s fused %s" % (j1 - len(d) + nz_a[:1], j1 - len(d) + nz_b[:1]))
        print("    lr differs at %d rows, first %d last %d of %d; first nonzero diff row: chunked %s fused %s" % (len(bad), bad[0], bad[-1], len(d), nz_a[:1], nz_b[:1]))
