#!/usr/bin/env python3
"""Twins under runtime changes (GPU box).  A batch of channels in KINDS of equal settings -- the channels of a kind are twins on one stream -- fed in calls of
uneven length (most of them not whole resampler blocks) while random setters and actions are applied to whole kinds between calls.  Whatever a kind's settings
are, its channels run the same arithmetic on the same data: a channel that differs from its twin in PCM, RDS bits or metaData has read memory it should
not have (stale LDS, a neighbour's rows) or raced.  No oracle involved: thousands of setter / call combinations per minute.
usage: twins_setters.py [seed] [rounds] [channels] [kinds] [calls per round] [pieces -1|0|n] [streams] [format f32|s16|u8] [input rate]
streams > 1: the kinds listen to different streams (kind k to stream k % streams; twins share their stream).  A raw format goes through fmx_process_host_raw."""
import importlib, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as ol   # (signal generator only)
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nch = int(sys.argv[3]) if len(sys.argv) > 3 else 130
nk = int(sys.argv[4]) if len(sys.argv) > 4 else 5
calls = int(sys.argv[5]) if len(sys.argv) > 5 else 10
pieces = int(sys.argv[6]) if len(sys.argv) > 6 else -1
nst = int(sys.argv[7]) if len(sys.argv) > 7 else 1
fmtname = sys.argv[8] if len(sys.argv) > 8 else "f32"
rate = int(sys.argv[9]) if len(sys.argv) > 9 else 2304000      # (1152000: the reference decimates by 6; 192000: not at all)
MAXB = 16384 * 16
SETTERS = [
    (M.P_FM_MODE, [0, 1, 2]), (M.P_FM_DECODER, [1, 2, 3, 4, 5, 6]), (M.P_SOUND_MODE, [0, 1, 2, 3, 4, 5, 6]), (M.P_STEREO_PANORAMA, [0, 60, 100, 140, 200]),
    (M.P_SOUND_BALANCE, [-100, -30, 0, 30, 100]), (M.P_DEEMPHASIS, [50, 75, 1]), (M.P_VOLUME_DB, [-6.0, -10.5, 0.0, -20.0]), (M.P_LF_CUTOFF, [15000, 12000, 0]),
    (M.P_BANDWIDTH, [165000, 130000, 200000, 0]), (M.P_ATTENUATION_L, [1.0, 0.9, 1.15]), (M.P_ATTENUATION_R, [1.0, 1.1, 0.85]), (M.P_RDS_MODE, [0, 1, 2, 3]),
    (M.P_LOCAL_OSCILLATOR, [0, 2500, -4000]), (M.P_AUTO_MONO, [0, 1]), (M.P_PSS, [0, 1]), (M.P_DC_REMOVE, [0, 1]), (M.P_SQUELCH_MODE, [0, 1, 2]),
    (M.P_TEST_TONE, [0, 1]), (M.P_SQUELCH_VALUE, [0, 20, 50, 80, 100]), (M.A_TRIGGER_FREQUENCY_CHANGE, [0]), (M.A_RESTART_PSS, [0]), (M.A_RESET_RDS, [0]),
]
bad = 0
for rnd in range(rounds):
    rng = np.random.default_rng(1000 * seed + rnd)
    lens = [int(rng.choice([16384 * 14, 16384 * 3, 230400, 100001, 16384 * 16, 57600, 7777, 192 * 12 * 50 + 12 * int(rng.integers(0, 192))])) for _ in range(calls)]
    n = sum(lens)
    iq = ol.synth_iq(n, stereo=1, noiseSeed=seed * 100 + rnd, noiseSigma=0.003, rds=1, rdsLevel=0.05, rdsBitsSeed=seed + rnd, inputRate=rate)
    env = np.ones(n, np.float32)
    a, b = sorted(int(v) for v in rng.integers(0, n, 2))
    env[a:b] = 0.004                                               # (a fade: the squelches get something to decide)
    iq = (iq * env[:, None]).astype(np.float32)
    streams = [iq] + [ol.synth_iq(n, stereo=1, noiseSeed=seed * 100 + rnd + 7 * k, noiseSigma=0.002 * k, leftHz=400.0 + 300 * k, rightHz=700.0 + 200 * k, rds=1, rdsLevel=0.05,
                                  rdsBitsSeed=seed + rnd + k, dcI=0.004 * (k % 2), dcQ=-0.003 * k, inputRate=rate) for k in range(1, nst)]
    iqs = np.stack(streams, axis=0)
    if fmtname == "s16": raw, code = np.clip(np.round(iqs * 1500.0 + 9.0), -32768, 32767).astype(np.int16), M.IQ_S16
    elif fmtname == "u8": raw, code = np.clip(np.round(iqs * 100.0 + 127.4), 0, 255).astype(np.uint8), M.IQ_U8
    else: raw, code = iqs, M.IQ_F32
    kind = [c % nk for c in range(nch)]
    f = pkg.Fmx(nch, streams=nst, stream_of_channel=[(c % nk) % nst for c in range(nch)], max_block=MAXB, inputRate=rate)
    for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0)): f.set_param(pid, v)
    f.set_param(M.P_CALL_PIECES, pieces)
    if rng.integers(0, 2): f.set_param(M.P_STAGEB_FORM, int(rng.integers(0, 3)))
    log = []

    def apply(k, pid, v):
        for c in range(k, nch, nk): f.set_param(pid, v, c)
        log.append((k, pid, v))
    for k in range(nk):
        for pid, vals in SETTERS[:19]:
            if rng.random() < 0.35: apply(k, pid, vals[int(rng.integers(0, len(vals)))])
    pos = 0
    for i, ln in enumerate(lens):
        for _ in range(int(rng.integers(0, 4))):
            pid, vals = SETTERS[int(rng.integers(0, len(SETTERS)))]
            apply(int(rng.integers(0, nk)), pid, vals[int(rng.integers(0, len(vals)))])
        pcm = f.process_host(raw[:, pos:pos + ln]) if code == M.IQ_F32 else f.process_host_raw(raw[:, pos:pos + ln], code); pos += ln
        assert np.isfinite(pcm).all()
        diff = [c for c in range(nk, nch) if not np.array_equal(pcm[c], pcm[kind[c]])]
        metas = [f.meta(c) for c in range(nch)]
        mdiff = [c for c in range(nk, nch) if any(getattr(metas[c], fld) != getattr(metas[kind[c]], fld) for fld, _ in pkg.fmx.FmxMeta._fields_)]
        if diff or mdiff:
            bad += 1
            c = (diff or mdiff)[0]
            w = np.flatnonzero((pcm[c] != pcm[kind[c]]).any(axis=1)) if diff else []
            print("seed %d round %d call %d (len %d, pieces %d): %d channels differ in PCM %s, %d in metaData %s; channel %d frames %s max %.2e; kind's settings so far: %s"
                  % (seed, rnd, i, ln, f.last_call_pieces(), len(diff), diff[:8], len(mdiff), mdiff[:8], c, list(w[:6]), float(np.abs(pcm[c] - pcm[kind[c]]).max()) if diff else 0.0,
                     [(p, v) for k, p, v in log if k == kind[c]][-12:]), flush=True)
    bits = [f.rds_bits(c, 1 << 16) for c in range(nch)]
    # (the RDS block filters run two real channels per complex transform, fmx_rds.hip: a channel's baseband carries its pair partner's rounding, 1e-7 of it,
    # which decides a bit where the programme has faded into the noise.  A channel's twin for the bits is the one with the same place in a pair and the same
    # kind of partner: 2 * kinds further on when the kind count is odd)
    step = 2 * nk if nk % 2 else nk
    bdiff = [c for c in range(step, nch) if not np.array_equal(bits[c], bits[c % step])]
    if bdiff:
        bad += 1
        c = bdiff[0]; a_, b_ = bits[c], bits[c % step]; m_ = min(len(a_), len(b_))
        w = np.flatnonzero(a_[:m_] != b_[:m_])
        print("seed %d round %d: RDS bits of %d channels differ from their twins %s; channel %d: %d bits against %d, %d of the common ones differ (first at %s); its kind's settings: %s"
              % (seed, rnd, len(bdiff), bdiff[:8], c, len(a_), len(b_), len(w), list(w[:5]), [(p, v) for k, p, v in log if k == kind[c]]), flush=True)
    del f
print("rounds %d, calls %d each, mismatch: %d" % (rounds, calls, bad))
sys.exit(0 if bad == 0 else 1)
