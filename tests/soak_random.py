#!/usr/bin/env python3
"""Randomised soak of the batch path against the oracle (GPU box: `python tests/soak_random.py [first seed] [seeds] [channels] [plain 0|1|2] [setters 0|1] [ragged 0|1]`; a bounded run is in the suite,
tests/test_gpu_round6.py::test_randomised_soak_against_the_oracle).
Per seed: a handle of `channels` channels on five streams with settings drawn at random -- input filter width / off, IQ balance, local oscillator, DC
removal, all six decoders, the three squelch modes with random thresholds, fm mode, selector, panorama, de-emphasis, volume, audio filter, auto-mono, and an
RDS decoder (0 .. 3) switched on at a random call (some switched off again later) -- fed in calls of uneven length; every channel's PCM, and its RDS bit count
where a decoder ran, against an oracle chain taking the same settings and switches.
plain 1: no oscillator and the input filter on everywhere (the matrix-pipe kernel's population); 2: oscillators, the filter on everywhere (its complex-tap
variant's, from 256 channels).  setters 1: at the second or third call one channel in seven takes a setBandwidth or a setlfcutoff -- the handle is promoted to the
block machines (fmx_promote.hip) and the oracle chains take the setter at the call fmx_filter_change_due () names.  ragged 1: calls of any length (not
multiples of the decimation: the f32 kernel takes what the matrix-pipe kernels leave, history handed back and forth); RDS decoders on from the first call."""
import importlib, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as ol
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nch = int(sys.argv[3]) if len(sys.argv) > 3 else 70
plain = int(sys.argv[4]) if len(sys.argv) > 4 else 0        # 1: no local oscillator, the input filter on everywhere (what the matrix-pipe input filter takes)
setters = int(sys.argv[5]) if len(sys.argv) > 5 else 0
ragged = int(sys.argv[6]) if len(sys.argv) > 6 else 0
TOL = 1e-5
nstreams = 5
blocks = [16384 * 3 * k for k in (5, 4, 6, 5, 3, 5)]                    # (the oracle applies a switch at its next 16384-sample block: calls are whole blocks, their fm counts multiples of 8)
n = sum(blocks)
bad = edge = 0
for seed in range(seed0, seed0 + nseeds):
    rng = np.random.default_rng(seed)
    if ragged:
        # (a switch is applied by the oracle at its next 16384-sample block and by the library at the call: ragged calls take none)
        cuts = np.sort(np.random.default_rng(9000 + seed).integers(20000, n - 20000, size=6))
        blocks = [int(v) for v in np.diff(np.concatenate([[0], cuts, [n]])) if v > 0]
    streams = []
    for sidx in range(nstreams):
        x = ol.synth_iq(n, stereo=1 if sidx != 1 else 0, noiseSeed=100 * seed + sidx, noiseSigma=0.002 * sidx, rds=1, rdsLevel=0.05, rdsBitsSeed=seed * 10 + sidx,
                        pilotLevel=float(rng.choice([0.10, 0.10, 0.05])))
        x[:, 0] += float(rng.choice([0.0, 0.007, -0.02])); x[:, 1] += float(rng.choice([0.0, -0.004, 0.015]))
        streams.append(x)
    iq = np.stack(streams, axis=0)
    cfgs, rdsplan = [], []
    for c in range(nch):
        kw = dict(inputFilterBw=int(rng.choice([0, 165000, 165000, 130000, 200000])), attL=float(rng.choice([1.0, 0.9, 1.15])), attR=float(rng.choice([1.0, 1.1, 0.85])),
                  loFrequency=int(rng.choice([0, 0, 0, 2500, -4000])), dcRemove=int(rng.choice([1, 1, 1, 0])), decoder=int(rng.choice([1, 2, 3, 3, 4, 5, 6])),
                  fmMode=int(rng.choice([0, 0, 1, 2])), soundSelector=int(rng.choice([0, 1, 4])), panorama=int(rng.choice([100, 60, 140])),
                  deemphasis=int(rng.choice([50, 75])), volumeDb=float(rng.choice([-6.0, -10.5, 0.0])), lfCutoff=int(rng.choice([15000, 12000, 0])),
                  autoMono=int(rng.choice([1, 0])), squelchMode=int(rng.choice([0, 0, 0, 1, 2])), squelchValue=int(rng.integers(20, 80)))
        if plain: kw["loFrequency"] = 0 if plain == 1 else int(rng.choice([0, 2500, -4000, 11000, -37500])); kw["inputFilterBw"] = int(rng.choice([165000, 130000, 200000]))
        cfgs.append(kw)
        mode = int(rng.choice([0, 0, 1, 2, 2, 3]))
        rdsplan.append((mode, int(rng.integers(0, 4)), int(rng.choice([99, 99, 4, 5]))))            # (mode, on at call, off at call)
        if ragged: rdsplan[-1] = (mode, 0, 99)
    # (drawn from a generator of its own: the draws above stay what they were without setters)
    rng2 = np.random.default_rng(7000 + seed)
    ev_call = int(rng2.integers(1, 3))
    events = [None] * nch
    if setters:
        for c in range(nch):
            if rng2.integers(0, 7) == 0:
                events[c] = (dict(inputFilterBw=int(rng2.choice([0, 120000, 165000, 200000]))) if rng2.integers(0, 2) else dict(lfCutoff=int(rng2.choice([9000, 12000, 15000]))))
    only = [int(v) for v in os.environ.get("SOAK_ONLY", "").split(",") if v]           # (a diagnostic: these channels of the draw only, in a handle of their own)
    if only:
        cfgs = [cfgs[c] for c in only]; rdsplan = [rdsplan[c] for c in only]; events = [events[c] for c in only]; chan_stream = [c % nstreams for c in only]; nch = len(only)
    else:
        chan_stream = [c % nstreams for c in range(nch)]
    f = pkg.Fmx(nch, streams=nstreams, stream_of_channel=chan_stream, max_block=max(blocks))
    pid = dict(inputFilterBw=M.P_BANDWIDTH, attL=M.P_ATTENUATION_L, attR=M.P_ATTENUATION_R, loFrequency=M.P_LOCAL_OSCILLATOR, dcRemove=M.P_DC_REMOVE,
               decoder=M.P_FM_DECODER, fmMode=M.P_FM_MODE, soundSelector=M.P_SOUND_MODE, panorama=M.P_STEREO_PANORAMA, deemphasis=M.P_DEEMPHASIS,
               volumeDb=M.P_VOLUME_DB, lfCutoff=M.P_LF_CUTOFF, autoMono=M.P_AUTO_MONO, squelchMode=M.P_SQUELCH_MODE, squelchValue=M.P_SQUELCH_VALUE)
    if os.environ.get("SOAK_FRONT"): f.set_param(M.P_FRONT_KERNEL, int(os.environ["SOAK_FRONT"]))     # (a diagnostic: 1 = the f32 kernel everywhere)
    for c, kw in enumerate(cfgs):
        for k, v in kw.items(): f.set_param(pid[k], v, c)
    chains = [ol.OracleChain(rdsMode=0, **kw) for kw in cfgs]
    outs, ref, pos = [], [[] for _ in range(nch)], 0
    waiting, ev_applied = False, None
    for k, b in enumerate(blocks):
        if setters and k == ev_call:
            for c in range(nch):
                for key, v in (events[c] or {}).items(): f.set_param(pid[key], v, c)
            waiting = any(e is not None for e in events)
        if waiting and f.filter_change_due() <= 0:
            for c in range(nch):
                if events[c]: chains[c].configure(**events[c])
            waiting, ev_applied = False, k
        for c, (mode, on_at, off_at) in enumerate(rdsplan):
            if mode and on_at == k: f.set_param(M.P_RDS_MODE, mode, c); chains[c].configure(rdsMode=mode)
            if mode and off_at == k: f.set_param(M.P_RDS_MODE, 0, c); chains[c].configure(rdsMode=0)
        outs.append(f.process_host(np.ascontiguousarray(iq[:, pos:pos + b])))
        for c in range(nch): ref[c].append(chains[c].process(iq[chan_stream[c], pos:pos + b]))
        if only:
            for c in range(nch):
                m_ = min(outs[-1].shape[1], ref[c][-1].shape[0])
                mg, mo = f.meta(c), chains[c].meta()
                print("   call %d channel %d: rms %.3e  squelch %d / %d  locked %d / %d" % (k, only[c], float(np.sqrt(np.mean((outs[-1][c][:m_].astype(np.float64) - ref[c][-1][:m_]) ** 2))),
                      mg.squelch_active if hasattr(mg, "squelch_active") else -1, getattr(mo, "squelchActive", -1), mg.PilotPllLocked, mo.pilotLocked))
        pos += b
    pcm = np.concatenate(outs, axis=1)
    worst, wc = 0.0, -1
    for c in range(nch):
        po = np.concatenate(ref[c])
        m = min(pcm.shape[1], po.shape[0])
        # (the first call on its own: the input filter's latency ends 28 ms into the stream and the limiter decides "|z| <= 0.001" on the filter's start-up
        # transient, sample by sample.  Round 5 exempted it -- the library filtered first and balanced afterwards, one last bit away from the reference's
        # "balance, then filter" (fm-processor.cpp:462-470), and a sample on the knife's edge became a click.  Since round 6 every stage-A kernel
        # multiplies by the balance in front of its filter, as the reference does: the first call counts like every other.)
        f0 = outs[0].shape[1]
        e0 = float(np.sqrt(np.mean((pcm[c][:f0].astype(np.float64) - po[:f0]) ** 2)))
        e = float(np.sqrt(np.mean((pcm[c][f0:m].astype(np.float64) - po[f0:m]) ** 2)))
        if e0 > TOL: print("   seed %d channel %d: first call %.3e" % (seed, c, e0))
        if e > worst: worst, wc = e, c
        # (the PLL decoder, decoder 2: round 5 accepted its channels at 2e-4 -- pllC senses its phase through two quantised tables, and behind round 5's start-up
        # click (gone since round 6) one run in three hundred stayed 9e-5 off.  The reference against ITSELF, built with other compiler flags, differs by 1e-6 on this
        # decoder (tools/pll_decoder_self_difference.py, profiles/r06_pll_decoder_reference_vs_itself.txt): no case for a wider bound.  One tolerance for all.)
        tol = TOL
        self_e = self_e0 = -1.0
        ok = m > 0.95 * pcm.shape[1] and e <= tol and e0 <= tol and np.isfinite(pcm[c]).all()
        if rdsplan[c][0]:
            nb_g, nb_o = len(f.rds_bits(c, 8192)), len(chains[c].rds_bits())
            # (RDS_1 takes a bit at every top of its recovered clock, rds-decoder-1.cpp:124-142: while that clock pulls in, a top more or less is rounding)
            ok = ok and abs(nb_g - nb_o) <= (3 if rdsplan[c][0] == 1 else 0)
            if nb_g != nb_o: print("   seed %d channel %d: RDS bits %d against %d, plan %s" % (seed, c, nb_g, nb_o, rdsplan[c]))
        if not ok and np.isfinite(pcm[c]).all() and m > 0.95 * pcm.shape[1]:
            # Out of tolerance -- or is this signal on a KNIFE'S EDGE of the reference itself?  The chain has hard decisions -- the limiter's "|z| <= 0.001" and its
            # z / |z| on the input filter's start-up transient (fm-demodulator.cpp:120-127), the pilot lock detector's "> 0.07" (pilot-recover.cpp:62-80) -- on values
            # that carry the ROUNDING NOISE OF THE REFERENCE'S OWN f32 FFT FILTER: ~3e-7 of the block's scale on every output (fft-complex.cpp:69-71), a
            # third of the filter's first outputs behind its latency (tools/diag/soak_case2.py prints them).  The library's filters are exact convolutions: they
            # agree with the reference's to that noise, not bit for bit.  Seed 22 channel 212 (round 6; DIFF decoder): the demodulator's spike at the signal's onset,
            # -318.86 against -318.84, kicks the pilot PLL 3e-5 rad apart, the lock metric's last rise through 0.07 falls one pilot period later, and half a second on
            # the stereo decoder switches on ten samples apart -- one call at 3.6e-3.  None of that is a property of the SIGNAL.  The test: the ORACLE against ITSELF
            # with another realisation of that noise (fmo_config::testFilterNoise = 3e-7 on the input filter's output -- the level at which the oracle differs from
            # itself at the fm rate as it differs from the library's exact filter, 3.8e-7 against 4.2e-7 rms: tools/diag/soak_noise_level.py --; with the input
            # filter off, white noise 114 dB below the carrier on the input).  If some draw moves the oracle's PCM beyond the tolerance too, the channel is
            # reported and not counted; if the oracle does not care, the library is wrong and the channel counts.  Up to 48 draws, until one does: the
            # demodulators' onset spike (DIFF decoder: -300 on the first sample over the limiter's 0.001, size +-0.03 rms under this noise) puts the first call
            # of seed 2 channel 58 at 1.26e-5 -- 2 draws in 40 move the oracle further, the median draw 2.7e-6 (six draws, as first written, saw 4.4e-6).
            xs = iq[chan_stream[c]]
            self_e = self_e0 = 0.0
            first_only = e <= tol and not ragged         # (the first call alone is out: the draws need not go further; a ragged call's frame count is the library's own)
            for trial in range(48):
                if (e <= tol or self_e > tol) and (e0 <= tol or self_e0 > tol): break
                filt = cfgs[c]["inputFilterBw"] > 0
                xp = xs if filt else (xs + np.float32(1e-6) * np.random.default_rng(1000 * trial + c).standard_normal(xs.shape).astype(np.float32)).astype(np.float32)
                ch2 = ol.OracleChain(rdsMode=0, testFilterNoise=3e-7 if filt else 0.0, testNoiseSeed=trial + 1, **cfgs[c])
                r2, pos2 = [], 0
                for k2, b2 in enumerate(blocks[:1] if first_only else blocks):
                    mode, on_at, off_at = rdsplan[c]
                    if mode and on_at == k2: ch2.configure(rdsMode=mode)
                    if mode and off_at == k2: ch2.configure(rdsMode=0)
                    if events[c] and ev_applied == k2: ch2.configure(**events[c])
                    r2.append(ch2.process(xp[pos2:pos2 + b2])); pos2 += b2
                p2 = np.concatenate(r2)
                m2 = min(m, p2.shape[0])
                self_e0 = max(self_e0, float(np.sqrt(np.mean((p2[:f0] - po[:f0].astype(np.float64)) ** 2))))
                if not first_only: self_e = max(self_e, float(np.sqrt(np.mean((p2[f0:m2] - po[f0:m2].astype(np.float64)) ** 2))))
                del ch2
            pcm_ok = (e <= tol or self_e > tol) and (e0 <= tol or self_e0 > tol)
            if pcm_ok and (e > tol or e0 > tol):
                edge += 1
                print("   seed %d channel %d: on a knife's edge of the reference -- library against oracle %.2e (first call %.2e), the oracle against itself under its own filter's noise %.2e (%.2e); settings %s"
                      % (seed, c, e, e0, self_e, self_e0, cfgs[c]))
                ok = True if not rdsplan[c][0] else abs(nb_g - nb_o) <= (3 if rdsplan[c][0] == 1 else 0)
        if not ok:
            bad += 1
            print("   seed %d channel %d: PCM rms %.3e (first call %.3e; the oracle against itself under its filter's noise %.2e, %.2e) settings %s rds %s"
                  % (seed, c, e, e0, self_e, self_e0, cfgs[c], rdsplan[c]))
    print("seed %d: %d channels, worst PCM rms %.3e (channel %d), front kernel %d%s" % (seed, nch, worst, wc, f.last_front_kernel(),
          "; setters on %d channels at call %d, applied at call %s" % (sum(e is not None for e in events), ev_call, ev_applied) if setters else ""), flush=True)
    if setters and any(e is not None for e in events) and ev_applied is None: bad += 1; print("   seed %d: the setters were never applied" % seed)
    del f
print("channels out of tolerance: %d; on a knife's edge of the reference itself (reported, not counted): %d" % (bad, edge))
sys.exit(1 if bad else 0)
