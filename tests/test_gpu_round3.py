"""Round-3 GPU tests: stage B as one kernel per call (fmx_stageb.hip) where its iterations can fail -- noise-only input, low CNR,
no pilot, a pilot that flaps across the lock threshold, a DC offset that crosses the RF limiter inside a call -- with both
solvers of the pilot PLL (FMX_P_PLL_SOLVER: 1 = sample by sample, 2 = what large batches run: Newton's method on the segment while
the pilot is comfortably in lock, sample by sample around every lock decision), against the oracle through the C ABI.  The bar everywhere: PCM <= 1e-5 RMS, lock / PSS flags equal call by call, the fallback counter reported."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PCM_RMS_TOL = 1e-5

import importlib  # noqa: E402

M = importlib.import_module("sdr-j-fm_amd").fmx


def rms(a):
    return float(np.sqrt(np.mean(np.square(np.asarray(a, np.float64)))))


def gui_defaults(f, bw=165000):
    f.set_param(M.P_BANDWIDTH, bw)
    f.set_param(M.P_LF_CUTOFF, 15000)
    f.set_param(M.P_DEEMPHASIS, 50)
    f.set_param(M.P_VOLUME_DB, -6.0)
    f.set_param(M.P_FM_MODE, 0)


def fm_modulate(mpx, dev=75000.0, rate=2304000, amp=0.5):
    """Complex baseband FM of a multiplex signal given at the input rate (float64 in, float32 [n, 2] out)."""
    ph = 2 * np.pi * dev / rate * np.cumsum(mpx)
    return np.stack([amp * np.cos(ph), amp * np.sin(ph)], axis=1).astype(np.float32)


def run_against_oracle(fmx_amd, ol, iq, blocks, solver, bw=165000, setup=None, oracle_kw=None, nch=1):
    """The same calls through the library and the oracle; returns PCM of both, the per-call flags of both, the handle."""
    f = fmx_amd.Fmx(nch, streams=1, stream_of_channel=[0] * nch, max_block=max(blocks))
    gui_defaults(f, bw)
    f.set_param(M.P_PLL_SOLVER, solver)
    if setup:
        setup(f)
    o = ol.OracleChain(inputFilterBw=bw, **(oracle_kw or {}))
    pg, po, fg, fo, live = [], [], [], [], []
    pos = 0
    for b in blocks:
        x = iq[pos:pos + b]; pos += b
        pg.append(f.process_host(x)[0]); po.append(o.process(x))
        assert pg[-1].shape == po[-1].shape
        a, m = f.meta(0), o.meta()
        fg.append((a.PilotPllLocked, a.PssState)); fo.append((m.pilotLocked, m.pssState))
        live.append(a.live_pilot_locked)
    f.live_locks = live
    f.per_call_rms = [rms(a - b) for a, b in zip(pg, po)]
    return np.concatenate(pg), np.concatenate(po), fg, fo, f


# whole reference blocks per call (the oracle pulls 16384 samples at a time, fm-processor.cpp:388), uneven, crossing segment boundaries
BLOCKS = [16384 * 3, 16384 * 5, 16384 * 2, 16384 * 14, 16384 * 7, 16384, 16384 * 9]


@pytest.mark.parametrize("solver", [1, 2])
def test_noise_only_channel(fmx_amd, ol, solver):
    """No station: complex white noise.  The discriminator output is full-scale noise (|5 demod gain| reaches 3e-3 rad per sample,
    well beyond what the PLL iteration was sized for), the pilot never locks, the output is mono noise."""
    blocks = BLOCKS * 3
    iq = ol.synth_iq(sum(blocks), carrierAmp=0.0, noiseSeed=5, noiseSigma=0.2)
    pg, po, fg, fo, f = run_against_oracle(fmx_amd, ol, iq, blocks, solver)
    e = rms(pg - po)
    print(f"\n[noise only, solver {solver}] PCM rms diff {e:.3e} (signal {rms(po):.3f}), PLL segments evaluated sequentially by the fail-safe: {f.pll_replays()}")
    assert fg == fo and all(x == (0, 0) for x in fo)
    assert e <= PCM_RMS_TOL
    if solver == 1:
        assert f.pll_replays() == 0                        # (the counter only counts Newton iterations that did not settle)


@pytest.mark.parametrize("solver", [1, 2])
def test_config2_at_17_db_cnr(fmx_amd, ol, solver):
    """configs[1] with wide-band noise at 17 dB CNR (SURVEY section 6's probe): carrier power 0.25, noise power 2 sigma^2."""
    blocks = BLOCKS * 5                                     # 2.9 s: lock at 0.5 s, PSS running behind it
    sigma = float(np.sqrt(0.25 / (2 * 10 ** 1.7)))
    iq = ol.synth_iq(sum(blocks), noiseSeed=11, noiseSigma=sigma)
    pg, po, fg, fo, f = run_against_oracle(fmx_amd, ol, iq, blocks, solver)
    e = rms(pg - po)
    print(f"\n[17 dB CNR, solver {solver}] PCM rms diff {e:.3e} (signal {rms(po):.3f}), flags {fo[-1]}, fail-safe segments {f.pll_replays()}")
    assert fg == fo
    assert fo[-1][0] == 1                                   # the pilot does lock in this noise
    assert e <= PCM_RMS_TOL


@pytest.mark.parametrize("solver", [1, 2])
def test_mono_station_without_pilot(fmx_amd, ol, solver):
    """A mono transmitter (no 19 kHz pilot) received in stereo mode: the PLL free-runs on programme material, never locks."""
    blocks = BLOCKS * 3
    iq = ol.synth_iq(sum(blocks), stereo=0)
    pg, po, fg, fo, f = run_against_oracle(fmx_amd, ol, iq, blocks, solver)
    e = rms(pg - po)
    print(f"\n[no pilot, solver {solver}] PCM rms diff {e:.3e} (signal {rms(po):.3f}), fail-safe segments {f.pll_replays()}")
    assert fg == fo and all(x[0] == 0 for x in fo)
    assert e <= PCM_RMS_TOL
    assert f.pll_replays() == 0


@pytest.mark.parametrize("solver", [1, 2])
def test_pilot_flapping_across_the_lock_threshold(fmx_amd, ol, solver):
    """The pilot level alternates between 6 % (0.70 s) and 1 % (0.25 s) of the deviation, with a 20 % ripple on top: the lock metric
    (threshold 0.07, 0.5 s of persistence, pilot-recover.cpp:62-80) goes up and down four times, each time switching the stereo decoder and the PSS
    (fm-processor.cpp:699-718) at one particular sample.  autoMono on (channel 0) and off (channel 1: L-R decoded throughout)."""
    rate = 2304000
    blocks = BLOCKS * 8                                     # 4.66 s
    n = sum(blocks)
    t = np.arange(n) / rate
    pil = np.where(np.mod(t, 0.95) < 0.70, 0.06, 0.01) * (1 + 0.2 * np.sin(2 * np.pi * t / 0.31))
    lft, rgt = 0.5 * np.sin(2 * np.pi * 1000 * t), 0.5 * np.sin(2 * np.pi * 400 * t)
    p19 = 2 * np.pi * 19000 * t
    mpx = 0.45 * (lft + rgt) + pil * np.sin(p19) + 0.45 * (lft - rgt) * np.sin(2 * p19)
    iq = fm_modulate(mpx)
    res = {}
    for auto_mono in (1, 0):
        def setup(f):
            f.set_param(M.P_AUTO_MONO, auto_mono)
        pg, po, fg, fo, f = run_against_oracle(fmx_amd, ol, iq, blocks, solver, setup=setup, oracle_kw=dict(autoMono=auto_mono))
        e = rms(pg - po)
        locks = f.live_locks
        ups = sum(1 for a, b in zip(locks, locks[1:]) if b > a)
        print(f"\n[flapping pilot, solver {solver}, autoMono {auto_mono}] PCM rms diff {e:.3e} (signal {rms(po):.3f}), lock acquired {ups} times, "
              f"flags differ in {sum(1 for a, b in zip(fg, fo) if a != b)} of {len(fo)} calls, fail-safe segments {f.pll_replays()}")
        assert ups >= 2 and 0 in locks[locks.index(1):]     # the lock does come and go
        assert fg == fo
        assert e <= PCM_RMS_TOL
        res[auto_mono] = e


def creeping_pilot_iq(n, phase=-0.5, period=1.6, rate=2304000):
    t = np.arange(n) / rate
    pil = 0.036 + 0.024 * np.sin(2 * np.pi * t / period + phase)
    lft, rgt = 0.5 * np.sin(2 * np.pi * 1000 * t), 0.5 * np.sin(2 * np.pi * 400 * t)
    p19 = 2 * np.pi * 19000 * t
    return fm_modulate(0.45 * (lft + rgt) + pil * np.sin(p19) + 0.45 * (lft - rgt) * np.sin(2 * p19))


@pytest.mark.parametrize("solver", [1, 2, 3])
def test_pilot_creeping_through_the_lock_threshold(fmx_amd, ol, solver):
    """The pilot level moves sinusoidally between 1.2 % and 6 % with a 1.6 s period, so the lock metric creeps through its threshold
    at ~1e-6 per sample and the sample at which it crosses depends on the seventh digit of the metric.  The decision itself is the
    reference's (pilot-recover.cpp:62-80); on the sequential trajectory the library takes it at the reference's sample (metric 2e-7
    from the oracle's).  Newton's method on a segment leaves the NCO phase ~2e-5 rad from the reference's and the metric 1.5e-6: the
    crossing then lands a pilot period later, and 0.5 s later (the persistence counter) the stereo decoder switches on ten samples late
    -- a click in one call (3e-4 RMS, 1.6e-2 peak in L-R).  Solver 2, what every large batch runs, therefore evaluates the segments
    around a lock decision sequentially (fmx_stageb.hip PLL_GUARD): no call above the tolerance.  Solver 3 (Newton
    always, a diagnostic) keeps the old bound: at most one such call per lock acquisition, none above 1e-3."""
    blocks = BLOCKS * 8
    iq = creeping_pilot_iq(sum(blocks))
    pg, po, fg, fo, f = run_against_oracle(fmx_amd, ol, iq, blocks, solver)
    locks = f.live_locks
    ups = sum(1 for a, b in zip(locks, locks[1:]) if b > a)
    over = [v for v in f.per_call_rms if v > PCM_RMS_TOL]
    print(f"\n[creeping pilot, solver {solver}] PCM rms diff {rms(pg - po):.3e}, lock acquired {ups} times, calls above 1e-5: {['%.1e' % v for v in over]}, "
          f"segments evaluated sequentially by the guard: {f.pll_exact_segments()} of {sum(-(-(b // 12) // 1536) for b in blocks)}")
    assert fg == fo and ups >= 2
    if solver != 3:
        assert not over
    else:
        assert len(over) <= ups and all(v <= 1e-3 for v in over)
    assert (f.pll_exact_segments() > 0) == (solver == 2)


def test_creeping_pilots_at_batch_scale(fmx_amd, ol):
    """the creeping-pilot case on the DEFAULT path of a large batch -- 4096 channels, four streams whose pilot levels
    creep through the threshold at different times (channel c listens to stream c % 4), the handle's automatic solver (above 64
    channels: Newton's method while in lock, sequential around the lock decisions).  Channels 0 .. 3 against their oracle chains call by
    call (no call above 1e-5, flags equal), every other channel bit-identical to channel c % 4."""
    nch, nst = 4096, 4
    blocks = BLOCKS * 8
    n = sum(blocks)
    iqs = np.stack([creeping_pilot_iq(n, phase=-0.5 + 0.9 * k, period=1.6 - 0.13 * k) for k in range(nst)])
    f = fmx_amd.Fmx(nch, streams=nst, stream_of_channel=[c % nst for c in range(nch)], max_block=max(blocks))
    gui_defaults(f)
    chains = [ol.OracleChain(inputFilterBw=165000) for _ in range(nst)]
    worst, ups, pos, prev = [0.0] * nst, [0] * nst, 0, [0] * nst
    for b in blocks:
        x = iqs[:, pos:pos + b]; pos += b
        pg = f.process_host(x)
        for k in range(nst):
            po = chains[k].process(x[k])
            assert pg[k].shape == po.shape
            worst[k] = max(worst[k], rms(pg[k] - po))
            a, m = f.meta(k), chains[k].meta()
            assert (a.PilotPllLocked, a.PssState) == (m.pilotLocked, m.pssState)
            ups[k] += 1 if a.live_pilot_locked > prev[k] else 0
            prev[k] = a.live_pilot_locked
        ref4 = pg[:nst]
        same = (pg.reshape(nch // nst, nst, -1) == ref4.reshape(1, nst, -1)).all(axis=2)
        if not same.all():                                   # (say where: a channel that differs from its twin is a race or a stray read)
            bad = np.argwhere(~same)
            c = int(bad[0][0] * nst + bad[0][1])
            nt = f.last_fm_samples()
            where = []
            for name, tap in (("fm IQ", M.TAP_FM_IQ), ("pre-resampler", M.TAP_PRE_RESAMPLER)):
                ta, tb = f.tap(tap, nt, c), f.tap(tap, nt, c % nst)
                wz = np.flatnonzero((ta != tb).reshape(nt, -1).any(axis=1))
                where.append("%s %s" % (name, "same" if len(wz) == 0 else "first at fm sample %d of %d (%d differ, max %.2e)" % (wz[0], nt, len(wz), float(np.abs(ta - tb).max()))))
            wf = np.flatnonzero((pg[c] != pg[c % nst]).any(axis=1))
            raise AssertionError("call %d: %d channels differ from their twins, e.g. %d vs %d: PCM frames %d..%d; %s; channels %s"
                                 % (len(worst) and blocks.index(b), len(bad), c, c % nst, wf[0], wf[-1], "; ".join(where), [int(x[0] * nst + x[1]) for x in bad[:12]]))
    ex = f.pll_exact_segments()
    print(f"\n[creeping pilots, {nch} channels] worst call per stream: {' '.join('%.1e' % v for v in worst)}; lock acquisitions {ups}; "
          f"guard: {ex / nch:.0f} of {sum(-(-(b // 12) // 1536) for b in blocks)} segments per channel sequential; fail-safe replays {f.pll_replays()}")
    assert max(worst) <= PCM_RMS_TOL and sum(ups) >= 4
    assert f.pll_replays() == 0 and 0 < ex < nch * sum(-(-(b // 12) // 1536) for b in blocks)


@pytest.mark.parametrize("solver", [1, 2])
def test_weak_pilot_just_above_the_lock_threshold(fmx_amd, ol, solver):
    """A pilot at 3 % of the deviation (the standard asks for 8-10 %): the lock metric settles at 0.105, 1.5 times its threshold, and
    the PLL's loop gain is a third of the usual one, so whatever disturbs its phase is integrated three times as long.  The
    sequential solver reproduces the reference's trajectory whatever the loop gain; Newton's method on the segment cannot
    reproduce the loop's own f32 rounding noise (fmx_stageb.hip), whose integral grows with the loop's time constant -- still
    inside the PCM tolerance here."""
    blocks = BLOCKS * 6
    iq = ol.synth_iq(sum(blocks), pilotLevel=0.03)
    pg, po, fg, fo, f = run_against_oracle(fmx_amd, ol, iq, blocks, solver)
    e = rms(pg - po)
    print(f"\n[weak pilot 3 %, solver {solver}] PCM rms diff {e:.3e} (signal {rms(po):.3f}), flags {fo[-1]}, per call after lock: "
          + " ".join("%.0e" % v for v in f.per_call_rms[-14:]))
    assert fg == fo and fo[-1][0] == 1
    assert e <= PCM_RMS_TOL


def test_dc_offset_crossing_the_rf_limiter_inside_a_call(fmx_amd, ol):
    """RF DC removal (fm-processor.cpp:423-446): the subtracted value is RfDC limited to +-0.01 per component.  A DC offset of 0.2
    makes RfDC (time constant 1 s) pass +0.01 about 51 ms into the first 0.1 s call; flipping the offset to -0.3 sends it back
    through +0.01 and on through -0.01, again inside calls.  Stage A subtracts behind the FIR with the limiter applied at the
    taps' centre of mass (fmx_front.hip): this is where that shortcut would show."""
    block = 16384 * 14
    calls = 8
    n = block * calls
    iq = ol.synth_iq(n).copy()
    dc = np.where(np.arange(n) < 3 * block, 0.2, -0.3).astype(np.float32)
    iq[:, 0] += dc
    iq[:, 1] -= 0.5 * dc
    pg, po, fg, fo, f = run_against_oracle(fmx_amd, ol, iq, [block] * calls, 1)
    per_call = f.per_call_rms
    print("\n[DC through the limiter] PCM rms diff per call:", " ".join("%.1e" % v for v in per_call))
    assert fg == fo
    assert rms(pg - po) <= PCM_RMS_TOL and max(per_call) <= 2 * PCM_RMS_TOL


def test_newton_solver_batch_equals_single_and_reports_rounds(fmx_amd, ol):
    """A batch above the automatic threshold (65 channels on one stream) uses Newton's method; its channels are bit-identical to each
    other, and equal to the oracle within the PCM tolerance; a 4-channel handle (sequential solver) gives the reference's pilot phase
    to 1e-6 rad, the Newton solver to its rounding-noise floor."""
    block = 16384 * 4
    n = block * 46                                          # 1.3 s
    iq = ol.synth_iq(n)
    o = ol.OracleChain(taps=[ol.TAP_PILOT], inputFilterBw=165000, tap_seconds=1.5)
    po = o.process(iq)
    nt = block // 12
    pil_o = o.tap(ol.TAP_PILOT)[-nt:].astype(np.float64)
    out = {}
    for nch in (4, 65):
        f = fmx_amd.Fmx(nch, streams=1, stream_of_channel=[0] * nch, max_block=block)
        f.set_param(M.P_SCOPE_TAPS, 1)                  # (the pilot-phase tap of a batch: a display feed it does not keep by default)
        gui_defaults(f)
        pcm = np.concatenate([f.process_host(iq[i:i + block]) for i in range(0, n, block)], axis=1)
        d = f.tap(M.TAP_PILOT_PHASE, nt, nch - 1).astype(np.float64) - pil_o
        d -= np.round(d / (2 * np.pi)) * 2 * np.pi
        out[nch] = (pcm, rms(d))
        for c in range(1, nch):
            assert np.array_equal(pcm[c], pcm[0])
        assert rms(pcm[0] - po) <= PCM_RMS_TOL
        assert f.pll_replays() == 0
    print(f"\n[PLL solvers] pilot phase against the oracle over the last call: sequential {out[4][1]:.2e} rad, Newton {out[65][1]:.2e} rad; "
          f"PCM of the two against each other {rms(out[4][0][0] - out[65][0][0]):.2e}")
    assert out[4][1] <= 2e-6
    assert out[65][1] <= 1e-4


def test_non_finite_samples_of_one_rds_channel_leave_its_pair_partner_alone(fmx_amd, ol):
    """channels 2p and 2p + 1 share one complex row of the RDS block transforms.  NaN / Inf bursts in the IQ stream
    of channel 0 must not reach channel 1: its RDS bit stream and PCM equal those of a run without the bursts, bit for bit, and
    channel 0 itself recovers (bits equal to the clean run's again once the burst has left its filters)."""
    block = 16384 * 16
    n = int(2.4 * 2304000) // block * block
    payloads = [ol.rds_programme_bits(pi=0xD3A1, ps="FMX-AMD ", text="CHANNEL ZERO"), ol.rds_programme_bits(pi=0x2468, ps="CHAN TWO", text="SECOND STREAM")]
    iqs = [ol.synth_iq(n, rds=1, rdsLevel=0.05, rds_payload=p) for p in payloads]
    clean = np.stack(iqs)
    dirty = clean.copy()
    burst0 = int(0.9 * 2304000)
    dirty[0, burst0:burst0 + 3000, 0] = np.nan
    dirty[0, burst0 + 5000:burst0 + 5200, 1] = np.inf
    dirty[0, burst0 + 9000:burst0 + 9100, :] = -np.inf

    def run(iq):
        f = fmx_amd.Fmx(2, streams=2, stream_of_channel=[0, 1], max_block=block)
        gui_defaults(f)
        f.set_param(M.P_RDS_MODE, 2)
        f.set_param(M.P_DC_REMOVE, 0)          # (the RF DC estimate is a recurrence: a NaN sample would stay in it for good, in the reference too)
        pcm = []
        for i in range(0, n, block):
            pcm.append(f.process_host(iq[:, i:i + block, :]))
        bits = [f.rds_bits(c, 8192) for c in range(2)]
        return np.concatenate(pcm, axis=1), bits

    pcm_c, bits_c = run(clean)
    pcm_d, bits_d = run(dirty)
    assert len(bits_c[1]) > 2000 and np.array_equal(bits_c[1], bits_d[1])                  # the partner: untouched
    assert np.array_equal(pcm_c[1], pcm_d[1])
    assert np.isfinite(pcm_d[1]).all()
    # the channel with the bursts: the same number of bits, equal again behind the burst (two 32000-sample block filters + slicer pull-in)
    assert len(bits_d[0]) == len(bits_c[0])
    tail = len(bits_c[0]) - 600
    assert np.array_equal(bits_c[0][tail:], bits_d[0][tail:])


@pytest.mark.parametrize("rate", [2400000, 2880000, 3200000, 1920000, 2048000, 1152000, 192000, 250000])
def test_input_rates(fmx_amd, ol, rate):
    """VERDICT r2 missing #1: the reference derives its decimators from the device's rate (fm-processor.cpp:36,68-75,471): fmBand_1 divides
    by 6, fmBand_2 by (inputRate / 6) / fmRate in integer arithmetic (2 from 2.304 MS/s, 1 below: colibri's 1.92 MS/s, rtl-sdr's 2.048),
    and 192 kS/s devices skip both; what is left is treated as 192 kS/s.  Filters, LO table and DC constant are designed for the rate
    given.  Same calls through library and oracle, configs[1] settings plus a local-oscillator offset and a DC offset, with the input
    filter on and -- a second pair of handles -- off (another tap alignment for every twin of the /6 and /1 cases)."""
    decim = 1 if rate // 192000 <= 1 else 6 * ((rate // 6) // 192000)
    block = 16384 * 5 if decim > 1 else 16384
    nblocks = 22 if decim > 1 else 40
    n = block * nblocks
    lo = 30000 if decim > 1 else 3000
    iq = ol.synth_iq(n, offsetHz=float(lo), dcI=0.004, dcQ=-0.003)     # (time base of the generator: 2.304 MS/s; the receivers are told `rate`)
    for bw in ((165000 if decim > 1 else 150000), 0):
        f = fmx_amd.Fmx(1, max_block=block, inputRate=rate)
        gui_defaults(f, bw)
        f.set_param(M.P_LOCAL_OSCILLATOR, lo)
        o = ol.OracleChain(inputRate=rate, inputFilterBw=bw, loFrequency=lo, taps=[ol.TAP_FM_IQ], tap_seconds=4.0)
        pg, po = [], []
        for k, i in enumerate(range(0, n, block)):
            pg.append(f.process_host(iq[i:i + block])[0]); po.append(o.process(iq[i:i + block]))
            assert pg[-1].shape == po[-1].shape, (k, pg[-1].shape, po[-1].shape)
        pg, po = np.concatenate(pg), np.concatenate(po)
        nt = block // decim
        nfm = n // decim
        z_g, z_o = f.tap(M.TAP_FM_IQ, nt), o.tap(ol.TAP_FM_IQ)[nfm - nt:nfm]
        print(f"\n[inputRate {rate}, decimation {decim}, input filter {bw}] fm-rate IQ rms {rms(z_g - z_o):.2e} (signal {rms(z_o):.3f}), "
              f"PCM rms {rms(pg - po):.2e} (signal {rms(po):.3f}), frames {len(pg)}")
        assert rms(z_g - z_o) <= 2e-6 * max(rms(z_o), 1e-3)
        assert rms(pg - po) <= PCM_RMS_TOL and len(pg) == n // decim // 192 * 48
        a, m = f.meta(0), o.meta()
        assert a.PilotPllLocked == m.pilotLocked


def test_input_rates_not_built_are_refused(fmx_amd):
    for rate in (1000000, 400000, 3456000):
        with pytest.raises(fmx_amd.FmxError):
            fmx_amd.Fmx(1, max_block=16384, inputRate=rate)


def test_stage_b_as_one_kernel_and_as_two_give_identical_results(fmx_amd, ol):
    """FMX_P_STAGEB_FORM: stage B as one kernel per call (handles up to 768 channels and the counts that fill its rounds of 3
    workgroups per CU) and as two (limiter .. lock detector | PSS .. de-emphasis, 4 workgroups per CU: what 4096 channels run) meet
    in the demodulator-output / pilot-phase rows and a byte of lock flags.  Same arithmetic: PCM, taps, metaData and RDS bits are
    bit-identical, over ragged calls, for every kind of channel (Newton and sequential PLL solver, PLL decoder and level squelch through
    the pre-pass, mono, PSS off) -- and the PCM equals the oracle's."""
    blocks = (BLOCKS * 4)[:22]                               # 1.2 s: through pilot lock, uneven, crossing segment boundaries
    iq = ol.synth_iq(sum(blocks), rds=1, rdsLevel=0.05, rds_payload=ol.rds_programme_bits(pi=0xD3A1, ps="FMX-AMD ", text="TWO KERNELS"))
    nch = 6

    def run(form):
        f = fmx_amd.Fmx(nch, streams=1, stream_of_channel=[0] * nch, max_block=max(blocks))
        gui_defaults(f)
        f.set_param(M.P_STAGEB_FORM, form)
        f.set_param(M.P_RDS_MODE, 2)
        f.set_param(M.P_PLL_SOLVER, 2)
        f.set_param(M.P_PLL_SOLVER, 1, channel=1)
        f.set_param(M.P_FM_DECODER, 2, channel=2)
        f.set_param(M.P_SQUELCH_MODE, 2, channel=3); f.set_param(M.P_SQUELCH_VALUE, 30, channel=3)
        f.set_param(M.P_FM_MODE, 2, channel=4)
        f.set_param(M.P_PSS, 0, channel=5)
        pcm, metas, pos = [], [], 0
        for k, b in enumerate(blocks):
            if form == 0:                                    # (the form may change from call to call: the state is the same)
                f.set_param(M.P_STAGEB_FORM, 1 + (k * 7 // 3) % 2)
            pcm.append(f.process_host(iq[pos:pos + b])); pos += b
            metas.append([(m.PilotPllLocked, m.PssState, m.PilotPllLockStrength, m.DcValIf, m.live_lock_strength) for m in (f.meta(c) for c in range(nch))])
        nt = blocks[-1] // 12
        taps = [f.tap(t, nt, c) for t in (M.TAP_PILOT_PHASE, M.TAP_DEMOD) for c in range(nch)]
        bits = [f.rds_bits(c, 8192) for c in range(nch)]
        return np.concatenate(pcm, axis=1), metas, taps, bits

    one, two, mixed = run(1), run(2), run(0)
    for other in (two, mixed):
        assert np.array_equal(one[0], other[0])
        assert one[1] == other[1]
        for a, b in zip(one[2], other[2]):
            assert np.array_equal(a, b)
        for a, b in zip(one[3], other[3]):
            assert len(a) > 800 and np.array_equal(a, b)
    po = ol.OracleChain(inputFilterBw=165000).process(iq)
    assert rms(two[0][0] - po) <= PCM_RMS_TOL
    with pytest.raises(Exception):
        fmx_amd.Fmx(1, max_block=16384).set_param(M.P_STAGEB_FORM, 3)


@pytest.mark.parametrize("bw", [165000, 0])
def test_stage_a_history_follows_lo_off_and_dc_removal_toggles(fmx_amd, ol, bw):
    """ADVICE r2 (low #1).  Stage A keeps a channel's FIR history raw while it has no local oscillator (RfDC and the IQ balance are applied
    behind the FIR) and processed while it has one; the reference's filter memory always holds processed samples of their own time
    (fm-processor.cpp:423-470).  Every change in between is a conversion of the 24 history columns on load: LO on, LO back to 0,
    setDCRemove off / on (which zeroes RfDC, :922-925) with and without an LO.  A stream with a DC offset and an unbalanced IQ pair, the
    fm-rate IQ of the call behind every switch against the oracle's (a history taken in the wrong format shows as up to 5e-3 over the
    first 24 samples), and the PCM over everything."""
    block = 16384 * 3                                       # 4096 fm samples per call
    switches = {4: dict(loFrequency=3000), 8: dict(loFrequency=0), 11: dict(dcRemove=0), 14: dict(dcRemove=1),
                17: dict(loFrequency=-2500), 19: dict(dcRemove=0), 21: dict(loFrequency=0), 23: dict(dcRemove=1)}
    nb = 26
    iq = ol.synth_iq(nb * block)
    iq[:, 0] += 0.007; iq[:, 1] -= 0.005
    o = ol.OracleChain(inputFilterBw=bw, attL=0.9, attR=1.1, taps=[ol.TAP_FM_IQ], tap_seconds=4.0)
    f = fmx_amd.Fmx(1, max_block=block)
    gui_defaults(f, bw)
    f.set_param(M.P_ATTENUATION_L, 0.9); f.set_param(M.P_ATTENUATION_R, 1.1)
    ids = dict(loFrequency=M.P_LOCAL_OSCILLATOR, dcRemove=M.P_DC_REMOVE)
    nt = block // 12
    po, pg, worst = [], [], {}
    for b in range(nb):
        for k, v in switches.get(b, {}).items():
            o.configure(**{k: v}); f.set_param(ids[k], v)
        x = iq[b * block:(b + 1) * block]
        po.append(o.process(x)); pg.append(f.process_host(x)[0])
        z_g, z_o = f.tap(M.TAP_FM_IQ, nt), o.tap(ol.TAP_FM_IQ)[b * nt:(b + 1) * nt]
        d = np.abs(z_g - z_o).max(axis=1)
        if b in switches or b - 1 in switches:
            worst[b] = (float(d[:64].max()), float(d.max()))
        assert d.max() <= 2e-5, (b, float(d.max()), int(d.argmax()))
    po, pg = np.concatenate(po), np.concatenate(pg)
    print(f"\n[stage A history, input filter {bw}] fm-rate IQ max |diff| in the calls behind a switch (first 64 samples, whole call): "
          + ", ".join(f"{b}: {a:.1e}/{w:.1e}" for b, (a, w) in sorted(worst.items())) + f"; PCM rms {rms(pg - po):.2e}")
    assert po.shape == pg.shape and rms(pg - po) <= PCM_RMS_TOL


def test_shard_collectives_over_rccl_one_rank(fmx_amd, ol, tmp_path):
    """SURVEY 8e / VERDICT r2 ("RCCL has never been exercised"): the gather / broadcast / max-over-ranks helpers of shard.py -- what
    bench.py --gpus N uses either side of the path -- over backend "nccl" (= RCCL) on the GPU box.  One GPU allows one rank (RCCL
    refuses two ranks on a device), so this is a one-rank group with the collectives forced on: communicator set-up and the
    collectives' kernels run on device tensors produced by the library; the two-rank logic is covered on gloo
    (test_distributed_cpu.py)."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = tmp_path / "w.py"
    worker.write_text(textwrap.dedent('''
        import importlib, os, sys
        import numpy as np, torch, torch.distributed as dist
        sys.path.insert(0, %r)
        import bench
        pkg = importlib.import_module("sdr-j-fm_amd"); m = pkg.fmx; shard = pkg.shard
        dev = torch.device("cuda", 0); torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
        assert dist.get_backend() == "nccl"
        ch, n = 5, 16384 * 6
        f = pkg.Fmx(ch, max_block=n)
        for pid, v in ((m.P_BANDWIDTH, 165000), (m.P_LF_CUTOFF, 15000), (m.P_DEEMPHASIS, 50), (m.P_VOLUME_DB, -6.0), (m.P_FM_MODE, 0)):
            f.set_param(pid, v)
        iq = bench.synth_device(torch, ch, n, dev)
        iq = shard.broadcast_stream(iq, src=0)
        pcm = torch.zeros((ch, n // 48 + 96, 2), dtype=torch.float32, device=dev)
        fr = 0
        for _ in range(3):
            fr = f.process_device(iq.data_ptr(), n, n, pcm.data_ptr(), pcm.shape[1], hip_stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        loc = pcm[:, :fr].contiguous()
        full = shard.gather_pcm(loc, ch, dst=0)
        torch.cuda.synchronize()
        assert full.data_ptr() != loc.data_ptr() and torch.equal(full, loc) and float(loc.abs().max()) > 0.01
        assert shard.max_over_ranks(3.25, device=dev) == 3.25
        t = torch.tensor([float(fr)], device=dev); dist.all_reduce(t); assert int(t.item()) == fr
        dist.barrier(); dist.destroy_process_group()
        print("RCCL_OK", fr, tuple(full.shape))
    ''' % root))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", FMX_SHARD_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(worker)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def _corner_signal(ol, n, seed):
    rng = np.random.default_rng(seed)
    steps = rng.choice([0, 1, 1, 1, 2, 3], size=n)            # quarter turns per sample
    k = np.cumsum(steps) % 4
    unit = np.array([[1, 0], [0, 1], [-1, 0], [0, -1]], np.float32)
    iq = 0.5 * unit[k]
    smooth = ol.synth_iq(n)[:, :]                             # an ordinary stereo signal (its time base does not matter here)
    use_smooth = (np.arange(n) // 3000) % 3 == (seed % 3)     # ordinary stretches in between: segments without a corner
    iq[use_smooth] = smooth[use_smooth]
    zero = (np.arange(n) // 1777) % 7 == 3
    iq[zero] = 0.0
    return iq


@pytest.mark.parametrize("variant", ["small", "solver2", "batch65", "batch4096"])
def test_discriminator_corner_arguments_of_the_atan_table(fmx_amd, ol, variant):
    """compAtan::atan2 (Xtan2.cpp:56-68) answers x = 0 without its table (+-pi/2, 0 for y = 0).  The kernel computes the table arm of a
    thread's six samples without those corners and looks for them once per wave (v_cmp_class), taking the general form when one turns
    up.  A 192 kS/s stream (no decimators: the samples reach the limiter as they are) that steps by exactly 90 degrees, flips by 180,
    repeats samples and falls to zero for stretches makes x = 0, y = 0 and the limiter's 0.001 floor appear in every combination, in
    some segments only: demodulator output and PCM against the oracle.
    variant "small": one channel on the automatic settings -- the reference's own divisions and a sequentially walked AFC (exact_disc): equal
    to the oracle to the last bit.  "solver2" (FMX_P_PLL_SOLVER = 2 on one channel), "batch65" (65 channels on the automatic settings: above
    the 64-channel limit of the exact forms) and "batch4096" (4096 channels on four such streams) run what every large batch runs -- the
    v_cmp_class fast arm; the AFC as a time-parallel scan and the short forms of the limiter's divisions for TUNED channels.  This signal's AFC
    average sits at ~1 rad, a hundred times an ordinary station's: rounds 3-4 bounded the batch path at 3e-4 of the demodulator's scale here
    (measured 1e-4: VERDICT r4 weak #1, next #5); since round 5 a batch channel whose average is above 0.15 rad takes the sequentially walked
    average and the reference's divisions for as long as it is (fmx_stageb.hip: AFC_EXACT_THR) -- 5e-7, one ulp of the scale, PCM 7e-8."""
    block = 16384
    nb = 30
    n = block * nb
    nst = 4 if variant == "batch4096" else 1
    iq = np.stack([_corner_signal(ol, n, 7 + sidx) for sidx in range(nst)])
    nch = {"batch65": 65, "batch4096": 4096}.get(variant, 1)
    for dec in (3, 4):
        f = fmx_amd.Fmx(nch, streams=nst, stream_of_channel=[c % nst for c in range(nch)], max_block=block, inputRate=192000)
        f.set_param(M.P_SCOPE_TAPS, 1)                  # (the demodulator tap of a batch: a display feed it does not keep by default)
        gui_defaults(f, 0)
        f.set_param(M.P_DC_REMOVE, 0); f.set_param(M.P_FM_DECODER, dec)
        if variant == "solver2":
            f.set_param(M.P_PLL_SOLVER, 2)
        os_ = [ol.OracleChain(inputRate=192000, inputFilterBw=0, dcRemove=0, decoder=dec, taps=[ol.TAP_DEMOD], tap_seconds=3.0) for _ in range(nst)]
        pg, po, worst = [[] for _ in range(nst)], [[] for _ in range(nst)], 0.0
        for b in range(nb):
            x = iq[:, b * block:(b + 1) * block]
            pc = f.process_host(x)
            assert all(np.array_equal(pc[c], pc[c % nst]) for c in range(nst, nch))      # (the channels of a batch on one stream: each other's twins)
            for sidx in range(nst):
                c = nch - nst + sidx                                                      # (the last listener of the stream)
                pg[sidx].append(pc[c]); po[sidx].append(os_[sidx].process(x[sidx]))
                d_g, d_o = f.tap(M.TAP_DEMOD, block, c), os_[sidx].tap(ol.TAP_DEMOD)[b * block:(b + 1) * block]
                worst = max(worst, float(np.abs(d_g - d_o).max()))
        e = max(rms(np.concatenate(pg[sidx]) - np.concatenate(po[sidx])) for sidx in range(nst))
        print(f"\n[atan corners, {variant}, decoder {dec}] demodulator output max |diff| {worst:.2e} (full scale {np.abs(os_[0].tap(ol.TAP_DEMOD)).max():.2f}), PCM rms {e:.2e}")
        # (a wrong corner would be off by pi/2 or more = 2.4 of the demodulator's scale)
        assert worst <= (0.0 if variant == "small" else 2e-5) and e <= (1e-6 if variant != "small" else PCM_RMS_TOL)


MID_ORDER = [dict(inputFilterBw=0), dict(inputFilterBw=120000), dict(lfCutoff=12000), dict(lfCutoff=0), dict(lfCutoff=15000), dict(inputFilterBw=165000),
             dict(inputFilterBw=165000, lfCutoff=9000)]


def run_mid_stream_changes(fmx_amd, ol, nch, gap_s, block=16384 * 3, restarts=None):
    """the changes of MID_ORDER, one every gap_s seconds, through the library and the oracle: per-call PCM RMS differences and the change calls"""
    per_s = 2304000 / block
    gap = int(gap_s * per_s)
    switches = {(i + 1) * gap: d for i, d in enumerate(MID_ORDER)}
    nb = (len(MID_ORDER) + 1) * gap
    iq = ol.synth_iq(nb * block)
    o = ol.OracleChain(inputFilterBw=165000)
    f = fmx_amd.Fmx(nch, streams=1, stream_of_channel=[0] * nch, max_block=block)
    if restarts is not None:
        f.set_param(M.P_FILTER_RESTARTS, restarts)
    gui_defaults(f)
    per_call = []
    for b in range(nb):
        for k, v in switches.get(b, {}).items():
            o.configure(**{k: v})
            f.set_param(M.P_BANDWIDTH if k == "inputFilterBw" else M.P_LF_CUTOFF, v)
        x = iq[b * block:(b + 1) * block]
        po, pg = o.process(x), f.process_host(x)
        assert pg[0].shape == po.shape and np.isfinite(pg).all() and float(np.abs(pg).max()) <= 1.5
        for c in range(1, nch):
            assert np.array_equal(pg[c], pg[0])
        per_call.append(rms(pg[0] - po))
    return per_call, switches, gap


def test_mid_stream_filter_changes_single_receiver(fmx_amd, ol):
    """VERDICT r3 missing #1 / next #5.  setBandwidth (radio.cpp:1706-1712 -> fm-processor.cpp:232-239,396-408) and setlfcutoff (:762-770)
    while the stream runs.  The reference's overlap-add filters restart their block position at every setLowPass
    (fft-filters.cpp:71-95: `inp = 0`, buffers kept): the reference plays the last completed output block again (65285 input samples =
    28 ms for the input filter, 7436 fm samples = 39 ms for the audio filter), drops the block in progress and adds the old block's tail
    to the first block of the new kernel; "Off" lets the undelayed samples through at once.  A handle of up to 64 channels runs the two
    filters as those block machines (fmx_ola.hip, FMX_P_FILTER_RESTARTS): "165kHz" -> "Off" -> "120kHz", three changes of the audio
    cut-off, back to "165kHz", and both filters at once, every 0.5 s -- the PCM stays within the tolerance in EVERY call, the glitches
    included."""
    per_call, switches, gap = run_mid_stream_changes(fmx_amd, ol, 1, 0.5)
    print("\n[mid-stream filter changes, single receiver] worst call behind each change: "
          + ", ".join(f"{switches[s_]}: {max(per_call[s_:s_ + gap]):.1e}" for s_ in sorted(switches)))
    assert max(per_call) <= PCM_RMS_TOL


def test_mid_stream_filter_changes_are_bounded_with_the_folded_filters_pinned(fmx_amd, ol):
    """(Round 6: a batch on the automatic setting is EXACT behind such a change -- tests/test_gpu_round6.py::test_mid_stream_filter_changes_are_exact_in_a_batch;
    this is what FMX_P_FILTER_RESTARTS = 2, the folded filters for good, still gives.)
    The same changes on a handle above 64 channels, whose filters are folded into the polyphase FIRs of stage A and stage C: the new tap
    set applies from the call's first sample and the rings are read at the new latency -- during one filter latency the reference's
    output and the library's are both glitches, and different ones.  Behind an AUDIO filter change nothing else has state: the PCM agrees
    again within 0.25 s.  Behind an INPUT filter change the 28 ms of different fm-rate IQ kick the pilot PLL, the lock detector and the PSS
    differently; where the glitch costs one side its pilot lock the stereo decoder comes back half a second apart and the PSS
    integrator re-converges behind it.  Asserted: PCM finite and bounded throughout (run_mid_stream_changes), within the tolerance again
    at most 0.25 s behind an audio filter change and 2.3 s behind an input filter change, and from there until the next change."""
    per_call, switches, gap = run_mid_stream_changes(fmx_amd, ol, 65, 2.5, restarts=2)
    per_s = 2304000 / (16384 * 3)
    back = {s_: next((i for i in range(gap) if all(v <= PCM_RMS_TOL for v in per_call[s_ + i:s_ + gap])), gap) for s_ in sorted(switches)}
    print("\n[mid-stream filter changes, 65 channels, folded filters] seconds behind each change until the PCM is back under 1e-5 for good: "
          + ", ".join(f"{switches[s_]}: {back[s_] / per_s:.2f} (worst call {max(per_call[s_:s_ + gap]):.1e})" for s_ in sorted(switches)))
    assert max(per_call[30:gap]) <= PCM_RMS_TOL
    for s_ in sorted(switches):
        assert back[s_] <= (0.25 if "inputFilterBw" not in switches[s_] else 2.3) * per_s, (switches[s_], back[s_])
