"""Round-5 GPU tests.  Stage A on the matrix pipe (FMX_P_FRONT_KERNEL = 3, csrc/fmx_front4.hip): the kernel takes the whole 1536-sample tiles of
a call, front_kernel the rest -- the results must be front_kernel's to the bounds of _close.  (The tests were written for two kernels: round 5's
six-wave VALU kernel, `kernel == 2`, bit-identical and slower, left the product in round 6: tools/experiments/fmx_front3.hip.)"""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
M = importlib.import_module("sdr-j-fm_amd").fmx

NCH, NST = 8, 3


def _handle(fmx_amd, kernel, max_block, nch=NCH, nst=NST):
    f = fmx_amd.Fmx(nch, streams=nst, stream_of_channel=[c % nst for c in range(nch)], max_block=max_block)
    for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FM_MODE, 0)):
        f.set_param(pid, v)
    f.set_param(M.P_FILTER_RESTARTS, 2)          # (the folded filters: what batches run; the block machines of small handles have a stage A of their own)
    f.set_param(M.P_FRONT_KERNEL, kernel)
    f.set_param(M.P_FRONT_PARTS, 1)
    # per-channel variety: another bandwidth (another tap set), IQ balance, RF DC removal off
    f.set_param(M.P_BANDWIDTH, 120000, 1)
    f.set_param(M.P_BANDWIDTH, 200000, 5)
    f.set_param(M.P_ATTENUATION_L, 0.9, 2); f.set_param(M.P_ATTENUATION_R, 1.1, 2)
    f.set_param(M.P_DC_REMOVE, 0, 3)
    return f


def _streams(ol, n):
    iq = np.stack([ol.synth_iq(n, leftHz=400.0 + 300 * k, rightHz=700.0 + 200 * k) for k in range(NST)])
    iq[0] += np.array([0.004, -0.003], np.float32)
    iq[1] += np.array([-0.02, 0.015], np.float32)             # (beyond the +-0.01 limiter)
    return iq


def _run(f, iq, blocks, events=None):
    """blocks: list of call lengths, or of tuples of lengths -- a tuple's samples are one stretch of the stream, made in that many calls"""
    pcm, taps, dcs, pos = [], [], [], 0
    for bi, b in enumerate(blocks):
        if events and bi in events:
            for pid, v, c in events[bi]:
                f.set_param(pid, v, c)
        for part in (b if isinstance(b, tuple) else (b,)):
            pcm.append(f.process_host(iq[:, pos:pos + part])); pos += part
            nt = f.last_fm_samples()
            taps.append(np.stack([f.tap(M.TAP_FM_IQ, nt, c) for c in range(NCH)]))
        dcs.append([(f.meta(c).live_rf_dc_re, f.meta(c).live_rf_dc_im) for c in range(NCH)])
    return np.concatenate(pcm, axis=1), np.concatenate(taps, axis=1), np.array(dcs)


def _close(a, b, kernel, what):
    """kernel 2 (the same f32 arithmetic in another order of waves): equal to the last bit.  kernel 3 (the filter on the matrix pipe, operands split
    into two f16 halves, products exact, the remainders' roundings at 2^-22 of a product; the RF DC recurrence with its sums taken in another
    order): the fm-rate IQ to 4e-7 of its amplitude at the worst sample (the f32 filter's own rounding is 1e-7), RfDC to 2e-8, PCM to 2e-6."""
    if kernel == 2:
        assert np.array_equal(a, b), what + " differs"
        return
    tol = {"fm-rate IQ": 8e-7 * float(np.abs(a).max()), "RF DC state": 2e-8, "PCM": 2e-6}[what]
    err = float(np.abs(a.astype(np.float64) - b).max())
    print("\n[front kernel 3 against 1] %s: max |diff| %.2e (bound %.1e, full scale %.2f)" % (what, err, tol, float(np.abs(a).max())))
    assert err <= tol, (what, err, tol)


@pytest.mark.parametrize("kernel", [3])
def test_front3_whole_tiles_bit_identical(fmx_amd, ol, kernel):
    """Calls of whole tiles (1 .. 150 of them: fewer tiles than the workgroup has waves, more, a multiple, not a multiple): six waves per channel
    against four -- PCM, the fm-rate IQ and the RF DC state bit for bit; a setDCRemove in mid-stream (its reset of RfDC) on the way.  And the
    filter on the matrix pipe (kernel 3) against the same, to the bounds of _close."""
    blocks = [1536 * 10, 1536 * 150, 1536 * 1, 1536 * 7, 1536 * 6, 1536 * 5, 1536 * 13]
    iq = _streams(ol, sum(blocks))
    ev = {3: [(M.P_DC_REMOVE, 1, 3), (M.P_DC_REMOVE, 0, 4)], 5: [(M.P_DC_REMOVE, 1, 4)]}
    outs = []
    for kn in (1, kernel):
        f = _handle(fmx_amd, kn, max(blocks))
        outs.append(_run(f, iq, blocks, ev))
        del f
    a, b = outs
    assert np.isfinite(a[0]).all() and float(np.abs(a[0]).max()) > 0.01
    _close(a[1], b[1], kernel, "fm-rate IQ")
    _close(a[2], b[2], kernel, "RF DC state")
    _close(a[0], b[0], kernel, "PCM")


@pytest.mark.parametrize("kernel", [3])
def test_front3_remainders_and_fallbacks(fmx_amd, ol, kernel):
    """Calls that are not whole tiles: the six-wave kernel takes the tiles, front_kernel the remainder as a call of its own -- so a handle on
    front_kernel alone that is given the same stretches in two calls (tiles, remainder) must agree bit for bit in the fm-rate IQ.  Calls shorter
    than a tile, calls that leave the 12-sample column grid (everything behind them is front_kernel's), a local oscillator switched on (the
    handle falls back) and off again (the six-wave kernel finds a history of mixed samples and converts it)."""
    T = 1536
    # (the first stretch carries the stream past the input filter's latency of 65285 samples: everything in front of it is zeros at the fm rate)
    stretches = [(50 * T, 480), (600,), (9 * T, 36), (2 * T, 1500), (7 * T,), (T, 12), (4 * T, 7), (5 * T,), (2 * T + 100,)]
    n = sum(sum(s) for s in stretches)
    iq = _streams(ol, n)
    ev = {4: [(M.P_LOCAL_OSCILLATOR, 200000, 6)], 5: [(M.P_LOCAL_OSCILLATOR, 0, 6)]}
    fa = _handle(fmx_amd, 1, 64 * T)
    a = _run(fa, iq, stretches, ev)
    del fa
    fb = _handle(fmx_amd, kernel, 64 * T)
    b = _run(fb, iq, [sum(s) for s in stretches], ev)
    del fb
    assert float(np.abs(a[1]).max()) > 0.1
    _close(a[1], b[1], kernel, "fm-rate IQ")
    _close(a[2], b[2], kernel, "RF DC state")
    # the stages behind are invariant to the cut to rounding only
    assert float(np.abs(a[0] - b[0]).max()) < 2e-6


@pytest.mark.parametrize("kernel", [3])
def test_front3_against_oracle_in_a_batch(fmx_amd, ol, kernel):
    """600 channels on 4 streams, six waves per channel (two channels per workgroup, an even count), three calls -- whole tiles,
    tiles and a remainder twice --, every 152nd channel against the oracle chain on its stream."""
    nch, nst, T = 600, 4, 1536
    blocks = [T * 100, T * 50 + 300, T * 42 - 300]          # (18 of the oracle's 16384-sample blocks)
    n = sum(blocks)
    iq = np.stack([ol.synth_iq(n, leftHz=500.0 + 250 * k, rightHz=900.0 + 150 * k) for k in range(nst)])
    f = fmx_amd.Fmx(nch, streams=nst, stream_of_channel=[c % nst for c in range(nch)], max_block=max(blocks))
    for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FRONT_KERNEL, kernel)):
        f.set_param(pid, v)
    pcm, pos = [], 0
    for b in blocks:
        pcm.append(f.process_host(iq[:, pos:pos + b])); pos += b
    pcm = np.concatenate(pcm, axis=1)
    for k in range(nst):
        ch = ol.OracleChain(inputFilterBw=165000)
        ref = ch.process(iq[k])
        for c in range(k, nch, 152):
            assert pcm[c].shape == ref.shape, (pcm[c].shape, ref.shape)
            err = float(np.sqrt(np.mean((pcm[c].astype(np.float64) - ref) ** 2)))
            if c < nst:
                print("\n[front kernel %d, 600 channels] channel %d PCM rms against the oracle %.2e" % (kernel, c, err))
            assert err <= 1e-5, (c, err)


def test_long_rds_calls_in_pieces_at_a_rate_decimated_by_six(fmx_amd, ol):
    """the pieces a long call is made in while a channel decodes RDS are 31999 FM samples -- 191994 input samples at 2.048 MS/s, which
    the reference decimates by 6, not the 383988 of 2.304 MS/s.  One call of 921600 samples (153600 fm samples) against four of 230400 (each of
    which is itself made in pieces: 38400 fm samples): the same PCM to the chain's block-size invariance, the same bits."""
    rate, n = 2048000, 921600
    iq = np.stack([ol.synth_iq(2 * n, inputRate=rate, rds=1, rdsLevel=0.05, rdsBitsSeed=sd) for sd in (5, 6)])
    res = []
    for blocks in ([230400] * 4 + [n], [230400] * 8):
        f = fmx_amd.Fmx(2, max_block=n, inputRate=rate)
        for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FILTER_RESTARTS, 2), (M.P_RDS_MODE, 2)):
            f.set_param(pid, v)
        pcm, pos = [], 0
        for b in blocks:
            pcm.append(f.process_host(iq[:, pos:pos + b])); pos += b
        res.append((np.concatenate(pcm, axis=1), [f.rds_bits(c, 8192) for c in range(2)], f.last_fm_samples()))
        del f
    (p1, b1, l1), (p4, b4, l4) = res
    assert p1.shape == p4.shape and float(np.sqrt(np.mean((p1.astype(np.float64) - p4) ** 2))) <= 2e-7
    assert all(len(a) == len(b) and np.array_equal(a, b) for a, b in zip(b1, b4))
    assert l1 == (n - 4 * 191994) // 6 and l4 == (230400 - 191994) // 6        # (the taps hold the last piece)


@pytest.mark.parametrize("rds,form,hard", [(0, 0, 0), (1, 1, 0), (0, 2, 1), (1, 0, 1)])
def test_twins_at_batch_scale_flake_hunt(rds, form, hard):
    """a race in stage B's second kernel made one channel-call in two million differ from its twin and was found by
    chance.  tools/diag/flake_hunt.py in the suite, bounded: 4096 channels, channel c on programme c % 4, handles created and destroyed in a
    loop, each run through pilot acquisition; every call every channel's PCM (and, with RDS on, its bits) must equal its twin's on the device.
    Both forms of stage B, RDS on for half, ordinary stations and the hard population (a pilot flapping across the lock threshold, one
    creeping through it, noise only, a station with a DC offset and a local oscillator)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "diag", "flake_hunt.py"), "10", "8", "115200", "4096", str(rds), str(form), str(hard)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    last = r.stdout.strip().splitlines()[-1]
    assert last.endswith("mismatch: 0"), r.stdout[-3000:]


@pytest.mark.parametrize("fmt", ["u8", "s8", "s16"])
def test_front4_raw_formats(fmx_amd, ol, fmt):
    """Raw integer samples through the matrix-pipe kernel (converted while they are loaded, as the reference's device handlers convert them):
    against front_kernel on the same calls -- whole tiles, tiles and a remainder -- to the bounds of _close."""
    T = 1536
    blocks = [150 * T, 40 * T + 600, 60 * T]
    n = sum(blocks)
    x = ol.synth_iq(n)
    if fmt == "u8":
        raw = np.clip(np.round(x * 100.0 + 127.4), 0, 255).astype(np.uint8); code = M.IQ_U8
    elif fmt == "s8":
        raw = np.clip(np.round(x * 100.0 + 0.4), -128, 127).astype(np.int8); code = M.IQ_S8
    else:
        raw = np.clip(np.round(x * 1500.0 + 9.0), -32768, 32767).astype(np.int16); code = M.IQ_S16
    outs = []
    for kernel in (1, 3):
        f = _handle(fmx_amd, kernel, max(blocks), nch=NCH, nst=1)
        pcm, taps, pos = [], [], 0
        for b in blocks:
            pcm.append(f.process_host_raw(raw[None, pos:pos + b], code)); pos += b
            assert f.last_front_kernel() == kernel
            taps.append(np.stack([f.tap(M.TAP_FM_IQ, f.last_fm_samples(), c) for c in range(NCH)]))
        outs.append((np.concatenate(pcm, axis=1), np.concatenate(taps, axis=1)))
        del f
    (pa, ta), (pb, tb) = outs
    assert float(np.abs(ta).max()) > 0.1 and float(np.abs(pa).max()) > 0.01
    # (channel 2 has an IQ balance and the stream a DC offset: front_kernel takes such a channel through its per-sample pass since round 6 -- the reference's own
    # RfDC recurrence in front of the balance --, the matrix-pipe kernel removes RfDC behind the filter: 2.5e-6 of the scale between the two)
    rest = [c for c in range(NCH) if c != 2]
    _close(ta[rest], tb[rest], 3, "fm-rate IQ")
    assert float(np.abs(ta[2].astype(np.float64) - tb[2]).max()) <= 4e-6 * float(np.abs(ta[2]).max())
    _close(pa, pb, 3, "PCM")


def test_front4_large_samples_are_linear_and_bad_ones_stay_in_their_channel(fmx_amd, ol):
    """Round 5's matrix-pipe kernel worked on f16 halves of the samples times a FIXED 2^12 and limited |x| >= 16.  Since round 6 every tile has a
    scale of its own (block floating point): a burst of samples of magnitude 100 in the middle of a stream goes through the filter as it goes
    through front_kernel's f32 one (the fm-rate IQ of both kernels agree to 8e-7 of the burst's scale), nothing is limited; and what happens to
    one stream is no other stream's business: the burst and a NaN / Inf burst on stream 1 leave the PCM of the channels on stream 0
    bit-identical, the channels on stream 1 finite where their input was (the reference's filter spreads a NaN over its own length too)."""
    T = 1536
    blocks = [100 * T, 100 * T]
    n = sum(blocks)
    iq = np.stack([ol.synth_iq(n, leftHz=400.0 + 300 * k, rightHz=700.0 + 200 * k) for k in range(2)])
    bad = iq.copy()
    bad[1, 50 * T:50 * T + 3000] *= 400.0                     # (magnitude ~100)
    big = bad.copy()
    bad[1, 120 * T:120 * T + 100] = np.nan
    bad[1, 121 * T:121 * T + 100, 0] = np.inf
    def run(x, kernel):
        f = fmx_amd.Fmx(6, streams=2, stream_of_channel=[0, 1, 0, 1, 0, 1], max_block=max(blocks))
        for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FILTER_RESTARTS, 2), (M.P_FRONT_KERNEL, kernel), (M.P_FRONT_PARTS, 1)):
            f.set_param(pid, v)
        pcm, taps, pos = [], [], 0
        for b in blocks:
            pcm.append(f.process_host(x[:, pos:pos + b])); pos += b
            assert f.last_front_kernel() == kernel
            taps.append(np.stack([f.tap(M.TAP_FM_IQ, f.last_fm_samples(), c) for c in range(2)]))
        return np.concatenate(pcm, axis=1), np.concatenate(taps, axis=1)
    a, _ = run(iq, 3)
    b, _ = run(bad, 3)
    for c in (0, 2, 4):
        assert np.array_equal(a[c], b[c]), c
    first_nan_frame = (120 * T) // 48
    assert np.isfinite(b[1][:first_nan_frame - 200]).all()
    # the burst through both kernels: the same fm-rate IQ (the burst arrives one filter latency later: the second call's ring)
    (_, t3), (_, t1) = run(big, 3), run(big, 1)
    scale = float(np.abs(t1[1]).max())
    err = float(np.abs(t3[1].astype(np.float64) - t1[1]).max())
    print("\n[a burst of magnitude 100 through kernels 3 and 1] fm-rate IQ max |diff| %.2e of a scale of %.1f" % (err, scale))
    assert scale > 100.0 and err <= 8e-7 * scale


def test_front_kernel_choice(fmx_amd, ol):
    """The automatic choice of the input-filter kernel (fmx_last_front_kernel): the matrix-pipe kernel for a handle that fills the GPU without
    splitting its channels in time, has no local oscillator and the input filter on everywhere, for calls on the 12-sample grid that hold a
    whole tile; front_kernel otherwise.  (Round 6: a handle with local oscillators runs the kernel's complex-tap variant, one channel per workgroup, when it has
    a channel per compute unit.)"""
    T = 1536
    x = ol.synth_iq(64 * T + 5)
    def run(nch, setup, n=4 * T, pre=0):
        f = fmx_amd.Fmx(nch, streams=1, stream_of_channel=[0] * nch, max_block=64 * T)
        for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_FILTER_RESTARTS, 2)):
            f.set_param(pid, v)
        setup(f)
        if pre:
            f.process_host(x[None, :pre])
        f.process_host(x[None, pre:pre + n])
        k = f.last_front_kernel()
        del f
        return k
    assert run(600, lambda f: None) == 3
    assert run(600, lambda f: None, n=T - 12) == 1                                   # (no whole tile)
    assert run(600, lambda f: None, n=2 * T, pre=5) == 1                             # (off the 12-sample grid)
    assert run(600, lambda f: f.set_param(M.P_LOCAL_OSCILLATOR, 200000, 7)) == 3     # (a local oscillator somewhere: since round 6 the complex-tap variant, fmx_front4lo.hip)
    assert run(200, lambda f: f.set_param(M.P_LOCAL_OSCILLATOR, 200000, 7), n=64 * T) == 1     # (... which wants a channel per compute unit)
    assert run(600, lambda f: f.set_param(M.P_BANDWIDTH, 0, 3)) == 1                 # (the input filter off somewhere)
    assert run(600, lambda f: f.set_param(M.P_FRONT_KERNEL, 1)) == 1
    assert run(40, lambda f: None, n=64 * T) == 1                                    # (too few channels: split in time on front_kernel)
    assert run(40, lambda f: f.set_param(M.P_FRONT_KERNEL, 3), n=64 * T) == 3


def test_scope_taps_switch(fmx_amd, ol):
    """FMX_P_SCOPE_TAPS: the demodulator output, the L-R difference in front of the matrix and the pilot phase are display feeds (the reference's scopes) --
    rows of stage B's work arrays --, and so is the peak-level meter: produced by a handle of up to 64 channels, not by a larger batch unless asked for (a batch without them runs stage B as
    one kernel that leaves the rows unwritten); the PCM does not know the difference, and a batch that keeps them has the small handle's values.  A channel
    that decodes RDS has its rows regardless of the taps: the RDS path reads them."""
    n = 16384 * 6
    x = ol.synth_iq(n, rds=1, rdsLevel=0.05)
    def run(nch, taps, rds=False):
        f = fmx_amd.Fmx(nch, streams=1, stream_of_channel=[0] * nch, max_block=n)
        for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_FILTER_RESTARTS, 2), (M.P_PLL_SOLVER, 1)):
            f.set_param(pid, v)
        if taps is not None: f.set_param(M.P_SCOPE_TAPS, taps)
        if rds: f.set_param(M.P_RDS_MODE, 2, nch - 1)
        pcm = f.process_host(x[None])
        out = []
        for t in (M.TAP_DEMOD, M.TAP_LR_RAW, M.TAP_PILOT_PHASE):
            try:
                out.append(f.tap(t, 4096, nch - 1))
            except Exception as e:
                out.append(str(e))
        try:
            out.append(f.peaks(nch - 1))
        except Exception as e:
            out.append(str(e))
        ring = f.tap(M.TAP_FM_IQ, 4096, nch - 1)
        iq24 = f.tap(M.TAP_RDS_IQ, f.last_rds_samples(nch - 1), nch - 1) if rds else None
        del f
        return pcm, out, ring, iq24
    p1, t1, r1, _ = run(1, None)
    assert all(isinstance(t, np.ndarray) for t in t1) and np.abs(t1[0]).max() > 0 and np.abs(t1[1][:, 0]).max() > 0 and len(t1[3]) >= 1
    p70, t70, r70, _ = run(70, None)
    assert all(isinstance(t, str) and "SCOPE_TAPS" in t for t in t70)      # (not kept: the library says so)
    assert np.array_equal(r70, r1)                                            # (the ring taps are always there)
    p70k, t70k, _, _ = run(70, 1)
    assert np.array_equal(p70, p70k) and np.array_equal(p70[69], p1[0])
    assert all(np.array_equal(a, b) for a, b in zip(t70k, t1))
    p1n, t1n, _, _ = run(1, 0)
    assert all(isinstance(t, str) for t in t1n) and np.array_equal(p1n, p1)
    # RDS in a batch without taps: the rows are written for the RDS path, the 24 kS/s baseband is the small handle's
    _, _, _, q1 = run(1, None, rds=True)
    p70r, _, _, q70 = run(70, None, rds=True)
    assert np.array_equal(q70, q1) and np.array_equal(p70r, p70)


def test_rds_block_phases_per_channel_in_a_batch(fmx_amd, ol):
    """A batch whose channels switch their RDS decoders on at different times (and one off and on again): three block phases in one handle, the
    block filters by channel list for the groups that are not the whole handle.  A representative of every group against an oracle chain that
    takes the same switches (24 kS/s baseband, bits); the channels of a group are bit-identical to each other."""
    block, calls, nch = 16384 * 15, 9, 70
    iq = ol.synth_iq(block * calls, rds=1, rdsLevel=0.05, rdsBitsSeed=777)
    f = fmx_amd.Fmx(nch, streams=1, stream_of_channel=[0] * nch, max_block=block)
    for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_FILTER_RESTARTS, 2), (M.P_PLL_SOLVER, 1)):
        f.set_param(pid, v)
    group = lambda c: 0 if c < 40 else (1 if c < 60 else 2)
    join = {0: 0, 1: 2, 2: 3}
    reps = {0: 0, 1: 40, 2: 60, 3: 5}                                 # (3: channel 5, which pauses from call 4 to call 6)
    chains = {g: ol.OracleChain(inputFilterBw=165000, rdsMode=0, taps=[ol.TAP_RDS_IQ], tap_seconds=1.2) for g in reps}
    taps = {g: [] for g in reps}
    on = [False] * nch
    for k in range(calls):
        for c in range(nch):
            if join[group(c)] == k: f.set_param(M.P_RDS_MODE, 2, c); on[c] = True
        for g in reps:
            if join[group(reps[g])] == k: chains[g].configure(rdsMode=2)
        if k == 4: f.set_param(M.P_RDS_MODE, 0, 5); chains[3].configure(rdsMode=0); on[5] = False
        if k == 6: f.set_param(M.P_RDS_MODE, 2, 5); chains[3].configure(rdsMode=2); on[5] = True
        x = iq[k * block:(k + 1) * block]
        f.process_host(x[None])
        for g in reps: chains[g].process(x)
        for g, c in reps.items():
            if on[c]: taps[g].append(f.tap(M.TAP_RDS_IQ, f.last_rds_samples(c), c))
        for c in (1, 39, 41, 59, 61, 69):                               # twins of the representatives 0, 40, 60
            r = reps[group(c)]
            if on[c]:
                assert f.last_rds_samples(c) == f.last_rds_samples(r)
                assert np.array_equal(f.tap(M.TAP_RDS_IQ, f.last_rds_samples(c), c), f.tap(M.TAP_RDS_IQ, f.last_rds_samples(r), r)), (k, c)
    for g, c in reps.items():
        gt = np.concatenate(taps[g]); o = chains[g].tap(ol.TAP_RDS_IQ)
        assert len(gt) == len(o), (g, len(gt), len(o))
        sig = float(np.sqrt(np.mean(o[len(o) // 2:].astype(np.float64) ** 2)))
        e = float(np.sqrt(np.mean((gt.astype(np.float64) - o) ** 2)))
        b_g, b_o = f.rds_bits(c, 8192), chains[g].rds_bits()
        print("\n[RDS batch, channel %d] baseband rms err %.2e (signal %.2e); bits %d / %d" % (c, e, sig, len(b_g), len(b_o)))
        assert sig > 1e-3 and e <= 1e-4 * sig and len(b_g) == len(b_o)
        assert np.count_nonzero(np.nonzero(b_g != b_o)[0] >= 460) <= 2


def test_call_made_in_overlapping_pieces_against_the_oracle(fmx_amd, ol):
    """FMX_P_CALL_PIECES (fmx_api.hip run_call): a batch whose channels run pllC or a squelch -- recurrences that walk a channel's samples one after
    the other, one wave per 64 channels -- is made in pieces on three streams, stage A of piece k + 1 and stage B / C of piece k - 1 running while the
    recurrences walk piece k.  A 70-channel handle forced into pieces of 2048 fm samples (nine per call), one stream, the channels on the PLL
    decoder, the AM decoder, the level and the noise squelch and the default decoder: every kind against an oracle chain fed the same calls whole, at the
    tolerances of the tests that compare a call made whole; channels of a kind bit-identical; the squelch flags call by call; a frequency change, a volume change and a squelch slider
    moved between calls."""
    block = 16384 * 14
    nb = 6
    iq = ol.synth_iq(nb * block, stereo=1, noiseSigma=0.002)
    env = np.ones(nb * block, np.float32)
    env[2 * block:4 * block] = 0.004                                   # (the carrier fades: the level squelch closes and opens again)
    iq = (iq * env[:, None]).astype(np.float32)
    kinds = {0: dict(decoder=2), 1: dict(decoder=1, fmMode=2), 2: dict(squelchMode=2, squelchValue=50), 3: dict(squelchMode=1, squelchValue=60), 4: dict()}
    nch = 70
    f = fmx_amd.Fmx(nch, streams=1, stream_of_channel=[0] * nch, max_block=block)
    for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FM_MODE, 0), (M.P_FM_DECODER, 3)): f.set_param(pid, v)
    f.set_param(M.P_CALL_PIECES, 2048)
    for c in range(nch):
        k = c % 5
        if k == 0: f.set_param(M.P_FM_DECODER, 2, c)
        elif k == 1: f.set_param(M.P_FM_DECODER, 1, c); f.set_param(M.P_FM_MODE, 2, c)
        elif k == 2: f.set_param(M.P_SQUELCH_MODE, 2, c); f.set_param(M.P_SQUELCH_VALUE, 50, c)
        elif k == 3: f.set_param(M.P_SQUELCH_MODE, 1, c); f.set_param(M.P_SQUELCH_VALUE, 60, c)
    chains = {k: ol.OracleChain(inputFilterBw=165000, **kw) for k, kw in kinds.items()}
    pg, po, fl_g, fl_o, pieces = [], {k: [] for k in kinds}, [], [], []
    for b in range(nb):
        # settings and a one-shot action between calls: they take effect with the call's FIRST piece, once (triggerFrequencyChange restarts the PSS
        # analyzer and the fade-in; a volume change runs the gain correction of the call's first frames; the noise squelch's slider)
        if b == 3:
            f.set_param(M.A_TRIGGER_FREQUENCY_CHANGE, 0)
            for k in kinds: chains[k].L.fmo_chain_trigger_frequency_change(chains[k].h)
        if b == 4:
            f.set_param(M.P_VOLUME_DB, -10.5)
            for k in kinds: chains[k].configure(volumeDb=-10.5)
            for c in range(3, nch, 5): f.set_param(M.P_SQUELCH_VALUE, 100, c)
            chains[3].configure(squelchValue=100)
        x = iq[b * block:(b + 1) * block]
        pg.append(f.process_host(x[None]))
        pieces.append(f.last_call_pieces())
        for k in kinds: po[k].append(chains[k].process(x))
        fl_g.append([f.meta(k).squelch_active for k in (2, 3)]); fl_o.append([chains[k].meta().squelchActive for k in (2, 3)])
    pg = np.concatenate(pg, axis=1)
    print("\n[overlapping pieces] pieces per call %s; squelch flags (level, noise) per call %s" % (pieces, fl_g))
    assert all(p == 9 for p in pieces)           # (round 6: the first call too -- the pre-pass's work arrays are there before the call is cut; the last third of a piece rides with the ninth)
    assert fl_g == fl_o and any(r[0] == 1 for r in fl_g) and any(r[1] == 1 for r in fl_g)
    for c in range(5, nch): assert np.array_equal(pg[c], pg[c % 5]), c
    for k in kinds:
        e = float(np.sqrt(np.mean((pg[k].astype(np.float64) - np.concatenate(po[k])) ** 2)))
        print("[overlapping pieces] kind %d %s: pcm rms err %.2e" % (k, kinds[k], e))
        assert e <= 1e-5, (k, e)


@pytest.mark.parametrize("mode", ["pll", "mix"])
def test_overlapping_pieces_equal_the_pieces_one_after_the_other(mode):
    """... and at batch scale (1024 channels on four programmes, three calls of 230400 samples, automatic piece length): the overlapping run against the same
    pieces one after the other on one stream -- the same kernels on the same data, so every channel bit for bit unless a stage of one piece races with a
    stage of another -- and against the calls made whole (the chain's invariance to how a stream is cut: 2e-5 of full scale; 1e-7 measured), the metaData
    snapshots included (tools/diag/pieces_check.py)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "diag", "pieces_check.py"), "1024", "3", "230400", mode, "-1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert r.stdout.strip().splitlines()[-1].endswith("mismatch: 0"), r.stdout[-3000:]


def test_volume_change_on_a_call_that_starts_inside_a_resampler_block(fmx_amd, ol):
    """A volume change between two calls (fm-processor.cpp:299-306, 630: the gain multiplies the sample that enters the resampler) when the call's
    first fm sample is NOT a multiple of 192: the call's frames begin at the resampler block that sample falls into, so the first frames' windows lie
    up to 191 + 127 samples in front of it and frames up to 78 straddle the change.  gain_fix_kernel kept 128 samples and 32 frames until round 5: the
    first frames of such a call summed stale LDS -- 1e-4 on five frames, differently from channel to channel.  130 channels on one stream: every channel
    bit-identical to channel 0, and the first 128 frames behind the change within 1e-5 of the oracle sample by sample (4e-6 measured)."""
    block = 16384 * 14
    iq = ol.synth_iq(3 * block, stereo=1, noiseSigma=0.002)
    nch = 130
    f = fmx_amd.Fmx(nch, streams=1, stream_of_channel=[0] * nch, max_block=block)
    for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FM_MODE, 0), (M.P_FM_DECODER, 3)): f.set_param(pid, v)
    o = ol.OracleChain(inputFilterBw=165000)
    for b in range(3):
        if b == 2: f.set_param(M.P_VOLUME_DB, -10.5); f.set_param(M.P_SOUND_BALANCE, 30); o.configure(volumeDb=-10.5, balance=30)
        pcm = f.process_host(iq[None, b * block:(b + 1) * block])
        po = o.process(iq[b * block:(b + 1) * block])
        assert all(np.array_equal(pcm[c], pcm[0]) for c in range(1, nch)), b
    assert (2 * block // 12) % 192 != 0
    e = float(np.abs(pcm[0][:128].astype(np.float64) - po[:128]).max())
    print("\n[volume change inside a resampler block] first 128 frames: max |err| %.2e (full scale %.2f)" % (e, float(np.abs(po[:128]).max())))
    assert e <= 1e-5 and float(np.abs(po[:128]).max()) > 0.02


@pytest.mark.parametrize("seed,channels,kinds,pieces,extra", [(1, 130, 5, -1, ()), (2, 70, 5, 2048, ()), (3, 130, 5, -1, ("3", "s16")), (4, 1040, 4, -1, ("2", "f32", "1152000"))])
def test_twins_under_runtime_changes(seed, channels, kinds, pieces, extra):
    """tools/diag/twins_setters.py, bounded: a batch in kinds of equal settings on one stream, calls of uneven length (most of them not whole resampler
    blocks), random setters and actions applied to whole kinds between calls; every call every channel's PCM and metaData equal its twin's, the RDS bits at
    the end.  The class of defect it looks for: memory a channel should not have read (gain_fix_kernel's stale LDS, rounds 2-5) and races (round 4).
    Variants: pieces forced on a 70-channel handle; three streams of raw int16 samples; 1040 channels (two stage-B / C channel groups) on two streams at 1.152 MS/s."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "diag", "twins_setters.py"), str(seed), "5" if channels < 1000 else "3", str(channels), str(kinds), "10", str(pieces)] + list(extra),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert r.stdout.strip().splitlines()[-1].endswith("mismatch: 0"), r.stdout[-3000:]


def test_stages_b_and_c_as_two_channel_groups_equal_one_group():
    """A plain batch of more channels than one round of stage-B workgroups runs stages B and C as two channel groups on two streams (fmx_api.hip run_call_one,
    fmx_last_second_group): 1100 channels on four programmes (768 + 332), four calls of uneven length, the first and the one behind a volume change made as one group (the gain correction runs) --
    against the same calls with FMX_TAIL_SPLIT=0, bit for bit; twins equal inside each run."""
    import os, subprocess, sys, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    with tempfile.TemporaryDirectory() as td:
        for flag in ("0", "1"):
            path = os.path.join(td, "g%s.npz" % flag)
            r = subprocess.run([sys.executable, os.path.join(root, "tools", "diag", "groups_check.py"), path, "1100", "4", "100000"],
                               env=dict(os.environ, FMX_TAIL_SPLIT=flag), capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
            res.append(np.load(path))
    one, two = res
    assert list(one["groups"]) == [0, 0, 0, 0] and list(two["groups"]) == [0, 332, 0, 332], (one["groups"], two["groups"])
    assert one["pcm"].shape == two["pcm"].shape and float(np.abs(one["pcm"]).max()) > 0.01
    assert np.array_equal(one["pcm"], two["pcm"])
