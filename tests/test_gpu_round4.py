"""Round-4 GPU tests: stage A with its channels split in time (FMX_P_FRONT_PARTS, fmx_front.hip): the results must not depend on the split."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
M = importlib.import_module("sdr-j-fm_amd").fmx


def _handle(fmx_amd, nch, nst, max_block, parts, restarts, lo, dc_remove=1):
    f = fmx_amd.Fmx(nch, streams=nst, stream_of_channel=[c % nst for c in range(nch)], max_block=max_block)
    for pid, v in ((M.P_BANDWIDTH, 165000), (M.P_LF_CUTOFF, 15000), (M.P_DEEMPHASIS, 50), (M.P_VOLUME_DB, -6.0), (M.P_FM_MODE, 0)):
        f.set_param(pid, v)
    f.set_param(M.P_FILTER_RESTARTS, restarts)
    f.set_param(M.P_FRONT_PARTS, parts)
    f.set_param(M.P_DC_REMOVE, dc_remove)
    for c, o in enumerate(lo):
        if o:
            f.set_param(M.P_LOCAL_OSCILLATOR, int(o), c)
    return f


@pytest.mark.parametrize("restarts", [2, 1])
def test_front_parts_give_identical_results(fmx_amd, ol, restarts):
    """Five channels on two streams (a DC offset on the streams, local oscillators on two channels -- the per-sample pass with its RF DC chain --,
    RF DC removal behind the FIR on the others), calls of uneven lengths that are not multiples of 12 or of the 1536-sample tile: one workgroup per
    channel against the automatic split against a forced split into 5 and into 32 parts -- PCM, fm-rate IQ and the RF DC value bit for bit.
    restarts 2: the folded filters (stage A does RF DC removal, balance and mix itself); 1: the block machines of handles up to 64 channels
    (stage A reads the streams pre_kernel has made)."""
    blocks = [230400, 16384 * 3 + 7, 100001, 230400, 1536 * 9, 20000, 230399]
    n = sum(blocks)
    iq = np.stack([ol.synth_iq(n, leftHz=400.0 + 300 * k, rightHz=700.0 + 200 * k) for k in range(2)])
    iq[0] += np.array([0.004, -0.003], np.float32)
    iq[1] += np.array([-0.02, 0.015], np.float32)             # (beyond the +-0.01 limiter)
    lo = [0, 200000, 0, -400000, 0]
    outs = []
    for parts in (1, 0, 5, 32):
        f = _handle(fmx_amd, 5, 2, max(blocks), parts, restarts, lo)
        pcm, taps, dcs = [], [], []
        pos = 0
        for b in blocks:
            pcm.append(f.process_host(iq[:, pos:pos + b])); pos += b
            nt = f.last_fm_samples()
            taps.append(np.stack([f.tap(M.TAP_FM_IQ, nt, c) for c in range(5)]))
            dcs.append([f.meta(c).DcValRf for c in range(5)])
        outs.append((np.concatenate(pcm, axis=1), np.concatenate(taps, axis=1), np.array(dcs)))
        del f
    ref = outs[0]
    assert np.isfinite(ref[0]).all() and float(np.abs(ref[0]).max()) > 0.01
    for k, o in enumerate(outs[1:]):
        assert np.array_equal(o[1], ref[1]), "fm-rate IQ differs (variant %d)" % (k + 1)
        assert np.array_equal(o[0], ref[0]), "PCM differs (variant %d)" % (k + 1)
        assert np.array_equal(o[2], ref[2]), "RF DC value differs (variant %d)" % (k + 1)


@pytest.mark.parametrize("fmt", ["u8", "s16"])
def test_front_parts_raw_formats(fmx_amd, ol, fmt):
    """Raw integer samples (converted while stage A loads them; the tile maps of the split convert them the same way)."""
    blocks = [230400, 100000, 230400]
    n = sum(blocks)
    x = ol.synth_iq(n)
    if fmt == "u8":
        raw = np.clip(np.round(x * 100.0 + 127.4), 0, 255).astype(np.uint8); code = M.IQ_U8
    else:
        raw = np.clip(np.round(x * 1500.0 + 9.0), -32768, 32767).astype(np.int16); code = M.IQ_S16
    outs = []
    for parts in (1, 0, 7):
        f = _handle(fmx_amd, 2, 1, max(blocks), parts, 2, [0, 250000])
        pcm, pos = [], 0
        for b in blocks:
            pcm.append(f.process_host_raw(raw[None, pos:pos + b], code)); pos += b
        outs.append(np.concatenate(pcm, axis=1))
        del f
    assert float(np.abs(outs[0]).max()) > 0.01
    assert np.array_equal(outs[1], outs[0]) and np.array_equal(outs[2], outs[0])


def test_long_calls_with_rds_on_are_made_in_pieces(fmx_amd, ol):
    """One launch sequence of the RDS front end covers one 32000-sample block of its filters (384000 input samples); a longer call is made in
    pieces inside the library (VERDICT r3 missing #4).  Two channels, RDS_2 on, 0.4 s in calls of 230400 samples for both handles (the
    slicer's loops pull in: the first ~450 bits are decided on a constellation that is still turning, where 1e-7 decides); then one handle
    takes the next 0.4 s in ONE call of 921600 samples, the other in four of 230400: the same PCM (to the chain's block-size invariance,
    2e-7) and the same RDS bits."""
    n = 921600
    iq = np.stack([ol.synth_iq(2 * n, rds=1, rdsLevel=0.05, rdsBitsSeed=sd) for sd in (5, 6)])
    res = []
    for blocks in ([230400] * 4 + [n], [230400] * 8):
        f = _handle(fmx_amd, 2, 2, n, 0, 2, [0, 0])
        f.set_param(M.P_RDS_MODE, 2)
        pcm, pos = [], 0
        for b in blocks:
            pcm.append(f.process_host(iq[:, pos:pos + b])); pos += b
        res.append((np.concatenate(pcm, axis=1), [f.rds_bits(c, 8192) for c in range(2)], f.last_fm_samples()))
        del f
    (p1, b1, l1), (p4, b4, l4) = res
    assert p1.shape == p4.shape and float(np.sqrt(np.mean((p1.astype(np.float64) - p4) ** 2))) <= 2e-7
    assert all(len(a) > 900 and len(a) == len(b) and np.array_equal(a, b) for a, b in zip(b1, b4))
    assert l1 == (n - 2 * 383988) // 12 and l4 == 19200            # (the taps of the long call hold its last piece)


def test_rds_decoders_switched_on_channel_by_channel(fmx_amd, ol):
    """setfmRdsSelector per processor and at any time (fm-processor.cpp:840-847; VERDICT r3 missing #4, r4 missing #2): three channels on one
    stream, channel 0 decodes RDS from the start, channel 1 switches its decoder on 0.3 s later, channel 2 at 0.5 s, and channel 0 goes off at
    0.6 s while the others go on -- each against an oracle chain that takes the same calls and the same switch.  Round 5: a channel's RDS path
    counts its OWN samples -- block boundaries, the phase delay line and the /8 phase are the processor's own in the reference, where all of it
    runs only while the decoder is on (:733-754, :551-553) -- so the late channels' 24 kS/s baseband agrees with their oracles as channel 0's
    does (3e-5 of the sub-carrier; 1.3e-3 while the batch had one block phase: the Hilbert filter is a frequency-domain mask whose impulse
    response wraps around the block, fft-filters.cpp:166-190), PCM within the tolerance, the bit streams equal once the slicer has pulled in."""
    block, calls = 16384 * 15, 15            # (the oracle takes whole 16384-sample blocks; 20480 fm samples per call)
    join = {0: 0, 1: 3, 2: 5}
    off0 = 6
    n = block * calls
    iq = ol.synth_iq(n, rds=1, rdsLevel=0.05, rdsBitsSeed=4242)
    f = _handle(fmx_amd, 3, 1, block, 0, 2, [0, 0, 0])
    chains = [ol.OracleChain(inputFilterBw=165000, rdsMode=0, taps=[ol.TAP_RDS_IQ], tap_seconds=1.7) for _ in range(3)]
    worst = [0.0] * 3
    taps_g = [[] for _ in range(3)]
    for k in range(calls):
        for c in range(3):
            if join[c] == k:
                f.set_param(M.P_RDS_MODE, 2, c); chains[c].configure(rdsMode=2)
        if k == off0:
            f.set_param(M.P_RDS_MODE, 0, 0); chains[0].configure(rdsMode=0)
        x = iq[k * block:(k + 1) * block]
        pg = f.process_host(x[None])
        for c in range(3):
            po = chains[c].process(x)
            assert pg[c].shape == po.shape
            worst[c] = max(worst[c], float(np.sqrt(np.mean((pg[c].astype(np.float64) - po) ** 2))))
            if k >= join[c] and not (c == 0 and k >= off0):
                taps_g[c].append(f.tap(M.TAP_RDS_IQ, f.last_rds_samples(c), c))
    print("\n[RDS channel by channel] worst PCM rms per channel:", " ".join("%.1e" % w for w in worst))
    assert max(worst) <= 1e-5
    for c in (1, 2):
        g = np.concatenate(taps_g[c]); o = chains[c].tap(ol.TAP_RDS_IQ)[:len(g)]
        sig = float(np.sqrt(np.mean(o[len(o) // 2:].astype(np.float64) ** 2)))
        e = float(np.sqrt(np.mean((g.astype(np.float64) - o) ** 2)))
        b_g, b_o = f.rds_bits(c, 8192), chains[c].rds_bits()
        where = np.nonzero(b_g[:min(len(b_g), len(b_o))] != b_o[:min(len(b_g), len(b_o))])[0]
        print("[RDS channel %d, on from call %d] baseband rms err %.2e (signal %.2e); bits %d / %d, differ at %s" % (c, join[c], e, sig, len(b_g), len(b_o), where.tolist()[:12]))
        assert len(g) == len(o) and sig > 1e-3 and e <= 1e-4 * sig
        assert len(b_g) == len(b_o) and len(b_o) > 900
        # (the first ~400 bits are decided on the filters' numerical dust -- 1e-9 here, where channel 0's spectrum leaks into its pair
        # partner's row; exact zeros in the oracle -- and on the slicer's pull-in)
        assert np.count_nonzero(where >= 460) <= 2
    b0_g, b0_o = f.rds_bits(0, 8192), chains[0].rds_bits()
    assert len(b0_g) == len(b0_o) and np.count_nonzero(b0_g[460:] != b0_o[460:]) <= 2        # (channel 0: what it decoded until it went off)
