"""GPU tests added in round 2: error paths and host-side bookkeeping found by the round-1 review (stalled stage-B
pipeline as a sticky error, resetRds / triggerFrequencyChange reaching the RDS group decoder, one-shot actions surviving
introspection calls, RDS off-everywhere -> on again), all through the C ABI."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

M = importlib.import_module("sdr-j-fm_amd").fmx
PCM_RMS_TOL = 1e-5


def rms(a):
    return float(np.sqrt(np.mean(np.asarray(a, np.float64) ** 2)))


def gui_defaults(f, bw=165000):
    f.set_param(M.P_BANDWIDTH, bw)
    f.set_param(M.P_LF_CUTOFF, 15000)
    f.set_param(M.P_DEEMPHASIS, 50)
    f.set_param(M.P_VOLUME_DB, -6.0)
    f.set_param(M.P_FM_MODE, 0)


def test_forced_stall_is_a_sticky_error(fmx_amd, ol, monkeypatch):
    """The persistent stage-B layout waits on words written by other kernels; a wait that runs out of patience (~2 s) must
    surface as FMX_E_HIP exactly once -- from the host call it happened in, or from the NEXT call / fmx_synchronize of an
    asynchronous caller -- never as FMX_OK with garbage PCM, and the handle must keep working on the event-driven layout.
    FMX_DEBUG_FORCE_STALL makes the start gate ask for one workgroup more than exist.  The PLL decoder keeps the handle on
    the chunked layouts (the fused per-channel kernel has no inter-workgroup waits)."""
    block = 16384
    nb = 24                                         # past the latencies of the input filter (65285 samples) and the audio filter (7436 fm samples)
    iq = ol.synth_iq(nb * block)
    monkeypatch.setenv("FMX_PERSISTENT_MIN_CHANNELS", "1")
    monkeypatch.setenv("FMX_DEBUG_FORCE_STALL", "1")
    f = fmx_amd.Fmx(1, max_block=block)
    gui_defaults(f)
    f.set_param(M.P_FM_DECODER, 2)
    with pytest.raises(fmx_amd.FmxError) as e:
        f.process_host(iq[:block])
    assert e.value.code == M.FMX_E_HIP and "stalled" in str(e.value)
    monkeypatch.delenv("FMX_DEBUG_FORCE_STALL")
    f.synchronize()                                   # reported once: the error is not repeated
    outs = [f.process_host(iq[i:i + block]) for i in range(block, nb * block, block)]     # event-driven layout from here on
    f.synchronize()
    pcm = np.concatenate(outs, axis=1)[0]
    assert np.all(np.isfinite(pcm)) and rms(pcm[-600:]) > 1e-3          # the chain runs again (fade-in under way)

    # asynchronous caller: the stalled call itself returns FMX_OK (nothing is known yet); the NEXT call must refuse
    import torch
    monkeypatch.setenv("FMX_DEBUG_FORCE_STALL", "1")
    g = fmx_amd.Fmx(1, max_block=block)
    gui_defaults(g)
    g.set_param(M.P_FM_DECODER, 2)
    d_iq = torch.from_numpy(iq[:block].copy()).cuda()
    d_pcm = torch.zeros((1, block // 48 + 8, 2), dtype=torch.float32, device="cuda")
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    g.process_device(d_iq.data_ptr(), block, block, d_pcm.data_ptr(), d_pcm.shape[1], hip_stream=st.cuda_stream)
    torch.cuda.synchronize()
    monkeypatch.delenv("FMX_DEBUG_FORCE_STALL")
    with pytest.raises(fmx_amd.FmxError) as e2:
        g.process_device(d_iq.data_ptr(), block, block, d_pcm.data_ptr(), d_pcm.shape[1], hip_stream=st.cuda_stream)
    assert e2.value.code == M.FMX_E_HIP
    g.process_device(d_iq.data_ptr(), block, block, d_pcm.data_ptr(), d_pcm.shape[1], hip_stream=st.cuda_stream)
    g.synchronize()


def test_reset_rds_and_retune_clear_the_programme(fmx_amd, ol):
    """resetRds() -> rdsGroupDecoder::reset (fm-processor.cpp:862-864, rds-groupdecoder.cpp:71-98) clears PI, PTY, the
    station label and the radio text; triggerFrequencyChange() does the same (:849-855).  After a retune to another
    programme the old PS / RT must not be reported until groups of the new PI arrive."""
    block = 16384 * 20
    n = int(2.4 * 2304000) // block * block
    pa = dict(pi=0xD3A1, pty=10, ps="FMX-AMD ", text="HIP KERNELS ON MI355X - RDS OK")
    pb = dict(pi=0x2468, pty=3, ps="CHAN TWO", text="SECOND STREAM")
    iq_a = ol.synth_iq(n, rds=1, rdsLevel=0.05, rds_payload=ol.rds_programme_bits(**pa))
    iq_b = ol.synth_iq(n, rds=1, rdsLevel=0.05, rds_payload=ol.rds_programme_bits(**pb))
    f = fmx_amd.Fmx(1, max_block=block)
    gui_defaults(f)
    f.set_param(M.P_RDS_MODE, 2)
    for i in range(0, n, block):
        f.process_host(iq_a[i:i + block])
    info = f.rds_decode(0)
    assert info.pi_code == pa["pi"] and info.station_label.decode() == pa["ps"] and info.groups_decoded >= 8
    f.set_param(M.A_RESET_RDS, 0)
    info = f.rds_decode(0)
    assert info.pi_code == 0 and info.pty_code == -1 and info.station_label.decode().strip() == "" and info.radio_text.decode() == ""
    assert info.last_group_type == -1
    # same station keeps sending: the programme comes back
    for i in range(0, 6 * block, block):
        f.process_host(iq_a[i:i + block])
    assert f.rds_decode(0).pi_code == pa["pi"]
    # retune: triggerFrequencyChange, then the other programme
    f.set_param(M.A_TRIGGER_FREQUENCY_CHANGE, 0)
    f.process_host(iq_b[:block])
    info = f.rds_decode(0)
    assert info.pi_code in (0, pb["pi"]) and info.station_label.decode() != pa["ps"] and info.radio_text.decode() != pa["text"]
    for i in range(block, n, block):
        f.process_host(iq_b[i:i + block])
    info = f.rds_decode(0)
    assert info.pi_code == pb["pi"] and info.station_label.decode() == pb["ps"] and info.radio_text.decode() == pb["text"]


def test_pending_action_survives_introspection_and_tiny_calls(fmx_amd, ol):
    """A one-shot action (triggerFrequencyChange: fade-in re-armed, PSS restarted) set before an introspection call
    (fmx_get_taps), a failing call (pcm capacity too small) and a call too short to hold one fm sample must still be applied
    by the first call that runs stage B, and only once."""
    block = 16384
    iq = ol.synth_iq(40 * block)
    f = fmx_amd.Fmx(1, max_block=block)
    gui_defaults(f, 0)
    ref = fmx_amd.Fmx(1, max_block=block)
    gui_defaults(ref, 0)
    for i in range(0, 30 * block, block):
        f.process_host(iq[i:i + block]); ref.process_host(iq[i:i + block])
    f.set_param(M.A_TRIGGER_FREQUENCY_CHANGE, 0)
    ref.set_param(M.A_TRIGGER_FREQUENCY_CHANGE, 0)
    assert f.taps(0).size == 37                     # introspection between the setter and the call
    import ctypes as C
    got = C.c_int64()
    one = np.zeros((1, 1, 2), np.float32)
    rc = f.L.fmx_process_host(f.h, iq[30 * block:].ctypes.data_as(C.POINTER(C.c_float)), block, block,
                              one.ctypes.data_as(C.POINTER(C.c_float)), 1, C.byref(got))
    assert rc == M.FMX_E_TOO_LARGE                  # fails AFTER the mailbox was flushed; nothing was processed
    a = [f.process_host(iq[30 * block + i: 30 * block + i + 4]) for i in range(0, 8, 4)]      # 8 samples: no fm sample yet
    assert all(x.shape[1] == 0 for x in a)
    b0 = ref.process_host(iq[30 * block: 30 * block + 8])
    assert b0.shape[1] == 0
    pa = np.concatenate([f.process_host(iq[30 * block + 8 + i: 30 * block + 8 + i + block]) for i in range(0, 8 * block, block)], axis=1)
    pb = np.concatenate([ref.process_host(iq[30 * block + 8 + i: 30 * block + 8 + i + block]) for i in range(0, 8 * block, block)], axis=1)
    assert np.abs(pa[0, :10]).max() < 1e-3          # the fade restarted from 0 in the first call that produced frames
    # ... and as for a handle that went straight to that call: to rounding only -- the two handles reached sample 30 * block + 8 in
    # calls of different lengths, and the RfDC state composes over a call's runs with call-dependent rounding (1e-9), which the
    # correction behind the FIR turns into last-bit differences of the fm-rate samples
    assert np.abs(pa - pb).max() <= 2e-7


def test_rds_off_everywhere_then_on_again(fmx_amd, ol):
    """When every channel switches RDS off, the shared block phase of the RDS front end ends; switching it on again later
    (any decoder) starts from fresh filters and slicer states instead of decoding stale blocks across the gap."""
    block = 16384 * 20
    n = int(2.2 * 2304000) // block * block
    p = dict(pi=0xD3A1, pty=10, ps="FMX-AMD ", text="HIP KERNELS ON MI355X - RDS OK")
    iq = ol.synth_iq(2 * n, rds=1, rdsLevel=0.05, rds_payload=ol.rds_programme_bits(**p))
    f = fmx_amd.Fmx(2, streams=1, stream_of_channel=[0, 0], max_block=block)
    gui_defaults(f)
    f.set_param(M.P_RDS_MODE, 2)
    for i in range(0, block * 4, block):
        f.process_host(iq[i:i + block])
    with pytest.raises(fmx_amd.FmxError):           # one channel off, the other still decoding: it cannot rejoin mid-block ...
        f.set_param(M.P_RDS_MODE, 0, 1); f.process_host(iq[4 * block:5 * block]); f.set_param(M.P_RDS_MODE, 1, 1)
    f.set_param(M.P_RDS_MODE, 0)                    # ... but after RDS went off everywhere
    for i in range(5 * block, 8 * block, block):
        f.process_host(iq[i:i + block])
    f.set_param(M.P_RDS_MODE, 2, 0); f.set_param(M.P_RDS_MODE, 1, 1)      # any decoder may start again
    k0 = 8 * block
    for i in range(k0, k0 + n, block):
        f.process_host(iq[i:i + block])
    for c in range(2):
        info = f.rds_decode(c)
        assert info.synchronized == 1 and info.pi_code == p["pi"] and info.station_label.decode() == p["ps"], c
        assert info.crc_errors <= 2
    # off everywhere and on again with NO call in between (a paused device, a programmatic reconfiguration): the same restart, no
    # error; a consumer that polls only at the very end still gets a fresh synchroniser / decoder picture
    g_before = [f.rds_decode(c).groups_decoded for c in range(2)]
    f.set_param(M.P_RDS_MODE, 0)
    f.set_param(M.P_RDS_MODE, 2)
    for i in range(0, n, block):
        f.process_host(iq[i:i + block])
    for c in range(2):
        info = f.rds_decode(c)
        assert info.synchronized == 1 and info.pi_code == p["pi"] and info.station_label.decode() == p["ps"], c
        assert 8 <= info.groups_decoded < g_before[c] + 8 and info.crc_errors <= 2      # counted from the restart, not on top of the old run


def _run_layout(fmx_amd, monkeypatch, layout, nch, iq, block, setup):
    if layout == "chunked":
        monkeypatch.setenv("FMX_STAGE_B", "chunked")
    else:
        monkeypatch.delenv("FMX_STAGE_B", raising=False)
    f = fmx_amd.Fmx(nch, streams=1, stream_of_channel=[0] * nch, max_block=block)
    gui_defaults(f)
    setup(f)
    outs = [f.process_host(iq[i:i + block]) for i in range(0, len(iq) - block + 1, block)]
    f.synchronize()
    nf = block // 12
    res = dict(pcm=np.concatenate(outs, axis=1), dem=[f.tap(M.TAP_DEMOD, nf, c) for c in range(nch)],
               lr=[f.tap(M.TAP_LR_RAW, nf, c) for c in range(nch)], pre=[f.tap(M.TAP_PRE_RESAMPLER, nf, c) for c in range(nch)],
               meta=[f.meta(c) for c in range(nch)])
    monkeypatch.delenv("FMX_STAGE_B", raising=False)
    return res


@pytest.mark.parametrize("noise", [0.0, 0.004])
def test_fused_stage_b_equals_chunked_layouts(fmx_amd, ol, monkeypatch, noise):
    """The fused per-channel kernel (fmx_stageb.hip) against the chunked lane-per-channel kernels (fmx_demod.hip) on the same
    stage-A output.  The pilot PLL and the PSS integrator are fixed points of the exact f32 trajectory: given the same
    demodulator output they are bit-identical.  The demodulator output itself goes through the AFC, whose state is carried
    across threads by a weighted scan (rounding differs from the sequential evaluation at the 1e-7 level), so the taps agree
    to rounding and the flags / counters exactly.  Mixed settings: four decoders, mono, PSS off, panorama, autoMono off;
    a lock acquisition, several calls of uneven length crossing segment boundaries."""
    nch = 10
    blocks = [16384 * 3, 16384 * 5 + 12 * 77, 16384 * 2, 230400, 16384 * 7, 1200, 16384 * 9]
    reps = 4
    n = sum(blocks) * reps
    kw = dict(noiseSeed=77, noiseSigma=noise) if noise > 0 else {}
    iq = ol.synth_iq(n, **kw)

    def setup(f):
        for c in range(nch):
            f.set_param(M.P_FM_DECODER, (3, 4, 5, 6)[c % 4], c)
        f.set_param(M.P_FM_MODE, 2, 4); f.set_param(M.P_PSS, 0, 5); f.set_param(M.P_FM_MODE, 1, 6); f.set_param(M.P_STEREO_PANORAMA, 60, 6)
        f.set_param(M.P_AUTO_MONO, 0, 7); f.set_param(M.P_SOUND_MODE, 5, 8); f.set_param(M.P_DEEMPHASIS, 75, 9)

    res = {}
    for layout in ("chunked", "fused"):
        if layout == "chunked":
            monkeypatch.setenv("FMX_STAGE_B", "chunked")
        else:
            monkeypatch.delenv("FMX_STAGE_B", raising=False)
        f = fmx_amd.Fmx(nch, streams=1, stream_of_channel=[0] * nch, max_block=max(blocks))
        gui_defaults(f)
        setup(f)
        pcm, metas, taps = [], [], []
        pos = 0
        for rep in range(reps):
            for b in blocks:
                pcm.append(f.process_host(iq[pos:pos + b])); pos += b
                metas.append([m.live_pilot_locked for m in (f.meta(c) for c in range(nch))])
                nf = min(b // 12, 500)
                taps.append([(f.tap(M.TAP_DEMOD, nf, c), f.tap(M.TAP_LR_RAW, nf, c)) for c in (0, 3, 7)])
        f.synchronize()
        res[layout] = (np.concatenate(pcm, axis=1), metas, taps, [f.meta(c) for c in range(nch)])
        monkeypatch.delenv("FMX_STAGE_B", raising=False)
    pa, ma, ta, fa = res["chunked"]; pb, mb, tb, fb = res["fused"]
    assert pa.shape == pb.shape
    assert ma == mb                                        # lock / PSS flags call by call, every channel
    errs = [rms(pa[c] - pb[c]) for c in range(nch)]
    worst = max(errs)
    print("\n[fused vs chunked] PCM RMS difference per channel:", " ".join("%.1e" % e for e in errs))
    for c in range(nch):
        # (channel 7: DIFF decoder with autoMono off decodes L-R during the pilot pull-in, where a 1e-5 rad difference of the PLL
        # phase is not second order: 3e-6; everything else stays below 1e-6)
        assert errs[c] <= (5e-6 if c == 7 else 2e-6), (c, errs)
        # (the fused kernels take the metaData snapshot at the reference's own sample, the chunked layout at the end of that call:
        # the slowly moving values differ by what they moved in between)
        assert abs(fa[c].PssPhaseShiftDegree - fb[c].PssPhaseShiftDegree) < 2e-2 and fa[c].PssState == fb[c].PssState
        assert abs(fa[c].PilotPllLockStrength - fb[c].PilotPllLockStrength) < 2e-3
    for xa, xb in zip(ta, tb):
        for (da, la), (db, lb) in zip(xa, xb):
            # (the raw L-R tap is 2 cos(table[idx]) demod at fm rate: the two PLL phases differ by ~1e-5 rad, 2x that against the
            # table's 3.3e-5 rad steps lands on the neighbouring entry in about half the samples -- white, gone behind the audio filter)
            assert rms(da - db) <= 5e-6 * max(1.0, float(np.abs(da).max())) and rms(la - lb) <= 1e-4
    print(f"\n[fused vs chunked, noise {noise}] worst PCM RMS difference {worst:.3e}")
    assert rms(pa[0]) > 0.01 and fa[0].PilotPllLocked == 1


def test_meta_snapshot_is_taken_at_the_reference_sample(fmx_amd, ol):
    """showMetaData (fm-processor.cpp:662-684): the reference snapshots its state behind every 96001st fm sample, inside the block
    loop.  With 0.1 s device-style calls (19200 fm samples) the snapshot sample lies somewhere inside a call; the fused kernels
    store the values of exactly that sample, so after every call the library's picture equals the oracle's last snapshot --
    during the pilot pull-in and the PSS swing-in, where the values move from sample to sample."""
    block = 16384 * 14                                        # 0.0996 s; whole reference blocks, so that the oracle has seen exactly the same samples
    calls = 22
    iq = ol.synth_iq(block * calls)
    f = fmx_amd.Fmx(1, max_block=block)
    gui_defaults(f)
    o = ol.OracleChain(inputFilterBw=165000)
    seen = 0
    for k in range(calls):
        x = iq[k * block:(k + 1) * block]
        f.process_host(x); o.process(x)
        a, b = f.meta(0), o.meta()
        if b.pilotLockStrength != 0.0:
            seen += 1
        assert a.PilotPllLocked == b.pilotLocked and a.PssState == b.pssState, k
        assert abs(a.PilotPllLockStrength - b.pilotLockStrength) <= 2e-5, (k, a.PilotPllLockStrength, b.pilotLockStrength)
        assert abs(a.DcValIf - b.dcValIf) <= 2e-6 + 1e-4 * abs(b.dcValIf), (k, a.DcValIf, b.dcValIf)
        assert abs(a.PssPhaseShiftDegree - b.pssPhaseShiftDegree) <= 2e-3, (k, a.PssPhaseShiftDegree, b.pssPhaseShiftDegree)
        assert abs(a.PssPhaseChange - b.pssPhaseChange) <= 2e-3 + 1e-3 * abs(b.pssPhaseChange), (k, a.PssPhaseChange, b.pssPhaseChange)
    assert seen >= 15


def test_full_size_config3_shared_wideband_streams(fmx_amd, ol):
    """BASELINE configs[2] at its full layout: 256 channels on 24 wide-band IQ streams, eleven (ten on the last streams) carriers
    per stream on a 200 kHz raster selected with set_localOscillator, every channel of a stream reading the SAME buffer
    (stream_of_channel).  The 24 streams carry three different contents (stream s = content s % 3), so (i) every channel must be
    bit-identical to the channel with the same carrier in the first stream of its content, and (ii) channels spot-checked against
    the oracle (which mixes the same wide-band stream with the same LO) stay within the north-star tolerance.  0.1 s device-sized
    calls through process_host, six calls."""
    block, calls, C, S = 230400, 6, 256, 24
    n = block * calls
    offs = [(k - 5) * 200000 for k in range(11)]
    contents = []
    for j in range(3):
        wide = np.zeros((n, 2), np.float64)
        for k, o in enumerate(offs):
            wide += ol.synth_iq(n, offsetHz=float(o), leftHz=300.0 + 170 * k + 40 * j, rightHz=800.0 + 90 * k + 25 * j, carrierAmp=0.085)
        contents.append(wide.astype(np.float32))
    smap = [c * S // C for c in range(C)]                      # 10 or 11 channels per stream, as bench.py's config3
    first_of_stream = {}
    lo = []
    for c in range(C):
        first_of_stream.setdefault(smap[c], c)
        lo.append(offs[c - first_of_stream[smap[c]]])
    f = fmx_amd.Fmx(C, streams=S, stream_of_channel=smap, max_block=block)
    gui_defaults(f)
    for c in range(C):
        f.set_param(M.P_LOCAL_OSCILLATOR, lo[c], c)
    iq = np.stack([contents[s % 3] for s in range(S)])          # [24, n, 2]
    pcm = np.concatenate([f.process_host(iq[:, i * block:(i + 1) * block]) for i in range(calls)], axis=1)
    assert pcm.shape[0] == C
    # (i) duplicates: channel c against the channel of stream (s % 3) with the same carrier index
    ref_of = {}
    for c in range(C):
        key = (smap[c] % 3, lo[c])
        if key in ref_of:
            assert np.array_equal(pcm[c], pcm[ref_of[key]]), (c, ref_of[key])
        else:
            ref_of[key] = c
    assert len(ref_of) >= 30                                     # (10 or 11 carriers in use per content)
    # (ii) oracle spot checks: an edge carrier, the centre one, one of each content
    for c in (0, 5, first_of_stream[1] + 10, first_of_stream[2] + 3, C - 1):
        want = ol.OracleChain(inputFilterBw=165000, loFrequency=int(lo[c])).process(contents[smap[c] % 3])
        m = want.shape[0]
        assert pcm[c].shape[0] - 16384 // 48 - 1 <= m <= pcm[c].shape[0]
        e = rms(pcm[c][:m] - want)
        assert e <= 1e-5, (c, lo[c], e)
    assert rms(pcm[0][-4800:]) > 1e-3


def test_front_pairs_layout_matches_classic_and_oracle(fmx_amd, ol, monkeypatch):
    """The opt-in stage-A layout of fmx_front2.hip (FMX_FRONT=pairs: producer / consumer wave pairs, the polyphase FIR as
    v_mfma_f32_16x16x4_f32 in Toeplitz form) against the default four-waves-per-channel kernel and the oracle: six channels
    (two workgroups, the second half empty) on two streams, DC offset, an LO shift on one channel, calls of uneven length
    (partial first / last tiles), input filter on."""
    nch = 6
    blocks = [16384 * 3, 16384 * 5 + 12 * 77 + 5, 1200, 16384 * 9, 230400]
    n = sum(blocks)
    iq = np.stack([ol.synth_iq(n), ol.synth_iq(n, stereo=0)], axis=0)
    iq[0, :, 0] += 0.004; iq[0, :, 1] -= 0.003              # a DC offset the RF DC removal has to track
    res = {}
    for layout in ("classic", "pairs"):
        monkeypatch.setenv("FMX_FRONT", layout)
        f = fmx_amd.Fmx(nch, streams=2, stream_of_channel=[c % 2 for c in range(nch)], max_block=max(blocks))
        gui_defaults(f)
        f.set_param(M.P_LOCAL_OSCILLATOR, 200000, 5)
        pcm, pos = [], 0
        for b in blocks:
            pcm.append(f.process_host(np.ascontiguousarray(iq[:, pos:pos + b]))); pos += b
        f.synchronize()
        res[layout] = (np.concatenate(pcm, axis=1), f.tap(M.TAP_FM_IQ, 4000, 0), f.tap(M.TAP_FM_IQ, 4000, 5))
    monkeypatch.delenv("FMX_FRONT")
    pa, za, wa = res["classic"]; pb, zb, wb = res["pairs"]
    assert pa.shape == pb.shape
    for c in range(nch):
        assert rms(pa[c] - pb[c]) <= 2e-6, (c, rms(pa[c] - pb[c]))
    assert rms(za - zb) <= 2e-6 * rms(za) and rms(wa - wb) <= 5e-7      # the fm-rate ring: summation order only (channel 5 is shifted out of its band: residue)
    assert np.array_equal(pb[0], pb[2]) and np.array_equal(pb[0], pb[4])                     # same stream, same settings
    ch = ol.OracleChain(inputFilterBw=165000)
    pcm_o = ch.process(iq[0])
    m = min(pb.shape[1], pcm_o.shape[0])                     # (the oracle works in the reference's 16384-sample blocks: its tail is still pending)
    assert m > 10000 and rms(pb[0][:m] - pcm_o[:m]) <= PCM_RMS_TOL


@pytest.mark.parametrize("audio_rate", [44100, 96000, 32000])
def test_second_converter_audio_rate(fmx_amd, ol, audio_rate):
    """audioRate != workingRate (main.cpp:57-65 -m; theConverter fm-processor.cpp:89-91, sendSampletoOutput :825-838): the 48 kHz
    frames go through the second converter -- like the first one libsamplerate in the reference, the documented fmx design
    (p / q polyphase Kaiser sinc) in the oracle and on the GPU.  Frame counts call by call and PCM against the oracle."""
    blocks = [16384 * 4, 16384 * 4 + 12 * 100, 230400, 16384 * 3, 16384 * 5 - 1200]
    n = sum(blocks)
    iq = ol.synth_iq(n)
    ch = ol.OracleChain(inputFilterBw=165000, audioRate=audio_rate)
    pcm_o = ch.process(iq)
    f = fmx_amd.Fmx(2, streams=1, stream_of_channel=[0, 0], max_block=max(blocks), audioRate=audio_rate)
    gui_defaults(f)
    f.set_param(M.P_VOLUME_DB, -12.0, 1)                 # channel 1 six dB down: the converter is linear and per channel
    outs, pos = [], 0
    for b in blocks:
        want = f.frames_for(b)
        o = f.process_host(iq[pos:pos + b]); pos += b
        assert o.shape[1] == want
        outs.append(o)
    pcm_g = np.concatenate(outs, axis=1)
    g = int(np.gcd(48000, audio_rate)); p_, q_ = audio_rate // g, 48000 // g
    frames48 = 48 * ((n // 12) // 192)
    assert pcm_g.shape[1] == (frames48 * p_ + q_ - 1) // q_ == f.meta(0).pcm_frames
    m = min(pcm_g.shape[1], pcm_o.shape[0])
    assert m > 0.9 * pcm_g.shape[1]
    e = rms(pcm_g[0][:m] - pcm_o[:m])
    print(f"\n[second converter {audio_rate}] {pcm_g.shape[1]} frames, rms diff vs oracle {e:.3e} (signal {rms(pcm_o):.3f})")
    assert e <= PCM_RMS_TOL and rms(pcm_o[m // 2:m]) > 0.01
    k = 10.0 ** (-6.0 / 20.0)
    assert rms(pcm_g[1]) > 0.005 and rms(pcm_g[1] - k * pcm_g[0]) <= 1e-6


def test_config1_cpp_file_source_realtime(fmx_amd, ol, tmp_path):
    """BASELINE configs[0] through the C++ host side only: a PCM16 stereo .wav at 2.304 MS/s -> fmx_host::FileSource (fileHulp's
    paced reader thread, real time ON) -> fmx_host::FmProcessor -> sink, mono FM, input filter off; PCM against the oracle fed
    with the floats libsndfile would deliver, and the wall time of the run against the signal's duration."""
    import os, struct, subprocess, time
    host = os.path.join(os.path.dirname(fmx_amd.__file__), "host")
    exe = str(tmp_path / "adapter_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", os.path.join(host, "adapter_demo.cpp"),
                           "-L" + os.path.dirname(fmx_amd.LIB_PATH), "-lfmx", "-Wl,-rpath," + os.path.dirname(fmx_amd.LIB_PATH), "-o", exe])
    blocks = 30
    n = 16384 * blocks
    iq = ol.synth_iq(n, stereo=0)
    s16 = np.clip(np.round(iq * 32768.0 * 0.9), -32768, 32767).astype(np.int16)
    body = s16.astype("<i2").tobytes()
    with open(tmp_path / "c1.wav", "wb") as f:
        f.write(struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(body), b"WAVE", b"fmt ", 16, 1, 2, 2304000, 2304000 * 4, 4, 16, b"data", len(body)) + body)
    t0 = time.time()
    subprocess.check_call([exe, str(tmp_path / "c1.wav"), str(tmp_path / "pcm.f32"), str(blocks), "1"], stdout=subprocess.DEVNULL)
    wall = time.time() - t0
    pcm = np.fromfile(str(tmp_path / "pcm.f32"), np.float32).reshape(-1, 2)
    want = ol.OracleChain(inputFilterBw=0, fmMode=2).process((s16.astype(np.float32) / np.float32(32768.0)).astype(np.float32))
    assert pcm.shape == want.shape and rms(pcm - want) <= PCM_RMS_TOL
    assert wall >= 0.9 * n / 2304000.0                         # paced: 0.21 s of signal does not arrive faster than real time


def test_lo_switched_on_mid_stream_with_dc_offset(fmx_amd, ol):
    """set_localOscillator going from 0 to a fine-tuning offset in the middle of a stream that carries a DC offset (what the GUI's AFC
    loop does, radio.cpp:1786-1809): channels without an LO keep their stage-A history raw and correct RfDC behind the FIR, channels with
    one keep it DC-corrected and mixed -- at the switch the raw history is converted, so the filter memory holds what the reference's
    holds and no click appears.  PCM against the oracle over the switch."""
    block = 16384
    nb = 40
    iq = ol.synth_iq(nb * block)
    iq[:, 0] += 0.006; iq[:, 1] -= 0.004
    ch = ol.OracleChain(inputFilterBw=165000)
    f = fmx_amd.Fmx(1, max_block=block)
    gui_defaults(f)
    po, pg = [], []
    for b in range(nb):
        if b == 24:
            ch.configure(loFrequency=3000); f.set_param(M.P_LOCAL_OSCILLATOR, 3000)
        po.append(ch.process(iq[b * block:(b + 1) * block])); pg.append(f.process_host(iq[b * block:(b + 1) * block])[0])
    po, pg = np.concatenate(po), np.concatenate(pg)
    assert po.shape == pg.shape
    k = 24 * block // 48                                     # frames in front of the switch
    e_all, e_sw = rms(pg - po), rms(pg[k - 200:k + 600] - po[k - 200:k + 600])
    print(f"\n[LO on mid-stream] rms diff all {e_all:.3e}, around the switch {e_sw:.3e}, max {np.abs(pg - po).max():.3e}")
    assert e_all <= PCM_RMS_TOL and e_sw <= PCM_RMS_TOL


def test_random_settings_batch_against_oracle(fmx_amd, ol):
    """Twelve channels with settings drawn at random (seeded) -- input filter width / off, IQ balance, LO offset, DC removal on / off,
    decoder, mode, selector, panorama, de-emphasis, volume, audio filter -- on three streams with different DC offsets and noise, calls of
    uneven length (not multiples of 12 or 192): every channel against an oracle chain with the same settings.  Covers the combinations
    the single-setting tests do not: IQ balance with and without an LO next to DC removal behind / in front of the FIR, tap sets of
    different widths in one batch."""
    rng = np.random.default_rng(2026)
    nch, nstreams = 12, 3
    blocks = [16384 * 4 + 7, 16384 * 6 - 5, 230400, 12 * 1000 + 1, 16384 * 9 + 100, 16384 * 5]
    n = sum(blocks)
    streams = []
    for sidx in range(nstreams):
        x = ol.synth_iq(n, stereo=1 if sidx != 1 else 0, noiseSeed=100 + sidx, noiseSigma=0.002 * sidx)
        x[:, 0] += (0.0, 0.007, -0.02)[sidx]; x[:, 1] += (0.0, -0.004, 0.015)[sidx]      # the last one beyond the +-0.01 limiter
        streams.append(x)
    iq = np.stack(streams, axis=0)
    bw_choices = [0, 165000, 130000, 200000]
    cfgs = []
    for c in range(nch):
        kw = dict(inputFilterBw=int(rng.choice(bw_choices)), attL=float(rng.choice([1.0, 0.9, 1.15])), attR=float(rng.choice([1.0, 1.1, 0.85])),
                  loFrequency=int(rng.choice([0, 0, 2500, -4000])), dcRemove=int(rng.choice([1, 1, 1, 0])), decoder=int(rng.choice([3, 4, 5, 6])),
                  fmMode=int(rng.choice([0, 0, 1, 2])), soundSelector=int(rng.choice([0, 1, 4])), panorama=int(rng.choice([100, 60, 140])),
                  deemphasis=int(rng.choice([50, 75])), volumeDb=float(rng.choice([-6.0, -10.5, 0.0])), lfCutoff=int(rng.choice([15000, 12000, 0])),
                  autoMono=int(rng.choice([1, 0])))
        cfgs.append(kw)
    f = fmx_amd.Fmx(nch, streams=nstreams, stream_of_channel=[c % nstreams for c in range(nch)], max_block=max(blocks))
    pid = dict(inputFilterBw=M.P_BANDWIDTH, attL=M.P_ATTENUATION_L, attR=M.P_ATTENUATION_R, loFrequency=M.P_LOCAL_OSCILLATOR, dcRemove=M.P_DC_REMOVE,
               decoder=M.P_FM_DECODER, fmMode=M.P_FM_MODE, soundSelector=M.P_SOUND_MODE, panorama=M.P_STEREO_PANORAMA, deemphasis=M.P_DEEMPHASIS,
               volumeDb=M.P_VOLUME_DB, lfCutoff=M.P_LF_CUTOFF, autoMono=M.P_AUTO_MONO)
    for c, kw in enumerate(cfgs):
        for k, v in kw.items():
            f.set_param(pid[k], v, c)
    outs, pos = [], 0
    for b in blocks:
        outs.append(f.process_host(np.ascontiguousarray(iq[:, pos:pos + b]))); pos += b
    pcm = np.concatenate(outs, axis=1)
    worst = 0.0
    for c, kw in enumerate(cfgs):
        po = ol.OracleChain(**kw).process(iq[c % nstreams])
        m = min(pcm.shape[1], po.shape[0])
        e = rms(pcm[c][:m] - po[:m])
        worst = max(worst, e)
        assert m > 0.95 * pcm.shape[1] and e <= PCM_RMS_TOL, (c, kw, e)
    print(f"\n[random settings] worst PCM RMS difference over {nch} channels {worst:.3e}")
