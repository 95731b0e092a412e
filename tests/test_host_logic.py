"""Host-side logic that needs no GPU: the fmProcessor-shaped mirror (setter names/argument parsing as in
includes/fm/fm-processor.h:104-156), the block loop of fmProcessor::run(), and the frame-count rule."""
import numpy as np
import pytest


class FakeFmx:
    def __init__(self):
        self.calls = []
        self.blocks = []

    def set_param(self, pid, value, channel=-1):
        self.calls.append((pid, value, channel))

    def process_host(self, iq):
        self.blocks.append(np.asarray(iq).shape)
        return np.zeros((1, 341, 2), np.float32)

    def meta(self, channel=0):
        class M:
            PilotPllLocked, PilotPllLockStrength, DcValIf = 1, 0.35, 0.01
            live_pilot_locked, live_lock_strength, live_dc_if = 1, 0.35, 0.01
        return M()


class FakeDevice:
    """deviceHandler-shaped source (devices/device-handler.h:71-74): Samples() / getSamples(n)."""
    def __init__(self, n):
        self.left = n

    def Samples(self):
        return self.left

    def getSamples(self, n):
        self.left -= n
        return np.zeros((n, 2), np.float32)


class FakeSink:
    """audioSink-shaped (includes/output/audiosink.h:45)."""
    def __init__(self):
        self.frames = 0

    def putSamples(self, pcm):
        self.frames += pcm.shape[0]


def test_setters_map_to_parameter_ids(fmx_amd):
    m = fmx_amd.fmx
    f = FakeFmx()
    p = fmx_amd.FmProcessor(fmx=f, channel=3)
    p.setBandwidth("165kHz"); p.setBandwidth("Off"); p.setBandwidth("82kHz")
    assert [c[1] for c in f.calls] == [165000, 0, 82000] and all(c[0] == m.P_BANDWIDTH and c[2] == 3 for c in f.calls)
    f.calls.clear()
    p.setfmMode("Mono"); p.setFMdecoder("FM PLL Decoder"); p.setFMdecoder("PLL Decoder")   # unknown name -> default: PLL
    p.setDeemphasis(50); p.setVolume(-6.0); p.setlfcutoff(15000); p.setSoundBalance(-20); p.setStereoPanorama(150)
    p.setAttenuation(0.9, 1.1); p.set_localOscillator(-200000); p.setAutoMonoMode(False); p.setPSSMode(True)
    p.setDCRemove(True); p.triggerFrequencyChange(); p.restartPssAnalyzer(); p.setSoundMode(4)
    got = [(c[0], c[1]) for c in f.calls]
    assert got == [(m.P_FM_MODE, 2), (m.P_FM_DECODER, 2), (m.P_FM_DECODER, 2), (m.P_DEEMPHASIS, 50), (m.P_VOLUME_DB, -6.0),
                   (m.P_LF_CUTOFF, 15000), (m.P_SOUND_BALANCE, -20), (m.P_STEREO_PANORAMA, 150),
                   (m.P_ATTENUATION_L, 0.9), (m.P_ATTENUATION_R, 1.1), (m.P_LOCAL_OSCILLATOR, -200000),
                   (m.P_AUTO_MONO, 0), (m.P_PSS, 1), (m.P_DC_REMOVE, 1), (m.A_TRIGGER_FREQUENCY_CHANGE, 0),
                   (m.A_RESTART_PSS, 0), (m.P_SOUND_MODE, 4)]
    assert p.isPilotLocked() == (True, 0.35)


def test_run_block_pulls_whole_blocks_only(fmx_amd):
    """fm-processor.cpp:388: the loop waits until Samples() >= 16384 and never consumes a partial block."""
    f, dev, sink = FakeFmx(), FakeDevice(16384 * 3 + 100), FakeSink()
    p = fmx_amd.FmProcessor(theDevice=dev, mySink=sink, fmx=f)
    n = 0
    while p.run_block():
        n += 1
    assert n == 3 and dev.left == 100 and sink.frames == 3 * 341
    assert f.blocks == [(16384, 2)] * 3


def frames_model(g, n):
    j0, j1 = g // 12, (g + n) // 12
    return 48 * (j1 // 192) - 48 * (j0 // 192)


def test_frame_count_rule_matches_oracle(ol):
    """192 fm samples in -> 48 PCM frames out (newconverter.cpp:55-80); the oracle only consumes whole 16384 blocks."""
    ch = ol.OracleChain(inputFilterBw=0)
    g = 0
    for k in (1, 3, 2, 7):
        n = 16384 * k
        got = ch.process(np.zeros((n, 2), np.float32)).shape[0]
        assert got == frames_model(g, n)
        g += n
    assert frames_model(0, 16384) == 336 and frames_model(16384, 16384) == 336 and frames_model(0, 230400) == 4800


def _write_wav(path, x, rate, fmt):
    import struct
    ch = x.shape[1]
    if fmt == "pcm16":
        body, tag, bits = x.astype("<i2").tobytes(), 1, 16
    elif fmt == "pcm8":
        body, tag, bits = x.astype(np.uint8).tobytes(), 1, 8
    else:
        body, tag, bits = x.astype("<f4").tobytes(), 3, 32
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(body), b"WAVE", b"fmt ", 16, tag, ch, rate,
                      rate * ch * bits // 8, ch * bits // 8, bits, b"data", len(body))
    with open(path, "wb") as f:
        f.write(hdr + body)


def test_wav_file_source(tmp_path):
    """fileHulp semantics (filehulp.cpp:41-147): header rate/channels, PCM16 / 32768, mono -> Q = 0, loop at EOF without
    padding, attenuation; raw() hands out the file's own int16 pairs."""
    import importlib
    pkg = importlib.import_module("sdr-j-fm_amd")
    rng = np.random.default_rng(5)
    s = rng.integers(-20000, 20000, size=(1000, 2)).astype(np.int16)
    _write_wav(tmp_path / "a.wav", s, 2304000, "pcm16")
    src = pkg.WavFileSource(str(tmp_path / "a.wav"))
    assert src.getRate() == 2304000 and src.numofChannels == 2 and src.samplesinFile == 1000
    a = src.getSamples(600)
    assert np.array_equal(a, s[:600].astype(np.float32) / np.float32(32768.0))
    b = src.getSamples(600)                                   # wraps: 400 from the end, 200 from the start again
    assert np.array_equal(b, np.concatenate([s[600:], s[:200]]).astype(np.float32) / np.float32(32768.0))
    assert np.array_equal(src.raw(300), s[200:500])
    src2 = pkg.WavFileSource(str(tmp_path / "a.wav"), attenuation=0.5)
    assert np.array_equal(src2.getSamples(10), (s[:10].astype(np.float32) / np.float32(32768.0)) * np.float32(0.5))
    m = rng.standard_normal((50, 1)).astype(np.float32)
    _write_wav(tmp_path / "m.wav", m, 192000, "f32")
    mono = pkg.WavFileSource(str(tmp_path / "m.wav"))
    v = mono.getSamples(50)
    assert mono.getRate() == 192000 and np.array_equal(v[:, 0], m[:, 0]) and not v[:, 1].any()


def test_cpp_file_source_paced_reader(tmp_path):
    """The C++ file source (sdr-j-fm_amd/host/file_source.h) against fileHulp's behaviour (filehulp.cpp:41-202): starts paused,
    plays at the file's own sample rate (10 ms of samples per 10 ms period), wraps at end of file without padding, attenuation,
    mono -> Q = 0; the same samples as the Python WavFileSource."""
    import importlib, os, subprocess, time
    pkg = importlib.import_module("sdr-j-fm_amd")
    host = os.path.join(os.path.dirname(pkg.__file__), "host")
    exe = str(tmp_path / "file_source_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", os.path.join(host, "file_source_demo.cpp"), "-o", exe])
    rng = np.random.default_rng(9)
    rate = 192000
    s = rng.integers(-20000, 20000, size=(rate // 4 + 123, 2)).astype(np.int16)        # 0.25 s and a bit: not a multiple of the 10 ms period
    _write_wav(tmp_path / "a.wav", s, rate, "pcm16")
    n = int(1.6 * len(s))                                                              # wraps once
    out = subprocess.check_output([exe, str(tmp_path / "a.wav"), str(tmp_path / "o.f32"), str(n), "16384", "1", "0.5"]).decode()
    f = dict(zip(out.split()[0::2], out.split()[1::2]))
    got = np.fromfile(str(tmp_path / "o.f32"), np.float32).reshape(-1, 2)
    want = pkg.WavFileSource(str(tmp_path / "a.wav"), attenuation=0.5).getSamples(n)
    assert int(f["rate"]) == rate and int(f["frames"]) == len(s)
    assert int(f["paused_before"]) == 0 and int(f["paused_after"]) == 0                 # readerPausing = true until restartReader ()
    assert np.array_equal(got, want)
    sec = float(f["seconds"])
    assert 0.9 * n / rate - 0.02 <= sec <= 1.5 * n / rate + 0.2, (sec, n / rate)         # real time, not as fast as the disk
    t0 = time.time()
    subprocess.check_call([exe, str(tmp_path / "a.wav"), str(tmp_path / "o2.f32"), str(n), "16384", "0"], stdout=subprocess.DEVNULL)
    assert time.time() - t0 < 0.5 * n / rate + 0.3                                       # realtime = false: only the deadline sleep is gone
    got2 = np.fromfile(str(tmp_path / "o2.f32"), np.float32).reshape(-1, 2)
    assert np.array_equal(got2, pkg.WavFileSource(str(tmp_path / "a.wav")).getSamples(n))
    m = rng.integers(0, 255, size=(5000, 1)).astype(np.uint8)
    _write_wav(tmp_path / "m.wav", m, 48000, "pcm8")
    subprocess.check_call([exe, str(tmp_path / "m.wav"), str(tmp_path / "o3.f32"), "7000", "1000", "0"], stdout=subprocess.DEVNULL)
    got3 = np.fromfile(str(tmp_path / "o3.f32"), np.float32).reshape(-1, 2)
    assert np.array_equal(got3, pkg.WavFileSource(str(tmp_path / "m.wav")).getSamples(7000)) and not got3[:, 1].any()


# ------------------------------------------------------------------------------------------------
# RDS block synchroniser + group decoder on the host: fmx_rds_decode_bits needs no device
# ------------------------------------------------------------------------------------------------
def _rds(bits):
    import importlib
    pkg = importlib.import_module("sdr-j-fm_amd")
    return pkg.fmx.rds_decode_bits(bits)


def test_rds_checkword_matches_the_synchronisers_syndrome():
    """The test encoder and the decoder's syndrome register are two independent statements of IEC 62106 annex B:
    every encoded block must have syndrome 0 against its own offset word and not against the others."""
    import oracle_lib as ol

    def syndrome(block26, off):      # rds-blocksynchronizer.cpp:126-142 restated
        reg, blk = 0, block26 ^ off
        for k in range(25, -1, -1):
            msb = reg & 0x200
            reg = (reg << 1) & 0xFFFFFFFF
            if msb:
                reg ^= 0x5B9
            if (blk >> k) & 1:
                reg ^= 0x31B
        return reg & 0xFFFFFFFF
    rng = np.random.default_rng(3)
    for _ in range(200):
        w = int(rng.integers(0, 65536))
        for name, off in ol.RDS_OFFSETS.items():
            blk = (w << 10) | ol.rds_checkword(w, off)
            assert syndrome(blk, off) == 0
            assert all(syndrome(blk, o2) != 0 for n2, o2 in ol.RDS_OFFSETS.items() if n2 != name)


def test_rds_groups_clean_stream():
    import oracle_lib as ol
    prog = ol.rds_programme_bits(pi=0xD3A1, pty=10, ps="FMX-AMD ", text="HIP KERNELS ON MI355X - RDS OK")
    junk = np.random.default_rng(1).integers(0, 2, 77).astype(np.uint8)
    info = _rds(np.concatenate([junk, prog, prog]))
    assert info.synchronized == 1 and info.pi_code == 0xD3A1 and info.pty_code == 10
    assert info.station_label == b"FMX-AMD "
    # prepareText drops the last character of what it is given and trims (rds-groupdecoder.cpp:262-278): the CR ends the text
    assert info.radio_text == b"HIP KERNELS ON MI355X - RDS OK"
    assert info.groups_decoded >= 2 * 12 - 1 and info.crc_errors == 0
    assert info.music_speech == 1 and info.af1_khz == 0 and info.af2_khz == 87500 + 100 * 12      # block C = (0xE1 'one AF follows', 12)
    assert info.last_group_type == 2 and info.bit_error_rate == 0.0


def test_rds_groups_errors_resync_and_pi_change():
    import oracle_lib as ol
    a = ol.rds_programme_bits(pi=0x1234, ps="STATION1", text="FIRST")
    b = ol.rds_programme_bits(pi=0xBEEF, ps="STATION2", text="SECOND PROGRAMME")
    bad = a.copy()
    bad[4 * 104 + 30] ^= 1                      # one wrong bit in block B of the fifth group: that block fails its CRC
    info = _rds(np.concatenate([a, bad, a]))
    assert info.pi_code == 0x1234 and info.station_label == b"STATION1" and info.radio_text == b"FIRST"
    assert info.crc_errors == 1 and info.synchronized == 1       # dropped out once, found block A again
    info = _rds(np.concatenate([a, a, b, b]))
    assert info.pi_code == 0xBEEF and info.station_label == b"STATION2" and info.radio_text == b"SECOND PROGRAMME"
    # noise only: nothing decodes
    info = _rds(np.random.default_rng(9).integers(0, 2, 5000).astype(np.uint8))
    assert info.groups_decoded <= 1 and info.pi_code in (0, info.pi_code)
    # type B groups carry PI / PTY but are not decoded further (rds-groupdecoder.cpp:118-120)
    gb = ol.rds_group_bits(0x4242, (0 << 12) | (1 << 11) | (5 << 5) | 1, 0x4242, (ord("X") << 8) | ord("Y"), type_b=True)
    info = _rds(np.array(gb * 6, np.uint8))
    assert info.pi_code == 0x4242 and info.pty_code == 5 and info.station_label == b"        "


def test_block_synchroniser_two_implementations_agree():
    """The block synchroniser exists twice: in the library's host decoder (fmx_rds_decode_bits) and, reduced to what decides
    a resynchronisation, next to the RDS_3 slicer on the GPU / in the oracle (fmo_bsync_run).  Same sync-error count and lock
    state on clean groups, on groups with bit errors, on type-B groups and on noise."""
    import ctypes as C
    import oracle_lib as ol
    L = ol.oracle()
    rng = np.random.default_rng(31)
    a = ol.rds_programme_bits(pi=0x1234, ps="STATION1", text="FIRST")
    gb = np.array(ol.rds_group_bits(0x4242, (0 << 12) | (1 << 11) | (5 << 5) | 1, 0x4242, (ord("X") << 8) | ord("Y"), type_b=True) * 6, np.uint8)
    cases = [a, np.concatenate([rng.integers(0, 2, 777).astype(np.uint8), a, a]), gb, rng.integers(0, 2, 20000).astype(np.uint8)]
    for k in range(6):                                        # groups with scattered bit errors: drops out and re-locks
        x = np.concatenate([a, a, a]).copy()
        x[rng.integers(0, x.size, 3 + 5 * k)] ^= 1
        cases.append(np.concatenate([rng.integers(0, 2, 13 * k).astype(np.uint8), x]))
    for x in cases:
        x = np.ascontiguousarray(x, np.uint8)
        info = _rds(x)
        se, sy = C.c_int32(), C.c_int32()
        L.fmo_bsync_run(ol.u8ptr(x), x.size, C.byref(se), C.byref(sy))
        assert (se.value, sy.value) == (info.sync_errors, info.synchronized)
