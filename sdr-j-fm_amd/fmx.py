"""ctypes binding of libfmx (include/fmx.h) plus a host-side mirror of the reference's
``fmProcessor`` interface (includes/fm/fm-processor.h:104-156) for the FM hot path.

This module never computes DSP itself and has no CPU fallback: if libfmx.so or a HIP device
is missing every call raises ``FmxError``.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FMX_LIB") or os.path.join(HERE, "lib", "libfmx.so")      # FMX_LIB: A/B runs of two builds

# error codes (include/fmx.h)
FMX_OK, FMX_E_INVALID, FMX_E_UNSUPPORTED, FMX_E_NO_DEVICE, FMX_E_HIP, FMX_E_NOMEM, FMX_E_TOO_LARGE = 0, -1, -2, -3, -4, -5, -6

# parameter ids (include/fmx.h fmx_param_id)
ABI_VERSION = 3               # FMX_ABI_VERSION of include/fmx.h this mirror follows
P_FM_MODE, P_FM_DECODER, P_SOUND_MODE, P_STEREO_PANORAMA, P_SOUND_BALANCE, P_DEEMPHASIS = 1, 2, 3, 4, 5, 6
P_VOLUME_DB, P_LF_CUTOFF, P_BANDWIDTH, P_ATTENUATION_L, P_ATTENUATION_R, P_RDS_MODE = 7, 8, 9, 10, 11, 12
P_LOCAL_OSCILLATOR, P_AUTO_MONO, P_PSS, P_DC_REMOVE, P_SQUELCH_MODE, P_TEST_TONE, P_SQUELCH_VALUE = 13, 14, 15, 16, 17, 18, 19
P_DISP_DELAY = 20
P_STAGEB_FORM = 22            # handle-wide: 0 automatic, 1 stage B as one kernel per call, 2 as two (bit-identical results)
P_FRONT_KERNEL = 25           # handle-wide: 0 automatic, 1 fmx_front.hip (packed f32 FMAs), 3 fmx_front4.hip (f16-split matrix FIR; 2 was the six-wave kernel now in tools/experiments)
P_SCOPE_TAPS = 26             # handle-wide: -1 automatic (up to 64 channels), 0 the display feeds (demodulator / LR / pilot-phase scope taps, peak meter) are not produced, 1 produced
P_CALL_PIECES = 27            # handle-wide: -1 automatic, 0 never, n > 0 fm samples per piece of a call made in overlapping pieces (pre-pass batches)
P_FRONT_PARTS = 24            # handle-wide: 0 automatic, 1 one workgroup per channel, 2..32 parts in time per channel (bit-identical results)
P_FILTER_RESTARTS = 23        # handle-wide, before the first call: 0 automatic, 1 the reference's block filters (<= 64 channels), 2 folded FIRs
P_PLL_SOLVER = 21             # 0 automatic, 1 sequential (the reference's trajectory), 2 Newton while in lock + sequential around lock decisions, 3 Newton always
A_TRIGGER_FREQUENCY_CHANGE, A_RESTART_PSS, A_RESET_RDS = 100, 101, 102

TAP_FM_IQ, TAP_DEMOD, TAP_LR_RAW, TAP_PRE_RESAMPLER, TAP_RDS_IQ, TAP_PILOT_PHASE = 0, 1, 2, 3, 4, 5
IQ_F32, IQ_U8, IQ_S8, IQ_S16 = 0, 1, 2, 3

EXPORTS = [
    "fmx_abi_version", "fmx_last_error", "fmx_create", "fmx_destroy", "fmx_set_param", "fmx_frames_for", "fmx_filter_change_due",
    "fmx_process_host", "fmx_process_device", "fmx_process_host_raw", "fmx_process_device_raw", "fmx_synchronize",
    "fmx_get_meta", "fmx_get_tap", "fmx_get_peaks",
    "fmx_rds_bits", "fmx_rds_symbols", "fmx_last_fm_samples", "fmx_pll_replays", "fmx_pll_exact_segments", "fmx_last_front_kernel", "fmx_last_call_pieces", "fmx_last_second_group", "fmx_last_rds_samples", "fmx_last_rds_samples_of", "fmx_rds_decode", "fmx_rds_decode_bits", "fmx_rds_pty_name", "fmx_rds_map_char", "fmx_rds_prepare_text", "fmx_get_taps", "fmx_profile_enable", "fmx_profile_read",
]


class FmxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libfmx error %d: %s" % (code, msg))
        self.code = code


class FmxConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("device", C.c_int32), ("channels", C.c_int32), ("streams", C.c_int32),
        ("stream_of_channel", C.POINTER(C.c_int32)),
        ("inputRate", C.c_int32), ("fmRate", C.c_int32), ("workingRate", C.c_int32), ("audioRate", C.c_int32),
        ("max_block", C.c_int32),
    ]


class FmxMeta(C.Structure):
    _fields_ = [
        ("DcValRf", C.c_float), ("DcValIf", C.c_float), ("PssPhaseShiftDegree", C.c_float),
        ("PssPhaseChange", C.c_float), ("PssState", C.c_int32), ("PilotPllLockStrength", C.c_float),
        ("PilotPllLocked", C.c_int32), ("fm_samples", C.c_int64), ("pcm_frames", C.c_int64),
        ("live_pilot_locked", C.c_int32), ("live_lock_strength", C.c_float), ("live_dc_if", C.c_float),
        ("squelch_active", C.c_int32), ("live_rf_dc_re", C.c_float), ("live_rf_dc_im", C.c_float),
    ]


class FmxRdsInfo(C.Structure):
    _fields_ = [("synchronized", C.c_int32), ("pi_code", C.c_int32), ("pty_code", C.c_int32), ("last_group_type", C.c_int32),
                ("groups_decoded", C.c_int32), ("crc_errors", C.c_int32), ("sync_errors", C.c_int32), ("bit_error_rate", C.c_float),
                ("station_label", C.c_char * 9), ("radio_text", C.c_char * 65), ("af1_khz", C.c_int32), ("af2_khz", C.c_int32),
                ("music_speech", C.c_int32), ("di_code", C.c_int32),
                ("radio_text_ucs2", C.c_uint16 * 65), ("radio_text_ucs2_len", C.c_int16)]

    @property
    def radio_text_unicode(self):
        """The radio text as the reference's setRadioText receives it (prepareText + mapEBUtoUnicode, trimmed)."""
        return "".join(chr(v) for v in self.radio_text_ucs2[:self.radio_text_ucs2_len])


class FmxProfile(C.Structure):
    _fields_ = [("launches", C.c_int64 * 4), ("ms", C.c_double * 4), ("input_samples", C.c_int64),
                ("channel_samples", C.c_int64)]


_lib = None


def load_library(path=None):
    """Load libfmx.so and declare every symbol of include/fmx.h.  Raises if it is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise FmxError(FMX_E_NO_DEVICE, "%s not built (run `python sdr-j-fm_amd/build.py`); there is no CPU fallback" % p)
    L = C.CDLL(p)
    vp, i32, i64, f32p = C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_float)
    L.fmx_abi_version.restype = C.c_int
    L.fmx_abi_version.argtypes = []
    if L.fmx_abi_version() != ABI_VERSION:     # the output structs (FmxMeta, FmxRdsInfo) carry no size field: refuse a library of another layout
        raise FmxError(FMX_E_INVALID, "%s has ABI version %d, this binding was written for %d" % (p, L.fmx_abi_version(), ABI_VERSION))
    L.fmx_last_error.restype = C.c_char_p
    L.fmx_last_error.argtypes = []
    L.fmx_create.restype = C.c_int
    L.fmx_create.argtypes = [C.POINTER(FmxConfig), C.POINTER(vp)]
    L.fmx_destroy.restype = C.c_int
    L.fmx_destroy.argtypes = [vp]
    L.fmx_set_param.restype = C.c_int
    L.fmx_set_param.argtypes = [vp, i32, i32, C.c_double]
    L.fmx_frames_for.restype = i64
    L.fmx_frames_for.argtypes = [vp, i64]
    L.fmx_process_host.restype = C.c_int
    L.fmx_process_host.argtypes = [vp, f32p, i64, i64, f32p, i64, C.POINTER(i64)]
    L.fmx_process_device.restype = C.c_int
    L.fmx_process_device.argtypes = [vp, vp, i64, i64, vp, i64, C.POINTER(i64), vp]
    L.fmx_process_host_raw.restype = C.c_int
    L.fmx_process_host_raw.argtypes = [vp, vp, i32, C.c_float, i64, i64, f32p, i64, C.POINTER(i64)]
    L.fmx_process_device_raw.restype = C.c_int
    L.fmx_process_device_raw.argtypes = [vp, vp, i32, C.c_float, i64, i64, vp, i64, C.POINTER(i64), vp]
    L.fmx_synchronize.restype = C.c_int
    L.fmx_synchronize.argtypes = [vp]
    L.fmx_get_meta.restype = C.c_int
    L.fmx_get_meta.argtypes = [vp, i32, C.POINTER(FmxMeta)]
    L.fmx_get_tap.restype = C.c_int
    L.fmx_get_tap.argtypes = [vp, i32, i32, f32p, i64]
    L.fmx_get_peaks.restype = C.c_int
    L.fmx_get_peaks.argtypes = [vp, i32, f32p, i32, C.POINTER(i32)]
    L.fmx_rds_decode.restype = C.c_int
    L.fmx_rds_decode.argtypes = [vp, i32, C.POINTER(FmxRdsInfo)]
    L.fmx_rds_decode_bits.restype = C.c_int
    L.fmx_rds_decode_bits.argtypes = [C.POINTER(C.c_uint8), i32, C.POINTER(FmxRdsInfo)]
    L.fmx_rds_pty_name.restype = C.c_char_p
    L.fmx_rds_pty_name.argtypes = [i32, i32]
    L.fmx_rds_map_char.restype = C.c_uint16
    L.fmx_rds_map_char.argtypes = [C.c_uint8, C.c_uint8]
    L.fmx_rds_prepare_text.restype = i32
    L.fmx_rds_prepare_text.argtypes = [C.POINTER(C.c_uint8), i32, C.POINTER(C.c_uint8), C.POINTER(C.c_uint16), i32]
    L.fmx_rds_bits.restype = C.c_int
    L.fmx_rds_bits.argtypes = [vp, i32, C.POINTER(C.c_uint8), i32, C.POINTER(i32)]
    L.fmx_rds_symbols.restype = C.c_int
    L.fmx_rds_symbols.argtypes = [vp, i32, f32p, i32, C.POINTER(i32)]
    L.fmx_last_fm_samples.restype = C.c_int64
    L.fmx_last_fm_samples.argtypes = [vp]
    L.fmx_pll_replays.restype = C.c_int64
    L.fmx_pll_replays.argtypes = [vp, C.c_int32]
    L.fmx_pll_exact_segments.restype = C.c_int64
    L.fmx_filter_change_due.restype = C.c_int64
    L.fmx_filter_change_due.argtypes = [vp]
    L.fmx_last_front_kernel.restype = C.c_int32
    L.fmx_last_front_kernel.argtypes = [vp]
    L.fmx_last_call_pieces.restype = C.c_int32
    L.fmx_last_call_pieces.argtypes = [vp]
    L.fmx_last_second_group.restype = C.c_int32
    L.fmx_last_second_group.argtypes = [vp]
    L.fmx_pll_exact_segments.argtypes = [vp, C.c_int32]
    L.fmx_last_rds_samples.restype = C.c_int64
    L.fmx_last_rds_samples.argtypes = [vp]
    L.fmx_last_rds_samples_of.restype = C.c_int64
    L.fmx_last_rds_samples_of.argtypes = [vp, C.c_int32]
    L.fmx_get_taps.restype = C.c_int
    L.fmx_get_taps.argtypes = [vp, i32, i32, f32p, i32, C.POINTER(i32)]
    L.fmx_profile_enable.restype = C.c_int
    L.fmx_profile_enable.argtypes = [vp, i32]
    L.fmx_profile_read.restype = C.c_int
    L.fmx_profile_read.argtypes = [vp, C.POINTER(FmxProfile), i32]
    if path is None:
        _lib = L
    return L


def rds_decode_bits(bits):
    """Decode a recorded RDS bit stream from a fresh state (host only); returns FmxRdsInfo."""
    L = load_library()
    b = np.ascontiguousarray(bits, np.uint8)
    info = FmxRdsInfo()
    rc = L.fmx_rds_decode_bits(b.ctypes.data_as(C.POINTER(C.c_uint8)), b.size, C.byref(info))
    if rc != FMX_OK:
        raise FmxError(rc, L.fmx_last_error().decode())
    return info


class Fmx:
    """Thin object wrapper over an fmx_handle (a batch of FM channels on one GPU)."""

    def __init__(self, channels=1, streams=0, stream_of_channel=None, device=0, max_block=16384,
                 inputRate=2304000, fmRate=192000, workingRate=48000, audioRate=48000):
        self.L = load_library()
        cfg = FmxConfig()
        cfg.struct_size = C.sizeof(FmxConfig)
        cfg.device, cfg.channels, cfg.streams = device, channels, streams
        self._map = None
        if stream_of_channel is not None:
            self._map = (C.c_int32 * channels)(*[int(x) for x in stream_of_channel])
            cfg.stream_of_channel = C.cast(self._map, C.POINTER(C.c_int32))
        cfg.inputRate, cfg.fmRate, cfg.workingRate, cfg.audioRate = inputRate, fmRate, workingRate, audioRate
        cfg.max_block = max_block
        self.channels = channels
        self.streams = streams if streams > 0 else channels
        self.max_block = max_block
        self.h = C.c_void_p()
        self._check(self.L.fmx_create(C.byref(cfg), C.byref(self.h)))

    def _check(self, rc):
        if rc != FMX_OK:
            raise FmxError(rc, self.L.fmx_last_error().decode("utf-8", "replace"))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.fmx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_param(self, pid, value, channel=-1):
        self._check(self.L.fmx_set_param(self.h, channel, pid, float(value)))

    def frames_for(self, n):
        return int(self.L.fmx_frames_for(self.h, n))

    def process_host(self, iq):
        """iq: float32 [streams, n, 2] (or [n, 2] for one stream) -> pcm float32 [channels, frames, 2]."""
        iq = np.ascontiguousarray(iq, np.float32)
        if iq.ndim == 2:
            iq = iq[None]
        assert iq.shape[0] == self.streams and iq.shape[2] == 2
        n = iq.shape[1]
        frames = self.frames_for(n)
        cap = max(frames, 1)
        pcm = np.zeros((self.channels, cap, 2), np.float32)
        got = C.c_int64()
        self._check(self.L.fmx_process_host(self.h, iq.ctypes.data_as(C.POINTER(C.c_float)), n, n,
                                            pcm.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(got)))
        return pcm[:, :got.value]

    def process_host_raw(self, iq, fmt, s16_denominator=2048.0):
        """Raw device samples: iq uint8 / int8 / int16 [streams, n, 2] (or [n, 2]), fmt = IQ_U8 / IQ_S8 /
        IQ_S16; converted on the GPU exactly as the reference's device handlers do on the host."""
        dt = {IQ_U8: np.uint8, IQ_S8: np.int8, IQ_S16: np.int16, IQ_F32: np.float32}[fmt]
        iq = np.ascontiguousarray(iq, dt)
        if iq.ndim == 2:
            iq = iq[None]
        assert iq.shape[0] == self.streams and iq.shape[2] == 2
        n = iq.shape[1]
        cap = max(self.frames_for(n), 1)
        pcm = np.zeros((self.channels, cap, 2), np.float32)
        got = C.c_int64()
        self._check(self.L.fmx_process_host_raw(self.h, iq.ctypes.data_as(C.c_void_p), fmt, float(s16_denominator), n, n,
                                                pcm.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(got)))
        return pcm[:, :got.value]

    def process_device(self, d_iq_ptr, stream_stride, n, d_pcm_ptr, pcm_stride, hip_stream=None):
        got = C.c_int64()
        self._check(self.L.fmx_process_device(self.h, C.c_void_p(d_iq_ptr), stream_stride, n, C.c_void_p(d_pcm_ptr),
                                              pcm_stride, C.byref(got), C.c_void_p(hip_stream or 0)))
        return got.value

    def synchronize(self):
        self._check(self.L.fmx_synchronize(self.h))

    def meta(self, channel=0):
        m = FmxMeta()
        self._check(self.L.fmx_get_meta(self.h, channel, C.byref(m)))
        return m

    def tap(self, tap_id, n, channel=0):
        width = 1 if tap_id in (TAP_DEMOD, TAP_PILOT_PHASE) else 2
        out = np.zeros((n, width), np.float32)
        self._check(self.L.fmx_get_tap(self.h, channel, tap_id, out.ctypes.data_as(C.POINTER(C.c_float)), n))
        return out[:, 0] if width == 1 else out

    def peaks(self, channel=0, capacity=256):
        """showPeakLevel events since the last fetch as [events, 2] (leftDb, rightDb)."""
        out = np.zeros((max(capacity, 1), 2), np.float32)
        n = C.c_int32()
        self._check(self.L.fmx_get_peaks(self.h, channel, out.ctypes.data_as(C.POINTER(C.c_float)), capacity, C.byref(n)))
        return out[:n.value].copy()

    def rds_bits(self, channel=0, capacity=8192):
        buf = (C.c_uint8 * capacity)()
        n = C.c_int32()
        self._check(self.L.fmx_rds_bits(self.h, channel, buf, capacity, C.byref(n)))
        return np.frombuffer(buf, np.uint8, n.value).copy()

    def rds_symbols(self, channel=0, capacity=1024):
        """The constellation points the pending bits were decided on, [n, 2]."""
        out = np.zeros((max(capacity, 1), 2), np.float32)
        n = C.c_int32()
        self._check(self.L.fmx_rds_symbols(self.h, channel, out.ctypes.data_as(C.POINTER(C.c_float)), capacity, C.byref(n)))
        return out[:n.value].copy()

    def last_fm_samples(self):
        return int(self.L.fmx_last_fm_samples(self.h))

    def filter_change_due(self):
        """Input samples per stream still to come before a pending mid-stream setBandwidth / setlfcutoff of a batch takes effect (fmx_filter_change_due):
        0 = the next call applies it at its first sample, -1 = nothing pending (setters apply with the next call)."""
        return int(self.L.fmx_filter_change_due(self.h))

    def last_front_kernel(self):
        """Which kernel ran the input-filter stage of the last call (fmx_last_front_kernel: FMX_P_FRONT_KERNEL's numbering)."""
        return int(self.L.fmx_last_front_kernel(self.h))

    def last_call_pieces(self):
        """Overlapping pieces the last call was made in (fmx_last_call_pieces, P_CALL_PIECES; 1: whole)."""
        return int(self.L.fmx_last_call_pieces(self.h))

    def last_second_group(self):
        """Channels of the second of the two groups the last call's stereo / audio stages ran as (fmx_last_second_group; 0: one group)."""
        return int(self.L.fmx_last_second_group(self.h))

    def last_rds_samples(self, channel=0):
        """24 kS/s RDS samples the last call produced on `channel` (fmx_last_rds_samples_of): the n that tap(TAP_RDS_IQ, n, channel) accepts."""
        return int(self.L.fmx_last_rds_samples_of(self.h, channel))

    def pll_replays(self, channel=-1):
        """Segments of the pilot PLL that were replayed sequentially (fmx_pll_replays)."""
        n = int(self.L.fmx_pll_replays(self.h, channel))
        if n < 0:
            self._check(n)
        return n

    def pll_exact_segments(self, channel=-1):
        """Segments FMX_P_PLL_SOLVER = 2 evaluated sequentially because the pilot was not comfortably in lock (fmx_pll_exact_segments)."""
        n = int(self.L.fmx_pll_exact_segments(self.h, channel))
        if n < 0:
            self._check(n)
        return n

    def rds_decode(self, channel=0):
        """Feed the bits sliced so far into the channel's block synchroniser / group decoder; returns FmxRdsInfo."""
        info = FmxRdsInfo()
        self._check(self.L.fmx_rds_decode(self.h, channel, C.byref(info)))
        return info

    def taps(self, which, channel=0):
        buf = np.zeros(1024, np.float32)
        n = C.c_int32()
        self._check(self.L.fmx_get_taps(self.h, channel, which, buf.ctypes.data_as(C.POINTER(C.c_float)), buf.size, C.byref(n)))
        return buf[:n.value].copy()

    def profile_enable(self, on=True):
        self._check(self.L.fmx_profile_enable(self.h, 1 if on else 0))

    def profile_read(self, reset=True):
        p = FmxProfile()
        self._check(self.L.fmx_profile_read(self.h, C.byref(p), 1 if reset else 0))
        return {"launches": list(p.launches), "ms": list(p.ms), "input_samples": p.input_samples,
                "channel_samples": p.channel_samples}


class FmProcessor:
    """Host-side mirror of the reference's ``fmProcessor`` (fm-processor.h:79-156) for ONE channel
    of an :class:`Fmx` batch: same method names and argument meaning as the reference setters, so a
    test written against the reference class reads the same here.  ``run_block`` is one iteration
    of ``fmProcessor::run()`` (fm-processor.cpp:387-686): pull a block from a deviceHandler-shaped
    source (``Samples()``/``getSamples(n)``) and push PCM into an audioSink-shaped sink
    (``putSamples(array)``)."""

    FM_Mode = {"Stereo": 0, "StereoPano": 1, "Mono": 2}
    DECODERS = {"AM": 1, "FM PLL Decoder": 2, "FM Mixed Demod": 3, "FM Complex Baseband Delay": 4,
                "FM Real Baseband Delay": 5, "FM Difference Based": 6}      # fm-demodulator.cpp:36-43

    def __init__(self, theDevice=None, mySink=None, inputRate=None, fmRate=192000, workingRate=48000,
                 audioRate=48000, fmx=None, channel=0, blockSize=16384):
        self.myRig, self.theSink = theDevice, mySink
        if inputRate is None:                             # radio.cpp:836: inputRate = theDevice -> getRate ()
            inputRate = theDevice.getRate() if hasattr(theDevice, "getRate") else 2304000
        self.fmx = fmx if fmx is not None else Fmx(1, max_block=blockSize, inputRate=inputRate, fmRate=fmRate,
                                                   workingRate=workingRate, audioRate=audioRate)
        self.channel = channel
        self.bufferSize = blockSize                       # fm-processor.cpp:374

    def _set(self, pid, v):
        self.fmx.set_param(pid, v, self.channel)

    # --- the reference's setters (same names) ---
    def setfmMode(self, m): self._set(P_FM_MODE, self.FM_Mode[m] if isinstance(m, str) else m)

    def setFMdecoder(self, name):
        # fm_Demodulator::setDecoder: an unknown name selects -1, which falls into `default:` = PLL
        self._set(P_FM_DECODER, self.DECODERS.get(name, 2) if isinstance(name, str) else name)

    def setSoundMode(self, selector): self._set(P_SOUND_MODE, selector)
    def setStereoPanorama(self, pan): self._set(P_STEREO_PANORAMA, pan)
    def setSoundBalance(self, balance): self._set(P_SOUND_BALANCE, balance)
    def setDeemphasis(self, us): self._set(P_DEEMPHASIS, us)
    def setVolume(self, gain_db): self._set(P_VOLUME_DB, gain_db)
    def setlfcutoff(self, hz): self._set(P_LF_CUTOFF, hz)

    def setBandwidth(self, s):
        # the reference takes the GUI string "165kHz" or "Off" (fm-processor.cpp:232-239)
        if isinstance(s, str):
            v = 0 if s == "Off" else int("".join(ch for ch in s if ch.isdigit() or ch == "-") or "0") * 1000
        else:
            v = int(s)
        self._set(P_BANDWIDTH, v)

    def setAttenuation(self, l, r):
        self._set(P_ATTENUATION_L, l)
        self._set(P_ATTENUATION_R, r)

    def setfmRdsSelector(self, m): self._set(P_RDS_MODE, m)
    def triggerFrequencyChange(self): self._set(A_TRIGGER_FREQUENCY_CHANGE, 0)
    def restartPssAnalyzer(self): self._set(A_RESTART_PSS, 0)
    def resetRds(self): self._set(A_RESET_RDS, 0)
    def set_localOscillator(self, lo): self._set(P_LOCAL_OSCILLATOR, lo)
    def set_squelchMode(self, m): self._set(P_SQUELCH_MODE, m)          # ESqMode: 0 OFF, 1 NSQ, 2 LSQ
    def set_squelchValue(self, v): self._set(P_SQUELCH_VALUE, v)         # fm-processor.cpp:213-215
    def getSquelchState(self): return bool(self.fmx.meta(self.channel).squelch_active)      # :217-219
    def pollPeakLevels(self): return self.fmx.peaks(self.channel)       # what showPeakLevel was emitted with since the last poll
    def setAutoMonoMode(self, b): self._set(P_AUTO_MONO, 1 if b else 0)
    def setPSSMode(self, b): self._set(P_PSS, 1 if b else 0)
    def setDCRemove(self, b): self._set(P_DC_REMOVE, 1 if b else 0)
    def setTestTone(self, b): self._set(P_TEST_TONE, 1 if b else 0)
    def setDispDelay(self, steps): self._set(P_DISP_DELAY, steps)

    def isPilotLocked(self):
        m = self.fmx.meta(self.channel)
        return bool(m.live_pilot_locked), m.live_lock_strength

    def get_demodDcComponent(self): return self.fmx.meta(self.channel).live_dc_if

    def run_block(self):
        """One loop iteration of fmProcessor::run(): returns False when the device has < bufferSize samples."""
        if self.myRig.Samples() < self.bufferSize:
            return False
        iq = self.myRig.getSamples(self.bufferSize)
        pcm = self.fmx.process_host(iq)
        if self.theSink is not None and pcm.shape[1] > 0:
            self.theSink.putSamples(pcm[self.channel])
        return True
