"""ctypes bindings for the TEST-ONLY checker libraries under oracle/.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
  * ``oracle()``  -> oracle/libfmoracle.so (this repo's C restatement, fm_oracle.c)
  * ``ref()``     -> oracle/_ref/libfmref.so (the reference's own leaf classes) or None
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

c_float_p = C.POINTER(C.c_float)
c_u8_p = C.POINTER(C.c_uint8)
c_i32_p = C.POINTER(C.c_int32)


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_float_p)


def u8ptr(a):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_u8_p)


class FmoConfig(C.Structure):
    _fields_ = [
        ("inputRate", C.c_int32), ("fmRate", C.c_int32), ("workingRate", C.c_int32), ("audioRate", C.c_int32),
        ("fmMode", C.c_int32), ("soundSelector", C.c_int32), ("decoder", C.c_int32),
        ("inputFilterBw", C.c_int32), ("lfCutoff", C.c_int32), ("deemphasis", C.c_int32),
        ("volumeDb", C.c_float), ("useCtorVolume", C.c_int32), ("balance", C.c_int32), ("panorama", C.c_int32),
        ("attL", C.c_float), ("attR", C.c_float), ("loFrequency", C.c_int32),
        ("dcRemove", C.c_int32), ("autoMono", C.c_int32), ("pssActive", C.c_int32), ("rdsMode", C.c_int32),
        ("squelchMode", C.c_int32), ("squelchValue", C.c_int32), ("testTone", C.c_int32), ("dispDelay", C.c_int32),
        ("touchInputFilter", C.c_int32), ("touchLfCutoff", C.c_int32),
        ("testFilterNoise", C.c_float), ("testNoiseSeed", C.c_int32),
    ]


class FmoMeta(C.Structure):
    _fields_ = [
        ("dcValRf", C.c_float), ("dcValIf", C.c_float), ("pssPhaseShiftDegree", C.c_float),
        ("pssPhaseChange", C.c_float), ("pssState", C.c_int32), ("pilotLockStrength", C.c_float),
        ("pilotLocked", C.c_int32), ("peakLeftDb", C.c_float), ("peakRightDb", C.c_float),
        ("fmSamples", C.c_int64), ("pcmFrames", C.c_int64), ("squelchActive", C.c_int32), ("pad_", C.c_int32),
    ]


class FmoSiggenConfig(C.Structure):
    _fields_ = [
        ("inputRate", C.c_int32), ("carrierAmp", C.c_double), ("deviationHz", C.c_double), ("offsetHz", C.c_double),
        ("leftHz", C.c_double), ("rightHz", C.c_double), ("leftAmp", C.c_double), ("rightAmp", C.c_double),
        ("stereo", C.c_int32), ("pilotLevel", C.c_double), ("rds", C.c_int32), ("rdsLevel", C.c_double),
        ("noiseSeed", C.c_uint64), ("noiseSigma", C.c_double), ("dcI", C.c_double), ("dcQ", C.c_double),
        ("rdsBitsSeed", C.c_uint64),
    ]


TAP_FM_IQ, TAP_DEMOD, TAP_LRRAW, TAP_PRE_RS, TAP_PILOT, TAP_PSS, TAP_RDS_IQ = range(7)

_oracle = None
_ref = None
_ref_tried = False


def build(force=False):
    """(Re)build the checker libraries with oracle/Makefile (gcc only; seconds)."""
    so = os.path.join(ORACLE_DIR, "libfmoracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(ORACLE_DIR, "fm_oracle.c")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libfmoracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)


def oracle():
    global _oracle
    if _oracle is not None:
        return _oracle
    build()
    L = C.CDLL(os.path.join(ORACLE_DIR, "libfmoracle.so"))
    vp, i32, f32, lng, dbl = C.c_void_p, C.c_int32, C.c_float, C.c_long, C.c_double
    sig = {
        "fmo_lowpass_kernel": (None, [C.c_int, i32, i32, c_float_p]),
        "fmo_decim_kernel": (None, [C.c_int, i32, i32, c_float_p]),
        "fmo_bandpass_kernel": (None, [C.c_int, i32, i32, i32, c_float_p]),
        "fmo_rrc_kernel": (C.c_int, [dbl, dbl, dbl, dbl, C.c_int, c_float_p]),
        "fmo_fft_radix2": (C.c_int, [c_float_p, lng, C.c_int]),
        "fmo_fftfilter_new": (vp, [C.c_int, C.c_int]),
        "fmo_fftfilter_free": (None, [vp]),
        "fmo_fftfilter_set_lowpass": (None, [vp, i32, i32]),
        "fmo_fftfilter_set_band": (None, [vp, i32, i32, i32]),
        "fmo_fftfilter_set_hilbert": (None, [vp]),
        "fmo_fftfilter_run_c": (None, [vp, c_float_p, c_float_p, lng]),
        "fmo_fftfilter_run_r": (None, [vp, c_float_p, c_float_p, lng]),
        "fmo_decim_new": (vp, [C.c_int, i32, i32, C.c_int]),
        "fmo_decim_free": (None, [vp]),
        "fmo_decim_run": (lng, [vp, c_float_p, lng, c_float_p]),
        "fmo_squelch_new": (vp, [i32, i32, i32, i32]),
        "fmo_squelch_free": (None, [vp]),
        "fmo_squelch_set_level": (None, [vp, C.c_int]),
        "fmo_squelch_run": (None, [vp, c_float_p, c_float_p, c_float_p, C.POINTER(C.c_uint8), lng]),
        "fmo_sincos_new": (vp, [i32]),
        "fmo_sincos_free": (None, [vp]),
        "fmo_sincos_sin": (f32, [vp, f32]),
        "fmo_sincos_cos": (f32, [vp, f32]),
        "fmo_sincos_table": (c_float_p, [vp]),
        "fmo_atan_new": (vp, []),
        "fmo_atan_free": (None, [vp]),
        "fmo_atan2": (f32, [vp, f32, f32]),
        "fmo_atan_table": (c_float_p, [vp, C.c_int]),
        "fmo_pi_constrain": (f32, [f32]),
        "fmo_pll_new": (vp, [i32, f32, f32, f32, f32, vp, vp]),
        "fmo_pll_free": (None, [vp]),
        "fmo_pll_phase_incr": (f32, [vp]),
        "fmo_demod_new": (vp, [i32]),
        "fmo_demod_free": (None, [vp]),
        "fmo_demod_set_decoder": (None, [vp, C.c_int]),
        "fmo_demod_dc": (f32, [vp]),
        "fmo_demod_carrier": (f32, [vp]),
        "fmo_demod_kfm": (f32, [vp]),
        "fmo_pilot_new": (vp, [i32, f32, f32, vp]),
        "fmo_pilot_free": (None, [vp]),
        "fmo_pilot_phase": (f32, [vp, f32]),
        "fmo_pilot_locked": (C.c_int, [vp]),
        "fmo_pilot_strength": (f32, [vp]),
        "fmo_pss_new": (vp, [i32, f32, vp]),
        "fmo_pss_free": (None, [vp]),
        "fmo_pss_reset": (None, [vp]),
        "fmo_pss_process": (f32, [vp, f32, f32]),
        "fmo_resampler_taps": (None, [c_float_p]),
        "fmo_sincos_eval": (None, [vp, c_float_p, lng, c_float_p, c_float_p, c_float_p]),
        "fmo_atan2_eval": (None, [c_float_p, c_float_p, lng, c_float_p]),
        "fmo_pi_constrain_eval": (None, [c_float_p, lng, c_float_p]),
        "fmo_pll_run": (None, [i32, f32, f32, f32, f32, c_float_p, lng, c_float_p]),
        "fmo_demod_run": (None, [i32, C.c_int, c_float_p, lng, c_float_p, c_float_p, c_float_p]),
        "fmo_pilot_run": (None, [i32, f32, f32, c_float_p, lng, c_float_p, c_u8_p, c_float_p]),
        "fmo_pss_run": (None, [i32, f32, c_float_p, c_float_p, lng, c_float_p, c_u8_p]),
        "fmo_agc_run": (None, [f32, f32, f32, c_float_p, lng, c_float_p]),
        "fmo_costas_run": (None, [f32, f32, f32, f32, c_float_p, lng, c_float_p]),
        "fmo_pilot_constants": (None, [i32, c_float_p, c_float_p, c_float_p]),
        "fmo_config_defaults": (None, [C.POINTER(FmoConfig)]),
        "fmo_chain_new": (vp, [C.POINTER(FmoConfig)]),
        "fmo_chain_free": (None, [vp]),
        "fmo_chain_configure": (None, [vp, C.POINTER(FmoConfig)]),
        "fmo_chain_trigger_frequency_change": (None, [vp]),
        "fmo_chain_set_tap": (None, [vp, C.c_int, c_float_p, lng]),
        "fmo_chain_tap_count": (lng, [vp, C.c_int]),
        "fmo_chain_process": (lng, [vp, c_float_p, lng, c_float_p, lng]),
        "fmo_chain_meta": (None, [vp, C.POINTER(FmoMeta)]),
        "fmo_chain_rds_bits": (lng, [vp, c_u8_p, lng]),
        "fmo_chain_peaks": (lng, [vp, c_float_p, lng]),
        "fmo_rds1_coeffs": (None, [c_float_p]),
        "fmo_bsync_run": (None, [c_u8_p, lng, c_i32_p, c_i32_p]),
        "fmo_iir_new": (vp, [C.c_int, C.c_int, i32, i32, i32, C.c_int]),
        "fmo_iir_free": (None, [vp]),
        "fmo_iir_coeffs": (C.c_int, [vp, c_float_p]),
        "fmo_iir_run": (None, [vp, c_float_p, lng, c_float_p]),
        "fmo_test_tone_burst": (None, [i32, c_float_p, lng]),
        "fmo_siggen_new": (vp, [C.POINTER(FmoSiggenConfig)]),
        "fmo_siggen_free": (None, [vp]),
        "fmo_siggen_run": (None, [vp, c_float_p, lng]),
        "fmo_siggen_rds_bits": (lng, [vp, c_u8_p, lng]),
        "fmo_siggen_set_rds_bits": (None, [vp, c_u8_p, lng]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _oracle = L
    return L


def ref():
    """The reference's own leaf classes (oracle/_ref/libfmref.so) or None when not built."""
    global _ref, _ref_tried
    if _ref_tried:
        return _ref
    _ref_tried = True
    try:
        build()
    except Exception:
        pass
    path = os.path.join(ORACLE_DIR, "_ref", "libfmref.so")
    if not os.path.exists(path):
        return None
    try:
        L = C.CDLL(path)
    except OSError:
        return None
    vp, i32, f32, lng, dbl = C.c_void_p, C.c_int32, C.c_float, C.c_long, C.c_double
    sig = {
        "ref_has_qt": (C.c_int, []),
        "ref_lowpass_kernel": (None, [C.c_int, i32, i32, c_float_p]),
        "ref_decim_kernel": (None, [C.c_int, i32, i32, c_float_p]),
        "ref_bandpass_kernel": (None, [C.c_int, i32, i32, i32, c_float_p]),
        "ref_rrc_kernel": (C.c_int, [dbl, dbl, dbl, dbl, C.c_int, c_float_p]),
        "ref_fft": (C.c_int, [c_float_p, lng, C.c_int]),
        "ref_fftfilter_new": (vp, [C.c_int, C.c_int]),
        "ref_fftfilter_hilbert_new": (vp, [C.c_int, C.c_int]),
        "ref_fftfilter_free": (None, [vp]),
        "ref_fftfilter_hilbert_free": (None, [vp]),
        "ref_fftfilter_set_lowpass": (None, [vp, i32, i32]),
        "ref_fftfilter_set_band": (None, [vp, i32, i32, i32]),
        "ref_fftfilter_run_c": (None, [vp, c_float_p, c_float_p, lng]),
        "ref_fftfilter_run_r": (None, [vp, c_float_p, c_float_p, lng]),
        "ref_fftfilter_hilbert_run": (None, [vp, c_float_p, c_float_p, lng]),
        "ref_decim_new": (vp, [C.c_int, i32, i32, C.c_int]),
        "ref_decim_free": (None, [vp]),
        "ref_decim_run": (lng, [vp, c_float_p, lng, c_float_p]),
        "ref_sincos_new": (vp, [i32]),
        "ref_sincos_free": (None, [vp]),
        "ref_sincos_eval": (None, [vp, c_float_p, lng, c_float_p, c_float_p, c_float_p]),
        "ref_atan2_eval": (None, [c_float_p, c_float_p, lng, c_float_p]),
        "ref_lo_run": (None, [i32, i32, lng, c_float_p]),
        "ref_lo_table": (None, [i32, c_i32_p, lng, c_float_p]),
        "ref_pi_constrain": (None, [c_float_p, lng, c_float_p]),
        "ref_pll_run": (None, [i32, f32, f32, f32, f32, c_float_p, lng, c_float_p]),
        "ref_pilot_run": (None, [i32, f32, f32, c_float_p, lng, c_float_p, c_u8_p, c_float_p]),
        "ref_pss_run": (None, [i32, f32, c_float_p, c_float_p, lng, c_float_p, c_u8_p]),
        "ref_agc_run": (None, [f32, f32, f32, c_float_p, lng, c_float_p]),
        "ref_costas_run": (None, [f32, f32, f32, f32, c_float_p, lng, c_float_p]),
        "ref_iir_new": (vp, [C.c_int, C.c_int, i32, i32, i32, C.c_int]),
        "ref_iir_free": (None, [vp]),
        "ref_iir_coeffs": (C.c_int, [vp, c_float_p]),
        "ref_iir_run": (None, [vp, c_float_p, lng, c_float_p]),
        "ref_chain_new": (vp, [i32, i32, C.c_int, C.c_int, C.c_int, C.c_int, f32, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_int, C.c_int]),
        "ref_chain_free": (None, [vp]),
        "ref_chain_run": (lng, [vp, c_float_p, lng, c_float_p, c_float_p, c_float_p, c_float_p]),
        "ref_rdsgroup_fields": (None, [C.POINTER(C.c_uint16), C.POINTER(C.c_int32)]),
        "ref_map_ebu": (C.c_uint16, [C.c_uint8, C.c_uint8]),
        "ref_pty_name": (C.c_char_p, [i32, i32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    if L.ref_has_qt():
        for name, (res, args) in {"ref_squelch_new": (vp, [i32, i32, i32, i32]), "ref_squelch_free": (None, [vp]),
                                  "ref_squelch_set_level": (None, [vp, C.c_int]),
                                  "ref_squelch_run": (None, [vp, c_float_p, c_float_p, c_float_p, C.POINTER(C.c_uint8), lng])}.items():
            fn = getattr(L, name); fn.restype = res; fn.argtypes = args
        L.ref_demod_run.restype = None
        L.ref_demod_run.argtypes = [i32, C.c_int, c_float_p, lng, c_float_p, c_float_p, c_float_p, c_float_p]
    _ref = L
    return L


# ---------------------------------------------------------------- convenience wrappers
def default_config(**kw):
    cfg = FmoConfig()
    oracle().fmo_config_defaults(C.byref(cfg))
    for k, v in kw.items():
        assert hasattr(cfg, k), k
        setattr(cfg, k, v)
    return cfg


def siggen_config(**kw):
    c = FmoSiggenConfig()
    c.inputRate = 2304000
    c.carrierAmp = 0.5
    c.deviationHz = 75000.0
    c.offsetHz = 0.0
    c.leftHz, c.rightHz = 1000.0, 400.0
    c.leftAmp, c.rightAmp = 0.5, 0.5
    c.stereo = 1
    c.pilotLevel = 0.10
    c.rds = 0
    c.rdsLevel = 0.03
    c.noiseSeed = 0
    c.noiseSigma = 0.0
    c.dcI = c.dcQ = 0.0
    c.rdsBitsSeed = 0
    for k, v in kw.items():
        assert hasattr(c, k), k
        setattr(c, k, v)
    return c


# ---- RDS group encoder for the tests (IEC 62106: 16 data bits + 10-bit checkword = CRC (x^10+x^8+x^7+x^5+x^4+x^3+1) ^ offset word)
RDS_OFFSETS = {"A": 0x0FC, "B": 0x198, "C": 0x168, "C'": 0x350, "D": 0x1B4}


def rds_checkword(data16, offset):
    reg = 0
    for k in range(15, -1, -1):
        fb = ((reg >> 9) & 1) ^ ((data16 >> k) & 1)
        reg = (reg << 1) & 0x3FF
        if fb:
            reg ^= 0x1B9
    return reg ^ offset


def rds_group_bits(a, b, c, d, type_b=False):
    out = []
    for word, off in ((a, "A"), (b, "B"), (c, "C'" if type_b else "C"), (d, "D")):
        blk = ((word & 0xFFFF) << 10) | rds_checkword(word & 0xFFFF, RDS_OFFSETS[off])
        out += [(blk >> k) & 1 for k in range(25, -1, -1)]
    return out


def rds_programme_bits(pi=0xD3A1, pty=10, ps="FMX-AMD ", text="HIP KERNELS ON MI355X - RDS OK", tp=0):
    """One cycle of groups: four 0A groups (PS name) followed by the 2A groups of the radio text (ended by CR)."""
    bits = []
    ps = (ps + " " * 8)[:8]
    for seg in range(4):
        b = (0 << 12) | (0 << 11) | (tp << 10) | (pty << 5) | (1 << 3) | seg         # group 0A, M/S = music
        bits += rds_group_bits(pi, b, (0xE0 + 1) << 8 | 12, (ord(ps[2 * seg]) << 8) | ord(ps[2 * seg + 1]))
    txt = text + "\r"
    txt += " " * (-len(txt) % 4)
    for seg in range(len(txt) // 4):
        b = (2 << 12) | (0 << 11) | (tp << 10) | (pty << 5) | (0 << 4) | seg           # group 2A, text A
        q = txt[4 * seg:4 * seg + 4]
        bits += rds_group_bits(pi, b, (ord(q[0]) << 8) | ord(q[1]), (ord(q[2]) << 8) | ord(q[3]))
    return np.array(bits, np.uint8)


def synth_iq(n, return_rds_bits=False, rds_payload=None, **kw):
    """n complex samples of synthetic FM IQ as float32 [n,2] (oracle's deterministic generator); with
    return_rds_bits also the (pre differential-encoding) RDS data bits the generator sent."""
    L = oracle()
    cfg = siggen_config(**kw)
    g = L.fmo_siggen_new(C.byref(cfg))
    if rds_payload is not None:
        pb = np.ascontiguousarray(rds_payload, np.uint8)
        L.fmo_siggen_set_rds_bits(g, u8ptr(pb), pb.size)
    out = np.empty((n, 2), np.float32)
    L.fmo_siggen_run(g, fptr(out), n)
    bits = None
    if return_rds_bits:
        nb = L.fmo_siggen_rds_bits(g, None, 0)
        bits = np.zeros(max(nb, 1), np.uint8)
        L.fmo_siggen_rds_bits(g, u8ptr(bits), nb)
        bits = bits[:nb]
    L.fmo_siggen_free(g)
    return (out, bits) if return_rds_bits else out


class OracleChain:
    """fmo_chain with numpy in/out and optional tap capture."""

    def __init__(self, cfg=None, taps=(), tap_seconds=6.0, **kw):
        self.L = oracle()
        self.cfg = cfg if cfg is not None else default_config(**kw)
        self.h = self.L.fmo_chain_new(C.byref(self.cfg))
        self.tapbufs = {}
        for t in taps:
            per = 2 if t in (TAP_FM_IQ, TAP_LRRAW, TAP_PRE_RS, TAP_RDS_IQ) else 1
            rate = 24000 if t == TAP_RDS_IQ else self.cfg.fmRate
            buf = np.zeros(int(tap_seconds * rate) * per, np.float32)
            self.tapbufs[t] = buf
            self.L.fmo_chain_set_tap(self.h, t, fptr(buf), buf.size)

    def configure(self, **kw):
        for k, v in kw.items():
            setattr(self.cfg, k, v)
        # a keyword given = the reference's setter called: setBandwidth / setlfcutoff restart their filter even when the value is the current one
        self.cfg.touchInputFilter = 1 if "inputFilterBw" in kw else 0
        self.cfg.touchLfCutoff = 1 if "lfCutoff" in kw else 0
        self.L.fmo_chain_configure(self.h, C.byref(self.cfg))
        self.cfg.touchInputFilter = 0; self.cfg.touchLfCutoff = 0

    def process(self, iq):
        iq = np.ascontiguousarray(iq, np.float32).reshape(-1, 2)
        n = iq.shape[0]
        dec = 12 if self.cfg.inputRate >= 2304000 else (6 if self.cfg.inputRate // self.cfg.fmRate > 1 else 1)      # fm-processor.cpp:68-75, 471
        cap = n // (4 * dec) + 64 + 16384 // (4 * dec) + 2  # (the chain works in the reference's 16384-sample blocks: up to one block may be pending)
        cap = cap * max(self.cfg.audioRate, self.cfg.workingRate) // self.cfg.workingRate + 8      # second converter (audioRate != workingRate)
        pcm = np.zeros((cap, 2), np.float32)
        got = self.L.fmo_chain_process(self.h, fptr(iq), n, fptr(pcm), cap)
        assert got <= cap
        return pcm[:got]

    def tap(self, t):
        n = self.L.fmo_chain_tap_count(self.h, t)
        a = self.tapbufs[t][:n]
        return a.reshape(-1, 2) if t in (TAP_FM_IQ, TAP_LRRAW, TAP_PRE_RS, TAP_RDS_IQ) else a

    def meta(self):
        m = FmoMeta()
        self.L.fmo_chain_meta(self.h, C.byref(m))
        return m

    def peaks(self):
        """showPeakLevel events so far as [events, 2] (leftDb, rightDb)."""
        n = self.L.fmo_chain_peaks(self.h, None, 0)
        a = np.zeros((max(n, 1), 2), np.float32)
        self.L.fmo_chain_peaks(self.h, fptr(a), n)
        return a[:n]

    def rds_bits(self):
        n = self.L.fmo_chain_rds_bits(self.h, None, 0)
        b = np.zeros(max(n, 1), np.uint8)
        self.L.fmo_chain_rds_bits(self.h, u8ptr(b), n)
        return b[:n]

    def close(self):
        if self.h:
            self.L.fmo_chain_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
