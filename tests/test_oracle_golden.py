"""The oracle (oracle/fm_oracle.c) against the committed golden vectors (tests/golden/ref_vectors.npz),
every "expected" array of which was produced by the REFERENCE's own classes compiled from
/root/reference (tests/golden/make_golden.py).  Bit-exact: the oracle restates the reference's
arithmetic including its f32/f64 promotion.  Runs on CPU, needs neither the reference tree nor a GPU."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import fptr, u8ptr

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_vectors.npz"))
O = ol.oracle()


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_bitexact(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, what
    same = bits(a) == bits(b)
    # NaN payloads aside, every element must be identical
    assert same.all(), "%s: %d of %d elements differ (max |d| %g)" % (
        what, (~same).sum(), same.size, np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64))))


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


@pytest.mark.parametrize("name,N,Fc,fs", [("lowpass_input_251", 251, 82500, 2304000),
                                          ("lowpass_audio_756", 756, 15000, 192000),
                                          ("lowpass_pss_295", 295, 15000, 192000)])
def test_lowpass_kernel(name, N, Fc, fs):
    a = np.zeros(N, np.float32)
    O.fmo_lowpass_kernel(N, Fc, fs, fptr(a))
    assert_bitexact(a, G[name], name)


@pytest.mark.parametrize("name,N,low,fs", [("decim_band1_25", 25, 96000, 2304000), ("decim_band2_3", 3, 96000, 384000),
                                           ("decim_rds_11", 11, 12000, 192000)])
def test_decim_kernel(name, N, low, fs):
    a = np.zeros(2 * N, np.float32)
    O.fmo_decim_kernel(N, low, fs, fptr(a))
    assert_bitexact(a, G[name], name)
    # the (h/sum, h) quirk: imaginary taps sum to the un-normalised sum (SURVEY 7 "quirks")
    assert abs(a[0::2].sum() - 1.0) < 1e-6 and a[1::2].sum() > 1.5


def test_bandpass_and_rrc():
    a = np.zeros(2 * 768, np.float32)
    O.fmo_bandpass_kernel(768, 54600, 59400, 192000, fptr(a))
    assert_bitexact(a, G["bandpass_rds_768"], "bandpass")
    t = np.zeros(64, np.float32)
    n = O.fmo_rrc_kernel(1.0, 24000.0, 2375.0, 1.0, 45, fptr(t))
    assert n == 45
    assert_bitexact(t[:n], G["rrc_45"], "rrc")


def test_fft_radix2():
    y = G["fft2048_in"].copy()
    assert O.fmo_fft_radix2(fptr(y), 2048, 0) == 1
    assert_bitexact(y, G["fft2048_out"], "fft 2048")
    bad = np.zeros(2 * 12, np.float32)
    assert O.fmo_fft_radix2(fptr(bad), 12, 0) == 0          # not a power of two


def test_overlap_add_filters():
    x = G["ola_pss_in"].copy()
    f = O.fmo_fftfilter_new(2048, 295)
    O.fmo_fftfilter_set_lowpass(f, 15000, 192000)
    y = np.zeros_like(x)
    O.fmo_fftfilter_run_c(f, fptr(x), fptr(y), x.size // 2)
    O.fmo_fftfilter_free(f)
    assert_bitexact(y, G["ola_pss_out"], "overlap-add complex (PSS low-pass)")
    assert not y[: 2 * 1753].any() and y[2 * 1753:].any()   # pure latency fftSize - degree (fft-filters.cpp:34)
    # real-input x3 variant (RDS band-pass): regenerate the generator's third draw
    rng = np.random.default_rng(20250929)
    rng.standard_normal(2 * 2048); rng.standard_normal(2 * 6000)
    xr = rng.standard_normal(40000).astype(np.float32)
    assert crc(xr) == int(G["ola_rdsbp_in_crc"])
    f = O.fmo_fftfilter_new(32768, 768)
    O.fmo_fftfilter_set_band(f, 54600, 59400, 192000)
    yr = np.zeros_like(xr)
    O.fmo_fftfilter_run_r(f, fptr(xr), fptr(yr), 40000)
    O.fmo_fftfilter_free(f)
    assert_bitexact(yr[32000:33000], G["ola_rdsbp_out_tail"], "overlap-add real x3 (RDS band-pass)")


def test_decimating_fir():
    x = G["decim25_in"].copy()
    d = O.fmo_decim_new(25, 96000, 2304000, 6)
    y = np.zeros(2 * 1001, np.float32)
    m = O.fmo_decim_run(d, fptr(x), 6001, fptr(y))
    O.fmo_decim_free(d)
    assert m == 1000                                          # output on every 6th call (fir-filters.cpp:401-405)
    assert_bitexact(y[:2 * m], G["decim25_out"], "DecimatingFIR 25/6")


def test_luts():
    ph = G["sincos_phase"].copy()
    n = ph.size
    t = O.fmo_sincos_new(192000)
    s = np.zeros(n, np.float32); c = np.zeros(n, np.float32); z = np.zeros(2 * n, np.float32)
    O.fmo_sincos_eval(t, fptr(ph), n, fptr(s), fptr(c), fptr(z))
    O.fmo_sincos_free(t)
    assert_bitexact(s, G["sincos_sin"], "SinCos::getSin")
    assert_bitexact(c, G["sincos_cos"], "SinCos::getCos")
    assert_bitexact(z, G["sincos_cplx"], "SinCos::getComplex")
    y, x = G["atan2_y"].copy(), G["atan2_x"].copy()
    o = np.zeros(y.size, np.float32)
    O.fmo_atan2_eval(fptr(y), fptr(x), y.size, fptr(o))
    assert_bitexact(o, G["atan2_out"], "compAtan::atan2")
    v = G["pic_in"].copy()
    o = np.zeros_like(v)
    O.fmo_pi_constrain_eval(fptr(v), v.size, fptr(o))
    assert_bitexact(o, G["pic_out"], "PI_Constrain")

    class c32(C.Structure):
        _fields_ = [("re", C.c_float), ("im", C.c_float)]
    O.fmo_lo_value.restype = c32
    O.fmo_lo_value.argtypes = [C.c_int32, C.c_int32]
    lo, p = [], 0
    for _ in range(64):                                       # Oscillator::nextValue(200000) oscillator.cpp:49-58
        p -= 200000
        if p < 0:
            p += 2304000
        w = O.fmo_lo_value(2304000, p)
        lo += [w.re, w.im]
    assert_bitexact(np.array(lo, np.float32), G["lo_step200k"], "Oscillator table")


@pytest.mark.parametrize("decoder", [1, 2, 3, 4, 5, 6])
def test_discriminators(decoder):
    z = G["demod_in"].copy()
    o = np.zeros(z.shape[0], np.float32)
    O.fmo_demod_run(192000, decoder, fptr(z), z.shape[0], fptr(o), None, None)
    assert_bitexact(o, G["demod_out_%d" % decoder], "fm_Demodulator decoder %d" % decoder)


def test_pilot_pss_agc_costas():
    om, g, pa = C.c_float(), C.c_float(), C.c_float()
    O.fmo_pilot_constants(192000, C.byref(om), C.byref(g), C.byref(pa))
    p5 = G["pilot_in"].copy(); n = p5.size
    ph = np.zeros(n, np.float32); lk = np.zeros(n, np.uint8); st = np.zeros(n, np.float32)
    O.fmo_pilot_run(192000, om.value, g.value, fptr(p5), n, fptr(ph), u8ptr(lk), fptr(st))
    assert_bitexact(ph, G["pilot_phase"], "pilotRecovery phase")
    assert_bitexact(st, G["pilot_strength"], "pilotRecovery lock strength")
    mux, phh, rs = G["pss_mux"].copy(), G["pss_phase"].copy(), G["pss_reset"].copy()
    o = np.zeros(mux.size, np.float32)
    O.fmo_pss_run(192000, pa.value, fptr(mux), fptr(phh), mux.size, fptr(o), u8ptr(rs))
    assert_bitexact(o, G["pss_out"], "PerfectStereoSeparation")
    x = G["agc_in"].copy(); o = np.zeros_like(x)
    O.fmo_agc_run(2e-3, 0.38, 9.0, fptr(x), x.size // 2, fptr(o))
    assert_bitexact(o, G["agc_out"], "AGC")
    O.fmo_costas_run(24000.0, 1.0, 0.02, 10.0, fptr(x), x.size // 2, fptr(o))
    assert_bitexact(o, G["costas_out"], "Costas")


def test_chain_against_reference_wired_leaf_classes():
    """fmProcessor::run() glue: the oracle chain vs the harness that wires the reference's own leaf
    objects in the same order (oracle/ref_wrap.cpp ref_chain_*), through the pre-resampler tap."""
    n = int(G["chain_iq_n"])
    iq = ol.synth_iq(n)
    assert crc(iq) == int(G["chain_iq_crc"]), "synthetic IQ not reproducible on this host (libm differs?)"
    ch = ol.OracleChain(taps=[ol.TAP_FM_IQ, ol.TAP_DEMOD, ol.TAP_LRRAW, ol.TAP_PRE_RS], inputFilterBw=165000, tap_seconds=0.3)
    ch.process(iq)
    m = int(G["chain_m"])
    for tap, key in [(ol.TAP_FM_IQ, "fm"), (ol.TAP_DEMOD, "demod"), (ol.TAP_LRRAW, "lr"), (ol.TAP_PRE_RS, "prers")]:
        a = ch.tap(tap)
        assert a.shape[0] == m
        assert crc(a) == int(G["chain_%s_crc" % key]), key
        assert_bitexact(a[m - 4096:], G["chain_%s_tail" % key], "chain tap " + key)
    iq1 = ol.synth_iq(n, stereo=0)
    assert crc(iq1) == int(G["chain1_iq_crc"])
    ch = ol.OracleChain(taps=[ol.TAP_DEMOD, ol.TAP_PRE_RS], inputFilterBw=0, fmMode=2, tap_seconds=0.3)
    ch.process(iq1)
    assert crc(ch.tap(ol.TAP_DEMOD)) == int(G["chain1_demod_crc"])
    assert crc(ch.tap(ol.TAP_PRE_RS)) == int(G["chain1_prers_crc"])
    assert_bitexact(ch.tap(ol.TAP_PRE_RS)[-4096:], G["chain1_prers_tail"], "config-1 pre-resampler tap")


def test_recursive_filters():
    """iir-filters.cpp (LowPassIIR / HighPassIIR / BandPassIIR, Chebyshev and Butterworth prototypes, bilinear transform, Basic_IIR::Pass):
    coefficients and responses of the reference's own classes (tests/golden/ref_iir.npz), among them the two order-20
    filters of the noise squelch (squelchClass.cpp:11-18) and the band-pass of rdsDecoder_1 (rds-decoder-1.cpp:45-48)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    GI = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_iir.npz"))
    x = GI["iir_in"]
    for name, kind, order, f1, ftype, f2, fs in mg.IIR_CASES:
        f = O.fmo_iir_new(kind, order, f1, f2, fs, ftype)
        c = np.zeros(128, np.float32)
        nq = O.fmo_iir_coeffs(f, fptr(c))
        y = np.zeros_like(x)
        O.fmo_iir_run(f, fptr(x), x.size, fptr(y))
        O.fmo_iir_free(f)
        assert_bitexact(c[:6 * nq + 1], GI[name + "_coef"], name + " coefficients")
        assert_bitexact(y, GI[name + "_out"], name + " response")


@pytest.mark.parametrize("mode", ["noise", "level"])
def test_squelch_object(mode):
    """squelch (squelchClass.cpp:11-113): getSquelchActive() transitions and the output of the reference's own object
    (tests/golden/ref_squelch.npz, made from the moc'ed class by make_golden.py) reproduced by the oracle, bit for bit."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    GS = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_squelch.npz"))
    x, carrier, schedule = mg.squelch_case(mode)
    q = O.fmo_squelch_new(1, 70000, 192000 // 20, 192000)
    y = np.zeros_like(x); fl = np.zeros(x.size, np.uint8)
    pos = 0
    for k, (level, chunk) in enumerate(schedule):
        if k:
            O.fmo_squelch_set_level(q, level)
        xin = np.ascontiguousarray(x[pos:pos + chunk]); yo = np.zeros(chunk, np.float32); fo = np.zeros(chunk, np.uint8)
        cin = None if carrier is None else fptr(np.ascontiguousarray(carrier[pos:pos + chunk]))
        O.fmo_squelch_run(q, fptr(xin), cin, fptr(yo), ol.u8ptr(fo), chunk)
        y[pos:pos + chunk] = yo; fl[pos:pos + chunk] = fo; pos += chunk
    O.fmo_squelch_free(q)
    assert fl[0] == GS[mode + "_flag0"][0]
    assert np.array_equal(np.nonzero(np.diff(fl.astype(np.int8)))[0], GS[mode + "_transitions"]) and len(GS[mode + "_transitions"]) >= 3
    assert_bitexact(y[:4096], GS[mode + "_out_head"], mode + " squelch output")
    assert mg.crc(y) == int(GS[mode + "_out_crc"][0])
