"""The oracle against the reference's own classes executed LIVE (oracle/_ref/libfmref.so, built from
/root/reference in place by oracle/Makefile).  Skipped where that library is absent; the committed
golden vectors (test_oracle_golden.py) cover the same ground there.  Uses fresh random inputs each
parametrisation, so it pins more inputs than the fixtures can hold."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import fptr, u8ptr

R = ol.ref()
O = ol.oracle()
pytestmark = pytest.mark.skipif(R is None, reason="oracle/_ref/libfmref.so not built (needs /root/reference)")


def same(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@pytest.mark.parametrize("seed", [1, 2])
def test_fft_and_overlap_add(seed):
    rng = np.random.default_rng(seed)
    for n in (8, 1024, 65536):
        x = rng.standard_normal(2 * n).astype(np.float32)
        a, b = x.copy(), x.copy()
        O.fmo_fft_radix2(fptr(a), n, 0); R.ref_fft(fptr(b), n, 0)
        assert same(a, b), n
    for size, deg, low, rate, n in [(2048, 295, 15000, 192000, 9000), (8192, 756, 15000, 192000, 20000),
                                    (65536, 251, 82500, 2304000, 140000)]:
        x = rng.standard_normal(2 * n).astype(np.float32)
        fo = O.fmo_fftfilter_new(size, deg); O.fmo_fftfilter_set_lowpass(fo, low, rate)
        fr = R.ref_fftfilter_new(size, deg); R.ref_fftfilter_set_lowpass(fr, low, rate)
        a, b = np.zeros_like(x), np.zeros_like(x)
        O.fmo_fftfilter_run_c(fo, fptr(x), fptr(a), n); R.ref_fftfilter_run_c(fr, fptr(x), fptr(b), n)
        O.fmo_fftfilter_free(fo); R.ref_fftfilter_free(fr)
        assert same(a, b), size
    n = 70000
    x = rng.standard_normal(n).astype(np.float32)
    ho = O.fmo_fftfilter_new(32768, 768); O.fmo_fftfilter_set_hilbert(ho)
    hr = R.ref_fftfilter_hilbert_new(32768, 768)
    xc = np.zeros(2 * n, np.float32); xc[0::2] = x
    a, b = np.zeros(2 * n, np.float32), np.zeros(2 * n, np.float32)
    O.fmo_fftfilter_run_c(ho, fptr(xc), fptr(a), n); R.ref_fftfilter_hilbert_run(hr, fptr(x), fptr(b), n)
    O.fmo_fftfilter_free(ho); R.ref_fftfilter_hilbert_free(hr)
    assert same(a, b)


@pytest.mark.parametrize("N,low,fs,D,n", [(25, 96000, 2304000, 6, 30000), (3, 96000, 384000, 2, 9999), (11, 12000, 192000, 8, 8000)])
def test_decimators(N, low, fs, D, n):
    rng = np.random.default_rng(N)
    x = rng.standard_normal(2 * n).astype(np.float32)
    do = O.fmo_decim_new(N, low, fs, D); dr = R.ref_decim_new(N, low, fs, D)
    a = np.zeros(2 * (n // D + 1), np.float32); b = np.zeros_like(a)
    ma = O.fmo_decim_run(do, fptr(x), n, fptr(a)); mb = R.ref_decim_run(dr, fptr(x), n, fptr(b))
    O.fmo_decim_free(do); R.ref_decim_free(dr)
    assert ma == mb == n // D and same(a, b)


def test_luts_dense():
    rng = np.random.default_rng(7)
    n = 300000
    ph = np.concatenate([rng.uniform(-40, 40, n // 2), rng.uniform(0, 2 * np.pi, n // 2)]).astype(np.float32)
    so = O.fmo_sincos_new(192000); sr = R.ref_sincos_new(192000)
    s1, c1, z1 = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(2 * n, np.float32)
    s2, c2, z2 = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(2 * n, np.float32)
    O.fmo_sincos_eval(so, fptr(ph), n, fptr(s1), fptr(c1), fptr(z1))
    R.ref_sincos_eval(sr, fptr(ph), n, fptr(s2), fptr(c2), fptr(z2))
    O.fmo_sincos_free(so); R.ref_sincos_free(sr)
    assert same(s1, s2) and same(c1, c2) and same(z1, z2)
    y = rng.standard_normal(n).astype(np.float32) * rng.choice([1e-3, 1.0, 50.0], n).astype(np.float32)
    x = rng.standard_normal(n).astype(np.float32)
    a, b = np.zeros(n, np.float32), np.zeros(n, np.float32)
    O.fmo_atan2_eval(fptr(y), fptr(x), n, fptr(a)); R.ref_atan2_eval(fptr(y), fptr(x), n, fptr(b))
    assert same(a, b)


@pytest.mark.skipif(R is not None and not R.ref_has_qt(), reason="fm-demodulator.cpp needs the image's QtCore")
@pytest.mark.parametrize("bw", [0, 165000])
def test_chain_long(bw):
    """>= 2.1 s of stereo FM: pilot lock at 0.5 s, PSS loop active; every tap bit-identical."""
    N = 16384 * 300
    iq = ol.synth_iq(N)
    ch = ol.OracleChain(taps=[ol.TAP_FM_IQ, ol.TAP_DEMOD, ol.TAP_LRRAW, ol.TAP_PRE_RS], inputFilterBw=bw, tap_seconds=2.3)
    ch.process(iq)
    assert ch.meta().pilotLocked == 1
    rc = R.ref_chain_new(2304000, 192000, 3, bw, 15000, 50, -6.0, 0, 1, 1, 1, 0, 0)
    nf = N // 12 + 8
    fm = np.zeros((nf, 2), np.float32); dm = np.zeros(nf, np.float32)
    lr = np.zeros((nf, 2), np.float32); pr = np.zeros((nf, 2), np.float32)
    m = R.ref_chain_run(rc, fptr(iq), N, fptr(fm), fptr(dm), fptr(lr), fptr(pr))
    R.ref_chain_free(rc)
    assert m == N // 12
    assert same(ch.tap(ol.TAP_FM_IQ), fm[:m]) and same(ch.tap(ol.TAP_DEMOD), dm[:m])
    assert same(ch.tap(ol.TAP_LRRAW), lr[:m]) and same(ch.tap(ol.TAP_PRE_RS), pr[:m])
    assert np.abs(lr[m - 1000:m, 1]).max() > 0.05          # the L-R path is alive (stereo decoded)


@pytest.mark.parametrize("kind,order,f1,ftype,f2,fs", [
    (1, 20, 69900, 0o100, 0, 192000), (0, 20, 70000, 0o100, 0, 192000), (0, 7, 3000, 0o101, 0, 192000), (1, 5, 10000, 0o101, 0, 192000),
    (0, 9, 20000, 0o100, 0, 192000), (0, 6, 96000, 0o100, 0, 192000), (1, 3, 500, 0o101, 0, 192000),
    (2, 7, 1181, 0o101, 1193, 24000), (2, 4, 3000, 0o101, 5000, 48000), (2, 6, 10000, 0o100, 20000, 192000), (2, 3, 100, 0o101, 200, 24000),
    (2, 5, 13000, 0o101, 14000, 24000)])
def test_recursive_filters_dense(kind, order, f1, ftype, f2, fs):
    """LowPassIIR / HighPassIIR / BandPassIIR of the reference (iir-filters.cpp) against the oracle's restatement: coefficients and a long
    response, bit for bit (among them the noise squelch's filters and the `2 * fpass >= fs` guard)."""
    R = ol.ref()
    if R is None:
        pytest.skip("oracle/_ref/libfmref.so not built (no reference tree)")
    L = ol.oracle()
    x = (0.3 * np.random.default_rng(order * 1000 + kind).standard_normal(50000)).astype(np.float32)
    a, b = L.fmo_iir_new(kind, order, f1, f2, fs, ftype), R.ref_iir_new(kind, order, f1, f2, fs, ftype)
    ca, cb = np.zeros(128, np.float32), np.zeros(128, np.float32)
    na, nb = L.fmo_iir_coeffs(a, ol.fptr(ca)), R.ref_iir_coeffs(b, ol.fptr(cb))
    ya, yb = np.zeros_like(x), np.zeros_like(x)
    L.fmo_iir_run(a, ol.fptr(x), x.size, ol.fptr(ya))
    R.ref_iir_run(b, ol.fptr(x), x.size, ol.fptr(yb))
    L.fmo_iir_free(a); R.ref_iir_free(b)
    assert na == nb
    assert np.array_equal(ca.view(np.uint32), cb.view(np.uint32))
    assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32))


@pytest.mark.skipif(R is not None and not R.ref_has_qt(), reason="squelchClass.cpp needs the image's QtCore + moc")
@pytest.mark.parametrize("mode", ["noise", "level"])
def test_squelch_object(mode):
    """squelch (src/various/squelchClass.cpp:11-113), the reference's own object (moc'ed header, QtCore) against the oracle's
    restatement: output AND getSquelchActive() after every sample, bit for bit -- a programme that fades into noise and back
    for the noise squelch, a carrier level that ramps through the threshold for the level squelch, with slider changes in
    between (setSquelchLevel :33-37) as fm-processor.cpp:410-413 applies them."""
    rng = np.random.default_rng(7)
    fs, n = 192000, 192000 * 2
    t = np.arange(n) / fs
    prog = 0.4 * np.sin(2 * np.pi * 1000 * t) + 0.05 * np.sin(2 * np.pi * 19000 * t)
    noise = rng.standard_normal(n) * 0.6
    fade = np.clip(np.abs(np.sin(2 * np.pi * 0.9 * t)) * 1.6 - 0.3, 0, 1)            # programme <-> wide-band noise
    x = (prog * fade + noise * (1 - fade)).astype(np.float32)
    carrier = (0.02 + 0.5 * np.abs(np.sin(2 * np.pi * 1.3 * t)) ** 3).astype(np.float32) if mode == "level" else None
    a = O.fmo_squelch_new(1, 70000, fs // 20, fs)
    b = R.ref_squelch_new(1, 70000, fs // 20, fs)
    ya, yb = np.zeros(n, np.float32), np.zeros(n, np.float32)
    fa, fb = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    pos = 0
    for k, (level, chunk) in enumerate([(0, 16384), (40, 100000), (70, 77777), (100, 50000), (55, n - 16384 - 100000 - 77777 - 50000)]):
        if k:
            O.fmo_squelch_set_level(a, level); R.ref_squelch_set_level(b, level)
        sl = slice(pos, pos + chunk)
        cin = None if carrier is None else fptr(np.ascontiguousarray(carrier[sl]))
        xin = np.ascontiguousarray(x[sl])
        oa, ob = np.zeros(chunk, np.float32), np.zeros(chunk, np.float32)
        ga, gb = np.zeros(chunk, np.uint8), np.zeros(chunk, np.uint8)
        O.fmo_squelch_run(a, fptr(xin), cin, fptr(oa), u8ptr(ga), chunk)
        R.ref_squelch_run(b, fptr(xin), cin, fptr(ob), u8ptr(gb), chunk)
        ya[sl], yb[sl], fa[sl], fb[sl] = oa, ob, ga, gb
        pos += chunk
    O.fmo_squelch_free(a); R.ref_squelch_free(b)
    assert np.array_equal(fa, fb) and same(ya, yb)
    assert 0 < int(fa.sum()) < n and int(np.abs(np.diff(fa.astype(np.int8))).sum()) >= 3          # the squelch opened and closed
