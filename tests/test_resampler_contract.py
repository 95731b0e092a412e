"""The one stage of the chain that cannot be pinned by reference code -- the 192 kHz -> 48 kHz converter (newconverter.cpp:37 wraps
libsamplerate's SRC_SINC_MEDIUM_QUALITY, a third-party library that is neither vendored nor present) -- gets a numeric contract
instead: what the fmx resampler (128-tap Kaiser windowed sinc, decimate by 4; the same taps in the oracle and, bit for bit, in the
GPU library: test_tap_sets_match_oracle_design) guarantees, next to what libsamplerate publishes for the converter it replaces
(documentation of 0.1.9 / 0.2.x: SRC_SINC_MEDIUM_QUALITY "121 dB SNR, 90 % bandwidth", linear phase).

    quantity                                   fmx resampler (asserted below)         libsamplerate medium (published)
    DC gain                                    1 to 1e-6                              1
    phase                                      linear, delay 63.5 samples @ 192 kHz   linear
    pass band 0 - 15 kHz (the audio filter's)  ripple < 0.001 dB                      flat
    response at 90 % of Nyquist (21.6 kHz)     -0.4 dB (> -0.5 dB)                    about -3 dB ("bandwidth")
    response at Nyquist (24 kHz)               -6.0 dB                                stop band begins
    images of 0 - 15 kHz (input >= 33 kHz)     below -95 dB                           below -121 dB
    images of 0 - 19 kHz (input >= 29 kHz)     below -90 dB                           below -121 dB
What reaches the converter has already passed the 15 kHz audio low-pass (756 taps, fm-processor.cpp:589-591) with the
de-emphasis behind it, so everything above 29 kHz is at least 80 dB down on entry: the aliased remainder is below -170 dB of full
scale, four orders of magnitude under the 1e-5 RMS tolerance of the PCM comparison."""
import ctypes as C

import numpy as np

import oracle_lib as ol


def taps():
    L = ol.oracle()
    L.fmo_resampler_taps.restype = None
    L.fmo_resampler_taps.argtypes = [C.POINTER(C.c_float)]
    h = np.zeros(128, np.float32)
    L.fmo_resampler_taps(ol.fptr(h))
    return h


def test_resampler_numeric_contract():
    h = taps()
    assert abs(float(h.astype(np.float64).sum()) - 1.0) < 1e-6                     # unity DC gain
    assert np.array_equal(h, h[::-1])                                              # linear phase: delay (128 - 1) / 2 input samples
    n = 1 << 16
    H = np.fft.rfft(h.astype(np.float64), n)
    f = np.arange(H.size) * 192000.0 / n
    mag = 20 * np.log10(np.maximum(np.abs(H), 1e-300))
    pb = mag[f <= 15000]
    assert pb.max() - pb.min() < 1e-3                                              # ripple over the audio band
    assert -0.5 < mag[np.searchsorted(f, 21600)] < 0.0                             # 90 % of the output Nyquist
    assert abs(mag[np.searchsorted(f, 24000)] + 6.02) < 0.05                       # half amplitude at Nyquist (windowed sinc)
    assert mag[f >= 33000].max() < -95.0                                           # what can fold into 0 - 15 kHz
    assert mag[f >= 29000].max() < -90.0                                           # ... into 0 - 19 kHz
    w = np.unwrap(np.angle(H))
    gd = -np.diff(w) / np.diff(2 * np.pi * f / 192000.0)
    sel = f[:-1] <= 15000
    assert abs(gd[sel].min() - 63.5) < 1e-6 and abs(gd[sel].max() - 63.5) < 1e-6   # constant group delay: 15.875 frames at 48 kHz



def test_second_converter_design():
    """fmo_conv2_design (the oracle's and, expression for expression, the library's design::design_conv2): p / q, taps per phase,
    unity DC gain of every phase, the prototype below -80 dB where images / aliases of the 15 kHz audio band fall (and from 8 %
    above the lower Nyquist rate on when that is higher), flat over the audio band."""
    import ctypes as C
    import numpy as np
    import oracle_lib as ol
    L = ol.oracle()
    L.fmo_conv2_design.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_float)]
    L.fmo_conv2_design.restype = C.c_int
    for rate, want in ((44100, (147, 160, 64)), (96000, (2, 1, 32)), (32000, (2, 3, 64)), (22050, (147, 320, 96))):
        p, q, nt = C.c_int32(), C.c_int32(), C.c_int32()
        assert L.fmo_conv2_design(48000, rate, C.byref(p), C.byref(q), C.byref(nt), None) == 0
        assert (p.value, q.value, nt.value) == want
        taps = np.zeros(p.value * nt.value, np.float32)
        assert L.fmo_conv2_design(48000, rate, C.byref(p), C.byref(q), C.byref(nt), taps.ctypes.data_as(C.POINTER(C.c_float))) == 0
        t = taps.reshape(p.value, nt.value).astype(np.float64)
        assert np.max(np.abs(t.sum(axis=1) - 1.0)) < 2e-3                     # every phase passes DC with gain 1
        proto = np.zeros(p.value * nt.value); proto[:] = t.T.reshape(-1) / p.value      # h[k p + ph]
        H = np.abs(np.fft.rfft(proto, 1 << 18))
        fgrid = np.arange(H.size) / float(1 << 18) * p.value * 48000.0        # prototype runs at p * 48 kHz
        lo = min(rate, 48000)
        stop = fgrid >= max(0.54 * lo, lo - 15000.0)
        assert 20 * np.log10(H[stop].max() / H[0]) < -80.0
        passb = fgrid <= min(15000.0, 0.4 * lo)
        assert np.max(np.abs(20 * np.log10(H[passb] / H[0]))) < 0.05          # flat over the 15 kHz audio band
