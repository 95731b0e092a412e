#!/usr/bin/env python3
"""How far is the REFERENCE from itself on the PLL decoder?  (the soak accepts PLL-decoder channels at 2e-4 instead of 1e-5.)

pllC (pllC.cpp:67-90) senses its phase through two quantised tables -- 192 000 NCO entries, 8192 arc-tangent entries per octant --, so two runs that are one
rounding bit apart somewhere walk through different table entries at sporadic samples from then on.  This script compiles the reference's own leaf classes
(oracle/Makefile's recipe: the sources where they lie under /root/reference, outputs under /tmp) twice -- with the reference's flags (-O2, no contraction:
fmreceiver.pro:12-18) and with -O3 -march=native -ffp-contract=fast (what a distribution or a user's CMAKE_CXX_FLAGS may well produce) -- and runs both over the
soak's kind of streams with the PLL decoder (2) and, as a control, the default decoder (3): RMS difference of the 192 kS/s stereo pair in front of the resampler.
CPU only; needs /root/reference (run in the build container).  Output: profiles/r06_pll_decoder_reference_vs_itself.txt."""
import ctypes as C, os, subprocess, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as ol
REF = "/root/reference"
if not os.path.exists(REF + "/src/various/pllC.cpp"):
    sys.exit("needs the reference tree")
mk = open(os.path.join(R, "oracle", "Makefile")).read()
inc = "-I{0}/includes -I{0}/includes/various -I{0}/includes/fm -I{0}/includes/rds -I{0}/src/rds".format(REF).split()
src = [REF + "/src/various/" + f for f in "fir-filters.cpp fft-filters.cpp fft-complex.cpp sincos.cpp oscillator.cpp pllC.cpp Xtan2.cpp shaping_filter.cpp iir-filters.cpp".split()]
src += [REF + "/src/fm/pilot-recover.cpp", REF + "/src/fm/stereo-separation.cpp", REF + "/src/rds/rds-group.cpp", REF + "/src/fm/fm-demodulator.cpp", REF + "/src/various/squelchClass.cpp"]
qt = "/opt/conda/include/qt"
os.makedirs("/tmp/pllself", exist_ok=True)
subprocess.check_call(["/opt/conda/bin/moc"] + inc + [REF + "/includes/various/squelchClass.h", "-o", "/tmp/pllself/moc_squelchClass.cpp"])
libs = {}
for name, flags in (("reference_flags", ["-O2", "-ffp-contract=off"]), ("O3_native_fma", ["-O3", "-march=native", "-ffp-contract=fast"])):
    out = "/tmp/pllself/libfmref_%s.so" % name
    subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-w", "-DFMREF_WITH_QT", "-I" + qt, "-I" + qt + "/QtCore"] + flags + inc + ["-shared", "-o", out,
                           os.path.join(R, "oracle", "ref_wrap.cpp")] + src + ["/tmp/pllself/moc_squelchClass.cpp", "-L/opt/conda/lib", "-lQt5Core", "-Wl,-rpath,/opt/conda/lib"])
    L = C.CDLL(out)
    L.ref_chain_new.restype = C.c_void_p
    L.ref_chain_new.argtypes = [C.c_int32, C.c_int32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float] + [C.c_int] * 6
    L.ref_chain_run.restype = C.c_long
    L.ref_chain_run.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_long] + [C.POINTER(C.c_float)] * 4
    libs[name] = L
n = 16384 * 90                      # 0.64 s, the soak's length
lines = ["reference against itself: oracle/ref_wrap.cpp + the reference's own leaf classes, g++ -O2 -ffp-contract=off against -O3 -march=native -ffp-contract=fast",
         "%d samples per stream (0.64 s), input filter 165 kHz, audio filter 15 kHz; RMS / max difference of the 192 kS/s stereo pair in front of the resampler" % n, ""]
for seed in (22, 42, 7, 3):
    for sidx in range(3):
        x = ol.synth_iq(n, noiseSeed=100 * seed + sidx, noiseSigma=0.002 * sidx, rds=1, rdsLevel=0.05, rdsBitsSeed=seed * 10 + sidx)
        x[:, 0] += (0.0, 0.007, -0.02)[sidx]; x[:, 1] += (0.0, -0.004, 0.015)[sidx]
        x = np.ascontiguousarray(x, np.float32)
        for dec in (2, 3):
            outs = []
            for name, L in libs.items():
                c = L.ref_chain_new(2304000, 192000, dec, 165000, 15000, 50, -6.0, 0, 1, 1, 1, 0, 0)
                o = np.zeros((n // 12 + 8, 2), np.float32)
                m = L.ref_chain_run(c, x.ctypes.data_as(C.POINTER(C.c_float)), n, None, None, None, o.ctypes.data_as(C.POINTER(C.c_float)))
                outs.append(o[:m].astype(np.float64))
            d = outs[0] - outs[1]
            lines.append("seed %2d stream %d decoder %d (%s): rms %.2e  max %.2e  (signal rms %.3f)" % (seed, sidx, dec, "PLL" if dec == 2 else "Mixed, the default", np.sqrt((d ** 2).mean()), np.abs(d).max(), np.sqrt((outs[0] ** 2).mean())))
txt = "\n".join(lines)
print(txt)
open(os.path.join(R, "profiles", "r06_pll_decoder_reference_vs_itself.txt"), "w").write(txt + "\n")
