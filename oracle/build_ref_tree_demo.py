#!/usr/bin/env python3
"""Recipe of the LINK-AND-RUN demonstration of the reference-tree binding (tests/test_reference_binding.py; VERDICT r3 next #9c).
Test infrastructure, like everything under oracle/: the three replacement sources of sdr-j-fm_amd/host/reference_tree are compiled against
the reference's real headers and linked with the reference's OWN, unchanged sources -- rds-blocksynchronizer.cpp, rds-groupdecoder.cpp,
rds-group.cpp (+ ebu-codetables.c) and the leaf classes includes/fm/fm-processor.h makes members of fmProcessor --, compiled where they lie
under /root/reference, plus the moc output of the reference's headers and the GUI stand-in of tests/ref_link.  Output ONLY into
oracle/_ref/ (git-ignored; it travels to the GPU box like oracle/_ref/libfmref.so): oracle/_ref/ref_tree_demo.  Nothing under
sdr-j-fm_amd/ links it; it pins nothing in the oracle.
usage: python oracle/build_ref_tree_demo.py   (needs /root/reference, the image's Qt 5.9 with moc, and sdr-j-fm_amd/lib/libfmx.so)"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
QT = "/opt/conda"
MOC = os.path.join(QT, "bin", "moc")
QTCORE = os.path.join(QT, "lib", "libQt5Core.so.5")
BIND = os.path.join(ROOT, "sdr-j-fm_amd", "host", "reference_tree")
SHIM = os.path.join(ROOT, "tests", "shim_headers")
LINKDIR = os.path.join(ROOT, "tests", "ref_link")
LIBDIR = os.path.join(ROOT, "sdr-j-fm_amd", "lib")
OUT = os.path.join(ROOT, "oracle", "_ref", "ref_tree_demo")

# (tests/ref_link/radio.h -- the slots the reference connects to -- in front of the compile-check stand-in of tests/shim_headers)
INCS = ["-I" + LINKDIR, "-I" + SHIM, "-I" + BIND, "-I" + os.path.join(ROOT, "include")] + \
       ["-I" + os.path.join(REF, d) for d in ("includes", "includes/fm", "includes/rds", "includes/various", "devices")] + \
       ["-I" + os.path.join(QT, "include", "qt")] + ["-I" + os.path.join(QT, "include", "qt", m) for m in ("QtCore", "QtGui", "QtWidgets")]
# the reference's own sources, unchanged and in place: the RDS byte work, and the leaf classes fmProcessor's header makes members of
# (constructed, not used: the DSP runs in libfmx)
REF_SOURCES = ["src/rds/rds-blocksynchronizer.cpp", "src/rds/rds-groupdecoder.cpp", "src/rds/rds-group.cpp",
               "src/fm/pilot-recover.cpp", "src/fm/stereo-separation.cpp",
               "src/various/fir-filters.cpp", "src/various/fft-filters.cpp", "src/various/fft-complex.cpp", "src/various/sincos.cpp",
               "src/various/oscillator.cpp", "src/various/pllC.cpp", "src/various/Xtan2.cpp", "src/various/newconverter.cpp",
               "src/various/squelchClass.cpp", "src/various/iir-filters.cpp", "devices/device-handler.cpp"]
REF_MOC = ["includes/fm/fm-processor.h", "includes/rds/rds-decoder.h", "includes/rds/rds-blocksynchronizer.h", "includes/rds/rds-groupdecoder.h",
           "includes/various/squelchClass.h", "devices/device-handler.h"]


def available():
    return os.path.isdir(os.path.join(REF, "includes")) and os.path.exists(MOC) and os.path.exists(QTCORE) and os.path.exists(os.path.join(LIBDIR, "libfmx.so"))


def build(out=OUT):
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        objs = []

        def cc(src, name):
            o = os.path.join(tmp, name + ".o")
            subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-w", "-c"] + INCS + [src, "-o", o])
            objs.append(o)

        for name in ("fm-processor-fmx.cpp", "fm-demodulator-fmx.cpp", "rds-decoder-fmx.cpp"):
            cc(os.path.join(BIND, name), name)
        for src in REF_SOURCES:
            cc(os.path.join(REF, src), "ref_" + os.path.basename(src))
        for hdr in REF_MOC + [os.path.join(LINKDIR, "radio.h")]:
            h = hdr if os.path.isabs(hdr) else os.path.join(REF, hdr)
            m = os.path.join(tmp, "moc_" + os.path.basename(h).replace(".h", ".cpp"))
            subprocess.check_call([MOC] + [i for i in INCS if not i.startswith("-I" + QT)] + [h, "-o", m])
            cc(m, os.path.basename(m))
        cc(os.path.join(LINKDIR, "ref_tree_demo.cpp"), "ref_tree_demo")
        subprocess.check_call(["g++"] + objs + ["-L" + LIBDIR, "-lfmx", QTCORE, "-lpthread", "-Wl,-rpath-link," + os.path.join(QT, "lib"),
                                                "-Wl,--allow-shlib-undefined", "-Wl,-rpath," + LIBDIR, "-o", out])
    return out


if __name__ == "__main__":
    if not available():
        sys.exit("needs /root/reference, %s and %s" % (MOC, os.path.join(LIBDIR, "libfmx.so")))
    print(build())
