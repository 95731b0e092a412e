"""The binding a maintainer adds to the reference tree, compiled for real: the three replacement sources of
sdr-j-fm_amd/host/reference_tree -- fm-processor-fmx.cpp, fm-demodulator-fmx.cpp, rds-decoder-fmx.cpp -- against the REFERENCE'S OWN
headers (includes/fm/fm-processor.h, fm-demodulator.h, includes/rds/rds-decoder.h, devices/device-handler.h, includes/various/ringbuffer.h
and everything they include), plus the moc output of those QObject headers, with the image's Qt 5.9.  Compile only (g++ -c): the GUI,
PortAudio, libsndfile and libsamplerate are not in the image (tests/shim_headers/README.md says what stands in for their headers).
Every member function and signal the reference's fm-processor.h / rds-decoder.h / fm-demodulator.h declare must be defined with the
declared signature -- a mismatch (setDeemphasis (int16_t), setfmRdsSelector (rdsDecoder::ERdsMode), ...) fails here.  CPU only; skipped
where the reference tree or Qt is absent."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
QT = "/opt/conda"
MOC = os.path.join(QT, "bin", "moc")
BIND = os.path.join(ROOT, "sdr-j-fm_amd", "host", "reference_tree")
SHIM = os.path.join(ROOT, "tests", "shim_headers")

pytestmark = pytest.mark.skipif(not (os.path.isdir(os.path.join(REF, "includes")) and os.path.exists(MOC)),
                                reason="needs the reference tree and the image's Qt (moc, QtCore headers)")

INCS = ["-I" + SHIM, "-I" + BIND, "-I" + os.path.join(ROOT, "include")] + \
       ["-I" + os.path.join(REF, d) for d in ("includes", "includes/fm", "includes/rds", "includes/various", "devices")] + \
       ["-I" + os.path.join(QT, "include", "qt")] + ["-I" + os.path.join(QT, "include", "qt", m) for m in ("QtCore", "QtGui", "QtWidgets")]
CXX = ["g++", "-std=c++17", "-O1", "-fPIC", "-w", "-c"]


def compile_obj(src, out):
    subprocess.check_call(CXX + INCS + [src, "-o", out])
    return out


def defined_symbols(obj):
    txt = subprocess.check_output(["nm", "-C", "--defined-only", obj]).decode()
    return txt


def test_binding_compiles_against_the_reference_headers(tmp_path):
    objs = {}
    for name in ("fm-processor-fmx.cpp", "fm-demodulator-fmx.cpp", "rds-decoder-fmx.cpp"):
        objs[name] = compile_obj(os.path.join(BIND, name), str(tmp_path / (name + ".o")))
    # the moc output of the reference's own QObject headers compiles next to it (signals = the reference's)
    for hdr in ("includes/fm/fm-processor.h", "includes/rds/rds-decoder.h"):
        m = str(tmp_path / ("moc_" + os.path.basename(hdr).replace(".h", ".cpp")))
        subprocess.check_call([MOC] + [i for i in INCS if not i.startswith("-I" + QT)] + [os.path.join(REF, hdr), "-o", m])
        compile_obj(m, m + ".o")
    # every non-inline member function the reference's class declarations name is defined by the replacement, with that signature
    want = {"fm-processor-fmx.cpp": ("includes/fm/fm-processor.h", "fmProcessor"),
            "fm-demodulator-fmx.cpp": ("includes/fm/fm-demodulator.h", "fm_Demodulator"),
            "rds-decoder-fmx.cpp": ("includes/rds/rds-decoder.h", "rdsDecoder")}
    for src, (hdr, cls) in want.items():
        syms = defined_symbols(objs[src])
        text = open(os.path.join(REF, hdr)).read()
        body = text[text.index("class " + cls if cls != "fm_Demodulator" else "class\tfm_Demodulator"):]
        body = body[:body.index("signals:")] if "signals:" in body else body[:body.index("};")]
        names = set(re.findall(r"\b([A-Za-z_][A-Za-z_0-9]*)\s*\(", re.sub(r"//.*", "", body)))
        names -= {cls, "DSPCOMPLEX", "DelayLine", "defined", "std", "complex", "float", "RingBuffer", "if", "sizeof", "int32_t"}
        for n in sorted(names):
            if not re.search(r"\b%s::%s\(" % (cls, n), syms):
                # (an enum, a member with an initialiser or an inline function is not a function to define)
                assert re.search(r"\b%s\s*\([^;{]*\)\s*(const)?\s*;" % n, body) is None, "%s::%s is declared by %s and not defined by %s" % (cls, n, hdr, src)
        assert re.search(r"\b%s::%s\(" % (cls, cls), syms) and re.search(r"\b%s::~%s\(" % (cls, cls), syms)
    s = defined_symbols(objs["fm-processor-fmx.cpp"])
    assert "fmProcessor::setDeemphasis(short)" in s and "fmProcessor::setfmRdsSelector(rdsDecoder::ERdsMode)" in s
    assert "fmProcessor::startDumping(sf_private_tag*)" in s and "fmProcessor::run()" in s
