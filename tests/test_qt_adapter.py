"""The Qt side of the drop-in (sdr-j-fm_amd/host/qt): a QThread with the reference fmProcessor's surface on top of the C ABI,
built with the image's Qt 5.9 (moc + g++ against QtCore) and driven like RadioInterface drives fmProcessor
(radio.cpp:915-948): construct with (device, GUI object, sink), setters, start(), signals by name into the GUI object's
slots (fm-processor.cpp:179-192), stop().  The build runs on CPU; the run needs the GPU."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QT = "/opt/conda"
MOC = os.path.join(QT, "bin", "moc")
QTCORE = os.path.join(QT, "lib", "libQt5Core.so.5")
HOSTQT = os.path.join(ROOT, "sdr-j-fm_amd", "host", "qt")
LIBDIR = os.path.join(ROOT, "sdr-j-fm_amd", "lib")
# conda ships an older libstdc++ next to its Qt; the ROCm runtime needs the system's
RUN_ENV = dict(os.environ, LD_LIBRARY_PATH=os.path.join(QT, "lib"), LD_PRELOAD="/usr/lib/x86_64-linux-gnu/libstdc++.so.6",
               QT_QPA_PLATFORM="offscreen")

needs_qt = pytest.mark.skipif(not (os.path.exists(MOC) and os.path.exists(QTCORE)), reason="no Qt (moc / QtCore) in this image")


def build_demo(outdir):
    exe = os.path.join(outdir, "qt_demo")
    mocs = []
    for h in ("fm_processor_qt.h", "qt_demo.h"):
        m = os.path.join(outdir, "moc_" + h.replace(".h", ".cpp"))
        subprocess.check_call([MOC, os.path.join(HOSTQT, h), "-o", m])
        mocs.append(m)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-I" + HOSTQT, "-I" + os.path.join(QT, "include", "qt"),
                           "-I" + os.path.join(QT, "include", "qt", "QtCore"),
                           os.path.join(HOSTQT, "fm_processor_qt.cpp"), os.path.join(HOSTQT, "qt_demo.cpp")] + mocs +
                          ["-L" + LIBDIR, "-lfmx", QTCORE, "-Wl,-rpath-link," + os.path.join(QT, "lib"), "-Wl,--allow-shlib-undefined",
                           "-Wl,-rpath," + LIBDIR, "-o", exe])
    return exe


@needs_qt
def test_qt_adapter_builds(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "libfmx.so")):
        pytest.skip("libfmx.so not built")
    exe = build_demo(str(tmp_path))
    r = subprocess.run([exe], env=RUN_ENV, capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@needs_qt
@pytest.mark.gpu
def test_qt_adapter_run(tmp_path, ol):
    exe = build_demo(str(tmp_path))
    nblocks = 180                                             # 1.28 s: the second showMetaData snapshot (1.0 s) sees the pilot lock
    n = 16384 * nblocks + 777                                 # the tail < 16384 is never pulled (fm-processor.cpp:388)
    iq = ol.synth_iq(n)
    iq.tofile(str(tmp_path / "iq.f32"))
    out = subprocess.check_output([exe, str(tmp_path / "iq.f32"), str(tmp_path / "pcm.f32")], env=RUN_ENV, timeout=300).decode()
    print("\n[qt adapter]", out.strip())
    pcm = np.fromfile(str(tmp_path / "pcm.f32"), np.float32).reshape(-1, 2)
    want = ol.OracleChain(inputFilterBw=165000).process(iq)
    assert pcm.shape == want.shape, out
    assert float(np.sqrt(np.mean((pcm.astype(np.float64) - want) ** 2))) <= 1e-5
    kv = dict(zip(out.split()[0::2], out.split()[1::2]))
    assert int(kv["hf"]) == nblocks                           # hfBufferLoaded once per block
    assert int(kv["peaks"]) == pcm.shape[0] // 961            # showPeakLevel once per 961 PCM frames
    assert int(kv["meta"]) >= 1                               # showMetaData every fmRate / 2 samples
    assert int(kv["locked"]) == 1 and int(kv["squelch"]) >= 1
