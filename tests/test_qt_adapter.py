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
    stub = os.path.join(HOSTQT, "gui_stub")
    mocs = []
    for h in (os.path.join(HOSTQT, "fm_processor_qt.h"), os.path.join(stub, "radio_stub.h")):
        m = os.path.join(outdir, "moc_" + os.path.basename(h).replace(".h", ".cpp"))
        subprocess.check_call([MOC, "-I" + HOSTQT, h, "-o", m])
        mocs.append(m)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-I" + HOSTQT, "-I" + stub, "-I" + os.path.join(QT, "include", "qt"),
                           "-I" + os.path.join(QT, "include", "qt", "QtCore"),
                           os.path.join(HOSTQT, "fm_processor_qt.cpp"), os.path.join(HOSTQT, "qt_demo.cpp")] + mocs +
                          ["-L" + LIBDIR, "-lfmx", QTCORE, "-Wl,-rpath-link," + os.path.join(QT, "lib"), "-Wl,--allow-shlib-undefined",
                           "-Wl,-rpath," + LIBDIR, "-o", exe])
    return exe


def parse(out):
    line = out.strip().splitlines()[0].split()
    kv = dict(zip(line[0::2], line[1::2]))
    txt = dict(f.split("=", 1) for f in out.strip().splitlines()[1].split("|"))
    return kv, txt


@needs_qt
def test_qt_adapter_builds(tmp_path):
    if not os.path.exists(os.path.join(LIBDIR, "libfmx.so")):
        pytest.skip("libfmx.so not built")
    exe = build_demo(str(tmp_path))
    r = subprocess.run([exe], env=RUN_ENV, capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@needs_qt
@pytest.mark.gpu
def test_qt_adapter_run(tmp_path, ol, fmx_amd):
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
    kv, _ = parse(out)
    assert int(kv["hf"]) == nblocks                           # hfBufferLoaded once per block
    assert int(kv["hfring"]) == 1                             # ... and the HF ring holds exactly the blocks pulled from the device (:420)
    assert int(kv["peaks"]) == pcm.shape[0] // 961            # showPeakLevel once per 961 PCM frames
    assert int(kv["meta"]) >= 1                               # showMetaData every fmRate / 2 samples
    assert int(kv["locked"]) == 1 and int(kv["squelch"]) >= 1
    # LF scope (:650-660): every fmRate / repeatRate + 1 = 19201 fm samples the first spectrumSize = 2048 entries of the vector
    nfm = 16384 * nblocks // 12
    assert int(kv["lf"]) == nfm // 19201 and int(kv["lfring"]) == 2048 * int(kv["lf"]) and int(kv["lfnew"]) == 1
    assert float(kv["lfimag"]) == 0.0                         # DEMODULATOR view: (demod, 0)
    # ... and they are the oracle's demodulator samples: entry k of emission e is fm sample 19201 e + k
    lf = np.fromfile(str(tmp_path / "pcm.f32.lf"), np.float32).reshape(-1, 2)[:, 0].reshape(-1, 2048)
    o = ol.OracleChain(taps=[ol.TAP_DEMOD], inputFilterBw=165000, tap_seconds=1.4)
    o.process(iq)
    dem = o.tap(ol.TAP_DEMOD)
    for e in range(lf.shape[0]):
        ref = dem[19201 * e: 19201 * e + 2048]
        assert float(np.sqrt(np.mean((lf[e] - ref) ** 2))) <= 2e-5 * max(1.0, float(np.abs(ref).max())), e
    assert int(kv["iq"]) == 0 and int(kv["iqring"]) == 0      # RDS off: nothing reaches the IQ ring


@needs_qt
@pytest.mark.gpu
def test_qt_adapter_rds_signals(tmp_path, ol, fmx_amd):
    """RDS on: the decided symbols reach the IQ ring (iqBufferLoaded every 101), and the group decoder's picture arrives as the
    reference's signals: PI 0xD3A1, PTY with its name, the PS name and the radio text of the generated programme -- a text with
    accented characters and an alphabet-switch pair (0x0F 0x0F), which must arrive as rdsGroupDecoder::prepareText would hand it to
    setRadioText (rds-groupdecoder.cpp:298-315: the pair leaves a blank and swallows the character behind it; the reference's own
    character table, tests/test_rds_text.py)."""
    from test_rds_text import G as RDS_GOLD, prepare_text_reference_loop
    exe = build_demo(str(tmp_path))
    nblocks = 520                                             # 3.7 s: block sync + a full pass over the programme's groups
    raw = b"Caf\x82 \x0f\x0fM\x97nchen \x91 - RDS OK 5\xa9"
    prog = dict(pi=0xD3A1, pty=10, ps="FMX-AMD ", text=raw.decode("latin1"))
    padded = raw + b"\r"
    padded += b" " * (-len(padded) % 4)
    want_text, _ = prepare_text_reference_loop(list(padded + b" " * (64 - len(padded)) + b"\0"), 64, 0, lambda a, c: int(RDS_GOLD["ebu_map"][a][c]))
    assert want_text == "Café  önchen ä - RDS OK 5X"
    bits = ol.rds_programme_bits(**prog)
    iq = ol.synth_iq(16384 * nblocks, rds=1, rdsLevel=0.05, rds_payload=bits)
    iq.tofile(str(tmp_path / "iq.f32"))
    out = subprocess.check_output([exe, str(tmp_path / "iq.f32"), str(tmp_path / "pcm.f32"), "rds"], env=RUN_ENV, timeout=600).decode()
    print("\n[qt adapter, RDS]", out.strip())
    kv, txt = parse(out)
    assert int(kv["rdssync"]) == 1 and int(kv["pi"]) == prog["pi"] and int(kv["pty"]) == prog["pty"]
    assert txt["ptyname"] == "Pop Music" and txt["label"] == prog["ps"] and txt["text"] == want_text
    assert int(kv["groups"]) >= 20 and int(kv["crc"]) <= 2
    nsym = int(kv["iqring"])
    assert abs(nsym - 16384 * nblocks / 2304000.0 * 1187.5) < 30          # one constellation point per RDS bit
    assert int(kv["iq"]) == nsym // 101                                    # iqBufferLoaded every 101 symbols (:558-562)
