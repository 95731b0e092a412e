"""GPU parity tests proper: the HIP path (through the C ABI, include/fmx.h) against the oracle
on identical synthetic IQ.  Tolerance from BASELINE.json north_star: <= 1e-5 RMS on float PCM."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PCM_RMS_TOL = 1e-5          # north_star: "within 1e-5 RMS on the float audio PCM"


def rms(a):
    return float(np.sqrt(np.mean(np.asarray(a, np.float64) ** 2)))


def run_gpu(fmx_amd, iq, block, setup, channels=1):
    f = fmx_amd.Fmx(channels, max_block=block)
    setup(f)
    outs = []
    for i in range(0, iq.shape[0] - block + 1, block):
        outs.append(f.process_host(iq[i:i + block]))
    return f, np.concatenate(outs, axis=1)


def cfg_setup(f, bw, stereo=True, decoder=3):
    P = f.__class__.__module__
    import importlib
    m = importlib.import_module("sdr-j-fm_amd").fmx
    f.set_param(m.P_BANDWIDTH, bw)
    f.set_param(m.P_LF_CUTOFF, 15000)
    f.set_param(m.P_DEEMPHASIS, 50)
    f.set_param(m.P_VOLUME_DB, -6.0)
    f.set_param(m.P_FM_MODE, 0 if stereo else 2)
    f.set_param(m.P_FM_DECODER, decoder)


@pytest.mark.parametrize("name,bw,stereo,seconds", [
    ("config1_mono_filter_off", 0, False, 1.0),
    ("config2_stereo_pss_filter_on", 165000, True, 1.3),
])
def test_chain_parity_short(fmx_amd, ol, name, bw, stereo, seconds):
    block = 16384 * 4
    n = int(seconds * 2304000) // block * block
    iq = ol.synth_iq(n, stereo=1 if stereo else 0)
    ch = ol.OracleChain(taps=[ol.TAP_FM_IQ, ol.TAP_DEMOD, ol.TAP_LRRAW], inputFilterBw=bw,
                        fmMode=0 if stereo else 2, tap_seconds=seconds + 0.1)
    pcm_o = ch.process(iq)
    f, pcm_g = run_gpu(fmx_amd, iq, block, lambda f: cfg_setup(f, bw, stereo))
    pcm_g = pcm_g[0]
    assert pcm_g.shape == pcm_o.shape
    m = fmx_amd.fmx
    nt = block // 12
    z_g = f.tap(m.TAP_FM_IQ, nt)
    z_o = ch.tap(ol.TAP_FM_IQ)[-nt:]
    d_g = f.tap(m.TAP_DEMOD, nt)
    d_o = ch.tap(ol.TAP_DEMOD)[-nt:]
    lr_g = f.tap(m.TAP_LR_RAW, nt)
    lr_o = ch.tap(ol.TAP_LRRAW)[-nt:]
    e_z, e_d, e_lr, e_pcm = rms(z_g - z_o), rms(d_g - d_o), rms(lr_g - lr_o), rms(pcm_g - pcm_o)
    print(f"\n[{name}] rms: fm_iq {e_z:.3e} (sig {rms(z_o):.3f}) demod {e_d:.3e} (sig {rms(d_o):.3f}) "
          f"lr {e_lr:.3e} pcm {e_pcm:.3e} (sig {rms(pcm_o):.3f}) max {np.max(np.abs(pcm_g - pcm_o)):.3e}")
    mg, mo = f.meta(0), ch.meta()
    assert mg.PilotPllLocked == mo.pilotLocked
    assert e_z <= 2e-6 * max(rms(z_o), 1e-3)
    assert e_pcm <= PCM_RMS_TOL
