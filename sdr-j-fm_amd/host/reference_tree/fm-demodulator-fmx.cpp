// src/fm/fm-demodulator-fmx.cpp -- replaces src/fm/fm-demodulator.cpp in the reference tree when fmProcessor runs on libfmx.
// includes/fm/fm-demodulator.h is unchanged.  The discriminators themselves run on the GPU (csrc/fmx_stageb.hip, fmx_demod.hip); what is
// left of the class is what the GUI uses: the list of decoder names (radio.cpp:255) and setDecoder (radio.cpp:1687), which hands the
// choice to fmProcessor through fmx_binding.  demodulate / get_DcComponent / get_carrier_ampl are never called by the replacement
// fmProcessor (it asks the library: fmx_get_meta) and return neutral values.
#include "fm-demodulator.h"
#include "fmx_binding.h"
#include <map>
#include <mutex>

namespace {
std::mutex g_mtx;
std::map<const fm_Demodulator *, int> g_code;
// the names the reference shows, in the order of its selector (fm-demodulator.cpp:60-71, 93-103): index + 1 = FMX_P_FM_DECODER
const char *const kNames[6] = { "AM", "FM PLL Decoder", "FM Mixed Demod", "FM Complex Baseband Delay", "FM Real Baseband Delay",
                                "FM Difference Based" };
}

int fmx_binding::decoder_of(const fm_Demodulator *d) {
    std::lock_guard<std::mutex> lk(g_mtx);
    auto it = g_code.find(d);
    return it == g_code.end() ? 3 : it->second;
}
void fmx_binding::publish_decoder(const fm_Demodulator *d, int code) {
    std::lock_guard<std::mutex> lk(g_mtx);
    g_code[d] = code;
}

fm_Demodulator::fm_Demodulator(int32_t Rate_in) : mySinCos(Rate_in) {
    rateIn = Rate_in; selectedDecoder = 3; myfm_pll = nullptr;
    max_freq_deviation = 0; fm_afc = 0; fm_cvt = 1; K_FM = 1; arcSineSize = 0;
    Imin1 = Qmin1 = Imin2 = Qmin2 = 0; am_carr_ampl = 0;
    fmx_binding::publish_decoder(this, selectedDecoder);
}
fm_Demodulator::~fm_Demodulator() {
    std::lock_guard<std::mutex> lk(g_mtx);
    g_code.erase(this);
}
void fm_Demodulator::setDecoder(const QString &decoder) {
    int code = 2;                                   // an unknown name selects the PLL decoder, as the reference's default branch does
    for (int i = 0; i < 6; i++) if (decoder == kNames[i]) code = i + 1;
    selectedDecoder = (int16_t)code;
    fmx_binding::publish_decoder(this, code);
}
QStringList fm_Demodulator::listNameofDecoder() const {
    QStringList l;
    for (const char *n : kNames) l << n;
    return l;
}
float fm_Demodulator::demodulate(std::complex<float>) { return 0.0f; }
float fm_Demodulator::decodeAM(std::complex<float>) { return 0.0f; }
float fm_Demodulator::get_DcComponent() { return 0.0f; }
float fm_Demodulator::get_carrier_ampl() { return 0.0f; }
