// src/fm/fm-processor-fmx.cpp -- replaces src/fm/fm-processor.cpp in the reference tree (fmreceiver.pro / CMakeLists.txt: swap the
// three sources named in fmx_binding.h, add -lfmx).  includes/fm/fm-processor.h is UNCHANGED: same class, same setters, same signals,
// so radio.cpp and every handle_* slot compile and behave as before; the per-sample work of run () happens in libfmx (include/fmx.h).
//
// What stays of the class's data members: everything the GUI-facing bookkeeping needs (rates, scope rings, plot type, counters,
// metaData, dump file, the RDS decoder object, the squelch object as the emitter of setSquelchIsActive).  The DSP members (filters,
// oscillator, pilot PLL, PSS, converter) still have to be constructed because the header declares them by value; they get their
// smallest legal sizes and are never called.  The library handle and the block buffers live in a side table keyed by `this` (the
// header has no room for them).
//
// tests/test_reference_binding.py compiles this file, fm-demodulator-fmx.cpp, rds-decoder-fmx.cpp and the moc output of the reference's
// own fm-processor.h / rds-decoder.h against the reference's real headers (stand-ins only for radio.h, audiosink.h and the two
// third-party headers the image lacks, sndfile.h / samplerate.h, as type declarations).
#include "fm-processor.h"
#include "device-handler.h"
#include "audiosink.h"
#include "radio.h"
#include "fmx_binding.h"
#include <fmx.h>

#include <cmath>
#include <map>
#include <mutex>
#include <vector>

namespace {

constexpr int32_t kBlock = 2 * 8192;                         // bufferSize of run () (fm-processor.cpp:374)
constexpr int32_t kRdsRate = 24000;                          // RDS_RATE (fm-processor.cpp:33)

struct Core {
    fmx_handle h = nullptr;
    std::vector<std::complex<float>> in, dumped, pcm;
    std::vector<float> tap;
    std::vector<uint8_t> bits;
    std::vector<float> sym;
    int32_t lfCount = 0, iqCounter = 0, lastDecoder = -1;
    int16_t lastSquelchValue = -1;
    bool lastSquelch = false, squelchKnown = false;
};
std::mutex g_mtx;
std::map<const fmProcessor *, Core *> g_core;
Core &core_of(const fmProcessor *p) { std::lock_guard<std::mutex> lk(g_mtx); return *g_core.at(p); }

void set(const fmProcessor *p, int id, double v) {
    Core &c = core_of(p);
    if (c.h && fmx_set_param(c.h, 0, id, v) != FMX_OK) qWarning("fmx: %s", fmx_last_error());
}

}  // namespace

// ---- construction (fm-processor.cpp:48-198) -------------------------------------------------------------------------------------
fmProcessor::fmProcessor(deviceHandler *theDevice, RadioInterface *RI, audioSink *mySink, fm_Demodulator *theDemodulator_,
                         int32_t inputRate_, int32_t fmRate_, int32_t workingRate_, int32_t audioRate_, int32_t displaySize_,
                         int spectrumSize_, int32_t repeatRate_, int ptyLocale_, RingBuffer<std::complex<float>> *hfBuffer_,
                         RingBuffer<std::complex<float>> *lfBuffer_, RingBuffer<std::complex<float>> *iqBuffer_, int16_t thresHold_)
    : myRdsDecoder(RI, kRdsRate),                            // used: the reference's own block synchroniser / group decoder
      localOscillator(16), mySinCos(16), pssAGC(1.0f, 0.3f, 2.0f),                        // (never called; smallest sizes)
      fmBand_1(3, 4, 16, 2), fmBand_2(3, 4, 16, 2), fmAudioFilter(16, 3), inputFilter(16, 3),
      pilotRecover(16, 0.1f, 0.1f, &mySinCos), pPSS(16, 0.1f, &mySinCos), rdsBandPassFilter(16, 3), rdsHilbertFilter(16, 3),
      mySquelch(1, 70000, fmRate_ / 20, fmRate_),            // used as the emitter of setSquelchIsActive (fm-processor.cpp:179-180)
      theConverter(workingRate_, audioRate_, workingRate_ / 20) {
    running.store(false);
    theDemodulator = theDemodulator_; myRig = theDevice; myRadioInterface = RI; theSink = mySink;
    inputRate = inputRate_; fmRate = fmRate_; workingRate = workingRate_; audioRate = audioRate_;
    displaySize = displaySize_; spectrumSize = spectrumSize_; repeatRate = repeatRate_; ptyLocale = ptyLocale_;
    hfBuffer = hfBuffer_; lfBuffer = lfBuffer_; iqBuffer = iqBuffer_; thresHold = thresHold_;
    rdsPhaseIndex = 0; newAudioFilter.store(false); inputFilterOn.store(false); newInputFilter.store(false);
    lowPassFrequency = 15000; fmAudioFilterActive.store(false); fmBandwidth = 0; fmFilterDegree = 0;
    lfBuffer_newFlag = true; scanning = false; squelchMode = ESqMode::OFF; loFrequency = 0;
    autoMono = true; pssActive = true; oldSquelchValue = 0; squelchValue = 0; dumping = false; dumpFile = nullptr; myCount = 0;
    Lgain = 1; Rgain = 1; peakLevelCurSampleCnt = 0; peakLevelSampleMax = 0; absPeakLeft = absPeakRight = 0;
    suppressAudioSampleCntMax = workingRate_ / 2; suppressAudioSampleCnt = suppressAudioSampleCntMax;
    pilotDelayPSS = 0; lastAudioSample = 0; deemphAlpha = 1; volumeFactor = 0.5f; panorama = 1; leftChannel = rightChannel = 1;
    fmModus = FM_Mode::Stereo; soundSelector = S_STEREO; rdsModus = rdsDecoder::ERdsMode::RDS_OFF; DCREnabled = true; RfDC = 0;
    lfPlotType = ELfPlot::DEMODULATOR; showFullSpectrum = false; spectrumSampleRate = fmRate_; zoomFactor = 1;

    Core *c = new Core;
    fmx_config cfg{};
    cfg.struct_size = (int32_t)sizeof cfg; cfg.device = 0; cfg.channels = 1;
    cfg.inputRate = inputRate_; cfg.fmRate = fmRate_; cfg.workingRate = workingRate_; cfg.audioRate = audioRate_;
    cfg.max_block = kBlock;
    if (fmx_abi_version() != FMX_ABI_VERSION) qFatal("fmx: libfmx has ABI version %d, this binding was built for %d", fmx_abi_version(), FMX_ABI_VERSION);
    if (fmx_create(&cfg, &c->h) != FMX_OK) qFatal("fmx: %s", fmx_last_error());
    c->in.resize(kBlock); c->dumped.resize(kBlock);
    // PCM frames of one block: the reference decimates inputRate by 12, 6 or not at all (fm-processor.cpp:68-75,471), four fm samples make
    // one 48 kHz frame, the second converter makes audioRate / workingRate of those -- a 192 kS/s device yields 4096 frames per block, not
    // 341.  Sized from what the library itself says the largest call produces (fmx_frames_for follows rate and block phase), plus slack.
    {
        const int64_t per_block = std::max<int64_t>(fmx_frames_for(c->h, kBlock), 0);
        c->pcm.resize((size_t)(per_block + (int64_t)(64 * std::max(audioRate_, workingRate_) / workingRate_) + 64));
    }
    { std::lock_guard<std::mutex> lk(g_mtx); g_core[this] = c; }

    // by name, exactly the connections of fm-processor.cpp:179-192
    connect(&mySquelch, SIGNAL(setSquelchIsActive(bool)), myRadioInterface, SLOT(setSquelchIsActive(bool)));
    connect(this, SIGNAL(hfBufferLoaded()), myRadioInterface, SLOT(hfBufferLoaded()));
    connect(this, SIGNAL(lfBufferLoaded(bool, bool, int)), myRadioInterface, SLOT(lfBufferLoaded(bool, bool, int)));
    connect(this, SIGNAL(iqBufferLoaded()), myRadioInterface, SLOT(iqBufferLoaded()));
    connect(this, SIGNAL(showPeakLevel(float, float)), myRadioInterface, SLOT(showPeakLevel(float, float)));
    connect(this, &fmProcessor::showMetaData, myRadioInterface, &RadioInterface::showMetaData);
    connect(this, SIGNAL(scanresult()), myRadioInterface, SLOT(scanresult()));
}

fmProcessor::~fmProcessor() {
    stop();
    Core *c = nullptr;
    { std::lock_guard<std::mutex> lk(g_mtx); auto it = g_core.find(this); if (it != g_core.end()) { c = it->second; g_core.erase(it); } }
    if (c) { if (c->h) fmx_destroy(c->h); delete c; }
}

void fmProcessor::stop() {                                   // fm-processor.cpp:204-211
    if (running.load()) {
        running.store(false);
        while (!isFinished()) usleep(100);
    }
}

// ---- the setters: the reference's members keep their meaning for the GUI-side logic below, the library gets the setting ----------
void fmProcessor::set_squelchValue(int16_t n) { squelchValue = n; }                        // :213-215 (taken over at the block boundary)
bool fmProcessor::getSquelchState() { fmx_meta m{}; return fmx_get_meta(core_of(this).h, 0, &m) == FMX_OK && m.squelch_active != 0; }
float fmProcessor::get_demodDcComponent() {                                                // :221-226
    if (!running.load()) return 0.0f;
    fmx_meta m{};
    return fmx_get_meta(core_of(this).h, 0, &m) == FMX_OK ? m.live_dc_if : 0.0f;
}
void fmProcessor::setBandwidth(const QString &f) {                                         // :232-239 ("165kHz" | "Off")
    if (f == "Off") { inputFilterOn.store(false); set(this, FMX_P_BANDWIDTH, 0); }
    else { fmBandwidth = Khz(std::stol(f.toStdString())); inputFilterOn.store(true); set(this, FMX_P_BANDWIDTH, fmBandwidth); }
}
void fmProcessor::setfmMode(FM_Mode m) { fmModus = m; set(this, FMX_P_FM_MODE, (int)m); }    // :241-243
void fmProcessor::setFMdecoder(const QString &name) { if (theDemodulator) theDemodulator->setDecoder(name); }
void fmProcessor::setlfPlotType(ELfPlot m) {                                               // :245-266
    lfPlotType = m;
    showFullSpectrum = (m == ELfPlot::IF_FILTERED || m == ELfPlot::RDS_INPUT || m == ELfPlot::RDS_DEMOD);
    spectrumSampleRate = m == ELfPlot::RDS_INPUT ? kRdsRate : (m == ELfPlot::RDS_DEMOD ? kRdsRate / 16 : fmRate);
    lfBuffer_newFlag = true;
}
void fmProcessor::setlfPlotZoomFactor(int32_t z) { zoomFactor = z; lfBuffer_newFlag = true; }
void fmProcessor::setSoundMode(uint8_t selector) { soundSelector = selector; set(this, FMX_P_SOUND_MODE, selector); }
void fmProcessor::setStereoPanorama(int16_t pan) { panorama = (float)pan / 100.0f; set(this, FMX_P_STEREO_PANORAMA, pan); }
void fmProcessor::setSoundBalance(int16_t balance) {                                       // :282-286
    leftChannel = balance > 0 ? (100 - balance) / 100.0 : 1.0f; rightChannel = balance < 0 ? (100 + balance) / 100.0 : 1.0f;
    set(this, FMX_P_SOUND_BALANCE, balance);
}
void fmProcessor::setDeemphasis(int16_t v) { Q_ASSERT(v >= 1); set(this, FMX_P_DEEMPHASIS, v); }   // :291-297
void fmProcessor::setVolume(const float db) { volumeFactor = std::pow(10.0f, db / 20.0f); set(this, FMX_P_VOLUME_DB, db); }
DSPCOMPLEX fmProcessor::audioGainCorrection(DSPCOMPLEX z) {                                // :303-306 (kept for completeness; the gain is applied on the GPU)
    return { volumeFactor * leftChannel * real(z), volumeFactor * rightChannel * imag(z) };
}
void fmProcessor::startDumping(SNDFILE *f) { if (dumping) return; dumpFile = f; dumping = true; }   // :338-345
void fmProcessor::stopDumping() { dumping = false; }
void fmProcessor::setAttenuation(DSPFLOAT l, DSPFLOAT r) { Lgain = l; Rgain = r; set(this, FMX_P_ATTENUATION_L, l); set(this, FMX_P_ATTENUATION_R, r); }
void fmProcessor::startScanning() { scanning = true; }                                     // (scan mode: SURVEY 8 a22, out of scope -- the flag is kept, run () ignores it)
void fmProcessor::stopScanning() { scanning = false; }
void fmProcessor::setlfcutoff(int32_t Hz) {                                                // :762-770
    if (Hz > 0) { lowPassFrequency = Hz; fmAudioFilterActive.store(true); } else fmAudioFilterActive.store(false);
    set(this, FMX_P_LF_CUTOFF, Hz);
}
void fmProcessor::setfmRdsSelector(rdsDecoder::ERdsMode m) {                               // :840-847
    rdsModus = m;
    set(this, FMX_P_RDS_MODE, (int)m);
    if (lfPlotType == ELfPlot::RDS_INPUT || lfPlotType == ELfPlot::RDS_DEMOD) new_lfSpectrum();
}
void fmProcessor::triggerFrequencyChange() {                                               // :849-855
    set(this, FMX_A_TRIGGER_FREQUENCY_CHANGE, 0);             // fade-in, PSS restart and RDS reset inside the library ...
    myRdsDecoder.reset();                                     // ... and the reference's own group decoder, which lives here
    new_lfSpectrum();
}
void fmProcessor::restartPssAnalyzer() { set(this, FMX_A_RESTART_PSS, 0); }                // :857-860
void fmProcessor::resetRds() { myRdsDecoder.reset(); }                                     // :862-864
void fmProcessor::set_localOscillator(int32_t lo) { loFrequency = lo; set(this, FMX_P_LOCAL_OSCILLATOR, lo); }
bool fmProcessor::isPilotLocked(float &oLockStrength) const {                              // :870-880
    fmx_meta m{};
    if (fmModus == FM_Mode::Mono || fmx_get_meta(core_of(this).h, 0, &m) != FMX_OK) { oLockStrength = 0; return false; }
    oLockStrength = m.live_lock_strength;
    return m.live_pilot_locked != 0;
}
void fmProcessor::set_squelchMode(ESqMode m) { squelchMode = m; set(this, FMX_P_SQUELCH_MODE, (int)m); }   // :882-884
void fmProcessor::setAutoMonoMode(const bool b) { autoMono = b; set(this, FMX_P_AUTO_MONO, b); }
void fmProcessor::setPSSMode(const bool b) { pssActive = b; set(this, FMX_P_PSS, b); }
void fmProcessor::setDCRemove(const bool b) { DCREnabled = b; RfDC = 0.0f; set(this, FMX_P_DC_REMOVE, b); }   // :922-925
void fmProcessor::new_lfSpectrum() { lfBuffer_newFlag = true; }
void fmProcessor::setTestTone(const bool b) { testTone.Enabled = b; set(this, FMX_P_TEST_TONE, b); }
void fmProcessor::setDispDelay(const int steps) { delayLine.set_delay_steps(steps); set(this, FMX_P_DISP_DELAY, steps); }
void fmProcessor::set_ptyLocale(int l) { ptyLocale = l; }

// (per-sample helpers of the reference's run (): their work is done on the GPU; kept because the header declares them)
void fmProcessor::sendSampletoOutput(DSPCOMPLEX s) { theSink->putSample(s); }
void fmProcessor::insertTestTone(DSPCOMPLEX &) {}
void fmProcessor::evaluatePeakLevel(const DSPCOMPLEX) {}
void fmProcessor::process_signal_with_rds(const float, std::complex<float> *, std::complex<float> *) {}
DSPFLOAT fmProcessor::getSignal(DSPCOMPLEX *, int32_t) { return 0; }
DSPFLOAT fmProcessor::getNoise(DSPCOMPLEX *, int32_t) { return 0; }
void fmProcessor::processLfSpectrum(std::vector<std::complex<float>> &v, int zoom, bool showFull, bool newFlag) {      // :906-912
    lfBuffer->putDataIntoBuffer(v.data(), spectrumSize);
    emit lfBufferLoaded(showFull, newFlag, zoom);
}

// ---- the thread (fm-processor.cpp:373-687) ------------------------------------------------------------------------------------------
void fmProcessor::run() {
    Core &c = core_of(this);
    running.store(true);
    while (running.load()) {
        while (running.load() && myRig->Samples() < kBlock) msleep(1);                     // :388-390
        if (!running.load()) break;
        // settings that the reference takes over at the block boundary (:396-413); the decoder the GUI chose on its fm_Demodulator
        if (squelchValue != c.lastSquelchValue) { c.lastSquelchValue = squelchValue; oldSquelchValue = squelchValue; fmx_set_param(c.h, 0, FMX_P_SQUELCH_VALUE, squelchValue); }
        const int dec = fmx_binding::decoder_of(theDemodulator);
        if (dec != c.lastDecoder) { c.lastDecoder = dec; fmx_set_param(c.h, 0, FMX_P_FM_DECODER, dec); }

        const int32_t amount = myRig->getSamples(c.in.data(), kBlock, IandQ);              // :416-417
        if (amount <= 0) continue;
        hfBuffer->putDataIntoBuffer(c.in.data(), amount);                                  // :420 the raw block to the HF scope
        emit hfBufferLoaded();                                                             // :421

        // RfDC in front of this block, for the dump below (the removal itself happens in the library)
        fmx_meta before{};
        const bool wantDump = dumping && dumpFile != nullptr;
        if (wantDump) (void)fmx_get_meta(c.h, 0, &before);

        int64_t frames = 0;
        if (fmx_process_host(c.h, reinterpret_cast<const float *>(c.in.data()), amount, amount, reinterpret_cast<float *>(c.pcm.data()),
                             (int64_t)c.pcm.size(), &frames) != FMX_OK) { qWarning("fmx: %s", fmx_last_error()); continue; }
        if (frames > 0) theSink->putSamples(c.pcm.data(), (int32_t)frames);                // :825-838

        if (wantDump) {
            // the reference dumps the block AFTER the RF DC removal (:423-455): its own recurrence from the library's RfDC in front of the block
            std::complex<float> dc(before.live_rf_dc_re, before.live_rf_dc_im);
            const float alpha = 1.0f / inputRate, lim = 0.01f;
            for (int32_t i = 0; i < amount; i++) {
                std::complex<float> x = c.in[(size_t)i];
                if (DCREnabled) {
                    dc = (x - dc) * alpha + dc;
                    x -= std::complex<float>(std::fmin(std::fmax(dc.real(), -lim), lim), std::fmin(std::fmax(dc.imag(), -lim), lim));
                }
                c.dumped[(size_t)i] = x;
            }
            sf_writef_float(dumpFile, reinterpret_cast<float *>(c.dumped.data()), amount);
        }

        // ---- RDS (:553-563): every bit the GPU slicer decided, with the point it was decided on, through the reference's own
        //      rdsBlockSynchronizer / rdsGroupDecoder (rds-decoder-fmx.cpp); the point goes to the IQ scope every 101st time
        if (rdsModus != rdsDecoder::ERdsMode::RDS_OFF) {
            c.bits.resize(1024); c.sym.resize(2 * 1024);
            int32_t nb = 0, ns = 0;
            if (fmx_rds_bits(c.h, 0, c.bits.data(), 1024, &nb) == FMX_OK && nb > 0) {
                if (fmx_rds_symbols(c.h, 0, c.sym.data(), 1024, &ns) != FMX_OK) ns = 0;
                for (int32_t k = 0; k < nb; k++) {
                    DSPCOMPLEX magCplx, point = k < ns ? DSPCOMPLEX(c.sym[(size_t)(2 * k)], c.sym[(size_t)(2 * k + 1)]) : DSPCOMPLEX(0, 0);
                    if (myRdsDecoder.doDecode(point, &magCplx, rdsModus, (ptyLocale & 0xFF) | (c.bits[(size_t)k] ? 0x100 : 0))) {
                        iqBuffer->putDataIntoBuffer(&magCplx, 1);
                        if (++c.iqCounter > 100) { emit iqBufferLoaded(); c.iqCounter = 0; }
                    }
                }
            }
        }

        // ---- LF scope (:566-627, 650-660): one entry per fm sample of this block (per 24 kS/s sample for the RDS views)
        const int64_t nfm = fmx_last_fm_samples(c.h);
        const ELfPlot type = lfPlotType;
        const bool rdsView = type == ELfPlot::RDS_INPUT || type == ELfPlot::RDS_DEMOD;
        const bool rdsOn = rdsModus != rdsDecoder::ERdsMode::RDS_OFF;
        int64_t nrds = 0;
        int per = 0;
        if (nfm > 0) {
            if (rdsView && rdsOn) {
                nrds = fmx_last_rds_samples(c.h);
                c.tap.resize((size_t)(2 * std::max<int64_t>(nrds, 1)));
                if (nrds > 0 && fmx_get_tap(c.h, 0, FMX_TAP_RDS_IQ, c.tap.data(), nrds) != FMX_OK) nrds = 0;
            } else if (!rdsView && type != ELfPlot::OFF) {
                int tap = FMX_TAP_PRE_RESAMPLER; per = 2;
                if (type == ELfPlot::IF_FILTERED) tap = FMX_TAP_FM_IQ;
                else if (type == ELfPlot::DEMODULATOR) { tap = FMX_TAP_DEMOD; per = 1; }
                else if (type == ELfPlot::AF_SUM || type == ELfPlot::AF_DIFF) tap = FMX_TAP_LR_RAW;
                c.tap.resize((size_t)(nfm * per));
                if (fmx_get_tap(c.h, 0, tap, c.tap.data(), nfm) != FMX_OK) per = 0;
            }
        }
        int64_t rdsPushed = 0;
        for (int64_t k = 0; k < nfm; k++) {
            if (rdsView && rdsOn) {
                const int64_t due = ((k + 1) * nrds) / nfm;                                // one decimator output per eight fm samples
                for (; rdsPushed < due; rdsPushed++)
                    spectrumBuffer_lf.push_back((type == ELfPlot::RDS_INPUT ? 20.0f : 1.0f) * std::complex<float>(c.tap[(size_t)(2 * rdsPushed)], c.tap[(size_t)(2 * rdsPushed + 1)]));
            } else {
                const float a = per ? c.tap[(size_t)(per * k)] : 0.f, b = per == 2 ? c.tap[(size_t)(2 * k + 1)] : 0.f;
                std::complex<float> e(0, 0);
                switch (type) {
                case ELfPlot::IF_FILTERED: e = std::complex<float>(a, b); break;
                case ELfPlot::DEMODULATOR: case ELfPlot::AF_SUM: case ELfPlot::AF_LEFT_FILTERED: e = std::complex<float>(a, 0); break;
                case ELfPlot::AF_DIFF: case ELfPlot::AF_RIGHT_FILTERED: e = std::complex<float>(b, 0); break;
                case ELfPlot::AF_MONO_FILTERED: e = std::complex<float>(a + b, 0); break;
                default: break;
                }
                spectrumBuffer_lf.push_back(e);
            }
            if (++c.lfCount > fmRate / repeatRate) {                                       // :650-660
                if (spectrumBuffer_lf.size() >= (unsigned)spectrumSize) {
                    processLfSpectrum(spectrumBuffer_lf, zoomFactor, showFullSpectrum, lfBuffer_newFlag);
                    lfBuffer_newFlag = false;
                    spectrumBuffer_lf.resize(0);
                }
                c.lfCount = 0;
            }
        }

        // ---- showPeakLevel every 961 PCM frames (:772-798), showMetaData every fmRate / 2 + 1 fm samples (:662-684) ----
        float lr[2 * 64]; int32_t npk = 0;
        if (fmx_get_peaks(c.h, 0, lr, 64, &npk) == FMX_OK) for (int32_t k = 0; k < npk; k++) emit showPeakLevel(lr[2 * k], lr[2 * k + 1]);
        myCount += (int32_t)nfm;
        if (myCount > (fmRate >> 1)) {
            fmx_meta m{};
            if (fmx_get_meta(c.h, 0, &m) == FMX_OK) {
                metaData.PilotPllLocked = m.PilotPllLocked != 0; metaData.PilotPllLockStrength = m.PilotPllLockStrength;
                metaData.DcValRf = m.DcValRf; metaData.DcValIf = m.DcValIf; metaData.PssPhaseShiftDegree = m.PssPhaseShiftDegree;
                metaData.PssPhaseChange = m.PssPhaseChange; metaData.PssState = (SMetaData::EPssState)m.PssState;
                emit showMetaData(&metaData);
                const bool sq = m.squelch_active != 0;                                     // squelchClass.cpp:74-77: on change
                if (!c.squelchKnown || sq != c.lastSquelch) { emit mySquelch.setSquelchIsActive(sq); c.lastSquelch = sq; c.squelchKnown = true; }
            }
            myCount -= (fmRate >> 1) + 1;
        }
    }
}
