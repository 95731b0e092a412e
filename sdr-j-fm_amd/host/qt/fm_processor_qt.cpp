// fm_processor_qt.cpp -- see fm_processor_qt.h.  Build: moc fm_processor_qt.h -o moc_fm_processor_qt.cpp;
//   g++ -std=c++17 -fPIC -I<dir of fmx_qt_host.h> -I$QT/include/qt -I$QT/include/qt/QtCore fm_processor_qt.cpp
//       moc_fm_processor_qt.cpp -lQt5Core -lfmx
#include "fm_processor_qt.h"
#include "fmx_qt_host.h"       // deviceHandler, audioSink, RadioInterface, RingBuffer<>: the GUI's own headers (see fm_processor_qt.h)

#include <cmath>
#include <cstring>

namespace fmx_qt {

void (*fmProcessor::dumpWriter)(sf_private_tag *, const float *, int32_t) = nullptr;

namespace {
// the two calls run() makes on the device and the sink, on the GUI's own classes
struct DeviceOfGui : fmx_host::DeviceHandler {
    deviceHandler *dev;
    explicit DeviceOfGui(deviceHandler *d) : dev(d) {}
    int32_t Samples() override { return dev->Samples(); }
    int32_t getSamples(std::complex<float> *dst, int32_t n) override { return dev->getSamples(dst, n, 0100 /* IandQ, fm-constants.h:86 */); }
    int32_t getRate() override { return dev->getRate(); }
};
struct SinkOfGui : fmx_host::AudioSink {
    audioSink *sink;
    explicit SinkOfGui(audioSink *s) : sink(s) {}
    int32_t putSamples(std::complex<float> *frames, int32_t n) override { return sink->putSamples(frames, n); }
};
class Core : public fmx_host::FmProcessor {
public:
    using fmx_host::FmProcessor::FmProcessor;
protected:
    void idle() override { QThread::msleep(1); }                          // fm-processor.cpp:389
};
}  // namespace

struct fmProcessor::Impl {
    DeviceOfGui dev;
    SinkOfGui sink;
    Core core;
    RingBuffer<std::complex<float>> *hfBuffer, *lfBuffer, *iqBuffer;
    // LF scope (fm-processor.cpp:566-627, 650-660)
    std::vector<std::complex<float>> spectrumBuffer_lf;
    std::vector<float> tapbuf;
    std::vector<std::complex<float>> dumped;
    int32_t lfCount = 0;
    // RDS (fm-processor.cpp:555-563 and the signals of the RDS objects)
    int iqCounter = 0;
    fmx_rds_info last{};
    bool haveLast = false;
    Impl(deviceHandler *d, audioSink *s, int32_t inputRate, int32_t fmRate, int32_t workingRate, int32_t audioRate, int gpu,
         RingBuffer<std::complex<float>> *hf, RingBuffer<std::complex<float>> *lf, RingBuffer<std::complex<float>> *iq)
        : dev(d), sink(s), core(&dev, &sink, inputRate, fmRate, workingRate, audioRate, gpu), hfBuffer(hf), lfBuffer(lf), iqBuffer(iq) {}
};

fmProcessor::fmProcessor(deviceHandler *theDevice, RadioInterface *RI, audioSink *mySink, fm_Demodulator *, int32_t inputRate,
                         int32_t fmRate_, int32_t workingRate, int32_t audioRate, int32_t, int spectrumSize_, int32_t repeatRate_,
                         int ptyLocale, RingBuffer<std::complex<float>> *hfBuffer, RingBuffer<std::complex<float>> *lfBuffer,
                         RingBuffer<std::complex<float>> *iqBuffer, int16_t, int gpu)
    : d(new Impl(theDevice, mySink, inputRate, fmRate_, workingRate, audioRate, gpu, hfBuffer, lfBuffer, iqBuffer)), fmRate(fmRate_),
      repeatRate(repeatRate_), spectrumSize(spectrumSize_) {
    d->core.set_ptyLocale(ptyLocale);
    qRegisterMetaType<const fmx_qt::fmProcessor::SMetaData *>("const fmx_qt::fmProcessor::SMetaData*");
    QObject *gui = RI;
    if (gui) {      // by name, as the reference's constructors do (a slot the GUI object lacks only prints Qt's warning, as there)
        connect(this, SIGNAL(setSquelchIsActive(bool)), gui, SLOT(setSquelchIsActive(bool)));
        connect(this, SIGNAL(hfBufferLoaded()), gui, SLOT(hfBufferLoaded()));
        connect(this, SIGNAL(lfBufferLoaded(bool, bool, int)), gui, SLOT(lfBufferLoaded(bool, bool, int)));
        connect(this, SIGNAL(iqBufferLoaded()), gui, SLOT(iqBufferLoaded()));
        connect(this, SIGNAL(showPeakLevel(float, float)), gui, SLOT(showPeakLevel(float, float)));
        connect(this, SIGNAL(showMetaData(const fmx_qt::fmProcessor::SMetaData *)), gui, SLOT(showMetaData(const fmx_qt::fmProcessor::SMetaData *)));
        connect(this, SIGNAL(scanresult()), gui, SLOT(scanresult()));
        connect(this, SIGNAL(setCRCErrors(int)), gui, SLOT(setCRCErrors(int)));
        connect(this, SIGNAL(setSyncErrors(int)), gui, SLOT(setSyncErrors(int)));
        connect(this, SIGNAL(setGroup(int)), gui, SLOT(setGroup(int)));
        connect(this, SIGNAL(setPTYCode(int, const QString &)), gui, SLOT(setPTYCode(int, const QString &)));
        connect(this, SIGNAL(setPiCode(int)), gui, SLOT(setPiCode(int)));
        connect(this, SIGNAL(setStationLabel(const QString &)), gui, SLOT(setStationLabel(const QString &)));
        connect(this, SIGNAL(setRadioText(const QString &)), gui, SLOT(setRadioText(const QString &)));
        connect(this, SIGNAL(clearRadioText()), gui, SLOT(clearRadioText()));
        connect(this, SIGNAL(setAFDisplay(int, int)), gui, SLOT(setAFDisplay(int, int)));
        connect(this, SIGNAL(setMusicSpeechFlag(int)), gui, SLOT(setMusicSpeechFlag(int)));
        connect(this, SIGNAL(clearMusicSpeechFlag()), gui, SLOT(clearMusicSpeechFlag()));
        connect(this, SIGNAL(setRDSisSynchronized(bool)), gui, SLOT(setRDSisSynchronized(bool)));
        connect(this, SIGNAL(setbitErrorRate(double)), gui, SLOT(setbitErrorRate(double)));
    }
}

fmProcessor::~fmProcessor() { stop(); }

void fmProcessor::stop() {                             // fm-processor.cpp:204-211
    if (running.load()) {
        running.store(false);
        while (!isFinished()) usleep(100);
    }
}

bool fmProcessor::ok() const { return d->core.ok(); }
std::string fmProcessor::lastError() const { return d->core.lastError(); }

void fmProcessor::setfmMode(FM_Mode m) { d->core.setfmMode(m); }
void fmProcessor::setFMdecoder(const QString &name) { d->core.setFMdecoder(name.toStdString()); }
void fmProcessor::setSoundMode(uint8_t selector) { d->core.setSoundMode(selector); }
void fmProcessor::setStereoPanorama(int16_t pan) { d->core.setStereoPanorama(pan); }
void fmProcessor::setSoundBalance(int16_t balance) { d->core.setSoundBalance(balance); }
void fmProcessor::setDeemphasis(int16_t us) { d->core.setDeemphasis(us); }
void fmProcessor::setVolume(float gainDb) { d->core.setVolume(gainDb); }
void fmProcessor::setlfcutoff(int32_t hz) { d->core.setlfcutoff(hz); }
void fmProcessor::setBandwidth(const QString &f) { d->core.setBandwidth(f.toStdString()); }
void fmProcessor::setAttenuation(float l, float r) { d->core.setAttenuation(l, r); }
void fmProcessor::setfmRdsSelector(rdsDecoder::ERdsMode mode) { rdsMode.store((int)mode); d->core.setfmRdsSelector((int)mode); }
void fmProcessor::triggerFrequencyChange() { d->core.triggerFrequencyChange(); }
void fmProcessor::restartPssAnalyzer() { d->core.restartPssAnalyzer(); }
void fmProcessor::resetRds() { d->core.resetRds(); }
void fmProcessor::set_localOscillator(int32_t lo) { d->core.set_localOscillator(lo); }
void fmProcessor::set_squelchMode(ESqMode m) { d->core.set_squelchMode((int)m); }
void fmProcessor::set_squelchValue(int16_t v) { d->core.set_squelchValue(v); }
void fmProcessor::setAutoMonoMode(bool b) { d->core.setAutoMonoMode(b); }
void fmProcessor::setPSSMode(bool b) { d->core.setPSSMode(b); }
void fmProcessor::setDCRemove(bool b) { dcRemove.store(b); d->core.setDCRemove(b); }
void fmProcessor::setTestTone(bool b) { d->core.setTestTone(b); }
void fmProcessor::setDispDelay(int steps) { d->core.setDispDelay(steps); }
void fmProcessor::set_ptyLocale(int l) { d->core.set_ptyLocale(l); }
bool fmProcessor::isPilotLocked(float &oLockStrength) { return d->core.isPilotLocked(oLockStrength); }
float fmProcessor::get_demodDcComponent() { return d->core.get_demodDcComponent(); }

// ---- LF scope: one entry per fm sample (per 24 kS/s RDS sample for the two RDS views) into spectrumBuffer_lf; every
//      fmRate / repeatRate + 1 fm samples the first spectrumSize entries go to the LF ring (fm-processor.cpp:566-627, 650-660, 906-912)
void fmProcessor::feed_lf_scope() {
    Impl &I = *d;
    fmx_handle h = I.core.handle();
    const int64_t nfm = fmx_last_fm_samples(h);
    if (nfm <= 0) return;
    const ELfPlot type = (ELfPlot)lfPlot.load();
    const bool rdsView = (type == ELfPlot::RDS_INPUT || type == ELfPlot::RDS_DEMOD);
    std::vector<std::complex<float>> fresh;               // this block's entries, in order
    if (type == ELfPlot::OFF) fresh.assign((size_t)nfm, std::complex<float>(0, 0));
    else if (!rdsView) {
        int tap = FMX_TAP_PRE_RESAMPLER, per = 2;
        if (type == ELfPlot::IF_FILTERED) tap = FMX_TAP_FM_IQ;
        else if (type == ELfPlot::DEMODULATOR) { tap = FMX_TAP_DEMOD; per = 1; }
        else if (type == ELfPlot::AF_SUM || type == ELfPlot::AF_DIFF) tap = FMX_TAP_LR_RAW;
        I.tapbuf.resize((size_t)(nfm * per));
        if (fmx_get_tap(h, 0, tap, I.tapbuf.data(), nfm) != FMX_OK) return;
        fresh.resize((size_t)nfm);
        for (int64_t k = 0; k < nfm; k++) {
            const float a = I.tapbuf[(size_t)(per * k)], b = per == 2 ? I.tapbuf[(size_t)(2 * k + 1)] : 0.f;
            switch (type) {
            case ELfPlot::IF_FILTERED: fresh[(size_t)k] = std::complex<float>(a, b); break;              // v
            case ELfPlot::DEMODULATOR: fresh[(size_t)k] = std::complex<float>(a, 0); break;              // demod
            case ELfPlot::AF_SUM: fresh[(size_t)k] = std::complex<float>(a, 0); break;                   // sumLR
            case ELfPlot::AF_DIFF: fresh[(size_t)k] = std::complex<float>(b, 0); break;                  // diffLR
            case ELfPlot::AF_MONO_FILTERED: fresh[(size_t)k] = std::complex<float>(a + b, 0); break;     // audio (this build: in front of the audio low-pass)
            case ELfPlot::AF_LEFT_FILTERED: fresh[(size_t)k] = std::complex<float>(a, 0); break;
            default: fresh[(size_t)k] = std::complex<float>(b, 0); break;                                // AF_RIGHT_FILTERED
            }
        }
    }
    // RDS views: an entry per rdsDecimator output (RDS on) or a zero per fm sample (RDS off), :566-589
    std::vector<std::complex<float>> rdsv;
    int64_t nrds = 0;
    const bool rdsOn = rdsMode.load() != 0;          // read ONCE per block: the GUI thread may switch RDS off while the loop below runs
    if (rdsView && rdsOn) {
        nrds = fmx_last_rds_samples(h);
        if (nrds > 0) {
            I.tapbuf.resize((size_t)(2 * nrds));
            if (fmx_get_tap(h, 0, FMX_TAP_RDS_IQ, I.tapbuf.data(), nrds) != FMX_OK) nrds = 0;
        }
        rdsv.resize((size_t)nrds);
        // RDS_INPUT: 20 rdsSample.  RDS_DEMOD shows the decoder's symbol (`magCplx`, which only changes once per bit): the same
        // decimator output stands in between the decisions; the decided symbols themselves go to the IQ ring (feed_rds)
        for (int64_t k = 0; k < nrds; k++) rdsv[(size_t)k] = (type == ELfPlot::RDS_INPUT ? 20.0f : 1.0f) * std::complex<float>(I.tapbuf[(size_t)(2 * k)], I.tapbuf[(size_t)(2 * k + 1)]);
    } else if (rdsView) fresh.assign((size_t)nfm, std::complex<float>(0, 0));
    // the reference's counters, sample by sample
    int64_t rdsPushed = 0;
    for (int64_t k = 0; k < nfm; k++) {
        if (rdsView && rdsOn) {
            // the decimator delivers one output per eight fm samples: spread this block's outputs evenly over it
            const int64_t due = ((k + 1) * nrds) / nfm;
            while (rdsPushed < due) I.spectrumBuffer_lf.push_back(rdsv[(size_t)rdsPushed++]);
        } else I.spectrumBuffer_lf.push_back(fresh[(size_t)k]);
        if (++I.lfCount > fmRate / repeatRate) {
            if (I.spectrumBuffer_lf.size() >= (size_t)spectrumSize) {
                if (I.lfBuffer) I.lfBuffer->putDataIntoBuffer(I.spectrumBuffer_lf.data(), spectrumSize);       // processLfSpectrum :906-912
                emit lfBufferLoaded(showFullSpectrum.load(), lfBuffer_newFlag.load(), zoomFactor.load());
                lfBuffer_newFlag.store(false);
                I.spectrumBuffer_lf.resize(0);
            }
            I.lfCount = 0;
        }
    }
}

// ---- RDS: the decided symbols into the IQ ring (fm-processor.cpp:555-563), the text / status signals of rdsDecoder,
//      rdsGroupDecoder and rdsBlockSynchronizer from the differences of the library's picture
void fmProcessor::feed_rds() {
    Impl &I = *d;
    fmx_handle h = I.core.handle();
    if (rdsMode.load() == 0) return;
    float sym[2 * 64]; int32_t n = 0;
    while (fmx_rds_symbols(h, 0, sym, 64, &n) == FMX_OK && n > 0) {
        for (int32_t k = 0; k < n; k++) {
            const std::complex<float> m(sym[2 * k], sym[2 * k + 1]);
            if (I.iqBuffer) I.iqBuffer->putDataIntoBuffer(&m, 1);
            if (++I.iqCounter > 100) { emit iqBufferLoaded(); I.iqCounter = 0; }
        }
        if (n < 64) break;
    }
    fmx_rds_info now{};
    if (fmx_rds_decode(h, 0, &now) != FMX_OK) return;
    const fmx_rds_info &was = I.last;
    const bool first = !I.haveLast;
    if (first || now.synchronized != was.synchronized) emit setRDSisSynchronized(now.synchronized != 0);          // rds-blocksynchronizer.cpp
    if (first || now.bit_error_rate != was.bit_error_rate) emit setbitErrorRate((double)now.bit_error_rate);
    if (first || now.crc_errors != was.crc_errors) emit setCRCErrors(now.crc_errors);                              // rds-decoder.cpp:116-121
    if (first || now.sync_errors != was.sync_errors) emit setSyncErrors(now.sync_errors);
    if (now.groups_decoded != (first ? 0 : was.groups_decoded) && now.last_group_type >= 0) emit setGroup(now.last_group_type);
    if (now.pi_code != (first ? 0 : was.pi_code)) emit setPiCode(now.pi_code);                                     // rds-groupdecoder.cpp:100-125
    if (now.pty_code >= 0 && (first || now.pty_code != was.pty_code)) emit setPTYCode(now.pty_code, pty_name(now.pty_code, I.core.get_ptyLocale()));
    if (first || std::memcmp(now.station_label, was.station_label, sizeof(now.station_label)) != 0)
        if (now.station_label[0]) emit setStationLabel(QString(now.station_label));            // QString (stationLabel) rds-groupdecoder.cpp:185-186
    if (first || now.radio_text_ucs2_len != was.radio_text_ucs2_len || std::memcmp(now.radio_text_ucs2, was.radio_text_ucs2, sizeof(now.radio_text_ucs2)) != 0) {
        if (now.radio_text_ucs2_len > 0) emit setRadioText(radio_text_qstring(now));               // prepareText -> setRadioText (outString. trimmed ())
        else if (!first) emit clearRadioText();
    }
    if ((now.af1_khz || now.af2_khz) && (first || now.af1_khz != was.af1_khz || now.af2_khz != was.af2_khz)) emit setAFDisplay(now.af1_khz, now.af2_khz);
    if (first || now.music_speech != was.music_speech) {
        if (now.music_speech >= 0) emit setMusicSpeechFlag(now.music_speech);
        else if (!first) emit clearMusicSpeechFlag();
    }
    I.last = now; I.haveLast = true;
}

void fmProcessor::run() {
    Impl &I = *d;
    running.store(true);
    bool lastSquelch = false, first = true;
    while (running.load()) {
        // the dump flag is latched once per block, with the RfDC value in front of that block (what the reference's `dumping` test at :448
        // sees); nothing is fetched on iterations that find no block
        fmx_meta dcBefore{};
        sf_private_tag *dumpNow = nullptr;
        const bool got = I.core.run_block([&] {
            dumpNow = dumpFile.load();
            if (dumpNow) (void)fmx_get_meta(I.core.handle(), 0, &dcBefore);
        });
        if (!got) { QThread::msleep(1); continue; }                       // fewer than 16384 samples waiting (:388-391)
        const int32_t amount = I.core.lastAmount();
        if (I.hfBuffer) I.hfBuffer->putDataIntoBuffer(I.core.lastBlock(), amount);          // :420
        emit hfBufferLoaded();                                                              // :421
        if (sf_private_tag *f = dumpNow) if (dumpWriter) {                                  // :448-455: the block behind the RF DC removal (:423-446)
            std::complex<float> dc(dcBefore.live_rf_dc_re, dcBefore.live_rf_dc_im);
            const float alpha = 1.0f / (float)I.dev.getRate(), lim = 0.01f;
            I.dumped.resize((size_t)amount);
            for (int32_t i = 0; i < amount; i++) {
                std::complex<float> x = I.core.lastBlock()[i];
                if (dcRemove.load()) {
                    dc = (x - dc) * alpha + dc;
                    x -= std::complex<float>(std::fmin(std::fmax(dc.real(), -lim), lim), std::fmin(std::fmax(dc.imag(), -lim), lim));
                }
                I.dumped[(size_t)i] = x;
            }
            dumpWriter(f, reinterpret_cast<const float *>(I.dumped.data()), amount);
        }
        feed_rds();
        feed_lf_scope();
        I.core.poll_peaks([this](float l, float r) { emit showPeakLevel(l, r); });          // :645, 772-798
        if (I.core.poll_meta(metaData)) {                                                   // every fmRate / 2 samples (:662-684)
            emit showMetaData(&metaData);
            const bool sq = I.core.getSquelchState();
            if (first || sq != lastSquelch) { emit setSquelchIsActive(sq); lastSquelch = sq; first = false; }   // squelchClass.cpp:74-77
            squelchState.store(sq);
        }
    }
}

// ---- RDS text: the library does the reference's byte work (fmx_rds_pty_name = pty_table [pty][locale] ebu-codetables.c:4-37;
// radio_text_ucs2 = prepareText + mapEBUtoUnicode, rds-groupdecoder.cpp:298-315); here only the QString wrapping -------------------
QString radio_text_qstring(const fmx_rds_info &info) {
    return QString::fromUtf16(reinterpret_cast<const char16_t *>(info.radio_text_ucs2), info.radio_text_ucs2_len);
}
QString pty_name(int pty, int ptyLocale) {
    const char *n = fmx_rds_pty_name(pty, ptyLocale);
    return n ? QString(n) : QString();             // (QString (const char *) = fromUtf8, what setPTYCode (int, const QString &) receives)
}

}  // namespace fmx_qt
