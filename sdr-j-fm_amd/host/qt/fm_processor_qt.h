// fm_processor_qt.h -- the Qt side of the drop-in: a QThread with the reference fmProcessor's public surface
// (includes/fm/fm-processor.h:79-294) whose run() is the reference's loop (src/fm/fm-processor.cpp:373-687) with the DSP
// replaced by calls through the C ABI of include/fmx.h.  Same setter names and argument meaning, same signals
// (showMetaData, showPeakLevel, hfBufferLoaded, lfBufferLoaded, iqBufferLoaded, setSquelchIsActive), connected to the
// GUI object by name exactly as fm-processor.cpp:179-192 does, so RadioInterface's slots receive what they received before.
//
// What stays with the maintainer when this replaces src/fm/fm-processor.cpp in the reference tree: derive the two
// interfaces below from the reference's own deviceHandler (devices/device-handler.h:60-85) and audioSink
// (includes/output/audiosink.h:36-76) -- the calls used are exactly theirs -- and hand the three RingBuffers of the scopes
// to setScopeBuffers().  Nothing here includes a reference header: the class compiles (moc + g++) against QtCore alone,
// which is what tests/test_qt_adapter.py does with the image's Qt 5.9.
#pragma once
#include <QThread>
#include <QObject>
#include <atomic>
#include <complex>
#include <cstdint>
#include <string>
#include <vector>

#include "../fm_processor_adapter.h"

namespace fmx_qt {

using fmx_host::DeviceHandler;     // Samples() / getSamples(): devices/device-handler.h:71-74
using fmx_host::AudioSink;         // putSamples(): includes/output/audiosink.h:45

class fmProcessor : public QThread {
    Q_OBJECT
public:
    typedef fmx_host::FmProcessor::FM_Mode FM_Mode;
    typedef fmx_host::FmProcessor::SMetaData SMetaData;
    enum class ELfPlot { OFF, IF_FILTERED, DEMODULATOR, AF_SUM, AF_DIFF, AF_MONO_FILTERED, AF_LEFT_FILTERED, AF_RIGHT_FILTERED, RDS_INPUT, RDS_DEMOD };   // fm-processor.h:84-87
    enum class ESqMode { OFF, NSQ, LSQ };                                  // fm-processor.h

    // fm-processor.cpp:48-63: (device, GUI object, sink, ..rates..).  The demodulator object, the scope ring buffers and
    // the scan threshold of the reference constructor have no DSP meaning here; the GUI keeps them.
    fmProcessor(DeviceHandler *theDevice, QObject *RI, AudioSink *mySink, int32_t inputRate = 2304000, int32_t fmRate = 192000,
                int32_t workingRate = 48000, int32_t audioRate = 48000, int32_t displaySize = 1024, int32_t spectrumSize = 2048,
                int32_t repeatRate = 10, int gpu = 0);
    ~fmProcessor() override;

    void stop();                                                           // fm-processor.cpp:204-211
    bool ok() const { return core.ok(); }
    std::string lastError() const { return core.lastError(); }

    // ---- the reference's setters, called from the GUI thread (fm-processor.h:104-156)
    void setfmMode(FM_Mode m) { core.setfmMode(m); }
    void setFMdecoder(const QString &name) { core.setFMdecoder(name.toStdString()); }
    void setSoundMode(uint8_t selector) { core.setSoundMode(selector); }
    void setStereoPanorama(int16_t pan) { core.setStereoPanorama(pan); }
    void setSoundBalance(int16_t balance) { core.setSoundBalance(balance); }
    void setDeemphasis(float us) { core.setDeemphasis(us); }
    void setVolume(float gainDb) { core.setVolume(gainDb); }
    void setlfcutoff(int32_t hz) { core.setlfcutoff(hz); }
    void setBandwidth(const QString &f) { core.setBandwidth(f.toStdString()); }
    void setAttenuation(float l, float r) { core.setAttenuation(l, r); }
    void setfmRdsSelector(int mode) { core.setfmRdsSelector(mode); }
    void triggerFrequencyChange() { core.triggerFrequencyChange(); }
    void restartPssAnalyzer() { core.restartPssAnalyzer(); }
    void resetRds() { core.resetRds(); }
    void set_localOscillator(int32_t lo) { core.set_localOscillator(lo); }
    void set_squelchMode(ESqMode m) { core.set_squelchMode((int)m); }
    void set_squelchValue(int16_t v) { core.set_squelchValue(v); }
    void setAutoMonoMode(bool b) { core.setAutoMonoMode(b); }
    void setPSSMode(bool b) { core.setPSSMode(b); }
    void setDCRemove(bool b) { core.setDCRemove(b); }
    void setTestTone(bool b) { core.setTestTone(b); }
    void setDispDelay(int steps) { core.setDispDelay(steps); }
    void setlfPlotType(ELfPlot t) { lfPlot.store((int)t); }
    void setlfPlotZoomFactor(int32_t z) { zoomFactor.store(z); }
    bool isPilotLocked(float &oLockStrength) { return core.isPilotLocked(oLockStrength); }
    float get_demodDcComponent() { return core.get_demodDcComponent(); }
    bool getSquelchState() { return squelchState.load(); }

signals:                                                                   // fm-processor.h:286-293, squelchClass.h
    void hfBufferLoaded();
    void lfBufferLoaded(bool, bool, int);
    void iqBufferLoaded();
    void showMetaData(const fmx_qt::fmProcessor::SMetaData *);
    void showPeakLevel(const float, const float);
    void setSquelchIsActive(bool);

protected:
    void run() override;                                                   // fm-processor.cpp:373-687

private:
    class Core : public fmx_host::FmProcessor {
    public:
        using fmx_host::FmProcessor::FmProcessor;
        fmProcessor *owner = nullptr;
    protected:
        void idle() override { QThread::msleep(1); }                      // fm-processor.cpp:389
    };
    Core core;
    std::atomic<bool> running{false};
    std::atomic<int> lfPlot{0}, zoomFactor{1};
    std::atomic<bool> squelchState{false};
    int32_t fmRate, repeatRate, spectrumSize;
    SMetaData metaData{};
};

}  // namespace fmx_qt

Q_DECLARE_METATYPE(const fmx_qt::fmProcessor::SMetaData *)
