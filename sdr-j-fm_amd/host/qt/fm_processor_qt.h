// fm_processor_qt.h -- the Qt side of the drop-in: a QThread with the reference fmProcessor's public surface
// (includes/fm/fm-processor.h:79-294) whose run() is the reference's loop (src/fm/fm-processor.cpp:373-687) with the DSP
// replaced by calls through the C ABI of include/fmx.h.
//
//   * Constructor: the reference's own argument list, in its order and with its types (fm-processor.cpp:48-63), so that
//     radio.cpp:915-930 compiles against it as it stands.  The GUI-side types (deviceHandler, audioSink, RadioInterface,
//     fm_Demodulator, RingBuffer<>) are only forward-declared here; fm_processor_qt.cpp pulls their definitions in through ONE
//     include, "fmx_qt_host.h": in the reference tree a four-line header naming device-handler.h, audiosink.h, radio.h and
//     ringbuffer.h; in this repository's test build the stand-in GUI of gui_stub/ (same names, the members used here).
//   * Setters: same names and argument meaning (fm-processor.h:104-156).
//   * Signals: everything fmProcessor and the RDS objects it owns send to the GUI, connected to the GUI object's slots BY NAME as
//     fm-processor.cpp:179-192, rds-decoder.cpp:52-55, rds-groupdecoder.cpp:44-63 and rds-blocksynchronizer.cpp:39-42 do.
//   * Side-band feeds of run(): the raw block into the HF scope ring (:420-421), the LF scope vector by lfPlotType every
//     fmRate / repeatRate samples (:566-627, 650-660), the RDS constellation points into the IQ ring with iqBufferLoaded every
//     101 symbols (:555-563), the RDS text / status signals from the differences of fmx_rds_decode's picture, showMetaData every
//     fmRate / 2 samples -- whose DcValIf is what RadioInterface's AFC loop reads (radio.cpp:1786-1809).
// `fmProcessor` in the global namespace is this class when FMX_QT_GLOBAL_NAMES is defined (the reference tree).
#pragma once
#include <QObject>
#include <QString>
#include <QThread>
#include <atomic>
#include <complex>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../fm_processor_adapter.h"

class deviceHandler;                       // devices/device-handler.h:60-85
class audioSink;                           // includes/output/audiosink.h:36-76
class RadioInterface;                      // radio.h
class fm_Demodulator;                      // includes/fm/fm-demodulator.h (the GUI's decoder object: its DSP runs on the GPU here)
template <class T> class RingBuffer;       // includes/various/ringbuffer.h
struct sf_private_tag;                     // sndfile.h: typedef struct sf_private_tag SNDFILE

namespace fmx_qt {

// the selector type of setfmRdsSelector (includes/rds/rds-decoder.h:60-65); in the reference tree this is the reference's own class
struct rdsDecoder { enum class ERdsMode { RDS_OFF, RDS_1, RDS_2, RDS_3 }; };

class fmProcessor : public QThread {
    Q_OBJECT
public:
    typedef fmx_host::FmProcessor::FM_Mode FM_Mode;
    typedef fmx_host::FmProcessor::SMetaData SMetaData;
    typedef fmx_host::FmProcessor::Channels Channels;
    enum class ELfPlot { OFF, IF_FILTERED, DEMODULATOR, AF_SUM, AF_DIFF, AF_MONO_FILTERED, AF_LEFT_FILTERED, AF_RIGHT_FILTERED, RDS_INPUT, RDS_DEMOD };   // fm-processor.h:84-87
    enum class ESqMode { OFF, NSQ, LSQ };

    fmProcessor(deviceHandler *theDevice, RadioInterface *RI, audioSink *mySink, fm_Demodulator *theDemodulator,
                int32_t inputRate, int32_t fmRate, int32_t workingRate, int32_t audioRate, int32_t displaySize, int spectrumSize,
                int32_t repeatRate, int ptyLocale, RingBuffer<std::complex<float>> *hfBuffer,
                RingBuffer<std::complex<float>> *lfBuffer, RingBuffer<std::complex<float>> *iqBuffer, int16_t thresHold,
                int gpu = 0);                                              // (gpu: the HIP device, the one argument the reference lacks)
    ~fmProcessor() override;

    void stop();                                                           // fm-processor.cpp:204-211
    bool ok() const;
    std::string lastError() const;

    // ---- the reference's setters, called from the GUI thread (fm-processor.h:104-156)
    void setfmMode(FM_Mode m);
    void setFMdecoder(const QString &name);
    void setSoundMode(uint8_t selector);
    void setStereoPanorama(int16_t pan);
    void setSoundBalance(int16_t balance);
    void setDeemphasis(int16_t us);                                        // fm-processor.h:125
    void setVolume(float gainDb);
    void setlfcutoff(int32_t hz);
    void setBandwidth(const QString &f);
    void setAttenuation(float l, float r);
    void setfmRdsSelector(rdsDecoder::ERdsMode mode);                      // fm-processor.h:134
    void triggerFrequencyChange();
    void restartPssAnalyzer();
    void resetRds();
    void set_localOscillator(int32_t lo);
    void set_squelchMode(ESqMode m);
    void set_squelchValue(int16_t v);
    void setAutoMonoMode(bool b);
    void setPSSMode(bool b);
    void setDCRemove(bool b);
    void setTestTone(bool b);
    void setDispDelay(int steps);
    void set_ptyLocale(int l);                                             // fm-processor.cpp:939-941
    void setlfPlotType(ELfPlot t) {                                        // fm-processor.cpp:244-265
        lfPlot.store((int)t);
        showFullSpectrum.store(t == ELfPlot::IF_FILTERED || t == ELfPlot::RDS_INPUT || t == ELfPlot::RDS_DEMOD);
        lfBuffer_newFlag.store(true);
    }
    void setlfPlotZoomFactor(int32_t z) { zoomFactor.store(z); lfBuffer_newFlag.store(true); }   // :267-270
    void new_lfSpectrum() { lfBuffer_newFlag.store(true); }                // fm-processor.cpp:927-929
    // input dump (fm-processor.cpp:337-349, 448-455 write every block with sf_writef_float): the image has no libsndfile, so the
    // handle is passed through to `dumpWriter` (a one-line function around sf_writef_float in the reference tree).  The block is
    // written behind the RF DC removal, as the reference writes it: the reference's own recurrence (:423-446) run over the block from
    // the library's RfDC in front of it (fmx_meta::live_rf_dc_*).
    void startDumping(sf_private_tag *f) { dumpFile.store(f); }
    void stopDumping() { dumpFile.store(nullptr); }
    static void (*dumpWriter)(sf_private_tag *f, const float *interleaved_iq, int32_t frames);
    bool isPilotLocked(float &oLockStrength);
    float get_demodDcComponent();
    bool getSquelchState() { return squelchState.load(); }

signals:
    // fm-processor.h:286-293, squelchClass.h
    void hfBufferLoaded();
    void lfBufferLoaded(bool, bool, int);
    void iqBufferLoaded();
    void showMetaData(const fmx_qt::fmProcessor::SMetaData *);
    void showPeakLevel(const float, const float);
    void setSquelchIsActive(bool);
    void scanresult();                                                     // (scan mode is out of scope: never emitted)
    // rds-decoder.h:84-85, rds-groupdecoder.h:90-101, rds-blocksynchronizer.h:102-104
    void setCRCErrors(int);
    void setSyncErrors(int);
    void setGroup(int);
    void setPTYCode(int, const QString &);
    void setMusicSpeechFlag(int);
    void clearMusicSpeechFlag();
    void setPiCode(int);
    void setStationLabel(const QString &);
    void clearRadioText();
    void setRadioText(const QString &);
    void setAFDisplay(int, int);
    void setRDSisSynchronized(bool);
    void setbitErrorRate(double);

protected:
    void run() override;                                                   // fm-processor.cpp:373-687

private:
    void feed_lf_scope();
    void feed_rds();
    struct Impl;
    std::unique_ptr<Impl> d;
    std::atomic<bool> running{false};
    std::atomic<int> lfPlot{0}, zoomFactor{1}, rdsMode{0};
    std::atomic<bool> dcRemove{true};
    std::atomic<bool> squelchState{false}, showFullSpectrum{false}, lfBuffer_newFlag{true};
    std::atomic<sf_private_tag *> dumpFile{nullptr};
    int32_t fmRate, repeatRate, spectrumSize;
    SMetaData metaData{};
};

// the radio text as the reference's setRadioText receives it (fmx_rds_info::radio_text_ucs2)
QString radio_text_qstring(const fmx_rds_info &info);
// programme type names: the reference's pty_table [pty][ptyLocale] (fmx_rds_pty_name)
QString pty_name(int pty, int ptyLocale);

}  // namespace fmx_qt

#ifdef FMX_QT_GLOBAL_NAMES
using fmProcessor = fmx_qt::fmProcessor;
#endif

Q_DECLARE_METATYPE(const fmx_qt::fmProcessor::SMetaData *)
