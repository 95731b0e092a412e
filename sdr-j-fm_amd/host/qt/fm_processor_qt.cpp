// fm_processor_qt.cpp -- see fm_processor_qt.h.  Build: moc fm_processor_qt.h -o moc_fm_processor_qt.cpp;
//   g++ -std=c++17 -fPIC -I$QT/include/qt -I$QT/include/qt/QtCore fm_processor_qt.cpp moc_fm_processor_qt.cpp -lQt5Core -lfmx
#include "fm_processor_qt.h"

namespace fmx_qt {

fmProcessor::fmProcessor(DeviceHandler *theDevice, QObject *RI, AudioSink *mySink, int32_t inputRate, int32_t fmRate_,
                         int32_t workingRate, int32_t audioRate, int32_t, int32_t spectrumSize_, int32_t repeatRate_, int gpu)
    : core(theDevice, mySink, inputRate, fmRate_, workingRate, audioRate, gpu), fmRate(fmRate_), repeatRate(repeatRate_),
      spectrumSize(spectrumSize_) {
    core.owner = this;
    qRegisterMetaType<const fmx_qt::fmProcessor::SMetaData *>("const fmx_qt::fmProcessor::SMetaData*");
    if (RI) {       // by name, as fm-processor.cpp:179-192 (a slot the GUI object lacks only prints Qt's warning, as there)
        connect(this, SIGNAL(setSquelchIsActive(bool)), RI, SLOT(setSquelchIsActive(bool)));
        connect(this, SIGNAL(hfBufferLoaded()), RI, SLOT(hfBufferLoaded()));
        connect(this, SIGNAL(lfBufferLoaded(bool, bool, int)), RI, SLOT(lfBufferLoaded(bool, bool, int)));
        connect(this, SIGNAL(iqBufferLoaded()), RI, SLOT(iqBufferLoaded()));
        connect(this, SIGNAL(showPeakLevel(float, float)), RI, SLOT(showPeakLevel(float, float)));
        connect(this, SIGNAL(showMetaData(const fmx_qt::fmProcessor::SMetaData *)), RI, SLOT(showMetaData(const fmx_qt::fmProcessor::SMetaData *)));
    }
}

fmProcessor::~fmProcessor() { stop(); }

void fmProcessor::stop() {                             // fm-processor.cpp:204-211
    if (running.load()) {
        running.store(false);
        while (!isFinished()) usleep(100);
    }
}

void fmProcessor::run() {
    running.store(true);
    int64_t fm = 0, lastLf = 0;
    bool lastSquelch = false, first = true;
    while (running.load()) {
        if (!core.run_block()) { QThread::msleep(1); continue; }          // fewer than 16384 samples waiting (:388-391)
        emit hfBufferLoaded();                                             // the raw block went to the HF scope ring (:420-421)
        core.poll_peaks([this](float l, float r) { emit showPeakLevel(l, r); });   // :645, 772-798
        fm += fmx_host::FmProcessor::bufferSize / 12;
        if (fm - lastLf > fmRate / repeatRate) {                           // LF scope (:650-660)
            emit lfBufferLoaded(false, false, zoomFactor.load());
            lastLf = fm;
        }
        if (core.poll_meta(metaData)) {                                    // every fmRate / 2 samples (:662-684)
            emit showMetaData(&metaData);
            const bool sq = core.getSquelchState();
            if (first || sq != lastSquelch) { emit setSquelchIsActive(sq); lastSquelch = sq; first = false; }   // squelchClass.cpp:74-77
            squelchState.store(sq);
        }
    }
}

}  // namespace fmx_qt
