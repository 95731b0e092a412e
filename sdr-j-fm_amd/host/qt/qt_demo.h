// qt_demo.h -- a stand-in for RadioInterface with the slots fm-processor.cpp:179-192 connects to (by name).
#pragma once
#include <QObject>
#include <atomic>
#include "fm_processor_qt.h"

class Receiver : public QObject {
    Q_OBJECT
public:
    std::atomic<int> nMeta{0}, nPeaks{0}, nHf{0}, nLf{0}, nSq{0};
    float lastL = 0, lastR = 0, lockStrength = 0; bool locked = false; int pssState = 0;
public slots:
    void showMetaData(const fmx_qt::fmProcessor::SMetaData *m) { nMeta++; locked = m->PilotPllLocked; lockStrength = m->PilotPllLockStrength; pssState = (int)m->PssState; }
    void showPeakLevel(float l, float r) { nPeaks++; lastL = l; lastR = r; }
    void hfBufferLoaded() { nHf++; }
    void lfBufferLoaded(bool, bool, int) { nLf++; }
    void iqBufferLoaded() {}
    void setSquelchIsActive(bool) { nSq++; }
};
