// gui_stub/radio_stub.h -- TEST SCAFFOLDING (see fmx_qt_host.h): the GUI object.  A QObject with the slots the reference's
// RadioInterface offers to fmProcessor and its RDS objects (radio.h: public slots), each recording what arrived; showMetaData
// also runs RadioInterface's AFC arithmetic on DcValIf (radio.cpp:1786-1809) to show that the read-back loop has its input.
#pragma once
#include <QObject>
#include <QString>
#include <atomic>
#include <cmath>
#include "../fm_processor_qt.h"

class RadioInterface : public QObject {
    Q_OBJECT
public:
    std::atomic<int> nMeta{0}, nPeaks{0}, nHf{0}, nLf{0}, nIq{0}, nSq{0}, nGroup{0};
    float lastL = 0, lastR = 0, lockStrength = 0, dcIf = 0;
    bool locked = false; int pssState = 0;
    int pi = 0, pty = -1, crc = -1, sync = -1, af1 = 0, af2 = 0, ms = -1, lfZoom = 0;
    bool rdsSync = false, lfFull = false, lfNew = false;
    double ber = -1;
    QString ptyName, label, text;
    // AFC (radio.cpp:1786-1809)
    float afcAlpha = 0.8f, afcCurrOffFreq = 0; int retunes = 0;
public slots:
    void hfBufferLoaded() { nHf++; }
    void lfBufferLoaded(bool full, bool isNew, int zoom) { nLf++; lfFull = full; lfNew = lfNew || isNew; lfZoom = zoom; }
    void iqBufferLoaded() { nIq++; }
    void showPeakLevel(const float l, const float r) { nPeaks++; lastL = l; lastR = r; }
    void setSquelchIsActive(bool) { nSq++; }
    void scanresult() {}
    void showMetaData(const fmx_qt::fmProcessor::SMetaData *m) {
        nMeta++; locked = m->PilotPllLocked; lockStrength = m->PilotPllLockStrength; pssState = (int)m->PssState; dcIf = m->DcValIf;
        const int32_t afcOffFreq = (int32_t)(m->DcValIf * 10000);
        afcCurrOffFreq = (1 - afcAlpha) * afcCurrOffFreq + afcAlpha * afcOffFreq;
        const float a = std::fabs(afcCurrOffFreq);
        afcAlpha = a < 10 ? 0.005f : (a < 100 ? 0.050f : 0.800f);
        if (a > 3) retunes++;
    }
    void setCRCErrors(int n) { crc = n; }
    void setSyncErrors(int n) { sync = n; }
    void setGroup(int) { nGroup++; }
    void setPTYCode(int c, const QString &s) { pty = c; ptyName = s; }
    void setMusicSpeechFlag(int f) { ms = f; }
    void clearMusicSpeechFlag() { ms = -1; }
    void setPiCode(int c) { pi = c; }
    void setStationLabel(const QString &s) { label = s; }
    void clearRadioText() { text.clear(); }
    void setRadioText(const QString &s) { text = s; }
    void setAFDisplay(int a, int b) { af1 = a; af2 = b; }
    void setRDSisSynchronized(bool b) { rdsSync = b; }
    void setbitErrorRate(double v) { ber = v; }
};
