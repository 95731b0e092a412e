// gui_stub/fmx_qt_host.h -- TEST SCAFFOLDING, not product code: the GUI side the Qt adapter binds to, reduced to the names and
// members fm_processor_qt.cpp uses, so that the adapter compiles and runs in this repository (tests/test_qt_adapter.py) exactly
// as it would inside the reference tree.  There this file is replaced by a four-line header:
//     #include "device-handler.h"      // deviceHandler   (devices/device-handler.h:60-85)
//     #include "audiosink.h"           // audioSink       (includes/output/audiosink.h:36-76)
//     #include "ringbuffer.h"          // RingBuffer<T>   (includes/various/ringbuffer.h:127-330)
//     #include "radio.h"               // RadioInterface  (radio.h)
// Written from the call sites, not from those headers: a device is "Samples() / getSamples(buffer, n, mode)", a sink is
// "putSamples(frames, n)", a ring is "putDataIntoBuffer / getDataFromBuffer / GetRingBufferReadAvailable", the GUI object is a
// QObject with the slots the signals are connected to by name.
#pragma once
#include <QObject>
#include <QString>
#include <atomic>
#include <complex>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>

class deviceHandler {
public:
    virtual ~deviceHandler() {}
    virtual int32_t getRate() { return 2304000; }
    virtual int32_t getSamples(std::complex<float> *, int32_t, uint8_t) = 0;
    virtual int32_t Samples() = 0;
};

class audioSink {
public:
    int32_t putSamples(std::complex<float> *f, int32_t n) { pcm.insert(pcm.end(), f, f + n); return n; }
    std::vector<std::complex<float>> pcm;
};

template <class T> class RingBuffer {
public:
    explicit RingBuffer(uint32_t elementCount) : buf(elementCount) {}
    int32_t putDataIntoBuffer(const void *data, int32_t n) {
        std::lock_guard<std::mutex> lk(m);
        const T *p = static_cast<const T *>(data);
        int32_t done = 0;
        for (; done < n && count < buf.size(); done++) { buf[(rd + count) % buf.size()] = p[done]; count++; }
        total += (uint64_t)done;
        return done;
    }
    int32_t getDataFromBuffer(void *data, int32_t n) {
        std::lock_guard<std::mutex> lk(m);
        T *p = static_cast<T *>(data);
        int32_t done = 0;
        for (; done < n && count > 0; done++) { p[done] = buf[rd]; rd = (rd + 1) % buf.size(); count--; }
        return done;
    }
    uint32_t GetRingBufferReadAvailable() { std::lock_guard<std::mutex> lk(m); return (uint32_t)count; }
    uint64_t totalWritten() { std::lock_guard<std::mutex> lk(m); return total; }
private:
    std::vector<T> buf; size_t rd = 0, count = 0; uint64_t total = 0; std::mutex m;
};

#include "radio_stub.h"
