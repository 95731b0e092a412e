// fm_processor_adapter.h -- host-side C++ mirror of the reference's fmProcessor for the FM hot path.
//
// The reference wires  deviceHandler -> fmProcessor -> audioSink  (SURVEY 1); `fmProcessor` is a
// QThread whose run() loop (src/fm/fm-processor.cpp:373-687) pulls 16384-sample blocks and pushes
// PCM frames.  This header provides the same class shape WITHOUT Qt on top of the C ABI of
// include/fmx.h, so that the GUI's call sites (radio.cpp:915-948, the handle_* slots) compile
// against it unchanged in spirit: same method names, same argument meaning, same error behaviour
// (setters never throw; unsupported settings are reported through lastError()).
//
// A Qt build derives this class from QThread and emits the signals from the values returned by
// poll_meta(); see INTEGRATION.md.  Header-only; links against libfmx.so only.
#pragma once
#include <atomic>
#include <algorithm>
#include <complex>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/fmx.h"

namespace fmx_host {

// devices/device-handler.h:60-85 -- the two calls the processing loop uses
struct DeviceHandler {
    virtual ~DeviceHandler() {}
    virtual int32_t Samples() = 0;                                        // complex samples available
    virtual int32_t getSamples(std::complex<float> *dst, int32_t n) = 0;  // copies n interleaved (I,Q) pairs
    virtual int32_t getRate() { return 2304000; }
};
// includes/output/audiosink.h:36-76 -- real = left, imag = right, float32 at audioRate
struct AudioSink {
    virtual ~AudioSink() {}
    virtual int32_t putSamples(std::complex<float> *frames, int32_t n) = 0;
};

class FmProcessor {
public:
    enum class FM_Mode { Stereo, StereoPano, Mono };                       // fm-processor.h:83
    enum Channels { S_STEREO, S_STEREO_SWAPPED, S_LEFT, S_RIGHT, S_LEFTplusRIGHT, S_LEFTminusRIGHT,
                    S_LEFTminusRIGHT_Test };                               // fm-processor.h:88-90
    struct SMetaData {                                                     // fm-processor.h:91-101
        enum class EPssState { OFF, ANALYZING, ESTABLISHED };
        float DcValRf, DcValIf, PssPhaseShiftDegree, PssPhaseChange;
        EPssState PssState;
        float PilotPllLockStrength;
        bool PilotPllLocked;
    };

    // fm-processor.cpp:48-63 (the scope ring buffers, RadioInterface* and fm_Demodulator* of the
    // reference constructor stay on the GUI side; the decoder is selected with setFMdecoder)
    FmProcessor(DeviceHandler *theDevice, AudioSink *mySink, int32_t inputRate = 2304000, int32_t fmRate = 192000,
                int32_t workingRate = 48000, int32_t audioRate = 48000, int device = 0)
        : myRig(theDevice), theSink(mySink) {
        fmx_config c{};
        c.struct_size = (int32_t)sizeof(c); c.device = device; c.channels = 1; c.streams = 0; c.stream_of_channel = nullptr;
        c.inputRate = inputRate; c.fmRate = fmRate; c.workingRate = workingRate; c.audioRate = audioRate;
        c.max_block = bufferSize;
        if (fmx_abi_version() != FMX_ABI_VERSION) { err = "libfmx has another ABI version than this adapter was built for"; h = nullptr; return; }
        check(fmx_create(&c, &h));
        inBuf.resize(bufferSize);
        // one PCM frame per 4 fm samples; an input rate the reference does not decimate (192 kS/s devices) gives bufferSize fm samples per
        // block; the second converter makes audioRate / workingRate frames of every 48 kHz frame
        outBuf.resize((size_t)((int64_t)(bufferSize / 4 + 64) * std::max(audioRate, workingRate) / workingRate + 8));
    }
    ~FmProcessor() { stop(); if (h) fmx_destroy(h); }
    FmProcessor(const FmProcessor &) = delete;
    FmProcessor &operator=(const FmProcessor &) = delete;

    bool ok() const { return h != nullptr; }
    const std::string &lastError() const { return err; }

    // ---- the reference's setters, fm-processor.h:104-156 ----
    void setfmMode(FM_Mode m) { set(FMX_P_FM_MODE, (double)(int)m); }
    void setFMdecoder(const std::string &name) {                          // fm-demodulator.cpp:93-103
        static const char *names[] = { "AM", "FM PLL Decoder", "FM Mixed Demod", "FM Complex Baseband Delay",
                                       "FM Real Baseband Delay", "FM Difference Based" };
        int code = 2;                                                      // unknown name -> `default:` = PLL
        for (int i = 0; i < 6; i++) if (name == names[i]) code = i + 1;
        set(FMX_P_FM_DECODER, code);
    }
    void setSoundMode(uint8_t selector) { set(FMX_P_SOUND_MODE, selector); }
    void setStereoPanorama(int16_t pan) { set(FMX_P_STEREO_PANORAMA, pan); }
    void setSoundBalance(int16_t balance) { set(FMX_P_SOUND_BALANCE, balance); }
    void setDeemphasis(int16_t us) { set(FMX_P_DEEMPHASIS, us); }
    void setVolume(float gainDb) { set(FMX_P_VOLUME_DB, gainDb); }
    void setlfcutoff(int32_t hz) { set(FMX_P_LF_CUTOFF, hz); }
    void setBandwidth(const std::string &s) {                              // "165kHz" | "Off" (fm-processor.cpp:232-239)
        set(FMX_P_BANDWIDTH, s == "Off" ? 0.0 : 1000.0 * std::strtol(s.c_str(), nullptr, 10));
    }
    void setAttenuation(float l, float r) { set(FMX_P_ATTENUATION_L, l); set(FMX_P_ATTENUATION_R, r); }
    void setfmRdsSelector(int mode) { set(FMX_P_RDS_MODE, mode); }
    void triggerFrequencyChange() { set(FMX_A_TRIGGER_FREQUENCY_CHANGE, 0); }
    void restartPssAnalyzer() { set(FMX_A_RESTART_PSS, 0); }
    void resetRds() { set(FMX_A_RESET_RDS, 0); }
    void set_localOscillator(int32_t lo) { set(FMX_P_LOCAL_OSCILLATOR, lo); }
    void set_squelchMode(int m) { set(FMX_P_SQUELCH_MODE, m); }
    void setAutoMonoMode(bool b) { set(FMX_P_AUTO_MONO, b); }
    void setPSSMode(bool b) { set(FMX_P_PSS, b); }
    void setDCRemove(bool b) { set(FMX_P_DC_REMOVE, b); }
    void setTestTone(bool b) { set(FMX_P_TEST_TONE, b); }
    void setDispDelay(int steps) { set(FMX_P_DISP_DELAY, steps); }       // fm-processor.cpp:935-937
    void set_squelchValue(int v) { set(FMX_P_SQUELCH_VALUE, v); }        // :213-215

    bool isPilotLocked(float &oLockStrength) {                             // fm-processor.cpp:870-880
        fmx_meta m{};
        if (fmx_get_meta(h, 0, &m) != FMX_OK) { oLockStrength = 0; return false; }
        oLockStrength = m.live_lock_strength;
        return m.live_pilot_locked != 0;
    }
    float get_demodDcComponent() { fmx_meta m{}; return fmx_get_meta(h, 0, &m) == FMX_OK ? m.live_dc_if : 0.0f; }
    bool getSquelchState() { fmx_meta m{}; return fmx_get_meta(h, 0, &m) == FMX_OK && m.squelch_active != 0; }      // fm-processor.cpp:217-219

    // ---- the processing loop ----
    // One iteration of the while loop of fmProcessor::run() (fm-processor.cpp:387-686).  Returns false
    // when the device holds fewer than bufferSize samples (the reference sleeps 1 ms and retries).
    // `before_block`, when given, runs once a block is known to be waiting and before it is taken: the place to read state that belongs in
    // front of the block (the Qt adapter latches its dump flag and the RfDC value there, not on iterations that find no block)
    template <class F> bool run_block(F before_block) {
        if (!h || myRig->Samples() < bufferSize) return false;
        before_block();
        return take_block();
    }
    bool run_block() { return run_block([] {}); }
    bool take_block() {
        const int32_t amount = myRig->getSamples(inBuf.data(), bufferSize);
        lastN = amount;
        int64_t frames = 0;
        if (!check(fmx_process_host(h, reinterpret_cast<const float *>(inBuf.data()), amount, amount,
                                    reinterpret_cast<float *>(outBuf.data()), (int64_t)outBuf.size(), &frames)))
            return false;
        if (frames > 0 && theSink) theSink->putSamples(outBuf.data(), (int32_t)frames);
        fmCount += fmx_last_fm_samples(h);                                    // (amount / the reference's decimation at this input rate)
        return true;
    }
    // QThread::run() equivalent; stop() as fm-processor.cpp:204-211
    void run() { running.store(true); while (running.load()) { if (!run_block()) idle(); } }
    void stop() { running.store(false); }

    // what the reference emits as showMetaData every fmRate/2 samples (fm-processor.cpp:662-684)
    bool poll_meta(SMetaData &out) {
        if (fmCount - lastMeta <= 96000) return false;
        lastMeta = fmCount;
        fmx_meta m{};
        if (fmx_get_meta(h, 0, &m) != FMX_OK) return false;
        out.DcValRf = m.DcValRf; out.DcValIf = m.DcValIf; out.PssPhaseShiftDegree = m.PssPhaseShiftDegree;
        out.PssPhaseChange = m.PssPhaseChange; out.PssState = (SMetaData::EPssState)m.PssState;
        out.PilotPllLockStrength = m.PilotPllLockStrength; out.PilotPllLocked = m.PilotPllLocked != 0;
        return true;
    }

    // what the reference emits as showPeakLevel (leftDb, rightDb) every 961 PCM frames (fm-processor.cpp:772-798):
    // calls `emit_fn(leftDb, rightDb)` once per window that closed since the last poll, oldest first
    template <class F> int poll_peaks(F emit_fn) {
        float lr[2 * 64]; int32_t n = 0;
        if (fmx_get_peaks(h, 0, lr, 64, &n) != FMX_OK) return 0;
        for (int32_t k = 0; k < n; k++) emit_fn(lr[2 * k], lr[2 * k + 1]);
        return n;
    }

    static constexpr int32_t bufferSize = 2 * 8192;                        // fm-processor.cpp:374

    // what the GUI side-band feeds of run() need (fm-processor.cpp:420-421, 555-563, 597-660): the raw block just processed
    // and the library handle for fmx_get_tap / fmx_rds_*
    const std::complex<float> *lastBlock() const { return inBuf.data(); }
    int32_t lastAmount() const { return lastN; }
    fmx_handle handle() const { return h; }
    void set_ptyLocale(int l) { ptyLocale = l; }                            // fm-processor.cpp:939-941 (used by the adapter's PTY names)
    int get_ptyLocale() const { return ptyLocale; }

protected:
    virtual void idle() {}                                                  // the Qt build calls msleep(1) here

private:
    bool check(int rc) { if (rc != FMX_OK) { err = fmx_last_error(); return false; } return true; }
    void set(int id, double v) { if (h) check(fmx_set_param(h, 0, id, v)); }
    fmx_handle h = nullptr;
    DeviceHandler *myRig;
    AudioSink *theSink;
    std::vector<std::complex<float>> inBuf, outBuf;
    std::atomic<bool> running{false};
    std::string err;
    int64_t fmCount = 0, lastMeta = 0;
    int32_t lastN = 0;
    int ptyLocale = 0;
};

}  // namespace fmx_host
