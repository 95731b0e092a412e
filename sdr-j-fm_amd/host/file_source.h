// file_source.h -- the file input of BASELINE configs[0] ("filereader .wav @ 2.304 MS/s") as a deviceHandler-shaped C++ source
// with the behaviour of the reference's fileHulp (devices/filereader/filehulp.cpp), real-time pacing included:
//   * rate and channel count come from the header (:61-63); 2 channels = (I, Q), 1 channel = I with Q = 0 (:127-137);
//   * samples are what libsndfile's sf_readf_float returns: PCM16 / 32768, PCM8 (unsigned) (v - 128) / 128, PCM32 / 2^31,
//     float32 as is -- parsed here from RIFF/WAVE directly (libsndfile is not in the image), streamed from the file 10 ms at a time;
//   * a reader thread (:159-202) moves 10 ms of samples (2 * rate / 100 floats) per 10 ms period into a ring of 32768 * 32
//     floats (:30), waits while the ring is full, pauses while the reader is stopped (restartReader / stopReader :83-92; it
//     starts PAUSED, :67), and sleeps until the next period's deadline: the file plays at its own sample rate;
//   * at end of file it seeks back to the start and goes on; the short read is NOT padded (:141-143); a read of nothing is a
//     period of zeros (:187-191);
//   * Samples () = complex samples in the ring (:94-98); getSamples (V, n, attenuation) blocks in 100 us naps until n are there
//     and multiplies by the attenuation (:100-119).
// `realtime = false` drops only the deadline sleep (tests, batch conversion): everything else is the same code path.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "fm_processor_adapter.h"

namespace fmx_host {

class FileSource : public DeviceHandler {
public:
    FileSource(const std::string &path, bool *success, bool realtime = true) : realtime_(realtime), ring_(kRing) {
        *success = parse(path);
        readerOK_ = *success;
        if (readerOK_) worker_ = std::thread([this] { run(); });
    }
    ~FileSource() override {
        exit_ = true;
        if (worker_.joinable()) worker_.join();
        if (file_) std::fclose(file_);
    }
    bool restartReader() { if (readerOK_) pausing_ = false; return readerOK_; }
    void stopReader() { if (readerOK_) pausing_ = true; }
    bool isWorking() const { return readerOK_; }
    int32_t getRate() override { return inputRate_; }
    int64_t samplesinFile() const { return frames_; }
    int64_t currPos() const { return currPos_; }
    int32_t Samples() override { return exit_ ? 0 : (int32_t)(avail() / 2); }
    int32_t getSamples(std::complex<float> *V, int32_t n) override { return getSamples(V, n, 1.0f); }
    int32_t getSamples(std::complex<float> *V, int32_t n, float attenuation) {
        while (!exit_ && avail() < (size_t)(2 * n)) std::this_thread::sleep_for(std::chrono::microseconds(100));
        if (exit_) return 0;
        size_t r = rd_.load(std::memory_order_relaxed);
        for (int32_t i = 0; i < n; i++) {
            V[i] = std::complex<float>(ring_[r % kRing] * attenuation, ring_[(r + 1) % kRing] * attenuation);
            r += 2;
        }
        rd_.store(r, std::memory_order_release);
        return n;
    }

private:
    static constexpr size_t kRing = 32768 * 32;
    size_t avail() const { return wr_.load(std::memory_order_acquire) - rd_.load(std::memory_order_acquire); }
    size_t space() const { return kRing - avail(); }

    // header only: the samples are read and converted 10 ms at a time by the reader thread, as sf_readf_float does
    bool parse(const std::string &path) {
        file_ = std::fopen(path.c_str(), "rb");
        if (!file_) { std::fprintf(stderr, "file %s no legitimate sound file\n", path.c_str()); return false; }
        uint8_t hd[12];
        if (std::fread(hd, 1, 12, file_) != 12 || std::memcmp(hd, "RIFF", 4) != 0 || std::memcmp(hd + 8, "WAVE", 4) != 0) return false;
        auto u16 = [](const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); };
        auto u32 = [&](const uint8_t *p) { return u16(p) | (u16(p + 2) << 16); };
        uint32_t tag = 0, ch = 0, bits = 0; bool have_fmt = false;
        for (;;) {
            uint8_t ck[8];
            if (std::fread(ck, 1, 8, file_) != 8) return false;
            const uint32_t size = u32(ck + 4);
            if (std::memcmp(ck, "fmt ", 4) == 0) {
                if (size < 16 || size > 4096) return false;                       // (the header is not trusted: a fmt chunk is 16 .. 40 bytes)
                std::vector<uint8_t> f(size);
                if (std::fread(f.data(), 1, size, file_) != size) return false;
                tag = u16(&f[0]); ch = u16(&f[2]); inputRate_ = (int32_t)u32(&f[4]); bits = u16(&f[14]);
                if (tag == 0xFFFE && size >= 26) tag = u16(&f[24]);              // WAVE_FORMAT_EXTENSIBLE: the sub-format's first word
                have_fmt = true;
                if (size & 1) std::fseek(file_, 1, SEEK_CUR);
            } else if (std::memcmp(ck, "data", 4) == 0) {
                dataStart_ = std::ftell(file_);
                if (dataStart_ < 0 || std::fseek(file_, 0, SEEK_END) != 0) return false;
                const long end = std::ftell(file_);
                if (end < dataStart_) return false;
                dataBytes_ = std::min<int64_t>((int64_t)size, (int64_t)end - dataStart_);
                break;
            } else if (std::fseek(file_, (long)size + (long)(size & 1), SEEK_CUR) != 0) return false;
        }
        if (!have_fmt || (ch != 1 && ch != 2)) return false;
        if (inputRate_ <= 0 || inputRate_ > 100000000) return false;            // (a rate of 0 or beyond 2^31 would make the 10 ms period empty / negative)
        if (tag == 1 && bits == 16) fmt_ = 0; else if (tag == 1 && bits == 8) fmt_ = 1; else if (tag == 1 && bits == 32) fmt_ = 2;
        else if (tag == 3 && bits == 32) fmt_ = 3; else return false;
        channels_ = (int)ch; bytesPerValue_ = (int)bits / 8;
        frames_ = dataBytes_ / ((int64_t)bytesPerValue_ * channels_);
        if (std::fseek(file_, dataStart_, SEEK_SET) != 0) return false;
        return frames_ > 0;
    }
    // filehulp.cpp:127-147: `length` floats wanted; returns the floats delivered; wraps to the start after a short read
    int32_t readBuffer(float *out, int32_t length) {
        const int64_t want = length / 2, left = frames_ - filePos_;
        const int64_t n = std::min(want, left);
        const size_t vals = (size_t)n * (size_t)channels_;
        raw_.resize(vals * (size_t)bytesPerValue_);
        const size_t got = raw_.empty() ? 0 : std::fread(raw_.data(), (size_t)bytesPerValue_ * (size_t)channels_, (size_t)n, file_);
        const uint8_t *b = raw_.data();
        auto value = [&](size_t i) -> float {                    // what sf_readf_float delivers for value i of the chunk
            switch (fmt_) {
            case 0: return (float)(int16_t)(b[2 * i] | (b[2 * i + 1] << 8)) / 32768.0f;
            case 1: return ((float)b[i] - 128.0f) / 128.0f;
            case 2: { int32_t v; std::memcpy(&v, b + 4 * i, 4); return (float)((double)v / 2147483648.0); }
            default: { float v; std::memcpy(&v, b + 4 * i, 4); return v; }
            }
        };
        for (size_t i = 0; i < got; i++) {
            if (channels_ == 2) { out[2 * i] = value(2 * i); out[2 * i + 1] = value(2 * i + 1); }
            else { out[2 * i] = value(i); out[2 * i + 1] = 0.f; }
        }
        filePos_ += (int64_t)got; currPos_ += (int64_t)got;
        if ((int64_t)got < want) { filePos_ = 0; std::fseek(file_, dataStart_, SEEK_SET); }
        return (int32_t)(2 * got);
    }
    void run() {
        using clock = std::chrono::steady_clock;
        const auto period = std::chrono::microseconds(10000);
        const int32_t bufferSize = 2 * inputRate_ / 100;
        std::vector<float> bi((size_t)bufferSize);
        auto nextStop = clock::now();
        while (!exit_) {
            if (pausing_) { std::this_thread::sleep_for(std::chrono::microseconds(1000)); nextStop = clock::now(); continue; }
            while (space() < (size_t)bufferSize + 10) {
                if (exit_) break;
                std::this_thread::sleep_for(std::chrono::microseconds(1000));
            }
            if (exit_) break;
            nextStop += period;
            int32_t t = readBuffer(bi.data(), bufferSize);
            if (t <= 0) { std::fill(bi.begin(), bi.end(), 0.f); t = bufferSize; }
            size_t w = wr_.load(std::memory_order_relaxed);
            for (int32_t i = 0; i < t; i++) ring_[(w + (size_t)i) % kRing] = bi[(size_t)i];
            wr_.store(w + (size_t)t, std::memory_order_release);
            if (realtime_ && nextStop > clock::now()) std::this_thread::sleep_until(nextStop);
        }
    }

    bool realtime_, readerOK_ = false;
    std::atomic<bool> exit_{false}, pausing_{true};
    int32_t inputRate_ = 192000; int channels_ = 2;
    int64_t frames_ = 0, filePos_ = 0;
    std::atomic<int64_t> currPos_{0};
    std::vector<float> ring_;
    std::vector<uint8_t> raw_;
    FILE *file_ = nullptr; long dataStart_ = 0; int64_t dataBytes_ = 0; int fmt_ = 0, bytesPerValue_ = 2;
    std::atomic<size_t> rd_{0}, wr_{0};
    std::thread worker_;
};

}  // namespace fmx_host
