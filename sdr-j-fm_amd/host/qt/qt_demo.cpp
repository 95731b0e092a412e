// qt_demo.cpp -- drives fmx_qt::fmProcessor the way RadioInterface drives the reference's fmProcessor (radio.cpp:915-948):
// construct with (device, GUI object, sink), apply the GUI's setters, start() the thread, let the event loop deliver the
// queued signals, stop().  Usage: qt_demo iq.f32 pcm_out.f32
#include <QCoreApplication>
#include <QTimer>
#include <cstdio>
#include <cstring>
#include <mutex>
#include "qt_demo.h"

struct MemDevice : fmx_qt::DeviceHandler {
    std::vector<std::complex<float>> data; std::atomic<size_t> pos{0};
    int32_t Samples() override { return (int32_t)(data.size() - pos.load()); }
    int32_t getSamples(std::complex<float> *dst, int32_t n) override {
        std::memcpy(dst, data.data() + pos.load(), sizeof(std::complex<float>) * (size_t)n); pos += (size_t)n; return n;
    }
};
struct MemSink : fmx_qt::AudioSink {
    std::vector<std::complex<float>> pcm;
    int32_t putSamples(std::complex<float> *f, int32_t n) override { pcm.insert(pcm.end(), f, f + n); return n; }
};

int main(int argc, char **argv) {
    QCoreApplication app(argc, argv);
    if (argc < 3) { std::fprintf(stderr, "usage: %s iq.f32 pcm_out.f32\n", argv[0]); return 2; }
    MemDevice dev; MemSink sink; Receiver gui;
    FILE *fi = std::fopen(argv[1], "rb");
    if (!fi) return 2;
    std::fseek(fi, 0, SEEK_END); long bytes = std::ftell(fi); std::fseek(fi, 0, SEEK_SET);
    dev.data.resize((size_t)bytes / sizeof(std::complex<float>));
    if (std::fread(dev.data.data(), 1, (size_t)bytes, fi) != (size_t)bytes) return 2;
    std::fclose(fi);
    fmx_qt::fmProcessor p(&dev, &gui, &sink);
    if (!p.ok()) { std::fprintf(stderr, "fmx: %s\n", p.lastError().c_str()); return 1; }
    p.setfmMode(fmx_qt::fmProcessor::FM_Mode::Stereo);
    p.setFMdecoder("FM Mixed Demod");
    p.setBandwidth("165kHz"); p.setlfcutoff(15000); p.setDeemphasis(50); p.setVolume(-6.0f);
    p.setAutoMonoMode(true); p.setPSSMode(true); p.setDCRemove(true);
    p.start();                                                     // QThread::start -> run()
    QTimer poll;
    QObject::connect(&poll, &QTimer::timeout, [&]() { if (dev.Samples() < 16384) { p.stop(); app.quit(); } });
    poll.start(5);
    app.exec();
    QCoreApplication::processEvents();                             // signals queued just before the thread ended
    std::printf("frames %zu meta %d peaks %d hf %d lf %d squelch %d locked %d strength %.4f pss %d peak %.2f %.2f\n", sink.pcm.size(), gui.nMeta.load(),
                gui.nPeaks.load(), gui.nHf.load(), gui.nLf.load(), gui.nSq.load(), (int)gui.locked, gui.lockStrength, gui.pssState, gui.lastL, gui.lastR);
    FILE *fo = std::fopen(argv[2], "wb");
    std::fwrite(sink.pcm.data(), sizeof(std::complex<float>), sink.pcm.size(), fo);
    std::fclose(fo);
    return 0;
}
