// qt_demo.cpp -- drives fmx_qt::fmProcessor the way RadioInterface drives the reference's fmProcessor (radio.cpp:915-948):
// construct with the reference's argument list, apply the GUI's setters, start() the thread, let the event loop deliver the
// queued signals, stop().  Then checks, like a GUI would use them, the three scope rings against the library's taps.
// Usage: qt_demo iq.f32 pcm_out.f32 [rds]
#include <QCoreApplication>
#include <QTimer>
#include <cstdio>
#include <cstring>
#include "fmx_qt_host.h"

struct MemDevice : deviceHandler {
    std::vector<std::complex<float>> data; std::atomic<size_t> pos{0};
    int32_t Samples() override { return (int32_t)(data.size() - pos.load()); }
    int32_t getSamples(std::complex<float> *dst, int32_t n, uint8_t) override {
        std::memcpy(dst, data.data() + pos.load(), sizeof(std::complex<float>) * (size_t)n); pos += (size_t)n; return n;
    }
};

int main(int argc, char **argv) {
    QCoreApplication app(argc, argv);
    if (argc < 3) { std::fprintf(stderr, "usage: %s iq.f32 pcm_out.f32 [rds]\n", argv[0]); return 2; }
    const bool rds = argc > 3;
    MemDevice dev; audioSink sink; RadioInterface gui;
    FILE *fi = std::fopen(argv[1], "rb");
    if (!fi) return 2;
    std::fseek(fi, 0, SEEK_END); long bytes = std::ftell(fi); std::fseek(fi, 0, SEEK_SET);
    dev.data.resize((size_t)bytes / sizeof(std::complex<float>));
    if (std::fread(dev.data.data(), 1, (size_t)bytes, fi) != (size_t)bytes) return 2;
    std::fclose(fi);
    const int spectrumSize = 2048, repeatRate = 10;
    RingBuffer<std::complex<float>> hfBuffer(1 << 25), lfBuffer(1 << 22), iqBuffer(1 << 16);
    // radio.cpp:915-930
    fmx_qt::fmProcessor p(&dev, &gui, &sink, nullptr, 2304000, 192000, 48000, 48000, 1024, spectrumSize, repeatRate, 0,
                          &hfBuffer, &lfBuffer, &iqBuffer, 20);
    if (!p.ok()) { std::fprintf(stderr, "fmx: %s\n", p.lastError().c_str()); return 1; }
    p.setfmMode(fmx_qt::fmProcessor::FM_Mode::Stereo);
    p.setFMdecoder("FM Mixed Demod");
    p.setBandwidth("165kHz"); p.setlfcutoff(15000); p.setDeemphasis(50); p.setVolume(-6.0f);
    p.setAutoMonoMode(true); p.setPSSMode(true); p.setDCRemove(true);
    p.setlfPlotType(fmx_qt::fmProcessor::ELfPlot::DEMODULATOR);
    if (rds) p.setfmRdsSelector(fmx_qt::rdsDecoder::ERdsMode::RDS_2);
    p.start();                                                     // QThread::start -> run()
    QTimer poll;
    QObject::connect(&poll, &QTimer::timeout, [&]() { if (dev.Samples() < 16384) { p.stop(); app.quit(); } });
    poll.start(5);
    app.exec();
    QCoreApplication::processEvents();                             // signals queued just before the thread ended
    // HF ring = the raw blocks, in order (fm-processor.cpp:420)
    const size_t pulled = dev.pos.load();
    std::vector<std::complex<float>> hf(pulled);
    const int32_t got = hfBuffer.getDataFromBuffer(hf.data(), (int32_t)pulled);
    const bool hfOk = (size_t)got == pulled && std::memcmp(hf.data(), dev.data.data(), sizeof(std::complex<float>) * pulled) == 0;
    // LF ring = spectrumSize demodulator samples per lfBufferLoaded (imaginary part 0), :605-607, 650-660
    const uint32_t lfAvail = lfBuffer.GetRingBufferReadAvailable();
    std::vector<std::complex<float>> lf(lfAvail);
    lfBuffer.getDataFromBuffer(lf.data(), (int32_t)lfAvail);
    double lfImag = 0, lfEnergy = 0;
    for (auto &v : lf) { lfImag += std::fabs(v.imag()); lfEnergy += (double)v.real() * v.real(); }
    FILE *fl = std::fopen((std::string(argv[2]) + ".lf").c_str(), "wb");
    std::fwrite(lf.data(), sizeof(std::complex<float>), lf.size(), fl); std::fclose(fl);
    const uint64_t iqTotal = iqBuffer.totalWritten();
    std::printf("frames %zu meta %d peaks %d hf %d hfring %d lf %d lfring %u lfimag %.3g lfenergy %.3g lfnew %d iq %d iqring %llu squelch %d locked %d "
                "strength %.4f pss %d peakl %.2f peakr %.2f dcif %.5f retunes %d rdssync %d pi %d pty %d crc %d sync %d groups %d ber %.4f\n",
                sink.pcm.size(), gui.nMeta.load(), gui.nPeaks.load(), gui.nHf.load(), (int)hfOk, gui.nLf.load(), lfAvail, lfImag, lfEnergy, (int)gui.lfNew,
                gui.nIq.load(), (unsigned long long)iqTotal, gui.nSq.load(), (int)gui.locked, gui.lockStrength, gui.pssState, gui.lastL, gui.lastR, gui.dcIf,
                gui.retunes, (int)gui.rdsSync, gui.pi, gui.pty, gui.crc, gui.sync, gui.nGroup.load(), gui.ber);
    std::printf("ptyname=%s|label=%s|text=%s\n", gui.ptyName.toUtf8().constData(), gui.label.toUtf8().constData(), gui.text.toUtf8().constData());
    FILE *fo = std::fopen(argv[2], "wb");
    std::fwrite(sink.pcm.data(), sizeof(std::complex<float>), sink.pcm.size(), fo);
    std::fclose(fo);
    return 0;
}
