// adapter_demo.cpp -- drives fmx_host::FmProcessor exactly the way RadioInterface drives fmProcessor
// (radio.cpp:915-948): a memory-backed deviceHandler, a capturing audioSink, the GUI's default setters.
// Build:  g++ -std=c++17 -O2 adapter_demo.cpp -L../lib -lfmx -Wl,-rpath,'$ORIGIN/../lib' -o adapter_demo
// Used by tests/test_gpu_parity.py::test_cpp_adapter (GPU) and compiled (syntax only) by the CPU suite.
#include <cstdio>
#include <cstring>
#include "fm_processor_adapter.h"
#include "file_source.h"

struct MemDevice : fmx_host::DeviceHandler {
    std::vector<std::complex<float>> data; size_t pos = 0;
    int32_t Samples() override { return (int32_t)(data.size() - pos); }
    int32_t getSamples(std::complex<float> *dst, int32_t n) override {
        std::memcpy(dst, data.data() + pos, sizeof(std::complex<float>) * (size_t)n); pos += (size_t)n; return n;
    }
};
struct MemSink : fmx_host::AudioSink {
    std::vector<std::complex<float>> pcm;
    int32_t putSamples(std::complex<float> *f, int32_t n) override { pcm.insert(pcm.end(), f, f + n); return n; }
};

int main(int argc, char **argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s iq.f32 pcm_out.f32 | in.wav pcm_out.f32 blocks [realtime]\n", argv[0]); return 2; }
    const std::string in = argv[1];
    if (in.size() > 4 && in.substr(in.size() - 4) == ".wav") {
        // BASELINE configs[0] literally: the file reader (fileHulp semantics, paced when asked) -> fmProcessor -> sink
        bool ok = false;
        fmx_host::FileSource src(in, &ok, argc > 4 && std::atoi(argv[4]) != 0);
        if (!ok || argc < 4) return 2;
        MemSink sink;
        fmx_host::FmProcessor p(&src, &sink, src.getRate());
        if (!p.ok()) { std::fprintf(stderr, "fmx: %s\n", p.lastError().c_str()); return 1; }
        p.setfmMode(fmx_host::FmProcessor::FM_Mode::Mono);
        p.setFMdecoder("FM Mixed Demod");
        p.setBandwidth("Off"); p.setlfcutoff(15000); p.setDeemphasis(50); p.setVolume(-6.0f);
        src.restartReader();
        int idle = 0;                                              // run_block () is false while the reader has less than a block: wait like run () does
        for (int b = 0; b < std::atoi(argv[3]) && idle < 20000; ) {
            if (p.run_block()) { b++; idle = 0; }
            else { if (!p.lastError().empty()) { std::fprintf(stderr, "fmx: %s\n", p.lastError().c_str()); return 1; } idle++; std::this_thread::sleep_for(std::chrono::milliseconds(1)); }
        }
        std::printf("frames %zu\n", sink.pcm.size());
        FILE *fo = std::fopen(argv[2], "wb");
        std::fwrite(sink.pcm.data(), sizeof(std::complex<float>), sink.pcm.size(), fo);
        std::fclose(fo);
        return 0;
    }
    MemDevice dev; MemSink sink;
    FILE *fi = std::fopen(argv[1], "rb");
    if (!fi) return 2;
    std::fseek(fi, 0, SEEK_END); long bytes = std::ftell(fi); std::fseek(fi, 0, SEEK_SET);
    dev.data.resize((size_t)bytes / sizeof(std::complex<float>));
    if (std::fread(dev.data.data(), 1, (size_t)bytes, fi) != (size_t)bytes) return 2;
    std::fclose(fi);
    fmx_host::FmProcessor p(&dev, &sink);
    if (!p.ok()) { std::fprintf(stderr, "fmx: %s\n", p.lastError().c_str()); return 1; }
    // the GUI's effective defaults
    p.setfmMode(fmx_host::FmProcessor::FM_Mode::Stereo);
    p.setFMdecoder("FM Mixed Demod");
    p.setBandwidth("165kHz"); p.setlfcutoff(15000); p.setDeemphasis(50); p.setVolume(-6.0f);
    p.setAutoMonoMode(true); p.setPSSMode(true); p.setDCRemove(true);
    while (p.run_block()) {}
    float strength = 0; bool locked = p.isPilotLocked(strength);
    std::printf("frames %zu locked %d strength %.4f\n", sink.pcm.size(), (int)locked, strength);
    FILE *fo = std::fopen(argv[2], "wb");
    std::fwrite(sink.pcm.data(), sizeof(std::complex<float>), sink.pcm.size(), fo);
    std::fclose(fo);
    return 0;
}
