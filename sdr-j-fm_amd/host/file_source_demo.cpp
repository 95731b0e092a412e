// file_source_demo.cpp -- fmx_host::FileSource on its own (no GPU): plays a .wav through the paced reader and writes what
// getSamples delivered, with the elapsed time.   usage: file_source_demo in.wav out.f32 n_complex block realtime(0|1) [attenuation]
// Build: g++ -std=c++17 -O2 -pthread file_source_demo.cpp -o file_source_demo     (tests/test_host_logic.py)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "file_source.h"

int main(int argc, char **argv) {
    if (argc < 6) return 2;
    bool ok = false;
    fmx_host::FileSource src(argv[1], &ok, std::atoi(argv[5]) != 0);
    if (!ok) { std::printf("open failed\n"); return 1; }
    const long n = std::atol(argv[3]); const int block = std::atoi(argv[4]);
    const float att = argc > 6 ? (float)std::atof(argv[6]) : 1.0f;
    std::vector<std::complex<float>> out((size_t)n), buf((size_t)block);
    const int32_t before = src.Samples();                       // the reader starts paused: nothing may arrive
    std::this_thread::sleep_for(std::chrono::milliseconds(30));
    const int32_t still = src.Samples();
    const auto t0 = std::chrono::steady_clock::now();
    src.restartReader();
    long got = 0;
    while (got < n) {
        const int32_t m = (int32_t)std::min<long>(block, n - got);
        if (src.getSamples(buf.data(), m, att) != m) return 1;
        std::copy(buf.begin(), buf.begin() + m, out.begin() + got); got += m;
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    src.stopReader();
    std::printf("rate %d frames %lld paused_before %d paused_after %d seconds %.4f\n", src.getRate(), (long long)src.samplesinFile(), before, still, sec);
    FILE *fo = std::fopen(argv[2], "wb");
    std::fwrite(out.data(), sizeof(std::complex<float>), out.size(), fo);
    std::fclose(fo);
    return 0;
}
