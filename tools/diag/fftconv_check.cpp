// Host check of csrc/fmx_fftconv.h (run by tests/test_fftconv_cpu.py): the forward / backward halves, thread by thread as the
// device runs them, against a direct convolution in double precision.  Prints the worst relative error of the valid outputs.
#include "../../sdr-j-fm_amd/csrc/fmx_fftconv.h"
#include <cstdio>
#include <cstdlib>
#include <random>
using namespace fmx::fftc;
int main(int argc, char **argv) {
    const int ntaps = argc > 1 ? atoi(argv[1]) : 295;
    std::mt19937 rng(12345);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float2> W(W_COUNT), Hs(N), in(N), slot(N), out(N);
    std::vector<float> taps(ntaps);
    for (auto &t : taps) t = nd(rng) * 0.05f;
    make_twiddles(W.data());
    make_spectrum(taps.data(), ntaps, Hs.data(), W.data());
    const int nin = N - 218;                     // 1536 + 294 inputs, the rest zero padding
    for (int n = 0; n < N; n++) in[n] = n < nin ? make_float2(nd(rng), nd(rng)) : make_float2(0.f, 0.f);
    host_forward(in.data(), slot.data(), W.data());
    for (int k = 0; k < N; k++) slot[k] = cmul(slot[k], Hs[k]);
    host_backward(slot.data(), out.data(), W.data());
    double worst = 0, scale = 0;
    for (int n = ntaps - 1; n < nin; n++) {
        double re = 0, im = 0;
        for (int k = 0; k < ntaps; k++) { re += (double)taps[k] * in[n - k].x; im += (double)taps[k] * in[n - k].y; }
        worst = std::fmax(worst, std::fmax(std::fabs(re - out[n].x), std::fabs(im - out[n].y)));
        scale = std::fmax(scale, std::fmax(std::fabs(re), std::fabs(im)));
    }
    // a pure transform check: backward (forward (x)) = N x
    host_forward(in.data(), slot.data(), W.data());
    host_backward(slot.data(), out.data(), W.data());
    double rt = 0;
    for (int n = 0; n < N; n++) rt = std::fmax(rt, std::fmax(std::fabs(out[n].x / N - in[n].x), std::fabs(out[n].y / N - in[n].y)));
    printf("{\"conv_worst_abs\": %.3e, \"conv_scale\": %.3e, \"roundtrip_worst_abs\": %.3e}\n", worst, scale, rt);
    return 0;
}
