// fmx_design.h -- host-side filter / table design for libfmx (product code, no oracle dependency).
//
// The reference designs its filters at run time with mixed f32/f64 arithmetic
// (src/various/fir-filters.cpp); the kernels here run FOLDED versions of those filters, so the
// host first reproduces the reference taps in the reference's own arithmetic and then convolves
// them in double precision.  Every function cites the lines it follows.
#pragma once
#include <complex>
#include <cmath>
#include <cstdint>
#include <vector>

namespace fmx {
namespace design {

constexpr double kPi = 3.14159265358979323846;

// windowed-sinc prototype shared by LowPassFIR / DecimatingFIR::newKernel
// (fir-filters.cpp:45-59, 331-343): f32 taps, f64 sin/cos, Blackman window on i/N, f32 running sum.
inline float sinc_blackman(int N, float f, std::vector<float> &tmp) {
    tmp.assign((size_t)N, 0.f);
    float sum = 0.0f;
    for (int i = 0; i < N; i++) {
        if (i == N / 2)
            tmp[i] = (float)(2 * kPi * (double)f);
        else
            tmp[i] = (float)(std::sin(2 * kPi * (double)f * (double)(i - N / 2)) / (double)(i - N / 2));
        tmp[i] = (float)((double)tmp[i] *
                         (0.42 - 0.50 * std::cos(2 * kPi * (double)(float)i / (double)(float)N) +
                          0.08 * std::cos(4 * kPi * (double)(float)i / (double)(float)N)));
        sum += tmp[i];
    }
    return sum;
}

// LowPassFIR::newKernel fir-filters.cpp:41-62 -> real taps
inline std::vector<float> lowpass(int N, int32_t Fc, int32_t fs) {
    std::vector<float> tmp;
    const float f = (float)Fc / (float)fs;
    const float sum = sinc_blackman(N, f, tmp);
    for (auto &v : tmp) v = v / sum;
    return tmp;
}

// DecimatingFIR::newKernel fir-filters.cpp:327-347: kernel = (tmp/sum, tmp).  Returned as the
// normalised real taps plus the f32 `sum`, i.e. kernel[i] ~= hn[i] * (1 + j*sum).
struct DecimKernel { std::vector<float> hn; float sum; };
inline DecimKernel decim(int N, int32_t low, int32_t fs) {
    DecimKernel k;
    std::vector<float> tmp;
    const float f = (float)low / (float)fs;
    k.sum = sinc_blackman(N, f, tmp);
    k.hn.resize((size_t)N);
    for (int i = 0; i < N; i++) k.hn[i] = tmp[i] / k.sum;
    return k;
}

// fmx resampler: 128-tap Kaiser(beta=9) windowed sinc, fc = 24 kHz @ 192 kHz, unity DC gain
// (replaces libsamplerate, which is third-party and absent; see DESIGN.md "Resampler").
inline double bessel_i0(double x) {
    double s = 1, t = 1;
    for (int k = 1; k < 64; k++) { t *= (x / (2.0 * k)) * (x / (2.0 * k)); s += t; if (t < 1e-20 * s) break; }
    return s;
}
inline std::vector<float> resampler(int N = 128) {
    const double beta = 9.0, fc = 0.125;
    std::vector<double> tmp((size_t)N);
    double sum = 0;
    for (int k = 0; k < N; k++) {
        const double t = k - (N - 1) / 2.0;
        const double x = 2.0 * k / (N - 1) - 1.0;
        const double w = bessel_i0(beta * std::sqrt(1.0 - x * x)) / bessel_i0(beta);
        const double s = (t == 0.0) ? 2 * fc : std::sin(2 * kPi * fc * t) / (kPi * t);
        tmp[k] = s * w; sum += tmp[k];
    }
    std::vector<float> h((size_t)N);
    for (int k = 0; k < N; k++) h[k] = (float)(tmp[k] / sum);
    return h;
}

// BandPassFIR::newKernel fir-filters.cpp:197-222: integer halving first, f32 divisions; `v` is DSPFLOAT so the
// cos/sin are the FLOAT overloads and tmp*cosf(v)/sum is evaluated in f32.  Returns interleaved (re, im).
inline std::vector<float> bandpass(int N, int32_t low, int32_t high, int32_t fs) {
    std::vector<float> tmp;
    const float lo = (float)((high - low) / 2) / (float)fs;
    const float shift = (float)((high + low) / 2) / (float)fs;
    const float sum = sinc_blackman(N, lo, tmp);
    std::vector<float> k((size_t)2 * N);
    for (int i = 0; i < N; i++) {
        const float v = (float)((double)(i - N / 2) * (2 * kPi * (double)shift));
        k[2 * i] = tmp[i] * std::cos(v) / sum;
        k[2 * i + 1] = tmp[i] * std::sin(v) / sum;
    }
    return k;
}
// DecimatingFIR complex kernel (tmp/sum, tmp) fir-filters.cpp:345-346, interleaved
inline std::vector<float> decim_complex(int N, int32_t low, int32_t fs) {
    std::vector<float> tmp;
    const float f = (float)low / (float)fs;
    const float sum = sinc_blackman(N, f, tmp);
    std::vector<float> k((size_t)2 * N);
    for (int i = 0; i < N; i++) { k[2 * i] = tmp[i] / sum; k[2 * i + 1] = tmp[i]; }
    return k;
}
// ShapingFilter::root_raised_cosine shaping_filter.cpp:4-54 (all f64, taps stored f32)
inline std::vector<float> rrc(double gain, double sampling_freq, double symbol_rate, double alpha, int ntaps) {
    ntaps |= 1;
    const double spb = sampling_freq / symbol_rate;
    std::vector<float> taps((size_t)ntaps);
    double scale = 0;
    for (int i = 0; i < ntaps; i++) {
        double x1, x2, x3, num, den;
        const double xindx = i - ntaps / 2;
        x1 = kPi * xindx / spb;
        x2 = 4 * alpha * xindx / spb;
        x3 = x2 * x2 - 1;
        if (std::fabs(x3) >= 0.000001) {
            if (i != ntaps / 2) num = std::cos((1 + alpha) * x1) + std::sin((1 - alpha) * x1) / (4 * alpha * xindx / spb);
            else num = std::cos((1 + alpha) * x1) + (1 - alpha) * kPi / (4 * alpha);
            den = x3 * kPi;
        } else {
            if (alpha == 1) { taps[i] = -1; scale += taps[i]; continue; }
            x3 = (1 - alpha) * x1; x2 = (1 + alpha) * x1;
            num = (std::sin(x2) * (1 + alpha) * kPi - std::cos(x3) * ((1 - alpha) * kPi * spb) / (4 * alpha * xindx)
                   + std::sin(x3) * spb * spb / (4 * alpha * xindx * xindx));
            den = -32 * kPi * alpha * alpha * xindx / spb;
        }
        taps[i] = (float)(4 * alpha * num / den);
        scale += taps[i];
    }
    for (int i = 0; i < ntaps; i++) taps[i] = (float)((double)taps[i] * gain / scale);
    return taps;
}

inline std::vector<double> convolve(const std::vector<double> &a, const std::vector<double> &b) {
    std::vector<double> c(a.size() + b.size() - 1, 0.0);
    for (size_t i = 0; i < a.size(); i++)
        for (size_t j = 0; j < b.size(); j++) c[i + j] += a[i] * b[j];
    return c;
}
inline std::vector<double> to_double(const std::vector<float> &a) { return std::vector<double>(a.begin(), a.end()); }


// ---- recursive filters of the noise squelch: iir-filters.cpp (Chebyshev prototype newChebyshev :165-218, LowPassIIR :451-490,
// HighPassIIR :497-540, Bilineair :73-109).  DSPFLOAT = float there; a libm call takes the overload of its argument type
// (float arguments -> float functions, expressions with M_PI / 10.0 / 0.1 -> double, narrowed on assignment).
struct Iir { int nq; float q[16][6]; float gain; };        // per biquad A0 A1 A2 B0 B1 B2
inline float iir_chebyshev(float q[][6], int nq, int order, int apass) {
    const float Eps = (float)std::sqrt(std::pow(10.0, -0.1 * apass) - 1);
    const float x = (float)(1.0 / Eps);
    const float D = std::log(x + std::sqrt(x * x + 1)) / order;                 // sinhm1 :48-51 (float overloads)
    const float sinhD = std::sinh(D), coshD = std::cosh(D);
    int i0 = 0;
    if (order & 1) { q[0][0] = 0; q[0][1] = 0; q[0][2] = sinhD; q[0][3] = 0; q[0][4] = 1; q[0][5] = sinhD; i0 = 1; }
    for (int i = i0; i < nq; i++) {
        const float Phim = (order & 1) ? (float)(kPi * (2 * (i - 1) + 1) / (2 * order)) : (float)(kPi * (2 * i + 1) / (2 * order));
        const float sigma = -sinhD * std::sin(Phim), omega = coshD * std::cos(Phim);
        q[i][0] = 0; q[i][1] = 0; q[i][2] = sigma * sigma + omega * omega;
        q[i][3] = 1; q[i][4] = -2 * sigma; q[i][5] = sigma * sigma + omega * omega;
    }
    return (order & 1) == 0 ? (float)std::pow(10.0, 0.05 * apass) : 1.0f;
}
inline float iir_bilinear(float q[][6], int fs, int nq) {
    float gain = 1.0f;
    const float f2 = (float)(2 * fs), f4 = f2 * f2;
    for (int i = 0; i < nq; i++) {
        float *c = q[i];
        const float N0 = c[0] * f4 + c[1] * f2 + c[2], N1 = 2 * (c[2] - c[0] * f4), N2 = c[0] * f4 - c[1] * f2 + c[2];
        const float D0 = c[3] * f4 + c[4] * f2 + c[5], D1 = 2 * (c[5] - c[3] * f4), D2 = c[3] * f4 - c[4] * f2 + c[5];
        c[0] = 1.0f; c[1] = N1 / N0; c[2] = N2 / N0; c[3] = 1.0f; c[4] = D1 / D0; c[5] = D2 / D0;
        gain *= (N0 / D0);
    }
    return gain;
}
inline Iir iir_chebyshev_lowhigh(bool highpass, int order, int32_t fpass, int32_t fs) {
    Iir f{}; f.nq = ((order + 1) & 0176) / 2;
    if (2 * fpass >= fs) fpass = fs / 4;
    const float omega = (float)(2.0 * fs * std::tan((2 * kPi * fpass) / (2 * fs)));      // warpDtoA :111-113
    f.gain = iir_chebyshev(f.q, f.nq, order, -1);
    for (int i = 0; i < f.nq; i++) {
        float *c = f.q[i];
        if (!highpass) { c[1] = c[1] * omega; c[4] = c[4] * omega; c[2] = c[2] * omega * omega; c[5] = c[5] * omega * omega; }
        else {
            const float A0 = c[0], A1 = c[1], A2 = c[2], B0 = c[3], B1 = c[4], B2 = c[5];
            f.gain *= A2 / B2;
            c[0] = 1.0f; c[3] = 1.0f; c[1] = (A1 / A2) * omega; c[4] = (B1 / B2) * omega;
            c[2] = (A0 / A2) * omega * omega; c[5] = (B0 / B2) * omega * omega;
        }
    }
    f.gain *= iir_bilinear(f.q, fs, f.nq);
    return f;
}

// BandPassIIR iir-filters.cpp:552-595 (Butterworth prototype newButterworth :120-163, unnormalizeBP :325-385, cQuadratic
// :308-313): DSPCOMPLEX = std::complex<float>, whose *, / and std::sqrt are used exactly as the reference writes them.
inline float iir_butterworth(float q[][6], int nq, int order, int apass) {
    const float Eps = (float)std::sqrt(std::pow(10.0, -0.1 * apass) - 1);
    const float R = (float)(1.0 / std::pow((double)Eps, 1.0 / order));
    int i0 = 0;
    if (order & 1) { q[0][0] = 0; q[0][1] = 0; q[0][2] = R; q[0][3] = 0; q[0][4] = 1; q[0][5] = R; i0 = 1; }
    for (int i = i0; i < nq; i++) {
        const float Phim = (order & 1) ? (float)(kPi * (2 * (i - 1) + order + 1) / (2 * order)) : (float)(kPi * (2 * i + order + 1) / (2 * order));
        const float sigma = R * std::cos(Phim), omega = R * std::sin(Phim);
        q[i][0] = 0; q[i][1] = 0; q[i][2] = sigma * sigma + omega * omega;
        q[i][3] = 1; q[i][4] = -2 * sigma; q[i][5] = sigma * sigma + omega * omega;
    }
    return 1.0f;
}
inline void iir_cquadratic(std::complex<float> A, std::complex<float> B, std::complex<float> C, std::complex<float> *D, std::complex<float> *E) {
    auto cmul = [](std::complex<float> x, float y) { return std::complex<float>(x.real() * y, x.imag() * y); };
    const std::complex<float> temp = std::sqrt(B * B - cmul(A * C, 4.0f));
    *D = (-B + temp) / cmul(A, 2.0f);
    *E = (-B - temp) / cmul(A, 2.0f);
}
inline Iir iir_butterworth_bandpass(int order, int32_t flow, int32_t fhigh, int32_t fs) {
    typedef std::complex<float> cf;
    Iir f{}; f.nq = (order + 1) & 0176; order = (order + 1) & 0176;
    float temp[16][6];
    if (flow >= fs / 2) flow = (int)(0.2 * fs);
    if (fhigh >= fs / 2) fhigh = (int)(0.3 * fs);
    const float omegaL = (float)(2.0 * fs * std::tan((2 * kPi * flow) / (2 * fs))), omegaH = (float)(2.0 * fs * std::tan((2 * kPi * fhigh) / (2 * fs)));
    const float Wo = std::sqrt(omegaL * omegaH), BW = omegaH - omegaL;
    const int nb = f.nq / 2;
    f.gain = iir_butterworth(temp, nb, order, -1);
    for (int i = 0; i < nb; i++) {
        float *t = temp[i], *q0 = f.q[2 * i], *q1 = f.q[2 * i + 1];
        cf A, B, C, D, E;
        // (Butterworth numerators have A0 == 0: the first branch of unnormalizeBP)
        q0[0] = 0.0f; q0[1] = std::sqrt(t[2]) * BW; q0[2] = 0.0f;
        q1[0] = 0.0f; q1[1] = std::sqrt(t[2]) * BW; q1[2] = 0.0f;
        A = cf(t[3], 0.0f); B = cf(t[4], 0.0f); C = cf(t[5], 0.0f);
        iir_cquadratic(A, B, C, &D, &E);
        A = cf(1.0f, 0.0f); B = cf((-D).real() * BW, (-D).imag() * BW); C = cf(Wo * Wo, 0.0f);
        iir_cquadratic(A, B, C, &D, &E);
        q0[3] = 1.0f; q0[4] = (float)(-2.0 * D.real()); q0[5] = (D * std::conj(D)).real();
        q1[3] = 1.0f; q1[4] = (float)(-2.0 * E.real()); q1[5] = (E * std::conj(E)).real();
    }
    f.gain *= iir_bilinear(f.q, fs, f.nq);
    return f;
}
// rdsDecoder_1's matched filter, rds-decoder-1.cpp:48-92 (43 taps at 24 kS/s)
inline std::vector<float> rds1_match_kernel(int32_t rate) {
    const float synchronizerSamples = rate / (float)1187.5;
    const int symbolCeiling = (int)std::ceil(synchronizerSamples);
    const int length = (symbolCeiling & ~01) + 1;
    std::vector<float> k((size_t)(2 * length + 1), 0.f);
    for (int i = 1; i <= length; i++) {
        const float x = (float)(((float)i) / rate * 1187.5);
        k[length + i] = (float)(0.75 * std::cos(4 * kPi * x) * ((1.0 / (1.0 / x - 64.01 * x)) - ((1.0 / (9.0 / x - 64.01 * x)))));
        k[length - i] = (float)(-0.75 * std::cos(4 * kPi * x) * ((1.0 / (1.0 / x - 64.01 * x)) - ((1.0 / (9.0 / x - 64.01 * x)))));
    }
    return k;
}
// Second converter workingRate -> audioRate (sendSampletoOutput fm-processor.cpp:825-838, theConverter :89-91: libsamplerate in
// the reference; the fmx design here, the same as the oracle's fmo_conv2_design): rational resampler p / q = audioRate /
// workingRate, polyphase Kaiser (beta 9) windowed sinc, cut-off 0.92 of the lower Nyquist rate, nt = 32 max (1, ceil (q / p))
// taps per phase; out[m] = sum_k taps[(m q) mod p][k] x[floor (m q / p) - k].
constexpr int CONV2_MAXP = 640, CONV2_MAXNT = 256;
inline bool design_conv2(int inRate, int outRate, int *pp, int *pq, int *pnt, std::vector<float> *taps) {
    int a = inRate, b = outRate; while (b) { const int t = a % b; a = b; b = t; }
    const int p = outRate / a, q = inRate / a;
    int nt = 32 * std::max(1, (q + p - 1) / p);
    if (nt > CONV2_MAXNT) nt = CONV2_MAXNT;
    *pp = p; *pq = q; *pnt = nt;
    if (p > CONV2_MAXP) return false;
    const long N = (long)nt * p;
    const double beta = 9.0, fc = 0.5 * 0.92 / (double)std::max(p, q), c = (N - 1) / 2.0;
    std::vector<double> h((size_t)N); double sum = 0;
    for (long i = 0; i < N; i++) {
        const double t = i - c, x = 2.0 * i / (N - 1) - 1.0;
        const double w = bessel_i0(beta * std::sqrt(1.0 - x * x)) / bessel_i0(beta);
        const double sv = (t == 0.0) ? 2 * fc : std::sin(2 * kPi * fc * t) / (kPi * t);
        h[(size_t)i] = sv * w; sum += h[(size_t)i];
    }
    taps->assign((size_t)p * nt, 0.f);
    for (int ph = 0; ph < p; ph++)
        for (int k = 0; k < nt; k++) (*taps)[(size_t)ph * nt + k] = (float)(h[(size_t)((long)k * p + ph)] * (double)p / sum);
    return true;
}
}  // namespace design

}  // namespace fmx
