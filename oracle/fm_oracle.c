/*
 * fm_oracle.c -- CPU restatement of the sdr-j-fm src/fm chain.  TEST INFRASTRUCTURE ONLY
 * (see fm_oracle.h).  Build: gcc -O2 -std=gnu11 -ffp-contract=off (no -march, no fast-math) so
 * that every expression is evaluated in the type the reference's C++ evaluates it in:
 * the reference is built -std=c++17 -O2 without FMA (fmreceiver.pro:12-18).
 *
 * Naming: "f32"/"f64" in comments = the type C++ promotion gives the reference expression.
 * NB: the reference says `using namespace std;` (fm-constants.h:48) and includes <math.h>, so a
 * libm call with a float argument resolves to the float overload (atan(float) == atanf, etc.).
 */
#define _GNU_SOURCE
#include "fm_oracle.h"
#include <complex.h>
#undef I   /* I,Q are sample names here; CMPLXF builds complex values */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_PI_4
#define M_PI_4 0.78539816339744830962
#endif

/* ------------------------------------------------------------------ complex helpers */
typedef fmo_c32 c32;
static inline c32 C(float re, float im) { c32 r = { re, im }; return r; }
/* std::complex<float> operator* : (ac-bd, ad+bc) in f32 (libstdc++ / __mulsc3 finite path) */
static inline c32 cmul(c32 a, c32 b) { return C(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static inline c32 cadd(c32 a, c32 b) { return C(a.re + b.re, a.im + b.im); }
static inline c32 csub(c32 a, c32 b) { return C(a.re - b.re, a.im - b.im); }
static inline c32 cscale(c32 a, float s) { return C(a.re * s, a.im * s); }
static inline c32 cconj(c32 a) { return C(a.re, -a.im); }
/* std::abs(std::complex<float>) -> cabsf -> hypotf */
static inline float cabs32(c32 a) { return hypotf(a.re, a.im); }
/* std::exp(std::complex<float>(0, x)) -> cexpf */
static inline c32 cexp_i(float x) {
    float complex e = cexpf(CMPLXF(0.0f, x));
    return C(crealf(e), cimagf(e));
}

/* ------------------------------------------------------------------ PI_Constrain */
/* fm-constants.h:148-158 : compares in f64 (2*M_PI is double), fmod in f64, result -> f32 */
float fmo_pi_constrain(float val) {
    if (0 <= val && val < 2 * M_PI) return val;
    if (val >= 2 * M_PI) return (float)fmod((double)val, 2 * M_PI);
    if (val > -2 * M_PI) return (float)((double)val + 2 * M_PI);
    return (float)(2 * M_PI - fmod((double)-val, 2 * M_PI));
}

/* ------------------------------------------------------------------ kernel design */
/* shared windowed-sinc prototype: fir-filters.cpp:45-59 / 201-214 / 331-343 (identical text) */
static float sinc_blackman(int N, float f, float *tmp) {
    float sum = 0.0f;
    for (int i = 0; i < N; i++) {
        if (i == N / 2)
            tmp[i] = (float)(2 * M_PI * (double)f);
        else
            tmp[i] = (float)(sin(2 * M_PI * (double)f * (double)(i - N / 2)) / (double)(i - N / 2));
        /* tmp[i] *= (double) : f32*f64 -> f64 -> f32 ; note i/N window (not i/(N-1)) */
        tmp[i] = (float)((double)tmp[i] *
                 (0.42 - 0.50 * cos(2 * M_PI * (double)(float)i / (double)(float)N)
                       + 0.08 * cos(4 * M_PI * (double)(float)i / (double)(float)N)));
        sum += tmp[i];
    }
    return sum;
}

void fmo_lowpass_kernel(int N, int32_t Fc, int32_t fs, float *h) {
    /* fir-filters.cpp:41-62 : f = (float)Fc / sampleRate (f32 division) */
    float *tmp = (float *)malloc(sizeof(float) * (size_t)N);
    float f = (float)Fc / (float)fs;
    float sum = sinc_blackman(N, f, tmp);
    for (int i = 0; i < N; i++) h[i] = tmp[i] / sum;
    free(tmp);
}

void fmo_decim_kernel(int N, int32_t low, int32_t fs, c32 *k) {
    /* fir-filters.cpp:327-347 : kernel = (tmp/sum, tmp)  -- the "complex gain" quirk */
    float *tmp = (float *)malloc(sizeof(float) * (size_t)N);
    float f = (float)low / (float)fs;
    float sum = sinc_blackman(N, f, tmp);
    for (int i = 0; i < N; i++) k[i] = C(tmp[i] / sum, tmp[i]);
    free(tmp);
}

void fmo_bandpass_kernel(int N, int32_t low, int32_t high, int32_t fs, c32 *k) {
    /* fir-filters.cpp:197-222 : integer halving first, then f32 division;
       v is DSPFLOAT so cos(v)/sin(v) are the FLOAT overloads; tmp*cosf(v)/sum all f32 */
    float *tmp = (float *)malloc(sizeof(float) * (size_t)N);
    float lo = (float)((high - low) / 2) / (float)fs;
    float shift = (float)((high + low) / 2) / (float)fs;
    float sum = sinc_blackman(N, lo, tmp);
    for (int i = 0; i < N; i++) {
        float v = (float)((double)(i - N / 2) * (2 * M_PI * (double)shift));
        k[i] = C(tmp[i] * cosf(v) / sum, tmp[i] * sinf(v) / sum);
    }
    free(tmp);
}

int fmo_rrc_kernel(double gain, double sampling_freq, double symbol_rate, double alpha,
                   int ntaps, float *taps) {
    /* shaping_filter.cpp:4-54 (all f64; taps stored f32; scale accumulates the f32 taps) */
    ntaps |= 1;
    double spb = sampling_freq / symbol_rate;
    double scale = 0;
    for (int i = 0; i < ntaps; i++) {
        double x1, x2, x3, num, den;
        double xindx = i - ntaps / 2;
        x1 = M_PI * xindx / spb;
        x2 = 4 * alpha * xindx / spb;
        x3 = x2 * x2 - 1;
        if (fabs(x3) >= 0.000001) {
            if (i != ntaps / 2)
                num = cos((1 + alpha) * x1) + sin((1 - alpha) * x1) / (4 * alpha * xindx / spb);
            else
                num = cos((1 + alpha) * x1) + (1 - alpha) * M_PI / (4 * alpha);
            den = x3 * M_PI;
        } else {
            if (alpha == 1) {
                taps[i] = -1;
                scale += taps[i];
                continue;
            }
            x3 = (1 - alpha) * x1;
            x2 = (1 + alpha) * x1;
            num = (sin(x2) * (1 + alpha) * M_PI
                   - cos(x3) * ((1 - alpha) * M_PI * spb) / (4 * alpha * xindx)
                   + sin(x3) * spb * spb / (4 * alpha * xindx * xindx));
            den = -32 * M_PI * alpha * alpha * xindx / spb;
        }
        taps[i] = (float)(4 * alpha * num / den);
        scale += taps[i];
    }
    for (int i = 0; i < ntaps; i++) taps[i] = (float)((double)taps[i] * gain / scale);
    return ntaps;
}

/* ------------------------------------------------------------------ FFT */
static size_t reverse_bits(size_t val, int width) {
    size_t result = 0;
    for (int i = 0; i < width; i++, val >>= 1) result = (result << 1) | (val & 1U);
    return result;
}

/* fft-complex.cpp:50-102 : iterative radix-2 DIT, twiddles = cexpf(j * (float)(-+2*pi*i/n)),
   recomputed per call in the reference; butterfly temp = vec[l]*w; vec[l]=vec[j]-temp; vec[j]+=temp */
int fmo_fft_radix2(c32 *vec, long n_, int inverse) {
    size_t n = (size_t)n_;
    int levels = 0;
    for (size_t t = n; t > 1U; t >>= 1) levels++;
    if ((size_t)1U << levels != n) return 0;
    c32 *exptable = (c32 *)malloc((n / 2) * sizeof(c32));
    for (size_t i = 0; i < n / 2; i++) {
        /* (inverse ? 2 : -2) * M_PI * i / n : int*f64, *(double)size_t, /(double)size_t -> f32 */
        double ang = (double)(inverse ? 2 : -2) * M_PI * (double)i / (double)n;
        exptable[i] = cexp_i((float)ang);
    }
    for (size_t i = 0; i < n; i++) {
        size_t j = reverse_bits(i, levels);
        if (j > i) { c32 t = vec[i]; vec[i] = vec[j]; vec[j] = t; }
    }
    for (size_t size = 2; size <= n; size *= 2) {
        size_t halfsize = size / 2, tablestep = n / size;
        for (size_t i = 0; i < n; i += size) {
            for (size_t j = i, k = 0; j < i + halfsize; j++, k += tablestep) {
                size_t l = j + halfsize;
                c32 temp = cmul(vec[l], exptable[k]);
                vec[l] = csub(vec[j], temp);
                vec[j] = cadd(vec[j], temp);
            }
        }
        if (size == n) break;
    }
    free(exptable);
    return 1;
}

/* ------------------------------------------------------------------ overlap-add filter */
struct fmo_fftfilter {
    int fftSize, degree, numSamples, inp;
    c32 *A, *Cc, *filt, *over;
};

fmo_fftfilter *fmo_fftfilter_new(int fftSize, int degree) {
    /* fft-filters.cpp:29-49 */
    fmo_fftfilter *f = (fmo_fftfilter *)calloc(1, sizeof(*f));
    f->fftSize = fftSize; f->degree = degree; f->numSamples = fftSize - degree; f->inp = 0;
    f->A = (c32 *)calloc((size_t)fftSize, sizeof(c32));
    f->Cc = (c32 *)calloc((size_t)fftSize, sizeof(c32));
    f->filt = (c32 *)calloc((size_t)fftSize, sizeof(c32));
    f->over = (c32 *)calloc((size_t)degree, sizeof(c32));
    return f;
}
void fmo_fftfilter_free(fmo_fftfilter *f) {
    if (!f) return;
    free(f->A); free(f->Cc); free(f->filt); free(f->over); free(f);
}
void fmo_fftfilter_set_lowpass(fmo_fftfilter *f, int32_t low, int32_t rate) {
    /* fft-filters.cpp:84-95 */
    float *h = (float *)malloc(sizeof(float) * (size_t)f->degree);
    fmo_lowpass_kernel(f->degree, low, rate, h);
    for (int i = 0; i < f->degree; i++) f->filt[i] = C(h[i], 0);
    for (int i = f->degree; i < f->fftSize; i++) f->filt[i] = C(0, 0);
    fmo_fft_radix2(f->filt, f->fftSize, 0);
    f->inp = 0;
    free(h);
}
void fmo_fftfilter_set_band(fmo_fftfilter *f, int32_t low, int32_t high, int32_t rate) {
    /* fft-filters.cpp:71-82 */
    fmo_bandpass_kernel(f->degree, low, high, rate, f->filt);
    for (int i = f->degree; i < f->fftSize; i++) f->filt[i] = C(0, 0);
    fmo_fft_radix2(f->filt, f->fftSize, 0);
    f->inp = 0;
}
void fmo_fftfilter_set_hilbert(fmo_fftfilter *f) {
    /* fft-filters.cpp:177-201 (even and odd sizes) */
    int n = f->fftSize;
    if ((n & 1) == 0) {
        f->filt[0] = C(1.0f, 0);
        for (int i = 1; i < n / 2; i++) f->filt[i] = C(2.0f, 0);
        f->filt[n / 2] = C(1.0f, 0);
        for (int i = n / 2 + 1; i < n; i++) f->filt[i] = C(0, 0);
    } else {
        f->filt[0] = C(1.0f, 0);
        for (int i = 1; i < (n + 1) / 2; i++) f->filt[i] = C(2.0f, 0);
        for (int i = (n - 1) / 2 + 1; i < n; i++) f->filt[i] = C(0, 0);
    }
    f->inp = 0;
}
static void fftfilter_block(fmo_fftfilter *f, int times3) {
    /* fft-filters.cpp:104-125 (real, x3) and 139-158 (complex) */
    int n = f->fftSize;
    for (int i = f->numSamples; i < n; i++) f->A[i] = C(0, 0);
    fmo_fft_radix2(f->A, n, 0);
    for (int j = 0; j < n; j++) {
        f->Cc[j] = cmul(f->A[j], f->filt[j]);
        if (times3) f->Cc[j] = C(f->Cc[j].re * 3, f->Cc[j].im * 3);
    }
    for (int j = 0; j < n; j++) f->Cc[j] = cconj(f->Cc[j]);
    fmo_fft_radix2(f->Cc, n, 0);
    /* real variant: (float)(1.0f / fftSize); complex variant: float factor = 1.0 / fftSize.
       Both are exact powers of two for the sizes used, hence identical. */
    float factor = (float)(1.0 / (double)n);
    for (int j = 0; j < n; j++) f->Cc[j] = cscale(cconj(f->Cc[j]), factor);
    for (int j = 0; j < f->degree; j++) {
        f->Cc[j] = cadd(f->Cc[j], f->over[j]);
        f->over[j] = f->Cc[f->numSamples + j];
    }
}
c32 fmo_fftfilter_pass_c(fmo_fftfilter *f, c32 z) {
    /* fft-filters.cpp:132-163 */
    c32 sample = f->Cc[f->inp];
    f->A[f->inp] = z;
    if (++f->inp >= f->numSamples) { f->inp = 0; fftfilter_block(f, 0); }
    return sample;
}
float fmo_fftfilter_pass_r(fmo_fftfilter *f, float x) {
    /* fft-filters.cpp:97-130 */
    float sample = f->Cc[f->inp].re;
    f->A[f->inp] = C(x, 0);
    if (++f->inp >= f->numSamples) { f->inp = 0; fftfilter_block(f, 1); }
    return sample;
}
void fmo_fftfilter_run_c(fmo_fftfilter *f, const c32 *in, c32 *out, long n) {
    for (long i = 0; i < n; i++) out[i] = fmo_fftfilter_pass_c(f, in[i]);
}
void fmo_fftfilter_run_r(fmo_fftfilter *f, const float *in, float *out, long n) {
    for (long i = 0; i < n; i++) out[i] = fmo_fftfilter_pass_r(f, in[i]);
}

/* ------------------------------------------------------------------ decimating FIR */
struct fmo_decim { int N, D, ip, cnt; c32 *kernel, *buf; };

fmo_decim *fmo_decim_new(int N, int32_t low, int32_t fs, int D) {
    /* fir-filters.cpp:316-325 + Basic_FIR ctor fir-filters.h:61-71 */
    fmo_decim *d = (fmo_decim *)calloc(1, sizeof(*d));
    d->N = N; d->D = D; d->ip = 0; d->cnt = 0;
    d->kernel = (c32 *)calloc((size_t)N, sizeof(c32));
    d->buf = (c32 *)calloc((size_t)N, sizeof(c32));
    fmo_decim_kernel(N, low, fs, d->kernel);
    return d;
}
void fmo_decim_free(fmo_decim *d) { if (d) { free(d->kernel); free(d->buf); free(d); } }
int fmo_decim_pass(fmo_decim *d, c32 z, c32 *out) {
    /* fir-filters.cpp:397-424 : MAC order newest -> oldest, kernel[0] * newest */
    c32 tmp = C(0, 0);
    d->buf[d->ip] = z;
    if (++d->cnt < d->D) { d->ip = (d->ip + 1) % d->N; return 0; }
    d->cnt = 0;
    for (int i = 0; i <= d->ip; i++) tmp = cadd(tmp, cmul(d->buf[d->ip - i], d->kernel[i]));
    for (int i = d->ip + 1; i < d->N; i++) tmp = cadd(tmp, cmul(d->buf[d->N + d->ip - i], d->kernel[i]));
    d->ip = (d->ip + 1) % d->N;
    *out = tmp;
    return 1;
}
long fmo_decim_run(fmo_decim *d, const c32 *in, long n, c32 *out) {
    long m = 0;
    for (long i = 0; i < n; i++) { c32 o; if (fmo_decim_pass(d, in[i], &o)) out[m++] = o; }
    return m;
}

/* ------------------------------------------------------------------ SinCos LUT */
struct fmo_sincos { int32_t rate; double Cc; c32 *tab; };

fmo_sincos *fmo_sincos_new(int32_t rate) {
    /* sincos.cpp:45-54 : table = ((float)cos(2*M_PI*i/Rate), (float)sin(...)) from f64 */
    fmo_sincos *s = (fmo_sincos *)calloc(1, sizeof(*s));
    s->rate = rate;
    s->tab = (c32 *)malloc(sizeof(c32) * (size_t)rate);
    for (int32_t i = 0; i < rate; i++)
        s->tab[i] = C((float)cos(2 * M_PI * i / rate), (float)sin(2 * M_PI * i / rate));
    s->Cc = rate / (2 * M_PI);
    return s;
}
void fmo_sincos_free(fmo_sincos *s) { if (s) { free(s->tab); free(s); } }
const c32 *fmo_sincos_table(const fmo_sincos *s) { return s->tab; }
static int32_t sincos_index(const fmo_sincos *s, float phase) {
    /* sincos.cpp:63-67 : Phase(f32) * C(f64) -> f64 -> int32 truncation -> % Rate */
    if (phase >= 0) return ((int32_t)((double)phase * s->Cc)) % s->rate;
    return s->rate - ((int32_t)((double)phase * s->Cc)) % s->rate;
}
float fmo_sincos_sin(const fmo_sincos *s, float phase) {
    /* sincos.cpp:81-85 */
    if (phase < 0) return -fmo_sincos_sin(s, -phase);
    return s->tab[sincos_index(s, phase)].im;
}
float fmo_sincos_cos(const fmo_sincos *s, float phase) {
    /* sincos.cpp:87-91 : Phase += 2*M_PI in f64 then back to f32; fmod in f64 -> f32 */
    while (phase < 0) phase = (float)((double)phase + 2 * M_PI);
    phase = (float)fmod((double)phase, 2 * M_PI);
    return s->tab[((int32_t)((double)phase * s->Cc)) % s->rate].re;
}
c32 fmo_sincos_complex(const fmo_sincos *s, float phase) {
    /* sincos.cpp:93-97 */
    while (phase < 0) phase = (float)((double)phase + 2 * M_PI);
    phase = (float)fmod((double)phase, 2 * M_PI);
    return s->tab[((int32_t)((double)phase * s->Cc)) % s->rate];
}

/* ------------------------------------------------------------------ atan2 LUT */
#define ATSIZE 8192
struct fmo_atan { float *t[8]; float stretch; };

fmo_atan *fmo_atan_new(void) {
    /* Xtan2.cpp:12-40 : f is float so atan(f) is atanf; "* Stretch" f32; "/ M_PI" f64 -> f32.
       Stretch*0.5f etc are f32. */
    fmo_atan *a = (fmo_atan *)calloc(1, sizeof(*a));
    a->stretch = (float)M_PI;
    for (int k = 0; k < 8; k++) a->t[k] = (float *)malloc(sizeof(float) * (ATSIZE + 1));
    float St = a->stretch;
    for (int i = 0; i <= ATSIZE; i++) {
        float f = (float)i / ATSIZE;
        float ppy = (float)((double)(atanf(f) * St) / M_PI);
        a->t[0][i] = ppy;                  /* PPY */
        a->t[1][i] = St * 0.5f - ppy;      /* PPX */
        a->t[2][i] = -ppy;                 /* PNY */
        a->t[3][i] = ppy - St * 0.5f;      /* PNX */
        a->t[4][i] = St - ppy;             /* NPY */
        a->t[5][i] = ppy + St * 0.5f;      /* NPX */
        a->t[6][i] = ppy - St;             /* NNY */
        a->t[7][i] = -St * 0.5f - ppy;     /* NNX */
    }
    return a;
}
void fmo_atan_free(fmo_atan *a) { if (a) { for (int k = 0; k < 8; k++) free(a->t[k]); free(a); } }
const float *fmo_atan_table(const fmo_atan *a, int which) { return a->t[which]; }

/* index expression: (int)(SIZE * y / x + 0.5): int*f32 -> f32, /x f32, +0.5 f64, truncation */
static inline int at_idx(int size, float num, float den) {
    return (int)((double)((float)size * num / den) + 0.5);
}
float fmo_atan2(const fmo_atan *a, float y, float x) {
    /* Xtan2.cpp:56-100 */
    if (isinf(x) || isinf(y)) return 0;
    if (isnan(x) || isnan(y)) return 0;
    if (x == 0) {
        if (y == 0) return 0;
        else if (y > 0) return (float)(M_PI / 2);
        else return (float)(-M_PI / 2);
    }
    if (x > 0) {
        if (y >= 0) {
            if (x >= y) return a->t[0][at_idx(ATSIZE, y, x)];
            else        return a->t[1][at_idx(ATSIZE, x, y)];
        } else {
            if (x >= -y) return a->t[2][at_idx(-ATSIZE, y, x)];
            else         return a->t[3][at_idx(-ATSIZE, x, y)];
        }
    } else {
        if (y >= 0) {
            if (-x >= y) return a->t[4][at_idx(-ATSIZE, y, x)];
            else         return a->t[5][at_idx(-ATSIZE, x, y)];
        } else {
            if (x <= y) return a->t[6][at_idx(ATSIZE, y, x)];
            else        return a->t[7][at_idx(ATSIZE, x, y)];
        }
    }
}

/* ------------------------------------------------------------------ LO */
c32 fmo_lo_value(int32_t rate, int32_t i) {
    /* oscillator.cpp:26-35 */
    return C((float)cos(2.0 * M_PI * i / rate), (float)sin(2.0 * M_PI * i / rate));
}

/* ------------------------------------------------------------------ pllC */
struct fmo_pll {
    int32_t rate, cf;
    float NcoPhase, phaseIncr, NcoHLimit, NcoLLimit, Beta, phaseError;
    const fmo_sincos *tab; const fmo_atan *at;
};
fmo_pll *fmo_pll_new(int32_t rate, float freq, float lofreq, float hifreq, float bandwidth,
                     const fmo_sincos *tab, const fmo_atan *at) {
    /* pllC.cpp:37-60 : fac is DSPFLOAT; cf is int32_t; Beta from f64 exp */
    fmo_pll *p = (fmo_pll *)calloc(1, sizeof(*p));
    float fac = (float)(2.0 * M_PI / rate);
    p->rate = rate; p->cf = (int32_t)freq;
    p->Beta = (float)exp(-2.0 * M_PI * (double)bandwidth / 2 / rate);
    p->NcoPhase = 0; p->phaseError = 0;
    p->phaseIncr = freq * fac;
    p->NcoLLimit = lofreq * fac;
    p->NcoHLimit = hifreq * fac;
    p->tab = tab; p->at = at;
    return p;
}
void fmo_pll_free(fmo_pll *p) { free(p); }
void fmo_pll_do(fmo_pll *p, c32 signal) {
    /* pllC.cpp:67-90 */
    c32 nco = fmo_sincos_complex(p->tab, p->NcoPhase);
    c32 delay = cmul(cconj(nco), signal);
    p->phaseError = fmo_atan2(p->at, delay.im, delay.re);
    /* (1 - Beta) : int - f32 -> f32 */
    p->phaseIncr = (1 - p->Beta) * p->phaseError + p->Beta * p->phaseIncr;
    if (p->phaseIncr < p->NcoLLimit || p->phaseIncr > p->NcoHLimit)
        p->phaseIncr = (float)(p->cf * 2 * M_PI / p->rate);
    p->NcoPhase += p->phaseIncr;
    if (p->NcoPhase >= 2 * M_PI)
        p->NcoPhase = (float)fmod((double)p->NcoPhase, 2 * M_PI);
    else
        while (p->NcoPhase < 0) p->NcoPhase = (float)((double)p->NcoPhase + 2 * M_PI);
}
float fmo_pll_phase_incr(const fmo_pll *p) { return p->phaseIncr; }

/* ------------------------------------------------------------------ discriminator */
struct fmo_demod {
    int32_t rateIn, selected, arcSineSize;
    float K_FM, fm_afc, fm_cvt, Imin1, Qmin1, Imin2, Qmin2, am_carr_ampl, max_dev;
    float *Arcsine;
    fmo_sincos *tab; fmo_atan *at; fmo_pll *pll;
};
fmo_demod *fmo_demod_new(int32_t rateIn) {
    /* fm-demodulator.cpp:51-87 */
    fmo_demod *d = (fmo_demod *)calloc(1, sizeof(*d));
    d->rateIn = rateIn;
    d->tab = fmo_sincos_new(rateIn);
    d->at = fmo_atan_new();
    float F_G = (float)(0.65 * rateIn / 2);
    float Delta_F = (float)(0.95 * rateIn / 2);
    float B_FM = 2 * (Delta_F + F_G);
    d->K_FM = (float)((double)(2 * B_FM) * M_PI / (double)F_G);
    d->selected = 3;
    d->max_dev = (float)(0.95 * (0.5 * rateIn));
    /* pllC(rateIn, 0, -max, +max, 0.85*rateIn, &mySinCos): bandwidth f64 -> DSPFLOAT */
    d->pll = fmo_pll_new(rateIn, 0, -d->max_dev, +d->max_dev, (float)(0.85 * rateIn), d->tab, d->at);
    d->arcSineSize = 4 * 8192;
    d->Arcsine = (float *)malloc(sizeof(float) * (size_t)(d->arcSineSize + 1));
    for (int i = 0; i <= d->arcSineSize; i++)
        d->Arcsine[i] = (float)(asin(2.0 * i / d->arcSineSize - 1.0) / 2.0);
    d->Imin1 = d->Qmin1 = d->Imin2 = d->Qmin2 = (float)0.01;
    d->fm_afc = 0; d->fm_cvt = 1.0f; d->am_carr_ampl = 0;
    return d;
}
void fmo_demod_free(fmo_demod *d) {
    if (!d) return;
    fmo_pll_free(d->pll); fmo_sincos_free(d->tab); fmo_atan_free(d->at); free(d->Arcsine); free(d);
}
void fmo_demod_set_decoder(fmo_demod *d, int code) { d->selected = code; }
float fmo_demod_dc(const fmo_demod *d) { return d->fm_afc; }
float fmo_demod_carrier(const fmo_demod *d) { return d->am_carr_ampl; }
float fmo_demod_kfm(const fmo_demod *d) { return d->K_FM; }

static float demod_am(fmo_demod *d, c32 z) {
    /* fm-demodulator.cpp:215-241 */
    float fmDcAlpha = 0.0001f, res;
    fmo_pll_do(d->pll, z);
    res = fmo_pll_phase_incr(d->pll);
    d->fm_afc = (1 - fmDcAlpha) * d->fm_afc + fmDcAlpha * res;
    float gainLimit = 0.01f;
    res = (cabs32(z) - d->am_carr_ampl) / (d->am_carr_ampl < gainLimit ? gainLimit : d->am_carr_ampl);
    float audioLimit = 1.0f;
    if (res > audioLimit) res = audioLimit;
    else if (res < -audioLimit) res = -audioLimit;
    return res;
}
float fmo_demod_demodulate(fmo_demod *d, c32 z) {
    /* fm-demodulator.cpp:111-205 */
    float res, I, Q;
    float carrierAlpha = 0.0010f, fmDcAlpha = 0.0001f;
    float zAbs = cabs32(z);
    if ((double)zAbs <= 0.001) { I = Q = (float)0.001; }
    else { I = z.re / zAbs; Q = z.im / zAbs; }
    d->am_carr_ampl = (1.0f - carrierAlpha) * d->am_carr_ampl + carrierAlpha * zAbs;
    if (d->selected == FMO_DEC_AM) return demod_am(d, z);
    z = C(I, Q);
    int index;
    float Scaler = (float)sqrt(2.0);
    switch (d->selected) {
    default:
    case FMO_DEC_PLL:
        fmo_pll_do(d->pll, z);
        res = fmo_pll_phase_incr(d->pll);
        break;
    case FMO_DEC_MIXED:
        res = fmo_atan2(d->at, Q * d->Imin1 - I * d->Qmin1, I * d->Imin1 + Q * d->Qmin1);
        break;
    case FMO_DEC_COMPLEX_BB: {
        c32 v = cmul(z, C(d->Imin1, -d->Qmin1));
        res = fmo_atan2(d->at, v.im, v.re);
        break; }
    case FMO_DEC_REAL_BB:
        /* (Imin1*Q - Qmin1*I + 1) / 2.0 : f32 sum, /2.0 f64 -> f32; floor(res*size) f32*int->f32,
           floor(float) -> float overload */
        res = (float)((double)(d->Imin1 * Q - d->Qmin1 * I + 1) / 2.0);
        index = (int)floorf(res * (float)d->arcSineSize);
        if (index < 0) index = 0;
        if (index >= d->arcSineSize) index = d->arcSineSize;
        res = d->Arcsine[index];
        break;
    case FMO_DEC_DIFF:
        res = (d->Imin1 * (Q - d->Qmin2) - d->Qmin1 * (I - d->Imin2));
        res /= (d->Imin1 * d->Imin1 + d->Qmin1 * d->Qmin1) * Scaler;
        d->Imin2 = d->Imin1; d->Qmin2 = d->Qmin1;
        break;
    }
    d->fm_afc = (1 - fmDcAlpha) * d->fm_afc + fmDcAlpha * res;
    res = 20.0f * (res - d->fm_afc) * d->fm_cvt / d->K_FM;
    d->Imin1 = I; d->Qmin1 = Q;
    return res;
}

/* ------------------------------------------------------------------ pilot PLL */
struct fmo_pilot {
    int32_t rate, stableCnt;
    float phase, oldValue, omega, gain, lock, quadRef;
    int locked;
    const fmo_sincos *tab;
};
fmo_pilot *fmo_pilot_new(int32_t rate, float omega, float gain, const fmo_sincos *tab) {
    fmo_pilot *p = (fmo_pilot *)calloc(1, sizeof(*p));
    p->rate = rate; p->omega = omega; p->gain = gain; p->tab = tab;
    return p;
}
void fmo_pilot_free(fmo_pilot *p) { free(p); }
int fmo_pilot_locked(const fmo_pilot *p) { return p->locked; }
float fmo_pilot_strength(const fmo_pilot *p) { return p->lock; }
float fmo_pilot_phase(fmo_pilot *p, float pilot) {
    /* pilot-recover.cpp:54-83 */
    float osc = fmo_sincos_sin(p->tab, p->phase);
    float err = pilot * osc;
    const float alpha = 1.0f / 3000.0f;
    p->phase += err * p->gain;
    float cur = fmo_pi_constrain(p->phase);
    p->phase = fmo_pi_constrain(p->phase + p->omega);
    p->quadRef = (osc - p->oldValue) / p->omega;
    p->oldValue = osc;
    /* alpha*(-quadRef*pilot) f32 ; pilot_Lock*(1.0 - alpha) f64 ; sum f64 -> f32 */
    p->lock = (float)((double)(alpha * (-p->quadRef * pilot)) + (double)p->lock * (1.0 - (double)alpha));
    int tmp = (p->lock > 0.07f);
    if (tmp) {
        if (p->locked || ++p->stableCnt > (p->rate >> 1)) p->locked = 1;
    } else { p->locked = 0; p->stableCnt = 0; }
    return cur;
}

/* ------------------------------------------------------------------ PSS */
struct fmo_pss {
    int32_t rate, lockCnt, unlockCnt;
    float acc, mean, alpha, lockAlpha;
    int minimized;
    const fmo_sincos *tab;
    fmo_fftfilter *lp;
};
fmo_pss *fmo_pss_new(int32_t rate, float alpha, const fmo_sincos *tab) {
    /* stereo-separation.cpp:27-40 */
    fmo_pss *p = (fmo_pss *)calloc(1, sizeof(*p));
    p->lp = fmo_fftfilter_new(2048, 295);
    p->lockAlpha = 1.0f / rate; p->rate = rate; p->tab = tab; p->alpha = alpha;
    fmo_pss_reset(p);
    fmo_fftfilter_set_lowpass(p->lp, 15000, rate);
    return p;
}
void fmo_pss_free(fmo_pss *p) { if (p) { fmo_fftfilter_free(p->lp); free(p); } }
void fmo_pss_reset(fmo_pss *p) {
    /* stereo-separation.cpp:46-54 : does NOT clear the filter */
    p->acc = 0; p->minimized = 0; p->mean = 0; p->lockCnt = 0; p->unlockCnt = 0;
}
int fmo_pss_minimized(const fmo_pss *p) { return p->minimized; }
float fmo_pss_mean_error(const fmo_pss *p) { return p->mean; }
float fmo_pss_process(fmo_pss *p, float mux, float phase) {
    /* stereo-separation.cpp:60-109 */
    c32 sc = cscale(fmo_sincos_complex(p->tab, phase), mux);
    sc = fmo_fftfilter_pass_c(p->lp, sc);
    float error = sc.re * sc.im;
    if (!p->minimized) error *= 10.0f;
    p->acc += p->alpha * error;
    p->mean = p->lockAlpha * error + p->mean * (1.0f - p->lockAlpha);
    int tmp = (fabsf(p->mean) < 0.001f);
    if (tmp) {
        if (p->minimized || (++p->lockCnt > 3 * p->rate)) p->minimized = 1;
        p->unlockCnt = 0;
    } else {
        if (!p->minimized || (++p->unlockCnt > 3 * p->rate)) p->minimized = 0;
        p->lockCnt = 0;
    }
    if (p->acc < -M_PI_4) p->acc = (float)-M_PI_4;
    else if (p->acc > M_PI_4) p->acc = (float)M_PI_4;
    return p->acc;
}

/* ------------------------------------------------------------------ AGC / Costas */
c32 fmo_agc_process(fmo_agc *a, c32 in) {
    /* agc.h:14-18 */
    c32 out = cscale(in, a->gain);
    a->gain += a->rate * (a->ref - cabs32(out));
    return out;
}
void fmo_costas_init(fmo_costas *c, float sr, float alpha, float beta, float limitHz) {
    /* costas.h:10-19 : freqLimit(2 * M_PI * iFreqLimitHz / iSR) f64 -> f32 */
    c->alpha = alpha; c->beta = beta;
    c->freqLimit = (float)(2 * M_PI * (double)limitHz / (double)sr);
    c->freq = 0; c->phase = 0;
}
c32 fmo_costas_process(fmo_costas *c, c32 z) {
    /* costas.h:21-33 */
    c32 r = cmul(z, cexp_i(-c->phase));
    float error = r.re * r.im;
    c->freq += (c->beta * error);
    if (fabsf(c->freq) > c->freqLimit) c->freq = 0;
    c->phase += c->freq + (c->alpha * error);
    c->phase = fmo_pi_constrain(c->phase);
    return r;
}

/* ------------------------------------------------------------------ fmx resampler (own) */
/* libsamplerate (SRC_SINC_MEDIUM_QUALITY, newconverter.cpp:37) is absent and unpinned.
   Replacement, shared verbatim by the HIP path:  128-tap Kaiser(beta=9) windowed sinc,
   cutoff 24 kHz at 192 kHz (fc = 1/8 cycles/sample), unity DC gain, decimate by 4:
       out[m] = sum_{k<128} h[k] * u[4m + 3 - k]      (u = resampler input, zero history)
   so each 192-frame call of newConverter::convert yields exactly 48 frames. */
static double bessel_i0(double x) {
    double s = 1, t = 1;
    for (int k = 1; k < 64; k++) { t *= (x / (2.0 * k)) * (x / (2.0 * k)); s += t; if (t < 1e-20 * s) break; }
    return s;
}
void fmo_resampler_taps(float *h) {
    const int N = FMO_RS_TAPS;
    const double beta = 9.0, fc = 0.125;
    double tmp[FMO_RS_TAPS], sum = 0;
    for (int k = 0; k < N; k++) {
        double t = k - (N - 1) / 2.0;
        double x = 2.0 * k / (N - 1) - 1.0;
        double w = bessel_i0(beta * sqrt(1.0 - x * x)) / bessel_i0(beta);
        double s = (t == 0.0) ? 2 * fc : sin(2 * M_PI * fc * t) / (M_PI * t);
        tmp[k] = s * w; sum += tmp[k];
    }
    for (int k = 0; k < N; k++) h[k] = (float)(tmp[k] / sum);
}

typedef struct { float h[FMO_RS_TAPS]; c32 hist[FMO_RS_TAPS]; int pos; long count; } resampler;
static void resampler_init(resampler *r) { memset(r, 0, sizeof(*r)); fmo_resampler_taps(r->h); }
/* push one input frame; returns 1 and writes *out when 4m+3 was just pushed */
static int resampler_push(resampler *r, c32 v, c32 *out) {
    r->hist[r->pos] = v;
    int newest = r->pos;
    r->pos = (r->pos + 1) % FMO_RS_TAPS;
    long idx = r->count++;
    if ((idx & 3) != 3) return 0;
    float ar = 0, ai = 0;
    for (int k = 0; k < FMO_RS_TAPS; k++) {
        int j = newest - k; if (j < 0) j += FMO_RS_TAPS;
        ar += r->h[k] * r->hist[j].re;
        ai += r->h[k] * r->hist[j].im;
    }
    *out = C(ar, ai);
    return 1;
}

/* ---------- second converter (sendSampletoOutput fm-processor.cpp:825-838: theConverter (workingRate, audioRate, ...), again
 * libsamplerate in the reference, again the documented fmx design here): rational resampler p / q = audioRate / workingRate,
 * polyphase Kaiser (beta 9) windowed sinc, cut-off 0.92 of the lower Nyquist rate, nt = 32 max (1, ceil (q / p)) taps per phase;
 * output m = sum_k taps[(m q) mod p][k] x[floor (m q / p) - k], produced as soon as its newest input exists. */
static int32_t gcd32(int32_t a, int32_t b) { while (b) { int32_t t = a % b; a = b; b = t; } return a; }
int fmo_conv2_design(int32_t inRate, int32_t outRate, int32_t *pp, int32_t *pq, int32_t *pnt, float *taps /* [p][nt], may be NULL */) {
    const int32_t g = gcd32(inRate, outRate), p = outRate / g, q = inRate / g;
    int32_t nt = 32 * ((q + p - 1) / p > 1 ? (q + p - 1) / p : 1);
    if (nt > FMO_CONV2_MAXNT) nt = FMO_CONV2_MAXNT;
    *pp = p; *pq = q; *pnt = nt;
    if (p > FMO_CONV2_MAXP) return -1;
    if (!taps) return 0;
    const long N = (long)nt * p;
    const double beta = 9.0, fc = 0.5 * 0.92 / (double)(p > q ? p : q), c = (N - 1) / 2.0;
    double *h = (double *)malloc(sizeof(double) * (size_t)N), sum = 0;
    for (long i = 0; i < N; i++) {
        const double t = i - c, x = 2.0 * i / (N - 1) - 1.0;
        const double w = bessel_i0(beta * sqrt(1.0 - x * x)) / bessel_i0(beta);
        const double sv = (t == 0.0) ? 2 * fc : sin(2 * M_PI * fc * t) / (M_PI * t);
        h[i] = sv * w; sum += h[i];
    }
    for (int32_t ph = 0; ph < p; ph++)
        for (int32_t k = 0; k < nt; k++) taps[(size_t)ph * nt + k] = (float)(h[(long)k * p + ph] * (double)p / sum);
    free(h);
    return 0;
}
typedef struct { int32_t p, q, nt; float *taps; c32 hist[FMO_CONV2_MAXNT]; int pos; long in_count, out_count; } conv2;
static void conv2_init(conv2 *r, int32_t inRate, int32_t outRate) {
    memset(r, 0, sizeof(*r));
    fmo_conv2_design(inRate, outRate, &r->p, &r->q, &r->nt, NULL);
    r->taps = (float *)malloc(sizeof(float) * (size_t)r->p * (size_t)r->nt);
    fmo_conv2_design(inRate, outRate, &r->p, &r->q, &r->nt, r->taps);
}
/* push one input frame; writes the output frames that became computable, returns their number */
static int conv2_push(conv2 *r, c32 v, c32 *out, int cap) {
    r->hist[r->pos] = v;
    const int newest = r->pos;
    r->pos = (r->pos + 1) % FMO_CONV2_MAXNT;
    r->in_count++;
    int n = 0;
    while ((long long)r->out_count * r->q < (long long)r->in_count * r->p && n < cap) {
        const long long mq = (long long)r->out_count * r->q;
        const long n0 = (long)(mq / r->p); const int ph = (int)(mq - (long long)n0 * r->p);
        const int back = (int)(r->in_count - 1 - n0);            /* newest input is in_count - 1 */
        const float *t = r->taps + (size_t)ph * r->nt;
        float ar = 0, ai = 0;
        for (int k = 0; k < r->nt; k++) {
            if (n0 - k < 0) break;
            int j = newest - back - k; j %= FMO_CONV2_MAXNT; if (j < 0) j += FMO_CONV2_MAXNT;
            ar += t[k] * r->hist[j].re; ai += t[k] * r->hist[j].im;
        }
        out[n++] = C(ar, ai);
        r->out_count++;
    }
    return n;
}

/* ------------------------------------------------------------------ RDS decoder 2 */
#define RDS_MF_TAPS 45
typedef struct {
    fmo_agc agc; fmo_costas costas;
    float sps, mMu, alpha;
    int32_t sampleCount, skip;
    c32 sb[3];
    float mf[RDS_MF_TAPS]; c32 mfbuf[RDS_MF_TAPS]; int mfidx, mfsize;
    int previousBit;
} rds2;
static void rds2_init(rds2 *r, int32_t rate) {
    /* rds-decoder-2.cpp:44-78 */
    memset(r, 0, sizeof(*r));
    r->agc.rate = 2e-3f; r->agc.ref = 0.38f; r->agc.gain = 9.0f;
    fmo_costas_init(&r->costas, (float)rate, 1.0f, 0.02f, 10.0f);
    r->sps = (float)rate / (float)1187.5f;
    r->mMu = 0; r->alpha = (float)0.01; r->skip = 3; r->sampleCount = 0;
    r->mfsize = fmo_rrc_kernel(1.0, rate, 2 * 1187.5f, 1.0, RDS_MF_TAPS, r->mf);
    r->mfidx = 0;
    r->previousBit = 0;   /* uninitialised in the reference (rds-decoder-2.h); first bit undefined */
}
static int rds2_decode(rds2 *r, c32 v, c32 *m, uint8_t *d) {
    /* rds-decoder-2.cpp:83-157 */
    c32 tmp = C(0, 0);
    r->mfbuf[r->mfidx] = v;
    for (int i = 0; i < r->mfsize; i++) {
        int index = r->mfidx - i; if (index < 0) index += r->mfsize;
        tmp = cadd(tmp, cscale(r->mfbuf[index], r->mf[i]));
    }
    r->mfidx = (r->mfidx + 1) % r->mfsize;
    v = fmo_agc_process(&r->agc, tmp);
    r->sb[0] = r->sb[1]; r->sb[1] = r->sb[2]; r->sb[2] = v;
    if (++r->sampleCount >= r->skip) {
        c32 rail[3];
        for (int i = 0; i < 3; i++)
            rail[i] = C(r->sb[i].re > 0.0f ? 1.0f : -1.0f, r->sb[i].im > 0.0f ? 1.0f : -1.0f);
        float x = (rail[2].re - rail[0].re) * r->sb[1].re + (rail[2].im - rail[0].im) * r->sb[1].im;
        float y = (r->sb[2].re - r->sb[0].re) * rail[1].re + (r->sb[2].im - r->sb[0].im) * rail[1].im;
        float mm = y - x;
        r->mMu += r->sps + r->alpha * mm;
        r->skip = (int32_t)(r->mMu);
        r->mMu -= r->skip;
        c32 o = r->sb[2];
        r->sampleCount = 0;
        o = fmo_costas_process(&r->costas, o);
        int bit = (o.re >= 0);
        *d = (uint8_t)(bit ^ r->previousBit);
        r->previousBit = bit;
        *m = o;
        return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------ whole chain */
#define FFT_SIZE_RDS (2 * 16384)
#define PILOTFILTER_SIZE (2 * 384)
#define RDS_SAMPLE_DELAY (2 * (FFT_SIZE_RDS - PILOTFILTER_SIZE))
#define RDS_WIDTH (2 * 2400)
#define BLOCK 16384


/* =====================================================================================
 * Recursive filters: iir-filters.cpp.  DSPFLOAT = float; libm calls take the overload of their
 * argument type (`using namespace std`, fm-constants.h:66): float arguments -> sinhf / cosf / ...,
 * double expressions (anything with M_PI, 10.0, 0.1 ...) -> the double functions, narrowed on
 * assignment to a DSPFLOAT.
 * ===================================================================================== */
static float iir_sinhm1(float x) { return logf(x + sqrtf(x * x + 1)); }                 /* :48-51 */
static float iir_warpDtoA(int fd, int fs) { return (float)(2.0 * fs * tan((2 * M_PI * fd) / (2 * fs))); }   /* :111-113 */

static float iir_butterworth(float q[][6], int nq, int order, int apass) {               /* newButterworth :120-163 */
    float Eps = (float)sqrt(pow(10.0, -0.1 * apass) - 1);
    float R = (float)(1.0 / pow((double)Eps, 1.0 / order));
    int i0 = 0;
    if (order & 1) { q[0][0] = 0; q[0][1] = 0; q[0][2] = R; q[0][3] = 0; q[0][4] = 1; q[0][5] = R; i0 = 1; }
    for (int i = i0; i < nq; i++) {
        float Phim = (order & 1) ? (float)(M_PI * (2 * (i - 1) + order + 1) / (2 * order)) : (float)(M_PI * (2 * i + order + 1) / (2 * order));
        float sigma = R * cosf(Phim), omega = R * sinf(Phim);
        q[i][0] = 0; q[i][1] = 0; q[i][2] = sigma * sigma + omega * omega;
        q[i][3] = 1; q[i][4] = -2 * sigma; q[i][5] = sigma * sigma + omega * omega;
    }
    return 1.0f;
}
static float iir_chebyshev(float q[][6], int nq, int order, int apass) {                 /* newChebyshev :165-218 */
    float Eps = (float)sqrt(pow(10.0, -0.1 * apass) - 1);
    float D = iir_sinhm1((float)(1.0 / Eps)) / order;
    float sinhD = sinhf(D), coshD = coshf(D);
    int i0 = 0;
    if (order & 1) { q[0][0] = 0; q[0][1] = 0; q[0][2] = sinhD; q[0][3] = 0; q[0][4] = 1; q[0][5] = sinhD; i0 = 1; }
    for (int i = i0; i < nq; i++) {
        float Phim = (order & 1) ? (float)(M_PI * (2 * (i - 1) + 1) / (2 * order)) : (float)(M_PI * (2 * i + 1) / (2 * order));
        float sigma = -sinhD * sinf(Phim), omega = coshD * cosf(Phim);
        q[i][0] = 0; q[i][1] = 0; q[i][2] = sigma * sigma + omega * omega;
        q[i][3] = 1; q[i][4] = -2 * sigma; q[i][5] = sigma * sigma + omega * omega;
    }
    return (order & 1) == 0 ? (float)pow(10.0, 0.05 * apass) : 1.0f;
}
static float iir_normalized(float q[][6], int nq, int order, int apass, int ftype) {     /* newNormalized :289-306 */
    if (order <= 0) order = 6;
    return ftype == FMO_IIR_CHEBYSHEV ? iir_chebyshev(q, nq, order, apass) : iir_butterworth(q, nq, order, apass);
}
static float iir_bilinear(float q[][6], int fs, int nq) {                                 /* Bilineair :73-109 */
    float gain = 1.0f;
    const float f2 = (float)(2 * fs), f4 = f2 * f2;
    for (int i = 0; i < nq; i++) {
        float *c = q[i];
        const float N0 = c[0] * f4 + c[1] * f2 + c[2];
        const float N1 = 2 * (c[2] - c[0] * f4);
        const float N2 = c[0] * f4 - c[1] * f2 + c[2];
        const float D0 = c[3] * f4 + c[4] * f2 + c[5];
        const float D1 = 2 * (c[5] - c[3] * f4);
        const float D2 = c[3] * f4 - c[4] * f2 + c[5];
        c[0] = 1.0f; c[1] = N1 / N0; c[2] = N2 / N0;
        c[3] = 1.0f; c[4] = D1 / D0; c[5] = D2 / D0;
        gain *= (N0 / D0);
    }
    return gain;
}
void fmo_iir_lowpass(fmo_iir *f, int order, int32_t fpass, int32_t fs, int ftype) {       /* LowPassIIR :451-490 */
    memset(f, 0, sizeof(*f));
    f->nq = ((order + 1) & 0176) / 2;
    if (2 * fpass >= fs) fpass = fs / 4;
    const float omega = iir_warpDtoA(fpass, fs);
    f->gain = iir_normalized(f->q, f->nq, order, -1, ftype);
    for (int i = 0; i < f->nq; i++) {
        float *c = f->q[i];
        c[1] = c[1] * omega; c[4] = c[4] * omega;
        c[2] = c[2] * omega * omega; c[5] = c[5] * omega * omega;
    }
    f->gain *= iir_bilinear(f->q, fs, f->nq);
}
void fmo_iir_highpass(fmo_iir *f, int order, int32_t fpass, int32_t fs, int ftype) {      /* HighPassIIR :497-540 */
    memset(f, 0, sizeof(*f));
    f->nq = ((order + 1) & 0176) / 2;
    if (2 * fpass >= fs) fpass = fs / 4;
    const float omega = iir_warpDtoA(fpass, fs);
    f->gain = iir_normalized(f->q, f->nq, order, -1, ftype);
    for (int i = 0; i < f->nq; i++) {
        float *c = f->q[i];
        const float A0 = c[0], A1 = c[1], A2 = c[2], B0 = c[3], B1 = c[4], B2 = c[5];
        f->gain *= A2 / B2;
        c[0] = 1.0f; c[3] = 1.0f;
        c[1] = (A1 / A2) * omega; c[4] = (B1 / B2) * omega;
        c[2] = (A0 / A2) * omega * omega; c[5] = (B0 / B2) * omega * omega;
    }
    f->gain *= iir_bilinear(f->q, fs, f->nq);
}
float fmo_iir_pass(fmo_iir *f, float v) {                                                  /* Basic_IIR::Pass (DSPFLOAT) iir-filters.h:89-103 */
    float o = v * f->gain;
    for (int i = 0; i < f->nq; i++) {
        const float *c = f->q[i];
        const float rm1 = f->m1[i], rm2 = f->m2[i];
        const float w = o - rm1 * c[4] - rm2 * c[5];
        o = w + rm1 * c[1] + rm2 * c[2];
        f->m2[i] = f->m1[i]; f->m1[i] = w;
    }
    return o;
}
void *fmo_iir_new(int kind, int order, int32_t f1, int32_t f2, int32_t fs, int ftype) {
    fmo_iir *f = (fmo_iir *)calloc(1, sizeof(*f));
    if (kind == 0) fmo_iir_lowpass(f, order, f1, fs, ftype); else if (kind == 1) fmo_iir_highpass(f, order, f1, fs, ftype);
    else fmo_iir_bandpass(f, order, f1, f2, fs, ftype);
    return f;
}
void fmo_iir_free(void *p) { free(p); }
int fmo_iir_coeffs(void *p, float *out) {
    fmo_iir *f = (fmo_iir *)p;
    for (int i = 0; i < f->nq; i++) for (int k = 0; k < 6; k++) out[6 * i + k] = f->q[i][k];
    out[6 * f->nq] = f->gain;
    return f->nq;
}
void fmo_iir_run(void *p, const float *in, long n, float *out) { fmo_iir *f = (fmo_iir *)p; for (long i = 0; i < n; i++) out[i] = fmo_iir_pass(f, in[i]); }


/* BandPassIIR iir-filters.cpp:552-595 with unnormalizeBP :325-385 and cQuadratic :308-313.  DSPCOMPLEX = std::complex<float>:
 * its * and / are the compiler's complex multiply / divide (__mulsc3 / __divsc3), std::sqrt(complex<float>) is csqrtf --
 * the same three library routines C99's float complex uses. */
static void iir_cquadratic(float complex A, float complex B, float complex Cc, float complex *D, float complex *E) {
    const float complex AC = A * Cc;
    const float complex t = csqrtf(B * B - CMPLXF(crealf(AC) * 4.0f, cimagf(AC) * 4.0f));
    const float complex A2 = CMPLXF(crealf(A) * 2.0f, cimagf(A) * 2.0f);
    *D = (-B + t) / A2;
    *E = (-B - t) / A2;
}
void fmo_iir_bandpass(fmo_iir *f, int order, int32_t flow, int32_t fhigh, int32_t fs, int ftype) {
    memset(f, 0, sizeof(*f));
    f->nq = (order + 1) & 0176;                       /* Basic_IIR ((order + 1) & MAXORDER) */
    order = (order + 1) & 0176;
    float temp[FMO_IIR_MAXQ][6];
    if (flow >= fs / 2) flow = (int)(0.2 * fs);
    if (fhigh >= fs / 2) fhigh = (int)(0.3 * fs);
    const float omegaL = iir_warpDtoA(flow, fs), omegaH = iir_warpDtoA(fhigh, fs);
    const float Wo = sqrtf(omegaL * omegaH), BW = omegaH - omegaL;
    const int nb = f->nq / 2;
    f->gain = iir_normalized(temp, nb, order, -1, ftype);
    for (int i = 0; i < nb; i++) {                    /* unnormalizeBP */
        float *t = temp[i], *q0 = f->q[2 * i], *q1 = f->q[2 * i + 1];
        float complex A, B, Cc, D, E;
        if (t[0] == 0.0) {
            q0[0] = 0.0f; q0[1] = sqrtf(t[2]) * BW; q0[2] = 0.0f;
            q1[0] = 0.0f; q1[1] = sqrtf(t[2]) * BW; q1[2] = 0.0f;
        } else {
            A = CMPLXF(t[0], 0.0f); B = CMPLXF(t[1], 0.0f); Cc = CMPLXF(t[2], 0.0f);
            iir_cquadratic(A, B, Cc, &D, &E);
            A = CMPLXF(1.0f, 0.0f); B = CMPLXF(crealf(-D) * BW, cimagf(-D) * BW); Cc = CMPLXF(Wo * Wo, 0.0f);
            iir_cquadratic(A, B, Cc, &D, &E);
            q0[0] = 1.0f; q0[1] = (float)(-2.0 * crealf(D)); q0[2] = crealf(D * conjf(D));
            q1[0] = 1.0f; q1[1] = (float)(-2.0 * crealf(E)); q1[2] = crealf(E * conjf(E));
        }
        A = CMPLXF(t[3], 0.0f); B = CMPLXF(t[4], 0.0f); Cc = CMPLXF(t[5], 0.0f);
        iir_cquadratic(A, B, Cc, &D, &E);
        A = CMPLXF(1.0f, 0.0f); B = CMPLXF(crealf(-D) * BW, cimagf(-D) * BW); Cc = CMPLXF(Wo * Wo, 0.0f);
        iir_cquadratic(A, B, Cc, &D, &E);
        q0[3] = 1.0f; q0[4] = (float)(-2.0 * crealf(D)); q0[5] = crealf(D * conjf(D));
        q1[3] = 1.0f; q1[4] = (float)(-2.0 * crealf(E)); q1[5] = crealf(E * conjf(E));
    }
    f->gain *= 1.0f;                                   /* unnormalizeBP returns 1.0 */
    f->gain *= iir_bilinear(f->q, fs, f->nq);
}

/* ------------------------------------------------------------------ RDS decoder 1 (rds-decoder-1.cpp:43-142), behind the
 * rdsDecoder's own Costas loop (rds-decoder.cpp:40-41,76-84) */
#define RDS1_MATCH 43
typedef struct {
    fmo_costas costas;                               /* my_costas (rate, 1/16, 0.02/16, 10) */
    float fir[21], firbuf[21]; int firip;            /* rdsFilter (21, RDS_WIDTH, rate): LowPassFIR, real Pass fir-filters.h:96-108 */
    float kernel[RDS1_MATCH], buf[RDS1_MATCH]; int ip;
    fmo_iir sharp;                                   /* sharpFilter (7, 1187.5 - 6, 1187.5 + 6, rate, S_BUTTERWORTH): int32 arguments */
    float lastSyncSlope, lastSync, lastData; int previousBit;
} rds1;
static void rds1_init(rds1 *r, int32_t rate) {
    memset(r, 0, sizeof(*r));
    fmo_costas_init(&r->costas, (float)rate, 1.0f / 16.0f, 0.02f / 16.0f, 10.0f);
    fmo_lowpass_kernel(21, 2 * 2400, rate, r->fir);
    fmo_iir_bandpass(&r->sharp, 7, (int32_t)(1187.5 - 6), (int32_t)(1187.5 + 6), rate, FMO_IIR_BUTTERWORTH);
    const float synchronizerSamples = rate / (float)1187.5;
    const int symbolCeiling = (int)ceilf(synchronizerSamples);
    const int length = (symbolCeiling & ~01) + 1;     /* 21; rdsBufferSize = 2 * length + 1 = 43 */
    r->kernel[length] = 0;
    for (int i = 1; i <= length; i++) {
        const float x = (float)(((float)i) / rate * 1187.5);
        r->kernel[length + i] = (float)(0.75 * cos(4 * M_PI * x) * ((1.0 / (1.0 / x - 64.01 * x)) - ((1.0 / (9.0 / x - 64.01 * x)))));
        r->kernel[length - i] = (float)(-0.75 * cos(4 * M_PI * x) * ((1.0 / (1.0 / x - 64.01 * x)) - ((1.0 / (9.0 / x - 64.01 * x)))));
    }
}
void fmo_rds1_coeffs(float *out /* 21 + 43 + 8 * 4 + 1 */) {
    rds1 r; rds1_init(&r, 24000);
    memcpy(out, r.fir, sizeof(float) * 21); memcpy(out + 21, r.kernel, sizeof(float) * RDS1_MATCH);
    for (int i = 0; i < r.sharp.nq; i++) { out[64 + 4 * i] = r.sharp.q[i][1]; out[64 + 4 * i + 1] = r.sharp.q[i][2]; out[64 + 4 * i + 2] = r.sharp.q[i][4]; out[64 + 4 * i + 3] = r.sharp.q[i][5]; }
    out[64 + 4 * r.sharp.nq] = r.sharp.gain;
}
static int rds1_decode(rds1 *r, c32 z, c32 *m, uint8_t *d) {
    z = fmo_costas_process(&r->costas, z);
    *m = cscale(z, 4.0f);
    float v = z.re;
    {   /* rdsFilter.Pass */
        float tmp = 0;
        r->firbuf[r->firip] = v;
        for (int i = 0; i < 21; i++) { int index = r->firip - i; if (index < 0) index += 21; tmp += r->firbuf[index] * r->fir[i]; }
        r->firip = (r->firip + 1) % 21;
        v = tmp;
    }
    {   /* Match :108-121 */
        float tmp = 0;
        r->buf[r->ip] = v;
        for (int i = 0; i < RDS1_MATCH; i++) { int index = r->ip - i; if (index < 0) index += RDS1_MATCH; tmp += r->buf[index] * r->kernel[i]; }
        r->ip = (r->ip + 1) % RDS1_MATCH;
        v = tmp;
    }
    const float rdsMag = fmo_iir_pass(&r->sharp, v * v);
    const float rdsSlope = rdsMag - r->lastSync;
    int res = 0;
    r->lastSync = rdsMag;
    if ((rdsSlope < 0.0) && (r->lastSyncSlope >= 0.0)) {        /* top of the sine wave: get the data */
        const int theBit = r->lastData >= 0 ? 1 : 0;
        *d = (uint8_t)(theBit ^ r->previousBit);
        r->previousBit = theBit;
        res = 1;
    }
    r->lastData = v;
    r->lastSyncSlope = rdsSlope;
    return res;
}


/* ------------------------------------------------------------------ RDS block synchroniser, as far as rdsDecoder_3 needs it
 * (rds-blocksynchronizer.cpp:57-336: syndrome register :126-142, offset words :197-213, pushBit :215-336; rdsDecoder::processBit
 * rds-decoder.cpp:104-131 resyncs after a failed block and clears the group after a complete one).  Only the state that
 * decides the NEXT result is kept: the bit-error bookkeeping and the Meggitt pass change no decision (the reference discards
 * the corrected block). */
typedef struct { uint32_t stream; int synced, cur, bits_in_blk, n_sync_err; uint16_t blk1; } bsync;
static uint32_t bsync_offset(int blk, int typeB) { return blk == 0 ? 0xFC : blk == 1 ? 0x198 : blk == 2 ? (typeB ? 0x350 : 0x168) : 0x1B4; }
static uint32_t bsync_syndrome(uint32_t bits, uint32_t off) {
    const uint32_t block = bits ^ off;
    uint32_t reg = 0;
    for (int k = 25; k >= 0; k--) {
        const uint32_t msb = reg & (1u << 9);
        reg <<= 1;
        if (msb) reg ^= 0x5B9;
        if ((block >> k) & 1u) reg ^= 0x31B;
    }
    return reg;
}
static void bsync_resync(bsync *b) { b->cur = 0; b->synced = 0; b->bits_in_blk = 0; }
static void bsync_push(bsync *b, int bit) {             /* pushBit + the reaction of processBit */
    const int typeB = (b->blk1 >> 11) & 1;
    b->stream = (b->stream << 1) | (bit ? 1u : 0u);
    if (b->synced) {
        if (++b->bits_in_blk < 26) return;
        b->bits_in_blk = 0;
        if (bsync_syndrome(b->stream, bsync_offset(b->cur, typeB)) != 0) { bsync_resync(b); return; }      /* RDS_NO_CRC */
        if (b->cur == 1) b->blk1 = (uint16_t)(b->stream >> 10);
        if (b->cur == 3) b->blk1 = 0;                                                                      /* group complete: cleared */
        b->cur = (b->cur + 1) & 3;
        return;
    }
    if (b->cur == 0) {
        if (bsync_syndrome(b->stream & 0x3FFFFFF, bsync_offset(0, typeB)) != 0) return;                    /* waiting for block A */
        b->bits_in_blk = 0; b->cur = 1;
        return;
    }
    if (b->bits_in_blk < 25) { b->bits_in_blk++; return; }
    b->bits_in_blk = 0;
    if (bsync_syndrome(b->stream, bsync_offset(b->cur, typeB)) != 0) { b->n_sync_err++; bsync_resync(b); return; }   /* RDS_NO_SYNC */
    if (b->cur == 1) b->blk1 = (uint16_t)(b->stream >> 10);
    if (b->cur < 2) { b->cur++; return; }
    b->synced = 1;
    if (b->cur == 3) b->blk1 = 0;
    b->cur = (b->cur + 1) & 3;
}

void fmo_bsync_run(const uint8_t *bits, long n, int32_t *sync_errors, int32_t *synced) {
    bsync b; memset(&b, 0, sizeof(b));
    for (long i = 0; i < n; i++) bsync_push(&b, bits[i]);
    *sync_errors = b.n_sync_err; *synced = b.synced;
}

/* ------------------------------------------------------------------ RDS decoder 3 (rds-decoder-3.cpp:44-154) behind the
 * rdsDecoder's Costas loop (rds-decoder.cpp:92-100) */
typedef struct {
    fmo_costas costas; fmo_sincos *sc;
    float fir[21], firbuf[21]; int firip;
    float syncBuffer[21]; int p, symbolCeiling, symbolFloor;
    float omegaRDS, bitIntegrator, bitClkPhase, prev_clkState; int previousBit, Resync;
    bsync bs;
} rds3;
static void rds3_init(rds3 *r, int32_t rate) {
    memset(r, 0, sizeof(*r));
    fmo_costas_init(&r->costas, (float)rate, 1.0f / 16.0f, 0.02f / 16.0f, 10.0f);
    r->sc = fmo_sincos_new(rate);
    fmo_lowpass_kernel(21, 2 * 2400, rate, r->fir);
    const float synchronizerSamples = rate / (float)1187.5;
    r->symbolCeiling = (int)ceilf(synchronizerSamples); r->symbolFloor = (int)floorf(synchronizerSamples);
    r->omegaRDS = (float)((2 * M_PI * 1187.5) / (float)rate);
    r->Resync = 1;
}
static void rds3_synchronize(rds3 *r, int first) {      /* synchronizeOnBitClk :116-153 */
    int isHigh = 0, k = 0;
    float corr[32];
    memset(corr, 0, sizeof(corr));
    for (int i = 0; i < r->symbolCeiling; i++) {
        const float phase = (float)fmod(i * (r->omegaRDS / 2), 2 * M_PI);
        if (fmo_sincos_sin(r->sc, phase) > 0 && !isHigh) { isHigh = 1; k = 0; }
        else if (fmo_sincos_sin(r->sc, phase) < 0 && isHigh) { isHigh = 0; k = 0; }
        corr[k++] += r->syncBuffer[(first + i) % r->symbolCeiling];
    }
    int iMin = 0;
    while (iMin < r->symbolFloor && corr[iMin++] > 0);
    while (iMin < r->symbolFloor && corr[iMin++] < 0);
    r->bitClkPhase = (float)fmod(-r->omegaRDS * (iMin - 1), 2 * M_PI);
    while (r->bitClkPhase < 0) r->bitClkPhase = (float)(r->bitClkPhase + 2 * M_PI);
}
static int rds3_decode(rds3 *r, c32 z, c32 *m, uint8_t *d) {
    z = fmo_costas_process(&r->costas, z);
    *m = cscale(z, 4.0f);
    const float v = z.re;
    int res = 0;
    {   /* syncBuffer [p] = rdsFilter. Pass (v) */
        float tmp = 0;
        r->firbuf[r->firip] = v;
        for (int i = 0; i < 21; i++) { int index = r->firip - i; if (index < 0) index += 21; tmp += r->firbuf[index] * r->fir[i]; }
        r->firip = (r->firip + 1) % 21;
        r->syncBuffer[r->p] = tmp;
    }
    r->p = (r->p + 1) % r->symbolCeiling;
    if (r->Resync || (r->bs.n_sync_err > 3)) {
        rds3_synchronize(r, r->p);
        bsync_resync(&r->bs);
        r->bs.n_sync_err = 0;
        r->Resync = 0;
    }
    const float clkState = fmo_sincos_sin(r->sc, r->bitClkPhase);
    r->bitIntegrator += clkState * v;
    if (r->prev_clkState <= 0 && clkState > 0) {         /* rising edge -> look at the integrator */
        const int theBit = r->bitIntegrator >= 0;
        *d = (uint8_t)(theBit ^ r->previousBit);
        r->bitIntegrator = 0;
        r->previousBit = theBit;
        res = 1;
    }
    r->prev_clkState = clkState;
    r->bitClkPhase = (float)fmod(r->bitClkPhase + r->omegaRDS, 2 * M_PI);
    if (res) bsync_push(&r->bs, *d);                     /* rdsDecoder::doDecode: if (b) processBit (theBit) */
    return res;
}

typedef struct { float *buf; long cap, n; } tapbuf;

/* ---- squelch (squelchClass.cpp): the object fmProcessor calls per demodulated sample ------------------------------ */
void fmo_squelch_set_level(fmo_squelch *q, int n) {                      /* squelchClass.cpp:33-37 */
    q->levelThr = powf(10.0f, (n - 80) / 30.0f);
    q->noiseThr = 1.0f - n / 100.0f;
}
void fmo_squelch_init(fmo_squelch *q, int32_t threshold, int32_t keyFrequency, int32_t bufsize, int32_t sampleRate) {   /* :11-31 */
    fmo_iir_highpass(&q->high, 20, keyFrequency - 100, sampleRate, FMO_IIR_CHEBYSHEV);
    fmo_iir_lowpass(&q->low, 20, keyFrequency, sampleRate, FMO_IIR_CHEBYSHEV);
    fmo_squelch_set_level(q, threshold);
    q->hold = bufsize; q->rate = sampleRate; q->suppress = 0; q->count = 0; q->avgHigh = 0; q->avgLow = 0;
}
static float sq_decaying_average(float old, float input, float weight) {  /* :40-45: the arithmetic is double (1.0 / weight) */
    if (weight <= 1) return input;
    return (float)(input * (1.0 / weight) + old * (1.0 - (1.0 / weight)));
}
float fmo_squelch_noise(fmo_squelch *q, float soundSample) {             /* do_noise_squelch :47-87 */
    const float val_1 = fabsf(fmo_iir_pass(&q->high, soundSample));
    const float val_2 = fabsf(fmo_iir_pass(&q->low, soundSample));
    q->avgHigh = sq_decaying_average(q->avgHigh, val_1, (float)(q->rate / 100));
    q->avgLow = sq_decaying_average(q->avgLow, val_2, (float)(q->rate / 100));
    if (++q->count >= q->hold) {
        q->count = 0;
        if (q->noiseThr < 0.001f) q->suppress = 1;                                        /* SQUELCH_HYSTERESIS_NSQ */
        else if (q->avgHigh < q->avgLow * q->noiseThr - 0.001f) q->suppress = 0;
        else if (q->avgHigh >= q->avgLow * q->noiseThr + 0.001f) q->suppress = 1;
    }
    return q->suppress ? soundSample * 0.000f : soundSample;                              /* LEVELREDUCTIONFACTOR */
}
float fmo_squelch_level(fmo_squelch *q, float soundSample, float carrierLevel) {          /* do_level_squelch :89-113 */
    if (++q->count >= q->hold) {
        q->count = 0;
        if (carrierLevel < q->levelThr - 0.000f) q->suppress = 1;                         /* SQUELCH_HYSTERESIS_LSQ */
        else if (carrierLevel >= q->levelThr + 0.000f) q->suppress = 0;
    }
    return q->suppress ? soundSample * 0.000f : soundSample;
}
fmo_squelch *fmo_squelch_new(int32_t threshold, int32_t keyFrequency, int32_t bufsize, int32_t sampleRate) {
    fmo_squelch *q = (fmo_squelch *)calloc(1, sizeof(fmo_squelch));
    fmo_squelch_init(q, threshold, keyFrequency, bufsize, sampleRate);
    return q;
}
void fmo_squelch_free(fmo_squelch *q) { free(q); }
int fmo_squelch_active(const fmo_squelch *q) { return q->suppress; }
/* n samples through do_noise_squelch (carrier == NULL) or do_level_squelch; flags[i] = getSquelchActive() after sample i */
void fmo_squelch_run(fmo_squelch *q, const float *in, const float *carrier, float *out, uint8_t *flags, long n) {
    for (long i = 0; i < n; i++) {
        out[i] = carrier ? fmo_squelch_level(q, in[i], carrier[i]) : fmo_squelch_noise(q, in[i]);
        if (flags) flags[i] = (uint8_t)q->suppress;
    }
}

struct fmo_chain {
    fmo_config cfg;           /* live settings */
    /* members of fmProcessor (fm-processor.h:157-280) */
    int32_t *lo_dummy;
    c32 *loTable; int32_t LOPhase;
    fmo_sincos *sincos;
    fmo_decim *band1, *band2, *rdsDecim;
    fmo_fftfilter *audioFilter, *inputFilter, *rdsBand, *rdsHilbert;
    fmo_pilot *pilot; fmo_pss *pss;
    fmo_demod *demod;
    float *rdsPhaseBuffer; int rdsPhaseIndex;
    conv2 cv2; int cv2On;                                                /* theConverter (fm-processor.cpp:89-91): audioRate != workingRate */
    fmo_squelch sq; int sqOldValue;                                      /* mySquelch, oldSquelchValue (fm-processor.cpp:87,195) */
    int newAudioFilter, inputFilterOn, newInputFilter, audioFilterActive;
    int32_t lowPassFrequency, fmBandwidth;
    float Lgain, Rgain, pilotDelayPSS, deemphAlpha, volumeFactor, panorama, leftChannel, rightChannel;
    c32 lastAudioSample, RfDC;
    int32_t suppressMax, suppressCnt;
    int32_t peakCnt, peakMax; float absPeakL, absPeakR, peakLdb, peakRdb;
    c32 *delayBuf; uint32_t delaySize, delayIdx;          /* DelayLine<DSPCOMPLEX> fm-processor.h:54-75 */
    float *peakEv; long peakEvN, peakEvCap;               /* what showPeakLevel was emitted with */
    struct { uint32_t periodCounter, remain; float curPhase, phaseIncr; } tt;   /* TestTone fm-processor.h:241-249 */
    int32_t myCount;
    fmo_meta meta;
    resampler rs; c32 rsIn[192]; int rsInp;
    rds2 rds; rds1 rdsA; rds3 rdsC; uint8_t *rdsBits; long rdsBitCount, rdsBitCap;
    /* block intake */
    c32 *pending; long npending;
    tapbuf taps[FMO_TAP_COUNT];
    int64_t fmCount, pcmCount;
    uint32_t noiseState;                                  /* fmo_config::testFilterNoise */
};

void fmo_config_defaults(fmo_config *c) {
    memset(c, 0, sizeof(*c));
    c->inputRate = 2304000; c->fmRate = 192000; c->workingRate = 48000; c->audioRate = 48000;
    c->fmMode = 0; c->soundSelector = 0; c->decoder = FMO_DEC_MIXED;
    c->inputFilterBw = 165000;    /* radio.cpp:2099 */
    c->lfCutoff = 15000;          /* radio.cpp:2135 */
    c->deemphasis = 50;           /* radio.cpp:2129-2130 */
    c->volumeDb = -6.0f;          /* radio.cpp:579,1502-1505 : -12 half-dB */
    c->useCtorVolume = 0;
    c->balance = 0; c->panorama = 100; c->attL = 1; c->attR = 1; c->loFrequency = 0;
    c->dcRemove = 1; c->autoMono = 1; c->pssActive = 1; c->rdsMode = 0; c->squelchMode = 0; c->squelchValue = 0;
}

static void delay_set_steps(fmo_chain *ch, uint32_t steps);
static void apply_settings(fmo_chain *ch, const fmo_config *c, int initial) {
    const fmo_config old = ch->cfg;
    ch->cfg = *c;
    ch->cfg.touchInputFilter = 0; ch->cfg.touchLfCutoff = 0;
    /* setBandwidth fm-processor.cpp:232-239 */
    if (initial || c->inputFilterBw != old.inputFilterBw || c->touchInputFilter) {
        if (c->inputFilterBw <= 0) ch->inputFilterOn = 0;
        else { ch->fmBandwidth = c->inputFilterBw; ch->newInputFilter = 1; }
    }
    /* setlfcutoff :762-770 */
    if (initial || c->lfCutoff != old.lfCutoff || c->touchLfCutoff) {
        if (c->lfCutoff > 0) { ch->lowPassFrequency = c->lfCutoff; ch->newAudioFilter = 1; }
        else ch->audioFilterActive = 0;
    }
    /* setDeemphasis :291-297 : Tau is float */
    if (c->deemphasis >= 1 && (initial || c->deemphasis != old.deemphasis)) {
        float Tau = (float)(1000000.0 / c->deemphasis);
        ch->deemphAlpha = (float)(1.0 / ((double)((float)ch->cfg.fmRate / Tau) + 1.0));
    }
    /* setVolume :299-301 : std::pow(10.0f, dB/20.0f) -> powf */
    if (c->useCtorVolume) ch->volumeFactor = 0.5f;
    else ch->volumeFactor = powf(10.0f, c->volumeDb / 20.0f);
    /* setSoundBalance :282-286 */
    ch->leftChannel = (c->balance > 0 ? (float)((100 - c->balance) / 100.0) : 1.0f);
    ch->rightChannel = (c->balance < 0 ? (float)((100 + c->balance) / 100.0) : 1.0f);
    /* setStereoPanorama :277-280 */
    ch->panorama = (float)(int16_t)c->panorama / 100.0f;
    ch->Lgain = c->attL; ch->Rgain = c->attR;
    /* setDCRemove :922-925 zeroes RfDC whenever called; we call it only on change */
    if (!initial && c->dcRemove != old.dcRemove) ch->RfDC = C(0, 0);
    /* setDispDelay :935-937 */
    if ((initial && c->dispDelay > 0) || (!initial && c->dispDelay != old.dispDelay)) delay_set_steps(ch, (uint32_t)c->dispDelay);
    /* setPSSMode / setAutoMonoMode / setfmMode / setSoundMode / set_localOscillator / setTestTone: plain stores */
    fmo_demod_set_decoder(ch->demod, c->decoder);
}

fmo_chain *fmo_chain_new(const fmo_config *c) {
    /* fm-processor.cpp:48-198 */
    fmo_chain *ch = (fmo_chain *)calloc(1, sizeof(*ch));
    int32_t inputRate = c->inputRate, fmRate = c->fmRate;
    int32_t IRate = inputRate / 6;
    ch->cfg = *c;
    ch->loTable = (c32 *)malloc(sizeof(c32) * (size_t)inputRate);
    for (int32_t i = 0; i < inputRate; i++) ch->loTable[i] = fmo_lo_value(inputRate, i);
    ch->LOPhase = 0;
    ch->sincos = fmo_sincos_new(fmRate);
    ch->band1 = fmo_decim_new(4 * inputRate / IRate + 1, fmRate / 2, inputRate, inputRate / IRate);
    ch->band2 = fmo_decim_new(IRate / fmRate + 1, fmRate / 2, IRate, IRate / fmRate);
    ch->audioFilter = fmo_fftfilter_new(2 * 4096, 756);
    ch->inputFilter = fmo_fftfilter_new(2 * 32768, 251);
    /* OMEGA_PILOT ((float(19000)) / fmRate) * (2*M_PI) -> DSPFLOAT ; gain 10*(2*M_PI)/fmRate */
    float omega = (float)((double)((float)19000 / (float)fmRate) * (2 * M_PI));
    float gain = (float)(10 * (2 * M_PI) / fmRate);
    ch->pilot = fmo_pilot_new(fmRate, omega, gain, ch->sincos);
    ch->pss = fmo_pss_new(fmRate, 10.0f / (float)fmRate, ch->sincos);
    ch->rdsBand = fmo_fftfilter_new(FFT_SIZE_RDS, PILOTFILTER_SIZE);
    ch->rdsHilbert = fmo_fftfilter_new(FFT_SIZE_RDS, PILOTFILTER_SIZE);
    fmo_fftfilter_set_hilbert(ch->rdsHilbert);
    ch->demod = fmo_demod_new(fmRate);
    ch->Lgain = ch->Rgain = 1;
    ch->lowPassFrequency = 15000;
    ch->volumeFactor = 0.5f; ch->panorama = 1.0f;
    ch->suppressMax = c->workingRate / 2; ch->suppressCnt = ch->suppressMax;
    ch->RfDC = C(0, 0);
    ch->peakMax = c->workingRate / 50;
    delay_set_steps(ch, 0);
    ch->fmBandwidth = (int32_t)(0.95 * fmRate);
    /* ctor :148 : inputFilter.setLowPass(0.95*fmRate/2, inputRate) -- f64 -> int32 arg */
    fmo_fftfilter_set_lowpass(ch->inputFilter, (int32_t)(0.95 * fmRate / 2), inputRate);
    ch->leftChannel = ch->rightChannel = 1.0f;
    fmo_fftfilter_set_band(ch->rdsBand, 3 * 19000 - RDS_WIDTH / 2, 3 * 19000 + RDS_WIDTH / 2, fmRate);
    ch->rdsPhaseBuffer = (float *)calloc(RDS_SAMPLE_DELAY, sizeof(float));
    ch->lastAudioSample = C(0, 0);
    /* ctor :174 (differs from setDeemphasis) */
    ch->deemphAlpha = (float)(1.0 / (fmRate / (1000000.0 / 50.0 + 1)));
    ch->rdsDecim = fmo_decim_new(11, 24000 / 2, fmRate, fmRate / 24000);
    resampler_init(&ch->rs);
    ch->cv2On = ch->cfg.audioRate != ch->cfg.workingRate;
    if (ch->cv2On) conv2_init(&ch->cv2, ch->cfg.workingRate, ch->cfg.audioRate);
    rds2_init(&ch->rds, 24000); rds1_init(&ch->rdsA, 24000); rds3_init(&ch->rdsC, 24000);
    ch->rdsBitCap = 1 << 16; ch->rdsBits = (uint8_t *)malloc((size_t)ch->rdsBitCap);
    ch->pending = (c32 *)malloc(sizeof(c32) * BLOCK);
    ch->meta.peakLeftDb = ch->meta.peakRightDb = -40.0f;
    /* mySquelch (1, 70000, fmRate / 20, fmRate) fm-processor.cpp:87; squelchValue = oldSquelchValue = 0 :194-195 */
    fmo_squelch_init(&ch->sq, 1, 70000, ch->cfg.fmRate / 20, ch->cfg.fmRate); ch->sqOldValue = 0;
    apply_settings(ch, c, 1);
    return ch;
}

void fmo_chain_free(fmo_chain *ch) {
    if (!ch) return;
    free(ch->loTable); fmo_sincos_free(ch->sincos);
    fmo_decim_free(ch->band1); fmo_decim_free(ch->band2); fmo_decim_free(ch->rdsDecim);
    fmo_fftfilter_free(ch->audioFilter); fmo_fftfilter_free(ch->inputFilter);
    fmo_fftfilter_free(ch->rdsBand); fmo_fftfilter_free(ch->rdsHilbert);
    fmo_pilot_free(ch->pilot); fmo_pss_free(ch->pss); fmo_demod_free(ch->demod);
    free(ch->delayBuf); free(ch->peakEv); fmo_sincos_free(ch->rdsC.sc);
    free(ch->rdsPhaseBuffer); free(ch->rdsBits); free(ch->pending); free(ch);
}

void fmo_chain_configure(fmo_chain *ch, const fmo_config *c) { apply_settings(ch, c, 0); }

void fmo_chain_trigger_frequency_change(fmo_chain *ch) {
    /* fm-processor.cpp:849-860 (resetRds only clears the group decoder: host side) */
    ch->suppressCnt = ch->suppressMax;
    ch->pilotDelayPSS = 0;
    fmo_pss_reset(ch->pss);
}

void fmo_chain_set_tap(fmo_chain *ch, int tap, float *buf, long cap) {
    ch->taps[tap].buf = buf; ch->taps[tap].cap = cap; ch->taps[tap].n = 0;
}
long fmo_chain_tap_count(const fmo_chain *ch, int tap) { return ch->taps[tap].n; }
static inline void tap1(fmo_chain *ch, int t, float v) {
    tapbuf *b = &ch->taps[t];
    if (b->buf && b->n + 1 <= b->cap) b->buf[b->n++] = v;
}
static inline void tap2(fmo_chain *ch, int t, c32 v) {
    tapbuf *b = &ch->taps[t];
    if (b->buf && b->n + 2 <= b->cap) { b->buf[b->n++] = v.re; b->buf[b->n++] = v.im; }
}

static void process_signal_with_rds(fmo_chain *ch, float demod, c32 *audioOut, c32 *rdsOut) {
    /* fm-processor.cpp:689-759 */
    float currentPilotPhase = fmo_pilot_phase(ch->pilot, 5 * demod);
    int pilotLocked = fmo_pilot_locked(ch->pilot);
    if (!pilotLocked) { ch->pilotDelayPSS = 0; fmo_pss_reset(ch->pss); }
    if (ch->cfg.fmMode != 2 && (pilotLocked || !ch->cfg.autoMono)) {
        /* 2 * (f32 + f64 + 0) - f32 : f64 -> f32 */
        float ph = (float)(2 * ((double)currentPilotPhase + M_PI_4 + 0) - (double)ch->pilotDelayPSS);
        if (ph < -2 * M_PI) ph = (float)((double)ph + 4 * M_PI);
        ph = (float)fmod((double)ph, 2 * M_PI);
        ch->pilotDelayPSS = ch->cfg.pssActive ? fmo_pss_process(ch->pss, demod, ph) : 0;
        /* 2.0 * (getSin|getCos) * demod : f64 -> f32 */
        float lut = (ch->cfg.soundSelector == 6) ? fmo_sincos_sin(ch->sincos, ph)
                                                 : fmo_sincos_cos(ch->sincos, ph);
        float LRDiff = (float)(2.0 * (double)lut * (double)demod);
        *audioOut = C(demod, LRDiff);
    } else {
        *audioOut = C(demod, 0);
    }
    tap1(ch, FMO_TAP_PILOT, currentPilotPhase);
    tap1(ch, FMO_TAP_PSS, ch->pilotDelayPSS);
    if (ch->cfg.rdsMode != 0) {
        float bp = fmo_fftfilter_pass_r(ch->rdsBand, demod);
        c32 hil = fmo_fftfilter_pass_c(ch->rdsHilbert, C(bp, 0));
        /* thePhase is float -> cos/sin are the float overloads */
        float thePhase = 3 * (ch->rdsPhaseBuffer[ch->rdsPhaseIndex] + 0);
        ch->rdsPhaseBuffer[ch->rdsPhaseIndex] = currentPilotPhase;
        ch->rdsPhaseIndex = (ch->rdsPhaseIndex + 1) % RDS_SAMPLE_DELAY;
        c32 osc = C(cosf(thePhase), -sinf(thePhase));
        *rdsOut = cmul(osc, hil);
    }
}

static void delay_set_steps(fmo_chain *ch, uint32_t steps) {
    /* DelayLine::set_delay_steps fm-processor.h:60-63: vector::resize keeps what is there, new entries = (-40, -40) */
    ch->delayBuf = (c32 *)realloc(ch->delayBuf, sizeof(c32) * (steps + 1));
    for (uint32_t i = ch->delaySize; i < steps + 1; i++) ch->delayBuf[i] = C(-40.0f, -40.0f);
    ch->delaySize = steps + 1; ch->delayIdx = 0;
}

static void insert_test_tone(fmo_chain *ch, c32 *s) {
    /* fm-processor.cpp:800-823; `sin` of a float under `using namespace std` (fm-constants.h:66) = sinf */
    const float toneFreqHz = 1000.0f, level = 0.9f;
    const float TimePeriod = 2.0f, SignalDuration = 0.025f;           /* fm-processor.h:243-244 */
    if (!ch->cfg.testTone) return;
    *s = cscale(*s, 1.0f - level);
    if (ch->tt.remain > 0) {
        ch->tt.remain--;
        ch->tt.curPhase += ch->tt.phaseIncr;
        ch->tt.curPhase = fmo_pi_constrain(ch->tt.curPhase);
        const float smpl = sinf(ch->tt.curPhase);
        s->re = s->re + level * smpl; s->im = s->im + level * smpl;
    } else if ((float)(++ch->tt.periodCounter) > (float)ch->cfg.workingRate * TimePeriod) {
        ch->tt.periodCounter = 0;
        ch->tt.remain = (uint32_t)((float)ch->cfg.workingRate * SignalDuration);
        ch->tt.curPhase = 0.0f;
        ch->tt.phaseIncr = (float)(2 * M_PI / ch->cfg.workingRate * toneFreqHz);
    }
}

void fmo_test_tone_burst(int32_t workingRate, float *dst, long n) {
    float ph = 0.0f; const float inc = (float)(2 * M_PI / workingRate * 1000.0f);
    for (long i = 0; i < n; i++) { ph += inc; ph = fmo_pi_constrain(ph); dst[i] = sinf(ph); }
}

static void evaluate_peak(fmo_chain *ch, c32 s) {
    /* fm-processor.cpp:772-798 */
    float aL = fabsf(s.re), aR = fabsf(s.im);
    if (aL > ch->absPeakL) ch->absPeakL = aL;
    if (aR > ch->absPeakR) ch->absPeakR = aR;
    ch->peakCnt++;
    if (ch->peakCnt > ch->peakMax) {
        ch->peakCnt = 0;
        ch->peakLdb = (ch->absPeakL > 0.0f ? 20.0f * log10f(ch->absPeakL) : -40.0f);
        ch->peakRdb = (ch->absPeakR > 0.0f ? 20.0f * log10f(ch->absPeakR) : -40.0f);
        /* delayLine.get_set_value fm-processor.h:65-69 */
        ch->delayBuf[ch->delayIdx] = C(ch->peakLdb, ch->peakRdb);
        ch->delayIdx = (ch->delayIdx + 1) % ch->delaySize;
        const c32 delayed = ch->delayBuf[ch->delayIdx];
        ch->meta.peakLeftDb = delayed.re; ch->meta.peakRightDb = delayed.im;
        if (ch->peakEvN == ch->peakEvCap) {
            ch->peakEvCap = ch->peakEvCap ? 2 * ch->peakEvCap : 256;
            ch->peakEv = (float *)realloc(ch->peakEv, sizeof(float) * 2 * (size_t)ch->peakEvCap);
        }
        ch->peakEv[2 * ch->peakEvN] = delayed.re; ch->peakEv[2 * ch->peakEvN + 1] = delayed.im; ch->peakEvN++;
        ch->absPeakL = 0.0f; ch->absPeakR = 0.0f;
    }
}

static long process_block(fmo_chain *ch, c32 *data, int32_t amount, float *pcm, long cap, long nout) {
    /* one iteration of the while loop of fmProcessor::run (fm-processor.cpp:387-686) */
    const int32_t inputRate = ch->cfg.inputRate, fmRate = ch->cfg.fmRate;
    float rfDcAlpha = 1.0f / inputRate;
    if (ch->newInputFilter) {
        fmo_fftfilter_set_lowpass(ch->inputFilter, ch->fmBandwidth / 2, inputRate);
        ch->inputFilterOn = 1; ch->newInputFilter = 0;
    }
    if (ch->newAudioFilter) {
        fmo_fftfilter_set_lowpass(ch->audioFilter, ch->lowPassFrequency, fmRate);
        ch->audioFilterActive = 1; ch->newAudioFilter = 0;
    }
    if (ch->cfg.squelchValue != ch->sqOldValue) {           /* fm-processor.cpp:410-413 */
        fmo_squelch_set_level(&ch->sq, ch->cfg.squelchValue);
        ch->sqOldValue = ch->cfg.squelchValue;
    }
    if (ch->cfg.dcRemove) {
        for (int32_t i = 0; i < amount; i++) {
            /* :425 */
            ch->RfDC = cadd(cscale(csub(data[i], ch->RfDC), rfDcAlpha), ch->RfDC);
            const float lim = 0.01f;
            float r = ch->RfDC.re, q = ch->RfDC.im;
            if (r > +lim) r = +lim; else if (r < -lim) r = -lim;
            if (q > +lim) q = +lim; else if (q < -lim) q = -lim;
            data[i] = csub(data[i], C(r, q));
        }
    }
    for (int32_t i = 0; i < amount; i++) {
        c32 v = C(data[i].re * ch->Lgain, data[i].im * ch->Rgain);
        /* Oscillator::nextValue oscillator.cpp:49-58 */
        ch->LOPhase -= ch->cfg.loFrequency;
        if (ch->LOPhase < 0) ch->LOPhase += inputRate;
        else if (ch->LOPhase >= inputRate) ch->LOPhase -= inputRate;
        v = cmul(v, ch->loTable[ch->LOPhase]);
        if (ch->inputFilterOn) v = fmo_fftfilter_pass_c(ch->inputFilter, v);
        if (ch->cfg.testFilterNoise > 0.f && ch->inputFilterOn) {     /* test hook, see fm_oracle.h: off in every parity comparison */
            if (ch->noiseState == 0) ch->noiseState = 0x9e3779b9u ^ (uint32_t)ch->cfg.testNoiseSeed;
            ch->noiseState = ch->noiseState * 1664525u + 1013904223u; const float a = (float)(int32_t)ch->noiseState * (1.0f / 2147483648.0f);
            ch->noiseState = ch->noiseState * 1664525u + 1013904223u; const float b = (float)(int32_t)ch->noiseState * (1.0f / 2147483648.0f);
            v = C(v.re + 1.7320508f * ch->cfg.testFilterNoise * a, v.im + 1.7320508f * ch->cfg.testFilterNoise * b);
        }
        if (inputRate / fmRate > 1) {
            if (!fmo_decim_pass(ch->band1, v, &v)) continue;
            if (!fmo_decim_pass(ch->band2, v, &v)) continue;
        }
        tap2(ch, FMO_TAP_FM_IQ, v);
        float demod = fmo_demod_demodulate(ch->demod, v);
        if (ch->cfg.squelchMode == 1) demod = fmo_squelch_noise(&ch->sq, demod);                                   /* fm-processor.cpp:499-509 */
        else if (ch->cfg.squelchMode == 2) demod = fmo_squelch_level(&ch->sq, demod, fmo_demod_carrier(ch->demod));
        tap1(ch, FMO_TAP_DEMOD, demod);
        c32 audio, rdsData = C(0, 0);
        process_signal_with_rds(ch, demod, &audio, &rdsData);
        tap2(ch, FMO_TAP_LRRAW, audio);
        const float sumLR = audio.re, diffLR = audio.im;
        const float diffW = diffLR * (ch->cfg.fmMode == 1 ? ch->panorama : 1.0f);
        const float left = sumLR + diffW, right = sumLR - diffW;
        switch (ch->cfg.soundSelector) {
        default:
        case 0: audio = C(left, right); break;
        case 1: audio = C(right, left); break;
        case 2: audio = C(left, left); break;
        case 3: audio = C(right, right); break;
        case 4: audio = C(sumLR, sumLR); break;
        case 5: case 6: audio = C(diffW, diffW); break;
        }
        if (ch->cfg.rdsMode != 0) {
            c32 rdsSample;
            if (fmo_decim_pass(ch->rdsDecim, rdsData, &rdsSample)) {
                tap2(ch, FMO_TAP_RDS_IQ, rdsSample);
                if (ch->cfg.rdsMode >= 1 && ch->cfg.rdsMode <= 3) {
                    c32 mag; uint8_t bit;
                    if (ch->cfg.rdsMode == 2 ? rds2_decode(&ch->rds, rdsSample, &mag, &bit)
                        : ch->cfg.rdsMode == 1 ? rds1_decode(&ch->rdsA, rdsSample, &mag, &bit) : rds3_decode(&ch->rdsC, rdsSample, &mag, &bit)) {
                        if (ch->rdsBitCount >= ch->rdsBitCap) {
                            ch->rdsBitCap *= 2;
                            ch->rdsBits = (uint8_t *)realloc(ch->rdsBits, (size_t)ch->rdsBitCap);
                        }
                        ch->rdsBits[ch->rdsBitCount++] = bit;
                    }
                }
            }
        }
        if (ch->audioFilterActive) audio = fmo_fftfilter_pass_c(ch->audioFilter, audio);
        /* de-emphasis :594-595 */
        audio = ch->lastAudioSample =
            cadd(cscale(csub(audio, ch->lastAudioSample), ch->deemphAlpha), ch->lastAudioSample);
        /* audioGainCorrection :303-306 : (volumeFactor*leftChannel)*re */
        audio = C(ch->volumeFactor * ch->leftChannel * audio.re,
                  ch->volumeFactor * ch->rightChannel * audio.im);
        tap2(ch, FMO_TAP_PRE_RS, audio);
        ch->fmCount++;
        /* newConverter::convert newconverter.cpp:55-80 with the fmx resampler behind it */
        ch->rsIn[ch->rsInp++] = audio;
        if (ch->rsInp >= fmRate / 1000) {
            for (int k = 0; k < ch->rsInp; k++) {
                c32 p;
                if (!resampler_push(&ch->rs, ch->rsIn[k], &p)) continue;
                /* :636-647 */
                if (ch->suppressCnt > 0) {
                    p = cscale(p, ((float)ch->suppressMax - (float)ch->suppressCnt) / (float)ch->suppressMax);
                    --ch->suppressCnt;
                }
                insert_test_tone(ch, &p);
                evaluate_peak(ch, p);
                if (!ch->cv2On) {                                /* sendSampletoOutput :825-838 */
                    if (nout < cap) { pcm[2 * nout] = p.re; pcm[2 * nout + 1] = p.im; }
                    nout++;
                } else {
                    c32 o2[FMO_CONV2_MAXP / 16 + 8];
                    const int n2 = conv2_push(&ch->cv2, p, o2, (int)(sizeof(o2) / sizeof(o2[0])));
                    for (int i2 = 0; i2 < n2; i2++) { if (nout < cap) { pcm[2 * nout] = o2[i2].re; pcm[2 * nout + 1] = o2[i2].im; } nout++; }
                }
                ch->pcmCount++;
            }
            ch->rsInp = 0;
        }
        if (++ch->myCount > (fmRate >> 1)) {
            /* :662-684 */
            float strength = 0; int locked = 0;
            if (ch->cfg.fmMode != 2) { strength = fmo_pilot_strength(ch->pilot); locked = fmo_pilot_locked(ch->pilot); }
            ch->meta.pilotLocked = locked; ch->meta.pilotLockStrength = strength;
            ch->meta.dcValRf = ch->cfg.dcRemove ? 20 * log10f(cabs32(ch->RfDC) + 1.0f / 32768) : (float)-99.99;
            ch->meta.dcValIf = fmo_demod_dc(ch->demod);
            ch->meta.pssPhaseShiftDegree = (float)((double)ch->pilotDelayPSS / M_PI * 180.0f);
            ch->meta.pssPhaseChange = fmo_pss_mean_error(ch->pss) * 1000;
            ch->meta.pssState = (ch->cfg.pssActive && locked) ? (fmo_pss_minimized(ch->pss) ? 2 : 1) : 0;
            ch->myCount = 0;
        }
    }
    return nout;
}

long fmo_chain_process(fmo_chain *ch, const float *iq, long n, float *pcm, long cap) {
    /* fm-processor.cpp:387-417 : the reference only ever consumes whole 16384 blocks */
    long nout = 0;
    const c32 *in = (const c32 *)iq;
    long i = 0;
    while (i < n) {
        long take = BLOCK - ch->npending;
        if (take > n - i) take = n - i;
        memcpy(ch->pending + ch->npending, in + i, sizeof(c32) * (size_t)take);
        ch->npending += take; i += take;
        if (ch->npending == BLOCK) {
            nout = process_block(ch, ch->pending, BLOCK, pcm, cap, nout);
            ch->npending = 0;
        }
    }
    return nout;
}

void fmo_chain_meta(const fmo_chain *ch, fmo_meta *m) {
    *m = ch->meta; m->fmSamples = ch->fmCount; m->pcmFrames = ch->pcmCount;
    m->squelchActive = ch->sq.suppress; m->pad_ = 0;
}
long fmo_chain_peaks(const fmo_chain *ch, float *lr_db, long cap) {
    long n = ch->peakEvN < cap ? ch->peakEvN : cap;
    if (lr_db && n > 0) memcpy(lr_db, ch->peakEv, sizeof(float) * 2 * (size_t)n);
    return ch->peakEvN;
}
long fmo_chain_rds_bits(const fmo_chain *ch, uint8_t *bits, long cap) {
    long n = ch->rdsBitCount < cap ? ch->rdsBitCount : cap;
    if (bits && n > 0) memcpy(bits, ch->rdsBits, (size_t)n);
    return ch->rdsBitCount;
}


/* ------------------------------------------------------------------ batch runners (tests) */
void fmo_sincos_eval(const fmo_sincos *t, const float *phase, long n, float *s, float *c, float *cplx) {
    for (long i = 0; i < n; i++) {
        s[i] = fmo_sincos_sin(t, phase[i]); c[i] = fmo_sincos_cos(t, phase[i]);
        c32 z = fmo_sincos_complex(t, phase[i]); cplx[2 * i] = z.re; cplx[2 * i + 1] = z.im;
    }
}
void fmo_atan2_eval(const float *y, const float *x, long n, float *out) {
    fmo_atan *a = fmo_atan_new();
    for (long i = 0; i < n; i++) out[i] = fmo_atan2(a, y[i], x[i]);
    fmo_atan_free(a);
}
void fmo_pi_constrain_eval(const float *in, long n, float *out) { for (long i = 0; i < n; i++) out[i] = fmo_pi_constrain(in[i]); }
void fmo_pll_run(int32_t rate, float freq, float lo, float hi, float bw, const float *sig, long n, float *incr) {
    fmo_sincos *tab = fmo_sincos_new(rate); fmo_atan *at = fmo_atan_new();
    fmo_pll *p = fmo_pll_new(rate, freq, lo, hi, bw, tab, at);
    for (long i = 0; i < n; i++) { fmo_pll_do(p, C(sig[2 * i], sig[2 * i + 1])); incr[i] = p->phaseIncr; }
    fmo_pll_free(p); fmo_sincos_free(tab); fmo_atan_free(at);
}
void fmo_demod_run(int32_t rate, int decoder, const float *z, long n, float *out, float *dc, float *carrier) {
    fmo_demod *d = fmo_demod_new(rate);
    fmo_demod_set_decoder(d, decoder);
    for (long i = 0; i < n; i++) out[i] = fmo_demod_demodulate(d, C(z[2 * i], z[2 * i + 1]));
    if (dc) *dc = d->fm_afc;
    if (carrier) *carrier = d->am_carr_ampl;
    fmo_demod_free(d);
}
void fmo_pilot_run(int32_t rate, float omega, float gain, const float *pilot, long n,
                   float *phase, uint8_t *locked, float *strength) {
    fmo_sincos *tab = fmo_sincos_new(rate);
    fmo_pilot *p = fmo_pilot_new(rate, omega, gain, tab);
    for (long i = 0; i < n; i++) {
        phase[i] = fmo_pilot_phase(p, pilot[i]); locked[i] = (uint8_t)p->locked; strength[i] = p->lock;
    }
    fmo_pilot_free(p); fmo_sincos_free(tab);
}
void fmo_pss_run(int32_t rate, float alpha, const float *mux, const float *ph, long n, float *out,
                 const uint8_t *reset_before) {
    fmo_sincos *tab = fmo_sincos_new(rate);
    fmo_pss *p = fmo_pss_new(rate, alpha, tab);
    for (long i = 0; i < n; i++) {
        if (reset_before && reset_before[i]) fmo_pss_reset(p);
        out[i] = fmo_pss_process(p, mux[i], ph[i]);
    }
    fmo_pss_free(p); fmo_sincos_free(tab);
}
void fmo_agc_run(float rate, float ref, float gain, const float *in, long n, float *out) {
    fmo_agc a = { rate, ref, gain };
    for (long i = 0; i < n; i++) { c32 o = fmo_agc_process(&a, C(in[2 * i], in[2 * i + 1])); out[2 * i] = o.re; out[2 * i + 1] = o.im; }
}
void fmo_costas_run(float sr, float alpha, float beta, float lim, const float *in, long n, float *out) {
    fmo_costas c; fmo_costas_init(&c, sr, alpha, beta, lim);
    for (long i = 0; i < n; i++) { c32 o = fmo_costas_process(&c, C(in[2 * i], in[2 * i + 1])); out[2 * i] = o.re; out[2 * i + 1] = o.im; }
}
/* pilot PLL constants as fmProcessor's ctor passes them (fm-processor.cpp:78-82) */
void fmo_pilot_constants(int32_t fmRate, float *omega, float *gain, float *pssAlpha) {
    *omega = (float)((double)((float)19000 / (float)fmRate) * (2 * M_PI));
    *gain = (float)(10 * (2 * M_PI) / fmRate);
    *pssAlpha = 10.0f / (float)fmRate;
}

/* ------------------------------------------------------------------ synthetic IQ */
struct fmo_siggen {
    fmo_siggen_config c;
    double phase;         /* FM phase accumulator (rad, wrapped) */
    uint64_t n;           /* sample index */
    uint64_t rng;
    /* RDS bit source */
    uint64_t bitrng; uint8_t *bits; long nbits, bitcap; int diffState; long lastk; int cur;
    uint8_t *fixed; long nfixed;
};
static uint64_t xorshift64s(uint64_t *s) {
    uint64_t x = *s; x ^= x >> 12; x ^= x << 25; x ^= x >> 27; *s = x;
    return x * 0x2545F4914F6CDD1DULL;
}
static double gauss(uint64_t *s) {
    /* Box-Muller on two 53-bit uniforms */
    double u1 = ((xorshift64s(s) >> 11) + 1.0) / 9007199254740993.0;
    double u2 = (xorshift64s(s) >> 11) / 9007199254740992.0;
    return sqrt(-2.0 * log(u1)) * cos(2 * M_PI * u2);
}
fmo_siggen *fmo_siggen_new(const fmo_siggen_config *c) {
    fmo_siggen *g = (fmo_siggen *)calloc(1, sizeof(*g));
    g->c = *c; g->rng = c->noiseSeed ? c->noiseSeed : 1; g->bitrng = c->rdsBitsSeed ? c->rdsBitsSeed : 0x9E3779B97F4A7C15ULL;
    g->bitcap = 4096; g->bits = (uint8_t *)malloc((size_t)g->bitcap);
    g->lastk = -1; g->cur = 0; g->diffState = 0;
    return g;
}
void fmo_siggen_free(fmo_siggen *g) { if (g) { free(g->bits); free(g->fixed); free(g); } }
long fmo_siggen_rds_bits(const fmo_siggen *g, uint8_t *bits, long cap) {
    long n = g->nbits < cap ? g->nbits : cap;
    if (bits && n > 0) memcpy(bits, g->bits, (size_t)n);
    return g->nbits;
}
/* test hook: send these data bits (e.g. encoded RDS groups), cyclically, instead of pseudo-random ones */
void fmo_siggen_set_rds_bits(fmo_siggen *g, const uint8_t *bits, long n) {
    free(g->fixed); g->fixed = NULL; g->nfixed = 0;
    if (bits && n > 0) { g->fixed = (uint8_t *)malloc((size_t)n); memcpy(g->fixed, bits, (size_t)n); g->nfixed = n; }
}
static int siggen_bit(fmo_siggen *g, long k) {
    if (g->nfixed > 0) {
        while (g->nbits <= k) {
            if (g->nbits >= g->bitcap) { g->bitcap *= 2; g->bits = (uint8_t *)realloc(g->bits, (size_t)g->bitcap); }
            g->bits[g->nbits] = g->fixed[g->nbits % g->nfixed]; g->nbits++;
        }
        return g->bits[k];
    }
    while (g->nbits <= k) {
        if (g->nbits >= g->bitcap) { g->bitcap *= 2; g->bits = (uint8_t *)realloc(g->bits, (size_t)g->bitcap); }
        g->bits[g->nbits++] = (uint8_t)((xorshift64s(&g->bitrng) >> 40) & 1);
    }
    return g->bits[k];
}
void fmo_siggen_run(fmo_siggen *g, float *iq, long n) {
    const fmo_siggen_config *c = &g->c;
    const double fs = c->inputRate;
    for (long i = 0; i < n; i++) {
        double t = (double)g->n / fs;
        double L = c->leftAmp * sin(2 * M_PI * c->leftHz * t);
        double R = c->rightAmp * sin(2 * M_PI * c->rightHz * t);
        double mpx;
        if (c->stereo) {
            double p19 = 2 * M_PI * 19000.0 * t;
            mpx = 0.45 * (L + R) + c->pilotLevel * sin(p19) + 0.45 * (L - R) * sin(2 * p19);
            if (c->rds) {
                /* differentially encoded biphase symbols on a 57 kHz DSB-SC carrier,
                   cosine-shaped half-bit pulses; carrier in quadrature with the 3rd pilot harmonic */
                double bt = t * 1187.5;
                long k = (long)bt; double frac = bt - (double)k;
                /* differential encoding is over the data bits; symbol = +1/-1 then biphase */
                if (k != g->lastk) {
                    g->diffState ^= siggen_bit(g, k); g->cur = g->diffState; g->lastk = k;
                }
                double sym = g->cur ? 1.0 : -1.0;
                double shape = sin(2 * M_PI * frac);   /* biphase: +half then -half */
                mpx += c->rdsLevel * sym * shape * cos(3 * p19);
            }
        } else {
            mpx = 0.9 * 0.5 * (L + R);
        }
        g->phase += 2 * M_PI * (c->deviationHz * mpx + c->offsetHz) / fs;
        if (g->phase > M_PI) g->phase -= 2 * M_PI; else if (g->phase < -M_PI) g->phase += 2 * M_PI;
        double I = c->carrierAmp * cos(g->phase) + c->dcI;
        double Q = c->carrierAmp * sin(g->phase) + c->dcQ;
        if (c->noiseSeed) { I += c->noiseSigma * gauss(&g->rng); Q += c->noiseSigma * gauss(&g->rng); }
        iq[2 * i] = (float)I; iq[2 * i + 1] = (float)Q;
        g->n++;
    }
}
