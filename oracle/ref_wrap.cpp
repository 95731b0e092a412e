// ref_wrap.cpp -- extern "C" handles around the REFERENCE's own leaf classes.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no DSP arithmetic of its own for the leaf
// wrappers: it instantiates the classes declared in /root/reference/includes/{various,fm}/*.h and
// forwards calls, so that tests can pin oracle/fm_oracle.c bit-for-bit against the reference.
// It is compiled together with the reference sources *where they lie* (see oracle/Makefile) into
// oracle/_ref/libfmref.so; no reference source is copied into this repository.
//
// Classes wrapped (all Qt-free): LowPassFIR, BandPassFIR, DecimatingFIR (fir-filters.h),
// fftFilter, fftFilterHilbert (fft-filters.h), Fft_transform (fft-complex.h), SinCos, Oscillator,
// compAtan, pllC, pilotRecovery, PerfectStereoSeparation, ShapingFilter, AGC, Costas, LowPassIIR, HighPassIIR, BandPassIIR,
// PI_Constrain (fm-constants.h), RDSGroup (rds-group.h) and the PTY / EBU tables of src/rds/ebu-codetables.c.  With -DFMREF_WITH_QT also fm_Demodulator (needs QString from
// the image's conda QtCore; built only when those headers exist).
//
// NOT built: src/fm/fm-processor.cpp (needs portaudio.h, sndfile.h, samplerate.h, qwt and
// moc/uic generated code, none of which exist in this image) -- the run() glue is therefore
// exercised through ref_chain_* below, which wires the reference leaf objects in the order of
// fm-processor.cpp:461-476,497,515-525,589-595,630 and process_signal_with_rds :689-730.
#include <complex>
#include <vector>
#include <cstring>
#include <cstdint>

#include "fm-constants.h"
#include "fir-filters.h"
#include "fft-filters.h"
#include "fft-complex.h"
#include "sincos.h"
#include "oscillator.h"
#include "Xtan2.h"
#include "pllC.h"
#include "pilot-recover.h"
#include "stereo-separation.h"
#include "shaping_filter.h"
#include "agc.h"
#include "costas.h"
#include "iir-filters.h"
#include "rds-group.h"          // RDSGroup (src/rds/rds-group.cpp, compiled with this file)
#include "ebu-codetables.c"     // pty_table, EBU_E1, mapEBUtoUnicode: pulled in by #include, as rds-groupdecoder.cpp:43 does
#ifdef FMREF_WITH_QT
#include "fm-demodulator.h"
#include "squelchClass.h"       // QObject + moc (oracle/Makefile runs the image's moc on the reference's header, output in _ref/)
#endif

typedef std::complex<float> cf;

extern "C" {

int ref_has_qt(void) {
#ifdef FMREF_WITH_QT
    return 1;
#else
    return 0;
#endif
}

// ---- kernels ----
void ref_lowpass_kernel(int N, int32_t Fc, int32_t fs, float *h) {
    LowPassFIR f(N, Fc, fs);
    for (int i = 0; i < N; i++) h[i] = real(f.getKernel()[i]);
}
void ref_decim_kernel(int N, int32_t low, int32_t fs, float *k /*2N*/) {
    DecimatingFIR f(N, low, fs, 1);
    std::memcpy(k, f.getKernel(), sizeof(cf) * N);
}
void ref_bandpass_kernel(int N, int32_t low, int32_t high, int32_t fs, float *k /*2N*/) {
    BandPassFIR f(N, low, high, fs);
    std::memcpy(k, f.getKernel(), sizeof(cf) * N);
}
int ref_rrc_kernel(double gain, double fs, double symrate, double alpha, int ntaps, float *taps) {
    std::vector<float> v = ShapingFilter().root_raised_cosine(gain, fs, symrate, alpha, ntaps);
    std::memcpy(taps, v.data(), sizeof(float) * v.size());
    return (int)v.size();
}

// ---- FFT ----
int ref_fft(float *vec, long n, int inverse) { return Fft_transform((cf *)vec, (size_t)n, inverse != 0) ? 1 : 0; }

// ---- overlap-add filter ----
void *ref_fftfilter_new(int size, int degree) { return new fftFilter(size, degree); }
void *ref_fftfilter_hilbert_new(int size, int degree) { return new fftFilterHilbert(size, degree); }
void ref_fftfilter_free(void *p) { delete (fftFilter *)p; }
void ref_fftfilter_hilbert_free(void *p) { delete (fftFilterHilbert *)p; }
void ref_fftfilter_set_lowpass(void *p, int32_t low, int32_t rate) { ((fftFilter *)p)->setLowPass(low, rate); }
void ref_fftfilter_set_band(void *p, int32_t lo, int32_t hi, int32_t rate) { ((fftFilter *)p)->setBand(lo, hi, rate); }
void ref_fftfilter_run_c(void *p, const float *in, float *out, long n) {
    fftFilter *f = (fftFilter *)p;
    for (long i = 0; i < n; i++) { cf o = f->Pass(cf(in[2 * i], in[2 * i + 1])); out[2 * i] = real(o); out[2 * i + 1] = imag(o); }
}
void ref_fftfilter_run_r(void *p, const float *in, float *out, long n) {
    fftFilter *f = (fftFilter *)p;
    for (long i = 0; i < n; i++) out[i] = f->Pass(in[i]);
}
void ref_fftfilter_hilbert_run(void *p, const float *in, float *out, long n) {
    fftFilterHilbert *f = (fftFilterHilbert *)p;
    for (long i = 0; i < n; i++) { cf o = f->Pass(in[i]); out[2 * i] = real(o); out[2 * i + 1] = imag(o); }
}

// ---- decimating FIR ----
void *ref_decim_new(int N, int32_t low, int32_t fs, int D) { return new DecimatingFIR(N, low, fs, D); }
void ref_decim_free(void *p) { delete (DecimatingFIR *)p; }
long ref_decim_run(void *p, const float *in, long n, float *out) {
    DecimatingFIR *d = (DecimatingFIR *)p;
    long m = 0;
    for (long i = 0; i < n; i++) {
        cf o;
        if (d->Pass(cf(in[2 * i], in[2 * i + 1]), &o)) { out[2 * m] = real(o); out[2 * m + 1] = imag(o); m++; }
    }
    return m;
}

// ---- LUTs ----
void *ref_sincos_new(int32_t rate) { return new SinCos(rate); }
void ref_sincos_free(void *p) { delete (SinCos *)p; }
void ref_sincos_eval(void *p, const float *phase, long n, float *s, float *c, float *cplx /*2n*/) {
    SinCos *t = (SinCos *)p;
    for (long i = 0; i < n; i++) {
        s[i] = t->getSin(phase[i]); c[i] = t->getCos(phase[i]);
        cf z = t->getComplex(phase[i]); cplx[2 * i] = real(z); cplx[2 * i + 1] = imag(z);
    }
}
void ref_atan2_eval(const float *y, const float *x, long n, float *out) {
    static compAtan a;
    for (long i = 0; i < n; i++) out[i] = a.atan2(y[i], x[i]);
}
void ref_lo_run(int32_t rate, int32_t step, long n, float *out /*2n*/) {
    Oscillator o(rate);
    for (long i = 0; i < n; i++) { cf z = o.nextValue(step); out[2 * i] = real(z); out[2 * i + 1] = imag(z); }
}
void ref_lo_table(int32_t rate, const int32_t *idx, long n, float *out) {
    // Oscillator exposes only nextValue(step): walk with step = -1 is O(rate); use steps instead
    Oscillator o(rate);
    int32_t cur = 0;   // LOPhase
    for (long i = 0; i < n; i++) {
        int32_t step = cur - idx[i];             // LOPhase -= step  -> idx[i]
        if (step <= -rate) step += rate; if (step >= rate) step -= rate;
        cf z = o.nextValue(step);
        cur = idx[i];
        out[2 * i] = real(z); out[2 * i + 1] = imag(z);
    }
}
void ref_pi_constrain(const float *in, long n, float *out) { for (long i = 0; i < n; i++) out[i] = PI_Constrain(in[i]); }

// ---- pllC ----
void ref_pll_run(int32_t rate, float freq, float lo, float hi, float bw, const float *sig /*2n*/, long n,
                 float *incr) {
    SinCos tab(rate);
    pllC p(rate, freq, lo, hi, bw, &tab);
    for (long i = 0; i < n; i++) { p.do_pll(cf(sig[2 * i], sig[2 * i + 1])); incr[i] = p.getPhaseIncr(); }
}

// ---- pilot PLL ----
void ref_pilot_run(int32_t rate, float omega, float gain, const float *pilot, long n,
                   float *phase, uint8_t *locked, float *strength) {
    SinCos tab(rate);
    pilotRecovery p(rate, omega, gain, &tab);
    for (long i = 0; i < n; i++) {
        phase[i] = p.getPilotPhase(pilot[i]);
        locked[i] = p.isLocked() ? 1 : 0; strength[i] = p.getLockedStrength();
    }
}

// ---- PSS ----
void ref_pss_run(int32_t rate, float alpha, const float *mux, const float *ph, long n, float *out,
                 const uint8_t *reset_before /* may be null */) {
    SinCos tab(rate);
    PerfectStereoSeparation p(rate, alpha, &tab);
    for (long i = 0; i < n; i++) {
        if (reset_before && reset_before[i]) p.reset();
        out[i] = p.process_sample(mux[i], ph[i]);
    }
}

// ---- AGC / Costas ----
void ref_agc_run(float rate, float ref, float gain, const float *in, long n, float *out) {
    AGC a(rate, ref, gain);
    for (long i = 0; i < n; i++) { cf o = a.process_sample(cf(in[2 * i], in[2 * i + 1])); out[2 * i] = real(o); out[2 * i + 1] = imag(o); }
}
void ref_costas_run(float sr, float alpha, float beta, float lim, const float *in, long n, float *out) {
    Costas c(sr, alpha, beta, lim);
    for (long i = 0; i < n; i++) { cf o = c.process_sample(cf(in[2 * i], in[2 * i + 1])); out[2 * i] = real(o); out[2 * i + 1] = imag(o); }
}

#ifdef FMREF_WITH_QT
// ---- discriminator (fm-demodulator.cpp) ----
static const char *decoder_name(int code) {
    switch (code) {
    case 1: return "AM"; case 2: return "FM PLL Decoder"; case 3: return "FM Mixed Demod";
    case 4: return "FM Complex Baseband Delay"; case 5: return "FM Real Baseband Delay";
    case 6: return "FM Difference Based"; default: return "";
    }
}
void ref_demod_run(int32_t rate, int decoder, const float *z /*2n*/, long n, float *out, float *dc, float *carrier, float *kfm_unused) {
    fm_Demodulator d(rate);
    d.setDecoder(QString(decoder_name(decoder)));
    for (long i = 0; i < n; i++) out[i] = d.demodulate(cf(z[2 * i], z[2 * i + 1]));
    if (dc) *dc = d.get_DcComponent();
    if (carrier) *carrier = d.get_carrier_ampl();
    (void)kfm_unused;
}
#endif

// -------------------------------------------------------------------------------------------
// ref_chain: the reference LEAF objects wired like fmProcessor (fm-processor.cpp).  The wiring
// below is this repo's restatement of the glue; every filter/LUT/PLL call is reference code.
// Stops at the resampler input (192 kS/s stereo), the last point pinned by reference code.
// -------------------------------------------------------------------------------------------
struct RefChain {
    int32_t inputRate, fmRate;
    Oscillator lo; SinCos sincos;
    DecimatingFIR band1, band2;
    fftFilter audioFilter, inputFilter;
    pilotRecovery pilot; PerfectStereoSeparation pss;
#ifdef FMREF_WITH_QT
    fm_Demodulator demod;
#endif
    bool inputFilterOn, audioFilterOn, dcr, autoMono, pssActive; int fmMode, sel, loFreq;
    float Lgain, Rgain, pilotDelayPSS, deemphAlpha, volume, panorama, lch, rch;
    cf last, RfDC;
    RefChain(int32_t ir, int32_t fr)
        : inputRate(ir), fmRate(fr), lo(ir), sincos(fr),
          band1(4 * ir / (ir / 6) + 1, fr / 2, ir, ir / (ir / 6)),
          band2((ir / 6) / fr + 1, fr / 2, ir / 6, (ir / 6) / fr),
          audioFilter(2 * 4096, 756), inputFilter(2 * 32768, 251),
          pilot(fr, ((float(19000)) / fr) * (2 * M_PI), 10 * (2 * M_PI) / fr, &sincos),
          pss(fr, 10.0f / fr, &sincos)
#ifdef FMREF_WITH_QT
          , demod(fr)
#endif
    {
        inputFilterOn = audioFilterOn = false; dcr = true; autoMono = true; pssActive = true;
        fmMode = 0; sel = 0; loFreq = 0; Lgain = Rgain = 1; pilotDelayPSS = 0;
        deemphAlpha = 1.0 / (fr / (1000000.0 / 50.0 + 1)); volume = 0.5f; panorama = 1.0f;
        lch = rch = 1.0f; last = 0; RfDC = 0;
    }
};

void *ref_chain_new(int32_t inputRate, int32_t fmRate, int decoder, int inputBw, int lfCutoff, int deemph,
                    float volumeDb, int fmMode, int autoMono, int pssActive, int dcr, int lo, int sel) {
    RefChain *c = new RefChain(inputRate, fmRate);
    if (inputBw > 0) { c->inputFilter.setLowPass(inputBw / 2, inputRate); c->inputFilterOn = true; }
    if (lfCutoff > 0) { c->audioFilter.setLowPass(lfCutoff, fmRate); c->audioFilterOn = true; }
    if (deemph >= 1) { float Tau = 1000000.0 / deemph; c->deemphAlpha = 1.0 / (float(fmRate) / Tau + 1.0); }
    c->volume = std::pow(10.0f, volumeDb / 20.0f);
    c->fmMode = fmMode; c->autoMono = autoMono != 0; c->pssActive = pssActive != 0; c->dcr = dcr != 0;
    c->loFreq = lo; c->sel = sel;
#ifdef FMREF_WITH_QT
    c->demod.setDecoder(QString(decoder_name(decoder)));
#else
    (void)decoder;
#endif
    return c;
}
void ref_chain_free(void *p) { delete (RefChain *)p; }

// returns number of fm-rate samples produced; outputs (may be null): fm IQ, demod, (sum,diff), pre-resampler
long ref_chain_run(void *p, const float *iq, long n, float *fmiq, float *demodOut, float *lrraw, float *prers) {
#ifndef FMREF_WITH_QT
    (void)p; (void)iq; (void)n; (void)fmiq; (void)demodOut; (void)lrraw; (void)prers;
    return -1;
#else
    RefChain *c = (RefChain *)p;
    long m = 0;
    const float alpha = 1.0f / c->inputRate;
    for (long i = 0; i < n; i++) {
        cf x(iq[2 * i], iq[2 * i + 1]);
        if (c->dcr) {
            c->RfDC = (x - c->RfDC) * alpha + c->RfDC;
            float r = real(c->RfDC), q = imag(c->RfDC);
            r = r > 0.01f ? 0.01f : (r < -0.01f ? -0.01f : r);
            q = q > 0.01f ? 0.01f : (q < -0.01f ? -0.01f : q);
            x -= cf(r, q);
        }
        cf v = cf(real(x) * c->Lgain, imag(x) * c->Rgain);
        v = v * c->lo.nextValue(c->loFreq);
        if (c->inputFilterOn) v = c->inputFilter.Pass(v);
        if (!c->band1.Pass(v, &v)) continue;
        if (!c->band2.Pass(v, &v)) continue;
        if (fmiq) { fmiq[2 * m] = real(v); fmiq[2 * m + 1] = imag(v); }
        float demod = c->demod.demodulate(v);
        if (demodOut) demodOut[m] = demod;
        float cur = c->pilot.getPilotPhase(5 * demod);
        bool locked = c->pilot.isLocked();
        if (!locked) { c->pilotDelayPSS = 0; c->pss.reset(); }
        cf audio;
        if (c->fmMode != 2 && (locked || !c->autoMono)) {
            float ph = 2 * (cur + M_PI_4 + 0) - c->pilotDelayPSS;
            if (ph < -2 * M_PI) ph += 4 * M_PI;
            ph = fmod(ph, 2 * M_PI);
            c->pilotDelayPSS = c->pssActive ? c->pss.process_sample(demod, ph) : 0;
            float LRDiff = 2.0 * (c->sel == 6 ? c->sincos.getSin(ph) : c->sincos.getCos(ph)) * demod;
            audio = cf(demod, LRDiff);
        } else audio = cf(demod, 0);
        if (lrraw) { lrraw[2 * m] = real(audio); lrraw[2 * m + 1] = imag(audio); }
        const float sumLR = real(audio), diffLR = imag(audio);
        const float dw = diffLR * (c->fmMode == 1 ? c->panorama : 1.0f);
        const float left = sumLR + dw, right = sumLR - dw;
        switch (c->sel) {
        default: case 0: audio = cf(left, right); break;
        case 1: audio = cf(right, left); break;
        case 2: audio = cf(left, left); break;
        case 3: audio = cf(right, right); break;
        case 4: audio = cf(sumLR, sumLR); break;
        case 5: case 6: audio = cf(dw, dw); break;
        }
        if (c->audioFilterOn) audio = c->audioFilter.Pass(audio);
        audio = c->last = (audio - c->last) * c->deemphAlpha + c->last;
        audio = cf(c->volume * c->lch * real(audio), c->volume * c->rch * imag(audio));
        if (prers) { prers[2 * m] = real(audio); prers[2 * m + 1] = imag(audio); }
        m++;
    }
    return m;
#endif
}


// ---- recursive filters (iir-filters.cpp): kind 0 LowPassIIR(order, f1, fs, type), 1 HighPassIIR, 2 BandPassIIR(order, f1, f2, fs, type)
void *ref_iir_new(int kind, int order, int32_t f1, int32_t f2, int32_t fs, int ftype) {
    if (kind == 0) return (Basic_IIR *)new LowPassIIR((int16_t)order, f1, fs, (int16_t)ftype);
    if (kind == 1) return (Basic_IIR *)new HighPassIIR((int16_t)order, f1, fs, (int16_t)ftype);
    return (Basic_IIR *)new BandPassIIR((int16_t)order, f1, f2, fs, (int16_t)ftype);
}
void ref_iir_free(void *p) { delete (Basic_IIR *)p; }
int ref_iir_coeffs(void *p, float *out /* 6 per quad: A0 A1 A2 B0 B1 B2, then the gain */) {
    Basic_IIR *f = (Basic_IIR *)p;
    for (int i = 0; i < f->numofQuads; i++) {
        out[6 * i] = f->Quads[i].A0; out[6 * i + 1] = f->Quads[i].A1; out[6 * i + 2] = f->Quads[i].A2;
        out[6 * i + 3] = f->Quads[i].B0; out[6 * i + 4] = f->Quads[i].B1; out[6 * i + 5] = f->Quads[i].B2;
    }
    out[6 * f->numofQuads] = f->gain;
    return f->numofQuads;
}
void ref_iir_run(void *p, const float *in, long n, float *out) {
    Basic_IIR *f = (Basic_IIR *)p;
    for (long i = 0; i < n; i++) out[i] = f->Pass(in[i]);
}
// ---- squelch (src/various/squelchClass.cpp): the reference's own object, sample by sample ----
void *ref_squelch_new(int32_t threshold, int32_t keyFrequency, int32_t bufsize, int32_t sampleRate) {
#ifdef FMREF_WITH_QT
    return new squelch(threshold, keyFrequency, bufsize, sampleRate);
#else
    (void)threshold; (void)keyFrequency; (void)bufsize; (void)sampleRate; return nullptr;
#endif
}
void ref_squelch_free(void *p) {
#ifdef FMREF_WITH_QT
    delete (squelch *)p;
#else
    (void)p;
#endif
}
void ref_squelch_set_level(void *p, int n) {
#ifdef FMREF_WITH_QT
    ((squelch *)p)->setSquelchLevel(n);
#else
    (void)p; (void)n;
#endif
}
// n samples through do_noise_squelch (carrier == NULL) or do_level_squelch; flags[i] = getSquelchActive() after sample i
void ref_squelch_run(void *p, const float *in, const float *carrier, float *out, uint8_t *flags, long n) {
#ifdef FMREF_WITH_QT
    squelch *q = (squelch *)p;
    for (long i = 0; i < n; i++) {
        out[i] = carrier ? q->do_level_squelch(in[i], carrier[i]) : q->do_noise_squelch(in[i]);
        if (flags) flags[i] = q->getSquelchActive() ? 1 : 0;
    }
#else
    (void)p; (void)in; (void)carrier; (void)out; (void)flags; (void)n;
#endif
}


// ---- RDS byte work: RDSGroup (rds-group.cpp:33-81) and the tables of ebu-codetables.c, which the reference itself
// pulls in by #include (rds-groupdecoder.cpp:43) -- included from the reference tree in place at the top of this file
void ref_rdsgroup_fields(const uint16_t *blk /*4*/, int32_t *out /*9*/) {
    RDSGroup g;
    g.setBlock(RDSGroup::BLOCK_A, blk[0]); g.setBlock(RDSGroup::BLOCK_B, blk[1]);
    g.setBlock(RDSGroup::BLOCK_C, blk[2]); g.setBlock(RDSGroup::BLOCK_D, blk[3]);
    out[0] = g.getBlock_A(); out[1] = g.getBlock_B(); out[2] = g.getBlock_C(); out[3] = g.getBlock_D();
    out[4] = g.getPiCode(); out[5] = g.getGroupType(); out[6] = g.isTypeBGroup() ? 1 : 0; out[7] = g.isTpFlagSet() ? 1 : 0;
    out[8] = g.getProgrammeType();
    g.clear();
    if (g.getBlock(RDSGroup::BLOCK_A) | g.getBlock(RDSGroup::BLOCK_B) | g.getBlock(RDSGroup::BLOCK_C) | g.getBlock(RDSGroup::BLOCK_D)) out[0] = -1;
}
uint16_t ref_map_ebu(uint8_t alfabet, uint8_t character) { return mapEBUtoUnicode(alfabet, character); }
const char *ref_pty_name(int32_t pty, int32_t locale) { return pty_table[pty][locale]; }   // rds-groupdecoder.cpp:115
}  // extern "C"
