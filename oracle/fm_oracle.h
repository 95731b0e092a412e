/*
 * fm_oracle.h -- CPU restatement of the sdr-j-fm `src/fm` processing chain.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The shipped path (libfmx, HIP) never
 * links, loads or falls back to anything in oracle/.
 *
 * Every function cites the reference file:line (relative to /root/reference) whose
 * arithmetic it restates, including the C++ float/double promotion of each expression.
 *
 * Parity status (see DESIGN.md "Oracle pinning"):
 *   - leaf stages (FIR design, overlap-add FFT filter, radix-2 FFT, decimating FIR, atan2 LUT,
 *     SinCos LUT, LO table, pllC, pilot PLL, PSS, discriminators, RRC taps, AGC, Costas):
 *     pinned bit-exactly against the reference's own classes compiled from
 *     /root/reference into oracle/_ref/libfmref.so (tests/test_oracle_vs_ref.py) and
 *     against committed fixtures generated from that library (tests/golden/).
 *   - chain glue of fmProcessor::run() (fm-processor.cpp:373-759): restated; the file itself
 *     is unbuildable here (needs portaudio/sndfile/samplerate/qwt headers and moc/uic output),
 *     so the glue is pinned only through a harness that wires the reference leaf classes
 *     in the same order (oracle/ref_wrap.cpp: ref_chain_*).
 *   - 192k->48k resampler: libsamplerate is third-party, absent and unpinned
 *     (newconverter.cpp:37) -> own documented design, "parity unpinned" for that stage.
 */
#ifndef FM_ORACLE_H
#define FM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } fmo_c32;

/* ---------- filter-kernel design (fir-filters.cpp) ---------- */
/* LowPassFIR::newKernel fir-filters.cpp:41-62 -> real taps h[N] (imag part is 0) */
void fmo_lowpass_kernel(int N, int32_t Fc, int32_t fs, float *h);
/* DecimatingFIR::newKernel(low) fir-filters.cpp:327-347 -> complex taps (h/sum, h) */
void fmo_decim_kernel(int N, int32_t low, int32_t fs, fmo_c32 *k);
/* BandPassFIR::newKernel fir-filters.cpp:197-222 */
void fmo_bandpass_kernel(int N, int32_t low, int32_t high, int32_t fs, fmo_c32 *k);
/* ShapingFilter::root_raised_cosine shaping_filter.cpp:4-54 ; returns ntaps|1 */
int  fmo_rrc_kernel(double gain, double fs, double symrate, double alpha, int ntaps, float *taps);

/* ---------- FFT (fft-complex.cpp:50-102) ---------- */
int  fmo_fft_radix2(fmo_c32 *vec, long n, int inverse);

/* ---------- overlap-add filter (fft-filters.cpp:29-201) ---------- */
typedef struct fmo_fftfilter fmo_fftfilter;
fmo_fftfilter *fmo_fftfilter_new(int fftSize, int degree);
void  fmo_fftfilter_free(fmo_fftfilter *);
void  fmo_fftfilter_set_lowpass(fmo_fftfilter *, int32_t low, int32_t rate);
void  fmo_fftfilter_set_band(fmo_fftfilter *, int32_t low, int32_t high, int32_t rate);
void  fmo_fftfilter_set_hilbert(fmo_fftfilter *);
fmo_c32 fmo_fftfilter_pass_c(fmo_fftfilter *, fmo_c32 z);
float   fmo_fftfilter_pass_r(fmo_fftfilter *, float x);
void  fmo_fftfilter_run_c(fmo_fftfilter *, const fmo_c32 *in, fmo_c32 *out, long n);
void  fmo_fftfilter_run_r(fmo_fftfilter *, const float *in, float *out, long n);

/* ---------- decimating FIR (fir-filters.cpp:316-424) ---------- */
typedef struct fmo_decim fmo_decim;
fmo_decim *fmo_decim_new(int N, int32_t low, int32_t fs, int D);
void  fmo_decim_free(fmo_decim *);
int   fmo_decim_pass(fmo_decim *, fmo_c32 z, fmo_c32 *out);
long  fmo_decim_run(fmo_decim *, const fmo_c32 *in, long n, fmo_c32 *out);

/* ---------- LUTs ---------- */
typedef struct fmo_sincos fmo_sincos;          /* sincos.cpp:36-91 */
fmo_sincos *fmo_sincos_new(int32_t rate);
void  fmo_sincos_free(fmo_sincos *);
float fmo_sincos_sin(const fmo_sincos *, float phase);
float fmo_sincos_cos(const fmo_sincos *, float phase);
fmo_c32 fmo_sincos_complex(const fmo_sincos *, float phase);
const fmo_c32 *fmo_sincos_table(const fmo_sincos *);

typedef struct fmo_atan fmo_atan;              /* Xtan2.cpp:12-100 */
fmo_atan *fmo_atan_new(void);
void  fmo_atan_free(fmo_atan *);
float fmo_atan2(const fmo_atan *, float y, float x);
const float *fmo_atan_table(const fmo_atan *, int which); /* 0..7: PPY PPX PNY PNX NPY NPX NNY NNX */

/* LO table entry i of Oscillator(rate) oscillator.cpp:26-35 */
fmo_c32 fmo_lo_value(int32_t rate, int32_t i);

float fmo_pi_constrain(float v);               /* fm-constants.h:148-158 */

/* ---------- complex PLL (pllC.cpp:37-90) ---------- */
typedef struct fmo_pll fmo_pll;
fmo_pll *fmo_pll_new(int32_t rate, float freq, float lofreq, float hifreq, float bandwidth,
                     const fmo_sincos *tab, const fmo_atan *at);
void  fmo_pll_free(fmo_pll *);
void  fmo_pll_do(fmo_pll *, fmo_c32 signal);
float fmo_pll_phase_incr(const fmo_pll *);

/* ---------- discriminator (fm-demodulator.cpp:51-241) ---------- */
enum { FMO_DEC_AM = 1, FMO_DEC_PLL = 2, FMO_DEC_MIXED = 3, FMO_DEC_COMPLEX_BB = 4,
       FMO_DEC_REAL_BB = 5, FMO_DEC_DIFF = 6 };
typedef struct fmo_demod fmo_demod;
fmo_demod *fmo_demod_new(int32_t rateIn);
void  fmo_demod_free(fmo_demod *);
void  fmo_demod_set_decoder(fmo_demod *, int code);
float fmo_demod_demodulate(fmo_demod *, fmo_c32 z);
float fmo_demod_dc(const fmo_demod *);
float fmo_demod_carrier(const fmo_demod *);
float fmo_demod_kfm(const fmo_demod *);

/* ---------- pilot PLL (pilot-recover.cpp:28-83) ---------- */
typedef struct fmo_pilot fmo_pilot;
fmo_pilot *fmo_pilot_new(int32_t rate, float omega, float gain, const fmo_sincos *tab);
void  fmo_pilot_free(fmo_pilot *);
float fmo_pilot_phase(fmo_pilot *, float pilot);
int   fmo_pilot_locked(const fmo_pilot *);
float fmo_pilot_strength(const fmo_pilot *);

/* ---------- PSS (stereo-separation.cpp:27-109) ---------- */
typedef struct fmo_pss fmo_pss;
fmo_pss *fmo_pss_new(int32_t rate, float alpha, const fmo_sincos *tab);
void  fmo_pss_free(fmo_pss *);
void  fmo_pss_reset(fmo_pss *);
float fmo_pss_process(fmo_pss *, float mux, float phase);
int   fmo_pss_minimized(const fmo_pss *);
float fmo_pss_mean_error(const fmo_pss *);

/* ---------- RDS leaf pieces ---------- */
typedef struct { float rate, ref, gain; } fmo_agc;                  /* agc.h:8-24 */
fmo_c32 fmo_agc_process(fmo_agc *, fmo_c32 in);
typedef struct { float alpha, beta, freqLimit, freq, phase; } fmo_costas;   /* costas.h:8-42 */
void    fmo_costas_init(fmo_costas *, float sr, float alpha, float beta, float limitHz);
fmo_c32 fmo_costas_process(fmo_costas *, fmo_c32 z);


/* ---------- recursive filters (iir-filters.cpp: Chebyshev / Butterworth prototypes :120-218, low-pass :451-490,
 * high-pass :497-540, Bilineair :73-109, Basic_IIR::Pass(float) iir-filters.h:89-103); all arithmetic in f32 as DSPFLOAT ---------- */
#define FMO_IIR_CHEBYSHEV 0100
#define FMO_IIR_BUTTERWORTH 0101
#define FMO_IIR_MAXQ 16
typedef struct { int nq; float q[FMO_IIR_MAXQ][6]; /* A0 A1 A2 B0 B1 B2 */ float gain; float m1[FMO_IIR_MAXQ], m2[FMO_IIR_MAXQ]; } fmo_iir;
void  fmo_iir_lowpass(fmo_iir *f, int order, int32_t fpass, int32_t fs, int ftype);
void  fmo_iir_highpass(fmo_iir *f, int order, int32_t fpass, int32_t fs, int ftype);
void  fmo_iir_bandpass(fmo_iir *f, int order, int32_t flow, int32_t fhigh, int32_t fs, int ftype);   /* :552-595 */
float fmo_iir_pass(fmo_iir *f, float v);
/* the block synchroniser as rdsDecoder_3 sees it (sync-error count, lock) over a bit array, from a fresh state */
void  fmo_bsync_run(const uint8_t *bits, long n, int32_t *sync_errors, int32_t *synced);
/* rdsDecoder_1's constants as the kernels hold them: rdsFilter taps [21], Match kernel [43], sharpFilter [8][A1 A2 B1 B2], gain */
void  fmo_rds1_coeffs(float *out);
/* test helpers with the signatures of ref_iir_* (kind 0 low-pass, 1 high-pass, 2 band-pass) */
/* squelch (squelchClass.cpp:11-113): state + the calls fmProcessor makes (fm-processor.cpp:87, 410-413, 499-509) */
typedef struct { float noiseThr, levelThr, avgHigh, avgLow; int32_t hold, rate, count, suppress; fmo_iir high, low; } fmo_squelch;
void  fmo_squelch_init(fmo_squelch *q, int32_t threshold, int32_t keyFrequency, int32_t bufsize, int32_t sampleRate);
void  fmo_squelch_set_level(fmo_squelch *q, int n);
float fmo_squelch_noise(fmo_squelch *q, float soundSample);
float fmo_squelch_level(fmo_squelch *q, float soundSample, float carrierLevel);
fmo_squelch *fmo_squelch_new(int32_t threshold, int32_t keyFrequency, int32_t bufsize, int32_t sampleRate);
void  fmo_squelch_free(fmo_squelch *q);
int   fmo_squelch_active(const fmo_squelch *q);
void  fmo_squelch_run(fmo_squelch *q, const float *in, const float *carrier, float *out, uint8_t *flags, long n);
void *fmo_iir_new(int kind, int order, int32_t f1, int32_t f2, int32_t fs, int ftype);
void  fmo_iir_free(void *);
int   fmo_iir_coeffs(void *, float *out);
void  fmo_iir_run(void *, const float *in, long n, float *out);

/* ---------- batch runners used by the tests (same signatures as oracle/ref_wrap.cpp's ref_*) ---------- */
void fmo_sincos_eval(const fmo_sincos *, const float *phase, long n, float *s, float *c, float *cplx);
void fmo_atan2_eval(const float *y, const float *x, long n, float *out);
void fmo_pi_constrain_eval(const float *in, long n, float *out);
void fmo_pll_run(int32_t rate, float freq, float lo, float hi, float bw, const float *sig, long n, float *incr);
void fmo_demod_run(int32_t rate, int decoder, const float *z, long n, float *out, float *dc, float *carrier);
void fmo_pilot_run(int32_t rate, float omega, float gain, const float *pilot, long n,
                   float *phase, uint8_t *locked, float *strength);
void fmo_pss_run(int32_t rate, float alpha, const float *mux, const float *ph, long n, float *out,
                 const uint8_t *reset_before);
void fmo_agc_run(float rate, float ref, float gain, const float *in, long n, float *out);
void fmo_costas_run(float sr, float alpha, float beta, float lim, const float *in, long n, float *out);
void fmo_pilot_constants(int32_t fmRate, float *omega, float *gain, float *pssAlpha);

/* ---------- fmx resampler (own design; replaces libsamplerate, see header note) ---------- */
#define FMO_RS_TAPS 128
void fmo_resampler_taps(float *h /* FMO_RS_TAPS */);
/* second converter workingRate -> audioRate (fm-processor.cpp:89-91, 825-838; own design like the first): p / q, taps per phase */
#define FMO_CONV2_MAXP 640
#define FMO_CONV2_MAXNT 256
int fmo_conv2_design(int32_t inRate, int32_t outRate, int32_t *p, int32_t *q, int32_t *nt, float *taps /* [p][nt] or NULL */);

/* ---------- whole chain (fm-processor.cpp:373-759,772-838) ---------- */
typedef struct {
    int32_t inputRate, fmRate, workingRate, audioRate;
    int32_t fmMode;           /* 0 Stereo, 1 StereoPano, 2 Mono          fm-processor.h:83 */
    int32_t soundSelector;    /* 0..6 Channels enum                        fm-processor.h:88-90 */
    int32_t decoder;          /* FMO_DEC_*                                 fm-demodulator.cpp:27-44 */
    int32_t inputFilterBw;    /* Hz, 0 = "Off"  (setBandwidth)             fm-processor.cpp:232-239 */
    int32_t lfCutoff;         /* Hz, <=0 = off  (setlfcutoff)              fm-processor.cpp:762-770 */
    int32_t deemphasis;       /* us, 0 = keep ctor default alpha           fm-processor.cpp:174,291-297 */
    float   volumeDb;         /* setVolume; NaN-free; ctor default 0.5 is used when useCtorVolume */
    int32_t useCtorVolume;    /* 1 -> volumeFactor = 0.5f (ctor :127)      */
    int32_t balance;          /* -100..100 setSoundBalance                 fm-processor.cpp:282-286 */
    int32_t panorama;         /* 0..200 setStereoPanorama                  fm-processor.cpp:277-280 */
    float   attL, attR;       /* setAttenuation (Lgain,Rgain)              fm-processor.cpp:351-359 */
    int32_t loFrequency;      /* set_localOscillator                       fm-processor.cpp:866-868 */
    int32_t dcRemove, autoMono, pssActive;
    int32_t rdsMode;          /* 0 off, 1..3 = RDS_1..3 */
    int32_t squelchMode;      /* 0 OFF, 1 NSQ (noise squelch), 2 LSQ (level squelch)   fm-processor.cpp:499-509 */
    int32_t squelchValue;     /* set_squelchValue 0..100: applied at a block start when it differs from the last one (:410-413) */
    int32_t testTone;         /* setTestTone (fm-processor.cpp:931-933): 1 kHz bursts of 25 ms every 2 s mixed into the PCM (:800-823) */
    int32_t dispDelay;        /* setDispDelay (:935-937): steps of the peak-level delay line */
    /* one-shot, consumed by fmo_chain_configure: the setter was CALLED, whatever the value -- setBandwidth (:232-239) / setlfcutoff (:762-770) set
     * newInputFilter / newAudioFilter even when the current value is selected again, and the loop then restarts the filter's block (:396-408) */
    int32_t touchInputFilter, touchLfCutoff;
    /* TEST HOOK (0 = off: the chain is the reference's, bit for bit).  The reference's input filter is an f32 FFT overlap-add (fft-filters.cpp:132-163 on
     * fft-complex.cpp's radix-2 transform): every output carries its rounding noise, ~3e-7 of the block's scale -- a third of the first outputs behind the
     * filter's latency, which the limiter's z / |z| turns into phase.  An exact convolution (libfmx), or the same FFT built with other compiler flags, has
     * another realisation of that noise.  testFilterNoise > 0 adds uniform pseudo-random noise of that standard deviation to each component of the input
     * filter's output: "the reference with another realisation of its own rounding noise" -- what tests/soak_random.py asks when a channel is out of
     * tolerance: does the reference's PCM move as far under its own noise? */
    float   testFilterNoise;
    int32_t testNoiseSeed;
} fmo_config;

void fmo_config_defaults(fmo_config *);   /* GUI-effective defaults, SURVEY 3.3 */

enum { FMO_TAP_FM_IQ = 0,      /* complex @fmRate after fmBand_2 (fm-processor.cpp:474)         */
       FMO_TAP_DEMOD = 1,      /* float   @fmRate after demodulate (:497)                        */
       FMO_TAP_LRRAW = 2,      /* complex @fmRate (sum,diff) out of process_signal_with_rds (:515) */
       FMO_TAP_PRE_RS = 3,     /* complex @fmRate into the resampler (:630-634)                  */
       FMO_TAP_PILOT = 4,      /* float   @fmRate currentPilotPhase (:695)                       */
       FMO_TAP_PSS = 5,        /* float   @fmRate pilotDelayPSS after the sample (:716)          */
       FMO_TAP_RDS_IQ = 6,     /* complex @24k after rdsDecimator (:553)                         */
       FMO_TAP_COUNT = 7 };

typedef struct {
    float   dcValRf, dcValIf, pssPhaseShiftDegree, pssPhaseChange;
    int32_t pssState;          /* 0 OFF 1 ANALYZING 2 ESTABLISHED */
    float   pilotLockStrength;
    int32_t pilotLocked;
    float   peakLeftDb, peakRightDb;
    int64_t fmSamples, pcmFrames;
    int32_t squelchActive;    /* getSquelchState */
    int32_t pad_;
} fmo_meta;

typedef struct fmo_chain fmo_chain;
fmo_chain *fmo_chain_new(const fmo_config *);
void  fmo_chain_free(fmo_chain *);
/* apply changed settings (takes effect like the reference: filters at the next 16384 block) */
void  fmo_chain_configure(fmo_chain *, const fmo_config *);
void  fmo_chain_trigger_frequency_change(fmo_chain *);   /* fm-processor.cpp:849-855 */
void  fmo_chain_set_tap(fmo_chain *, int tap, float *buf, long cap_floats);
long  fmo_chain_tap_count(const fmo_chain *, int tap);   /* floats written */
/* feed n complex samples (interleaved I,Q); returns PCM frames written (interleaved L,R) */
long  fmo_chain_process(fmo_chain *, const float *iq, long n, float *pcm, long pcm_cap_frames);
void  fmo_chain_meta(const fmo_chain *, fmo_meta *);
/* showPeakLevel events so far (fm-processor.cpp:772-798): (leftDb, rightDb) pairs as emitted, i.e. behind the
 * delay line; copies up to cap events, returns the total count */
long  fmo_chain_peaks(const fmo_chain *, float *lr_db, long cap_events);
/* the 1200 samples of one test-tone burst (fm-processor.cpp:808-813,818-821): the same for every burst */
void  fmo_test_tone_burst(int32_t workingRate, float *dst, long n);
/* RDS bits produced so far (rdsMode==2): copies up to cap, returns total count */
long  fmo_chain_rds_bits(const fmo_chain *, uint8_t *bits, long cap);

/* ---------- deterministic synthetic IQ (own generator, used by tests and bench) ---------- */
typedef struct {
    int32_t inputRate;
    double  carrierAmp;        /* 0.5 */
    double  deviationHz;       /* 75000 */
    double  offsetHz;          /* carrier offset in the IQ stream */
    double  leftHz, rightHz;   /* programme tones */
    double  leftAmp, rightAmp;
    int32_t stereo;            /* 0: mono MPX = 0.9*(L+R)/2 ; 1: pilot + 38k DSB-SC */
    double  pilotLevel;        /* 0.10 */
    int32_t rds;               /* 1: add 57 kHz RDS sub-carrier */
    double  rdsLevel;          /* 0.03 */
    uint64_t noiseSeed;        /* 0 = no noise */
    double  noiseSigma;        /* per-component std dev */
    double  dcI, dcQ;          /* DC offset added */
    uint64_t rdsBitsSeed;
} fmo_siggen_config;
typedef struct fmo_siggen fmo_siggen;
fmo_siggen *fmo_siggen_new(const fmo_siggen_config *);
void fmo_siggen_free(fmo_siggen *);
void fmo_siggen_run(fmo_siggen *, float *iq, long n);
void fmo_siggen_set_rds_bits(fmo_siggen *, const uint8_t *bits, long n);   /* test hook: data bits to send, cyclically */
long fmo_siggen_rds_bits(const fmo_siggen *, uint8_t *bits, long cap);

#ifdef __cplusplus
}
#endif
#endif
