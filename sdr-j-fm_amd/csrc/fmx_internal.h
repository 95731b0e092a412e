// fmx_internal.h -- structures shared by the host side (fmx_api.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fmx {

constexpr int DECIM = 12;              // inputRate / fmRate (2304000 / 192000)
constexpr int A_TILE_COLS = 256;       // front-end tile: 256 fm-rate outputs = 3072 input samples
constexpr int A_HIST_COLS = 25;        // history columns kept per channel (>= max taps/12 + 1)
constexpr int A_MAX_ND = 25;           // tap columns: 287 taps at off=6 -> 25 columns of 12
constexpr int A_TAPS_STRIDE = (A_MAX_ND + 2) * DECIM;   // 324: host image Tz[(d+1)*12 + r], d = -1..25, zero rows at both ends
constexpr int A_TAPS_ROW = 32;         // device image Trd[r][d] (d contiguous, zero padded): one row = the taps of phase r
constexpr int A_TAPS_DEV = DECIM * A_TAPS_ROW;
constexpr int PSS_TAPS = 295;          // stereo-separation.cpp:31
constexpr int PSS_DELAY = 2048 - 295;  // overlap-add latency fftSize - degree (fft-filters.cpp:34)
constexpr int PSS_CHUNK = 1744;        // <= PSS_DELAY and a multiple of the work-array tile (16 rows)        // PSS feedback lag: the only chunked part of stage B (<= PSS_DELAY)
constexpr int RS_TAPS = 128;           // fmx resampler (oracle/fm_oracle.c fmo_resampler_taps)
constexpr int AUDIO_TAPS = 756;        // fm-processor.cpp:76
constexpr int AUDIO_DELAY = 8192 - 756;
constexpr int C_MAX_TAPS = AUDIO_TAPS + RS_TAPS - 1;   // 883
constexpr int C_TAPS_STRIDE = 896;     // padded
constexpr int C_TILE = 256;            // PCM frames per audio-FIR tile
constexpr int GAIN_FIX_BACK = RS_TAPS + 192;   // fm samples in front of a call's first that its first frames' resampler windows can reach: a call's frames begin at the
                                               // 192-sample block its first fm sample falls into (frames_geom: M0 = 48 (J0 / 192)), up to 191 samples before J0
constexpr int GAIN_FIX_FRAMES = GAIN_FIX_BACK / 4;   // PCM frames whose resampler memory straddles a gain change (80; 32 when J0 is a multiple of 192)
constexpr int NSQ_QUADS = 10;          // ((20 + 1) & 0176) / 2 biquads per filter (iir-filters.cpp:454)
constexpr int TT_SILENT = 96001;       // ++TimePeriodCounter > workingRate * 2.0f fires on the 96001st silent frame (fm-processor.cpp:816-817)
constexpr int TT_BURST = 1200;         // workingRate * 0.025f (:819)
constexpr int TT_CYCLE = TT_SILENT + TT_BURST;
constexpr int PK_WIN = 961;            // ++peakLevelCurSampleCnt > workingRate / 50 (:781-782, :142)
constexpr int PK_RING = 256;           // finished windows kept per channel (5 s)
constexpr int SINCOS_N = 192000;
constexpr int ATAN_N = 8192;
constexpr int ARCSINE_N = 4 * 8192;
constexpr int TRIG2_A = 750, TRIG2_B = 256;   // 192000 = 750 * 256: idx = 256 a + b
constexpr int TRIG2_APAD = 2;                  // wrap-around entries a = 750, 751 (idx may reach N)
constexpr int TRIG2_N = TRIG2_A + TRIG2_APAD + TRIG2_B;

// front-end filter description of one tap set
struct FrontSet {
    int32_t nd;          // tap columns used
    int32_t off;         // newest input sample of output j is 12*j + off
    int32_t delay_fm;    // pure delay in fm samples applied after the FIR (overlap-add latency)
    float   gain_re, gain_im;   // complex gain (1 + j*S1)(1 + j*S2) of the two DecimatingFIRs
    // RF DC removal applied BEHIND the FIR (channels without LO mix, fmx_front.hip): the taps' sum, and where the RfDC value an
    // output takes sits -- the taps' centre of mass: boundary column = output column - dc_k, weight dc_w towards the next one
    float   hsum, dc_w;
    int32_t dc_k;
    int32_t zshift;      // twins (CallGeom::twins > 1): output column q of this twin is fm sample twins * (q + zshift) + twin index of the ring
};
constexpr int DCV_SAVE = 16;           // RfDC boundary values kept per channel between calls (14 used)

struct AudioSet {
    int32_t ntaps;       // 883 (audio filter on) or 128
    int32_t delay;       // AUDIO_DELAY or 0
};

// per-channel settings (host mirror uploaded before each call when dirty)
struct ChanParams {
    int32_t stream;
    int32_t front_set, audio_set;
    int32_t fm_mode, sound_sel, decoder, auto_mono, pss_active, dc_remove, rds_mode;
    int32_t lo_freq;
    int32_t lo_period;   // inputRate / gcd(|lo_freq|, inputRate) when <= LO_LDS_MAX (the LO phase sequence repeats), else 0
    float   att_l, att_r;
    float   deemph_alpha, volume, left_ch, right_ch, panorama;
    int32_t actions;     // one-shot ACT_* bits consumed by the kernels of the next call
    int32_t squelch_mode;   // 0 off, 2 level squelch (set_squelchMode)
    float   squelch_thr;    // levelSquelchThreshold squelchClass.cpp:35
    int32_t test_tone;      // setTestTone fm-processor.cpp:931-933
    float   squelch_nthr;   // noiseSquelchThreshold squelchClass.cpp:36
    float   deemph_l2;      // log2 (1.0f - deemph_alpha) (fused stage B: scan weights)
    float   hlo_re, hlo_im; // a channel with a local oscillator on the matrix-pipe input filter (fmx_front4lo.hip): the sum of its complex taps, sum_m G [m] table [(m lo) mod R]
                            // -- what the filter makes of the RF DC value the reference subtracts in front of the mix; (hsum, 0) without an oscillator
    int32_t pll_seq;        // FMX_P_PLL_SOLVER resolved on the host: 1 the pilot PLL of this channel is evaluated sample by sample in every segment;
                            // 0 Newton's method on the segment while the pilot is comfortably in lock, sample by sample otherwise (PLL_GUARD in
                            // fmx_stageb.hip: the lock decisions are then taken on the reference's own trajectory); 2 Newton's method always (diagnostic)
};
enum { ACT_TRIGGER_FREQ = 1, ACT_RESTART_PSS = 2, ACT_DC_RESET = 4 };

// per-channel DSP state carried between calls
struct ChanState {
    // front end
    float   dc_re, dc_im;
    int32_t lo_phase;
    // discriminator (fm-demodulator.cpp:79-86)
    float   Imin1, Qmin1, Imin2, Qmin2, fm_afc, am_carr;
    // pllC (pllC.cpp:37-60)
    float   nco_phase, phase_incr;
    // pilot PLL (pilot-recover.cpp:28-47)
    float   pil_phase, pil_old, pil_lock;
    int32_t pil_stable, pil_locked;
    // PSS (stereo-separation.cpp:46-54)
    float   pss_acc, pss_mean;
    int32_t pss_lock_cnt, pss_unlock_cnt, pss_minimized;
    float   pilot_delay_pss;
    int64_t pss_count;           // number of process_sample calls so far (filter time base)
    int32_t pss_call_total;      // process_sample calls made by the current call (seq1 -> deemph advances pss_count)
    int32_t pad0;
    // de-emphasis (fm-processor.cpp:594-595)
    float   de_l, de_r;
    // audioGainCorrection (fm-processor.cpp:303-306): the gains of the previous call, for the transient of a change
    float   prev_gl, prev_gr;
    int32_t gain_valid, pad_g;
    // fade-in (fm-processor.cpp:130-131,638-642)
    int64_t fade_start_frame;    // PCM frame index at which suppressAudioSampleCnt was (re)armed
    // meta snapshot (fm-processor.cpp:662-684)
    int32_t my_count;
    float   meta_dc_rf, meta_dc_if, meta_pss_deg, meta_pss_change, meta_lock_strength;
    int32_t meta_pss_state, meta_locked;
    // level squelch (squelchClass.cpp:20-28)
    int32_t sq_count, sq_suppress;
    // test tone (fm-processor.cpp:800-823): position in the 97201-frame cycle (96001 silent frames, then the 1200-frame burst)
    int32_t tt_pos;
    // peak meter (fm-processor.cpp:772-798): frames in the current 961-frame window, its maxima so far, windows finished so far
    int32_t pk_cnt;
    float   pk_l, pk_r;
    int32_t pk_events;
    int32_t hist_fmt;    // stage-A history (DeviceBuffers::hist): 0 raw samples (channel without LO), 1 DC-corrected, balanced and mixed ones
    // noise squelch (squelchClass.cpp:47-87): decaying averages and the (m1, m2) memories of the two order-20 filters
    float   sq_avg_hi, sq_avg_lo;
    float   sq_m[2][NSQ_QUADS][2];
    // segments of the pilot PLL that did not settle and were replayed sample by sample (fmx_stageb.hip), since fmx_create
    int32_t pll_replays;
    // ChanParams::pll_seq == 0: the next segment may use Newton's method (the pilot was in lock behind the last segment and the lock metric stayed
    // above PLL_GUARD throughout it); 0 after fmx_create: the acquisition runs sample by sample
    int32_t pll_newton_ok;
    int32_t pll_exact_segs, pad_x;   // segments the guard sent through the sample-by-sample evaluation, since fmx_create
};

// Work arrays of stage B (w_*): element (row r, channel ch) lives at ((r / 16) * pitch + ch) * 16 + r % 16 -- tiles of 16
// consecutive fm samples per channel.  A lane-per-channel recurrence kernel moves its 16 samples with four dwordx4
// operations (64 contiguous bytes per lane), a time-parallel kernel maps threads as (row-in-tile fastest, channel next)
// and stays fully coalesced (1 KB contiguous per 64 channels).
constexpr int LO_LDS_MAX = 1024;       // LO phase periods up to this are tabulated in LDS by the input-FIR kernel
constexpr int WT = 16;
__host__ __device__ __forceinline__ size_t widx(int64_t r, int ch, int pitch) {
    return ((size_t)(r >> 4) * (size_t)pitch + (size_t)ch) * WT + (size_t)(r & 15);
}

// The SinCos table (sincos.cpp:45-54: (float)cos / (float)sin of 2 pi i / Rate in f64) WITHOUT the table: octant reduction of
// the index, two f64 Taylor polynomials on [0, pi/4] (sin to x^15, cos to x^16: truncation < 5e-17), rounded to f32.
// fmx_create runs the very same operation sequence (explicit fma) over all 192000 indices against the reference's entries;
// the entries that differ (the exact zero crossings, where the reference's f64 angle is not exact) go into the exception
// lists; with more than four per list `ok` stays 0 and the kernels gather from the table in memory.
struct SinPoly {
    double step;                 // 2 pi / Rate
    int32_t ok, ns, nc, pad_;
    int32_t s_idx[4]; float s_val[4];     // sine entries that differ
    int32_t c_idx[4]; float c_val[4];     // cosine entries that differ
};
__host__ __device__ __forceinline__ void sincos_poly(int idx, double step, float *sn, float *cs) {
    constexpr int OCT = SINCOS_N / 8;
    const int q = idx / OCT, r = idx - OCT * q, k = q & 3;
    const int rr = (k & 1) ? OCT - r : r;
    const double x = (double)rr * step, z = x * x;
    double ps = -1.0 / 1307674368000.0;
    ps = __builtin_fma(ps, z, 1.0 / 6227020800.0); ps = __builtin_fma(ps, z, -1.0 / 39916800); ps = __builtin_fma(ps, z, 1.0 / 362880);
    ps = __builtin_fma(ps, z, -1.0 / 5040); ps = __builtin_fma(ps, z, 1.0 / 120); ps = __builtin_fma(ps, z, -1.0 / 6);
    const double sv = __builtin_fma(x * z, ps, x);
    double pc = 1.0 / 20922789888000.0;
    pc = __builtin_fma(pc, z, -1.0 / 87178291200.0); pc = __builtin_fma(pc, z, 1.0 / 479001600); pc = __builtin_fma(pc, z, -1.0 / 3628800);
    pc = __builtin_fma(pc, z, 1.0 / 40320); pc = __builtin_fma(pc, z, -1.0 / 720); pc = __builtin_fma(pc, z, 1.0 / 24); pc = __builtin_fma(pc, z, -0.5);
    const double cv = __builtin_fma(z, pc, 1.0);
    // angle = k pi/4 + a (k even, x = a) or (k + 1) pi/4 - b (k odd, x = b)
    double s, c;
    switch (k) {
    default:
    case 0: s = sv; c = cv; break;
    case 1: s = cv; c = sv; break;
    case 2: s = cv; c = -sv; break;
    case 3: s = sv; c = -cv; break;
    }
    if (q >= 4) { s = -s; c = -c; }
    *sn = (float)s; *cs = (float)c;
}

struct DeviceTables {
    const float2 *sincos;        // [SINCOS_N] (cos, sin)
    const float  *atan_ppy;      // [ATAN_N + 1]
    const float  *arcsine;       // [ARCSINE_N + 1]
    const float2 *lo_table;      // [inputRate] or null when every lo == 0
    const double2 *trig2;        // [TRIG2_N] f64 (cos,sin) factors exp(j2pi 256a/N), exp(j2pi b/N); null if the host check failed
    const float  *front_taps;    // [sets][DECIM][A_TAPS_ROW]  Trd[r][d] = G[12 d + off - r] (0 outside)
    const FrontSet *front_sets;
    const float  *pss_taps;      // [PSS_TAPS]
    const float2 *fft_w;         // [fftc::W_COUNT] stage twiddles of fmx_fftconv.h
    const float2 *pss_hs;        // [2048] spectrum of the PSS taps in the forward transform's slot order, 1 / N included; null: direct FIR (FMX_PSS_FIR=direct)
    const float  *audio_taps;    // [sets][C_TAPS_STRIDE]
    const float  *audio_lp_taps; // [sets][AUDIO_TAPS] the audio low-pass alone (gain_fix_kernel), h[0] first
    const float  *rs_taps;       // [RS_TAPS] the resampler alone, h[0] first
    const float2 *audio_spec;    // [sets][4][2048] spectra of the folded FIR's four decimation phases, slot order of fmx_fftconv.h, 1 / N
                                 // included; null: the direct form (FMX_AUDIO_FIR=direct)
    const AudioSet *audio_sets;
    double  sincos_C;            // Rate / (2*M_PI)  (sincos.cpp:42)
    float   K_FM, K_FM_rcp, pil_omega_rcp;   // rcp = RN(1/c) for fdiv_const
    float   pil_omega, pil_gain, pss_alpha, pss_lock_alpha;
    float   afc_l2, lock_l2, pssmean_l2;   // log2 of the decays of the AFC, the lock metric and the PSS mean error (fused stage B: scan weights)
    float   pll_beta, pll_lo, pll_hi, pll_center;
    const float *nsq_coef;       // [2][NSQ_QUADS][4] (A1, A2, B1, B2) + [2] gains: high-pass 69.9 kHz, low-pass 70 kHz (squelchClass.cpp:11-18); null until used
    SinPoly sp;
    float   wrap32_c;            // fl32(fl32_above(2 pi) - 2 pi)
    int32_t wrap32_ok;           // the f32 form of the pilot-phase wrap was verified on the host for every float it can see
};

struct CallGeom {
    int64_t g0;          // global index of the first input sample of this call
    int64_t n;           // input samples per stream
    int64_t J0, J1;      // fm samples [J0, J1) are demodulated in this call
    int64_t M0, M1;      // PCM frames [M0, M1) are produced in this call
    int32_t ring_mask;   // fm-rate ring capacity - 1
    int32_t dring_mask;  // d ring capacity - 1
    int32_t sring_mask;  // s ring capacity - 1
    int32_t input_rate;
    int32_t pitch;       // row pitch (elements) of the sample-major work arrays: channels + pad, so that
                         // consecutive rows do not land on the same HBM channel/bank (power-of-two strides do)
    int32_t streams_private;   // every stream is read by exactly one channel (stage A then loads it nontemporally)
    int64_t stream_stride, pcm_stride;   // in complex samples / frames
    int32_t iq_format;   // fmx_iq_format of the input buffer
    float   iq_scale;    // 1/128 (U8, S8) or 1/denominator (S16)
    int32_t gain_fix;    // the call's first GAIN_FIX_FRAMES frames take a correction from B.gfix (a volume / balance change, fmx_audio.hip)
    int32_t stageb_form; // FMX_P_STAGEB_FORM (host side only: launch_demod_fused)
    int32_t twins;       // stage A always decimates by 12; an input rate the reference decimates by 12 / twins (6: twins = 2, 1: twins = 12) runs
                         // `twins` workgroups per channel, twin p with the tap alignment of output phase p, interleaved in the fm-rate ring
    int32_t channels;
    int32_t n_cus;       // compute units of the handle's device (host side only: the one-kernel / two-kernel choice of stage B)
    int32_t pre_processed;   // stage A reads a stream that pre_kernel / the overlap-add machine (fmx_ola.hip) have already made: one stream per
    int32_t no_deemph;       // channel, no RF DC removal, balance or LO mix here, history always raw, no front-end state written.  no_deemph: stage B
                             // writes the stereo pair to the d ring as it is: the audio low-pass (a block machine then) comes first, deemph_kernel behind it
    int32_t streams;         // IQ streams of the handle
    int32_t front4;          // stage A: != 0 the handle runs front4_kernel (fmx_front4.hip: the filter on the matrix pipe; 2: its complex-tap variant for handles with
                             // local oscillators, fmx_front4lo.hip; every tap set the
                             // long fold with RfDC taken 12 columns back); launch_front gives it the whole tiles of a call that starts on a column
                             // boundary, front_kernel the remainder and everything else
    int32_t cont;            // front_kernel: this launch continues a call whose head another launch has made (the one-shot actions are done)
    int32_t ch_count;        // stage B: workgroups of the launch (0: one per channel of the handle); with ch0:
    int32_t ch0;             // stage B / C: the launch covers the channels ch0 ... ch0 + ch_count - 1 (stage C: + its channel argument - 1) (a batch whose last round of stage-B workgroups
                             // would leave the chip two thirds empty gives that round to a second stream: fmx_api.hip run_call_one); 0 everywhere else
    int32_t host_count1;     // != 0: the demodulator pre-pass takes the reference's myCount (fm-processor.cpp:662) in front of this call from here (the count
                             // + 1) instead of the channel state -- a call made in overlapping pieces (fmx_api.hip: run_call), where the previous piece's
                             // stage B, which keeps the count, may still be running
    int32_t parts, part_tiles;   // stage A, handles that leave the chip empty (one workgroup per channel, few channels): a channel's tiles are
                             // split in time over `parts` workgroups of `part_tiles` tiles each (FMX_P_FRONT_PARTS); parts <= 1: one per channel
};
// what a workgroup of stage A reads of its channel's state when the channel is split in time (the workgroup of the last part rewrites the
// live state while others may not have started): taken by front_pre_kernel in front of the launch
struct FrontSnap { int32_t lo_phase, hist_fmt; float dc_re, dc_im; };

constexpr int DBG_SLOTS = 96;
struct DeviceBuffers {
    float2 *hist;        // [channels * twins][DECIM][A_HIST_COLS]   input history (column layout): raw samples; DC-corrected and mixed ones for channels with an LO
    float2 *dcv_hist;    // [channels * twins][DCV_SAVE] RfDC in front of the 13 columns before the next call's first column, and of that column
    float2 *zring;       // [channels][ring]  front-end output v[j]
    float2 *sring;       // [channels][sring] PSS filter input history
    float2 *dring;       // [channels][dring] de-emphasised, gained stereo @ fmRate
    ChanState *state;
    ChanState *state_tw; // [twins - 1][channels] stage-A state (RfDC, LO phase, history format) of the twins p > 0 (null when twins == 1)
    const ChanParams *params;
    unsigned long long *dbg;   // optional [channels][DBG_SLOTS] diagnostics (null = off): 0-7 per-phase cycles of front_kernel, 8-12 path / round
                               // counters of stage B, 16-27 per-phase cycles of stageb_seg_kernel
    // per-call work arrays of stage B
    float   *w_dem;      // [channel][lin_rows] demodulator output of this call (DEMODULATOR scope tap; input of the RDS path)
    float   *w_cur;      // [channel][lin_rows] pilot phase (currentPilotPhase; input of the RDS path)
    float   *w_diff;     // [channel][lin_rows] L-R difference before the matrix (LR scope tap, with w_dem)
    float2  *w_iq;       // demodulator pre-pass (fmx_demod.hip; allocated when a channel first needs it), 16-row tiles of widx(): limited
                         // (PLL decoder) / unlimited (AM) samples, |z| for the level squelch of the other decoders
    float   *w_osc;      // ... and its output: the demodulator output behind AFC, scaling and squelch
    float2  *gfix;       // [channels][GAIN_FIX_FRAMES] what a gain change adds to the first frames of the call (gain_fix_kernel)
    uint8_t *w_lockm;    // [channel][lockm_stride] stage B as two kernels: the pilot-lock flags of this call, one byte (six samples + the
                         // whole segment's) per thread and segment
    int32_t lockm_stride;
    int32_t prepass;     // != 0: disc_kernel / afc_kernel<true> run as the pre-pass of the fused stage B (launch_demod_fused): only the channels
                         // with a recurrence of their own in the demodulator (PLL / AM decoder, a squelch) are touched, their demodulator output
                         // goes to the 16-row tiles of w_osc (given to them as w_dem), the metaData snapshot of the AFC value is taken there
    int32_t peaks_on;    // != 0: the peak-level meter's maxima are taken (showPeakLevel: a display feed, FMX_P_SCOPE_TAPS)
    int32_t rows_on;     // != 0: the per-call rows w_dem / w_cur are wanted beyond stage B's own hand-over -- a scope tap is kept (FMX_P_SCOPE_TAPS) or a channel decodes RDS
    int32_t prepass_var; // what the pre-pass channels of the handle use: bit 0 the PLL decoder, 1 the AM decoder, 2 the level squelch, 3 a squelch behind another decoder (afc_kernel's variants)
    // PCM tail (fmx_audio.hip)
    const float *tone;   // [TT_BURST] one test-tone burst (the same for every burst: phase restarts at 0)
    float4  *pk_part;    // [channels][pk_tiles] per audio tile: max |L|, |R| of the frames of the tile's first window, then of its second
    float2  *pk_ring;    // [channels][PK_RING] maxima of the finished windows
    int32_t pk_tiles;
    int32_t lin_rows;    // rows per channel of w_dem / w_diff / w_cur; 0 inside the demodulator pre-pass: its arrays are the 16-row tiles of widx()
    // stage A split in time (CallGeom::parts > 1; allocated when first needed)
    float4  *dc_tiles;   // [streams][dc_pitch][2] the RF DC recurrence over tile t of a stream as the map r -> r (1 - u) + a: (u, a.re, a.im, -), [0] as the
                         // channels without an LO compute it (sums), [1] as the ones with an LO do (the recurrence itself)
    int32_t dc_pitch;
    FrontSnap *fsnap;    // [channels] front-end state in front of the call
    float2  *hist_snap;  // [channels][DECIM][A_HIST_COLS] ... and the history,
    float2  *dcv_snap;   // [channels][DCV_SAVE] ... and the saved RfDC boundaries
};
// element (row r, channel ch) of a per-call work array in either layout
__host__ __device__ __forceinline__ size_t tap_idx(const DeviceBuffers &B, int64_t r, int ch, int pitch) {
    return B.lin_rows ? (size_t)ch * (size_t)B.lin_rows + (size_t)r : widx(r, ch, pitch);
}

// ---- the reference's overlap-add filters as block machines (fmx_ola.hip; handles of few channels) ----------------------
constexpr int OLA_MAX_CH = 64;              // FMX_P_FILTER_RESTARTS automatic: handles up to this many channels
constexpr int OLA_MAX_TAPS = 768;           // >= 756 (fmAudioFilter) and 251 (inputFilter)
struct OlaChan { int32_t off, len, inp; int16_t on, conv; };   // this step of one channel: samples [off, off + len) of the call enter the block at inp; conv: the block is complete behind them
struct OlaStep { OlaChan ch[OLA_MAX_CH]; };
// a step as the launchers take it: by value (handles of up to OLA_MAX_CH channels: the receiver's call pays no copy), or as a table in device memory
// (tab != null; larger handles: a batch that switched to the block machines behind a mid-stream filter change, fmx_api.hip)
struct OlaStepRef { OlaStep val; const OlaChan *tab; };
struct OlaBuffers {
    const float2 *src; float2 *dst;         // input / output streams, channel c at + c * stride, sample i of the call at (pos + i) & mask
    int64_t src_stride, dst_stride, src_mask, dst_mask, src_pos, dst_pos;
    float2 *A, *C;                          // [channels][L] FFT_A (block being filled) and FFT_C (result of the last complete block), fft-filters.h:58-62
    float2 *over, *over_new;                // [channels][OLA_MAX_TAPS] Overloop, and where the block transform puts the new one
    const float *taps;                      // [channels][OLA_MAX_TAPS] the kernel of each channel
    int32_t L, degree;                      // NumofSamples = fftSize - degree (fft-filters.cpp:34), degree
};
// pre_kernel's look-back between the tiles of a channel: the tiles' RF DC maps and "published in launch `epoch`" flags (zeroed once; the
// epoch counts the handle's launches, so nothing is ever reset)
// (tickets: a workgroup takes its tile index from its channel's counter -- a tile's workgroup only ever waits for tiles whose workgroups have
// STARTED, whatever order the dispatcher starts them in; the counters run on from launch to launch, ticket_base = what earlier launches handed out)
struct PreLook { float4 *maps; int32_t *flags; int32_t max_tiles, epoch; uint32_t *tickets; uint32_t ticket_base; };
constexpr int PRE_TILE_SAMPLES = 8192;
// (S, O given: the whole call is ONE run of every channel's filter, and the kernel does ola_io_kernel's work itself)
void launch_pre(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, const void *iq, float2 *vbuf, int64_t vstride, int channels, hipStream_t s,
                const OlaStepRef *S, const OlaBuffers *O, const PreLook &LB);
void launch_ola_io(const OlaStepRef &S, const OlaBuffers &O, int channels, int maxlen, hipStream_t s);
void launch_ola_conv(const OlaStepRef &S, const OlaBuffers &O, int channels, hipStream_t s);
// de-emphasis (fm-processor.cpp:594-595) of fm samples [J0, J1) of every channel, in place in `ring`
void launch_deemph(const DeviceBuffers &B, const CallGeom &G, float2 *ring, int channels, hipStream_t s, const OlaStepRef *S, const OlaBuffers *O);

// ---- a batch handle's promotion to the block machines behind a mid-stream filter change (fmx_promote.hip, fmx_api.hip promote)
constexpr int PROMO_TAIL_IN = 3 * (2 * 32768 - 251) + 1024;      // input samples kept per stream: the block in progress, the two blocks in front of it, the filter's length
constexpr int PROMO_TAIL_AU = 3 * (2 * 4096 - AUDIO_TAPS) + 1200; // fm samples of the d ring the audio machine is run over (and the de-emphasis behind it settles in)
// ... and back (demote): once its machines have been quiet for DEMO_QUIET input samples (three blocks: they are the LTI filters again, which the folded FIRs
// reproduce), a promoted batch keeps DEMO_TAIL_IN samples of its streams and runs the folded stage A over them -- the fm-rate ring's entries in flight and
// the filter history a folded call finds --, and takes the d ring's last DEMO_TAIL_AU entries through the de-emphasis the folded stage B applies
constexpr int64_t DEMO_QUIET = 3 * (2 * 32768 - 251) + 4096;
constexpr int DEMO_TAIL_IN = (2 * 32768 - 251) + 4096;
constexpr int DEMO_TAIL_AU = AUDIO_DELAY + C_MAX_TAPS + 2048;
void launch_capture(const void *iq, int fmt, float qs, int64_t stream_stride, int64_t n, int streams, float2 *tail, int64_t tail_cap, int64_t pos, hipStream_t s);
void launch_promo_state(ChanState *st, FrontSnap *snap, int channels, int mode, hipStream_t s);
void launch_promo_hist(float2 *hist, const float2 *u, int64_t u_stride, int64_t len, int r0, int twins, float2 *zring, int ring_mask, int64_t J0,
                       const ChanParams *params, const FrontSet *old_sets, const int32_t *old_set_of, int channels, hipStream_t s);
void launch_promo_inv_deemph(const float2 *dring, int dmask, int64_t J0, int NA, const ChanParams *params, float2 *out, int channels, hipStream_t s);

// ---- RDS path (fmx_rds.hip) -------------------------------------------------------------------
constexpr int RDS_BLK = 32000;              // overlap-add block of the two 32768-pt filters (fft-filters.cpp:34)
constexpr int RDS_PHASE_RING = 131072;      // pilot-phase delay line (>= 64000 + one block + one call)
constexpr int RDS24_RING = 8192;            // 24 kS/s decimator output ring
constexpr int RDS_BITS_CAP = 8192;          // per-channel bit ring (>= 6 s of bits)
constexpr int RDS_SYM_CAP = 1024;           // per-channel ring of the decided symbols (the IQ scope's constellation points)
struct RdsState {                           // rdsDecoder_2 + AGC + Costas state (rds-decoder-2.cpp:44-78)
    float gain, mu, c_freq, c_phase, c_limit;
    int32_t sample_count, skip, prev_bit, nbits;
    float2 sb0, sb1, sb2;
};
constexpr int RDS1_FIR = 21, RDS1_MATCH = 43, RDS1_QUADS = 8;
struct Rds1State {                          // rdsDecoder's Costas (rds-decoder.cpp:40-41) + rdsDecoder_1 (rds-decoder-1.h:47-58)
    float c_freq, c_phase;
    float m1[RDS1_QUADS], m2[RDS1_QUADS];   // sharpFilter memories
    float last_sync_slope, last_sync, last_data;
    int32_t prev_bit;
};
struct Rds3State {                          // rdsDecoder's Costas + rdsDecoder_3 (rds-decoder-3.h:55-70) + the block synchroniser's
    float c_freq, c_phase;                  // state as far as it decides the next resynchronisation (rds-blocksynchronizer.h:93-99)
    float bit_integrator, bit_clk_phase, prev_clk_state;
    int32_t prev_bit, resync_pending, started;
    uint32_t bs_stream; int32_t bs_synced, bs_cur, bs_bits_in_blk, bs_sync_err, bs_blk1;
};
struct RdsBuffers {
    float  *in_blk;      // [ch][32000]     demod samples of the block being filled
    float  *bpreal;      // [ch][2][32000]  real part of the band-pass block results (parity = block index & 1)
    float2 *bp_over;     // [2][ch][768]    band-pass Overloop (paired transforms: parity = block index & 1; one channel per transform: [0] only)
    float2 *hil;         // [ch][2][32000]  Hilbert block results
    float2 *hil_over;    // [2][ch][768]
    float  *phase_ring;  // [ch][RDS_PHASE_RING]
    float2 *U, *V;       // [ch][32768]     FFT scratch
    float2 *rds24;       // [ch][RDS24_RING]
    float2 *mf;          // [rows][pitch]   RDS_1: matched-filter output of the call (sample-major)
    float2 *mfc;         // [ch][mfc_stride] RDS_2: two AGC outputs of the previous call, then the call's matched-filter outputs, AGC'd in place
    float  *mfm;         // [ch][mfc_stride] RDS_2: |matched-filter output| (same indexing)
    int64_t mfc_stride;
    RdsState *state;
    uint8_t *bits;       // [ch][RDS_BITS_CAP]
    float2 *sym;         // [ch][RDS_SYM_CAP] RDS_2: the sample every bit was decided on (rds-decoder-2.cpp:108-114, `*m = r`), index = bit count
    const float2 *S_bp, *S_hil;   // [32768] filter spectra
    const float2 *S_bp_re;        // [32768] spectrum of the band-pass kernel's real part (paired transforms, fmx_rds.hip)
    const float2 *dec_taps;       // [11] rdsDecimator kernel (h/sum, h)
    const float *rrc;             // [45] matched filter
    // RDS_1 (rds-decoder-1.cpp)
    float  *c_ring, *f_ring;      // [ch][RDS24_RING] Re of the Costas output, and of rdsFilter's output
    Rds1State *state1;
    const float *rds1_coef;       // [21] rdsFilter taps, [43] match kernel, [8][A1 A2 B1 B2] sharpFilter, gain
    // RDS_3 (rds-decoder-3.cpp)
    Rds3State *state3;
    const float2 *sincos24;       // [24000] SinCos (rate) table of the bit-clock NCO (sincos.cpp:45-54)
    int32_t pitch;
    // Every channel's RDS path counts the fm samples IT has processed (round 5): the reference's processor runs its block filters, its phase delay
    // line and its decimator only while its decoder is on (fm-processor.cpp:733-754, :551-553), so a channel that switches on later than its
    // neighbours -- or off and on again -- has block boundaries, a /8 phase and filter contents of its own.
    const int64_t *nc0;           // [ch] samples the channel's path had processed in front of this call; < 0: the decoder is off in this call
    int *chlist;                  // [ch] scratch: the channels of one block phase (launch_rds_block)
};
#define C_RDS_PITCH(Rb) ((Rb).pitch)
// modes: bit k = some channel runs RDS_k; h_nc0: the host's copy of RdsBuffers::nc0
void launch_rds(const DeviceBuffers &B, const RdsBuffers &Rb, const CallGeom &G, int C, const int64_t *h_nc0, int modes, hipStream_t s);

void launch_front(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, const void *iq,
                  int channels, hipStream_t s);
// fmx_front4.hip: whole tiles of the call front4_kernel can take (0: none), and its launch over that many
int front4_tiles(const CallGeom &G, const void *iq);
void launch_front4(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, const void *iq, int channels, hipStream_t s);
// fmx_front4lo.hip: the same kernel with complex taps, one channel per workgroup (CallGeom::front4 == 2: some channel of the handle has a local oscillator)
void launch_front4_lo(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, const void *iq, int channels, hipStream_t s);
// The FMX_* environment switches (diagnostics and A/B runs of one build; none is needed by a user): read ONCE, by the first fmx_create of the process --
// nothing on the per-call path asks the environment.
struct EnvSwitches {
    int call_pieces;      // FMX_CALL_PIECES: fm samples per piece of an overlapping call (-1: the handle's setting)
    int call_pieces_ends; // FMX_CALL_PIECES_ENDS: fm samples of such a call's first and last piece (-1: the library's choice)
    std::vector<int> call_pieces_list;   // FMX_CALL_PIECES_LIST=a,b,c: the pieces' fm samples spelt out (a diagnostic)
    int pieces_serial;    // FMX_CALL_PIECES_SERIAL=1: the same pieces one after the other on the caller's stream
    int front_kernel;     // FMX_FRONT_KERNEL: stage-A kernel where the handle says automatic
    int prof_double;      // FMX_PROF_DOUBLE: a throw-away event in front of each profiling event
    int tail_split;       // FMX_TAIL_SPLIT=0: stages B and C as one channel group
    int tail_ch;          // FMX_TAIL_CH=n: channels of the second group
    int stageb_split;     // FMX_STAGEB_SPLIT=0 / 1: stage B as one kernel / two (-1: the round arithmetic)
    int rows_off_split;   // FMX_ROWS_OFF_SPLIT: the round arithmetic also for batches that keep no scope-tap rows
    int no_sinpoly;       // FMX_DEBUG_NO_SINPOLY: the SinCos table from memory
    int host_zerocopy;    // FMX_HOST_ZEROCOPY=0: fmx_process_host through staged copies
    int rds_pair;         // FMX_RDS_PAIR=0: one channel per RDS block transform
};
const EnvSwitches &env_switches();
// first HIP error of the launches / event calls of the current fmx_process_* call (they are enqueued by void helpers);
// run_call clears it before the launches and turns it into FMX_E_HIP behind them
extern thread_local hipError_t g_launch_err;
inline void note_hip(hipError_t e) { if (e != hipSuccess && g_launch_err == hipSuccess) g_launch_err = e; }
#define FMX_LAUNCHED() ::fmx::note_hip(hipGetLastError())
// the demodulators with a recurrence of their own (PLL / AM decoder, squelches), one lane per channel over the whole call, in front of
// the fused kernel and on the same stream (fmx_demod.hip: disc_kernel + afc_kernel<true> with DeviceBuffers::prepass set)
// A call made in overlapping pieces (fmx_api.hip run_call) spreads the pre-pass over streams: disc_kernel (a parallel kernel) behind the piece's stage A,
// the lone-wave recurrences of afc_kernel on a stream of their own (with compute units of their own), nsq_kernel and stage B on `s`.
struct PrepassStreams { hipStream_t s_disc, s_afc; hipEvent_t ev_disc, ev_afc; };
void launch_demod_prepass(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, int channels, hipStream_t s, const PrepassStreams *ps = nullptr);
// stage B as one time-parallel workgroup per channel and segment (fmx_stageb.hip): everything on the caller's stream
void launch_demod_fused(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, int channels, hipStream_t s, const PrepassStreams *ps = nullptr);
void launch_audio(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, float2 *pcm,
                  int channels, hipStream_t s);
// second converter workingRate -> audioRate (fm-processor.cpp:825-838): x48 = [channels][x_stride] with nt history frames in front of
// the call's frames_in 48 kHz frames; writes output frames out0 .. out0 + nout - 1, then moves the history up
void launch_conv2(float2 *x48, int64_t x_stride, const float *taps, int p, int q, int nt, int64_t in0, int64_t frames_in,
                  int64_t out0, int64_t nout, float2 *pcm, int64_t pcm_stride, int channels, hipStream_t s);
// before launch_audio of a call in front of which volume / balance may have changed (and of the first call)
void launch_gain_fix(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, int channels, hipStream_t s);

}  // namespace fmx
