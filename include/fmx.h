/*
 * fmx.h -- C ABI of libfmx: the MI355X (gfx950) implementation of sdr-j-fm's FM processing
 * chain (reference: /root/reference/src/fm/fm-processor.cpp:373-759 and the leaf classes in
 * src/various).  One fmx handle owns N independent FM channels ("fmProcessor" instances) that
 * are demodulated as one batch on one GPU.
 *
 * Each entry point names the reference interface it replaces (file:line relative to the
 * reference tree).  No C++, Qt or torch types cross this boundary: plain pointers and sizes.
 *
 * Threading: exactly one processing thread per handle (the adapter's QThread replacing
 * fmProcessor::run()).  fmx_set_param may be called from any thread; values take effect at the
 * next fmx_process_* call, mirroring the settings mailbox of fm-processor.cpp:396-413.
 * Ownership: the caller owns every in/out buffer; the handle owns all device memory and state.
 * Errors: 0 = FMX_OK, negative = error; text via fmx_last_error().  There is NO CPU fallback:
 * every call fails with FMX_E_NO_DEVICE / FMX_E_HIP when the GPU path is unavailable.
 */
#ifndef FMX_H
#define FMX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: fmx_meta and fmx_rds_info grew (live_rf_dc_*, radio_text_ucs2*), FMX_P_PLL_SOLVER 3, fmx_pll_exact_segments.
 * 3 (round 6): handles above 64 channels keep no display feeds unless asked (FMX_P_SCOPE_TAPS: fmx_get_tap / fmx_get_peaks answer FMX_E_UNSUPPORTED there), round 5's
 * exports (fmx_last_front_kernel, fmx_last_call_pieces, fmx_last_second_group, fmx_last_rds_samples_of ...), FMX_P_FRONT_KERNEL without its value 2.  The output
 * structs are caller-allocated and carry no size field: a caller built against another version must not call in -- the adapters
 * compare fmx_abi_version () with the FMX_ABI_VERSION they were compiled with when they load the library. */
#define FMX_ABI_VERSION 3

typedef struct fmx_handle_s *fmx_handle;

enum {
    FMX_OK = 0,
    FMX_E_INVALID = -1,        /* bad argument / unknown parameter id / out-of-range value */
    FMX_E_UNSUPPORTED = -2,    /* a setting of the reference this build does not implement */
    FMX_E_NO_DEVICE = -3,      /* no HIP device: the product path never falls back to the CPU */
    FMX_E_HIP = -4,            /* HIP runtime error (text in fmx_last_error) */
    FMX_E_NOMEM = -5,
    FMX_E_TOO_LARGE = -6,      /* n > max_block, or pcm capacity too small */
};

/* Construction arguments: replaces the fmProcessor constructor
 * (fm-processor.cpp:48-63; GUI values radio.cpp:231-252,915-930).  Only the DSP-relevant
 * arguments exist here; the scope ring buffers / Qt objects stay in the adapter. */
typedef struct {
    int32_t struct_size;       /* sizeof(fmx_config), for ABI growth */
    int32_t device;            /* HIP device ordinal */
    int32_t channels;          /* number of FM channels in this batch (>=1) */
    int32_t streams;           /* number of distinct IQ input streams; 0 => one per channel */
    const int32_t *stream_of_channel; /* [channels] or NULL (identity); several channels may share a
                                  wide-band stream and differ in set_localOscillator (BASELINE config 3) */
    int32_t inputRate;         /* the device's rate (deviceHandler::getRate, radio.cpp:836): 2304000 (fm-constants.h:35), or any rate the reference
                                  decimates by 12 (2304000 <= inputRate < 3456000), by 6 (1152000 <= inputRate < 2304000: its second decimator
                                  then only filters) or not at all (inputRate / fmRate <= 1: the 192 kS/s devices) -- fm-processor.cpp:68-75,
                                  :471; what the decimation leaves is treated as fmRate, as in the reference.  Other rates: FMX_E_UNSUPPORTED */
    int32_t fmRate;            /* 192000 */
    int32_t workingRate;       /* 48000 */
    int32_t audioRate;         /* 48000 = workingRate: sendSampletoOutput's direct path (:826-829); any other rate in 8000 .. 192000
                                  with audioRate / gcd <= 640 runs the second converter (theConverter :89-91, :830-837; main.cpp:57-65 -m) */
    int32_t max_block;         /* max complex samples per stream per call (reference block = 16384, :374) */
} fmx_config;

/* One id per fmProcessor setter (fm-processor.h:104-156).  Values are passed as double. */
typedef enum {
    FMX_P_FM_MODE = 1,         /* setfmMode: 0 Stereo, 1 StereoPano, 2 Mono          (:241-243)  */
    FMX_P_FM_DECODER = 2,      /* fm_Demodulator::setDecoder: 1 AM 2 PLL 3 Mixed 4 ComplexBB 5 RealBB 6 Diff
                                  (fm-demodulator.cpp:27-44,93-103,215-241) */
    FMX_P_SOUND_MODE = 3,      /* setSoundMode: Channels enum 0..6                   (:273-275)  */
    FMX_P_STEREO_PANORAMA = 4, /* setStereoPanorama 0..200                           (:277-280)  */
    FMX_P_SOUND_BALANCE = 5,   /* setSoundBalance -100..100                          (:282-286)  */
    FMX_P_DEEMPHASIS = 6,      /* setDeemphasis, microseconds >= 1                   (:291-297)  */
    FMX_P_VOLUME_DB = 7,       /* setVolume, dB                                      (:299-301)  */
    FMX_P_LF_CUTOFF = 8,       /* setlfcutoff, Hz; <= 0 switches the audio filter off (:762-770) */
    FMX_P_BANDWIDTH = 9,       /* setBandwidth, Hz (the GUI's "165kHz" -> 165000); 0 = "Off" (:232-239) */
    FMX_P_ATTENUATION_L = 10,  /* setAttenuation (Lgain)                             (:351-359)  */
    FMX_P_ATTENUATION_R = 11,  /* setAttenuation (Rgain)                                         */
    FMX_P_RDS_MODE = 12,       /* setfmRdsSelector: 0 off, 1 = RDS_1 (rds-decoder-1.cpp), 2 = RDS_2 (rds-decoder-2.cpp),
                                  3 = RDS_3 (rds-decoder-3.cpp)                      (:840-847).  Per channel, at any time: a channel's RDS path
                                  (block filters, phase delay line, decimator, slicers) runs while its decoder is on and keeps what it holds while
                                  it is off, as the reference's processor does (:733-754, :551-553) -- nothing is restarted by a switch */
    FMX_P_LOCAL_OSCILLATOR = 13,/* set_localOscillator, Hz                           (:866-868)  */
    FMX_P_AUTO_MONO = 14,      /* setAutoMonoMode                                    (:914-916)  */
    FMX_P_PSS = 15,            /* setPSSMode                                         (:918-920)  */
    FMX_P_DC_REMOVE = 16,      /* setDCRemove (also zeroes RfDC)                     (:922-925)  */
    FMX_P_SQUELCH_MODE = 17,   /* set_squelchMode (fm-processor.cpp:882): 0 OFF, 1 NSQ (noise squelch, squelchClass.cpp:47-87: two order-20
                                  Chebyshev filters around 70 kHz on the demodulator output), 2 LSQ (level squelch, :89-113) */
    FMX_P_TEST_TONE = 18,      /* setTestTone (:931-933): 1 kHz bursts of 25 ms every 2 s at 0.9, programme at 0.1 (:800-823) */
    FMX_P_SQUELCH_VALUE = 19,  /* set_squelchValue 0..100 (:213-215): takes effect at the next call when it differs  */
    FMX_P_DISP_DELAY = 20,     /* setDispDelay (:935-937): steps of the peak-level delay line; applies to the windows
                                  fmx_get_peaks has not handed out yet */
    FMX_P_PLL_SOLVER = 21,     /* how stage B evaluates the pilot PLL loop of pilot-recover.cpp:54-61 (no counterpart in the reference, whose
                                  loop runs sample by sample): 1 = sequentially in every segment -- the reference's f32 trajectory (one thread
                                  per channel walks the recurrence, the table arithmetic is prepared by all); 2 = all samples of a segment at
                                  once by Newton's method (the trajectory to ~1e-5 rad: the loop's own f32 rounding noise, integrated) WHILE
                                  THE PILOT IS COMFORTABLY IN LOCK, sequentially otherwise -- during acquisition and whenever the lock metric
                                  came within 0.05 of its threshold in the previous segment, so that every lock decision
                                  (pilot-recover.cpp:62-80) is taken on the reference's own trajectory: what large batches use;
                                  3 = Newton's method always (diagnostic: lock decisions may then fall a few samples apart from the
                                  reference's when the metric creeps through its threshold); 0 = automatic: 1 up to 64 channels per
                                  handle, 2 above (default).  "The reference's trajectory" to this bound: every evaluation of the loop step,
                                  the sequential one included, takes the NCO sine from the GPU's sine unit, within 1.2e-7 of the
                                  reference's table entry (not that entry bit for bit, and the unit's rounding is this architecture's);
                                  the lock decisions were equal to the oracle's in every tested case, which a metric creeping at 1e-6
                                  per sample through its threshold does not guarantee to the sample. */
    FMX_P_STAGEB_FORM = 22,    /* (handle-wide: the channel argument is ignored) stage B -- limiter .. de-emphasis -- as 1 = one kernel per call,
                                  2 = two kernels (limiter .. lock detector, then PSS .. de-emphasis: four workgroups per CU instead of
                                  three); 0 = automatic (default): whichever wastes less of its last round of workgroups for the
                                  handle's channel count.  The results of the two forms are bit-identical. */
    FMX_P_FILTER_RESTARTS = 23,/* (handle-wide, before the first call) how the two overlap-add filters of the reference -- inputFilter (65536 points, 251
                                  taps, :469-470) and fmAudioFilter (8192 points, 756 taps, :589-591) -- are built: 1 = as the block machines they
                                  are (fft-filters.cpp:33-163): a setBandwidth / setlfcutoff in the middle of a stream then replays the last output
                                  block, drops the block in progress and carries the old tail over, sample for sample as the reference does (any
                                  channel count; about seven times the cost of 2 on a batch); 2 = folded into stage A's / stage C's polyphase FIRs
                                  for good -- the same filters wherever the settings were made before the first call, a different glitch of one
                                  filter latency behind a change in mid-stream; 0 = automatic (default): 1 up to 64 channels; above, folded UNTIL a
                                  filter setter arrives in mid-stream -- the handle then becomes a block-machine handle before the setter takes
                                  effect (fmx_filter_change_due), and the change is the reference's, sample for sample */
    FMX_P_FRONT_PARTS = 24,    /* (handle-wide: the channel argument is ignored) the input-filter stage runs one workgroup per channel; a handle with
                                  fewer channels than the GPU has workgroup slots splits every channel's call in time over several workgroups
                                  (each later one recomputes one 1536-sample tile to get its filter history): 0 = automatic (default), 1 = never,
                                  2..32 = that many parts wherever a call is long enough.  The results are bit-identical. */
    FMX_P_FRONT_KERNEL = 25,   /* (handle-wide: the channel argument is ignored) which kernel runs the input-filter stage.
                                  1 = fmx_front.hip: four waves per channel, the folded filter as packed f32 FMAs; every input format, local
                                  oscillators, any call.
                                  3 = fmx_front4.hip: the filter on the matrix pipe -- samples and taps split into two f16 halves each, their
                                  products exact in the f32 accumulator, the remainders' roundings at 2^-22 of a product: the fm-rate IQ
                                  agrees with kernel 1's to 5e-7 of its amplitude, PCM against the oracle is unchanged.  BLOCK FLOATING
                                  POINT: every 1536-sample tile of a channel is split behind a power-of-two scale of its own (its largest
                                  balanced sample goes to [2^14, 2^15)), so the stage is linear at any level, as the reference's f32 filter
                                  is -- nothing is clamped, weak signals keep 22 bits.  (Limit: two adjacent tiles of one channel whose
                                  largest samples differ by more than 2^126.)  The IQ balance is applied in front of the filter, where the
                                  reference has it (fm-processor.cpp:462-464).
                                  3 takes the whole 1536-sample tiles of the calls it can -- any sample format, the input filter on
                                  everywhere, a balance between 1e-6 and 1e6 in magnitude, a call that starts on a multiple of 12 samples --
                                  and leaves the rest to kernel 1.  A handle with LOCAL OSCILLATORS runs the kernel's complex-tap variant
                                  (the mix folded into a tap set per channel built from the reference's own oscillator table; one channel
                                  per workgroup: automatic for handles with at least one channel per compute unit).  (2 was round 5's six-wave VALU
                                  kernel: measured slower, now tools/experiments/fmx_front3.hip; the value is refused.)
                                  0 = automatic (default): 3 where a handle qualifies and has the channels to fill the GPU, else 1. */
    FMX_P_SCOPE_TAPS = 26,     /* (handle-wide: the channel argument is ignored) whether the DISPLAY FEEDS are produced: the three scope taps that are rows
                                  of stage B's work arrays -- FMX_TAP_DEMOD, FMX_TAP_LR_RAW, FMX_TAP_PILOT_PHASE (fm-processor.cpp:608-613 and the
                                  demodulator / pilot scopes): 12 bytes written per channel and fm sample that nothing else reads (a channel that
                                  decodes RDS keeps its demodulator and pilot rows regardless: the RDS path reads them) -- and the peak-level
                                  meter's maxima (showPeakLevel: fmx_get_peaks).  1 = produced, 0 = not (fmx_get_tap answers FMX_E_UNSUPPORTED for
                                  those taps, fmx_get_peaks likewise; FMX_TAP_FM_IQ, FMX_TAP_PRE_RESAMPLER and FMX_TAP_RDS_IQ are read from rings
                                  and always there), -1 = automatic (default): produced by handles of up to 64 channels -- the receiver with a
                                  display --, not by larger batches.  The PCM does not depend on it.  Takes effect at the next call. */
    FMX_P_CALL_PIECES = 27,    /* (handle-wide: the channel argument is ignored) fm samples per PIECE of a call that is made in overlapping pieces.  A batch
                                  whose channels run the PLL / AM decoder or a squelch (pllC.cpp:67-90, squelchClass.cpp:47-113: recurrences that walk a
                                  channel's samples one after the other, one wave per 64 channels) would leave the GPU idle for most of such a call:
                                  the library cuts the call into pieces -- the chain does not depend on how a stream is cut into calls -- and runs the
                                  input filter of piece k + 1 and the stereo / audio stages of piece k - 1 while the recurrences walk piece k.
                                  -1 = automatic (default): 3840 (AM decoder) or 4608 (PLL decoder alone) with a short last piece (19200 fm samples are cut 4608 4608 4608 3840 1536)
                                  or 4608 (squelches only) for handles of 1024 channels and more, calls of two pieces and more, no RDS decoder on, no second converter;
                                  0 = never; n > 0: pieces of n fm samples (rounded up to 16) for any handle above 64 channels.  Takes effect at the next call. */
    /* actions (value ignored) */
    FMX_A_TRIGGER_FREQUENCY_CHANGE = 100, /* triggerFrequencyChange (:849-855) */
    FMX_A_RESTART_PSS = 101,              /* restartPssAnalyzer     (:857-860) */
    FMX_A_RESET_RDS = 102,                /* resetRds               (:862-864) */
} fmx_param_id;

/* What fmProcessor reports upward: SMetaData (fm-processor.h:91-101, emitted :662-684) */
typedef struct {
    float   DcValRf, DcValIf, PssPhaseShiftDegree, PssPhaseChange;
    int32_t PssState;          /* 0 OFF, 1 ANALYZING, 2 ESTABLISHED */
    float   PilotPllLockStrength;
    int32_t PilotPllLocked;
    int64_t fm_samples;        /* fm-rate samples processed so far */
    int64_t pcm_frames;        /* PCM frames produced so far */
    /* live values (not the 0.5 s snapshot): what isPilotLocked() (:870-880) / get_demodDcComponent() (:221-226) read */
    int32_t live_pilot_locked;
    float   live_lock_strength;
    float   live_dc_if;
    int32_t squelch_active;    /* getSquelchState (:217-219): the level squelch is muting the demodulator output */
    float   live_rf_dc_re, live_rf_dc_im;   /* RfDC (:423-446) behind the last sample processed: what a caller needs to reproduce the
                                               DC-corrected block the reference dumps (:448-455) with the reference's own recurrence */
} fmx_meta;

typedef enum {
    FMX_TAP_FM_IQ = 0,         /* complex @fmRate after fmBand_2 (IF_FILTERED scope, :601-604)   */
    FMX_TAP_DEMOD = 1,         /* float   @fmRate after demodulate (DEMODULATOR scope, :605-607) */
    FMX_TAP_LR_RAW = 2,        /* complex @fmRate (sum,diff) (AF_SUM/AF_DIFF scopes, :608-613)   */
    FMX_TAP_PRE_RESAMPLER = 3, /* complex @fmRate after de-emphasis (:594-595), before gain; this
                                  build applies the audio low-pass AFTER this point */
    FMX_TAP_RDS_IQ = 4,        /* complex @24 kS/s after rdsDecimator (RDS_INPUT scope, :566-569)        */
    FMX_TAP_PILOT_PHASE = 5,   /* float   @fmRate currentPilotPhase (:695; the value the RDS mixer's phase buffer takes, :747) */
} fmx_tap_id;

/* per-kernel timing collected with HIP events on the processing stream */
typedef struct {
    int64_t launches[4];       /* 0 front-end (input FIR), 1 demod/pilot/PSS, 2 audio FIR+resample, 3 reserved */
    double  ms[4];             /* accumulated GPU time of each kernel */
    int64_t input_samples;     /* complex input samples (summed over streams) covered by the timings */
    int64_t channel_samples;   /* complex input samples summed over channels */
} fmx_profile;

int  fmx_abi_version(void);
const char *fmx_last_error(void);

/* replaces `new fmProcessor(...)` (radio.cpp:915-930) */
int  fmx_create(const fmx_config *cfg, fmx_handle *out);
/* replaces fmProcessor::stop() + delete (fm-processor.cpp:200-211) */
int  fmx_destroy(fmx_handle h);

/* replaces the ~25 setters; channel = -1 addresses every channel */
int  fmx_set_param(fmx_handle h, int32_t channel, int32_t param_id, double value);

/* number of PCM frames the next call with n complex samples will produce per channel */
int64_t fmx_frames_for(fmx_handle h, int64_t n_complex);
/* replaces the moment at which the reference's loop takes a setBandwidth / setlfcutoff over (fm-processor.cpp:396-408: the next block start).  A handle
 * that runs its filters as the reference's block machines takes it with its next call: -1.  A BATCH (above 64 channels, FMX_P_FILTER_RESTARTS
 * automatic) runs the filters folded into its polyphase FIRs until a filter setter arrives in mid-stream; it then keeps what its streams deliver for
 * three blocks of the input filter (196 879 samples, 85 ms) and becomes a block-machine handle at the first call boundary behind that, where the setter
 * restarts its filter exactly as the reference's setLowPass does (fft-filters.cpp:84-95).  Returns the input samples per stream still to come before
 * that call (0: the next call applies the change at its first sample), -1 when nothing is pending. */
int64_t fmx_filter_change_due(fmx_handle h);

/* Replaces one iteration of the loop in fmProcessor::run() (fm-processor.cpp:387-686) for every
 * channel: `iq` = what deviceHandler::getSamples (device-handler.h:71-74) delivered, interleaved
 * (I,Q) float32, stream s at iq + 2*s*stream_stride; `pcm` = what is handed to
 * audioSink::putSamples (audiosink.h:45), interleaved (L,R) float32, channel c at
 * pcm + 2*c*pcm_stride.  Host buffers; synchronous. */
int  fmx_process_host(fmx_handle h, const float *iq, int64_t stream_stride, int64_t n_complex,
                      float *pcm, int64_t pcm_stride, int64_t *n_frames);
/* Same with DEVICE pointers (IQ already resident in HBM); asynchronous on `hip_stream`
 * (a hipStream_t, NULL = the handle's own stream, which first waits for the work queued on HIP's default stream -- the
 * stream NULL names -- at the time of the call, so a producer there needs no extra synchronisation; the results are
 * ordered by fmx_synchronize).  *n_frames is known on return. */
int  fmx_process_device(fmx_handle h, const float *d_iq, int64_t stream_stride, int64_t n_complex,
                        float *d_pcm, int64_t pcm_stride, int64_t *n_frames, void *hip_stream);
/* The same two calls for RAW device samples (the conversion the device handlers do on the host moves into
 * the input-FIR kernel, so 2 or 4 bytes per complex sample cross PCIe / HBM instead of 8).  Interleaved (I,Q) pairs of
 *   FMX_IQ_F32  float32, as above
 *   FMX_IQ_U8   uint8,  value = (u8 - 127) / 128        rtlsdr-handler.cpp:291-292
 *   FMX_IQ_S8   int8,   value = s8 / 128                hackrf-handler.cpp:364-365
 *   FMX_IQ_S16  int16,  value = s16 / s16_denominator   lime-handler.cpp:250, pluto-handler.cpp:578,
 *                                                       sdrplay-handler-v3.cpp:261 (2048 or 4096); 32768 for PCM16 files
 * stream_stride stays in COMPLEX SAMPLES; s16_denominator must be a power of two (the division is then exact, as in the
 * reference) and is ignored for the other formats. */
typedef enum { FMX_IQ_F32 = 0, FMX_IQ_U8 = 1, FMX_IQ_S8 = 2, FMX_IQ_S16 = 3 } fmx_iq_format;
int  fmx_process_host_raw(fmx_handle h, const void *iq, int32_t format, float s16_denominator, int64_t stream_stride,
                          int64_t n_complex, float *pcm, int64_t pcm_stride, int64_t *n_frames);
int  fmx_process_device_raw(fmx_handle h, const void *d_iq, int32_t format, float s16_denominator, int64_t stream_stride,
                            int64_t n_complex, float *d_pcm, int64_t pcm_stride, int64_t *n_frames, void *hip_stream);
int  fmx_synchronize(fmx_handle h);

/* replaces the showMetaData signal payload / isPilotLocked / get_demodDcComponent */
int  fmx_get_meta(fmx_handle h, int32_t channel, fmx_meta *meta);
/* replaces the showPeakLevel signal (fm-processor.h:293, evaluatePeakLevel fm-processor.cpp:772-798): every 961 PCM
 * frames the reference emits (leftDb, rightDb) of the window's absolute maxima, behind a display delay line.  Copies
 * the (leftDb, rightDb) pairs of the windows that closed since the last fetch, oldest first, at most `capacity` pairs
 * (the maxima are taken on the GPU next to the audio FIR; dB conversion and delay line run here).  The library keeps
 * the last 256 windows (5 s) per channel. */
int  fmx_get_peaks(fmx_handle h, int32_t channel, float *lr_db, int32_t capacity, int32_t *n_events);
/* replaces the hf/lf/iq scope ring feeds: copies the most recent n samples of a tap
 * (n * 1 or 2 floats) to host memory; n <= samples produced by the last call (fmx_last_fm_samples / fmx_last_rds_samples: while a channel
 * decodes RDS, a call of more than 31999 fm samples -- 383988 input samples at the rates the reference decimates by 12, 191994 at those it
 * decimates by 6, 31999 where it does not decimate -- is made in pieces of that length, and the taps hold the last piece) */
int  fmx_get_tap(fmx_handle h, int32_t channel, int32_t tap_id, float *dst, int64_t n);
/* What the reference's RDS classes tell the GUI through Qt signals (rds-groupdecoder.cpp:44-63, rds-blocksynchronizer.cpp:39-42):
 * setPiCode, setPTYCode, setStationLabel, setRadioText / clearRadioText, setAFDisplay, setMusicSpeechFlag, setGroup,
 * setRDSisSynchronized, setbitErrorRate, setCRCErrors, setSyncErrors.  station_label / radio_text hold the raw RDS characters, NUL
 * terminated (the reference shows the label as QString (stationLabel), rds-groupdecoder.cpp:185-186); radio_text_ucs2 is the text as
 * setRadioText receives it. */
typedef struct fmx_rds_info {
    int32_t synchronized;      /* block synchroniser locked (A..C received without error) */
    int32_t pi_code;           /* block A of the last group; 0 until a group has been decoded */
    int32_t pty_code;          /* programme type 0..31, -1 until known */
    int32_t last_group_type;   /* 0..15, -1 until known */
    int32_t groups_decoded;    /* complete groups so far */
    int32_t crc_errors, sync_errors;
    float   bit_error_rate;
    char    station_label[9];  /* PS name (group 0A), blank padded */
    char    radio_text[65];    /* radio text (group 2A) as the reference would display it (trimmed) */
    int32_t af1_khz, af2_khz;  /* alternative frequencies of the last 0A group, 0 = none */
    int32_t music_speech;      /* -1 unknown, else the M/S flag */
    int32_t di_code;
    /* the radio text as rdsGroupDecoder::prepareText hands it to setRadioText (rds-groupdecoder.cpp:298-315): walked pair by pair with
     * the alphabet-switch pairs 0x0F0F / 0x0E0E / 0x1B6E handled as the reference handles them (the pair's second byte is emitted, the
     * character behind it is not), every character through mapEBUtoUnicode (ebu-codetables.c:65-72), QString::trimmed; UTF-16 code
     * units, 0 terminated */
    uint16_t radio_text_ucs2[65];
    int16_t  radio_text_ucs2_len;
} fmx_rds_info;
/* replaces rdsDecoder::processBit + rdsBlockSynchronizer + rdsGroupDecoder (rds-decoder.cpp:104-131,
 * rds-blocksynchronizer.cpp:114-336, rds-groupdecoder.cpp:100-290) on the host: feeds every bit the slicer has produced
 * since the last call into the channel's block synchroniser / group decoder and returns the current picture.
 * Independent of fmx_rds_bits (own read position). */
int  fmx_rds_decode(fmx_handle h, int32_t channel, fmx_rds_info *info);
/* the same decoder over a caller-supplied bit array, from a fresh state (host only, needs no device): what a
 * recorded bit stream decodes to */
int  fmx_rds_decode_bits(const uint8_t *bits, int32_t n_bits, fmx_rds_info *info);
/* The byte work behind the text signals, exactly as the reference does it (tests/test_rds_text.py pins all three against the
 * reference's own code in oracle/_ref and tests/golden/ref_rds_tables.npz):
 *   fmx_rds_pty_name      pty_table [pty][locale] (ebu-codetables.c:4-37; rds-groupdecoder.cpp:115), UTF-8; NULL outside 0..31 x 0..1
 *   fmx_rds_map_char      mapEBUtoUnicode (alfabet, character) (ebu-codetables.c:65-72)
 *   fmx_rds_prepare_text  rdsGroupDecoder::prepareText (v, length) (:298-315) with alfabetSwitcher / setAlfabetTo (:317-343): writes the
 *                         trimmed text as UTF-16 code units (0 terminated when capacity allows) and returns their count; *alfabet is
 *                         the decoder's theAlfabet (in / out, may be NULL) */
const char *fmx_rds_pty_name(int32_t pty_code, int32_t pty_locale);
uint16_t fmx_rds_map_char(uint8_t alfabet, uint8_t character);
int32_t  fmx_rds_prepare_text(const uint8_t *v, int32_t length, uint8_t *alfabet, uint16_t *out, int32_t capacity);
/* replaces rdsDecoder::doDecode's bit output (rds-decoder.cpp:69-104): pending RDS bits */
int  fmx_rds_bits(fmx_handle h, int32_t channel, uint8_t *bits, int32_t capacity, int32_t *n_bits);
/* replaces doDecode's second output (`*m`, rds-decoder-2.cpp:108-114), the constellation point every bit was decided on, which
 * fmProcessor::run pushes into the IQ scope ring (fm-processor.cpp:555-563): pending symbols as interleaved (I, Q), oldest
 * first, own read position.  RDS_2 only; the library keeps the last 1024. */
int  fmx_rds_symbols(fmx_handle h, int32_t channel, float *iq, int32_t capacity, int32_t *n_symbols);
/* fm-rate samples (inputRate / the reference's decimation: 12, 6 or 1) the last fmx_process_* call produced per channel: the n that fmx_get_tap accepts for the
 * fm-rate taps, and the number of entries the reference's run() pushed into its LF scope vector for the same block.  A call the library made in pieces -- with a
 * channel decoding RDS (pieces of 31999 fm samples) or as overlapping pieces (FMX_P_CALL_PIECES) -- reports its LAST piece here, and the row taps hold that piece. */
int64_t fmx_last_fm_samples(fmx_handle h);
/* Health counter of the pilot PLL (no counterpart in the reference, whose loop is sequential): stage B finds the loop's
 * trajectory of a 1536-sample segment by Newton's method on the whole segment; a segment that does not settle within the
 * round limit is replayed sample by sample by one thread -- correct, only slower.  Returns the number of such segments
 * of `channel` since fmx_create (channel < 0: summed over all channels), or a negative fmx error code. */
int64_t fmx_pll_replays(fmx_handle h, int32_t channel);
/* Segments (of up to 1536 fm samples) that FMX_P_PLL_SOLVER = 2 evaluated sequentially because the pilot was not comfortably in lock,
 * of `channel` since fmx_create (channel < 0: summed over all channels), or a negative fmx error code: what the guard costs. */
int64_t fmx_pll_exact_segments(fmx_handle h, int32_t channel);
/* Which kernel the last fmx_process_* call gave its input-filter stage to (FMX_P_FRONT_KERNEL's numbering: 1, 2 or 3; the remainder of a call
 * that is not whole 1536-sample tiles always goes to kernel 1): what a benchmark names beside its number. */
int32_t fmx_last_front_kernel(fmx_handle h);
/* ... and the number of overlapping pieces it was made in (FMX_P_CALL_PIECES; 1: the call was made whole). */
int32_t fmx_last_call_pieces(fmx_handle h);
/* ... and the channels of the SECOND of the two channel groups its stereo / audio stages ran as (0: one group).  A batch of more channels than one round of
 * the stereo stage's workgroups, without RDS, scope taps or demodulator pre-pass, runs those stages for the channels 0 .. C - n - 1 on the caller's stream and
 * for the last n on a side stream of the handle, so that the audio stage of one group fills what the stereo stage of the other leaves free; the input
 * filter stage stays one launch.  The results do not depend on it (every channel's arithmetic is its own); FMX_TAIL_SPLIT=0 in the environment switches it off. */
int32_t fmx_last_second_group(fmx_handle h);
/* ... and 24 kS/s RDS samples (rdsDecimator outputs, fm-processor.cpp:553): the n that fmx_get_tap accepts for FMX_TAP_RDS_IQ */
int64_t fmx_last_rds_samples(fmx_handle h);
/* ... of one channel.  A channel's RDS path counts the fm samples IT has processed -- it runs while the channel's decoder is on and stands still
 * otherwise, as a processor of the reference leaves its block filters, its phase delay line and its decimator alone while rdsModus is RDS_OFF
 * (fm-processor.cpp:733-754, :551-553) -- so channels that switched their decoders on at different times divide by eight on different phases
 * and a call may give one of them an output more than another.  fmx_last_rds_samples is this for channel 0. */
int64_t fmx_last_rds_samples_of(fmx_handle h, int32_t channel);

/* introspection used by the parity tests: the filter taps the kernels run with.
 * which: 0 front-end polyphase taps, 1 PSS low-pass, 2 audio+resampler FIR, 3 resampler alone,
 * 4 the noise squelch's two order-20 filters: [2][10] x (A1, A2, B1, B2), then the two gains (high-pass first),
 * 5 the RDS_1 decoder's constants: rdsFilter taps [21], matched filter [43], band-pass [8] x (A1, A2, B1, B2), gain */
int  fmx_get_taps(fmx_handle h, int32_t channel, int32_t which, float *dst, int32_t capacity, int32_t *n);

int  fmx_profile_enable(fmx_handle h, int32_t on);
int  fmx_profile_read(fmx_handle h, fmx_profile *out, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif
