// fmx_api.hip -- the C ABI of libfmx (include/fmx.h): handle, settings mailbox, table/tap
// design + upload, per-call geometry and the three kernel launches.  No CPU fallback exists:
// every entry point fails when HIP is unavailable.
#include "../../include/fmx.h"
#include "../../include/fmx_debug.h"
#include "fmx_internal.h"
#include "fmx_fftconv.h"
#include "fmx_rdsgroups.h"
#include "fmx_design.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

using namespace fmx;

namespace {

thread_local std::string g_err;
int env_int(const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; }
int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(FMX_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

int next_pow2(int64_t v) { int64_t p = 1; while (p < v) p <<= 1; return (int)p; }

// per-channel user-level settings (what the fmProcessor setters store)
struct ChanUser {
    int32_t bandwidth = 0;       // 0 = Off.  NB ctor default: inputFilterOn=false (fm-processor.cpp:149)
    int32_t lf_cutoff = 0;       // <=0 = off. ctor default fmAudioFilterActive=false (:164)
    int32_t bw_applied = 0, lf_applied = 0;    // what a FOLDED handle's tap sets are built from: the setter's value -- except while a mid-stream change of a batch
                                 // is pending (fmx_handle_s::promo_pending): then the values in force when the setter arrived
    bool bw_event = false, lf_event = false;   // setBandwidth / setlfcutoff CALLED with a value since the last call: the reference sets newInputFilter /
                                 // newAudioFilter whatever the value (:232-239, :762-770), and its loop restarts the filter's block (:396-408)
    int32_t deemph_us = 0;       // 0 = ctor default alpha (:174)
    bool    ctor_volume = true;  // volumeFactor = 0.5f (:127) until setVolume
    float   volume_db = 0.f;
    int32_t balance = 0, panorama = 100;
    int32_t squelch_value = 0, squelch_old = 0, squelch_level = 1;   // fm-processor.cpp:194-195; mySquelch (1, ...) :87
    // peak meter, host half: DelayLine<DSPCOMPLEX> (fm-processor.h:54-75) over the dB pairs, windows already fetched
    std::vector<float2> delay = std::vector<float2>(1, make_float2(-40.0f, -40.0f));
    uint32_t delay_idx = 0;
    int32_t pk_read = 0;
};

struct ProfRec { hipEvent_t e[4]; int64_t in_samples, ch_samples; int n; };

}  // namespace

struct fmx_handle_s {
    fmx_config cfg{};
    int channels = 0, streams = 0;
    bool streams_private = false;                                    // no two channels listen to the same stream
    hipStream_t stream = nullptr;
    int n_cus = 256; size_t lds_per_block = 65536;     // device limits the stage-A layout choice looks at
    std::vector<int32_t> act_up;                                     // one-shot action bits uploaded with the last parameter upload
    std::vector<uint8_t> rds_reset_req;                              // resetRds / triggerFrequencyChange asked for the group decoder's reset
    hipEvent_t ev_in = nullptr, ev_dummy = nullptr;
    std::mutex mtx;                          // guards the mailbox (set_param from any thread)
    std::vector<ChanUser> user;
    std::vector<ChanParams> params;          // host mirror
    bool params_dirty = true, sets_dirty = true;
    bool gain_dirty = true;                                          // volume / balance set since the last call (and before the first); under mtx
    bool gain_pending = false;                                       // ... taken over by flush_mailbox together with the settings themselves (processing
                                                                     // thread only): gain_fix_kernel runs in the first call that produces frames
    float *d_audio_lp = nullptr, *d_rs_taps = nullptr;
    float2 *d_audio_spec = nullptr;               // the audio low-pass and the resampler alone (gain_fix_kernel)
    // unique tap sets
    std::vector<int32_t> front_keys, audio_keys;
    int front_cap = 0, audio_cap = 0;
    std::vector<float> h_front_taps, h_audio_taps, h_pss_taps, h_rs_taps, h_nsq, h_rds1;
    float *d_nsq = nullptr;
    std::vector<FrontSet> h_front_sets; std::vector<AudioSet> h_audio_sets;
    // device
    float *d_front_taps = nullptr, *d_audio_taps = nullptr, *d_pss_taps = nullptr;
    float2 *d_fft_w = nullptr, *d_pss_hs = nullptr;
    FrontSet *d_front_sets = nullptr; AudioSet *d_audio_sets = nullptr;
    float2 *d_sincos = nullptr, *d_lo = nullptr; float *d_atan = nullptr, *d_arcsine = nullptr; double2 *d_trig3 = nullptr;
    ChanParams *d_params = nullptr;
    DeviceTables T{}; DeviceBuffers B{};
    int ring = 0, dring = 0, sring = 0;
    int decim = DECIM, twins = 1;            // the reference's total decimation for cfg.inputRate (12, 6 or 1) and 12 / decim (CallGeom::twins)
    int64_t g_total = 0;                     // input samples consumed per stream
    int64_t work_nj = 0;                     // rows of the sample-major work arrays
    int pitch = 0;                           // their row pitch in elements
    // staging for the host-pointer entry point
    float2 *d_iq = nullptr, *d_pcm = nullptr; int64_t pcm_cap = 0;
    // second converter (audioRate != workingRate, fm-processor.cpp:89-91,825-838)
    int cv_p = 1, cv_q = 1, cv_nt = 0; float *d_cv_taps = nullptr; float2 *d_x48 = nullptr; int64_t x48_stride = 0;
    // profiling
    bool prof_on = false; std::vector<ProfRec> prof; fmx_profile prof_acc{};
    int64_t last_J0 = 0, last_J1 = 0;
    // RDS path (allocated when a channel first switches RDS on)
    bool rds_alloc = false; RdsBuffers R{};
    // per channel: the fm samples its RDS path has processed (it stands still while the channel's decoder is off, as the reference's
    // processor leaves its block filters, phase delay line and decimator alone then: fm-processor.cpp:733-754, :551-553), what RdsBuffers::nc0
    // was in the last call, and the device copy of the latter
    std::vector<int64_t> rds_nc, rds_nc0;
    int64_t *d_rds_nc0 = nullptr;
    std::vector<int32_t> rds_read;          // per channel: bits already handed out by fmx_rds_bits
    std::vector<int32_t> rds_read_dec;      // ... and by fmx_rds_decode
    std::vector<int32_t> rds_read_sym;      // ... symbols handed out by fmx_rds_symbols
    std::atomic<int32_t> rds_gen{0};        // (rounds 2-4 counted restarts of the RDS path here; nothing restarts it any more) a consumer that saw an older generation starts over (its own read position and,
    std::vector<int32_t> rds_gen_dec, rds_gen_sym;   // for fmx_rds_decode, the channel's block synchroniser / group decoder), however late it polls
    std::vector<fmx::RdsGroupDecoderHost> rds_dec;
    std::vector<int64_t> last_m0, last_m1;  // per channel: its 24 kS/s outputs of the last call
    std::vector<void *> rds_ptrs, tail_ptrs;
    // what a call needs of `params` beyond the device copy, taken under `mtx` by flush_mailbox: fmx_set_param may write `params` from another thread
    // while the processing thread enqueues the call
    std::vector<int8_t> call_rds_mode; bool call_any_rds = false;
    std::atomic<int> stageb_form{0};     // FMX_P_STAGEB_FORM
    std::atomic<int> front_parts{0};     // FMX_P_FRONT_PARTS
    std::atomic<int> front_kernel{0};    // FMX_P_FRONT_KERNEL
    std::atomic<int> scope_taps{-1};     // FMX_P_SCOPE_TAPS
    bool taps_kept = true;               // the last call kept the scope-tap rows (fmx_get_tap)
    float *w_diff_mem = nullptr;         // the LR scope tap's rows (allocated when first wanted; DeviceBuffers::w_diff is null while the tap is off)
    bool front4_ok = false;              // every channel qualifies for front4_kernel (flush_mailbox)
    bool front4_lo = false;              // ... and some channel has a local oscillator: the complex-tap variant (fmx_front4lo.hip)
    std::vector<std::pair<int32_t, int32_t>> hlo_key;   // per channel: the (bandwidth, oscillator) its ChanParams::hlo_* were computed for
    int last_front_kernel = 1;           // what the last call's stage A was given (FMX_P_FRONT_KERNEL numbering): fmx_last_front_kernel
    // a batch call whose channels need the demodulator pre-pass (a lone wave per 64 channels walking the call sample by sample: 6 % of the chip for
    // most of the call's time) is made in pieces whose stages overlap: run_call
    hipStream_t pipe_sA = nullptr, pipe_sP = nullptr, pipe_sB = nullptr;   // stage A (and the parallel part of the pre-pass) of piece k + 1, the recurrences of piece k, stage B / C of piece k - 1
    hipEvent_t pipe_ev0 = nullptr, pipe_evA = nullptr, pipe_evD = nullptr, pipe_evP = nullptr, pipe_evE = nullptr, pipe_evB[2] = {nullptr, nullptr};
    std::atomic<int> pipe_rows{-1};      // FMX_P_CALL_PIECES: fm samples per piece (-1 automatic, 0 never)
    int last_pieces = 1;                 // overlapping pieces the last call was made in (fmx_last_call_pieces)
    int last_second_group = 0;           // channels of the second stage-B / C group of the last call (fmx_last_second_group)
    int my_count_host = 0;               // the reference's myCount (fm-processor.cpp:662-684) as stage B keeps it in every channel's state: the same in all of them
    PreLook pre_look{};                  // pre_kernel's look-back buffers (ensure_ola)
    void *hp_iq = nullptr; float2 *hp_pcm = nullptr; size_t hp_iq_bytes = 0; int64_t hp_pcm_cap = 0;   // fmx_process_host: pinned, device-visible staging of small calls
    // the reference's two overlap-add filters as the block machines they are (fmx_ola.hip): handles of up to OLA_MAX_CH channels
    bool ola_mode = false;               // FMX_P_FILTER_RESTARTS resolved (fixed once the first call has been made)
    struct OlaSide {
        int L = 0, degree = 0;
        float2 *A = nullptr, *C = nullptr, *over = nullptr, *over_new = nullptr; float *taps = nullptr;
        std::vector<int32_t> inp, on, key;   // per channel: block position, Pass () in use, the setting the kernel was designed for (-1: none yet)
    } ola_in, ola_au;
    float2 *d_v = nullptr, *d_u = nullptr, *d2ring = nullptr;   // pre_kernel's output, the input filter's output ([channels][max_block]), the audio filter's output ring
    // A batch (folded filters) whose filters are changed in mid-stream: the setter stays pending while the library keeps the streams' samples (fmx_promote.hip),
    // then the handle switches to the block machines (promote).  FMX_P_FILTER_RESTARTS = 2 pins the folded filters (the change then applies at once, as
    // rounds 1-5 applied it: a different glitch of one filter latency).
    bool folded_pinned = false;          // FMX_P_FILTER_RESTARTS = 2 was asked for
    bool promoted = false;               // a block-machine handle by promotion: goes back to the folded filters once its machines have been quiet (demote)
    bool demo_capture = false;           // ... and is keeping its streams for that
    int64_t last_filter_event_g = 0;     // stream position at which a filter setter was last applied (ola_take_settings)
    std::vector<int64_t> origin_in, origin_au;   // per channel: the stream position (input samples / fm samples) at which the filter's block counter was last 0, as far as a
                                         // FOLDED handle knows it (0 from fmx_create; the machines' own counters at a demotion)
    bool call_head = true;               // the launch sequence being enqueued is the first of its fmx_process_* call (a call may be made in pieces, run_call): a handle
                                         // changes its filter structure there only -- fmx_filter_change_due speaks of CALLS
    bool promo_pending = false;          // a filter setter arrived behind the first call
    bool promo_recapture = false;        // ... and a setter of what pre_kernel applies (RF DC removal, balance, oscillator) behind it: the kept samples start over
    int64_t promo_have = 0, promo_g0 = 0;   // samples kept per stream, and the stream position of the first
    float2 *tail_iq = nullptr; int64_t tail_cap = 0; FrontSnap *tail_snap = nullptr;
    size_t old_sets_cap = 0;
    ChanParams *d_params_replay = nullptr; FrontSet *d_old_sets = nullptr; int32_t *d_old_set_of = nullptr; float2 *d_au_tail = nullptr;
    // the block machines' steps of a handle above OLA_MAX_CH channels: tables in device memory (the launchers take a step by value up to 64 channels),
    // a ring of STEP_SLOTS tables, each copied from pinned host memory on the call's stream in front of the kernels that read it
    static constexpr int STEP_SLOTS = 32;
    OlaChan *step_dev = nullptr, *step_host = nullptr; hipEvent_t step_ev[STEP_SLOTS] = {}; bool step_used[STEP_SLOTS] = {}; int step_next = 0;
};

namespace {

// ---- tap-set construction -----------------------------------------------------------------
// Front end = inputFilter (optional) * fmBand_1 * fmBand_2 folded into one real polyphase FIR evaluated by stage A's /12 kernel.
// `decim` = the reference's total decimation at this input rate (12: both decimators; 6: fmBand_2 does not decimate; 1: no decimators
// at all, fm-processor.cpp:471), `twins` = 12 / decim, `p` = which of the twins this set is for: twin p computes the outputs
// j = twins * m + p, whose newest input decim * j + decim - 1 - L = 12 (m - delay_p) + off_p.  Returns delay_p.
int build_front_set(int32_t bw, int32_t inputRate, int32_t fmRate, int decim, int p, float *taps /*A_TAPS_STRIDE*/, FrontSet *fs) {
    std::vector<double> g(1, 1.0);
    double S1 = 0.0, S2 = 0.0;
    if (decim > 1) {
        const int32_t IRate = inputRate / 6;                                       // fm-processor.cpp:36
        design::DecimKernel k1 = design::decim(4 * inputRate / IRate + 1, fmRate / 2, inputRate);   // :68-71
        design::DecimKernel k2 = design::decim(IRate / fmRate + 1, fmRate / 2, IRate);              // :72-75 (decimation IRate / fmRate: 2 or 1)
        const int D1 = inputRate / IRate;                                          // 6
        // y[m] = sum k1[l] x[6m+5-l]; z[j] = sum k2[i] y[D2 j + D2 - 1 - i]  (fir-filters.cpp:397-424)
        //  => z[j] = sum_k g[k] x[decim j + decim - 1 - k],  g[k] = sum_{D1*i+l=k} h2[i] h1[l]
        g.assign((size_t)(D1 * (k2.hn.size() - 1) + k1.hn.size()), 0.0);
        for (size_t i = 0; i < k2.hn.size(); i++)
            for (size_t l = 0; l < k1.hn.size(); l++) g[D1 * i + l] += (double)k2.hn[i] * (double)k1.hn[l];
        S1 = k1.sum; S2 = k2.sum;
    }
    int L = 0;                                                                 // overlap-add latency in input samples
    if (bw > 0) {
        // inputFilter.setLowPass(fmBandwidth / 2, inputRate), fftFilter(2*32768, 251) (:77,398)
        std::vector<float> hin = design::lowpass(251, bw / 2, inputRate);
        g = design::convolve(g, design::to_double(hin));
        L = 2 * 32768 - 251;                                                   // NumofSamples fft-filters.cpp:34
    }
    const int NT = (int)g.size();
    // newest input of this twin's output column m: 12 (m - delay) + off
    const int64_t e = (int64_t)decim * p + decim - 1 - L;
    int64_t delay = -(e >= 0 ? e / 12 : -((-e + 11) / 12));
    const int off = (int)(e + 12 * delay);
    fs->off = off; fs->delay_fm = (int32_t)delay; fs->zshift = 0;
    int nd = 0;
    std::fill(taps, taps + A_TAPS_STRIDE, 0.f);
    for (int d = 0; d < A_MAX_ND; d++)
        for (int r = 0; r < DECIM; r++) {
            const int k = 12 * d + off - r;
            if (k >= 0 && k < NT) { taps[(d + 1) * DECIM + r] = (float)g[k]; nd = d + 1; }
        }
    fs->nd = nd;
    // complex gain of the (h/sum, h) kernels: (1 + j S1)(1 + j S2)  (fir-filters.cpp:345-346); 1 without the decimators
    fs->gain_re = (float)(1.0 - S1 * S2); fs->gain_im = (float)(S1 + S2);
    // the FIR's response to the slowly moving RfDC value: sum of the taps (as the kernel sums them: f32 taps) times the value at
    // the taps' centre of mass.  Output column q has its newest input at sample 12 q + off; "RfDC applied to sample s" is the state
    // behind s, interpolated between the column boundaries: boundary column q - dc_k, weight dc_w towards the next boundary.
    double hs = 0, mk = 0;
    for (int d = 0; d < A_MAX_ND; d++)
        for (int r = 0; r < DECIM; r++) {
            const double v = (double)taps[(d + 1) * DECIM + r];
            hs += v; mk += v * (double)(12 * d + off - r);
        }
    fs->hsum = (float)hs;
    const double q0 = (double)off - mk / hs + 1.0;
    const int K = -(int)std::floor(q0 / 12.0);
    fs->dc_k = K; fs->dc_w = (float)((q0 + 12.0 * K) / 12.0);
    return (int)delay;
}

void build_audio_set(int32_t lf, int32_t fmRate, const std::vector<float> &rs, float *taps /*C_TAPS_STRIDE*/, AudioSet *as) {
    std::vector<double> g = design::to_double(rs);
    as->delay = 0;
    if (lf > 0) {
        // fmAudioFilter.setLowPass(lowPassFrequency, fmRate), fftFilter(2*4096, 756) (:76,404)
        std::vector<float> ha = design::lowpass(AUDIO_TAPS, lf, fmRate);
        g = design::convolve(design::to_double(ha), g);
        as->delay = AUDIO_DELAY;
    }
    as->ntaps = (int)g.size();
    std::fill(taps, taps + C_TAPS_STRIDE, 0.f);
    for (int kk = 0; kk < as->ntaps; kk++) taps[kk] = (float)g[as->ntaps - 1 - kk];   // reversed
}

int ensure_sets(fmx_handle h) {
    // deduplicate bandwidth / lf-cutoff values into tap sets, upload when changed
    std::vector<int32_t> fk, ak;
    for (int c = 0; c < h->channels; c++) {
        int32_t b = h->user[c].bw_applied, l = h->user[c].lf_applied > 0 ? h->user[c].lf_applied : 0;      // (folded handles: see ChanUser)
        if (h->ola_mode) { b = 0; l = 0; }      // (the two filters run as block machines in front of stage A's decimators / stage C's resampler)
        auto it = std::find(fk.begin(), fk.end(), b);
        if (it == fk.end()) { fk.push_back(b); it = fk.end() - 1; }
        h->params[c].front_set = (int)(it - fk.begin()) * h->twins;          // (index of the channel's first twin set)
        auto ia = std::find(ak.begin(), ak.end(), l);
        if (ia == ak.end()) { ak.push_back(l); ia = ak.end() - 1; }
        h->params[c].audio_set = (int)(ia - ak.begin());
    }
    if (fk != h->front_keys) {
        h->front_keys = fk;
        const size_t TW = (size_t)h->twins, NS = fk.size() * TW;              // one tap set per bandwidth value and twin
        h->h_front_taps.assign(NS * A_TAPS_STRIDE, 0.f);
        h->h_front_sets.resize(NS);
        for (size_t i = 0; i < fk.size(); i++) {
            int dmin = 0x7fffffff;
            std::vector<int> dl(TW);
            for (size_t p = 0; p < TW; p++) {
                dl[p] = build_front_set(fk[i], h->cfg.inputRate, h->cfg.fmRate, h->decim, (int)p, &h->h_front_taps[(i * TW + p) * A_TAPS_STRIDE], &h->h_front_sets[i * TW + p]);
                dmin = std::min(dmin, dl[p]);
            }
            // fm sample j sits at ring position j - delay_fm: twin p's column m is j = TW (m + delay_p) + p
            for (size_t p = 0; p < TW; p++) { h->h_front_sets[i * TW + p].zshift = dl[p] - dmin; h->h_front_sets[i * TW + p].delay_fm = (int32_t)(TW * (size_t)dmin); }
        }
        if ((int)NS > h->front_cap) {
            if (h->d_front_taps) { (void)hipFree(h->d_front_taps); (void)hipFree(h->d_front_sets); }
            h->front_cap = std::max<int>((int)NS, 4);
            HIPCHK(hipMalloc(&h->d_front_taps, sizeof(float) * A_TAPS_DEV * h->front_cap));
            HIPCHK(hipMalloc(&h->d_front_sets, sizeof(FrontSet) * h->front_cap));
        }
        HIPCHK(hipDeviceSynchronize());
        std::vector<float> dev(NS * A_TAPS_DEV, 0.f);                    // device image: [set][r][d]
        for (size_t i = 0; i < NS; i++)
            for (int d = 0; d < A_MAX_ND; d++)
                for (int r = 0; r < DECIM; r++)
                    dev[i * A_TAPS_DEV + r * A_TAPS_ROW + d] = h->h_front_taps[i * A_TAPS_STRIDE + (d + 1) * DECIM + r];
        HIPCHK(hipMemcpy(h->d_front_taps, dev.data(), sizeof(float) * dev.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_front_sets, h->h_front_sets.data(), sizeof(FrontSet) * NS, hipMemcpyHostToDevice));
        h->T.front_taps = h->d_front_taps; h->T.front_sets = h->d_front_sets;
    }
    if (ak != h->audio_keys) {
        h->audio_keys = ak;
        h->h_audio_taps.assign(ak.size() * C_TAPS_STRIDE, 0.f);
        h->h_audio_sets.resize(ak.size());
        for (size_t i = 0; i < ak.size(); i++)
            build_audio_set(ak[i], h->cfg.fmRate, h->h_rs_taps, &h->h_audio_taps[i * C_TAPS_STRIDE], &h->h_audio_sets[i]);
        if ((int)ak.size() > h->audio_cap) {
            if (h->d_audio_taps) { (void)hipFree(h->d_audio_taps); (void)hipFree(h->d_audio_sets); }
            h->audio_cap = std::max<int>((int)ak.size(), 4);
            HIPCHK(hipMalloc(&h->d_audio_taps, sizeof(float) * C_TAPS_STRIDE * h->audio_cap));
            HIPCHK(hipMalloc(&h->d_audio_sets, sizeof(AudioSet) * h->audio_cap));
        }
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(h->d_audio_taps, h->h_audio_taps.data(), sizeof(float) * h->h_audio_taps.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_audio_sets, h->h_audio_sets.data(), sizeof(AudioSet) * ak.size(), hipMemcpyHostToDevice));
        h->T.audio_taps = h->d_audio_taps; h->T.audio_sets = h->d_audio_sets;
        {   // the two factors of the folded FIR, separately (gain_fix_kernel)
            std::vector<float> lp(ak.size() * AUDIO_TAPS, 0.f);
            for (size_t i = 0; i < ak.size(); i++) if (ak[i] > 0) { const std::vector<float> ha = design::lowpass(AUDIO_TAPS, ak[i], h->cfg.fmRate); std::copy(ha.begin(), ha.end(), lp.begin() + i * AUDIO_TAPS); }
            if (h->d_audio_lp) (void)hipFree(h->d_audio_lp);
            HIPCHK(hipMalloc(&h->d_audio_lp, sizeof(float) * lp.size()));
            HIPCHK(hipMemcpy(h->d_audio_lp, lp.data(), sizeof(float) * lp.size(), hipMemcpyHostToDevice));
            if (!h->d_rs_taps) { HIPCHK(hipMalloc(&h->d_rs_taps, sizeof(float) * RS_TAPS)); HIPCHK(hipMemcpy(h->d_rs_taps, h->h_rs_taps.data(), sizeof(float) * RS_TAPS, hipMemcpyHostToDevice)); }
            h->T.audio_lp_taps = h->d_audio_lp; h->T.rs_taps = h->d_rs_taps;
            // spectra of the four decimation phases g_p[i] = g[4 i + p] of the folded FIR (audio_fft_kernel); h_audio_taps holds g reversed
            std::vector<float2> W(fftc::W_COUNT), spec(ak.size() * 4 * fftc::N);
            fftc::make_twiddles(W.data());
            for (size_t i = 0; i < ak.size(); i++) {
                const int nt = h->h_audio_sets[i].ntaps;
                const float *rev = &h->h_audio_taps[i * C_TAPS_STRIDE];
                for (int p = 0; p < 4; p++) {
                    std::vector<float> gp;
                    for (int kk = p; kk < nt; kk += 4) gp.push_back(rev[nt - 1 - kk]);
                    fftc::make_spectrum(gp.data(), (int)gp.size(), &spec[(i * 4 + p) * fftc::N], W.data());
                }
            }
            if (h->d_audio_spec) (void)hipFree(h->d_audio_spec);
            HIPCHK(hipMalloc(&h->d_audio_spec, sizeof(float2) * spec.size()));
            HIPCHK(hipMemcpy(h->d_audio_spec, spec.data(), sizeof(float2) * spec.size(), hipMemcpyHostToDevice));
            h->T.audio_spec = h->d_audio_spec;
        }
    }
    h->sets_dirty = false;
    return FMX_OK;
}

void refresh_derived(fmx_handle h, int c) {
    // the arithmetic of the setters themselves
    const ChanUser &u = h->user[c];
    ChanParams &p = h->params[c];
    const int32_t fmRate = h->cfg.fmRate;
    if (u.deemph_us >= 1) {                     // setDeemphasis fm-processor.cpp:291-297 (Tau is float)
        const float Tau = (float)(1000000.0 / u.deemph_us);
        p.deemph_alpha = (float)(1.0 / ((double)((float)fmRate / Tau) + 1.0));
    } else {                                    // ctor :174
        p.deemph_alpha = (float)(1.0 / (fmRate / (1000000.0 / 50.0 + 1)));
    }
    p.deemph_l2 = (float)std::log2((double)(1.0f - p.deemph_alpha));
    p.volume = u.ctor_volume ? 0.5f : std::pow(10.0f, u.volume_db / 20.0f);        // :127, :299-301
    p.left_ch = (u.balance > 0 ? (float)((100 - u.balance) / 100.0) : 1.0f);       // :282-286
    p.right_ch = (u.balance < 0 ? (float)((100 + u.balance) / 100.0) : 1.0f);
    p.panorama = (float)(int16_t)u.panorama / 100.0f;                               // :277-280
}

int ensure_lo_table(fmx_handle h) {
    if (h->d_lo) return FMX_OK;
    const int32_t R = h->cfg.inputRate;
    std::vector<float2> tab((size_t)R);
    for (int32_t i = 0; i < R; i++)             // Oscillator ctor oscillator.cpp:26-35
        tab[i] = make_float2((float)std::cos(2.0 * design::kPi * i / R), (float)std::sin(2.0 * design::kPi * i / R));
    HIPCHK(hipMalloc(&h->d_lo, sizeof(float2) * (size_t)R));
    HIPCHK(hipMemcpy(h->d_lo, tab.data(), sizeof(float2) * (size_t)R, hipMemcpyHostToDevice));
    h->T.lo_table = h->d_lo;
    return FMX_OK;
}

int ensure_rds_body(fmx_handle h);
int ensure_rds(fmx_handle h) {
    if (h->rds_alloc) return FMX_OK;
    const int rc = ensure_rds_body(h);
    if (rc) {                                       // partial failure: give everything back, the next enable tries again
        for (void *p : h->rds_ptrs) if (p) (void)hipFree(p);
        h->rds_ptrs.clear(); h->R = RdsBuffers{};
    }
    return rc;
}
int ensure_rds_body(fmx_handle h) {
    const size_t C = (size_t)h->channels;
    const int32_t fmRate = h->cfg.fmRate;
    auto dalloc = [&](void **p, size_t bytes, bool zero) -> int {
        HIPCHK(hipMalloc(p, bytes));
        if (zero) HIPCHK(hipMemset(*p, 0, bytes));
        h->rds_ptrs.push_back(*p);
        return FMX_OK;
    };
    RdsBuffers &R = h->R;
    int rc;
    if ((rc = dalloc((void **)&R.in_blk, sizeof(float) * C * RDS_BLK, true))) return rc;
    if ((rc = dalloc((void **)&R.bpreal, sizeof(float) * C * 2 * RDS_BLK, true))) return rc;
    if ((rc = dalloc((void **)&R.bp_over, sizeof(float2) * 2 * C * 768, true))) return rc;
    if ((rc = dalloc((void **)&R.hil, sizeof(float2) * C * 2 * RDS_BLK, true))) return rc;
    if ((rc = dalloc((void **)&R.hil_over, sizeof(float2) * 2 * C * 768, true))) return rc;
    if ((rc = dalloc((void **)&R.phase_ring, sizeof(float) * C * RDS_PHASE_RING, true))) return rc;
    if ((rc = dalloc((void **)&R.U, sizeof(float2) * C * 32768, false))) return rc;
    if ((rc = dalloc((void **)&R.V, sizeof(float2) * C * 32768, false))) return rc;
    if ((rc = dalloc((void **)&R.rds24, sizeof(float2) * C * RDS24_RING, true))) return rc;
    R.pitch = h->pitch;
    if ((rc = dalloc((void **)&R.mf, sizeof(float2) * (size_t)(h->work_nj / 8 + 8) * h->pitch, false))) return rc;
    R.mfc_stride = ((h->work_nj / 8 + 8 + 2 + 7) / 8) * 8;
    if ((rc = dalloc((void **)&R.mfc, sizeof(float2) * (size_t)R.mfc_stride * C, true))) return rc;
    if ((rc = dalloc((void **)&R.mfm, sizeof(float) * (size_t)R.mfc_stride * C, true))) return rc;
    if ((rc = dalloc((void **)&R.bits, C * RDS_BITS_CAP, true))) return rc;
    if ((rc = dalloc((void **)&R.sym, sizeof(float2) * C * RDS_SYM_CAP, true))) return rc;
    {   // rdsDecoder_2 / AGC / Costas constructor state (rds-decoder-2.cpp:44-78, rds-decoder.cpp:41-43)
        RdsState s0; std::memset(&s0, 0, sizeof(s0));
        s0.gain = 9.0f; s0.mu = 0.f; s0.skip = 3; s0.sample_count = 0;
        s0.c_limit = (float)(2 * design::kPi * (double)10.0f / (double)(float)24000);
        std::vector<RdsState> init(C, s0);
        if ((rc = dalloc((void **)&R.state, sizeof(RdsState) * C, false))) return rc;
        HIPCHK(hipMemcpy(R.state, init.data(), sizeof(RdsState) * C, hipMemcpyHostToDevice));
    }
    {   // spectra: the DFT of the zero-padded band-pass kernel (fft-filters.cpp:71-82, setBand(57000 -+ 2400)), and the
        // analytic-signal mask of fftFilterHilbert::setHilbert (:177-190)
        const int N = 32768, D = 768;
        std::vector<float> k = design::bandpass(D, 3 * 19000 - 4800 / 2, 3 * 19000 + 4800 / 2, fmRate);
        std::vector<float2> S((size_t)N), M((size_t)N);
        std::vector<double> cs((size_t)N), sn((size_t)N);
        for (int i = 0; i < N; i++) { cs[i] = std::cos(-2 * design::kPi * i / N); sn[i] = std::sin(-2 * design::kPi * i / N); }
        for (int f = 0; f < N; f++) {
            double re = 0, im = 0;
            for (int i = 0; i < D; i++) {
                const int a = (int)(((int64_t)f * i) & (N - 1));
                re += (double)k[2 * i] * cs[a] - (double)k[2 * i + 1] * sn[a];
                im += (double)k[2 * i] * sn[a] + (double)k[2 * i + 1] * cs[a];
            }
            S[f] = make_float2((float)re, (float)im);
            M[f] = make_float2(f == 0 || f == N / 2 ? 1.0f : (f < N / 2 ? 2.0f : 0.0f), 0.f);
        }
        float2 *dS, *dM;
        if ((rc = dalloc((void **)&dS, sizeof(float2) * N, false))) return rc;
        if ((rc = dalloc((void **)&dM, sizeof(float2) * N, false))) return rc;
        HIPCHK(hipMemcpy(dS, S.data(), sizeof(float2) * N, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dM, M.data(), sizeof(float2) * N, hipMemcpyHostToDevice));
        R.S_bp = dS; R.S_hil = dM;
        {   // the band-pass kernel is complex and only the real part of its result is kept (fm-processor.cpp:741): for a real input
            // Re (a * s) = a * Re (s), the DFT of Re (s) is the Hermitian part of S -- the filter vector of the paired transforms
            std::vector<float2> Sr((size_t)N);
            for (int f = 0; f < N; f++) {
                const float2 a = S[(size_t)f], b = S[(size_t)((N - f) & (N - 1))];
                Sr[(size_t)f] = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y - b.y));
            }
            float2 *dSr;
            if ((rc = dalloc((void **)&dSr, sizeof(float2) * N, false))) return rc;
            HIPCHK(hipMemcpy(dSr, Sr.data(), sizeof(float2) * N, hipMemcpyHostToDevice));
            R.S_bp_re = dSr;
        }
        std::vector<float> dk = design::decim_complex(11, 24000 / 2, fmRate);      // rdsDecimator fm-processor.cpp:382
        std::vector<float> rr = design::rrc(1.0, 24000, 2 * 1187.5f, 1.0, 45);       // rds-decoder-2.cpp:67-71
        float2 *dd; float *dr;
        if ((rc = dalloc((void **)&dd, sizeof(float2) * 11, false))) return rc;
        if ((rc = dalloc((void **)&dr, sizeof(float) * 45, false))) return rc;
        HIPCHK(hipMemcpy(dd, dk.data(), sizeof(float2) * 11, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dr, rr.data(), sizeof(float) * 45, hipMemcpyHostToDevice));
        R.dec_taps = dd; R.rrc = dr;
    }
    {   // RDS_1: Costas state, rdsFilter / Match / sharpFilter coefficients (rds-decoder-1.cpp:43-100), rings at 24 kS/s
        if ((rc = dalloc((void **)&R.c_ring, sizeof(float) * C * RDS24_RING, true))) return rc;
        if ((rc = dalloc((void **)&R.f_ring, sizeof(float) * C * RDS24_RING, true))) return rc;
        if ((rc = dalloc((void **)&R.state1, sizeof(Rds1State) * C, true))) return rc;
        std::vector<float> co = design::lowpass(RDS1_FIR, 2 * 2400, 24000);            // rdsFilter (21, RDS_WIDTH, rate)
        const std::vector<float> mk = design::rds1_match_kernel(24000);
        if ((int)mk.size() != RDS1_MATCH) return fail(FMX_E_HIP, "unexpected length of the RDS_1 matched filter");
        co.insert(co.end(), mk.begin(), mk.end());
        const design::Iir bp = design::iir_butterworth_bandpass(7, (int32_t)(1187.5 - 6), (int32_t)(1187.5 + 6), 24000);   // sharpFilter
        if (bp.nq != RDS1_QUADS) return fail(FMX_E_HIP, "unexpected biquad count of the RDS_1 band-pass");
        for (int i = 0; i < RDS1_QUADS; i++) { co.push_back(bp.q[i][1]); co.push_back(bp.q[i][2]); co.push_back(bp.q[i][4]); co.push_back(bp.q[i][5]); }
        co.push_back(bp.gain);
        float *dc;
        if ((rc = dalloc((void **)&dc, sizeof(float) * co.size(), false))) return rc;
        HIPCHK(hipMemcpy(dc, co.data(), sizeof(float) * co.size(), hipMemcpyHostToDevice));
        R.rds1_coef = dc; h->h_rds1 = co;
    }
    {   // RDS_3: state, and mySinCos (rate) of rdsDecoder_3 (rds-decoder-3.cpp:49): SinCos ctor sincos.cpp:45-54 at Rate 24000
        if ((rc = dalloc((void **)&R.state3, sizeof(Rds3State) * C, true))) return rc;
        std::vector<float2> sc24(24000);
        for (int i = 0; i < 24000; i++) sc24[i] = make_float2((float)std::cos(2 * design::kPi * i / 24000), (float)std::sin(2 * design::kPi * i / 24000));
        float2 *d24;
        if ((rc = dalloc((void **)&d24, sizeof(float2) * 24000, false))) return rc;
        HIPCHK(hipMemcpy(d24, sc24.data(), sizeof(float2) * 24000, hipMemcpyHostToDevice));
        R.sincos24 = d24;
    }
    h->rds_read.assign(C, 0);
    {
        int *cl = nullptr;
        if ((rc = dalloc((void **)&h->d_rds_nc0, sizeof(int64_t) * C, true))) return rc;
        if ((rc = dalloc((void **)&cl, sizeof(int) * C, true))) return rc;
        R.nc0 = h->d_rds_nc0; R.chlist = cl;
        h->rds_nc.assign(C, 0); h->rds_nc0.assign(C, -1); h->last_m0.assign(C, 0); h->last_m1.assign(C, 0);
    }
    h->rds_alloc = true;
    return FMX_OK;
}

// ---- the overlap-add block machines (fmx_ola.hip) ----------------------------------------------------------------------------------
int ola_alloc_side(fmx_handle h, fmx_handle_s::OlaSide &S, int L, int degree) {
    const size_t C = (size_t)h->channels;
    S.L = L; S.degree = degree;
    HIPCHK(hipMalloc(&S.A, sizeof(float2) * C * L)); HIPCHK(hipMemset(S.A, 0, sizeof(float2) * C * L));
    HIPCHK(hipMalloc(&S.C, sizeof(float2) * C * L)); HIPCHK(hipMemset(S.C, 0, sizeof(float2) * C * L));
    HIPCHK(hipMalloc(&S.over, sizeof(float2) * C * OLA_MAX_TAPS)); HIPCHK(hipMemset(S.over, 0, sizeof(float2) * C * OLA_MAX_TAPS));
    HIPCHK(hipMalloc(&S.over_new, sizeof(float2) * C * OLA_MAX_TAPS)); HIPCHK(hipMemset(S.over_new, 0, sizeof(float2) * C * OLA_MAX_TAPS));
    HIPCHK(hipMalloc(&S.taps, sizeof(float) * C * OLA_MAX_TAPS)); HIPCHK(hipMemset(S.taps, 0, sizeof(float) * C * OLA_MAX_TAPS));
    S.inp.assign(C, 0); S.on.assign(C, 0); S.key.assign(C, -1);
    for (void *p : {(void *)S.A, (void *)S.C, (void *)S.over, (void *)S.over_new, (void *)S.taps}) h->tail_ptrs.push_back(p);
    return FMX_OK;
}
int ensure_ola(fmx_handle h) {
    if (h->d_v) return FMX_OK;
    const size_t C = (size_t)h->channels;
    HIPCHK(hipMalloc(&h->d_v, sizeof(float2) * C * h->cfg.max_block));
    HIPCHK(hipMalloc(&h->d_u, sizeof(float2) * C * h->cfg.max_block));
    HIPCHK(hipMalloc(&h->d2ring, sizeof(float2) * C * h->dring));
    HIPCHK(hipMemset(h->d2ring, 0, sizeof(float2) * C * h->dring));
    for (void *p : {(void *)h->d_v, (void *)h->d_u, (void *)h->d2ring}) h->tail_ptrs.push_back(p);
    h->pre_look.max_tiles = (int32_t)(h->cfg.max_block / PRE_TILE_SAMPLES + 1);
    HIPCHK(hipMalloc(&h->pre_look.maps, sizeof(float4) * C * h->pre_look.max_tiles));
    HIPCHK(hipMalloc(&h->pre_look.flags, sizeof(int32_t) * C * h->pre_look.max_tiles));
    HIPCHK(hipMemset(h->pre_look.flags, 0, sizeof(int32_t) * C * h->pre_look.max_tiles));
    HIPCHK(hipMalloc(&h->pre_look.tickets, sizeof(uint32_t) * C));
    HIPCHK(hipMemset(h->pre_look.tickets, 0, sizeof(uint32_t) * C));
    h->pre_look.ticket_base = 0;
    h->tail_ptrs.push_back(h->pre_look.maps); h->tail_ptrs.push_back(h->pre_look.flags); h->tail_ptrs.push_back(h->pre_look.tickets);
    int rc = ola_alloc_side(h, h->ola_in, 2 * 32768 - 251, 251);          // inputFilter (2 * 32768, 251) fm-processor.cpp:77
    if (rc) return rc;
    return ola_alloc_side(h, h->ola_au, 2 * 4096 - AUDIO_TAPS, AUDIO_TAPS);   // fmAudioFilter (2 * 4096, 756) :76
}
// setBandwidth / setlfcutoff as the reference's loop takes them over at a block start (fm-processor.cpp:396-408): a new value designs the
// kernel and RESTARTS the block position (fftFilter::setLowPass fft-filters.cpp:84-95: inp = 0, buffers kept); "Off" / <= 0 stops using the
// filter, whose buffers stay as they are
int ola_take_settings(fmx_handle h) {
    bool synced = false;
    for (int side = 0; side < 2; side++) {
        fmx_handle_s::OlaSide &S = side ? h->ola_au : h->ola_in;
        for (int c = 0; c < h->channels; c++) {
            const int32_t want = side ? (h->user[c].lf_cutoff > 0 ? h->user[c].lf_cutoff : 0) : (h->user[c].bandwidth > 0 ? h->user[c].bandwidth : 0);
            bool &ev = side ? h->user[c].lf_event : h->user[c].bw_event;
            const bool again = ev && want != 0 && S.key[(size_t)c] >= 0;        // (the value in use selected again: the block restarts, the kernel stays)
            ev = false;
            if (want == S.key[(size_t)c]) { if (again) { S.on[(size_t)c] = 1; S.inp[(size_t)c] = 0; h->last_filter_event_g = h->g_total; } continue; }
            S.key[(size_t)c] = want;
            h->last_filter_event_g = h->g_total;
            if (want == 0) { S.on[(size_t)c] = 0; continue; }
            const std::vector<float> k = side ? design::lowpass(AUDIO_TAPS, want, h->cfg.fmRate) : design::lowpass(251, want / 2, h->cfg.inputRate);
            if (!synced) { HIPCHK(hipDeviceSynchronize()); synced = true; }      // (an earlier call may still be reading the kernels)
            HIPCHK(hipMemcpy(S.taps + (size_t)c * OLA_MAX_TAPS, k.data(), sizeof(float) * k.size(), hipMemcpyHostToDevice));
            S.on[(size_t)c] = 1; S.inp[(size_t)c] = 0;
        }
    }
    return FMX_OK;
}
// one step of every channel's block machine, as the host works it out; step_ref hands it to the launchers
struct HStep { std::vector<OlaChan> ch; };
int step_ref(fmx_handle h, const HStep &st, hipStream_t s, OlaStepRef *out) {
    *out = OlaStepRef{};
    if (h->channels <= OLA_MAX_CH) { for (int c = 0; c < h->channels; c++) out->val.ch[c] = st.ch[(size_t)c]; return FMX_OK; }
    const size_t C = (size_t)h->channels;
    if (!h->step_dev) {
        HIPCHK(hipMalloc(&h->step_dev, sizeof(OlaChan) * C * fmx_handle_s::STEP_SLOTS));
        HIPCHK(hipHostMalloc((void **)&h->step_host, sizeof(OlaChan) * C * fmx_handle_s::STEP_SLOTS, hipHostMallocDefault));
        for (auto &e : h->step_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->tail_ptrs.push_back(h->step_dev);
    }
    const int k = h->step_next; h->step_next = (k + 1) % fmx_handle_s::STEP_SLOTS;
    if (h->step_used[k]) HIPCHK(hipEventSynchronize(h->step_ev[k]));          // (the copy that last used this slot's host side has run)
    h->step_used[k] = true;
    std::memcpy(h->step_host + (size_t)k * C, st.ch.data(), sizeof(OlaChan) * C);
    HIPCHK(hipMemcpyAsync(h->step_dev + (size_t)k * C, h->step_host + (size_t)k * C, sizeof(OlaChan) * C, hipMemcpyHostToDevice, s));
    HIPCHK(hipEventRecord(h->step_ev[k], s));
    out->tab = h->step_dev + (size_t)k * C;
    return FMX_OK;
}
void ola_fill(const fmx_handle_s::OlaSide &S, OlaBuffers &O) { O.A = S.A; O.C = S.C; O.over = S.over; O.over_new = S.over_new; O.taps = S.taps; O.L = S.L; O.degree = S.degree; }
// does every channel's filter take the whole call as one run (no block boundary before its last sample)?  Then the step is returned and
// the caller's own kernel (pre_kernel / deemph_kernel) does the copy; ola_finish_single runs the block transforms that fall due behind it.
bool ola_single_step(fmx_handle h, const fmx_handle_s::OlaSide &S, int64_t len, HStep *st) {
    st->ch.assign((size_t)h->channels, OlaChan{});
    for (int c = 0; c < h->channels; c++) {
        const bool on = S.on[(size_t)c] != 0;
        if (on && S.inp[(size_t)c] + len > S.L) return false;
        OlaChan &d = st->ch[(size_t)c];
        d.off = 0; d.len = (int32_t)len; d.inp = S.inp[(size_t)c]; d.on = on ? 1 : 0; d.conv = (on && len > 0 && S.inp[(size_t)c] + len == S.L) ? 1 : 0;
    }
    return true;
}
void ola_finish_single(fmx_handle h, fmx_handle_s::OlaSide &S, const HStep &st, const OlaStepRef &ref, const OlaBuffers &O, hipStream_t s) {
    bool any_conv = false;
    for (int c = 0; c < h->channels; c++) any_conv |= st.ch[(size_t)c].conv != 0;
    if (any_conv) { launch_ola_conv(ref, O, h->channels, s); FMX_LAUNCHED(); }
    for (int c = 0; c < h->channels; c++) if (st.ch[(size_t)c].on) S.inp[(size_t)c] = st.ch[(size_t)c].conv ? 0 : S.inp[(size_t)c] + st.ch[(size_t)c].len;
}
// Pass () of every channel over `len` samples: runs up to the block boundary, the block transform where a block completes, and on
int run_ola(fmx_handle h, fmx_handle_s::OlaSide &S, OlaBuffers O, int64_t len, hipStream_t s) {
    ola_fill(S, O);
    std::vector<int64_t> off((size_t)h->channels, 0);
    for (;;) {
        HStep st; st.ch.assign((size_t)h->channels, OlaChan{});
        int maxlen = 0; bool any_conv = false;
        for (int c = 0; c < h->channels; c++) {
            const int64_t rem = len - off[(size_t)c];
            const bool on = S.on[(size_t)c] != 0;
            const int64_t run = on ? std::min<int64_t>(rem, S.L - S.inp[(size_t)c]) : rem;
            OlaChan &d = st.ch[(size_t)c];
            d.off = (int32_t)off[(size_t)c]; d.len = (int32_t)run; d.inp = S.inp[(size_t)c]; d.on = on ? 1 : 0;
            d.conv = (on && run > 0 && S.inp[(size_t)c] + run == S.L) ? 1 : 0;
            maxlen = std::max<int>(maxlen, (int)run); any_conv |= d.conv != 0;
        }
        if (maxlen <= 0) break;
        OlaStepRef ref; { const int rc = step_ref(h, st, s, &ref); if (rc) return rc; }
        launch_ola_io(ref, O, h->channels, maxlen, s); FMX_LAUNCHED();
        if (any_conv) { launch_ola_conv(ref, O, h->channels, s); FMX_LAUNCHED(); }
        for (int c = 0; c < h->channels; c++) {
            const OlaChan &d = st.ch[(size_t)c];
            off[(size_t)c] += d.len;
            if (d.on) S.inp[(size_t)c] = d.conv ? 0 : S.inp[(size_t)c] + d.len;
        }
    }
    return FMX_OK;
}

// the input side of a block-machine handle's call: pre_kernel (RF DC removal, balance, oscillator per sample) and the input filter's machine over the call's
// G.n samples; the machines' output stream is h->d_u
int ola_input_side(fmx_handle h, const CallGeom &G, const DeviceBuffers &B, const void *d_iq, hipStream_t s) {
    OlaBuffers O{}; O.src = h->d_v; O.dst = h->d_u; O.src_stride = O.dst_stride = h->cfg.max_block; O.src_mask = O.dst_mask = -1;
    HStep st1;
    int rc;
    if (ola_single_step(h, h->ola_in, G.n, &st1)) {
        OlaStepRef ref; rc = step_ref(h, st1, s, &ref); if (rc) return rc;
        ola_fill(h->ola_in, O);
        h->pre_look.epoch += 1;
        launch_pre(h->T, B, G, d_iq, h->d_v, h->cfg.max_block, h->channels, s, &ref, &O, h->pre_look); FMX_LAUNCHED();
        { const uint32_t nt = (uint32_t)((G.n + PRE_TILE_SAMPLES - 1) / PRE_TILE_SAMPLES); if (nt > 1) h->pre_look.ticket_base += nt; }
        ola_finish_single(h, h->ola_in, st1, ref, O, s);
    } else {
        h->pre_look.epoch += 1;
        launch_pre(h->T, B, G, d_iq, h->d_v, h->cfg.max_block, h->channels, s, nullptr, nullptr, h->pre_look); FMX_LAUNCHED();
        { const uint32_t nt = (uint32_t)((G.n + PRE_TILE_SAMPLES - 1) / PRE_TILE_SAMPLES); if (nt > 1) h->pre_look.ticket_base += nt; }
        rc = run_ola(h, h->ola_in, O, G.n, s); if (rc) return rc;
    }
    return FMX_OK;
}

// PROMOTION of a folded batch to the block machines at the stream position g_total (fmx_promote.hip has the story).  Rare and heavy: the device is idle around it.
int promote(fmx_handle h, hipStream_t s) {
    std::lock_guard<std::mutex> lk(h->mtx);
    const size_t C = (size_t)h->channels;
    HIPCHK(hipDeviceSynchronize());
    int rc = ensure_ola(h); if (rc) return rc;
    if (h->h_front_sets.size() > h->old_sets_cap) {
        if (h->d_old_sets) (void)hipFree(h->d_old_sets);
        h->d_old_sets = nullptr; h->old_sets_cap = h->h_front_sets.size() + 8;
        HIPCHK(hipMalloc(&h->d_old_sets, sizeof(FrontSet) * h->old_sets_cap));
    }
    if (!h->d_old_set_of) {
        HIPCHK(hipMalloc(&h->d_old_set_of, sizeof(int32_t) * C));
        HIPCHK(hipMalloc(&h->d_params_replay, sizeof(ChanParams) * C));
        HIPCHK(hipMalloc(&h->d_au_tail, sizeof(float2) * C * PROMO_TAIL_AU));
        for (void *p : {(void *)h->d_old_set_of, (void *)h->d_params_replay, (void *)h->d_au_tail}) h->tail_ptrs.push_back(p);
    }
    {   // what the folded stage A used: its tap sets' delays (the ring's newest entries move by them), and the parameters the kept samples were taken under
        std::vector<int32_t> set_of(C);
        std::vector<ChanParams> pr(h->params);
        for (size_t c = 0; c < C; c++) { set_of[c] = h->params[c].front_set; pr[c].actions = 0; }
        HIPCHK(hipMemcpy(h->d_old_sets, h->h_front_sets.data(), sizeof(FrontSet) * h->h_front_sets.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_old_set_of, set_of.data(), sizeof(int32_t) * C, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_params_replay, pr.data(), sizeof(ChanParams) * C, hipMemcpyHostToDevice));
    }
    const int64_t t0 = h->g_total, have = h->promo_have;
    const int64_t J0 = t0 / h->decim;
    // the machines as the settings in force leave them at the first kept sample / at the d ring's tail: kernels designed, block positions as the reference's
    // (both filters count from the stream's first sample: a folded handle has had no restart), buffers empty
    for (int side = 0; side < 2; side++) {
        fmx_handle_s::OlaSide &S = side ? h->ola_au : h->ola_in;
        const int64_t start = side ? J0 - PROMO_TAIL_AU : h->promo_g0;
        HIPCHK(hipMemset(S.A, 0, sizeof(float2) * C * S.L)); HIPCHK(hipMemset(S.C, 0, sizeof(float2) * C * S.L));
        HIPCHK(hipMemset(S.over, 0, sizeof(float2) * C * OLA_MAX_TAPS));
        for (size_t c = 0; c < C; c++) {
            const int32_t want = side ? (h->user[c].lf_applied > 0 ? h->user[c].lf_applied : 0) : (h->user[c].bw_applied > 0 ? h->user[c].bw_applied : 0);
            S.key[c] = want; S.on[c] = want != 0 ? 1 : 0;
            const int64_t org = (side ? h->origin_au : h->origin_in).size() == C ? (side ? h->origin_au : h->origin_in)[c] : 0;
            S.inp[c] = want != 0 ? (int32_t)((((start - org) % S.L) + S.L) % S.L) : 0;
            if (want == 0) continue;
            const std::vector<float> k = side ? design::lowpass(AUDIO_TAPS, want, h->cfg.fmRate) : design::lowpass(251, want / 2, h->cfg.inputRate);
            HIPCHK(hipMemcpy(S.taps + c * OLA_MAX_TAPS, k.data(), sizeof(float) * k.size(), hipMemcpyHostToDevice));
        }
    }
    g_launch_err = hipSuccess;
    // ---- the input side: the channels' state back to where it was at the first kept sample, then pre_kernel and the machine over the kept samples
    launch_promo_state(h->B.state, h->tail_snap, h->channels, 1, s); FMX_LAUNCHED();
    DeviceBuffers Bc = h->B; Bc.params = h->d_params_replay;
    int64_t pos = 0, last_len = 0;
    while (pos < have) {
        int64_t len = (pos == 0 && have % h->cfg.max_block) ? have % h->cfg.max_block : h->cfg.max_block;
        if (len > have - pos) len = have - pos;
        CallGeom Gc{};
        Gc.g0 = h->promo_g0 + pos; Gc.n = len; Gc.input_rate = h->cfg.inputRate; Gc.iq_format = 0; Gc.iq_scale = 1.0f; Gc.stream_stride = h->tail_cap;
        Gc.channels = h->channels; Gc.streams = h->streams; Gc.twins = h->twins; Gc.n_cus = h->n_cus;
        rc = ola_input_side(h, Gc, Bc, h->tail_iq + pos, s); if (rc) return rc;
        pos += len; last_len = len;
    }
    if (last_len < 12 * A_HIST_COLS + 12) return fail(FMX_E_HIP, "promotion: the last kept piece is shorter than the decimators' history");
    // ---- the decimators' history and the ring's newest entries
    launch_promo_hist(h->B.hist, h->d_u, h->cfg.max_block, last_len, (int)(t0 % DECIM), h->twins, h->B.zring, h->ring - 1, J0, h->d_params, h->d_old_sets, h->d_old_set_of,
                      h->channels, s); FMX_LAUNCHED();
    // ---- the audio side: the d ring's tail with the de-emphasis taken out, the audio machine over it, the de-emphasis behind the machine into the ring stage C reads
    launch_promo_inv_deemph(h->B.dring, h->dring - 1, J0, PROMO_TAIL_AU, h->d_params_replay, h->d_au_tail, h->channels, s); FMX_LAUNCHED();
    {
        OlaBuffers O{}; O.src = h->d_au_tail; O.src_stride = PROMO_TAIL_AU; O.src_mask = -1; O.src_pos = 0;
        O.dst = h->d2ring; O.dst_stride = h->dring; O.dst_mask = h->dring - 1; O.dst_pos = J0 - PROMO_TAIL_AU;
        rc = run_ola(h, h->ola_au, O, PROMO_TAIL_AU, s); if (rc) return rc;
        CallGeom Gx{}; Gx.J0 = J0 - PROMO_TAIL_AU; Gx.J1 = J0; Gx.dring_mask = h->dring - 1;
        launch_deemph(Bc, Gx, h->d2ring, h->channels, s, nullptr, nullptr); FMX_LAUNCHED();
    }
    HIPCHK(g_launch_err);
    HIPCHK(hipDeviceSynchronize());
    // ---- a block-machine handle from here on: the pending setters reach their filters through ola_take_settings, as a small handle's do
    h->ola_mode = true; h->sets_dirty = true; h->params_dirty = true;
    h->promo_pending = false; h->promo_have = 0; h->promoted = true; h->demo_capture = false; h->last_filter_event_g = t0;
    for (auto &u : h->user) { u.bw_applied = u.bandwidth; u.lf_applied = u.lf_cutoff; }
    return FMX_OK;
}

// may a promoted handle go back?  Every filter that has ever run is running and has been undisturbed for three blocks (a filter that is switched off keeps
// buffers the reference would replay when it is switched on again: such a handle stays on the machines), nothing is pending
bool demotable(fmx_handle h) {
    if (!h->promoted || !h->ola_mode || h->folded_pinned || h->twins != 1) return false;
    if (h->g_total - h->last_filter_event_g < DEMO_QUIET) return false;
    for (int side = 0; side < 2; side++) {
        const fmx_handle_s::OlaSide &S = side ? h->ola_au : h->ola_in;
        for (int c = 0; c < h->channels; c++) {
            const ChanUser &u = h->user[(size_t)c];
            if ((side ? u.lf_event : u.bw_event)) return false;
            const int32_t want = side ? (u.lf_cutoff > 0 ? u.lf_cutoff : 0) : (u.bandwidth > 0 ? u.bandwidth : 0);
            if (want != S.key[(size_t)c] && !(want == 0 && S.key[(size_t)c] <= 0)) return false;      // (a setter on its way)
            if (S.key[(size_t)c] > 0 && !S.on[(size_t)c]) return false;
            if (S.key[(size_t)c] == 0 && want == 0 && S.inp[(size_t)c] != 0) return false;             // (switched off in mid-block: stale buffers)
        }
    }
    return true;
}
// DEMOTION: the folded filters again at the stream position g_total (see DEMO_QUIET in fmx_internal.h)
int demote(fmx_handle h, hipStream_t s) {
    std::lock_guard<std::mutex> lk(h->mtx);
    const size_t C = (size_t)h->channels;
    HIPCHK(hipDeviceSynchronize());
    const int64_t t1 = h->g_total, have = h->promo_have, J1 = t1 / h->decim;
    h->origin_in.assign(C, 0); h->origin_au.assign(C, 0);
    for (size_t c = 0; c < C; c++) {
        if (h->ola_in.on[c]) h->origin_in[c] = t1 - h->ola_in.inp[c];
        if (h->ola_au.on[c]) h->origin_au[c] = J1 - h->ola_au.inp[c];
    }
    h->ola_mode = false; h->promoted = false; h->demo_capture = false; h->promo_have = 0;
    for (auto &u : h->user) { u.bw_applied = u.bandwidth; u.lf_applied = u.lf_cutoff; }
    h->sets_dirty = true;
    int rc = ensure_sets(h); if (rc) return rc;                   // (the folded tap sets of the settings in force; the channels' set indices with them)
    {
        std::vector<ChanParams> pr(h->params);
        for (auto &p : pr) p.actions = 0;
        HIPCHK(hipMemcpy(h->d_params_replay, pr.data(), sizeof(ChanParams) * C, hipMemcpyHostToDevice));
    }
    g_launch_err = hipSuccess;
    // the channels' stage-A state back to the first kept sample, an empty filter history (the run's first 24 outputs land in ring entries nobody reads any more)
    launch_promo_state(h->B.state, h->tail_snap, h->channels, 1, s); FMX_LAUNCHED();
    HIPCHK(hipMemsetAsync(h->B.hist, 0, sizeof(float2) * C * DECIM * A_HIST_COLS, s));
    HIPCHK(hipMemsetAsync(h->B.dcv_hist, 0, sizeof(float2) * C * DCV_SAVE, s));
    DeviceBuffers Bc = h->B; Bc.params = h->d_params_replay;
    int64_t pos = 0;
    while (pos < have) {
        const int64_t len = std::min<int64_t>(h->cfg.max_block, have - pos);
        CallGeom Gc{};
        Gc.g0 = h->promo_g0 + pos; Gc.n = len; Gc.J0 = Gc.g0 / h->decim; Gc.J1 = (Gc.g0 + len) / h->decim;
        Gc.ring_mask = h->ring - 1; Gc.dring_mask = h->dring - 1; Gc.sring_mask = h->sring - 1; Gc.input_rate = h->cfg.inputRate; Gc.pitch = h->pitch;
        Gc.iq_format = 0; Gc.iq_scale = 1.0f; Gc.stream_stride = h->tail_cap; Gc.channels = h->channels; Gc.streams = h->streams; Gc.twins = h->twins; Gc.n_cus = h->n_cus;
        Gc.parts = 1;
        launch_front(h->T, Bc, Gc, h->tail_iq + pos, h->channels, s); FMX_LAUNCHED();
        pos += len;
    }
    // the d ring as the folded stage B leaves it: de-emphasised (its last entries, from a standing start 2048 entries in front of what stage C reads)
    {
        CallGeom Gx{}; Gx.J0 = J1 - DEMO_TAIL_AU; Gx.J1 = J1; Gx.dring_mask = h->dring - 1;
        launch_deemph(Bc, Gx, h->B.dring, h->channels, s, nullptr, nullptr); FMX_LAUNCHED();
    }
    HIPCHK(g_launch_err);
    HIPCHK(hipDeviceSynchronize());
    h->params_dirty = true;
    return FMX_OK;
}

static bool any_rds_on(fmx_handle h) { return h->call_any_rds; }     // (as of the call's flush_mailbox)

constexpr int PIPE_ROWS_AUTO = 3072;      // fm samples per piece of an overlapping call where pllC runs (two of stage B's segments; measured at 4096 channels:
                                          // 2048 / 3072 / 4608 / 6400 fm samples per piece give 6.37 / 6.18 / 6.46 / 6.79 ms per step, the call made whole 8.24)
constexpr int PIPE_ROWS_AUTO_PLL = 4608;  // ... where pllC runs for the PLL decoder only (round 6: its chain is 66 issue slots per sample instead of 80 and the launches' own cost counts more:
                                          // pieces of 3072 x 6 / 4608 x 4 / 5376 x 3 + 3072 / 4608 x 3 + 3072 + 2304 / 4608 x 3 + 3840 + 1536 give 5.29 / 5.21 / 4.92 / 4.82 / 4.83 ms per step)
constexpr int PIPE_ROWS_AUTO_AM = 3840;   // ... where the AM decoder runs (its chain is longer per sample): 3072 x 6 / 3840 x 4 + 2304 + 1536 / 3840 x 4 + 1536 + 2304 / 3072 x 5 + 2304 + 1536 give 6.0-6.1 / 5.72 / 5.93 / 5.77 ms per step
constexpr int PIPE_ROWS_AUTO_SQ = 4608;   // ... where only squelches do (noise squelch 6.61 / 5.73 / 5.46 / 5.54 against 5.90 whole, level squelch 5.43 / 4.73 / 4.59 / 4.72 against 5.23)
constexpr int TAIL_MIN_CHANNELS = 128;    // the second stage-B / C channel group: at least this many channels
constexpr int PIPE_MIN_CHANNELS = 1024;   // automatic: batches that fill the chip
constexpr int PLL_SEQ_AUTO_MAX = 64;   // FMX_P_PLL_SOLVER = 0: handles up to this many channels evaluate the pilot PLL sequentially

// the tiled work arrays of the demodulator pre-pass (fmx_demod.hip): limited / unlimited samples in, demodulator output out.  (Allocated by the first call
// that needs them -- before run_call decides how the call is made: a call in pieces needs them too.)
int ensure_prepass_arrays(fmx_handle h) {
    if (h->B.w_iq) return FMX_OK;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMalloc(&h->B.w_iq, sizeof(float2) * (size_t)h->work_nj * h->pitch));
    HIPCHK(hipMemset(h->B.w_iq, 0, sizeof(float2) * (size_t)h->work_nj * h->pitch));
    HIPCHK(hipMalloc(&h->B.w_osc, sizeof(float) * (size_t)h->work_nj * h->pitch));
    HIPCHK(hipMemset(h->B.w_osc, 0, sizeof(float) * (size_t)h->work_nj * h->pitch));
    return FMX_OK;
}
int flush_mailbox(fmx_handle h) {
    std::lock_guard<std::mutex> lk(h->mtx);
    if (h->gain_dirty) { h->gain_pending = true; h->gain_dirty = false; }   // (a change arriving behind this point belongs to the next call, flag and value)
    if (h->ola_mode) { int rc = ensure_ola(h); if (rc) return rc; rc = ola_take_settings(h); if (rc) return rc; }
    else if (!h->promo_pending) for (auto &u : h->user) { u.bw_event = false; u.lf_event = false; }     // (a folded handle applies a filter value, not the setter's call)
    if (h->sets_dirty) { int rc = ensure_sets(h); if (rc) return rc; h->params_dirty = true; }
    bool any_lo = false;
    for (auto &p : h->params) any_lo |= (p.lo_freq != 0);
    if (any_lo) { int rc = ensure_lo_table(h); if (rc) return rc; }
    {
        // stage A on the matrix pipe (fmx_front4.hip): every tap set the long fold with its RfDC taken 12 columns back; a handle with a local oscillator somewhere
        // runs the complex-tap variant (fmx_front4lo.hip)
        bool ok = h->twins == 1 && !h->ola_mode;
        for (auto &p : h->params) { if (!ok) break; ok = h->h_front_sets[(size_t)p.front_set].nd > 4; }
        // front4_kernel applies the IQ balance in front of its filter together with the tile's scale and takes the RF DC recurrence's column sums back
        // through 1 / balance: a balance of 0 (the slider's end, radio.cpp:989-995) or one no slider produces goes to front_kernel's per-sample pass
        for (auto &p : h->params) {
            if (!ok) break;
            const float al = std::fabs(p.att_l), ar = std::fabs(p.att_r);
            ok = h->h_front_sets[(size_t)p.front_set].dc_k == 12 && al >= 1e-6f && al <= 1e6f && ar >= 1e-6f && ar <= 1e6f;
        }
        h->front4_ok = ok; h->front4_lo = any_lo;
        if (ok && any_lo) {
            // the sum of a channel's complex taps, Hlo = sum_m G [m] e^(j 2 pi ((m lo) mod R) / R): what the filter makes of the RF DC value (fmx_front4.hip)
            // (computed when a channel's tap set or oscillator changes, not per call: 300 sine / cosine pairs per channel)
            const int R = h->cfg.inputRate;
            h->hlo_key.resize(h->params.size(), std::make_pair(-1, 0x7fffffff));
            for (size_t ci = 0; ci < h->params.size(); ci++) {
                ChanParams &p = h->params[ci];
                const std::pair<int32_t, int32_t> key(h->front_keys[(size_t)p.front_set / (size_t)h->twins], p.lo_freq);
                if (h->hlo_key[ci] == key) continue;
                h->hlo_key[ci] = key;
                const FrontSet &fs = h->h_front_sets[(size_t)p.front_set];
                const float *tz = &h->h_front_taps[(size_t)p.front_set * A_TAPS_STRIDE];
                double hr = 0, hi = 0;
                for (int d = 0; d < A_MAX_ND; d++)
                    for (int r = 0; r < DECIM; r++) {
                        const int m = 12 * d + fs.off - r;
                        if (m < 0) continue;
                        const int64_t ph = (((int64_t)m * p.lo_freq) % R + R) % R;
                        const double g = (double)tz[(d + 1) * DECIM + r], a = 2.0 * design::kPi * (double)ph / (double)R;
                        hr += g * std::cos(a); hi += g * std::sin(a);
                    }
                const float fr = (float)hr, fi = (float)hi;
                if (fr != p.hlo_re || fi != p.hlo_im) { p.hlo_re = fr; p.hlo_im = fi; h->params_dirty = true; }
            }
        }
    }
    bool any_rds = false;
    h->call_rds_mode.resize((size_t)h->channels);
    for (int c = 0; c < h->channels; c++) { h->call_rds_mode[(size_t)c] = (int8_t)h->params[(size_t)c].rds_mode; any_rds |= (h->params[(size_t)c].rds_mode != 0); }
    h->call_any_rds = any_rds;
    // (a channel's RDS path runs while its decoder is on and stands still otherwise -- block filters, phase delay line, decimator and slicer
    // keep what they hold, as the reference's processor's do: nothing restarts when a decoder is switched; run_call_one counts per channel)
    if (any_rds) { const int rc = ensure_rds(h); if (rc) return rc; }
    bool any_pll = false;
    for (auto &p : h->params) any_pll |= (p.decoder == 2 || p.decoder == 1 || p.squelch_mode != 0);     // pllC on the fm-rate IQ; |z| for the level squelch; the general AFC body for the noise squelch
    {   // the scope taps that are rows of stage B's work arrays (FMX_P_SCOPE_TAPS): display feeds, kept where there is a display; the RDS path reads two of them
        const int want = h->scope_taps.load();
        const bool keep = want < 0 ? h->channels <= 64 : want != 0;
        if (keep && !h->w_diff_mem) {
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipMalloc(&h->w_diff_mem, sizeof(float) * (size_t)h->work_nj * h->pitch));
            HIPCHK(hipMemset(h->w_diff_mem, 0, sizeof(float) * (size_t)h->work_nj * h->pitch));
            h->tail_ptrs.push_back(h->w_diff_mem);
        }
        h->B.w_diff = keep ? h->w_diff_mem : nullptr;
        h->taps_kept = keep;
        h->B.rows_on = (keep || any_rds) ? 1 : 0;
        h->B.peaks_on = keep ? 1 : 0;
    }
    {
        int var = 0;
        for (auto &p : h->params)          // (of the channels the pre-pass touches: fmx_demod.hip, afc_kernel's variants)
            var |= (p.decoder == 2 ? 1 : 0) | (p.decoder == 1 ? 2 : 0) | (p.squelch_mode == 2 ? 4 : 0) | ((p.decoder > 2 && p.squelch_mode != 0) ? 8 : 0);
        h->B.prepass_var = var;
    }
    bool any_nsq = false;
    for (auto &p : h->params) any_nsq |= (p.squelch_mode == 1);
    if (any_nsq && !h->d_nsq) {
        // squelch ctor squelchClass.cpp:11-18 with mySquelch (1, 70000, fmRate / 20, fmRate) fm-processor.cpp:87
        const design::Iir hp = design::iir_chebyshev_lowhigh(true, 20, 70000 - 100, h->cfg.fmRate);
        const design::Iir lp = design::iir_chebyshev_lowhigh(false, 20, 70000, h->cfg.fmRate);
        h->h_nsq.assign(2 * NSQ_QUADS * 4 + 2, 0.f);
        for (int f = 0; f < 2; f++) {
            const design::Iir &F = f ? lp : hp;
            if (F.nq != NSQ_QUADS) return fail(FMX_E_HIP, "unexpected biquad count of the squelch filters");
            for (int i = 0; i < NSQ_QUADS; i++) {
                h->h_nsq[(f * NSQ_QUADS + i) * 4 + 0] = F.q[i][1]; h->h_nsq[(f * NSQ_QUADS + i) * 4 + 1] = F.q[i][2];
                h->h_nsq[(f * NSQ_QUADS + i) * 4 + 2] = F.q[i][4]; h->h_nsq[(f * NSQ_QUADS + i) * 4 + 3] = F.q[i][5];
            }
            h->h_nsq[2 * NSQ_QUADS * 4 + f] = F.gain;
        }
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMalloc(&h->d_nsq, sizeof(float) * h->h_nsq.size()));
        HIPCHK(hipMemcpy(h->d_nsq, h->h_nsq.data(), sizeof(float) * h->h_nsq.size(), hipMemcpyHostToDevice));
        h->T.nsq_coef = h->d_nsq; h->tail_ptrs.push_back(h->d_nsq);
    }
    if (any_pll) { const int rc = ensure_prepass_arrays(h); if (rc) return rc; }
    for (int c = 0; c < h->channels; c++) {            // set_squelchValue takes effect at a block start, when it differs (fm-processor.cpp:410-413)
        ChanUser &u = h->user[c];
        if (u.squelch_value != u.squelch_old) {
            u.squelch_level = u.squelch_value; u.squelch_old = u.squelch_value;
            h->params[c].squelch_thr = std::pow(10.0f, (float)(u.squelch_level - 80) / 30.0f);       // squelchClass.cpp:33-37
            h->params[c].squelch_nthr = 1.0f - (float)u.squelch_level / 100.0f;
            h->params_dirty = true;
        }
    }
    if (h->params_dirty) {
        HIPCHK(hipDeviceSynchronize());   // the previous call may still run on the caller's stream and the side streams
        HIPCHK(hipMemcpy(h->d_params, h->params.data(), sizeof(ChanParams) * h->channels, hipMemcpyHostToDevice));
        // one-shot actions stay pending on the host until a call's kernels have consumed them (actions_consumed): an
        // upload alone -- a call that fails afterwards, or one too short to hold an fm sample -- must not lose them
        h->act_up.resize((size_t)h->channels);
        for (int c = 0; c < h->channels; c++) h->act_up[(size_t)c] = h->params[(size_t)c].actions;
        h->params_dirty = false;
    }
    return FMX_OK;
}

// The kernels of a call have been enqueued: the action bits they consume are done.  ACT_DC_RESET belongs to the input-FIR
// kernel (every call), the other two to the PSS integrator, which only looks at them in a call with at least one fm sample.
void actions_consumed(fmx_handle h, bool had_fm_samples) {
    std::lock_guard<std::mutex> lk(h->mtx);
    if (h->act_up.size() != (size_t)h->channels) return;
    const int32_t done = ACT_DC_RESET | (had_fm_samples ? (ACT_TRIGGER_FREQ | ACT_RESTART_PSS) : 0);
    for (int c = 0; c < h->channels; c++) {
        const int32_t clr = h->act_up[(size_t)c] & done;
        if (clr) { h->params[(size_t)c].actions &= ~clr; h->act_up[(size_t)c] &= ~clr; h->params_dirty = true; }   // upload the cleared bits before the next call
    }
}

void frames_geom(const fmx_handle h, int64_t n, CallGeom *G) {
    G->g0 = h->g_total; G->n = n;
    G->J0 = h->g_total / h->decim; G->J1 = (h->g_total + n) / h->decim;
    // newConverter: 192 frames in -> 48 out (newconverter.cpp:55-80, inputLimit = fmRate/1000)
    G->M0 = 48 * (G->J0 / 192); G->M1 = 48 * (G->J1 / 192);
}
// frames the second converter has put out after `in` frames at the working rate: outputs m with m q < in p
int64_t conv2_out(const fmx_handle h, int64_t in) { return h->cv_nt ? (in * h->cv_p + h->cv_q - 1) / h->cv_q : in; }

// Stage A runs one workgroup per channel, two per CU: a handle with fewer channels than that leaves compute units idle while each workgroup
// walks its channel's tiles one after the other.  Such a call splits every channel in time (fmx_front.hip, CallGeom::parts): as many parts as
// fill the chip, none shorter than FRONT_MIN_PART_TILES tiles (every later part computes one tile twice).  The results are the same bit for bit.
constexpr int FRONT_MIN_PART_TILES = 6, FRONT_MAX_PARTS = 32, FRONT_TILE = 128 * DECIM;
int front_parts_for(fmx_handle h, CallGeom &G) {
    G.parts = 1; G.part_tiles = 0; G.streams = h->streams;
    const int want = h->front_parts.load();
    if (h->twins != 1 || want == 1) return FMX_OK;
    const int64_t r0 = G.g0 % DECIM;
    const int NT = (int)((r0 + G.n - 1) / FRONT_TILE) + 1;
    int parts = want > 1 ? want : (2 * h->n_cus) / h->channels;
    if (want <= 1 && parts > NT / FRONT_MIN_PART_TILES) parts = NT / FRONT_MIN_PART_TILES;
    if (parts > FRONT_MAX_PARTS) parts = FRONT_MAX_PARTS;
    if (parts > NT / 2) parts = NT / 2;              // (forced: at least two tiles per part)
    if (parts < 2) return FMX_OK;
    const int pt = (NT + parts - 1) / parts;
    parts = (NT + pt - 1) / pt;
    if (parts < 2) return FMX_OK;
    if (!h->B.fsnap) {
        const size_t C = (size_t)h->channels;
        h->B.dc_pitch = (int32_t)(h->cfg.max_block / FRONT_TILE + 2);
        HIPCHK(hipMalloc(&h->B.dc_tiles, sizeof(float4) * 2 * (size_t)h->streams * h->B.dc_pitch));
        HIPCHK(hipMalloc(&h->B.fsnap, sizeof(FrontSnap) * C));
        HIPCHK(hipMalloc(&h->B.hist_snap, sizeof(float2) * C * DECIM * A_HIST_COLS));
        HIPCHK(hipMalloc(&h->B.dcv_snap, sizeof(float2) * C * DCV_SAVE));
        for (void *p : {(void *)h->B.dc_tiles, (void *)h->B.fsnap, (void *)h->B.hist_snap, (void *)h->B.dcv_snap}) h->tail_ptrs.push_back(p);
    }
    G.parts = parts; G.part_tiles = pt;
    return FMX_OK;
}

// the handle's side streams (a call made in overlapping pieces: run_call; stage B's last round of workgroups beside stage C: run_call_one), created when first wanted
int ensure_pipe_streams(fmx_handle h) {
    if (h->pipe_sA) return FMX_OK;
    // (plain streams, the recurrences' with priority.  Measured and not kept: compute units of their own for the recurrences -- CU-masked streams,
    // 16 / 32 / 64 CUs for the 64 waves, the rest for the other streams: 6.7 / 6.3 / 6.0 ms per step against 6.2 without masks.  A lone wave walks a
    // sample in 216 ns with the chip to itself and in 270-300 ns while the other stages run, on CUs of its own or not: what it loses is the
    // chip's clock under load, not its SIMD.)
    int lo = 0, hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPCHK(hipStreamCreateWithPriority(&h->pipe_sA, hipStreamNonBlocking, lo));
    HIPCHK(hipStreamCreateWithPriority(&h->pipe_sB, hipStreamNonBlocking, lo));
    HIPCHK(hipStreamCreateWithPriority(&h->pipe_sP, hipStreamNonBlocking, hi));     // (the chain everything waits for)
    for (hipEvent_t *e : {&h->pipe_ev0, &h->pipe_evA, &h->pipe_evD, &h->pipe_evP, &h->pipe_evE, &h->pipe_evB[0], &h->pipe_evB[1]}) HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    return FMX_OK;
}

struct PipePiece { int k; };       // a piece of a call made in overlapping pieces (its number)
int run_call_one(fmx_handle h, const void *d_iq, int32_t fmt, float s16_den, int64_t stream_stride, int64_t n, float2 *d_pcm,
                 int64_t pcm_stride, int64_t *n_frames, hipStream_t s, const PipePiece *pp = nullptr);
// One call of the boundary.  The RDS front end works on blocks of RDS_BLK fm samples, every channel on its own block phase, and one launch sequence covers
// at most one block boundary per channel: while a channel decodes RDS, a longer call is made in pieces (the chain is invariant to how a stream is cut into calls).
int run_call_pieces(fmx_handle h, const void *d_iq, int32_t fmt, float s16_den, int64_t stream_stride, int64_t n, float2 *d_pcm,
                    int64_t pcm_stride, int64_t *n_frames, hipStream_t s);
int run_call(fmx_handle h, const void *d_iq, int32_t fmt, float s16_den, int64_t stream_stride, int64_t n, float2 *d_pcm,
             int64_t pcm_stride, int64_t *n_frames, hipStream_t s) {
    h->call_head = true;
    const int rc = run_call_pieces(h, d_iq, fmt, s16_den, stream_stride, n, d_pcm, pcm_stride, n_frames, s);
    h->call_head = true;
    return rc;
}
int run_call_pieces(fmx_handle h, const void *d_iq, int32_t fmt, float s16_den, int64_t stream_stride, int64_t n, float2 *d_pcm,
                    int64_t pcm_stride, int64_t *n_frames, hipStream_t s) {
    const int64_t PIECE = (int64_t)(RDS_BLK - 1) * h->decim;       // (J1 - J0 <= RDS_BLK whatever the call's phase in the fm-rate grid; decim: input samples per
                                                                   // fm sample at this handle's rate -- 12, 6 or 1 as the reference decimates)
    bool any_rds = false;
    { std::lock_guard<std::mutex> lk(h->mtx); for (auto &p : h->params) any_rds |= (p.rds_mode != 0); }
    h->last_pieces = 1;
    if (!any_rds && fmt >= 0 && fmt <= 3 && n <= h->cfg.max_block) {
        // A batch whose channels need the demodulator pre-pass (PLL / AM decoder, a squelch: fmx_demod.hip).  Its recurrences are one wave per 64 channels
        // walking the call's fm samples one after the other -- 64 waves on 1024 SIMDs at 4096 channels, for longer than stages A, B and C together take --
        // and nothing else can run meanwhile: stage B waits for its output, it waits for stage A.  Such a call is made in PIECES (the chain is invariant
        // to how a stream is cut into calls) on three streams: stage A of piece k + 1 and stage B / C of piece k - 1 fill the chip while the pre-pass
        // walks piece k.  The pre-pass's work arrays alternate between two halves of their allocation; an event per stage boundary orders the rest.
        // What the stages share beyond their hand-over arrays: the myCount of the metaData snapshot (kept by stage B, read by the pre-pass: handed over by
        // the host, CallGeom::host_count) and the RF DC level stage B's snapshot reads from stage A's state (a display value that moves by 1e-7 of its
        // distance per sample: it may be the next piece's).
        const int want = h->pipe_rows.load() >= 0 ? h->pipe_rows.load() : env_switches().call_pieces;
        bool special = false, chain = false, am_chain = false;
        { std::lock_guard<std::mutex> lk(h->mtx); for (auto &p : h->params) { special |= (p.decoder == 2 || p.decoder == 1 || p.squelch_mode != 0); chain |= (p.decoder == 2 || p.decoder == 1); am_chain |= (p.decoder == 1); } }
        if (special && !h->ola_mode) { const int rc = ensure_prepass_arrays(h); if (rc) return rc; }
        // (pllC's chain: longer pieces with a short last one, PIPE_ROWS_AUTO_PLL / _AM)
        const bool taper = want < 0 && chain;
        // (a call too short for two of the PLL decoder's longer pieces is cut into the shorter ones)
        const int64_t rows = want > 0 ? ((want + 15) / 16) * 16
                                      : (chain ? (n < 2 * (int64_t)(am_chain ? PIPE_ROWS_AUTO_AM : PIPE_ROWS_AUTO_PLL) * h->decim ? PIPE_ROWS_AUTO : (am_chain ? PIPE_ROWS_AUTO_AM : PIPE_ROWS_AUTO_PLL))
                                               : PIPE_ROWS_AUTO_SQ);
        const int64_t ends_rows = env_switches().call_pieces_ends >= 0 ? ((env_switches().call_pieces_ends + 15) / 16) * 16 : 0;
        const int64_t half = (h->work_nj / 2) & ~(int64_t)15;
        const int64_t piece = rows * h->decim;
        // (only these batches.  Measured: the headline's batch -- no pre-pass; stage A bound by HBM, stage B by instruction issue -- made in 13 / 6 / 4 / 3 overlapping
        // pieces takes 4.97 / 3.89 / 3.60 / 3.60 ms per step against 3.46 whole: stage B's workgroups fill the register files, the stages do not share a CU)
        if (special && h->B.w_iq && want != 0 && !h->ola_mode && !h->cv_nt && (want > 0 || h->channels >= PIPE_MIN_CHANNELS) && rows + 2 <= half && n >= 2 * piece) {
            {
                CallGeom G{}; frames_geom(h, n, &G);
                if (conv2_out(h, G.M1) - conv2_out(h, G.M0) > pcm_stride) return fail(FMX_E_TOO_LARGE, "pcm_stride smaller than the frames this call produces");
            }
            { const int rc = ensure_pipe_streams(h); if (rc) return rc; }
            // the pieces run on the handle's three streams, behind what the caller's stream held when the call began; the caller's stream goes on behind the last of them
            HIPCHK(hipEventRecord(h->pipe_ev0, s));
            HIPCHK(hipStreamWaitEvent(h->pipe_sA, h->pipe_ev0, 0)); HIPCHK(hipStreamWaitEvent(h->pipe_sB, h->pipe_ev0, 0));
            const int64_t bps = (fmt == 0) ? 8 : (fmt == 3 ? 4 : 2);
            // the pieces' lengths.  What the call pays beyond the chain of the lone waves is its FIRST piece's stage A (nothing to overlap it with yet) and its
            // LAST piece's stages B and C (the chain has ended): short end pieces of `ends` fm samples, equal ones in between.
            std::vector<int64_t> lens;
            const int64_t ends = ends_rows * h->decim;
            if (ends > 0 && n >= 2 * ends + piece) {
                const int64_t inner = n - 2 * ends;
                int64_t m = (inner + piece / 2) / piece; if (m < 1) m = 1;
                const int64_t grid = 16 * h->decim;
                int64_t mid = ((inner / m) / grid) * grid;
                if (mid / h->decim + 2 > half) { m++; mid = ((inner / m) / grid) * grid; }
                lens.push_back(ends);
                for (int64_t i = 0; i < m; i++) lens.push_back(mid);
                lens.push_back(n - ends - m * mid);
            }
            if (!env_switches().call_pieces_list.empty()) {      // (a diagnostic: the pieces' fm samples spelt out; the last one takes what is left)
                lens.clear(); int64_t used = 0;
                for (int r : env_switches().call_pieces_list) { const int64_t l = (int64_t)r * h->decim; if (used + l < n) { lens.push_back(l); used += l; } }
                lens.push_back(n - used);
            }
            bool fits = !lens.empty();
            for (int64_t l : lens) fits = fits && l > 0 && l / h->decim + 2 <= half;
            if (!fits && taper) {
                // whole pieces, then the rest (between one and two pieces) as a multiple of half a segment of stage B's and a SHORT last piece of one segment to one
                // and a half: what follows the chain's end is the last piece's stages B and C (19200 fm samples: 4608 4608 4608 3840 1536; AM decoder: 3840 x 4, 2304, 1536)
                lens.clear();
                const int64_t seg = (int64_t)1536 * h->decim;           // (stage B's segment, fmx_stageb.hip FB_W)
                int64_t pos = 0;
                while (n - pos >= 2 * piece) { lens.push_back(piece); pos += piece; }
                const int64_t R = n - pos;
                int64_t a = ((R - seg) / (seg / 2)) * (seg / 2);
                if (a > piece) a = piece;
                if (a >= seg) { lens.push_back(a); lens.push_back(R - a); } else lens.push_back(R);
                fits = true;
                for (int64_t l : lens) fits = fits && l > 0 && l / h->decim + 2 <= half;
            }
            if (!fits) {
                lens.clear();
                for (int64_t pos = 0; pos < n;) {
                    // (a last piece shorter than half a piece rides with the one before it: the arrays' halves hold a piece and a half)
                    int64_t len = (n - pos < piece) ? n - pos : piece;
                    if (n - pos - len > 0 && n - pos - len < piece / 2 && (rows * 3) / 2 + 2 <= half) len = n - pos;
                    lens.push_back(len); pos += len;
                }
            }
            int64_t total = 0, pos = 0; int k = 0;
            for (int64_t len : lens) {
                int64_t got = 0;
                const PipePiece pp{k};
                const int rc = run_call_one(h, reinterpret_cast<const char *>(d_iq) + pos * bps, fmt, s16_den, stream_stride, len, d_pcm + total, pcm_stride, &got, h->pipe_sB, &pp);
                if (rc) { (void)hipDeviceSynchronize(); return rc; }
                total += got; pos += len; k++;
            }
            HIPCHK(hipEventRecord(h->pipe_evE, h->pipe_sB)); HIPCHK(hipStreamWaitEvent(s, h->pipe_evE, 0));
            h->last_pieces = k;
            if (n_frames) *n_frames = total;
            return FMX_OK;
        }
    }
    if (!any_rds || n <= PIECE || fmt < 0 || fmt > 3 || n > h->cfg.max_block) return run_call_one(h, d_iq, fmt, s16_den, stream_stride, n, d_pcm, pcm_stride, n_frames, s);
    {
        CallGeom G{}; frames_geom(h, n, &G);
        if (conv2_out(h, G.M1) - conv2_out(h, G.M0) > pcm_stride) return fail(FMX_E_TOO_LARGE, "pcm_stride smaller than the frames this call produces");
    }
    const int64_t bps = (fmt == 0) ? 8 : (fmt == 3 ? 4 : 2);
    int64_t total = 0;
    for (int64_t pos = 0; pos < n; pos += PIECE) {
        int64_t got = 0;
        const int rc = run_call_one(h, reinterpret_cast<const char *>(d_iq) + pos * bps, fmt, s16_den, stream_stride, (n - pos < PIECE) ? n - pos : PIECE,
                                    d_pcm + total, pcm_stride, &got, s);
        if (rc) return rc;
        total += got;
    }
    if (n_frames) *n_frames = total;
    return FMX_OK;
}

int run_call_one(fmx_handle h, const void *d_iq, int32_t fmt, float s16_den, int64_t stream_stride, int64_t n, float2 *d_pcm,
                 int64_t pcm_stride, int64_t *n_frames, hipStream_t s, const PipePiece *pp) {
    if (fmt < 0 || fmt > 3) return fail(FMX_E_INVALID, "unknown IQ format");
    if (fmt == 3) {
        int ex = 0; const float m = std::frexp(s16_den, &ex);
        if (!(s16_den >= 1.0f) || m != 0.5f) return fail(FMX_E_INVALID, "s16_denominator must be a power of two >= 1");
    }
    if (n <= 0 || n > h->cfg.max_block) return fail(FMX_E_TOO_LARGE, "n_complex must be in [1, max_block]");
    int rc;
    bool capture = false;
    {   // a folded batch with a filter change pending (fmx_promote.hip): promoted once enough of its streams is kept, else this call's samples are kept too
        bool do_promote = false, do_demote = false;
        {
            std::lock_guard<std::mutex> lk(h->mtx);
            if (h->ola_mode) {
                // (a promoted handle whose machines have been quiet goes back to the folded filters: it keeps a block of its streams first)
                const bool ok = demotable(h);
                if (!ok || h->promo_recapture) { h->demo_capture = false; h->promo_have = 0; }
                else if (!h->demo_capture) { h->demo_capture = true; h->promo_have = 0; capture = true; }
                else if (h->promo_have >= DEMO_TAIL_IN && h->call_head) do_demote = true;
                else capture = true;
                h->promo_recapture = false;
            }
            else if (!h->promo_pending) h->promo_recapture = false;
            else if (h->promo_recapture) { h->promo_recapture = false; h->promo_have = 0; }      // (what pre_kernel applies changed: the kept samples start over, behind this call)
            else if (h->promo_have >= PROMO_TAIL_IN && h->call_head) do_promote = true;
            else capture = true;
        }
        if (do_promote) { rc = promote(h, s); if (rc) return rc; }
        if (do_demote) { rc = demote(h, s); if (rc) return rc; }
    }
    rc = flush_mailbox(h);
    if (rc) return rc;
    CallGeom G{};
    frames_geom(h, n, &G);
    G.ring_mask = h->ring - 1; G.dring_mask = h->dring - 1; G.sring_mask = h->sring - 1;
    G.input_rate = h->cfg.inputRate; G.pitch = h->pitch; G.streams_private = (h->streams_private && h->twins == 1) ? 1 : 0; G.twins = h->twins; G.channels = h->channels; G.stream_stride = stream_stride; G.pcm_stride = pcm_stride;
    G.iq_format = fmt; G.iq_scale = (fmt == 3) ? 1.0f / s16_den : 1.0f / 128.0f; G.n_cus = h->n_cus;
    const int64_t frames = conv2_out(h, G.M1) - conv2_out(h, G.M0);
    if (frames > pcm_stride) return fail(FMX_E_TOO_LARGE, "pcm_stride smaller than the frames this call produces");
    if (h->rds_alloc && any_rds_on(h) && G.J1 - G.J0 > RDS_BLK)
        return fail(FMX_E_TOO_LARGE, "with RDS on, a call may cover at most 32000 fm samples (384000 input samples)");
    // (a piece of an overlapping call: stage A on its own stream, behind what the caller's stream held when the call began)
    // (FMX_CALL_PIECES_SERIAL=1, a diagnostic: the same pieces one after the other on the caller's stream -- what the overlapping run must equal bit for bit)
    const bool piped = pp != nullptr && !h->ola_mode && !env_switches().pieces_serial;
    hipStream_t sa = piped ? h->pipe_sA : s;
    G.host_count1 = piped ? h->my_count_host + 1 : 0;
    ProfRec pr{}; const bool prof = h->prof_on;
    if (prof) {
        for (int i = 0; i < 4; i++) HIPCHK(hipEventCreate(&pr.e[i]));
        pr.in_samples = n * h->streams; pr.ch_samples = n * h->channels;
        HIPCHK(hipEventRecord(pr.e[0], sa));
    }
    g_launch_err = hipSuccess;
    h->B.lin_rows = (int32_t)h->work_nj;
    if (capture && h->tail_iq && h->promo_have + n > h->tail_cap) capture = false;      // (cannot happen: a change is applied at the first call boundary behind PROMO_TAIL_IN samples)
    if (capture) {
        if (!h->tail_iq) {
            h->tail_cap = PROMO_TAIL_IN + h->cfg.max_block;
            HIPCHK(hipMalloc(&h->tail_iq, sizeof(float2) * (size_t)h->streams * h->tail_cap));
            HIPCHK(hipMalloc(&h->tail_snap, sizeof(FrontSnap) * (size_t)h->channels));
            h->tail_ptrs.push_back(h->tail_iq); h->tail_ptrs.push_back(h->tail_snap);
        }
        if (h->promo_have == 0) { h->promo_g0 = h->g_total; launch_promo_state(h->B.state, h->tail_snap, h->channels, 0, sa); FMX_LAUNCHED(); }
        launch_capture(d_iq, fmt, G.iq_scale, stream_stride, n, h->streams, h->tail_iq, h->tail_cap, h->promo_have, sa); FMX_LAUNCHED();
        h->promo_have += n;
    }
    rc = front_parts_for(h, G);
    if (rc) return rc;
    {
        const int fk = h->front_kernel.load() ? h->front_kernel.load() : env_switches().front_kernel;     // (the environment: A/B runs of one build)
        // automatic: the filter on the matrix pipe wherever a handle qualifies and has the channels to fill the chip without splitting them in time
        // (measured at 4096 channels on one box: 1.52 ms per launch against 1.75 for the four-wave kernel and 1.84 for the six-wave VALU kernel, which
        // both sit at the packed-FMA power limit)
        // (the complex-tap variant runs one channel per workgroup: a handle of one channel per compute unit fills the chip)
        if (h->front4_ok && !h->front4_lo && (fk == 3 || (fk == 0 && G.parts <= 1))) { G.parts = 1; G.front4 = 1; }
        if (h->front4_ok && h->front4_lo && (fk == 3 || (fk == 0 && h->channels >= h->n_cus))) { G.parts = 1; G.front4 = 2; }
        // (what is reported is what runs: a call without a whole tile on the kernel's grid goes to front_kernel in launch_front)
        h->last_front_kernel = (G.front4 && front4_tiles(G, d_iq) > 0) ? 3 : 1;
    }
    if (h->ola_mode) {
        // RF DC removal / balance / LO mix per sample, the input filter as the reference's block machine, then the decimators
        rc = ola_input_side(h, G, h->B, d_iq, s); if (rc) return rc;
        CallGeom Gp = G; Gp.pre_processed = 1; Gp.iq_format = 0; Gp.iq_scale = 1.0f; Gp.stream_stride = h->cfg.max_block; Gp.streams_private = h->twins == 1 ? 1 : 0;
        launch_front(h->T, h->B, Gp, h->d_u, h->channels, s);
    } else
    launch_front(h->T, h->B, G, d_iq, h->channels, sa);   // stage A: four waves per channel, packed-FMA FIR (fmx_front.hip)
    FMX_LAUNCHED();
    const bool prof_double = env_switches().prof_double != 0;    // (diagnostic: a throw-away event in front of each boundary event)
    if (prof && prof_double && !h->ev_dummy) HIPCHK(hipEventCreate(&h->ev_dummy));      // (one per handle, destroyed with it)
    hipEvent_t pdummy = h->ev_dummy;
    if (prof && prof_double) HIPCHK(hipEventRecord(pdummy, sa));
    if (prof) HIPCHK(hipEventRecord(pr.e[1], sa));
    G.stageb_form = h->stageb_form.load(); G.no_deemph = h->ola_mode ? 1 : 0;
    // A plain batch (no pre-pass, no RDS, no scope-tap rows: stage B as the whole kernel, three workgroups per CU) of more channels than one round of stage-B
    // workgroups: stages B and C run as TWO CHANNEL GROUPS on two streams.  Stage B is bound by instruction issue and leaves its last round of workgroups partly
    // filled (4096 channels are 5.33 rounds of 768); stage C moves the d ring and the PCM.  With the second group's stage B finished while the first group's still
    // runs, its stage C fills what stage B leaves free: 3.48 -> 3.32 ms per step at 4096 channels.  Stage A stays one launch with the chip to itself -- the stage the
    // roofline line is measured on.  The groups touch disjoint channels (CallGeom::ch0 / ch_count); the caller's stream goes on behind both.  FMX_TAIL_SPLIT=0: off.
    int tail_ch = 0;
    {
        const bool tail_on = env_switches().tail_split != 0;
        const int slots = 3 * (h->n_cus > 0 ? h->n_cus : 256);
        const bool plain = !piped && !h->ola_mode && !h->B.w_iq && !h->cv_nt && !h->B.rows_on && !(h->rds_alloc && any_rds_on(h)) && !h->gain_pending && G.stageb_form == 0 &&
                           env_switches().stageb_split < 0 && G.J1 > G.J0 && G.M1 > G.M0;
        if (tail_on && plain && h->channels > slots) {
            // (the first group: 2304 of 4096 channels on 256 CUs -- three whole rounds.  Measured there, second group of 256 / 512 / 1024 /
            // 1792 / 2048 / 3072 channels: 3.47 / 3.45 / 3.35 / 3.32 / 3.39 / 3.43 ms per step against 3.47-3.55 with one group.  FMX_TAIL_CH=n: a diagnostic)
            // other counts, one group -> two, ms per step: 3840 (2304 + 1536) 3.20 -> 3.12, 3000 (2304 + 696) 2.51 -> 2.43 (1536 + 1464: 2.51), 2048 (1536 + 512)
            // 1.68 -> 1.63, 1536 (768 + 768) 1.30 -> 1.27, 1024 (768 + 256) 0.915 -> 0.910: the first group is two thirds of the rounds, in whole rounds
            int first_rounds = (int)(0.65 * (double)h->channels / (double)slots + 0.5);
            if (first_rounds < 1) first_rounds = 1;
            tail_ch = h->channels - first_rounds * slots;
            const int force = env_switches().tail_ch;
            if (force > 0 && force < h->channels) tail_ch = force;
            if (tail_ch < TAIL_MIN_CHANNELS) tail_ch = 0;             // (a second group of a few dozen channels is three launches for nothing)
            if (tail_ch > 0) { const int rc2 = ensure_pipe_streams(h); if (rc2) return rc2; }
        }
        h->last_second_group = tail_ch;
    }
    if (piped) { HIPCHK(hipEventRecord(h->pipe_evA, sa)); HIPCHK(hipStreamWaitEvent(s, h->pipe_evA, 0)); }      // (stage B of this piece behind its stage A)
    if (piped) {
        // the pre-pass of this piece: its parallel part (disc_kernel) behind the piece's stage A on that stream -- and behind the stage B that read the half
        // of the work arrays it is about to overwrite --, the recurrences on theirs, the noise squelch's pipeline and stage B on `s`
        if (pp->k >= 2) HIPCHK(hipStreamWaitEvent(sa, h->pipe_evB[pp->k & 1], 0));
        DeviceBuffers Bk = h->B;
        if (pp->k & 1) { const size_t off = (size_t)((h->work_nj / 2) & ~(int64_t)15) * (size_t)h->pitch; Bk.w_iq += off; Bk.w_osc += off; }
        const PrepassStreams ps{sa, h->pipe_sP, h->pipe_evD, h->pipe_evP};
        launch_demod_fused(h->T, Bk, G, h->channels, s, &ps);
        HIPCHK(hipEventRecord(h->pipe_evB[pp->k & 1], s));
    } else if (tail_ch > 0) {
        // (the second group on the handle's side stream, behind stage A)
        HIPCHK(hipEventRecord(h->pipe_evA, s)); HIPCHK(hipStreamWaitEvent(h->pipe_sA, h->pipe_evA, 0));
        CallGeom G1 = G, G2 = G;
        G1.ch0 = 0; G1.ch_count = h->channels - tail_ch; G2.ch0 = h->channels - tail_ch; G2.ch_count = tail_ch;
        launch_demod_fused(h->T, h->B, G1, h->channels, s);
        launch_demod_fused(h->T, h->B, G2, h->channels, h->pipe_sA);
    } else
    launch_demod_fused(h->T, h->B, G, h->channels, s);      // (with its pre-pass for the PLL / AM decoders and the squelches)
    if (G.J1 > G.J0) {       // (stage B's bookkeeping behind the call, fmx_stageb.hip)
        int cnt = h->my_count_host + (int)(G.J1 - G.J0);
        if (cnt > (SINCOS_N >> 1)) cnt -= (SINCOS_N >> 1) + 1;
        h->my_count_host = cnt;
    }
    if (h->rds_alloc && any_rds_on(h)) {
        const int64_t nj = G.J1 - G.J0;
        int modes = 0;
        for (int c = 0; c < h->channels; c++) {
            const int m = h->call_rds_mode[(size_t)c];
            modes |= 1 << m;
            h->rds_nc0[(size_t)c] = m != 0 ? h->rds_nc[(size_t)c] : -1;
        }
        // (pageable source: the copy is staged before the call returns, the vector is free again)
        HIPCHK(hipMemcpyAsync(h->d_rds_nc0, h->rds_nc0.data(), sizeof(int64_t) * (size_t)h->channels, hipMemcpyHostToDevice, s));
        launch_rds(h->B, h->R, G, h->channels, h->rds_nc0.data(), modes, s);
        for (int c = 0; c < h->channels; c++) {
            const int64_t a = h->rds_nc0[(size_t)c];
            if (a < 0) { h->last_m0[(size_t)c] = h->last_m1[(size_t)c]; continue; }       // (decoder off in this call: no outputs, not the last on-call's)
            h->last_m0[(size_t)c] = a / 8; h->last_m1[(size_t)c] = (a + nj) / 8;
            h->rds_nc[(size_t)c] = a + nj;
        }
    }
    if (prof && prof_double) HIPCHK(hipEventRecord(pdummy, s));
    if (prof) HIPCHK(hipEventRecord(pr.e[2], s));
    DeviceBuffers Bq = h->B;
    if (h->ola_mode && G.J1 > G.J0) {
        // the audio low-pass as the reference's block machine on the de-emphasised stream; the resampler reads its output
        OlaBuffers O{}; O.src = h->B.dring; O.dst = h->d2ring; O.src_stride = O.dst_stride = h->dring; O.src_mask = O.dst_mask = h->dring - 1;
        O.src_pos = O.dst_pos = G.J0;
        HStep st1;
        // de-emphasis behind the filter, as the reference orders them (:589-595)
        if (ola_single_step(h, h->ola_au, G.J1 - G.J0, &st1)) {
            OlaStepRef ref; rc = step_ref(h, st1, s, &ref); if (rc) return rc;
            ola_fill(h->ola_au, O);
            launch_deemph(h->B, G, h->d2ring, h->channels, s, &ref, &O); FMX_LAUNCHED();
            ola_finish_single(h, h->ola_au, st1, ref, O, s);
        } else {
            rc = run_ola(h, h->ola_au, O, G.J1 - G.J0, s); if (rc) return rc;
            launch_deemph(h->B, G, h->d2ring, h->channels, s, nullptr, nullptr); FMX_LAUNCHED();
        }
    }
    if (h->ola_mode) Bq.dring = h->d2ring;
    if (h->gain_pending && G.M1 > G.M0) { G.gain_fix = 1; launch_gain_fix(h->T, Bq, G, h->channels, s); FMX_LAUNCHED(); h->gain_pending = false; }
    if (tail_ch > 0) {
        CallGeom G1 = G, G2 = G;
        G1.ch0 = 0; G2.ch0 = h->channels - tail_ch;
        launch_audio(h->T, Bq, G1, d_pcm, h->channels - tail_ch, s);
        launch_audio(h->T, Bq, G2, d_pcm, tail_ch, h->pipe_sA);
        HIPCHK(hipEventRecord(h->pipe_evE, h->pipe_sA)); HIPCHK(hipStreamWaitEvent(s, h->pipe_evE, 0));
    } else if (!h->cv_nt) launch_audio(h->T, Bq, G, d_pcm, h->channels, s);
    else {
        // the audio stage writes its 48 kHz frames behind the converter's history; theConverter's output goes to the caller
        CallGeom G48 = G; G48.pcm_stride = h->x48_stride;
        launch_audio(h->T, Bq, G48, h->d_x48 + h->cv_nt, h->channels, s);
        launch_conv2(h->d_x48, h->x48_stride, h->d_cv_taps, h->cv_p, h->cv_q, h->cv_nt, G.M0, G.M1 - G.M0, conv2_out(h, G.M0), frames,
                     d_pcm, pcm_stride, h->channels, s);
    }
    if (prof) { HIPCHK(hipEventRecord(pr.e[3], s)); h->prof.push_back(pr); }
    FMX_LAUNCHED();
    HIPCHK(g_launch_err);
    actions_consumed(h, G.J1 > G.J0);
    h->last_J0 = G.J0; h->last_J1 = G.J1;
    h->g_total += n;
    h->call_head = false;
    if (n_frames) *n_frames = frames;
    return FMX_OK;
}

int prof_drain(fmx_handle h) {
    for (auto &pr : h->prof) {
        HIPCHK(hipEventSynchronize(pr.e[3]));
        for (int k = 0; k < 3; k++) {
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, pr.e[k], pr.e[k + 1]));
            h->prof_acc.ms[k] += ms; h->prof_acc.launches[k] += 1;
        }
        h->prof_acc.input_samples += pr.in_samples; h->prof_acc.channel_samples += pr.ch_samples;
        for (int i = 0; i < 4; i++) (void)hipEventDestroy(pr.e[i]);
    }
    h->prof.clear();
    return FMX_OK;
}

}  // namespace

namespace fmx {
const EnvSwitches &env_switches() {
    static const EnvSwitches sw = [] {
        EnvSwitches e{};
        e.call_pieces = env_int("FMX_CALL_PIECES", -1); e.call_pieces_ends = env_int("FMX_CALL_PIECES_ENDS", -1);
        if (const char *v = getenv("FMX_CALL_PIECES_LIST")) { for (const char *q = v; *q;) { e.call_pieces_list.push_back(atoi(q)); while (*q && *q != ',') q++; if (*q) q++; } } e.pieces_serial = env_int("FMX_CALL_PIECES_SERIAL", 0) != 0; e.front_kernel = env_int("FMX_FRONT_KERNEL", 0);
        e.prof_double = getenv("FMX_PROF_DOUBLE") != nullptr; e.tail_split = env_int("FMX_TAIL_SPLIT", 1) != 0; e.tail_ch = env_int("FMX_TAIL_CH", 0);
        e.stageb_split = env_int("FMX_STAGEB_SPLIT", -1); e.rows_off_split = getenv("FMX_ROWS_OFF_SPLIT") != nullptr; e.no_sinpoly = getenv("FMX_DEBUG_NO_SINPOLY") != nullptr;
        e.host_zerocopy = env_int("FMX_HOST_ZEROCOPY", 1) != 0; e.rds_pair = env_int("FMX_RDS_PAIR", 1) != 0;
        return e;
    }();
    return sw;
}
}  // namespace fmx

// ---- diagnostics: the practical HBM ceiling (SURVEY 8d asks for the measured device-copy bandwidth next to the nominal 8 TB/s)
namespace fmx {
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// mode 0: dst[i] = src[i] (float2 copy, 16 B per lane per access);  mode 1: stage A's traffic shape: read 12 float2, write 1;  mode 2: the same reads and NO
// write (a sum that is never the sentinel): what this GPU reads at when nothing is written -- the ceiling of `roofline.frac_read_only`.
template <int MODE>
__global__ __launch_bounds__(256) void stream_probe_kernel(const f32x4_t *__restrict__ src, f32x4_t *__restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    if (MODE == 0) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
    } else {
        // each lane reads 6 x 16 B (12 float2) spaced a wave apart (coalesced), sums, and writes one float2 per 12 read
        const size_t ngroups = n16 / (6 * 64);
        const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = stride >> 6;
        const int lane = threadIdx.x & 63;
        for (size_t g = wave; g < ngroups; g += nwaves) {
            f32x4_t a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 6; k++) a += __builtin_nontemporal_load(src + (g * 6 + k) * 64 + lane);
            if (MODE == 2) { if (a.x + a.z == 1.2345e33f) reinterpret_cast<float2 *>(dst)[lane] = make_float2(a.y, a.w); }
            else reinterpret_cast<float2 *>(dst)[g * 64 + lane] = make_float2(a.x + a.z, a.y + a.w);
        }
    }
}
}  // namespace fmx

extern "C" {

int fmx_abi_version(void) { return FMX_ABI_VERSION; }
const char *fmx_last_error(void) { return g_err.c_str(); }

int fmx_create(const fmx_config *cfg, fmx_handle *out) {
    if (!cfg || !out) return fail(FMX_E_INVALID, "null argument");
    (void)env_switches();                       // (the FMX_* environment switches: read here, once per process)
    if (cfg->struct_size != (int32_t)sizeof(fmx_config)) return fail(FMX_E_INVALID, "fmx_config.struct_size mismatch");
    if (cfg->channels < 1) return fail(FMX_E_INVALID, "channels must be >= 1");
    if (cfg->fmRate != 192000 || cfg->workingRate != 48000)
        return fail(FMX_E_UNSUPPORTED, "this build implements fmRate 192000 / workingRate 48000 (radio.cpp:68,231-233)");
    int decim = DECIM;
    {   // The reference derives its two decimators from inputRate (fm-processor.cpp:36,68-75): fmBand_1 always divides by 6, fmBand_2 by
        // (inputRate / 6) / fmRate in INTEGER arithmetic, and with inputRate / fmRate <= 1 there are no decimators at all (:471) -- whatever
        // rate that leaves is then treated as fmRate.  Stage A's kernel decimates by 12; a total of 6 or 1 runs it as 2 or 12 "twins" per
        // channel, one per output phase (CallGeom::twins).  Filters, LO table and DC constant are designed for the rate given.
        if (cfg->inputRate < 1 || cfg->inputRate > 100000000) return fail(FMX_E_INVALID, "inputRate out of range");
        if (cfg->inputRate / cfg->fmRate <= 1) decim = 1;
        else {
            const int32_t IRate = cfg->inputRate / 6;
            const int64_t d2 = IRate / cfg->fmRate;
            if ((d2 != 1 && d2 != 2) || 4 * (int64_t)cfg->inputRate / IRate + 1 != 25 || cfg->inputRate / IRate != 6)
                return fail(FMX_E_UNSUPPORTED, "inputRate: built are the rates the reference decimates by 12 (2304000 <= inputRate < 3456000), by 6 "
                                               "(1152000 <= inputRate < 2304000) and not at all (inputRate < 384000); below 1152000 the reference's "
                                               "second decimator has one tap and a zero gain (fir-filters.cpp:327-347), from 3456000 on it divides by 3+");
            decim = 6 * (int)d2;
        }
    }
    std::vector<float> cv_taps; int cv_p = 1, cv_q = 1, cv_nt = 0;
    if (cfg->audioRate != cfg->workingRate) {
        if (cfg->audioRate < 8000 || cfg->audioRate > 192000 ||
            !design::design_conv2(cfg->workingRate, cfg->audioRate, &cv_p, &cv_q, &cv_nt, &cv_taps))
            return fail(FMX_E_UNSUPPORTED, "audioRate: 8000 .. 192000 with audioRate / gcd (audioRate, workingRate) <= 640 (the second converter's phases)");
    }
    if (cfg->max_block < 12 || cfg->max_block > (1 << 20)) return fail(FMX_E_INVALID, "max_block must be in [12, 1048576]");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(FMX_E_NO_DEVICE, "no HIP device visible: libfmx has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(FMX_E_INVALID, "device ordinal out of range");
    HIPCHK(hipSetDevice(cfg->device));

    fmx_handle h = new (std::nothrow) fmx_handle_s();
    if (!h) return fail(FMX_E_NOMEM, "out of host memory");
    // every failure below goes through fmx_destroy: nothing of a half-built handle stays behind
    auto init = [&]() -> int {
    h->cfg = *cfg; h->cfg.stream_of_channel = nullptr;
    h->decim = decim; h->twins = DECIM / decim;
    h->channels = cfg->channels;
    h->streams = cfg->streams > 0 ? cfg->streams : cfg->channels;
    if (cv_nt) {
        h->cv_p = cv_p; h->cv_q = cv_q; h->cv_nt = cv_nt;
        h->x48_stride = cv_nt + cfg->max_block / (4 * h->decim) + 96;
        HIPCHK(hipMalloc(&h->d_cv_taps, sizeof(float) * cv_taps.size()));
        HIPCHK(hipMemcpy(h->d_cv_taps, cv_taps.data(), sizeof(float) * cv_taps.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMalloc(&h->d_x48, sizeof(float2) * (size_t)h->channels * h->x48_stride));
        HIPCHK(hipMemset(h->d_x48, 0, sizeof(float2) * (size_t)h->channels * h->x48_stride));
    }
    h->ola_mode = h->channels <= OLA_MAX_CH;          // FMX_P_FILTER_RESTARTS = 0 (automatic)
    h->user.assign(h->channels, ChanUser());
    h->params.assign(h->channels, ChanParams());
    for (int c = 0; c < h->channels; c++) {
        ChanParams &p = h->params[c];
        std::memset(&p, 0, sizeof(p));
        int s = cfg->stream_of_channel ? cfg->stream_of_channel[c] : (cfg->streams > 0 ? c % h->streams : c);
        if (s < 0 || s >= h->streams) return fail(FMX_E_INVALID, "stream_of_channel entry out of range");
        p.stream = s;
        // constructor defaults fm-processor.cpp:110-160 / fm-demodulator.cpp:66
        p.fm_mode = 0; p.sound_sel = 0; p.decoder = 3; p.auto_mono = 1; p.pss_active = 1; p.dc_remove = 1;
        p.rds_mode = 0; p.lo_freq = 0; p.lo_period = 0; p.att_l = 1.f; p.att_r = 1.f;
        p.pll_seq = h->channels <= PLL_SEQ_AUTO_MAX ? 1 : 0;
        p.squelch_mode = 0; p.squelch_thr = std::pow(10.0f, (float)(1 - 80) / 30.0f); p.squelch_nthr = 1.0f - 1 / 100.0f;
        refresh_derived(h, c);
    }
    {
        std::vector<int> users((size_t)h->streams, 0);
        h->streams_private = true;
        for (auto &p : h->params) if (++users[(size_t)p.stream] > 1) h->streams_private = false;
    }
    {
        hipDeviceProp_t dp;
        HIPCHK(hipGetDeviceProperties(&dp, cfg->device));
        h->n_cus = dp.multiProcessorCount;
        int optin = 0;
        if (hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, cfg->device) != hipSuccess) optin = 0;
        h->lds_per_block = std::max((size_t)dp.sharedMemPerBlock, (size_t)optin);
    }
    HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));

    // ---- tables -------------------------------------------------------------------------
    const int32_t fmRate = cfg->fmRate;
    {
        std::vector<float2> sc((size_t)SINCOS_N);                       // SinCos ctor sincos.cpp:45-54
        for (int i = 0; i < SINCOS_N; i++)
            sc[i] = make_float2((float)std::cos(2 * design::kPi * i / fmRate), (float)std::sin(2 * design::kPi * i / fmRate));
        HIPCHK(hipMalloc(&h->d_sincos, sizeof(float2) * SINCOS_N));
        HIPCHK(hipMemcpy(h->d_sincos, sc.data(), sizeof(float2) * SINCOS_N, hipMemcpyHostToDevice));
        {   // the table as two polynomials (SinPoly): accept it only if it reproduces every entry, bar a few listed exceptions
            SinPoly sp{}; sp.step = 2 * design::kPi / fmRate;
            bool fits = (fmRate == SINCOS_N);
            for (int i = 0; fits && i < SINCOS_N; i++) {
                float sn, cs;
                sincos_poly(i, sp.step, &sn, &cs);
                if (std::memcmp(&sn, &sc[i].y, 4) != 0) { if (sp.ns < 4) { sp.s_idx[sp.ns] = i; sp.s_val[sp.ns] = sc[i].y; sp.ns++; } else fits = false; }
                if (std::memcmp(&cs, &sc[i].x, 4) != 0) { if (sp.nc < 4) { sp.c_idx[sp.nc] = i; sp.c_val[sp.nc] = sc[i].x; sp.nc++; } else fits = false; }
            }
            sp.ok = (fits && !env_switches().no_sinpoly) ? 1 : 0;
            h->T.sp = sp;
        }
        {   // 2-level factorisation of the sine column for the sequential pilot PLL (LDS resident):
            // sin(2 pi idx/N) = Im(EA[a] EB[b]), idx = 256 a + b.  Used only if the f64 expression rounds to
            // exactly the reference's f32 table entry for EVERY idx (checked here with the same unfused
            // multiply/add sequence the kernel runs); otherwise the kernel reads the global table.
            std::vector<double2> tr((size_t)TRIG2_N);
            for (int a = 0; a < TRIG2_A; a++) { double t = 2 * design::kPi * (256.0 * a) / fmRate; tr[a] = make_double2(std::cos(t), std::sin(t)); }
            for (int a = 0; a < TRIG2_APAD; a++) tr[TRIG2_A + a] = tr[a];          // idx in [N, N + 512) wraps to idx - N
            for (int b = 0; b < TRIG2_B; b++) { double t = 2 * design::kPi * (double)b / fmRate; tr[TRIG2_A + TRIG2_APAD + b] = make_double2(std::cos(t), std::sin(t)); }
            bool exact = (fmRate == TRIG2_A * TRIG2_B);
            for (int i = 0; exact && i < SINCOS_N; i++) {
                const double2 ea = tr[i >> 8], eb = tr[TRIG2_A + TRIG2_APAD + (i & 255)];
                const float sn = (float)(ea.y * eb.x + ea.x * eb.y);
                if (std::memcmp(&sn, &sc[i].y, 4) != 0) exact = false;
            }
            if (exact) {
                HIPCHK(hipMalloc(&h->d_trig3, sizeof(double2) * TRIG2_N));
                HIPCHK(hipMemcpy(h->d_trig3, tr.data(), sizeof(double2) * TRIG2_N, hipMemcpyHostToDevice));
            }
        }
        {   // the pilot-phase wrap: (float)((double)v - 2 pi) == (v - P32) + C32 in f32 for every float v the PLL can
            // produce there, v in [P32, P32 + 0.7)?  (P32 = the float just above 2 pi.)  Checked bit for bit.
            const float P32 = 6.2831855f;
            const float C32 = (float)((double)P32 - 2 * design::kPi);
            bool ok = true;
            for (float v = P32; ok && v < P32 + 0.7f; v = std::nextafterf(v, 100.f)) {
                const float want = (float)((double)v - 2 * design::kPi);
                volatile float d = v - P32;
                const float got = d + C32;
                if (std::memcmp(&want, &got, 4) != 0) ok = false;
            }
            h->T.wrap32_c = C32; h->T.wrap32_ok = ok ? 1 : 0;
        }
        std::vector<float> at((size_t)ATAN_N + 1);                      // compAtan ctor Xtan2.cpp:28-31
        const float St = (float)design::kPi;
        for (int i = 0; i <= ATAN_N; i++) { float f = (float)i / ATAN_N; at[i] = (float)((double)(std::atan(f) * St) / design::kPi); }
        HIPCHK(hipMalloc(&h->d_atan, sizeof(float) * (ATAN_N + 1)));
        HIPCHK(hipMemcpy(h->d_atan, at.data(), sizeof(float) * (ATAN_N + 1), hipMemcpyHostToDevice));
        std::vector<float> as((size_t)ARCSINE_N + 1);                   // fm-demodulator.cpp:74-77
        for (int i = 0; i <= ARCSINE_N; i++) as[i] = (float)(std::asin(2.0 * i / ARCSINE_N - 1.0) / 2.0);
        HIPCHK(hipMalloc(&h->d_arcsine, sizeof(float) * (ARCSINE_N + 1)));
        HIPCHK(hipMemcpy(h->d_arcsine, as.data(), sizeof(float) * (ARCSINE_N + 1), hipMemcpyHostToDevice));
        h->h_pss_taps = design::lowpass(PSS_TAPS, 15000, fmRate);      // stereo-separation.cpp:31,39
        HIPCHK(hipMalloc(&h->d_pss_taps, sizeof(float) * PSS_TAPS));
        HIPCHK(hipMemcpy(h->d_pss_taps, h->h_pss_taps.data(), sizeof(float) * PSS_TAPS, hipMemcpyHostToDevice));
        h->h_rs_taps = design::resampler(RS_TAPS);
    }
    h->T.sincos = h->d_sincos; h->T.atan_ppy = h->d_atan; h->T.arcsine = h->d_arcsine; h->T.lo_table = nullptr; h->T.trig2 = h->d_trig3;
    h->T.pss_taps = h->d_pss_taps;
    {   // fast-convolution form of the PSS low-pass (fmx_fftconv.h): twiddles and the taps' spectrum in the device transform's own order
        std::vector<float2> W(fftc::W_COUNT), Hs(fftc::N);
        fftc::make_twiddles(W.data());
        fftc::make_spectrum(h->h_pss_taps.data(), PSS_TAPS, Hs.data(), W.data());
        HIPCHK(hipMalloc(&h->d_fft_w, sizeof(float2) * fftc::W_COUNT));
        HIPCHK(hipMalloc(&h->d_pss_hs, sizeof(float2) * fftc::N));
        HIPCHK(hipMemcpy(h->d_fft_w, W.data(), sizeof(float2) * fftc::W_COUNT, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->d_pss_hs, Hs.data(), sizeof(float2) * fftc::N, hipMemcpyHostToDevice));
        h->T.fft_w = h->d_fft_w;
        h->T.pss_hs = h->d_pss_hs;
    }
    h->T.sincos_C = fmRate / (2 * design::kPi);
    {   // fm_Demodulator ctor fm-demodulator.cpp:57-72
        const float F_G = (float)(0.65 * fmRate / 2), Delta_F = (float)(0.95 * fmRate / 2);
        const float B_FM = 2 * (Delta_F + F_G);
        h->T.K_FM = (float)((double)(2 * B_FM) * design::kPi / (double)F_G);
        const float max_dev = (float)(0.95 * (0.5 * fmRate));
        const float fac = (float)(2.0 * design::kPi / fmRate);         // pllC ctor pllC.cpp:42-53
        h->T.pll_beta = (float)std::exp(-2.0 * design::kPi * (double)(float)(0.85 * fmRate) / 2 / fmRate);
        h->T.pll_lo = -max_dev * fac; h->T.pll_hi = max_dev * fac; h->T.pll_center = (float)(0 * 2 * design::kPi / fmRate);
    }
    // pilotRecovery / PSS ctor args fm-processor.cpp:78-82, stereo-separation.cpp:32
    h->T.pil_omega = (float)((double)((float)19000 / (float)fmRate) * (2 * design::kPi));
    h->T.pil_gain = (float)(10 * (2 * design::kPi) / fmRate);
    h->T.K_FM_rcp = 1.0f / h->T.K_FM; h->T.pil_omega_rcp = 1.0f / h->T.pil_omega;
    h->T.pss_alpha = 10.0f / (float)fmRate;
    h->T.pss_lock_alpha = 1.0f / fmRate;
    h->T.afc_l2 = (float)std::log2((double)(1 - 0.0001f));                              // fm-demodulator.cpp:197 (1 - fmDcAlpha)
    h->T.lock_l2 = (float)std::log2(1.0 - (double)(1.0f / 3000.0f));                    // pilot-recover.cpp:66 (1.0 - alpha)
    h->T.pssmean_l2 = (float)std::log2((double)(1.0f - h->T.pss_lock_alpha));          // stereo-separation.cpp:90 (1.0f - lockAlpha)

    // ---- per-channel buffers ------------------------------------------------------------
    const int64_t fm_per_call = cfg->max_block / h->decim + 2;
    h->ring = next_pow2((2 * 32768 - 251) / h->decim + 1 + fm_per_call + 512);      // the input filter's latency in fm samples + a call
    // (a batch keeps three blocks of the audio filter in its d ring: what its promotion to the block machines runs the audio machine over, fmx_promote.hip)
    h->dring = next_pow2(std::max<int64_t>(AUDIO_DELAY + C_MAX_TAPS, h->channels > OLA_MAX_CH ? PROMO_TAIL_AU + 8 : 0) + fm_per_call + 192 + 4 * C_TILE);
    h->sring = 4096;
    const size_t C = (size_t)h->channels;
    const size_t CT = C * (size_t)h->twins;                    // stage-A workgroups: `twins` per channel
    HIPCHK(hipMalloc(&h->B.hist, sizeof(float2) * CT * DECIM * A_HIST_COLS));
    HIPCHK(hipMalloc(&h->B.dcv_hist, sizeof(float2) * CT * DCV_SAVE));
    HIPCHK(hipMemset(h->B.dcv_hist, 0, sizeof(float2) * CT * DCV_SAVE));
    if (h->twins > 1) {
        HIPCHK(hipMalloc(&h->B.state_tw, sizeof(ChanState) * C * (size_t)(h->twins - 1)));
        HIPCHK(hipMemset(h->B.state_tw, 0, sizeof(ChanState) * C * (size_t)(h->twins - 1)));
        h->tail_ptrs.push_back(h->B.state_tw);
    }
    HIPCHK(hipMalloc(&h->B.zring, sizeof(float2) * C * h->ring));
    HIPCHK(hipMalloc(&h->B.sring, sizeof(float2) * C * h->sring));
    HIPCHK(hipMalloc(&h->B.dring, sizeof(float2) * C * h->dring));
    {
        const size_t NJ = (((size_t)fm_per_call + 1 + WT - 1) / WT) * WT;      // whole work-array tiles (widx)
        h->work_nj = (int64_t)NJ;
        h->pitch = ((h->channels + 63) / 64) * 64 + 64;
        const size_t C = (size_t)h->pitch;   // rows are padded (see CallGeom.pitch)
        HIPCHK(hipMalloc(&h->B.w_dem, sizeof(float) * NJ * C));
        HIPCHK(hipMalloc(&h->B.w_cur, sizeof(float) * NJ * C));
        h->B.w_diff = nullptr;      // (flush_mailbox: FMX_P_SCOPE_TAPS)
        h->B.lockm_stride = (int32_t)(NJ / 6 + 512);
        HIPCHK(hipMalloc(&h->B.w_lockm, (size_t)h->B.lockm_stride * C));
        HIPCHK(hipMemset(h->B.w_lockm, 0, (size_t)h->B.lockm_stride * C));
        h->tail_ptrs.push_back(h->B.w_lockm);
        HIPCHK(hipMalloc(&h->B.gfix, sizeof(float2) * GAIN_FIX_FRAMES * C));
        HIPCHK(hipMemset(h->B.gfix, 0, sizeof(float2) * GAIN_FIX_FRAMES * C));
        h->B.w_iq = nullptr;
        // the recurrence kernels move whole 64-channel row blocks, pad columns included: keep those finite
        HIPCHK(hipMemset(h->B.w_dem, 0, sizeof(float) * NJ * C));
        HIPCHK(hipMemset(h->B.w_cur, 0, sizeof(float) * NJ * C));
    }
    HIPCHK(hipMalloc(&h->B.state, sizeof(ChanState) * C));
    HIPCHK(hipMalloc(&h->d_params, sizeof(ChanParams) * C));
    {   // PCM tail: one test-tone burst (fm-processor.cpp:808-813,818-821: float phase += float incr, PI_Constrain, sinf)
        // and the peak meter's per-tile maxima / window ring
        std::vector<float> tone((size_t)TT_BURST);
        float ph = 0.0f; const float inc = (float)(2 * design::kPi / cfg->workingRate * 1000.0f);
        for (int i = 0; i < TT_BURST; i++) {
            ph += inc;
            if (!(0 <= ph && ph < 2 * design::kPi)) ph = (float)std::fmod((double)ph, 2 * design::kPi);   // PI_Constrain fm-constants.h:149-153 (ph > 0 here)
            tone[i] = std::sin(ph);
        }
        float *d_tone = nullptr;
        HIPCHK(hipMalloc(&d_tone, sizeof(float) * TT_BURST));
        HIPCHK(hipMemcpy(d_tone, tone.data(), sizeof(float) * TT_BURST, hipMemcpyHostToDevice));
        h->B.tone = d_tone; h->tail_ptrs.push_back(d_tone);
        h->B.pk_tiles = (int32_t)((cfg->max_block / (4 * h->decim) + 96) / C_TILE + 8);
        HIPCHK(hipMalloc(&h->B.pk_part, sizeof(float4) * C * h->B.pk_tiles));
        HIPCHK(hipMalloc(&h->B.pk_ring, sizeof(float2) * C * PK_RING));
        HIPCHK(hipMemset(h->B.pk_part, 0, sizeof(float4) * C * h->B.pk_tiles));
        HIPCHK(hipMemset(h->B.pk_ring, 0, sizeof(float2) * C * PK_RING));
        h->tail_ptrs.push_back(h->B.pk_part); h->tail_ptrs.push_back(h->B.pk_ring);
    }
    HIPCHK(hipMemset(h->B.hist, 0, sizeof(float2) * CT * DECIM * A_HIST_COLS));
    HIPCHK(hipMemset(h->B.zring, 0, sizeof(float2) * C * h->ring));
    HIPCHK(hipMemset(h->B.sring, 0, sizeof(float2) * C * h->sring));
    HIPCHK(hipMemset(h->B.dring, 0, sizeof(float2) * C * h->dring));
    {
        ChanState s0; std::memset(&s0, 0, sizeof(s0));
        s0.Imin1 = s0.Qmin1 = s0.Imin2 = s0.Qmin2 = (float)0.01;        // fm-demodulator.cpp:79-82
        std::vector<ChanState> init(C, s0);
        HIPCHK(hipMemcpy(h->B.state, init.data(), sizeof(ChanState) * C, hipMemcpyHostToDevice));
    }
    h->B.params = h->d_params;
    return ensure_sets(h);
    };
    const int rc = init();
    if (rc) { const std::string msg = g_err; (void)fmx_destroy(h); g_err = msg; return rc; }
    *out = h;
    return FMX_OK;
}

int fmx_destroy(fmx_handle h) {
    if (!h) return FMX_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto &pr : h->prof) for (int i = 0; i < 4; i++) (void)hipEventDestroy(pr.e[i]);
    void *ptrs[] = { h->d_audio_spec, h->d_audio_lp, h->d_rs_taps, h->B.gfix, h->d_fft_w, h->d_pss_hs, h->d_front_taps, h->d_audio_taps, h->d_pss_taps, h->d_front_sets, h->d_audio_sets, h->d_sincos,
                     h->d_lo, h->d_atan, h->d_arcsine, h->d_trig3, h->d_params, h->B.hist, h->B.dcv_hist, h->B.zring,
                     h->B.sring, h->B.dring, h->B.state, h->d_iq, h->d_pcm, h->B.w_dem, h->B.w_iq, h->B.w_cur,
                     h->B.w_osc, h->d_cv_taps, h->d_x48 };      // (the LR tap's rows are in tail_ptrs)
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (void *p : h->rds_ptrs) if (p) (void)hipFree(p);
    for (void *p : h->tail_ptrs) if (p) (void)hipFree(p);
    if (h->d_old_sets) (void)hipFree(h->d_old_sets);
    if (h->step_host) (void)hipHostFree(h->step_host);
    for (hipEvent_t e : h->step_ev) if (e) (void)hipEventDestroy(e);
    if (h->hp_iq) (void)hipHostFree(h->hp_iq);
    if (h->hp_pcm) (void)hipHostFree(h->hp_pcm);
    for (hipStream_t st : {h->pipe_sA, h->pipe_sP, h->pipe_sB}) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (hipEvent_t e : {h->pipe_ev0, h->pipe_evA, h->pipe_evD, h->pipe_evP, h->pipe_evE, h->pipe_evB[0], h->pipe_evB[1]}) if (e) (void)hipEventDestroy(e);
    if (h->ev_in) (void)hipEventDestroy(h->ev_in);
    if (h->ev_dummy) (void)hipEventDestroy(h->ev_dummy);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return FMX_OK;
}

int fmx_set_param(fmx_handle h, int32_t channel, int32_t id, double value) {
    if (!h) return fail(FMX_E_INVALID, "null handle");
    if (channel < -1 || channel >= h->channels) return fail(FMX_E_INVALID, "channel out of range");
    const int iv = (int)std::llround(value);
    // validate once
    switch (id) {
    case FMX_P_FM_MODE: if (iv < 0 || iv > 2) return fail(FMX_E_INVALID, "fm mode must be 0..2"); break;
    case FMX_P_FM_DECODER:
        if (iv < 1 || iv > 6) return fail(FMX_E_INVALID, "decoder must be 1..6"); break;
    case FMX_P_SOUND_MODE: if (iv < 0 || iv > 6) return fail(FMX_E_INVALID, "sound mode must be 0..6"); break;
    case FMX_P_STEREO_PANORAMA: if (iv < 0 || iv > 200) return fail(FMX_E_INVALID, "panorama must be 0..200"); break;
    case FMX_P_SOUND_BALANCE: if (iv < -100 || iv > 100) return fail(FMX_E_INVALID, "balance must be -100..100"); break;
    case FMX_P_DEEMPHASIS: if (iv < 1) return fail(FMX_E_INVALID, "de-emphasis must be >= 1 us (Q_ASSERT fm-processor.cpp:293)"); break;
    case FMX_P_BANDWIDTH: if (iv < 0 || iv > h->cfg.inputRate) return fail(FMX_E_INVALID, "bandwidth out of range"); break;
    case FMX_P_RDS_MODE: if (iv < 0 || iv > 3) return fail(FMX_E_INVALID, "rds mode must be 0..3"); break;
    case FMX_P_LOCAL_OSCILLATOR:
        if (std::abs(iv) > h->cfg.inputRate) return fail(FMX_E_INVALID, "|lo| must be <= inputRate (oscillator.cpp:49-58)"); break;
    case FMX_P_SQUELCH_MODE:
        if (iv < 0 || iv > 2) return fail(FMX_E_INVALID, "squelch mode must be 0 (off), 1 (noise squelch) or 2 (level squelch)"); break;
    case FMX_P_SQUELCH_VALUE: if (iv < 0 || iv > 100) return fail(FMX_E_INVALID, "squelch value must be 0..100"); break;
    case FMX_P_PLL_SOLVER: if (iv < 0 || iv > 3) return fail(FMX_E_INVALID, "PLL solver must be 0 (automatic), 1 (sequential), 2 (Newton, sequential around lock decisions) or 3 (Newton always)"); break;
    case FMX_P_FRONT_KERNEL:
        if (iv < 0 || iv > 3 || iv == 2) return fail(FMX_E_INVALID, "front kernel must be 0 (automatic), 1 (four waves per channel, packed f32 FMAs) or 3 (the filter on the matrix pipe); 2 was round 5's six-wave kernel, now tools/experiments/fmx_front3.hip");
        h->front_kernel.store(iv); return FMX_OK;
    case FMX_P_SCOPE_TAPS:
        if (iv < -1 || iv > 1) return fail(FMX_E_INVALID, "scope taps must be -1 (automatic), 0 (not kept) or 1 (kept)");
        h->scope_taps.store(iv); return FMX_OK;
    case FMX_P_CALL_PIECES:
        if (iv < -1 || iv > (1 << 20)) return fail(FMX_E_INVALID, "call pieces must be -1 (automatic), 0 (never) or the fm samples per piece");
        h->pipe_rows.store(iv); return FMX_OK;
    case FMX_P_FRONT_PARTS:
        if (iv < 0 || iv > 32) return fail(FMX_E_INVALID, "front parts must be 0 (automatic), 1 (one workgroup per channel) or 2..32");
        h->front_parts.store(iv); return FMX_OK;
    case FMX_P_STAGEB_FORM:
        if (iv < 0 || iv > 2) return fail(FMX_E_INVALID, "stage B form must be 0 (automatic), 1 (one kernel) or 2 (two kernels)");
        h->stageb_form.store(iv); return FMX_OK;
    case FMX_P_FILTER_RESTARTS: {
        if (iv < 0 || iv > 2) return fail(FMX_E_INVALID, "filter restarts must be 0 (automatic), 1 (the reference's block filters) or 2 (folded FIRs)");
        std::lock_guard<std::mutex> lk(h->mtx);
        if (h->g_total != 0) return fail(FMX_E_UNSUPPORTED, "the filter structure of a handle is fixed by its first call");
        const bool want = iv == 1 || (iv == 0 && h->channels <= OLA_MAX_CH);
        if (want != h->ola_mode) { h->ola_mode = want; h->sets_dirty = true; }
        h->folded_pinned = iv == 2;
        return FMX_OK; }
    case FMX_P_DISP_DELAY: if (iv < 0 || iv > 100000) return fail(FMX_E_INVALID, "display delay must be 0..100000 steps"); break;
    case FMX_P_TEST_TONE:
    case FMX_P_VOLUME_DB: case FMX_P_LF_CUTOFF: case FMX_P_ATTENUATION_L: case FMX_P_ATTENUATION_R:
    case FMX_P_AUTO_MONO: case FMX_P_PSS: case FMX_P_DC_REMOVE:
    case FMX_A_TRIGGER_FREQUENCY_CHANGE: case FMX_A_RESTART_PSS: case FMX_A_RESET_RDS: break;
    default: return fail(FMX_E_INVALID, "unknown parameter id");
    }
    std::lock_guard<std::mutex> lk(h->mtx);
    const int c0 = channel < 0 ? 0 : channel, c1 = channel < 0 ? h->channels : channel + 1;
    if (id == FMX_A_RESET_RDS || id == FMX_A_TRIGGER_FREQUENCY_CHANGE) {
        if (h->rds_reset_req.size() != (size_t)h->channels) h->rds_reset_req.assign((size_t)h->channels, 0);
        for (int c = c0; c < c1; c++) h->rds_reset_req[(size_t)c] = 1;      // resetRds (:862-864); triggerFrequencyChange calls it (:852)
    }
    for (int c = c0; c < c1; c++) {
        ChanUser &u = h->user[c]; ChanParams &p = h->params[c];
        switch (id) {
        case FMX_P_FM_MODE: p.fm_mode = iv; break;
        case FMX_P_FM_DECODER: p.decoder = iv; break;
        case FMX_P_SOUND_MODE: p.sound_sel = iv; break;
        case FMX_P_STEREO_PANORAMA: u.panorama = iv; break;
        case FMX_P_SOUND_BALANCE: u.balance = iv; h->gain_dirty = true; break;
        case FMX_P_DEEMPHASIS: u.deemph_us = iv; break;
        case FMX_P_VOLUME_DB: u.volume_db = (float)value; u.ctor_volume = false; h->gain_dirty = true; break;
        case FMX_P_LF_CUTOFF: case FMX_P_BANDWIDTH: {
            if (id == FMX_P_LF_CUTOFF) { u.lf_cutoff = iv > 0 ? iv : 0; u.lf_event = iv > 0; } else { u.bandwidth = iv; u.bw_event = iv > 0; }
            // a folded handle in mid-stream: the setter stays pending until the handle has kept enough of its streams to become a block-machine handle
            // (promote); everywhere else -- before the first call, a block-machine handle, folded filters pinned -- it applies with the next call
            const bool defer = !h->ola_mode && !h->folded_pinned && h->g_total > 0 && h->twins >= 1 && h->cfg.max_block >= 4096;
            if (defer) { if (!h->promo_pending) { h->promo_pending = true; h->promo_have = 0; } }
            else { u.bw_applied = u.bandwidth; u.lf_applied = u.lf_cutoff; h->sets_dirty = true; }
            break; }
        case FMX_P_ATTENUATION_L: p.att_l = (float)value; h->promo_recapture = true; break;
        case FMX_P_ATTENUATION_R: p.att_r = (float)value; h->promo_recapture = true; break;
        case FMX_P_RDS_MODE: p.rds_mode = iv; break;
        case FMX_P_LOCAL_OSCILLATOR: {
            p.lo_freq = iv; h->promo_recapture = true;
            int64_t a = iv < 0 ? -(int64_t)iv : iv, b = h->cfg.inputRate;
            while (b) { const int64_t r = a % b; a = b; b = r; }                  // gcd(|lo|, inputRate)
            const int64_t per = a ? h->cfg.inputRate / a : 0;
            p.lo_period = (iv != 0 && per <= LO_LDS_MAX) ? (int32_t)per : 0;
            break;
        }
        case FMX_P_AUTO_MONO: p.auto_mono = iv != 0; break;
        case FMX_P_PSS: p.pss_active = iv != 0; break;
        case FMX_P_SQUELCH_MODE: p.squelch_mode = iv; break;
        case FMX_P_SQUELCH_VALUE: u.squelch_value = iv; break;
        case FMX_P_TEST_TONE: p.test_tone = iv != 0; break;
        case FMX_P_PLL_SOLVER: p.pll_seq = (iv == 1 || (iv == 0 && h->channels <= PLL_SEQ_AUTO_MAX)) ? 1 : (iv == 3 ? 2 : 0); break;
        case FMX_P_DISP_DELAY:                       // DelayLine::set_delay_steps fm-processor.h:60-63: resize keeps what is there
            u.delay.resize((size_t)iv + 1, make_float2(-40.0f, -40.0f)); u.delay_idx = 0; break;
        case FMX_P_DC_REMOVE: p.dc_remove = iv != 0; p.actions |= ACT_DC_RESET; h->promo_recapture = true; break;
        case FMX_A_TRIGGER_FREQUENCY_CHANGE: p.actions |= ACT_TRIGGER_FREQ; break;
        case FMX_A_RESTART_PSS: p.actions |= ACT_RESTART_PSS; break;
        default: break;
        }
        refresh_derived(h, c);
    }
    h->params_dirty = true;
    return FMX_OK;
}

int64_t fmx_filter_change_due(fmx_handle h) {
    if (!h) return -1;
    std::lock_guard<std::mutex> lk(h->mtx);
    if (!h->promo_pending || h->ola_mode) return -1;
    if (h->promo_recapture) return PROMO_TAIL_IN + 1;
    return h->promo_have >= PROMO_TAIL_IN ? 0 : PROMO_TAIL_IN - h->promo_have;
}

int64_t fmx_frames_for(fmx_handle h, int64_t n) {
    if (!h || n < 0) return -1;
    CallGeom G{}; frames_geom(h, n, &G);
    return conv2_out(h, G.M1) - conv2_out(h, G.M0);
}

static int bytes_per_sample(int32_t fmt) { return fmt == FMX_IQ_F32 ? 8 : (fmt == FMX_IQ_S16 ? 4 : 2); }

int fmx_process_device_raw(fmx_handle h, const void *d_iq, int32_t format, float s16_den, int64_t stream_stride, int64_t n,
                           float *d_pcm, int64_t pcm_stride, int64_t *n_frames, void *hip_stream) {
    if (!h || !d_iq || !d_pcm) return fail(FMX_E_INVALID, "null argument");
    if (stream_stride < n) return fail(FMX_E_INVALID, "stream_stride < n_complex");
    HIPCHK(hipSetDevice(h->cfg.device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
    if (!hip_stream) {
        // NULL also names HIP's legacy default stream, where such a caller's producer (e.g. PyTorch's current stream) runs;
        // the handle's stream is non-blocking, so order this call behind what is queued there now.  (Measured: the default
        // stream's implicit synchronisation with the CU-masked streams of stage B costs ~0.4 ms per call at 4096 channels;
        // a caller that passes a stream of its own, as bench.py does, does not pay it.)
        HIPCHK(hipEventRecord(h->ev_in, nullptr));
        HIPCHK(hipStreamWaitEvent(h->stream, h->ev_in, 0));
    }
    return run_call(h, d_iq, format, s16_den, stream_stride, n, reinterpret_cast<float2 *>(d_pcm), pcm_stride, n_frames, s);
}
int fmx_process_device(fmx_handle h, const float *d_iq, int64_t stream_stride, int64_t n, float *d_pcm,
                       int64_t pcm_stride, int64_t *n_frames, void *hip_stream) {
    return fmx_process_device_raw(h, d_iq, FMX_IQ_F32, 0.f, stream_stride, n, d_pcm, pcm_stride, n_frames, hip_stream);
}

int fmx_process_host_raw(fmx_handle h, const void *iq, int32_t format, float s16_den, int64_t stream_stride, int64_t n,
                         float *pcm, int64_t pcm_stride, int64_t *n_frames) {
    if (!h || !iq || !pcm) return fail(FMX_E_INVALID, "null argument");
    if (format < 0 || format > 3) return fail(FMX_E_INVALID, "unknown IQ format");
    if (stream_stride < n) return fail(FMX_E_INVALID, "stream_stride < n_complex");
    if (n <= 0 || n > h->cfg.max_block) return fail(FMX_E_TOO_LARGE, "n_complex must be in [1, max_block]");
    HIPCHK(hipSetDevice(h->cfg.device));
    const int64_t cap = conv2_out(h, h->cfg.max_block / (4 * h->decim) + 96) + 2;
    if (!h->d_iq) {
        HIPCHK(hipMalloc(&h->d_iq, sizeof(float2) * (size_t)h->streams * h->cfg.max_block));    // sized for the widest format
        HIPCHK(hipMalloc(&h->d_pcm, sizeof(float2) * (size_t)h->channels * cap));
        h->pcm_cap = cap;
    }
    const int64_t frames = fmx_frames_for(h, n);
    if (frames > pcm_stride) return fail(FMX_E_TOO_LARGE, "pcm_stride smaller than the frames this call produces");
    const size_t bps = (size_t)bytes_per_sample(format);
    // The single receiver's call (one stream, 16384 samples: 128 KB in, 2.7 KB out) spends a third of its time in the two copies' fixed costs
    // (a pageable buffer is staged by the runtime, each copy is a submission of its own).  Small calls go through pinned, device-visible
    // memory instead: the samples are copied there by the CPU and READ BY STAGE A over the bus, the PCM is written there by stage C.
    const bool zc_env = env_switches().host_zerocopy != 0;
    constexpr size_t ZC_MAX_IN = (size_t)1 << 20;
    const size_t in_bytes = bps * (size_t)n * (size_t)h->streams;
    if (zc_env && in_bytes <= ZC_MAX_IN && (size_t)h->channels * (size_t)cap * sizeof(float2) <= ZC_MAX_IN) {
        if (!h->hp_pcm) {
            HIPCHK(hipHostMalloc(&h->hp_iq, ZC_MAX_IN, hipHostMallocDefault)); h->hp_iq_bytes = ZC_MAX_IN;
            HIPCHK(hipHostMalloc((void **)&h->hp_pcm, sizeof(float2) * (size_t)h->channels * cap, hipHostMallocDefault)); h->hp_pcm_cap = cap;
        }
        for (int sidx = 0; sidx < h->streams; sidx++)
            std::memcpy((char *)h->hp_iq + (size_t)sidx * bps * n, (const char *)iq + (size_t)sidx * bps * stream_stride, bps * (size_t)n);
        int64_t got = 0;
        int rc = run_call(h, h->hp_iq, format, s16_den, n, n, h->hp_pcm, cap, &got, h->stream);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(h->stream));
        for (int c = 0; c < h->channels && got > 0; c++)
            std::memcpy(pcm + 2 * (size_t)c * pcm_stride, h->hp_pcm + (size_t)c * cap, sizeof(float2) * (size_t)got);
        if (n_frames) *n_frames = got;
        return FMX_OK;
    }
    HIPCHK(hipMemcpy2DAsync(h->d_iq, bps * h->cfg.max_block, iq, bps * stream_stride, bps * n, h->streams,
                            hipMemcpyHostToDevice, h->stream));
    int64_t got = 0;
    int rc = run_call(h, h->d_iq, format, s16_den, h->cfg.max_block, n, h->d_pcm, cap, &got, h->stream);
    if (rc) return rc;
    if (got > 0)
        HIPCHK(hipMemcpy2DAsync(pcm, sizeof(float2) * pcm_stride, h->d_pcm, sizeof(float2) * cap, sizeof(float2) * got,
                                h->channels, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (n_frames) *n_frames = got;
    return FMX_OK;
}
int fmx_process_host(fmx_handle h, const float *iq, int64_t stream_stride, int64_t n, float *pcm,
                     int64_t pcm_stride, int64_t *n_frames) {
    return fmx_process_host_raw(h, iq, FMX_IQ_F32, 0.f, stream_stride, n, pcm, pcm_stride, n_frames);
}

int fmx_synchronize(fmx_handle h) {
    if (!h) return fail(FMX_E_INVALID, "null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipDeviceSynchronize());
    return FMX_OK;
}

int fmx_get_meta(fmx_handle h, int32_t channel, fmx_meta *m) {
    if (!h || !m || channel < 0 || channel >= h->channels) return fail(FMX_E_INVALID, "bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipDeviceSynchronize());
    ChanState st;
    HIPCHK(hipMemcpy(&st, h->B.state + channel, sizeof(st), hipMemcpyDeviceToHost));
    m->DcValRf = st.meta_dc_rf; m->DcValIf = st.meta_dc_if; m->PssPhaseShiftDegree = st.meta_pss_deg;
    m->PssPhaseChange = st.meta_pss_change; m->PssState = st.meta_pss_state;
    m->PilotPllLockStrength = st.meta_lock_strength; m->PilotPllLocked = st.meta_locked;
    m->live_pilot_locked = (h->params[channel].fm_mode != 2) ? st.pil_locked : 0;
    m->live_lock_strength = (h->params[channel].fm_mode != 2) ? st.pil_lock : 0.f;
    m->live_dc_if = st.fm_afc; m->squelch_active = (h->params[channel].squelch_mode != 0) ? st.sq_suppress : 0;
    m->fm_samples = h->g_total / h->decim; m->pcm_frames = conv2_out(h, 48 * ((h->g_total / h->decim) / 192));
    m->live_rf_dc_re = st.dc_re; m->live_rf_dc_im = st.dc_im;
    return FMX_OK;
}

int fmx_get_peaks(fmx_handle h, int32_t channel, float *lr_db, int32_t capacity, int32_t *n_events) {
    if (!h || !n_events || channel < 0 || channel >= h->channels || capacity < 0 || (capacity > 0 && !lr_db))
        return fail(FMX_E_INVALID, "bad argument");
    if (!h->taps_kept) return fail(FMX_E_UNSUPPORTED, "this handle does not run the peak-level meter: a display feed, automatic only up to 64 channels -- fmx_set_param (h, -1, FMX_P_SCOPE_TAPS, 1) and one call switch it on");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipDeviceSynchronize());
    ChanState st;
    HIPCHK(hipMemcpy(&st, h->B.state + channel, sizeof(st), hipMemcpyDeviceToHost));
    std::lock_guard<std::mutex> lk(h->mtx);
    ChanUser &u = h->user[channel];
    if (st.pk_events - u.pk_read > PK_RING) u.pk_read = st.pk_events - PK_RING;      // the caller fell behind: the oldest windows are gone
    const int avail = st.pk_events - u.pk_read;
    const int take = std::min(avail, (int)capacity);
    if (take > 0) {
        std::vector<float2> ring((size_t)PK_RING);
        HIPCHK(hipMemcpy(ring.data(), h->B.pk_ring + (size_t)channel * PK_RING, sizeof(float2) * PK_RING, hipMemcpyDeviceToHost));
        for (int k = 0; k < take; k++) {
            const float2 pk = ring[(size_t)((u.pk_read + k) & (PK_RING - 1))];
            // fm-processor.cpp:785-794: float log10 (std::log10 of a float), -40 dB for silence, then the display delay line
            const float ldb = pk.x > 0.0f ? 20.0f * std::log10(pk.x) : -40.0f;
            const float rdb = pk.y > 0.0f ? 20.0f * std::log10(pk.y) : -40.0f;
            u.delay[u.delay_idx] = make_float2(ldb, rdb);
            u.delay_idx = (u.delay_idx + 1) % (uint32_t)u.delay.size();
            lr_db[2 * k] = u.delay[u.delay_idx].x; lr_db[2 * k + 1] = u.delay[u.delay_idx].y;
        }
        u.pk_read += take;
    }
    *n_events = take;
    return FMX_OK;
}

int fmx_get_tap(fmx_handle h, int32_t channel, int32_t tap, float *dst, int64_t n) {
    if (!h || !dst || channel < 0 || channel >= h->channels || n < 0) return fail(FMX_E_INVALID, "bad argument");
    const int64_t J1 = h->g_total / h->decim;
    if (tap != 4 && (n > J1 || n > (h->last_J1 - h->last_J0))) return fail(FMX_E_INVALID, "n exceeds the samples produced by the last call");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipDeviceSynchronize());
    const char *base; int64_t cap, elem, delay = 0;
    switch (tap) {
    case FMX_TAP_FM_IQ: base = (const char *)(h->B.zring + (size_t)channel * h->ring); cap = h->ring; elem = sizeof(float2);
        delay = h->h_front_sets[h->params[channel].front_set].delay_fm; break;
    case FMX_TAP_DEMOD: case FMX_TAP_LR_RAW: case FMX_TAP_PILOT_PHASE: {
        // these taps are read back from the last call's work arrays (channel-major rows of the call), rows [nj - n, nj)
        const int64_t nj = h->last_J1 - h->last_J0, r0 = nj - n;
        if (!h->taps_kept) return fail(FMX_E_UNSUPPORTED, "this handle does not keep the demodulator / LR / pilot-phase scope taps: display feeds, automatic only up to 64 channels -- fmx_set_param (h, -1, FMX_P_SCOPE_TAPS, 1) and one call switch them on");
        if (n == 0) return FMX_OK;
        {                            // this call's rows are contiguous per channel
            const size_t off = (size_t)channel * (size_t)h->work_nj + (size_t)r0;
            std::vector<float> a((size_t)n), b;
            HIPCHK(hipMemcpy(a.data(), (tap == FMX_TAP_PILOT_PHASE ? h->B.w_cur : h->B.w_dem) + off, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
            if (tap == FMX_TAP_LR_RAW) {
                b.resize((size_t)n);
                HIPCHK(hipMemcpy(b.data(), h->B.w_diff + off, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
                for (int64_t i = 0; i < n; i++) { dst[2 * i] = a[(size_t)i]; dst[2 * i + 1] = b[(size_t)i]; }
            } else std::memcpy(dst, a.data(), sizeof(float) * (size_t)n);
        }
        return FMX_OK; }
    case FMX_TAP_PRE_RESAMPLER: base = (const char *)((h->ola_mode && h->d2ring ? h->d2ring : h->B.dring) + (size_t)channel * h->dring); cap = h->dring; elem = sizeof(float2); break;
    case 4: {   // FMX_TAP_RDS_IQ: complex @24 kS/s after rdsDecimator (:553): the last n outputs of the last call
        if (!h->rds_alloc) return fail(FMX_E_INVALID, "RDS is off");
        const int64_t lm0 = h->last_m0[(size_t)channel], lm1 = h->last_m1[(size_t)channel];      // (the channel's own count: fmx_last_rds_samples_of)
        if (n > lm1 - lm0) return fail(FMX_E_INVALID, "n exceeds the RDS samples the channel produced in the last call");
        char *o = (char *)dst;
        for (int64_t m = lm1 - n; m < lm1;) {
            const int64_t pos = m & (RDS24_RING - 1);
            const int64_t run = std::min<int64_t>(RDS24_RING - pos, lm1 - m);
            HIPCHK(hipMemcpy(o, (const char *)(h->R.rds24 + (size_t)channel * RDS24_RING) + pos * sizeof(float2), (size_t)run * sizeof(float2), hipMemcpyDeviceToHost));
            o += run * sizeof(float2); m += run;
        }
        return FMX_OK; }
    default: return fail(FMX_E_INVALID, "unknown tap id");
    }
    // samples j in [J1-n, J1) live at ring index (j - delay) & (cap-1); before the stream start they are 0
    char *out = (char *)dst;
    for (int64_t j = J1 - n; j < J1;) {
        const int64_t jv = j - delay;
        if (jv < 0) { std::memset(out, 0, elem); out += elem; j++; continue; }
        const int64_t pos = jv & (cap - 1);
        const int64_t run = std::min<int64_t>(cap - pos, J1 - j);
        HIPCHK(hipMemcpy(out, base + pos * elem, (size_t)(run * elem), hipMemcpyDeviceToHost));
        out += run * elem; j += run;
    }
    return FMX_OK;
}

int fmx_rds_bits(fmx_handle h, int32_t channel, uint8_t *bits, int32_t capacity, int32_t *n_bits) {
    if (!h || channel < 0 || channel >= h->channels || !n_bits || capacity < 0) return fail(FMX_E_INVALID, "bad argument");
    *n_bits = 0;
    if (!h->rds_alloc) return FMX_OK;
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipDeviceSynchronize());
    RdsState st;
    HIPCHK(hipMemcpy(&st, h->R.state + channel, sizeof(st), hipMemcpyDeviceToHost));
    int32_t have = st.nbits - h->rds_read[channel];
    if (have > RDS_BITS_CAP) { h->rds_read[channel] = st.nbits - RDS_BITS_CAP; have = RDS_BITS_CAP; }   // ring overrun: oldest bits lost
    const int32_t take = have < capacity ? have : capacity;
    std::vector<uint8_t> ring((size_t)RDS_BITS_CAP);
    if (take > 0 && bits) {
        HIPCHK(hipMemcpy(ring.data(), h->R.bits + (size_t)channel * RDS_BITS_CAP, RDS_BITS_CAP, hipMemcpyDeviceToHost));
        for (int32_t i = 0; i < take; i++) bits[i] = ring[(size_t)((h->rds_read[channel] + i) & (RDS_BITS_CAP - 1))];
        h->rds_read[channel] += take;
    }
    *n_bits = (take > 0 && bits) ? take : 0;
    return FMX_OK;
}

int fmx_rds_symbols(fmx_handle h, int32_t channel, float *iq, int32_t capacity, int32_t *n_symbols) {
    if (!h || channel < 0 || channel >= h->channels || !n_symbols || capacity < 0) return fail(FMX_E_INVALID, "bad argument");
    *n_symbols = 0;
    if (!h->rds_alloc) return FMX_OK;
    if ((int)h->rds_read_sym.size() != h->channels) h->rds_read_sym.assign((size_t)h->channels, 0);
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipDeviceSynchronize());
    RdsState st;
    HIPCHK(hipMemcpy(&st, h->R.state + channel, sizeof(st), hipMemcpyDeviceToHost));
    int32_t &rd = h->rds_read_sym[(size_t)channel];
    if (h->rds_gen_sym.size() != (size_t)h->channels) h->rds_gen_sym.assign((size_t)h->channels, 0);
    const int32_t gen = h->rds_gen.load();
    if (h->rds_gen_sym[(size_t)channel] != gen) { h->rds_gen_sym[(size_t)channel] = gen; rd = 0; }   // (the generation moves with resetRds only)
    int32_t have = st.nbits - rd;
    if (have < 0) { rd = 0; have = st.nbits; }
    if (have > RDS_SYM_CAP) { rd = st.nbits - RDS_SYM_CAP; have = RDS_SYM_CAP; }        // ring overrun: oldest symbols lost
    const int32_t take = have < capacity ? have : capacity;
    if (take > 0 && iq) {
        std::vector<float2> ring((size_t)RDS_SYM_CAP);
        HIPCHK(hipMemcpy(ring.data(), h->R.sym + (size_t)channel * RDS_SYM_CAP, sizeof(float2) * RDS_SYM_CAP, hipMemcpyDeviceToHost));
        for (int32_t i = 0; i < take; i++) { const float2 v = ring[(size_t)((rd + i) & (RDS_SYM_CAP - 1))]; iq[2 * i] = v.x; iq[2 * i + 1] = v.y; }
        rd += take;
        *n_symbols = take;
    }
    return FMX_OK;
}

int64_t fmx_last_fm_samples(fmx_handle h) { return h ? (int64_t)(h->last_J1 - h->last_J0) : 0; }
int64_t fmx_pll_replays(fmx_handle h, int32_t channel) {
    if (!h || channel >= h->channels) return (int64_t)fail(FMX_E_INVALID, "bad argument");
    if (hipSetDevice(h->cfg.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return (int64_t)fail(FMX_E_HIP, "device error");
    const int c0 = channel < 0 ? 0 : channel, c1 = channel < 0 ? h->channels : channel + 1;
    std::vector<ChanState> st((size_t)(c1 - c0));
    if (hipMemcpy(st.data(), h->B.state + c0, sizeof(ChanState) * st.size(), hipMemcpyDeviceToHost) != hipSuccess) return (int64_t)fail(FMX_E_HIP, "device error");
    int64_t n = 0;
    for (auto &s : st) n += s.pll_replays;
    return n;
}
int64_t fmx_pll_exact_segments(fmx_handle h, int32_t channel) {
    if (!h || channel >= h->channels) return (int64_t)fail(FMX_E_INVALID, "bad argument");
    if (hipSetDevice(h->cfg.device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return (int64_t)fail(FMX_E_HIP, "device error");
    const int c0 = channel < 0 ? 0 : channel, c1 = channel < 0 ? h->channels : channel + 1;
    std::vector<ChanState> st((size_t)(c1 - c0));
    if (hipMemcpy(st.data(), h->B.state + c0, sizeof(ChanState) * st.size(), hipMemcpyDeviceToHost) != hipSuccess) return (int64_t)fail(FMX_E_HIP, "device error");
    int64_t n = 0;
    for (auto &s : st) n += s.pll_exact_segs;
    return n;
}
int32_t fmx_last_front_kernel(fmx_handle h) { return h ? h->last_front_kernel : 0; }
int32_t fmx_last_call_pieces(fmx_handle h) { return h ? h->last_pieces : 0; }
int32_t fmx_last_second_group(fmx_handle h) { return h ? h->last_second_group : 0; }
int64_t fmx_last_rds_samples_of(fmx_handle h, int32_t channel) {
    if (!h || !h->rds_alloc || channel < 0 || channel >= h->channels) return 0;
    return h->last_m1[(size_t)channel] - h->last_m0[(size_t)channel];
}
int64_t fmx_last_rds_samples(fmx_handle h) { return fmx_last_rds_samples_of(h, 0); }

int fmx_rds_decode(fmx_handle h, int32_t channel, fmx_rds_info *info) {
    if (!h || channel < 0 || channel >= h->channels || !info) return fail(FMX_E_INVALID, "bad argument");
    if ((int)h->rds_dec.size() != h->channels) { h->rds_dec.assign((size_t)h->channels, fmx::RdsGroupDecoderHost()); h->rds_read_dec.assign((size_t)h->channels, 0); }
    fmx::RdsGroupDecoderHost &D = h->rds_dec[(size_t)channel];
    bool do_reset = false;
    {
        std::lock_guard<std::mutex> lk(h->mtx);
        if (h->rds_reset_req.size() == (size_t)h->channels && h->rds_reset_req[(size_t)channel]) { h->rds_reset_req[(size_t)channel] = 0; do_reset = true; }
    }
    if (h->rds_alloc) {
        HIPCHK(hipSetDevice(h->cfg.device));
        HIPCHK(hipDeviceSynchronize());
        RdsState st;
        HIPCHK(hipMemcpy(&st, h->R.state + channel, sizeof(st), hipMemcpyDeviceToHost));
        int32_t &rd = h->rds_read_dec[(size_t)channel];
        if (h->rds_gen_dec.size() != (size_t)h->channels) h->rds_gen_dec.assign((size_t)h->channels, 0);
        const int32_t gen = h->rds_gen.load();
        if (h->rds_gen_dec[(size_t)channel] != gen) { h->rds_gen_dec[(size_t)channel] = gen; rd = 0; D.reset_all(); }   // (the generation moves with resetRds only)
        int32_t have = st.nbits - rd;
        if (have < 0) { rd = 0; have = st.nbits; D.reset_all(); }
        if (do_reset) {
            // rdsGroupDecoder::reset: PI / PTY / labels back to unknown.  The reference resets between two blocks of samples;
            // here the decoder runs behind the slicer, so the bits still pending belong to the time before the reset (the
            // old station after a retune) and are dropped.
            D.reset_groups(); rd = st.nbits; have = 0;
        }
        if (have > RDS_BITS_CAP) { rd = st.nbits - RDS_BITS_CAP; have = RDS_BITS_CAP; }   // ring overrun: oldest bits lost
        if (have > 0) {
            std::vector<uint8_t> ring((size_t)RDS_BITS_CAP);
            HIPCHK(hipMemcpy(ring.data(), h->R.bits + (size_t)channel * RDS_BITS_CAP, RDS_BITS_CAP, hipMemcpyDeviceToHost));
            for (int32_t i = 0; i < have; i++) D.push_bit(ring[(size_t)((rd + i) & (RDS_BITS_CAP - 1))] != 0);
            rd += have;
        }
    }
    else if (do_reset) D.reset_groups();
    *info = D.info();
    return FMX_OK;
}

// host-only entry (no device needed): run a fresh block synchroniser / group decoder over a bit array
int fmx_rds_decode_bits(const uint8_t *bits, int32_t n_bits, fmx_rds_info *info) {
    if ((!bits && n_bits > 0) || n_bits < 0 || !info) return fail(FMX_E_INVALID, "bad argument");
    fmx::RdsGroupDecoderHost D;
    for (int32_t i = 0; i < n_bits; i++) D.push_bit(bits[i] != 0);
    *info = D.info();
    return FMX_OK;
}

const char *fmx_rds_pty_name(int32_t pty_code, int32_t pty_locale) { return fmx::rds_pty_name(pty_code, pty_locale); }
uint16_t fmx_rds_map_char(uint8_t alfabet, uint8_t character) { return fmx::rds_map_char(alfabet, character); }
int32_t fmx_rds_prepare_text(const uint8_t *v, int32_t length, uint8_t *alfabet, uint16_t *out, int32_t capacity) {
    if (!v || !out || length < 0 || capacity < 0) { (void)fail(FMX_E_INVALID, "bad argument"); return -1; }
    return fmx::rds_prepare_text(v, length, alfabet, out, capacity);
}

int fmx_get_taps(fmx_handle h, int32_t channel, int32_t which, float *dst, int32_t capacity, int32_t *n) {
    if (!h || !dst || !n || channel < 0 || channel >= h->channels) return fail(FMX_E_INVALID, "bad argument");
    {   // the tap sets of the CURRENT settings; nothing is uploaded and no pending action is touched (introspection)
        std::lock_guard<std::mutex> lk(h->mtx);
        if (h->sets_dirty) { HIPCHK(hipSetDevice(h->cfg.device)); int rc = ensure_sets(h); if (rc) return rc; h->params_dirty = true; }
    }
    const float *src = nullptr; int cnt = 0;
    std::vector<float> tmp;
    switch (which) {
    case 0: {   // front-end taps back in FIR order G[k], k = 12 d + off - r
        const FrontSet &fs = h->h_front_sets[h->params[channel].front_set];
        const float *t = &h->h_front_taps[(size_t)h->params[channel].front_set * A_TAPS_STRIDE];
        int NT = 0;
        tmp.assign(A_TAPS_STRIDE, 0.f);
        for (int d = 0; d < fs.nd; d++) for (int r = 0; r < DECIM; r++) {
            int k = 12 * d + fs.off - r;
            if (k >= 0 && k < A_TAPS_STRIDE) { tmp[k] = t[(d + 1) * DECIM + r]; if (t[(d + 1) * DECIM + r] != 0.f) NT = std::max(NT, k + 1); }
        }
        src = tmp.data(); cnt = NT; break; }
    case 1: src = h->h_pss_taps.data(); cnt = PSS_TAPS; break;
    case 2: {
        const AudioSet &as = h->h_audio_sets[h->params[channel].audio_set];
        const float *t = &h->h_audio_taps[(size_t)h->params[channel].audio_set * C_TAPS_STRIDE];
        tmp.resize(as.ntaps);
        for (int k = 0; k < as.ntaps; k++) tmp[k] = t[as.ntaps - 1 - k];
        src = tmp.data(); cnt = as.ntaps; break; }
    case 3: src = h->h_rs_taps.data(); cnt = RS_TAPS; break;
    case 4: {   // noise-squelch filters as the kernel holds them: [2][10][A1 A2 B1 B2], then the two gains (high-pass first)
        const design::Iir hp = design::iir_chebyshev_lowhigh(true, 20, 70000 - 100, h->cfg.fmRate);
        const design::Iir lp = design::iir_chebyshev_lowhigh(false, 20, 70000, h->cfg.fmRate);
        tmp.assign(2 * NSQ_QUADS * 4 + 2, 0.f);
        for (int f = 0; f < 2; f++) {
            const design::Iir &F = f ? lp : hp;
            for (int i = 0; i < NSQ_QUADS; i++) { tmp[(f * NSQ_QUADS + i) * 4] = F.q[i][1]; tmp[(f * NSQ_QUADS + i) * 4 + 1] = F.q[i][2]; tmp[(f * NSQ_QUADS + i) * 4 + 2] = F.q[i][4]; tmp[(f * NSQ_QUADS + i) * 4 + 3] = F.q[i][5]; }
            tmp[2 * NSQ_QUADS * 4 + f] = F.gain;
        }
        src = tmp.data(); cnt = (int)tmp.size(); break; }
    case 5: {   // RDS_1 constants as the kernels hold them: rdsFilter taps [21], Match kernel [43], sharpFilter [8][A1 A2 B1 B2], gain
        tmp = design::lowpass(RDS1_FIR, 2 * 2400, 24000);
        const std::vector<float> mk = design::rds1_match_kernel(24000);
        tmp.insert(tmp.end(), mk.begin(), mk.end());
        const design::Iir bp = design::iir_butterworth_bandpass(7, (int32_t)(1187.5 - 6), (int32_t)(1187.5 + 6), 24000);
        for (int i = 0; i < bp.nq; i++) { tmp.push_back(bp.q[i][1]); tmp.push_back(bp.q[i][2]); tmp.push_back(bp.q[i][4]); tmp.push_back(bp.q[i][5]); }
        tmp.push_back(bp.gain);
        src = tmp.data(); cnt = (int)tmp.size(); break; }
    default: return fail(FMX_E_INVALID, "unknown tap-set id");
    }
    if (cnt > capacity) return fail(FMX_E_TOO_LARGE, "capacity too small");
    std::memcpy(dst, src, sizeof(float) * cnt);
    *n = cnt;
    return FMX_OK;
}

// diagnostics (include/fmx_debug.h): streaming bandwidth of this GPU in GB/s over `bytes` of float2 data, the mean of
// `iters` launches timed with HIP events.  mode 0: copy (counts bytes read + written); mode 1: read 12, write 1 (stage A's shape).
int fmx_debug_stream_bandwidth(int32_t device, int32_t mode, int64_t bytes, int32_t iters, double *gbps) {
    if (!gbps || bytes < (1 << 20) || iters < 1 || mode < 0 || mode > 2) return fail(FMX_E_INVALID, "bad argument");
    HIPCHK(hipSetDevice(device));
    const size_t n16 = (size_t)bytes / 16 / (6 * 64) * (6 * 64);
    fmx::f32x4_t *src = nullptr, *dst = nullptr;
    HIPCHK(hipMalloc(&src, n16 * 16));
    if (hipMalloc(&dst, mode == 0 ? n16 * 16 : n16 * 16 / 12 + 64) != hipSuccess) { (void)hipFree(src); return fail(FMX_E_HIP, "hipMalloc"); }
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipMemset(src, 0, n16 * 16));
    const int grid = 256 * 8;
    for (int it = -2; it < iters; it++) {
        if (it == 0) HIPCHK(hipEventRecord(e0, 0));
        if (mode == 0) hipLaunchKernelGGL(fmx::stream_probe_kernel<0>, dim3(grid), dim3(256), 0, 0, src, dst, n16);
        else if (mode == 1) hipLaunchKernelGGL(fmx::stream_probe_kernel<1>, dim3(grid), dim3(256), 0, 0, src, dst, n16);
        else hipLaunchKernelGGL(fmx::stream_probe_kernel<2>, dim3(grid), dim3(256), 0, 0, src, dst, n16);
    }
    HIPCHK(hipEventRecord(e1, 0));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    const double moved = mode == 0 ? 2.0 * n16 * 16 : (mode == 1 ? n16 * 16 * (1.0 + 1.0 / 12) : (double)n16 * 16);
    *gbps = moved * iters / (ms * 1e-3) * 1e-9;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(src); (void)hipFree(dst);
    return FMX_OK;
}

// diagnostics (include/fmx_debug.h): per-phase shader-cycle counters of front_kernel, summed over channels
int fmx_debug_phase_cycles(fmx_handle h, int32_t enable, unsigned long long *out /*[DBG_SLOTS = 96], may be null*/) {
    if (!h) return fail(FMX_E_INVALID, "null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipDeviceSynchronize());
    const size_t nb = sizeof(unsigned long long) * DBG_SLOTS * (size_t)h->channels;
    if (out && h->B.dbg) {
        std::vector<unsigned long long> tmp(DBG_SLOTS * (size_t)h->channels);
        HIPCHK(hipMemcpy(tmp.data(), h->B.dbg, nb, hipMemcpyDeviceToHost));
        for (int k = 0; k < DBG_SLOTS; k++) { out[k] = 0; for (int c = 0; c < h->channels; c++) out[k] += tmp[(size_t)c * DBG_SLOTS + k]; }
    }
    if (enable && !h->B.dbg) { HIPCHK(hipMalloc(&h->B.dbg, nb)); }
    if (h->B.dbg) HIPCHK(hipMemset(h->B.dbg, 0, nb));
    if (!enable && h->B.dbg) { (void)hipFree(h->B.dbg); h->B.dbg = nullptr; }
    return FMX_OK;
}

int fmx_profile_enable(fmx_handle h, int32_t on) {
    if (!h) return fail(FMX_E_INVALID, "null handle");
    h->prof_on = on != 0;
    return FMX_OK;
}
int fmx_profile_read(fmx_handle h, fmx_profile *out, int32_t reset) {
    if (!h || !out) return fail(FMX_E_INVALID, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    int rc = prof_drain(h);
    if (rc) return rc;
    *out = h->prof_acc;
    if (reset) std::memset(&h->prof_acc, 0, sizeof(h->prof_acc));
    return FMX_OK;
}

}  // extern "C"
