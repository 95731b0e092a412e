// fmx_audio.hip -- stage C: 192 kS/s stereo -> 48 kS/s PCM.
//
// Replaces per channel:
//   fmAudioFilter (8192-pt overlap-add, 756 taps)  fm-processor.cpp:76,589-591, fft-filters.cpp:132-163
//   newConverter audioDecimator (192k -> 48k)      fm-processor.cpp:380,633-634, newconverter.cpp:55-80
//   start-up fade                                  fm-processor.cpp:636-642
//   insertTestTone, evaluatePeakLevel              fm-processor.cpp:800-823, 772-798
//
// MI355X design: the audio low-pass and the decimate-by-4 resampler are both LTI, so they are
// folded on the host into ONE polyphase FIR (756 + 128 - 1 = 883 taps) that is only evaluated at
// the 48 kHz output instants (4x fewer MACs than filtering at 192 kS/s); de-emphasis commutes with it and is
// applied upstream (fmx_demod.hip), the volume/balance gain at the FIR output.  The
// overlap-add latency (7436 samples = 1859 PCM frames) is reproduced by reading further back in
// the per-channel d ring.  libsamplerate itself is third-party and absent: the resampler is the
// documented fmx design (oracle/fm_oracle.c fmo_resampler_taps), "parity unpinned" for that stage.
#include "fmx_internal.h"
#include "fmx_fftconv.h"
#include <cstdlib>

namespace fmx {

typedef float v2f __attribute__((ext_vector_type(2)));

// PCM tail bookkeeping, one thread per channel, after audio_fft_kernel: folds the tiles' maxima into the open window, writes
// the maxima of every window that closed in this call to the channel's ring (the host turns them into dB and runs the
// display delay line: fmx_get_peaks), advances the test-tone cycle position.
__global__ __launch_bounds__(64) void pcm_tail_kernel(DeviceBuffers B, CallGeom G, int channels) {
    const int ch = blockIdx.x * 64 + threadIdx.x + G.ch0;
    if (ch >= G.ch0 + channels) return;
    ChanState *st = &B.state[ch];
    const int frames = (int)(G.M1 - G.M0);
    const int cnt0 = st->pk_cnt;
    float L = st->pk_l, R = st->pk_r;
    int ev = st->pk_events;
    const int tiles = B.peaks_on ? (frames + C_TILE - 1) / C_TILE : 0;      // (the meter off: the window grid goes on counting, nothing is folded or handed out)
    for (int tile = 0; tile < tiles; tile++) {
        const int i0 = tile * C_TILE, nfr = min(C_TILE, frames - i0);
        const float4 p = B.pk_part[(size_t)ch * B.pk_tiles + tile];
        const int w_first = (cnt0 + i0) / PK_WIN;
        const int to_end = (w_first + 1) * PK_WIN - (cnt0 + i0);          // frames of the tile's first window from i0 on
        L = fmaxf(L, p.x); R = fmaxf(R, p.y);
        if (to_end <= nfr) {                                              // the window closes inside this tile
            B.pk_ring[(size_t)ch * PK_RING + (ev & (PK_RING - 1))] = make_float2(L, R);
            ev++;
            L = p.z; R = p.w;
        }
    }
    st->pk_l = L; st->pk_r = R; st->pk_events = ev; st->pk_cnt = (cnt0 + frames) % PK_WIN;
    if (B.params[ch].test_tone) st->tt_pos = (int)(((int64_t)st->tt_pos + frames) % TT_CYCLE);
}

// The same FIR by fast convolution (fmx_fftconv.h) -- the reference's own method for its audio filter (fft-filters.cpp:132-163,
// 8192 points).  The folded FIR is only wanted at every fourth fm sample, so it is split into its four decimation phases:
//   out[m] = sum_p (g_p * x_p)[m],  g_p[i] = g[4 i + p] (221 taps),  x_p[n] = d[4 n + 3 - delay - p]   (all at 48 kHz),
// four forward transforms of 2048 points (the (L, R) pair rides as one complex number: the taps are real), the four spectra
// products summed in registers, ONE backward transform: 2048 - 220 outputs, of which a workgroup keeps 1792 = seven 256-frame
// tiles (the peak meter's unit).  A third of the direct form's instructions.  One 256-thread workgroup per (block, channel);
// the four phases of a thread's eight window entries are 32 contiguous bytes of the d ring each.
constexpr int AF_HIST = (C_MAX_TAPS + 3) / 4 - 1;        // 220 frames of history in front of a block
constexpr int AF_VALID = 7 * C_TILE;                     // frames a block delivers
static_assert(AF_HIST + AF_VALID <= fftc::N, "block + history fit one transform");
static_assert(AUDIO_DELAY % 4 == 0, "the four phases of a frame are one aligned group of four ring entries");
#ifndef AF_WAVES_PER_SIMD
#define AF_WAVES_PER_SIMD 3
#endif
template <bool PEAKS>      // (the peak meter's maxima: a display feed that a batch does not take, FMX_P_SCOPE_TAPS -- 7 % of the kernel)
__global__ __launch_bounds__(fftc::T, AF_WAVES_PER_SIMD) void audio_fft_kernel(DeviceTables T, DeviceBuffers B, CallGeom G, float2 *__restrict__ pcm) {
    __shared__ __attribute__((aligned(16))) float2 X[fftc::LDS_N];
    __shared__ int pkt[7][4];
    const int ch = blockIdx.y + G.ch0, t = threadIdx.x, lane = t & 63;
    const int64_t mb = G.M0 + (int64_t)blockIdx.x * AF_VALID;
    if (mb >= G.M1) return;
    const ChanParams P = B.params[ch];
    const AudioSet AS = T.audio_sets[P.audio_set];
    const float2 *__restrict__ dring = B.dring + (size_t)ch * (G.dring_mask + 1);
    const float2 *__restrict__ Gs = T.audio_spec + (size_t)P.audio_set * 4 * fftc::N;
    if (PEAKS && t < 28) pkt[t >> 2][t & 3] = 0;
    // window entry n' <-> frame mb - 220 + n': its four phases are d[4 (mb - 220 + n') - delay + 0 .. 3], phase p = entry 3 - p.
    // A phase's eight entries are loaded in front of its transform (the four phases share their cache lines; holding all four
    // windows in registers costs 48 VGPRs and a third of the occupancy)
    float2 acc[8];
#pragma unroll
    for (int q = 0; q < 8; q++) acc[q] = make_float2(0.f, 0.f);
    auto load_phase = [&](int p, float2 *a) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int64_t fr = mb - AF_HIST + t + fftc::T * k;
            const int64_t f0 = 4 * fr - AS.delay;
            a[k] = (f0 >= 0 && fr < G.M1) ? dring[(f0 + 3 - p) & G.dring_mask] : make_float2(0.f, 0.f);     // (frames past the call's last are never used)
        }
    };
    float2 a[8];
    load_phase(0, a);
#pragma unroll 1
    for (int p = 0; p < 4; p++) {
        __syncthreads();                                  // (the previous transform's last reads of X are done)
        fftc::forward_slots(t, a, X, T.fft_w);
        fftc::times_spectrum(t, a, Gs + (size_t)p * fftc::N);
#pragma unroll
        for (int q = 0; q < 8; q++) { acc[q].x += a[q].x; acc[q].y += a[q].y; }
        if (p < 3) load_phase(p + 1, a);                  // (a second register set that lands under the transform spills: measured, slower)
    }
    __syncthreads();
    fftc::backward_slots(t, acc, X, T.fft_w);
    // ---- per frame: gain, fade, test tone, peaks, store (as audio_kernel)
    const float gl = P.volume * P.left_ch, gr = P.volume * P.right_ch;
    const ChanState *__restrict__ st = &B.state[ch];
    const int64_t F = st->fade_start_frame;
    const int Max = 24000;
    const int cnt0 = st->pk_cnt, tt0 = st->tt_pos;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int rel = t + fftc::T * k - AF_HIST;        // frame of the block
        const int64_t m = mb + rel;
        const bool live = rel >= 0 && rel < AF_VALID && m < G.M1;
        const int tile = rel >= 0 ? rel / C_TILE : 0;
        float al = acc[k].x * gl, ar = acc[k].y * gr;
        float pv[4] = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            if (G.gain_fix && m - G.M0 < GAIN_FIX_FRAMES) {
                const float2 c = B.gfix[(size_t)ch * GAIN_FIX_FRAMES + (m - G.M0)];
                al += c.x; ar += c.y;
            }
            const int64_t since = m - F;                  // start-up fade fm-processor.cpp:638-642
            if (since >= 0 && since < Max) {
                const float cnt = (float)(Max - (int)since);
                const float f = ((float)Max - cnt) / (float)Max;
                al *= f; ar *= f;
            }
            const int i = (int)(m - G.M0);
            if (P.test_tone) {                            // insertTestTone fm-processor.cpp:800-823 (see audio_kernel)
#pragma clang fp contract(off)
                const float level = 0.9f;
                al = al * (1.0f - level); ar = ar * (1.0f - level);
                const int pos = (int)(((int64_t)tt0 + i) % TT_CYCLE);
                if (pos >= TT_SILENT) {
                    const float smpl = level * B.tone[pos - TT_SILENT];
                    al = al + smpl; ar = ar + smpl;
                }
            }
            const int i_tile = (int)(mb - G.M0) + tile * C_TILE;
            const bool second = (cnt0 + i) / PK_WIN != (cnt0 + i_tile) / PK_WIN;
            if (PEAKS) { pv[second ? 2 : 0] = fabsf(al); pv[second ? 3 : 1] = fabsf(ar); }
            pcm[(size_t)ch * G.pcm_stride + (m - G.M0)] = make_float2(al, ar);
        }
        // peak maxima per 256-frame tile: for a given k a wave's frames lie in one tile except the wave that holds t = 220
        if (!PEAKS) {
        } else if (t < 192) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                // (the wave's maximum of four non-negative values per step k: 32 reductions per thread -- as DPP moves on the vector pipe, not
                // as 192 ds_bpermute that queue up with the transforms' exchanges; non-negative floats order like their bit patterns)
                int v = __float_as_int(pv[q]);
#define AF_MAX_STEP(ctrl, rmask) v = max(v, __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, false))
                AF_MAX_STEP(0x111, 0xf); AF_MAX_STEP(0x112, 0xf); AF_MAX_STEP(0x114, 0xf); AF_MAX_STEP(0x118, 0xf);   // row_shr 1, 2, 4, 8: lane 15 of a row has the row's
                AF_MAX_STEP(0x142, 0xa); AF_MAX_STEP(0x143, 0xc);                                                     // row_bcast 15 / 31: lane 63 has the wave's
#undef AF_MAX_STEP
                if (lane == 63 && k >= 1) atomicMax(&pkt[k - 1][q], v);
            }
        } else if (live) {
#pragma unroll
            for (int q = 0; q < 4; q++) atomicMax(&pkt[tile][q], __float_as_int(pv[q]));
        }
    }
    if (!PEAKS) return;
    __syncthreads();
    const int tiles = (int)(((G.M1 - mb) < AF_VALID ? (G.M1 - mb) : AF_VALID) + C_TILE - 1) / C_TILE;
    if (t < tiles) B.pk_part[(size_t)ch * B.pk_tiles + 7 * blockIdx.x + t] = make_float4(__int_as_float(pkt[t][0]), __int_as_float(pkt[t][1]), __int_as_float(pkt[t][2]), __int_as_float(pkt[t][3]));
}


// A volume / balance change between two calls (fm-processor.cpp:299-306, 630): the reference multiplies the sample that enters
// the resampler, so PCM frame m = sum_k h_rs[k] g(4 m + 3 - k) a[4 m + 3 - k] with a = audio low-pass output; for the frames
// whose window reaches behind the call's first fm sample J0 the old gain still weighs in.  audio_kernel computes
// g_new * (folded FIR); this kernel computes the rest, (g_old - g_new) * sum over the window entries older than J0, for the
// call's first frames: A[q] = a[J0 - q] (one thread each, 756 taps of the d ring), then
// corr[r] = sum_q h_rs[e_r + q] A[q], e_r = 4 (M0 + r) + 3 - J0.  One workgroup per channel; channels whose gain did not change
// write zeros.  Also records the gains for the next change.
// The call's frames begin at the 192-sample block J0 falls into: e_0 is 3 when J0 is a multiple of 192 and as low as -188 otherwise -- a frame
// whose whole window lies in front of J0 reaches q = 127 - e_r <= 315 samples back, and frames up to r = 78 still straddle J0.  (Rounds 2-5 kept 128
// entries and 32 frames: right for calls of whole 192-sample blocks, the bench's and most tests'; otherwise the first -e_0 / 4 frames summed LDS
// beyond A[] -- whatever the CU's previous workgroup had left there: 1e-4 on five frames of a channel, differently from channel to channel -- and
// frames 32 ... 78 went without their part.  Found by test_call_made_in_overlapping_pieces_against_the_oracle's twins.)
__global__ __launch_bounds__(GAIN_FIX_BACK) void gain_fix_kernel(DeviceTables T, DeviceBuffers B, CallGeom G, int channels) {
    __shared__ float2 A[GAIN_FIX_BACK];
    const int ch = blockIdx.x, t = threadIdx.x;
    const ChanParams &P = B.params[ch];
    ChanState *st = &B.state[ch];
    const float gl = P.volume * P.left_ch, gr = P.volume * P.right_ch;
    const bool valid = st->gain_valid != 0;
    const float dl = valid ? st->prev_gl - gl : 0.f, dr = valid ? st->prev_gr - gr : 0.f;
    float2 *out = B.gfix + (size_t)ch * GAIN_FIX_FRAMES;
    __syncthreads();                                              // (every thread has read the previous gains)
    if (t == 0) { st->prev_gl = gl; st->prev_gr = gr; st->gain_valid = 1; }
    if (dl == 0.f && dr == 0.f) { if (t < GAIN_FIX_FRAMES) out[t] = make_float2(0.f, 0.f); return; }
    const AudioSet AS = T.audio_sets[P.audio_set];
    const float2 *__restrict__ dring = B.dring + (size_t)ch * (G.dring_mask + 1);
    {
        const int q = t;                                          // A[q] = a[J0 - q]; q = 0 unused
        float2 acc = make_float2(0.f, 0.f);
        if (q >= 1) {
            const int64_t j = G.J0 - q - AS.delay;
            if (AS.delay > 0) {                                   // audio low-pass on: a[j'] = sum_i h_a[i] d[j' - delay - i]
                const float *__restrict__ ha = T.audio_lp_taps + (size_t)P.audio_set * AUDIO_TAPS;
                for (int i = 0; i < AUDIO_TAPS; i++) {
                    const int64_t idx = j - i;
                    const float2 v = idx >= 0 ? dring[idx & G.dring_mask] : make_float2(0.f, 0.f);
                    acc.x = fmaf(ha[i], v.x, acc.x); acc.y = fmaf(ha[i], v.y, acc.y);
                }
            } else acc = j >= 0 ? dring[j & G.dring_mask] : make_float2(0.f, 0.f);
        }
        A[q] = acc;
    }
    __syncthreads();
    if (t < GAIN_FIX_FRAMES) {
        const int e = (int)(4 * (G.M0 + t) + 3 - G.J0);           // newest window entry of frame M0 + t, relative to J0
        float2 c = make_float2(0.f, 0.f);
        for (int q = (e < -1 ? -e : 1); q < GAIN_FIX_BACK && e + q < RS_TAPS; q++) {
            const float h = T.rs_taps[e + q];
            c.x = fmaf(h, A[q].x, c.x); c.y = fmaf(h, A[q].y, c.y);
        }
        out[t] = make_float2(dl * c.x, dr * c.y);
    }
}

void launch_gain_fix(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, int channels, hipStream_t s) {
    hipLaunchKernelGGL(gain_fix_kernel, dim3(channels), dim3(GAIN_FIX_BACK), 0, s, T, B, G, channels);
}

void launch_audio(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, float2 *pcm,
                  int channels, hipStream_t s) {
    const int64_t frames = G.M1 - G.M0;
    if (frames <= 0) return;
    if (B.peaks_on) hipLaunchKernelGGL(audio_fft_kernel<true>, dim3((unsigned)((frames + AF_VALID - 1) / AF_VALID), channels), dim3(fftc::T), 0, s, T, B, G, pcm);
    else hipLaunchKernelGGL(audio_fft_kernel<false>, dim3((unsigned)((frames + AF_VALID - 1) / AF_VALID), channels), dim3(fftc::T), 0, s, T, B, G, pcm);
    hipLaunchKernelGGL(pcm_tail_kernel, dim3((channels + 63) / 64), dim3(64), 0, s, B, G, channels);
}

// ---- second converter (sendSampletoOutput fm-processor.cpp:825-838 with audioRate != workingRate; design::design_conv2): one
// thread per output frame, out[m] = sum_k taps[(m q) mod p][k] x[floor (m q / p) - k] in the oracle's order (k ascending,
// unfused).  x = the 48 kHz frames of this call behind the nt frames in front of them (x48[ch][nt + i], i relative to the call).
__global__ __launch_bounds__(256) void conv2_kernel(const float2 *__restrict__ x48, int64_t x_stride, const float *__restrict__ taps,
                                                    int p, int q, int nt, int64_t in0, int64_t out0, int64_t nout,
                                                    float2 *__restrict__ pcm, int64_t pcm_stride) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nout) return;
    const int ch = blockIdx.y;
    const int64_t mq = (out0 + i) * q, n0 = mq / p;
    const int ph = (int)(mq - n0 * p);
    const float *t = taps + (size_t)ph * nt;
    const float2 *x = x48 + (size_t)ch * x_stride + nt + (n0 - in0);       // x[-k] = frame n0 - k
    float ar = 0.f, ai = 0.f;
    for (int k = 0; k < nt; k++) {
        if (n0 - k < 0) break;                                             // in front of the stream: nothing
        const float2 v = x[-k];
        ar = __fadd_rn(ar, __fmul_rn(t[k], v.x)); ai = __fadd_rn(ai, __fmul_rn(t[k], v.y));
    }
    pcm[(size_t)ch * pcm_stride + i] = make_float2(ar, ai);
}
// the last nt frames of the call become the history in front of the next one (one block per channel)
__global__ __launch_bounds__(256) void conv2_shift_kernel(float2 *x48, int64_t x_stride, int nt, int64_t frames) {
    float2 *x = x48 + (size_t)blockIdx.x * x_stride;
    const int t = threadIdx.x;
    float2 v = make_float2(0.f, 0.f);
    if (t < nt) v = x[frames + t];
    __syncthreads();
    if (t < nt) x[t] = v;
}
void launch_conv2(float2 *x48, int64_t x_stride, const float *taps, int p, int q, int nt, int64_t in0, int64_t frames_in,
                  int64_t out0, int64_t nout, float2 *pcm, int64_t pcm_stride, int channels, hipStream_t s) {
    if (nout > 0)
        hipLaunchKernelGGL(conv2_kernel, dim3((unsigned)((nout + 255) / 256), channels), dim3(256), 0, s, x48, x_stride, taps, p, q, nt,
                           in0, out0, nout, pcm, pcm_stride);
    if (frames_in > 0) hipLaunchKernelGGL(conv2_shift_kernel, dim3(channels), dim3(256), 0, s, x48, x_stride, nt, frames_in);
}

}  // namespace fmx
