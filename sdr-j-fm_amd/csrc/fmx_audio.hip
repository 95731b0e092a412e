// fmx_audio.hip -- stage C: 192 kS/s stereo -> 48 kS/s PCM.
//
// Replaces per channel:
//   fmAudioFilter (8192-pt overlap-add, 756 taps)  fm-processor.cpp:76,589-591, fft-filters.cpp:132-163
//   newConverter audioDecimator (192k -> 48k)      fm-processor.cpp:380,633-634, newconverter.cpp:55-80
//   start-up fade                                  fm-processor.cpp:636-642
//
// MI355X design: the audio low-pass and the decimate-by-4 resampler are both LTI, so they are
// folded on the host into ONE polyphase FIR (756 + 128 - 1 = 883 taps) that is only evaluated at
// the 48 kHz output instants (4x fewer MACs than filtering at 192 kS/s); de-emphasis and the
// volume/balance gains commute with it and were applied upstream (fmx_demod.hip).  The
// overlap-add latency (7436 samples = 1859 PCM frames) is reproduced by reading further back in
// the per-channel d ring.  libsamplerate itself is third-party and absent: the resampler is the
// documented fmx design (oracle/fm_oracle.c fmo_resampler_taps), "parity unpinned" for that stage.
#include "fmx_internal.h"

namespace fmx {

constexpr int CW = C_TILE + (C_MAX_TAPS + 3) / 4 + 1;     // columns of the 4-phase window: 256 + 221 + 1
constexpr int CWS = CW + 1;

__global__ __launch_bounds__(256) void audio_kernel(DeviceTables T, DeviceBuffers B, CallGeom G,
                                                    float2 *__restrict__ pcm) {
    __shared__ float2 X[4][CWS];
    const int ch = blockIdx.y;
    const int t = threadIdx.x;
    const int64_t m0 = G.M0 + (int64_t)blockIdx.x * C_TILE;
    if (m0 >= G.M1) return;
    const ChanParams P = B.params[ch];
    const AudioSet AS = T.audio_sets[P.audio_set];
    const float *__restrict__ taps = T.audio_taps + (size_t)P.audio_set * C_TAPS_STRIDE;   // reversed order
    const float2 *__restrict__ dring = B.dring + (size_t)ch * (G.dring_mask + 1);
    const int NC = AS.ntaps;
    // window entry w <-> fm index fbase + w ; output t reads w = 4 t + kk, kk = NC-1-k
    const int64_t fbase = 4 * m0 + 3 - AS.delay - (NC - 1);
    const int nw = 4 * (C_TILE - 1) + NC;
    for (int w = t; w < nw; w += 256) {
        const int64_t f = fbase + w;
        float2 v = make_float2(0.f, 0.f);
        if (f >= 0) v = dring[f & G.dring_mask];
        X[w & 3][w >> 2] = v;
    }
    __syncthreads();
    const int64_t m = m0 + t;
    if (m >= G.M1) return;
    float al = 0.f, ar = 0.f;
    int kk = 0;
    for (; kk + 4 <= NC; kk += 4) {
        const int c = t + (kk >> 2);
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const float w = taps[kk + p];
            const float2 v = X[p][c];
            al = fmaf(w, v.x, al); ar = fmaf(w, v.y, ar);
        }
    }
    for (; kk < NC; kk++) {
        const float w = taps[kk];
        const float2 v = X[kk & 3][t + (kk >> 2)];
        al = fmaf(w, v.x, al); ar = fmaf(w, v.y, ar);
    }
    // audioGainCorrection fm-processor.cpp:303-306: (volumeFactor * leftChannel) * sample.  Applied here, at
    // the output of the folded FIR, so that a volume/balance change takes effect at the call boundary as in
    // the reference (it sits behind the audio low-pass there) rather than one filter latency late.
    al *= P.volume * P.left_ch; ar *= P.volume * P.right_ch;
    // start-up fade fm-processor.cpp:638-642: factor = (Max - cnt)/Max with cnt = Max - (m - F)
    const int64_t F = B.state[ch].fade_start_frame;
    const int64_t since = m - F;
    const int Max = 24000;                               // workingRate / 2
    if (since >= 0 && since < Max) {
        const float cnt = (float)(Max - (int)since);
        const float f = ((float)Max - cnt) / (float)Max;
        al *= f; ar *= f;
    }
    pcm[(size_t)ch * G.pcm_stride + (m - G.M0)] = make_float2(al, ar);
}

void launch_audio(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, float2 *pcm,
                  int channels, hipStream_t s) {
    const int64_t frames = G.M1 - G.M0;
    if (frames <= 0) return;
    const int tiles = (int)((frames + C_TILE - 1) / C_TILE);
    hipLaunchKernelGGL(audio_kernel, dim3(tiles, channels), dim3(256), 0, s, T, B, G, pcm);
}

}  // namespace fmx
