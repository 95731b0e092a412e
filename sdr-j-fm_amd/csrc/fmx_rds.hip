// fmx_rds.hip -- the RDS path, run after stage B when a channel has RDS on.
//
// Replaces per channel:
//   rdsBandPassFilter.Pass(float)   fm-processor.cpp:741, fft-filters.cpp:97-130  (32768-pt overlap-add, x3, real part)
//   rdsHilbertFilter.Pass(float)    fm-processor.cpp:742-743, fft-filters.cpp:165-201 (spectrum mask 1,2..2,1,0..0)
//   57 kHz mix with 3 x the pilot phase of 64000 samples ago   fm-processor.cpp:744-754
//   rdsDecimator (11 taps, /8)      fm-processor.cpp:382,553, fir-filters.cpp:397-424
//   rdsDecoder_2::doDecode          rds-decoder-2.cpp:83-157 (RRC matched filter, AGC agc.h, Mueller&Mueller
//                                   timing, Costas costas.h, differential decode)
//
// The two overlap-add filters are NOT equivalent to short FIRs: the Hilbert mask is applied to the
// zero-padded 32768-point block CIRCULARLY, so each output depends on the whole 32000-sample block.  They are
// therefore reproduced as what they are -- block FFT filters with the reference's block phase, latency
// (32000 samples each) and overlap buffers -- on a hand-written four-step FFT (32768 = 128 x 256) in LDS.
// Everything downstream of them is cheap: the mix + decimator is time-parallel (each 24 kS/s output recomputes its
// 11 inputs), the matched filter is time-parallel, and only AGC / M&M / Costas run as a lane-per-channel recurrence.
#include "fmx_internal.h"
#include "fmx_fftconv.h"
#include <algorithm>
#include <vector>

namespace fmx {

constexpr int RN = 32768, RN1 = 128, RN2 = 256;      // FFT size and its four-step factorisation
constexpr int RBLK = RDS_BLK;                         // NumofSamples = fftSize - degree (fft-filters.cpp:34)
constexpr int RDEG = 768;

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// ---- 16-point DFT in registers (two 8-point DFTs of fmx_fftconv.h + the 16th roots of unity), natural order in and out
__device__ __forceinline__ void dft16_fwd(float2 *v) {
    float2 e[8], o[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
    fftc::dft8<-1>(e); fftc::dft8<-1>(o);
    // exp (-2 pi i k / 16), k = 0 .. 7
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, r2 = 0.70710678118654752440f;
    const float2 w[8] = { make_float2(1.f, 0.f), make_float2(c1, -s1), make_float2(r2, -r2), make_float2(s1, -c1),
                          make_float2(0.f, -1.f), make_float2(-s1, -c1), make_float2(-r2, -r2), make_float2(-c1, -s1) };
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float2 t = k == 0 ? o[0] : fftc::cmul(o[k], w[k]);
        v[k] = fftc::cadd(e[k], t); v[k + 8] = fftc::csub(e[k], t);
    }
}

// ---- four-step FFT, step 1: 256 column FFTs of length 128 (+ twiddle), 32 columns per workgroup (256-byte row segments in HBM).
//      in  : x[n], n = 256*n1 + n2          out : a[k1*256 + n2] = W_N^(n2 k1) * sum_n1 x[256 n1 + n2] W_128^(n1 k1)
// Eight threads per column, sixteen points per thread, 128 = 16 x 8: n1 = 8 p + t, k1 = q + 16 u,
//      X[q + 16 u] = sum_t W_8^(t u) W_128^(t q) sum_p x[8 p + t] W_16^(p q)
// -- a 16-point DFT in registers, one exchange through LDS, an 8-point DFT in registers.  (Rounds 1-2: radix 2 in LDS, seven passes with
// a barrier each: LDS bound at 2.4-2.9 TB/s of HBM traffic.)
// pair_src != null: the row is not read from `in` but built on the fly from two real 32000-sample blocks, channels 2 p and 2 p + 1 as
// real and imaginary part, zero padded (rds_load_real_pair fused into the transform's first pass)
constexpr int S1C = 32;
__global__ __launch_bounds__(256) void rds_fft_step1(const float2 *__restrict__ in, float2 *__restrict__ out, int nch,
                                                     const int *__restrict__ chlist, const float *__restrict__ pair_src = nullptr,
                                                     size_t pair_stride = 0, int C = 0) {
    __shared__ __attribute__((aligned(16))) float2 sx[S1C][RN1 + 2];     // [column][q * 8 + t]; the pad keeps the 16-byte reads of adjacent columns apart
    __shared__ float2 twA[RN1], twB[RN2];                  // W_128^a, W_32768^b
    const int tid = threadIdx.x;
    const int ch = chlist ? chlist[blockIdx.y] : blockIdx.y;
    const int c0 = blockIdx.x * S1C;
    const float2 *x = in + (size_t)ch * RN;
    float2 *a = out + (size_t)ch * RN;
    if (tid < RN1) { float sn, cs; sincospif(-2.0f * (float)tid / (float)RN1, &sn, &cs); twA[tid] = make_float2(cs, sn); }
    { float sn, cs; sincospif(-2.0f * (float)tid / (float)RN, &sn, &cs); twB[tid] = make_float2(cs, sn); }
    const int col = tid & (S1C - 1), t = tid >> 5, n2 = c0 + col;
    float2 v[16];
    if (pair_src) {
        const float *pa = pair_src + (size_t)(2 * ch) * pair_stride, *pb = pair_src + (size_t)(2 * ch + 1) * pair_stride;
        const bool hb = 2 * ch + 1 < C;
#pragma unroll
        for (int p = 0; p < 16; p++) { const int n = (8 * p + t) * RN2 + n2; v[p] = make_float2(n < RBLK ? pa[n] : 0.f, (n < RBLK && hb) ? pb[n] : 0.f); }
    } else {
#pragma unroll
        for (int p = 0; p < 16; p++) v[p] = x[(8 * p + t) * RN2 + n2];
    }
    __syncthreads();                                       // (the twiddle tables)
    dft16_fwd(v);
#pragma unroll
    for (int q = 0; q < 16; q++) sx[col][q * 8 + t] = q == 0 ? v[0] : fftc::cmul(v[q], twA[t * q]);
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int q = t + 8 * h;
        float2 w[8];
        const float4 *r4 = reinterpret_cast<const float4 *>(&sx[col][q * 8]);
#pragma unroll
        for (int i = 0; i < 4; i++) { const float4 f = r4[i]; w[2 * i] = make_float2(f.x, f.y); w[2 * i + 1] = make_float2(f.z, f.w); }
        fftc::dft8<-1>(w);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k1 = q + 16 * u;
            const int m = (k1 * n2) & (RN - 1);            // twiddle W_N^(n2 k1): the exponent split as 256 a + b
            a[k1 * RN2 + n2] = fftc::cmul(w[u], fftc::cmul(twA[m >> 8], twB[m & 255]));
        }
    }
}
// ---- step 2: 128 row FFTs of length 256, 16 rows per workgroup; X[k1 + 128 k2] = sum_n2 a[k1][n2] W_256^(n2 k2).
// Sixteen threads per row, sixteen points per thread, 256 = 16 x 16: n2 = 16 p + t, k2 = q + 16 u, two 16-point DFTs in registers around
// one exchange through LDS (the second one with the threads regrouped so that adjacent lanes hold adjacent k1: 128-byte segments out).
// EPI: what happens to the transform's result on its way out (the element-wise passes of the block filters fused into the
// transform's last pass): 0 stored as it is; 1 times the filter vector S (and `scale`), conjugated (rds_spectrum); 2 band-pass
// pair behind the second transform -- conj / N, overlap add, the two real block results, the two new tails (rds_finish_pair +
// rds_save_tail_pair); 3 one channel's complex result likewise (rds_finish + rds_save_tail).  The tails of block b go to the
// overlap buffer of parity b & 1 and are read from the other one: no pass reads what another one is writing.
struct RdsEpi {
    const float2 *S; float scale;                 // 1
    const float2 *over_old; float2 *over_new;     // 2, 3: [ch][RDEG]
    float *out_real; float2 *out_cplx; size_t out_stride; int C;
};
template <int EPI>
__global__ __launch_bounds__(256) void rds_fft_step2(const float2 *__restrict__ in, float2 *__restrict__ out, int nch,
                                                     const int *__restrict__ chlist, RdsEpi E) {
    constexpr int QS = 18, RS = 16 * QS + 2;               // [row][q * 18 + t]: strides that keep both sides' LDS accesses conflict-free
    __shared__ __attribute__((aligned(16))) float2 sy[16][RS];
    __shared__ float2 tw[RN2];                             // W_256^i
    const int tid = threadIdx.x;
    const int ch = chlist ? chlist[blockIdx.y] : blockIdx.y;
    const int k10 = blockIdx.x * 16;
    const float2 *a = in + (size_t)ch * RN;
    float2 *X = out + (size_t)ch * RN;
    { float sn, cs; sincospif(-2.0f * (float)tid / (float)RN2, &sn, &cs); tw[tid] = make_float2(cs, sn); }
    float2 v[16];
    {
        const int row = tid >> 4, t = tid & 15;
#pragma unroll
        for (int p = 0; p < 16; p++) v[p] = a[(k10 + row) * RN2 + 16 * p + t];
        __syncthreads();                                   // (the twiddle table)
        dft16_fwd(v);
#pragma unroll
        for (int q = 0; q < 16; q++) sy[row][q * QS + t] = q == 0 ? v[0] : fftc::cmul(v[q], tw[t * q]);
    }
    __syncthreads();
    const int r = tid & 15, q = tid >> 4;
    {
        const float4 *r4 = reinterpret_cast<const float4 *>(&sy[r][q * QS]);
#pragma unroll
        for (int i = 0; i < 8; i++) { const float4 f = r4[i]; v[2 * i] = make_float2(f.x, f.y); v[2 * i + 1] = make_float2(f.z, f.w); }
        dft16_fwd(v);
    }
#pragma unroll
    for (int u = 0; u < 16; u++) {
        const int k2 = q + 16 * u, k = (k10 + r) + RN1 * k2;
        const float2 vv = v[u];
        if (EPI == 0) X[k] = vv;
        else if (EPI == 1) { float2 w = cmulf(vv, E.S[k]); X[k] = make_float2(w.x * E.scale, -(w.y * E.scale)); }
        else if (EPI == 2) {
            const int ca = 2 * ch, cb = 2 * ch + 1;
            const bool hb = cb < E.C;
            const float f = 1.0f / (float)RN;
            float va = vv.x * f, vb = -vv.y * f;
            if (k < RBLK) {
                if (k < RDEG) { va += E.over_old[(size_t)ca * RDEG + k].x; if (hb) vb += E.over_old[(size_t)cb * RDEG + k].x; }
                E.out_real[(size_t)ca * E.out_stride + k] = va; if (hb) E.out_real[(size_t)cb * E.out_stride + k] = vb;
            } else {
                E.over_new[(size_t)ca * RDEG + (k - RBLK)] = make_float2(va, 0.f);
                if (hb) E.over_new[(size_t)cb * RDEG + (k - RBLK)] = make_float2(vb, 0.f);
            }
        } else {
            const float f = 1.0f / (float)RN;
            float2 w = make_float2(vv.x * f, -vv.y * f);
            if (k < RBLK) {
                if (k < RDEG) { const float2 o = E.over_old[(size_t)ch * RDEG + k]; w.x += o.x; w.y += o.y; }
                E.out_cplx[(size_t)ch * E.out_stride + k] = w;
            } else E.over_new[(size_t)ch * RDEG + (k - RBLK)] = w;
        }
    }
}

// ---- load a 32000-sample real block, zero padded (fft-filters.cpp:104-107)
__global__ void rds_load_real(const float *__restrict__ src, size_t src_stride, float2 *__restrict__ U, const int *chlist) {
    const int ch = chlist ? chlist[blockIdx.y] : blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < RN) U[(size_t)ch * RN + i] = make_float2(i < RBLK ? src[(size_t)ch * src_stride + i] : 0.f, 0.f);
}
// ---- FFT_C = conj(FFT_A * filterVector [* 3])   (fft-filters.cpp:111-118 / 145-150)
__global__ void rds_spectrum(float2 *__restrict__ U, const float2 *__restrict__ S, float scale, const int *chlist) {
    const int ch = chlist ? chlist[blockIdx.y] : blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < RN) {
        float2 v = cmulf(U[(size_t)ch * RN + i], S[i]);
        v.x *= scale; v.y *= scale;
        U[(size_t)ch * RN + i] = make_float2(v.x, -v.y);
    }
}
// ---- FFT_C = conj(FFT_C)/N; overlap add; keep the tail   (fft-filters.cpp:120-125 / 152-157)
//      writes the 32000 outputs either as real parts (band-pass) or complex (Hilbert)
__global__ void rds_finish(const float2 *__restrict__ U, float2 *__restrict__ over, float *__restrict__ out_real,
                           float2 *__restrict__ out_cplx, size_t out_stride, const int *chlist) {
    const int ch = chlist ? chlist[blockIdx.y] : blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= RN) return;
    const float f = 1.0f / (float)RN;
    float2 v = U[(size_t)ch * RN + i];
    v = make_float2(v.x * f, -v.y * f);
    if (i < RDEG) { const float2 o = over[(size_t)ch * RDEG + i]; v.x += o.x; v.y += o.y; }
    if (i < RBLK) {
        if (out_real) out_real[(size_t)ch * out_stride + i] = v.x;
        if (out_cplx) out_cplx[(size_t)ch * out_stride + i] = v;
    }
}
__global__ void rds_save_tail(const float2 *__restrict__ U, float2 *__restrict__ over, const int *chlist) {
    // Overloop[j] = FFT_C[NumofSamples + j] AFTER the scaling (and before the overlap is added: j >= 32000 > 768)
    const int ch = chlist ? chlist[blockIdx.y] : blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < RDEG) {
        const float f = 1.0f / (float)RN;
        const float2 v = U[(size_t)ch * RN + RBLK + j];
        over[(size_t)ch * RDEG + j] = make_float2(v.x * f, -v.y * f);
    }
}

// ---- append this call's demod samples to the block being filled and the pilot phases to the delay ring
__global__ void rds_collect(DeviceBuffers B, RdsBuffers Rb, CallGeom G, int C, int64_t row0, int nrows) {
    const int ch = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int64_t nc0 = Rb.nc0[ch];
    if (q >= nrows || nc0 < 0) return;              // (a channel whose decoder is off: its filters, its phase delay line and its decimator stand still, as the reference's do)
    const int64_t r = row0 + q, n = nc0 + r;        // the channel's own sample count
    const float demod = B.w_dem[tap_idx(B, r, ch, G.pitch)];
    const float cur = B.w_cur[tap_idx(B, r, ch, G.pitch)];   // unconstrained pilot phase; PI_Constrain gives currentPilotPhase
    // Channels of one block phase ride through the block transforms in pairs, as the real and imaginary part of one row: a non-finite sample of
    // one (the raw IQ formats cannot carry one, float32 input can) would turn the whole pair's spectrum into NaN and leave the
    // neighbour's slicer state NaN for good.  It enters the block as zero instead.
    Rb.in_blk[(size_t)ch * RBLK + (int)(n % RBLK)] = (fabsf(demod) < __builtin_inff()) ? demod : 0.f;
    float c = (fabsf(cur) < __builtin_inff()) ? cur : 0.f;
    {   // PI_Constrain fm-constants.h:148-158 (arguments are within (-2pi, 4pi))
        const double v = (double)c;
        if (!(c >= 0.f && c < 6.2831855f)) c = (v >= 6.283185307179586) ? (float)(v - 6.283185307179586) : (float)(v + 6.283185307179586);
    }
    Rb.phase_ring[(size_t)ch * RDS_PHASE_RING + (int)(n & (RDS_PHASE_RING - 1))] = c;
}

// ---- 57 kHz mix + decimate by 8: one thread per 24 kS/s output m (rds sample index 8m+7).  A workgroup's 256 outputs read 2058 consecutive inputs:
//      they are mixed ONCE each, from coalesced loads, into LDS (round 5; before, every output gathered and mixed its own eleven inputs -- 64-byte
//      strides through the Hilbert block and 1.4 oscillator evaluations per input: the RDS path's largest kernel), and the decimator's eleven taps
//      read them from there.  The same operations on the same values in the same order.
constexpr int MD_OUT = 256, MD_IN = 8 * MD_OUT + 10;
__device__ __forceinline__ int md_pos(int j) { return j + (j >> 3); }         // one entry of padding per eight: the outputs' reads, 64 bytes apart, spread over the banks
__global__ __launch_bounds__(MD_OUT) void rds_mix_decim(DeviceBuffers B, RdsBuffers Rb, CallGeom G, int C, int64_t row0, int nrows) {
    __shared__ float2 sx[MD_IN + MD_IN / 8 + 2];
    const int ch = blockIdx.y;
    const int tid = threadIdx.x;
    const int64_t nc0 = Rb.nc0[ch];
    if (nc0 < 0) return;
    // the channel's outputs whose newest input 8 m + 7 lies in rows [row0, row0 + nrows) of the call (its own count: nc0 + row)
    const int64_t m0 = (nc0 + row0) / 8;
    const int nout = (int)((nc0 + row0 + nrows) / 8 - m0);
    if ((int64_t)blockIdx.x * MD_OUT >= nout) return;
    const int64_t mb = m0 + (int64_t)blockIdx.x * MD_OUT;    // the workgroup's first output
    const int64_t n_lo = 8 * mb + 7 - 10;                    // ... and the oldest input it reads (negative in front of the stream's start)
    // block and position of that input (one 64-bit division per thread; the others follow by counting up, across at most one block boundary)
    const int64_t nb = n_lo >= 0 ? n_lo : 0;
    const int64_t blk_b = nb / RBLK; const int inp_b = (int)(nb - blk_b * RBLK);
    const float2 *hil = Rb.hil + (size_t)ch * 2 * RBLK;
    const float *pring = Rb.phase_ring + (size_t)ch * RDS_PHASE_RING;
    const int nin = (int)((nout - (int64_t)blockIdx.x * MD_OUT < MD_OUT ? nout - (int64_t)blockIdx.x * MD_OUT : MD_OUT) * 8 + 3);       // inputs 0 .. 8 (outputs - 1) + 10
    for (int j = tid; j < nin; j += MD_OUT) {
        const int64_t n = n_lo + j;
        float2 x = make_float2(0.f, 0.f);
        if (n >= 0) {
            int inp = inp_b + (int)(n - nb); int64_t blk = blk_b;
            if (inp >= RBLK) { inp -= RBLK; blk += 1; }
            // during block blk the Hilbert filter returns the result of block blk-1 (zeros for the first block)
            float2 hv = make_float2(0.f, 0.f);
            if (blk >= 1) hv = hil[(size_t)((blk - 1) & 1) * RBLK + inp];
            float th = 0.f;
            if (n >= 2 * RBLK) th = pring[(int)((n - 2 * RBLK) & (RDS_PHASE_RING - 1))];
            th = 3 * th;                                     // thePhase = 3 * rdsPhaseBuffer[idx]  (:744)
            // cos / sin (thePhase) (:749-750) from the hardware's sine unit: thePhase lies in [0, 6 pi), the unit takes turns and is good to
            // ~5e-7 absolute -- on a sub-carrier of 0.07 that is 4e-8, forty times below what the block filters' own rounding leaves
            const float turns = th * 0.15915494309189535f;
            const float2 osc = make_float2(__builtin_amdgcn_cosf(turns), -__builtin_amdgcn_sinf(turns));
            x = cmulf(osc, hv);                              // *rdsValueCmpl = oscValue * rdsBaseHilb  (:754)
        }
        sx[md_pos(j)] = x;
    }
    __syncthreads();
    const int q = blockIdx.x * MD_OUT + tid;
    if (q >= nout) return;
    const int64_t m = m0 + q;
    float2 acc = make_float2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 11; i++) {                           // newest -> oldest, kernel[0] * newest (fir-filters.cpp:409-418)
        const float2 x = sx[md_pos(8 * tid + 10 - i)];
        const float2 k = Rb.dec_taps[i];
        const float2 p = cmulf(x, k);
        acc.x += p.x; acc.y += p.y;
    }
    Rb.rds24[(size_t)ch * RDS24_RING + (int)(m & (RDS24_RING - 1))] = acc;
}

// ---- RRC matched filter (rds-decoder-2.cpp:83-98), time-parallel; output channel-major: mfc[ch][2 + q] (entries 0, 1 are the last
//      two AGC outputs of the previous call, put there by rds_agc)
__global__ void rds_matched(DeviceBuffers B, RdsBuffers Rb, int nj) {
    const int ch = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int64_t nc0 = Rb.nc0[ch];
    if (nc0 < 0) return;
    const int64_t m0 = nc0 / 8; const int nout = (int)((nc0 + nj) / 8 - m0);          // the channel's own 24 kS/s outputs of this call
    if (q >= nout || B.params[ch].rds_mode != 2) return;
    const int64_t m = m0 + q;
    float2 acc = make_float2(0.f, 0.f);
    for (int i = 0; i < 45; i++) {
        const int64_t j = m - i;
        if (j < 0) continue;
        const float2 v = Rb.rds24[(size_t)ch * RDS24_RING + (int)(j & (RDS24_RING - 1))];
        const float w = Rb.rrc[i];
        acc.x += v.x * w; acc.y += v.y * w;
    }
    Rb.mfc[(size_t)ch * Rb.mfc_stride + 2 + q] = acc;
    // |x| for the AGC (std::abs (complex) = hypotf: f64 square root of the f64 sum of squares), taken here where it is time-parallel
    Rb.mfm[(size_t)ch * Rb.mfc_stride + 2 + q] = (float)sqrt((double)acc.x * (double)acc.x + (double)acc.y * (double)acc.y);
}

// The reference's rdsDecoder_2::doDecode runs AGC, Mueller & Mueller timing, Costas loop and slicer per 24 kS/s sample
// (rds-decoder-2.cpp:101-157).  Only the AGC works on every sample; the rest acts once per symbol (every ~20.2 samples), on the last
// three AGC outputs, and nothing of it feeds back into the AGC.  One lane per channel walking the samples with the symbol branch
// inside -- round 1 / 2 -- made a wave pay for the branch at nearly every sample (64 lanes at unrelated symbol phases: some lane
// is in it 96 % of the time; 0.87 ms per call at 2048 channels on 32 waves).  Two kernels instead:
//   rds_agc      the AGC (agc.h:14-18) of every sample, in place in mfc: a workgroup per channel, the gain by a scan (see there);
//   rds_symbols  lane per channel over the SYMBOLS: the sample a symbol falls on follows from the skip count in closed form
//                (`++sampleCount >= skipNrSamples`), so a lane jumps from symbol to symbol and every lane is in the branch together.
constexpr int AGC_T = 256, AGC_K = 10;                         // threads per channel, samples per thread (2400 outputs per 0.1 s call)
__global__ __launch_bounds__(AGC_T) void rds_agc(DeviceBuffers B, RdsBuffers Rb, int C, int nj) {
    // AGC (2e-3, 0.38, start 9), agc.h:14-18: out = in * gain; gain += rate * (ref - |out|).  |out| = |in * gain| is taken as gain * |in|
    // with |in| from rds_matched (the same value to an ulp), which makes the gain an AFFINE recurrence, gain' = gain (1 - rate |in|) +
    // rate ref: one workgroup per channel, every thread composes the maps of its AGC_K adjacent samples in f64, a scan over the
    // workgroup gives the gain in front of each thread, and the thread then runs its own samples in the reference's f32 expression.
    // (Lane per channel over the samples -- the first form of this kernel -- had 64 lanes walking 64 rows in 64 different pages:
    // 0.16-0.2 ms per call at 2048 channels, all of it address translation and load latency.)
    const int ch = blockIdx.x;
    if (ch >= C || B.params[ch].rds_mode != 2) return;
    const int64_t nc0 = Rb.nc0[ch];
    if (nc0 < 0) return;
    const int64_t m0 = nc0 / 8; const int nout = (int)((nc0 + nj) / 8 - m0);          // the channel's own 24 kS/s outputs of this call
    (void)m0;
    __shared__ double sA[4], sB[4];
    __shared__ float2 sLast[3];
    __shared__ float sGain;                                        // the gain behind a round of AGC_T * AGC_K samples
    RdsState *sp = Rb.state + ch;
    const float g0 = sp->gain;
    float2 *row = Rb.mfc + (size_t)ch * Rb.mfc_stride;
    const float *mrow = Rb.mfm + (size_t)ch * Rb.mfc_stride + 2;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float2 p0 = sp->sb0, p1 = sp->sb1, p2 = sp->sb2;
    __syncthreads();                                               // (everybody has read the state the last thread rewrites below)
    if (tid == 0) { row[0] = p1; row[1] = p2; }                   // what the first symbols of this call may look back at
    for (int base = 0; base < nout; base += AGC_T * AGC_K) {       // (one round for calls of up to 2560 outputs)
        const int q0 = base + tid * AGC_K;
        float a[AGC_K]; float2 x[AGC_K];
        double A = 1.0, Bv = 0.0;
#pragma unroll
        for (int i = 0; i < AGC_K; i++) {
            const bool ok = q0 + i < nout;
            a[i] = ok ? mrow[q0 + i] : 0.f; x[i] = ok ? row[2 + q0 + i] : make_float2(0.f, 0.f);
            if (ok) { const double m = 1.0 - (double)2e-3f * (double)a[i]; A *= m; Bv = Bv * m + (double)2e-3f * (double)0.38f; }
        }
        // inclusive scan of the maps g -> A g + B over the wave, then over the four waves
        double cA = A, cB = Bv;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double pA = __shfl_up(cA, o, 64), pB = __shfl_up(cB, o, 64);
            if (lane >= o) { cB = cA * pB + cB; cA = cA * pA; }
        }
        if (lane == 63) { sA[wv] = cA; sB[wv] = cB; }
        __syncthreads();
        double gin = (double)((base == 0) ? g0 : sGain);
        for (int v = 0; v < wv; v++) gin = sA[v] * gin + sB[v];      // gain behind the waves in front
        {
            const double eA = __shfl_up(cA, 1, 64), eB = __shfl_up(cB, 1, 64);
            if (lane > 0) gin = eA * gin + eB;                          // ... and behind the lanes in front
        }
        float gain = (float)gin;
#pragma unroll
        for (int i = 0; i < AGC_K; i++) {
            if (q0 + i < nout) {
                const float2 v = make_float2(x[i].x * gain, x[i].y * gain);
                gain += 2e-3f * (0.38f - gain * a[i]);
                row[2 + q0 + i] = v;
                if (q0 + i >= nout - 3) sLast[q0 + i - (nout - 3)] = v;
            }
        }
        __syncthreads();                                           // (everybody has taken the last round's gain)
        if (q0 < nout && q0 + AGC_K >= nout) sp->gain = gain;       // the thread that owns the call's last sample
        if (tid == AGC_T - 1) sGain = gain;                         // (a further round starts from here)
        __syncthreads();
    }
    // sampleBuffer [0..2] behind the call = the last three AGC outputs (rds-decoder-2.cpp:123-125)
    if (tid == 0) {
        float2 n0 = p0, n1 = p1, n2 = p2;
        if (nout >= 3) { n0 = sLast[0]; n1 = sLast[1]; n2 = sLast[2]; }
        else for (int q = 0; q < nout; q++) { n0 = n1; n1 = n2; n2 = row[2 + q]; }
        sp->sb0 = n0; sp->sb1 = n1; sp->sb2 = n2;
    }
}

// ---- Mueller & Mueller timing -> Costas -> slicer -> differential decode, one step per symbol   [lane per channel]
//      rds-decoder-2.cpp:120-157, costas.h:21-33
__global__ __launch_bounds__(64) void rds_symbols(DeviceBuffers B, RdsBuffers Rb, int C, int nj) {
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= C || B.params[ch].rds_mode != 2) return;
    const int64_t nc0 = Rb.nc0[ch];
    if (nc0 < 0) return;
    const int64_t m0 = nc0 / 8; const int nout = (int)((nc0 + nj) / 8 - m0);          // the channel's own 24 kS/s outputs of this call
    (void)m0;
    RdsState st = Rb.state[ch];
    const float sps = 24000.0f / 1187.5f;                   // samplesPerSymbol = rate / (float)RDS_BITCLK_HZ
    const float mm_alpha = (float)0.01;
    uint8_t *bits = Rb.bits + (size_t)ch * RDS_BITS_CAP;
    const float2 *v = Rb.mfc + (size_t)ch * Rb.mfc_stride + 2;     // v[q], q = -2 .. nout - 1: AGC outputs
    // `if (++sampleCount >= skip)` at sample q of the call: sampleCount = count0 + q + 1
    int q = st.skip - st.sample_count - 1;
    q = q < 0 ? 0 : q;
    int count = st.sample_count + nout;                     // (no symbol in this call)
    while (q < nout) {
        const float2 s0 = v[q - 2], s1 = v[q - 1], s2 = v[q];
        const float2 r0 = make_float2(s0.x > 0.f ? 1.f : -1.f, s0.y > 0.f ? 1.f : -1.f);
        const float2 r1 = make_float2(s1.x > 0.f ? 1.f : -1.f, s1.y > 0.f ? 1.f : -1.f);
        const float2 r2 = make_float2(s2.x > 0.f ? 1.f : -1.f, s2.y > 0.f ? 1.f : -1.f);
        const float x = (r2.x - r0.x) * s1.x + (r2.y - r0.y) * s1.y;
        const float y = (s2.x - s0.x) * r1.x + (s2.y - s0.y) * r1.y;
        const float mm = y - x;
        st.mu += sps + mm_alpha * mm;
        st.skip = (int)st.mu;
        st.mu -= (float)st.skip;
        // Costas (alpha 1, beta 0.02, limit 2*pi*10/24000)
        const float2 e = make_float2(cosf(-st.c_phase), sinf(-st.c_phase));       // std::exp(complex(0, -phase))
        const float2 r = cmulf(s2, e);
        const float err = r.x * r.y;
        st.c_freq += 0.02f * err;
        if (fabsf(st.c_freq) > st.c_limit) st.c_freq = 0.f;
        st.c_phase += st.c_freq + 1.0f * err;
        {   // PI_Constrain, generic form
            const double pv = (double)st.c_phase;
            if (!(0.0 <= pv && pv < 6.283185307179586)) {
                if (pv >= 6.283185307179586) st.c_phase = (float)fmod(pv, 6.283185307179586);
                else if (pv > -6.283185307179586) st.c_phase = (float)(pv + 6.283185307179586);
                else st.c_phase = (float)(6.283185307179586 - fmod(-pv, 6.283185307179586));
            }
        }
        const int bit = r.x >= 0.f ? 1 : 0;
        bits[st.nbits & (RDS_BITS_CAP - 1)] = (uint8_t)(bit ^ st.prev_bit);
        Rb.sym[(size_t)ch * RDS_SYM_CAP + (st.nbits & (RDS_SYM_CAP - 1))] = r;      // *m = r: what the IQ scope shows (fm-processor.cpp:555-563)
        st.prev_bit = bit;
        st.nbits++;
        count = nout - 1 - q;                               // sampleCount behind the call, should this be its last symbol
        q += st.skip < 1 ? 1 : st.skip;                     // the next sample with sampleCount >= skip (a skip below 1 fires at once)
    }
    st.sample_count = count;
    RdsState *sp = Rb.state + ch;                           // (gain and the sample buffer are rds_agc's)
    sp->mu = st.mu; sp->c_freq = st.c_freq; sp->c_phase = st.c_phase; sp->sample_count = st.sample_count; sp->skip = st.skip;
    sp->prev_bit = st.prev_bit; sp->nbits = st.nbits;
}


// =================================================================================================
// RDS_1 (setfmRdsSelector 1): rdsDecoder::doDecode rds-decoder.cpp:76-84 -> rdsDecoder_1::doDecode rds-decoder-1.cpp:124-142
//   Costas (1/16, 0.02/16, +-10 Hz)  [lane per channel]  ->  rdsFilter (21-tap low-pass) -> Match (43 taps)  [time-parallel,
//   each output summed in the reference's tap order]  ->  sharpFilter (order-7 Butterworth band-pass on v^2, eight biquads),
//   slope detector, differential decode  [lane per channel]
// =================================================================================================
__device__ __forceinline__ float pi_constrain_generic(float ph) {                 // PI_Constrain fm-constants.h:149-158
    const double pv = (double)ph;
    if (0.0 <= pv && pv < 6.283185307179586) return ph;
    if (pv >= 6.283185307179586) return (float)fmod(pv, 6.283185307179586);
    if (pv > -6.283185307179586) return (float)(pv + 6.283185307179586);
    return (float)(6.283185307179586 - fmod(-pv, 6.283185307179586));
}
__global__ __launch_bounds__(64) void rds1_costas(DeviceBuffers B, RdsBuffers Rb, int C, int nj) {
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= C || B.params[ch].rds_mode != 1) return;
    const int64_t nc0 = Rb.nc0[ch];
    if (nc0 < 0) return;
    const int64_t m0 = nc0 / 8; const int nout = (int)((nc0 + nj) / 8 - m0);          // the channel's own 24 kS/s outputs of this call

    Rds1State *st = Rb.state1 + ch;
    float freq = st->c_freq, phase = st->c_phase;
    const float alpha = 1.0f / 16.0f, beta = 0.02f / 16.0f, lim = (float)(2 * 3.14159265358979323846 * (double)10.0f / (double)(float)24000);
    const float2 *in = Rb.rds24 + (size_t)ch * RDS24_RING;
    float *out = Rb.c_ring + (size_t)ch * RDS24_RING;
    for (int q = 0; q < nout; q++) {
        const int64_t m = m0 + q;
        const float2 z = in[(int)(m & (RDS24_RING - 1))];
        const float2 r = cmulf(z, make_float2(cosf(-phase), sinf(-phase)));       // z * std::exp(complex(0, -phase))  costas.h:22
        const float err = r.x * r.y;
        freq += beta * err;
        if (fabsf(freq) > lim) freq = 0.f;
        phase += freq + alpha * err;
        phase = pi_constrain_generic(phase);
        out[(int)(m & (RDS24_RING - 1))] = r.x;                                   // decoder_1 -> doDecode (real (v), ...)
    }
    st->c_freq = freq; st->c_phase = phase;
}
// out[m] = sum_i ring_in[m - i] * taps[i], i ascending from a zero accumulator (Basic_FIR::Pass fir-filters.h:96-108, Match :108-121)
template <int NT>
__global__ __launch_bounds__(256) void rds1_fir(DeviceBuffers B, const float *__restrict__ rin, float *__restrict__ rout,
                                                const float *__restrict__ taps, int nj, int to_mf, RdsBuffers Rb) {
    const int ch = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int64_t nc0 = Rb.nc0[ch];
    if (nc0 < 0) return;
    const int64_t m0 = nc0 / 8; const int nout = (int)((nc0 + nj) / 8 - m0);          // the channel's own 24 kS/s outputs of this call
    if (q >= nout || B.params[ch].rds_mode != 1) return;
    const int64_t m = m0 + q;
    const float *in = rin + (size_t)ch * RDS24_RING;
    float tmp = 0.f;
#pragma unroll
    for (int i = 0; i < NT; i++) {
        const int64_t j = m - i;
        const float v = j >= 0 ? in[(int)(j & (RDS24_RING - 1))] : 0.f;
        tmp += v * taps[i];
    }
    if (to_mf) Rb.mf[(size_t)q * C_RDS_PITCH(Rb) + ch] = make_float2(tmp, 0.f);   // sample-major for the lane-per-channel slicer
    else rout[(size_t)ch * RDS24_RING + (int)(m & (RDS24_RING - 1))] = tmp;
}
__global__ __launch_bounds__(64) void rds1_slicer(DeviceBuffers B, RdsBuffers Rb, int C, int nj) {
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= C || B.params[ch].rds_mode != 1) return;
    const int64_t nc0 = Rb.nc0[ch];
    if (nc0 < 0) return;
    const int64_t m0 = nc0 / 8; const int nout = (int)((nc0 + nj) / 8 - m0);          // the channel's own 24 kS/s outputs of this call
    (void)m0;
    Rds1State st = Rb.state1[ch];
    RdsState s2 = Rb.state[ch];                          // the bit ring's write counter is shared with the RDS_2 slicer
    const int pitch = C_RDS_PITCH(Rb);
    const float *cf = Rb.rds1_coef + RDS1_FIR + RDS1_MATCH;
    const float gain = cf[RDS1_QUADS * 4];
    uint8_t *bits = Rb.bits + (size_t)ch * RDS_BITS_CAP;
    for (int q = 0; q < nout; q++) {
        const float v = Rb.mf[(size_t)q * pitch + ch].x;
        float o = (v * v) * gain;                         // sharpFilter.Pass (v * v): Basic_IIR::Pass iir-filters.h:89-103
#pragma unroll
        for (int i = 0; i < RDS1_QUADS; i++) {
            const float rm1 = st.m1[i], rm2 = st.m2[i];
            const float w = o - rm1 * cf[4 * i + 2] - rm2 * cf[4 * i + 3];
            o = w + rm1 * cf[4 * i] + rm2 * cf[4 * i + 1];
            st.m2[i] = rm1; st.m1[i] = w;
        }
        const float slope = o - st.last_sync;
        st.last_sync = o;
        if ((slope < 0.0f) && (st.last_sync_slope >= 0.0f)) {                     // top of the sine wave: take the data
            const int bit = st.last_data >= 0.f ? 1 : 0;
            bits[s2.nbits & (RDS_BITS_CAP - 1)] = (uint8_t)(bit ^ st.prev_bit);
            st.prev_bit = bit;
            s2.nbits++;
        }
        st.last_data = v;
        st.last_sync_slope = slope;
    }
    Rb.state1[ch] = st;
    Rb.state[ch].nbits = s2.nbits;
}


// =================================================================================================
// RDS_3 (setfmRdsSelector 3): rdsDecoder::doDecode rds-decoder.cpp:92-100 -> rdsDecoder_3::doDecode rds-decoder-3.cpp:86-113.
// Everything is one recurrence per channel [lane per channel]: Costas, bit-clock NCO (SinCos table of 24000 entries),
// integrate-and-dump, differential decode -- and the block synchroniser (rds-blocksynchronizer.cpp:215-336 with the reaction
// of rdsDecoder::processBit rds-decoder.cpp:104-131), because its sync-error count re-synchronises the bit clock (:95-100).
// rdsFilter (21 taps) only feeds the 21-entry syncBuffer that synchronizeOnBitClk (:116-153) reads: its outputs are computed
// when a resynchronisation happens, from the ring of Costas outputs, each summed in the reference's tap order.
// =================================================================================================
__device__ __forceinline__ uint32_t bs_offset(int blk, int typeB) { return blk == 0 ? 0xFCu : blk == 1 ? 0x198u : blk == 2 ? (typeB ? 0x350u : 0x168u) : 0x1B4u; }
__device__ __forceinline__ uint32_t bs_syndrome(uint32_t bits, uint32_t off) {     // :126-142
    const uint32_t block = bits ^ off;
    uint32_t reg = 0;
    for (int k = 25; k >= 0; k--) {
        const uint32_t msb = reg & (1u << 9);
        reg <<= 1;
        if (msb) reg ^= 0x5B9u;
        if ((block >> k) & 1u) reg ^= 0x31Bu;
    }
    return reg;
}
__device__ __forceinline__ void bs_resync(Rds3State &s) { s.bs_cur = 0; s.bs_synced = 0; s.bs_bits_in_blk = 0; }
__device__ __forceinline__ void bs_push(Rds3State &s, int bit) {
    const int typeB = (s.bs_blk1 >> 11) & 1;
    s.bs_stream = (s.bs_stream << 1) | (bit ? 1u : 0u);
    if (s.bs_synced) {
        if (++s.bs_bits_in_blk < 26) return;
        s.bs_bits_in_blk = 0;
        if (bs_syndrome(s.bs_stream, bs_offset(s.bs_cur, typeB)) != 0) { bs_resync(s); return; }          // RDS_NO_CRC -> resync
        if (s.bs_cur == 1) s.bs_blk1 = (int)((s.bs_stream >> 10) & 0xFFFFu);
        if (s.bs_cur == 3) s.bs_blk1 = 0;                                                                 // complete group: cleared
        s.bs_cur = (s.bs_cur + 1) & 3;
        return;
    }
    if (s.bs_cur == 0) {
        if (bs_syndrome(s.bs_stream & 0x3FFFFFFu, bs_offset(0, typeB)) != 0) return;                      // waiting for block A
        s.bs_bits_in_blk = 0; s.bs_cur = 1;
        return;
    }
    if (s.bs_bits_in_blk < 25) { s.bs_bits_in_blk++; return; }
    s.bs_bits_in_blk = 0;
    if (bs_syndrome(s.bs_stream, bs_offset(s.bs_cur, typeB)) != 0) { s.bs_sync_err++; bs_resync(s); return; }   // RDS_NO_SYNC
    if (s.bs_cur == 1) s.bs_blk1 = (int)((s.bs_stream >> 10) & 0xFFFFu);
    if (s.bs_cur < 2) { s.bs_cur++; return; }
    s.bs_synced = 1;
    if (s.bs_cur == 3) s.bs_blk1 = 0;
    s.bs_cur = (s.bs_cur + 1) & 3;
}
__device__ __forceinline__ float sin24(const float2 *__restrict__ tab, float phase) {        // SinCos::getSin sincos.cpp:81-85, Rate 24000
    const double C = 24000 / (2 * 3.14159265358979323846);
    if (phase < 0) return -tab[((int)((double)(-phase) * C)) % 24000].y;
    return tab[((int)((double)phase * C)) % 24000].y;
}
__global__ __launch_bounds__(64) void rds3_slicer(DeviceBuffers B, RdsBuffers Rb, int C, int nj) {
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= C || B.params[ch].rds_mode != 3) return;
    const int64_t nc0 = Rb.nc0[ch];
    if (nc0 < 0) return;
    const int64_t m0 = nc0 / 8; const int nout = (int)((nc0 + nj) / 8 - m0);          // the channel's own 24 kS/s outputs of this call

    Rds3State st = Rb.state3[ch];
    RdsState s2 = Rb.state[ch];
    if (!st.started) { st.started = 1; st.resync_pending = 1; }          // Resync = true (rds-decoder-3.cpp:81)
    const float alpha = 1.0f / 16.0f, beta = 0.02f / 16.0f, lim = (float)(2 * 3.14159265358979323846 * (double)10.0f / (double)(float)24000);
    const float omegaRDS = (float)((2 * 3.14159265358979323846 * 1187.5) / (float)24000);
    const int symbolCeiling = 21, symbolFloor = 20;                      // ceil / floor (24000 / 1187.5f)
    const float2 *in = Rb.rds24 + (size_t)ch * RDS24_RING;
    float *cr = Rb.c_ring + (size_t)ch * RDS24_RING;
    const float *fir = Rb.rds1_coef;                                    // rdsFilter (21, RDS_WIDTH, rate): the same taps as rdsDecoder_1's
    uint8_t *bits = Rb.bits + (size_t)ch * RDS_BITS_CAP;
    for (int q = 0; q < nout; q++) {
        const int64_t m = m0 + q;
        const float2 z = in[(int)(m & (RDS24_RING - 1))];
        const float2 r = cmulf(z, make_float2(cosf(-st.c_phase), sinf(-st.c_phase)));
        const float err = r.x * r.y;
        st.c_freq += beta * err;
        if (fabsf(st.c_freq) > lim) st.c_freq = 0.f;
        st.c_phase += st.c_freq + alpha * err;
        st.c_phase = pi_constrain_generic(st.c_phase);
        const float v = r.x;
        cr[(int)(m & (RDS24_RING - 1))] = v;
        if (st.resync_pending || st.bs_sync_err > 3) {
            // synchronizeOnBitClk: the syncBuffer holds rdsFilter's outputs at samples m - 20 .. m, oldest first
            float corr[21];
#pragma unroll
            for (int i = 0; i < 21; i++) corr[i] = 0.f;
            bool isHigh = false; int k = 0;
            for (int i = 0; i < symbolCeiling; i++) {
                const float phase = (float)fmod((double)((float)i * (omegaRDS / 2)), 6.283185307179586);
                const float sn = sin24(Rb.sincos24, phase);
                if (sn > 0 && !isHigh) { isHigh = true; k = 0; }
                else if (sn < 0 && isHigh) { isHigh = false; k = 0; }
                const int64_t t = m - 20 + i;
                float f = 0.f;
                for (int j = 0; j < RDS1_FIR; j++) {
                    const int64_t u = t - j;
                    const float cv = u >= 0 ? cr[(int)(u & (RDS24_RING - 1))] : 0.f;
                    f += cv * fir[j];
                }
                // corr[k++] += f with a run-time k: select instead of a dynamically indexed register array
#pragma unroll
                for (int kk = 0; kk < 21; kk++) if (kk == k) corr[kk] += f;
                k++;
            }
            int iMin = 0;
            auto cget = [&](int idx) { float o = 0.f;
#pragma unroll
                for (int kk = 0; kk < 21; kk++) if (kk == idx) o = corr[kk];
                return o; };
            while (iMin < symbolFloor && cget(iMin++) > 0) {}
            while (iMin < symbolFloor && cget(iMin++) < 0) {}
            st.bit_clk_phase = (float)fmod((double)(-omegaRDS * (float)(iMin - 1)), 6.283185307179586);
            while (st.bit_clk_phase < 0) st.bit_clk_phase = (float)((double)st.bit_clk_phase + 6.283185307179586);
            bs_resync(st); st.bs_sync_err = 0; st.resync_pending = 0;
        }
        const float clk = sin24(Rb.sincos24, st.bit_clk_phase);
        st.bit_integrator += clk * v;
        if (st.prev_clk_state <= 0 && clk > 0) {                          // rising edge: look at the integrator
            const int bit = st.bit_integrator >= 0 ? 1 : 0;
            const int d = bit ^ st.prev_bit;
            st.bit_integrator = 0.f;
            st.prev_bit = bit;
            bits[s2.nbits & (RDS_BITS_CAP - 1)] = (uint8_t)d;
            s2.nbits++;
            bs_push(st, d);                                               // rdsDecoder::doDecode: processBit (theBit)
        }
        st.prev_clk_state = clk;
        st.bit_clk_phase = (float)fmod((double)(st.bit_clk_phase + omegaRDS), 6.283185307179586);
    }
    Rb.state3[ch] = st;
    Rb.state[ch].nbits = s2.nbits;
}

static void fft_fwd(const RdsBuffers &Rb, int nch, const int *chlist, hipStream_t s) {
    hipLaunchKernelGGL(rds_fft_step1, dim3(RN2 / S1C, nch), dim3(256), 0, s, Rb.U, Rb.V, nch, chlist, (const float *)nullptr, (size_t)0, 0);
    hipLaunchKernelGGL(rds_fft_step2<0>, dim3(RN1 / 16, nch), dim3(256), 0, s, Rb.V, Rb.U, nch, chlist, RdsEpi{});
}
// the transform of `nrows` rows: first pass A -> Bs (or from the real pair source), last pass Bs -> A with the epilogue
template <int EPI>
static void fft_rows(float2 *A, float2 *Bs, int nrows, const RdsEpi &E, hipStream_t s, const float *pair_src = nullptr, size_t pair_stride = 0, int C = 0) {
    hipLaunchKernelGGL(rds_fft_step1, dim3(RN2 / S1C, nrows), dim3(256), 0, s, A, Bs, nrows, (const int *)nullptr, pair_src, pair_stride, C);
    hipLaunchKernelGGL(rds_fft_step2<EPI>, dim3(RN1 / 16, nrows), dim3(256), 0, s, Bs, A, nrows, (const int *)nullptr, E);
}

// ---- Two channels per transform.  The inputs of both block filters are REAL, so channels 2 p and 2 p + 1 ride as the real and
// imaginary part of one complex row: the band-pass -- complex taps s, but only Re (a * s) = a * Re (s) is kept, and with the real
// kernel Re (s), (a + j b) * Re (s) = a * Re (s) + j b * Re (s) -- keeps the pair through the backward transform as well and the two
// results are the row's real and imaginary parts; the Hilbert filter's output is complex,
// so its spectra are taken apart behind the forward transform (A[k] = (Z[k] + conj Z[N - k]) / 2, B[k] = (Z[k] - conj Z[N - k]) / 2j)
// and the backward transforms run per channel.  2.5 instead of 4 transforms per channel and block: the four-step transform is
// bound by its HBM passes -- for the same reason the element-wise passes (load, filter vector, finish, tails) are fused into the
// transforms' first / last passes (rds_fft_step1's pair source, rds_fft_step2's epilogues).
// Hilbert: the pair's spectrum Z apart, each times the filter vector, conjugated (rds_spectrum's step) into the channels' own rows
__global__ void rds_hil_split(const float2 *__restrict__ Z, const float2 *__restrict__ S, float2 *__restrict__ U, int C) {
    const int p = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= RN) return;
    const int a = 2 * p, b = 2 * p + 1;
    const float2 zk = Z[(size_t)p * RN + k], zn = Z[(size_t)p * RN + ((RN - k) & (RN - 1))];
    const float2 xa = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
    const float2 xb = make_float2(0.5f * (zk.y + zn.y), 0.5f * (zn.x - zk.x));
    const float2 sk = S[k];
    float2 v = cmulf(xa, sk);
    U[(size_t)a * RN + k] = make_float2(v.x, -v.y);
    if (b < C) { v = cmulf(xb, sk); U[(size_t)b * RN + k] = make_float2(v.x, -v.y); }
}

// one block boundary of the channels h_list[0 .. nlist) (one block phase and parity): BP filter of the demod block just completed (block index blk
// of their own count), Hilbert of the previous BP result.  Every channel of the handle at once: pairs of channels per transform; a subset (channels
// that joined at different times): one channel per transform, by list.  Both forms keep their overlap tails by block parity.
void launch_rds_block(const RdsBuffers &Rb, int C, int64_t blk, const int *h_list, int nlist, hipStream_t s) {
    const bool pair = env_switches().rds_pair != 0;
    const size_t ov = (size_t)C * RDEG;                           // one parity of an overlap buffer
    if (pair && C >= 2 && nlist == C) {
        const int P = (C + 1) / 2;
        const dim3 gp(RN / 256, P);
        // Hilbert of the previous band-pass result: pairs forward straight from the real blocks (rows of V, U as scratch), apart into U,
        // backward per channel with the finishing pass in the transform
        fft_rows<0>(Rb.V, Rb.U, P, RdsEpi{}, s, Rb.bpreal + (size_t)((blk + 1) & 1) * RBLK, (size_t)2 * RBLK, C);
        hipLaunchKernelGGL(rds_hil_split, gp, dim3(256), 0, s, Rb.V, Rb.S_hil, Rb.U, C);
        {
            RdsEpi E{}; E.over_old = Rb.hil_over + (size_t)((blk + 1) & 1) * ov; E.over_new = Rb.hil_over + (size_t)(blk & 1) * ov;
            E.out_cplx = Rb.hil + (size_t)(blk & 1) * RBLK; E.out_stride = (size_t)2 * RBLK; E.C = C;
            fft_rows<3>(Rb.U, Rb.V, C, E, s);
        }
        // band-pass of the demod block just completed: pairs all the way, the filter vector and the finishing pass in the transforms
        {
            RdsEpi E{}; E.S = Rb.S_bp_re; E.scale = 3.0f;
            fft_rows<1>(Rb.U, Rb.V, P, E, s, Rb.in_blk, (size_t)RBLK, C);
        }
        {
            RdsEpi E{}; E.over_old = Rb.bp_over + (size_t)((blk + 1) & 1) * ov; E.over_new = Rb.bp_over + (size_t)(blk & 1) * ov;
            E.out_real = Rb.bpreal + (size_t)(blk & 1) * RBLK; E.out_stride = (size_t)2 * RBLK; E.C = C;
            fft_rows<2>(Rb.U, Rb.V, P, E, s);
        }
        return;
    }
    const int *chl = nullptr;
    if (nlist != C) {
        // (pageable source: the copy is staged before the call returns)
        note_hip(hipMemcpyAsync(Rb.chlist, h_list, sizeof(int) * (size_t)nlist, hipMemcpyHostToDevice, s));
        chl = Rb.chlist;
    }
    const dim3 g(RN / 256, nlist);
    // Hilbert first: its input is bpreal[(blk-1)&1] (zeros when blk == 0), output hil[blk & 1]
    hipLaunchKernelGGL(rds_load_real, g, dim3(256), 0, s, Rb.bpreal + (size_t)((blk + 1) & 1) * RBLK, (size_t)2 * RBLK, Rb.U, chl);
    fft_fwd(Rb, nlist, chl, s);
    hipLaunchKernelGGL(rds_spectrum, g, dim3(256), 0, s, Rb.U, Rb.S_hil, 1.0f, chl);
    fft_fwd(Rb, nlist, chl, s);
    hipLaunchKernelGGL(rds_finish, g, dim3(256), 0, s, Rb.U, Rb.hil_over + (size_t)((blk + 1) & 1) * ov, (float *)nullptr, Rb.hil + (size_t)(blk & 1) * RBLK, (size_t)2 * RBLK, chl);
    hipLaunchKernelGGL(rds_save_tail, dim3(3, nlist), dim3(256), 0, s, Rb.U, Rb.hil_over + (size_t)(blk & 1) * ov, chl);
    // band-pass: input in_blk, output bpreal[blk & 1]
    hipLaunchKernelGGL(rds_load_real, g, dim3(256), 0, s, Rb.in_blk, (size_t)RBLK, Rb.U, chl);
    fft_fwd(Rb, nlist, chl, s);
    hipLaunchKernelGGL(rds_spectrum, g, dim3(256), 0, s, Rb.U, Rb.S_bp, 3.0f, chl);
    fft_fwd(Rb, nlist, chl, s);
    hipLaunchKernelGGL(rds_finish, g, dim3(256), 0, s, Rb.U, Rb.bp_over + (size_t)((blk + 1) & 1) * ov, Rb.bpreal + (size_t)(blk & 1) * RBLK, (float2 *)nullptr, (size_t)2 * RBLK, chl);
    hipLaunchKernelGGL(rds_save_tail, dim3(3, nlist), dim3(256), 0, s, Rb.U, Rb.bp_over + (size_t)(blk & 1) * ov, chl);
}

// the RDS work of one call: rows [0, nj) of the work arrays; channel c's path has processed h_nc0[c] samples before (< 0: its decoder is off)
void launch_rds(const DeviceBuffers &B, const RdsBuffers &Rb, const CallGeom &G, int C, const int64_t *h_nc0, int modes, hipStream_t s) {
    const int64_t nj = G.J1 - G.J0;
    if (nj <= 0) return;
    // the block boundaries that fall into this call (a call holds at most one per channel: fmx_api.hip makes longer ones in pieces): the
    // channels of one boundary row and block parity are one launch of the block filters
    struct Cls { int64_t e; int64_t blk; std::vector<int> ch; };
    std::vector<Cls> cls;
    for (int c = 0; c < C; c++) {
        if (h_nc0[c] < 0) continue;
        const int64_t e = RBLK - (h_nc0[c] % RBLK);              // rows until the channel's block is complete (1 .. RBLK)
        if (e > nj) continue;
        const int64_t blk = (h_nc0[c] + e) / RBLK - 1;
        size_t k = 0;
        for (; k < cls.size(); k++) if (cls[k].e == e && ((cls[k].blk ^ blk) & 1) == 0) break;
        if (k == cls.size()) cls.push_back(Cls{e, blk, {}});
        cls[k].ch.push_back(c);
    }
    std::sort(cls.begin(), cls.end(), [](const Cls &a, const Cls &b) { return a.e < b.e; });
    int64_t row = 0;
    size_t k = 0;
    while (row < nj) {
        const int64_t end = k < cls.size() ? cls[k].e : nj;
        const int64_t take = end - row;
        if (take > 0) {
            hipLaunchKernelGGL(rds_collect, dim3((unsigned)((take + 255) / 256), C), dim3(256), 0, s, B, Rb, G, C, row, (int)take);
            // the 24 kS/s outputs whose newest input 8m+7 lies in this stretch: mixed BEFORE the block transform below
            // replaces the Hilbert result of two blocks ago, which the 10-sample look-back may still need
            hipLaunchKernelGGL(rds_mix_decim, dim3((unsigned)((take / 8 + 1 + MD_OUT - 1) / MD_OUT), C), dim3(MD_OUT), 0, s, B, Rb, G, C, row, (int)take);
        }
        row = end;
        for (; k < cls.size() && cls[k].e == end; k++) launch_rds_block(Rb, C, cls[k].blk, cls[k].ch.data(), (int)cls[k].ch.size(), s);
    }
    const int nmax = (int)(nj / 8 + 1);            // most outputs any channel has in this call
    if (modes & (1 << 2)) {
        hipLaunchKernelGGL(rds_matched, dim3((unsigned)((nmax + 255) / 256), C), dim3(256), 0, s, B, Rb, (int)nj);
        hipLaunchKernelGGL(rds_agc, dim3((unsigned)C), dim3(AGC_T), 0, s, B, Rb, C, (int)nj);
        hipLaunchKernelGGL(rds_symbols, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, s, B, Rb, C, (int)nj);
    }
    if (modes & (1 << 1)) {          // (the mf rows of an RDS_1 channel are its own: the two slicers never share a channel)
        const dim3 gt((unsigned)((nmax + 255) / 256), C);
        hipLaunchKernelGGL(rds1_costas, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, s, B, Rb, C, (int)nj);
        hipLaunchKernelGGL(rds1_fir<RDS1_FIR>, gt, dim3(256), 0, s, B, Rb.c_ring, Rb.f_ring, Rb.rds1_coef, (int)nj, 0, Rb);
        hipLaunchKernelGGL(rds1_fir<RDS1_MATCH>, gt, dim3(256), 0, s, B, Rb.f_ring, (float *)nullptr, Rb.rds1_coef + RDS1_FIR, (int)nj, 1, Rb);
        hipLaunchKernelGGL(rds1_slicer, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, s, B, Rb, C, (int)nj);
    }
    if (modes & (1 << 3)) hipLaunchKernelGGL(rds3_slicer, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, s, B, Rb, C, (int)nj);
}

}  // namespace fmx
