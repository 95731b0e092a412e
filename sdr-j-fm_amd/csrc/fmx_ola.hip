// fmx_ola.hip -- the reference's two overlap-add filters AS THE BLOCK MACHINES THEY ARE, for handles of few channels.
// COMPILED WITH -ffp-contract=off.
//
// fftFilter (fft-filters.cpp:33-163) is not an LTI filter once its settings move: Pass () stores the sample at FFT_A [inp], returns
// FFT_C [inp] and runs the block transform when inp reaches NumofSamples = fftSize - degree; setLowPass (:84-95) computes a new kernel and
// sets inp = 0 WITHOUT touching FFT_A, FFT_C or Overloop.  A setBandwidth / setlfcutoff in the middle of a stream (radio.cpp:1706-1712 ->
// fm-processor.cpp:232-239,396-408,762-770) therefore makes the reference play the last completed output block again from its start,
// drop the block in progress, and add the old block's tail to the first block of the new kernel; switched "Off", the filter keeps its
// buffers until it is switched on again.  The folded polyphase FIRs of stage A / stage C (fmx_front.hip, fmx_audio.hip) reproduce the
// filter where it is LTI -- every setting made before the first call -- and not this.  Large batches keep the folded form (the backlog
// the glitch is a function of would double stage A's traffic).  Handles of up to 64 channels -- the drop-in receiver -- run what the
// reference runs (fmx_handle_s::ola_mode, FMX_P_FILTER_RESTARTS):
//   pre_kernel        RF DC removal, IQ balance, LO mix per sample (fm-processor.cpp:423-446,462-466), as the reference orders them
//   ola_io_kernel     Pass () for a run of samples that ends at or before the block boundary: block buffer in, last block's result out
//   ola_conv_kernel   the block transform as a direct convolution: C [k] = sum_i h [i] A [k - i] + Overloop [k], new Overloop = the tail
// and the decimators (stage A's kernel with the 37-tap set of "input filter off") / the resampler (stage C with its 128-tap set) read
// the filtered stream.  The block position of every (channel, filter) is kept on the host, which knows every sample count and reset.
// The direct convolution differs from the reference's f32 FFTs by rounding only (1e-7 relative, as the folded form does).
#include "fmx_internal.h"

namespace fmx {

// ---------------------------------------------------------------------------------------------------------------------
// pre_kernel: one workgroup per channel.  RfDC = (x - RfDC) * alpha + RfDC is an affine recurrence: every thread runs its eight
// consecutive samples from zero, a scan over the workgroup gives the state in front of each thread's run (to 1e-7 of it: the
// cross-thread carry is summed in another order), then the thread runs the reference's own f32 expression from there.
// ---------------------------------------------------------------------------------------------------------------------
// (one workgroup walks a channel's call tile by tile, and every tile ends in stores that the next tile's loads queue up behind: tiles as
// large as a workgroup can hold -- 1024 threads x 16 samples = the reference's block of 16384 in one tile -- keep those round trips few)
constexpr int PRE_T = 1024, PRE_K = 8, PRE_TILE = PRE_T * PRE_K;
// Global memory is read and written in thread order (sample base + kk * PRE_T + t: coalesced); the recurrences want every thread on eight
// CONSECUTIVE samples: the tile goes through LDS, sample j of the tile at pad (j) = j + j / 8 (nine-entry rows: both sides conflict-free).
__device__ __forceinline__ int pre_pad(int j) { return j + (j >> 3); }
constexpr int PRE_LDS = PRE_TILE + PRE_TILE / 8;
static_assert(PRE_TILE == PRE_TILE_SAMPLES, "the host sizes the look-back buffers by this");

// The maps r -> r (1 - u) + a of the workgroup's threads composed in thread order: the map of everything in front of this thread
// (eu, er, ei) and of the whole workgroup (tu, tr, ti).  Six shuffle steps inside the wave, the four wave totals through LDS.
struct AffW { float u, r, i; };
__device__ __forceinline__ AffW aff_after(const AffW &p, const AffW &c) {      // p first, then c
    AffW o; o.u = p.u + c.u - p.u * c.u; o.r = p.r + c.r - p.r * c.u; o.i = p.i + c.i - p.i * c.u; return o;
}
__device__ __forceinline__ void wg_affine_scan(AffW m, AffW *excl, AffW *total, AffW *sW /* [4] LDS */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    AffW inc = m;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        AffW o; o.u = __shfl_up(inc.u, d, 64); o.r = __shfl_up(inc.r, d, 64); o.i = __shfl_up(inc.i, d, 64);
        if (lane >= d) inc = aff_after(o, inc);
    }
    AffW ex; ex.u = __shfl_up(inc.u, 1, 64); ex.r = __shfl_up(inc.r, 1, 64); ex.i = __shfl_up(inc.i, 1, 64);
    if (lane == 0) { ex.u = 0.f; ex.r = 0.f; ex.i = 0.f; }
    if (lane == 63) sW[wv] = inc;
    __syncthreads();
    AffW pre; pre.u = 0.f; pre.r = 0.f; pre.i = 0.f;
    AffW tot = pre;
#pragma unroll
    for (int v = 0; v < PRE_T / 64; v++) { const AffW w = sW[v]; if (v < wv) pre = aff_after(pre, w); tot = aff_after(tot, w); }
    *excl = aff_after(pre, ex); *total = tot;
    __syncthreads();
}

template <int FMT>
__global__ __launch_bounds__(PRE_T) void pre_kernel(DeviceTables T, DeviceBuffers B, CallGeom G, const void *__restrict__ iq_raw, float2 *__restrict__ vbuf, int64_t vstride,
                                                       int fused, OlaStep S, const OlaChan *__restrict__ Stab, OlaBuffers O, PreLook LB) {
    // One workgroup per TILE (8192 samples) and channel.  The only thing a tile needs from the tiles in front of it is the RF DC state at its
    // first sample: every workgroup publishes its tile's map r -> r (1 - u) + a (it depends on the tile's samples only) and walks the maps of
    // the tiles in front of its own from the call's state, one after the other as a single workgroup walking the call would -- the same
    // values, without the call's tiles queueing up behind each other's round trips to memory (one channel, 0.1 s: 29 tiles, 0.28 -> 0.02 ms).
    const int ntiles = gridDim.x;
    const int ch = blockIdx.y, t = threadIdx.x;
    // the tile of this workgroup: the next one of its channel nobody has taken yet (NOT blockIdx.x: the look-back below spins on the tiles in front
    // of its own, which is only safe when their workgroups are known to have started -- an order of dispatch HIP does not promise)
    __shared__ int s_tile;
    if (t == 0) s_tile = ntiles > 1 ? (int)(atomicAdd(&LB.tickets[ch], 1u) - LB.ticket_base) : 0;
    __syncthreads();
    const int tile = s_tile;
    const ChanParams P = B.params[ch];
    ChanState *st = B.state + ch;
    constexpr int BPS = (FMT == 0) ? 8 : (FMT == 3 ? 4 : 2);
    const char *inb = reinterpret_cast<const char *>(iq_raw) + (size_t)P.stream * G.stream_stride * BPS;
    float2 *out = vbuf + (size_t)ch * vstride;
    const int n = (int)G.n, R = G.input_rate;
    const float qs = G.iq_scale;
    const float alpha = 1.0f / (float)R;                       // rfDcAlpha fm-processor.cpp:379
    const bool dcr = P.dc_remove != 0;
    const bool rst = (P.actions & ACT_DC_RESET) != 0;          // setDCRemove zeroes RfDC (:922-925)
    float dr = rst ? 0.f : st->dc_re, di = rst ? 0.f : st->dc_im;
    const int lo = P.lo_freq, lo_phase0 = st->lo_phase;
    const bool have_tab = T.lo_table != nullptr;
    __shared__ AffW sW[PRE_T / 64];
    auto load = [&](int i) -> float2 {
        if (FMT == 0) return reinterpret_cast<const float2 *>(inb)[i];
        if (FMT == 1) { const uint8_t *p = reinterpret_cast<const uint8_t *>(inb) + 2 * (size_t)i; return make_float2((float)((int)p[0] - 127) * qs, (float)((int)p[1] - 127) * qs); }
        if (FMT == 2) { const int8_t *p = reinterpret_cast<const int8_t *>(inb) + 2 * (size_t)i; return make_float2((float)p[0] * qs, (float)p[1] * qs); }
        const int16_t *p = reinterpret_cast<const int16_t *>(inb) + 2 * (size_t)i;
        return make_float2((float)p[0] * qs, (float)p[1] * qs);
    };
    // (where this kernel also does the input filter's Pass (), the block results that go out in the samples' place are requested with them)
    const OlaChan od = fused ? (Stab ? Stab[ch] : S.ch[ch]) : OlaChan{};     // (handles above OLA_MAX_CH channels: the step's table in device memory)
    const bool pass = fused && od.on;
    const float2 *Cc = pass ? O.C + (size_t)ch * O.L + od.inp : nullptr;
    float2 *Ab = pass ? O.A + (size_t)ch * O.L + od.inp : nullptr;
    __shared__ float2 sx[PRE_LDS];
    __shared__ float sCarry[2];
    float4 *const lb_map = LB.maps + (size_t)ch * LB.max_tiles;
    int *const lb_flag = LB.flags + (size_t)ch * LB.max_tiles;
    {
        const int base = tile * PRE_TILE;
        const int i0 = base + t * PRE_K;
        float2 x[PRE_K], cb[PRE_K];
#pragma unroll
        for (int kk = 0; kk < PRE_K; kk++) {
            const int j = kk * PRE_T + t, i = base + j;
            sx[pre_pad(j)] = (i < n) ? load(i) : make_float2(0.f, 0.f);
            cb[kk] = (pass && i < n) ? Cc[i] : make_float2(0.f, 0.f);       // (what goes out in sample i's place: needed at the store, requested now)
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PRE_K; k++) x[k] = sx[pre_pad(t * PRE_K + k)];
        if (dcr) {
            // this thread's run as the map r -> r (1 - u) + a
            float u = 0.f, ar = 0.f, ai = 0.f;
#pragma unroll
            for (int k = 0; k < PRE_K; k++) if (i0 + k < n) { u = (1.0f - u) * alpha + u; ar = (x[k].x - ar) * alpha + ar; ai = (x[k].y - ai) * alpha + ai; }
            AffW mine; mine.u = u; mine.r = ar; mine.i = ai;
            AffW ex, tot;
            wg_affine_scan(mine, &ex, &tot, sW);
            const float tu = tot.u, tr = tot.r, ti = tot.i;
            if (ntiles > 1) {
                if (t == 0) {
                    lb_map[tile] = make_float4(tu, tr, ti, 0.f);
                    __hip_atomic_store(&lb_flag[tile], LB.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (t < 64) {                                   // the state in front of this tile: the maps of tiles 0 .. tile - 1, in order
                    float cr = dr, ci = di;
                    for (int b = 0; b < tile; b += 64) {
                        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (b + t < tile) {
                            while (__hip_atomic_load(&lb_flag[b + t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != LB.epoch) __builtin_amdgcn_s_sleep(2);
                            m = lb_map[b + t];
                        }
                        const int cnt = tile - b < 64 ? tile - b : 64;
                        for (int k = 0; k < cnt; k++) {
                            const float mu = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m.x), k));
                            const float mr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m.y), k));
                            const float mi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m.z), k));
                            cr = cr - cr * mu + mr; ci = ci - ci * mu + mi;
                        }
                    }
                    if (t == 0) { sCarry[0] = cr; sCarry[1] = ci; }
                }
                __syncthreads();
                dr = sCarry[0]; di = sCarry[1];
            }
            float r0 = dr - dr * ex.u + ex.r, q0 = di - di * ex.u + ex.i;
#pragma unroll
            for (int k = 0; k < PRE_K; k++) if (i0 + k < n) {
                r0 = (x[k].x - r0) * alpha + r0; q0 = (x[k].y - q0) * alpha + q0;              // :425
                x[k].x -= __builtin_amdgcn_fmed3f(r0, -0.01f, 0.01f); x[k].y -= __builtin_amdgcn_fmed3f(q0, -0.01f, 0.01f);   // DCRlimit :429-442
            }
            dr = dr - dr * tu + tr; di = di - di * tu + ti;       // the state behind the tile (every thread the same)
        } else if (ntiles > 1) {
            // (no RF DC removal: the flags still say "this workgroup has read the channel's state", which the last tile waits for below)
            if (t == 0) __hip_atomic_store(&lb_flag[tile], LB.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        // IQ balance and LO mix (:462-466; Oscillator::nextValue oscillator.cpp:49-58: LOPhase after sample i of the call = (P0 - (i + 1) lo) mod R)
        long long ph = 0;
        if (have_tab) {
            ph = ((long long)lo_phase0 - (((long long)(i0 + 1) * (long long)lo) % (long long)R)) % (long long)R;
            if (ph < 0) ph += R;
        }
#pragma unroll
        for (int k = 0; k < PRE_K; k++) if (i0 + k < n) {
            float2 v = make_float2(x[k].x * P.att_l, x[k].y * P.att_r);
            if (have_tab) {
                const float2 w = T.lo_table[ph];
                v = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
                ph -= lo; if (ph < 0) ph += R; else if (ph >= R) ph -= R;
            }
            sx[pre_pad(t * PRE_K + k)] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < PRE_K; kk++) {
            const int j = kk * PRE_T + t, i = base + j;
            if (i < n) {
                const float2 v = sx[pre_pad(j)];
                if (fused) {
                    // (the whole call is one run of this channel's input filter -- the usual case: ola_io_kernel's work done here, into O.dst)
                    if (pass) Ab[i] = v;
                    O.dst[(size_t)ch * O.dst_stride + i] = pass ? cb[kk] : v;
                } else out[i] = v;
            }
        }
    }
    // the channel's state behind the call: the last tile's workgroup, once every other one has read the state in front of it
    if (tile != ntiles - 1) return;
    if (ntiles > 1 && !dcr && t < 64) {
        for (int b = 0; b < tile; b += 64)
            if (b + t < tile) while (__hip_atomic_load(&lb_flag[b + t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != LB.epoch) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    if (t == 0) {
        if (dcr || rst) { st->dc_re = dr; st->dc_im = di; }
        if (lo != 0) {
            long long m = ((long long)n * (long long)lo) % (long long)R;
            int p2 = (int)(((long long)lo_phase0 - m) % (long long)R);
            if (p2 < 0) p2 += R;
            st->lo_phase = p2;
        }
    }
}

void launch_pre(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, const void *iq, float2 *vbuf, int64_t vstride, int channels, hipStream_t s,
                const OlaStepRef *S, const OlaBuffers *O, const PreLook &LB) {
    const int fused = S != nullptr;
    const OlaStep S0 = (S && !S->tab) ? S->val : OlaStep{}; const OlaBuffers O0 = O ? *O : OlaBuffers{};
    const OlaChan *Stab = S ? S->tab : nullptr;
    const dim3 grid((unsigned)((G.n + PRE_TILE - 1) / PRE_TILE), channels);
    switch (G.iq_format) {
    case 1: hipLaunchKernelGGL(pre_kernel<1>, grid, dim3(PRE_T), 0, s, T, B, G, iq, vbuf, vstride, fused, S0, Stab, O0, LB); break;
    case 2: hipLaunchKernelGGL(pre_kernel<2>, grid, dim3(PRE_T), 0, s, T, B, G, iq, vbuf, vstride, fused, S0, Stab, O0, LB); break;
    case 3: hipLaunchKernelGGL(pre_kernel<3>, grid, dim3(PRE_T), 0, s, T, B, G, iq, vbuf, vstride, fused, S0, Stab, O0, LB); break;
    default: hipLaunchKernelGGL(pre_kernel<0>, grid, dim3(PRE_T), 0, s, T, B, G, iq, vbuf, vstride, fused, S0, Stab, O0, LB); break;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// ola_io_kernel: Pass () over a run of `len` samples of every channel that does not cross the block boundary (the host cuts the
// call there).  grid = (chunks, channels).  A filter that is switched off passes its input through and keeps its buffers.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ola_io_kernel(OlaStep S, const OlaChan *__restrict__ Stab, OlaBuffers O) {
    const int c = blockIdx.y;
    const OlaChan d = Stab ? Stab[c] : S.ch[c];
    const int64_t smask = O.src_mask, dmask = O.dst_mask;
    const float2 *src = O.src + (size_t)c * O.src_stride;
    float2 *dst = O.dst + (size_t)c * O.dst_stride;
    float2 *A = O.A + (size_t)c * O.L, *Cc = O.C + (size_t)c * O.L;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < d.len; i += gridDim.x * 256) {
        const float2 v = src[(O.src_pos + d.off + i) & smask];
        float2 r = v;
        if (d.on) { r = Cc[d.inp + i]; A[d.inp + i] = v; }
        dst[(O.dst_pos + d.off + i) & dmask] = r;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// ola_conv_kernel: the block transform of the channels whose block is complete (fft-filters.cpp:139-158 as a direct convolution):
//   C [k] = sum_i h [i] A [k - i]  (A = 0 outside the block)  + Overloop [k] (k < degree);   Overloop [k] = sum_{i > k} h [i] A [L + k - i]
// grid = (ceil (L / 256) + ceil (degree / 256), channels): the last workgroups of a channel compute the new tail into a scratch row that
// ola_tail_kernel moves into place (the old Overloop is an input of the first workgroups).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ola_conv_kernel(OlaStep S, const OlaChan *__restrict__ Stab, OlaBuffers O) {
    const int c = blockIdx.y;
    if (!(Stab ? Stab[c].conv : S.ch[c].conv)) return;
    const int L = O.L, D = O.degree;
    const float2 *A = O.A + (size_t)c * L;
    float2 *Cc = O.C + (size_t)c * L;
    const float *h = O.taps + (size_t)c * OLA_MAX_TAPS;
    __shared__ float sh[OLA_MAX_TAPS];
    __shared__ float2 sa[256 + OLA_MAX_TAPS];
    for (int i = threadIdx.x; i < D; i += 256) sh[i] = h[i];
    const int body = (L + 255) / 256;
    const bool tail = (int)blockIdx.x >= body;
    const int k0 = tail ? L + 256 * ((int)blockIdx.x - body) : (int)blockIdx.x * 256;      // first output of this workgroup (tail: virtual outputs L .. L + D - 1)
    // window A [k0 - (D - 1) .. k0 + 255]
    for (int i = threadIdx.x; i < 256 + D - 1; i += 256) {
        const int idx = k0 - (D - 1) + i;
        sa[i] = (idx >= 0 && idx < L) ? A[idx] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    const int k = k0 + threadIdx.x;
    if (k < L + D) {
        float ar = 0.f, ai = 0.f;
        for (int i = 0; i < D; i++) { const float2 v = sa[threadIdx.x + (D - 1) - i]; ar += sh[i] * v.x; ai += sh[i] * v.y; }
        if (k < L) {
            if (k < D) { const float2 ov = O.over[(size_t)c * OLA_MAX_TAPS + k]; ar += ov.x; ai += ov.y; }
            Cc[k] = make_float2(ar, ai);
        } else O.over_new[(size_t)c * OLA_MAX_TAPS + (k - L)] = make_float2(ar, ai);
    }
}
__global__ __launch_bounds__(256) void ola_tail_kernel(OlaStep S, const OlaChan *__restrict__ Stab, OlaBuffers O) {
    const int c = blockIdx.x;
    if (!(Stab ? Stab[c].conv : S.ch[c].conv)) return;
    for (int i = threadIdx.x; i < O.degree; i += 256) O.over[(size_t)c * OLA_MAX_TAPS + i] = O.over_new[(size_t)c * OLA_MAX_TAPS + i];
}

// ---------------------------------------------------------------------------------------------------------------------
// deemph_kernel: audio = last = (audio - last) * deemphAlpha + last (fm-processor.cpp:594-595) behind the audio filter, one workgroup per
// channel over the call's fm samples; the same scan as pre_kernel's (the carry across threads is summed in another order: 1e-7 of it)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PRE_T) void deemph_kernel(DeviceBuffers B, CallGeom G, float2 *__restrict__ ring, int fused, OlaStep S, const OlaChan *__restrict__ Stab, OlaBuffers O) {
    const int ch = blockIdx.x, t = threadIdx.x;
    ChanState *st = B.state + ch;
    const float a = B.params[ch].deemph_alpha;
    float2 *r = ring + (size_t)ch * (G.dring_mask + 1);
    const int n = (int)(G.J1 - G.J0);
    float yl = st->de_l, yr = st->de_r;
    __shared__ AffW sW[PRE_T / 64];
    __shared__ float2 sx[PRE_LDS];
    const OlaChan od = fused ? (Stab ? Stab[ch] : S.ch[ch]) : OlaChan{};
    const bool pass = fused && od.on;
    for (int base = 0; base < n; base += PRE_TILE) {
        const int i0 = base + t * PRE_K;
#pragma unroll
        for (int kk = 0; kk < PRE_K; kk++) {
            const int j = kk * PRE_T + t, i = base + j;
            float2 v = make_float2(0.f, 0.f);
            if (i < n) {
                if (fused) {
                    // (the whole call is one run of this channel's audio filter: ola_io_kernel's work done here, straight from the source ring)
                    const float2 sv = O.src[(size_t)ch * O.src_stride + ((O.src_pos + i) & O.src_mask)];
                    v = sv;
                    if (pass) { v = O.C[(size_t)ch * O.L + od.inp + i]; O.A[(size_t)ch * O.L + od.inp + i] = sv; }
                } else v = r[(G.J0 + i) & G.dring_mask];
            }
            sx[pre_pad(j)] = v;
        }
        __syncthreads();
        float2 x[PRE_K];
#pragma unroll
        for (int k = 0; k < PRE_K; k++) x[k] = sx[pre_pad(t * PRE_K + k)];
        float u = 0.f, ar = 0.f, ai = 0.f;
#pragma unroll
        for (int k = 0; k < PRE_K; k++) if (i0 + k < n) { u = (1.0f - u) * a + u; ar = (x[k].x - ar) * a + ar; ai = (x[k].y - ai) * a + ai; }
        AffW mine; mine.u = u; mine.r = ar; mine.i = ai;
        AffW ex, tot;
        wg_affine_scan(mine, &ex, &tot, sW);
        float l0 = yl - yl * ex.u + ex.r, r0 = yr - yr * ex.u + ex.i;
#pragma unroll
        for (int k = 0; k < PRE_K; k++) if (i0 + k < n) {
            l0 = (x[k].x - l0) * a + l0; r0 = (x[k].y - r0) * a + r0;
            sx[pre_pad(t * PRE_K + k)] = make_float2(l0, r0);
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < PRE_K; kk++) {
            const int j = kk * PRE_T + t, i = base + j;
            if (i < n) r[(G.J0 + i) & G.dring_mask] = sx[pre_pad(j)];
        }
        yl = yl - yl * tot.u + tot.r; yr = yr - yr * tot.u + tot.i;
        __syncthreads();
    }
    // (the state behind the call as the LAST sample's own thread computed it would be exact; the tile total is within 1e-7 of it and is what
    // the next call's carry would see anyway)
    if (t == 0 && n > 0) { st->de_l = yl; st->de_r = yr; }
}
void launch_deemph(const DeviceBuffers &B, const CallGeom &G, float2 *ring, int channels, hipStream_t s, const OlaStepRef *S, const OlaBuffers *O) {
    if (G.J1 <= G.J0) return;
    const OlaStep S0 = (S && !S->tab) ? S->val : OlaStep{}; const OlaBuffers O0 = O ? *O : OlaBuffers{};
    hipLaunchKernelGGL(deemph_kernel, dim3(channels), dim3(PRE_T), 0, s, B, G, ring, S != nullptr ? 1 : 0, S0, S ? S->tab : nullptr, O0);
}

void launch_ola_io(const OlaStepRef &S, const OlaBuffers &O, int channels, int maxlen, hipStream_t s) {
    if (maxlen <= 0) return;
    const int chunks = maxlen > 256 * 64 ? 64 : (maxlen + 255) / 256;
    hipLaunchKernelGGL(ola_io_kernel, dim3((unsigned)chunks, (unsigned)channels), dim3(256), 0, s, S.tab ? OlaStep{} : S.val, S.tab, O);
}
void launch_ola_conv(const OlaStepRef &S, const OlaBuffers &O, int channels, hipStream_t s) {
    static_assert(OLA_MAX_TAPS <= 1024, "the tail fits the workgroups below");
    const int body = (O.L + 255) / 256, tails = (O.degree + 255) / 256;
    hipLaunchKernelGGL(ola_conv_kernel, dim3((unsigned)(body + tails), (unsigned)channels), dim3(256), 0, s, S.tab ? OlaStep{} : S.val, S.tab, O);
    hipLaunchKernelGGL(ola_tail_kernel, dim3((unsigned)channels), dim3(256), 0, s, S.tab ? OlaStep{} : S.val, S.tab, O);
}

}  // namespace fmx
