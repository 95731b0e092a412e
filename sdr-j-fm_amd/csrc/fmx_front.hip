// fmx_front.hip -- stage A, the input-FIR kernel (the roofline-graded stage).
//
// Replaces, per channel and per call, the per-input-sample part of fmProcessor::run():
//   RF DC removal            fm-processor.cpp:423-446
//   IQ balance + LO mix      fm-processor.cpp:462-466, oscillator.cpp:49-58
//   inputFilter (251 taps)   fm-processor.cpp:469-470, fft-filters.cpp:132-163 (overlap-add, delay 65285)
//   fmBand_1 (25 taps, /6)   fm-processor.cpp:472,  fir-filters.cpp:397-424
//   fmBand_2 (3 taps, /2)    fm-processor.cpp:474
//
// MI355X design (not the reference's structure): the three LTI stages are folded on the host into ONE
// real polyphase decimate-by-12 FIR (37 taps, or 287 with the input filter) evaluated once per
// 12 inputs; the overlap-add latency is reproduced as a pure delay (5 input samples folded into
// the tap alignment `off`, 5440 fm-rate samples applied by the consumer of the ring).  One
// persistent workgroup per channel streams the call's samples tile by tile: each thread owns one
// 12-sample column (= one output), so the DC-removal recurrence is a per-thread run plus one
// f64 affine block scan, and the mixed samples sit in LDS in a [12 phases][columns] layout whose
// FIR reads are conflict-free ds_read_b64 with wave-uniform taps in SGPRs.
#include "fmx_internal.h"

namespace fmx {

constexpr int HL = A_HIST_COLS - 1;            // 24 full history columns in front of a tile
constexpr int XCOLS = HL + A_TILE_COLS;        // 280
constexpr int XSTRIDE = XCOLS + 1;             // +1 float2 pad

struct Aff { double m, ar, ai; };              // r -> m*r + a   (a complex)

__device__ __forceinline__ Aff aff_then(const Aff &first, const Aff &second) {
    // apply `first`, then `second`
    Aff o;
    o.m = first.m * second.m;
    o.ar = first.ar * second.m + second.ar;
    o.ai = first.ai * second.m + second.ai;
    return o;
}
__device__ __forceinline__ double shfl_up_d(double v, int delta) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_up(lo, delta, 64); hi = __shfl_up(hi, delta, 64);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(256) void front_kernel(DeviceTables T, DeviceBuffers B, CallGeom G,
                                                    const float2 *__restrict__ iq) {
    __shared__ float2 X[DECIM][XSTRIDE];
    __shared__ double wave_tot[4][3];
    __shared__ double carry[2];

    const int ch = blockIdx.x;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const ChanParams P = B.params[ch];
    const FrontSet FS = T.front_sets[P.front_set];
    const float *__restrict__ taps = T.front_taps + (size_t)P.front_set * A_TAPS_STRIDE;
    const float2 *__restrict__ in = iq + (size_t)P.stream * G.stream_stride;
    ChanState *st = B.state + ch;
    float2 *hist = B.hist + (size_t)ch * DECIM * A_HIST_COLS;
    float2 *zring = B.zring + (size_t)ch * (G.ring_mask + 1);

    const int64_t g0 = G.g0, n = G.n, gend = g0 + n;
    const int off = FS.off, nd = FS.nd;
    const int64_t ja = (g0 - off + 11) / 12;          // first output completed by this call
    const int64_t jb = (gend - off + 11) / 12;        // one past the last
    const int64_t qa = g0 / 12;                       // column holding the first fresh sample
    const int64_t qb = (gend - 1) / 12;               // column holding the last fresh sample
    const int r0 = (int)(g0 - qa * 12);

    // ---- history -> LDS (columns qa-24 .. qa-1 at cc 0..23, partial column qa at cc 24)
    for (int i = t; i < DECIM * A_HIST_COLS; i += 256) {
        int r = i / A_HIST_COLS, c = i - r * A_HIST_COLS;
        float2 v = hist[i];
        if (c == HL && r >= r0) v = make_float2(0.f, 0.f);
        X[r][c] = v;
    }
    if (t == 0) {
        const bool rst = (P.actions & ACT_DC_RESET) != 0;        // setDCRemove zeroes RfDC (:922-925)
        carry[0] = rst ? 0.0 : (double)st->dc_re; carry[1] = rst ? 0.0 : (double)st->dc_im;
    }

    const bool dcr = P.dc_remove != 0;
    const int lo = P.lo_freq;
    const bool mix = (lo != 0) && (T.lo_table != nullptr);
    const int R = G.input_rate;
    const float alpha = 1.0f / (float)R;              // rfDcAlpha fm-processor.cpp:379
    const double beta = 1.0 - (double)alpha;
    const float Lg = P.att_l, Rg = P.att_r;
    const bool aligned16 = ((g0 & 1) == 0) && ((G.stream_stride & 1) == 0) &&
                           ((reinterpret_cast<uintptr_t>(iq) & 15) == 0);
    const int lo_phase0 = st->lo_phase;
    __syncthreads();

    for (int64_t qt = qa; qt <= qb; qt += A_TILE_COLS) {
        const int64_t q = qt + t;                     // this thread's column
        const int64_t base = q * 12;                  // global index of row 0
        const int cc = HL + t;
        // ---- load the column's fresh samples
        float2 x[DECIM];
        const bool full = (base >= g0) && (base + 12 <= gend);
        int nfresh = 0;
        if (full) {
            const float2 *p = in + (base - g0);
            if (aligned16) {
                const float4 *p4 = reinterpret_cast<const float4 *>(p);
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    float4 v = p4[k];
                    x[2 * k] = make_float2(v.x, v.y); x[2 * k + 1] = make_float2(v.z, v.w);
                }
            } else {
#pragma unroll
                for (int k = 0; k < DECIM; k++) x[k] = p[k];
            }
            nfresh = 12;
        } else {
#pragma unroll
            for (int k = 0; k < DECIM; k++) {
                int64_t idx = base + k;
                bool fr = (idx >= g0) && (idx < gend);
                x[k] = fr ? in[idx - g0] : make_float2(0.f, 0.f);
                nfresh += fr ? 1 : 0;
            }
        }
        const int first = (base >= g0) ? 0 : (int)((g0 - base) < 12 ? (g0 - base) : 12);   // first fresh row
        const int lastp1 = first + nfresh;            // fresh rows are [first, lastp1)

        // ---- RF DC removal: per-column affine map, block scan in f64, then the reference's own
        //      f32 recurrence RfDC = (x - RfDC)*alpha + RfDC from the scanned prefix.
        if (dcr) {
            Aff a; a.m = 1.0; a.ar = 0.0; a.ai = 0.0;
#pragma unroll
            for (int k = 0; k < DECIM; k++) {
                if (k >= first && k < lastp1) {
                    a.m *= beta;
                    a.ar = a.ar * beta + (double)alpha * (double)x[k].x;
                    a.ai = a.ai * beta + (double)alpha * (double)x[k].y;
                }
            }
            Aff inc = a;                              // inclusive scan over the wave
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                Aff o; o.m = shfl_up_d(inc.m, d); o.ar = shfl_up_d(inc.ar, d); o.ai = shfl_up_d(inc.ai, d);
                if (lane >= d) inc = aff_then(o, inc);
            }
            if (lane == 63) { wave_tot[wave][0] = inc.m; wave_tot[wave][1] = inc.ar; wave_tot[wave][2] = inc.ai; }
            Aff exc;                                  // exclusive prefix within the wave
            exc.m = shfl_up_d(inc.m, 1); exc.ar = shfl_up_d(inc.ar, 1); exc.ai = shfl_up_d(inc.ai, 1);
            if (lane == 0) { exc.m = 1.0; exc.ar = 0.0; exc.ai = 0.0; }
            __syncthreads();
            Aff pre; pre.m = 1.0; pre.ar = 0.0; pre.ai = 0.0;
            for (int w = 0; w < wave; w++) {
                Aff wv; wv.m = wave_tot[w][0]; wv.ar = wave_tot[w][1]; wv.ai = wave_tot[w][2];
                pre = aff_then(pre, wv);
            }
            pre = aff_then(pre, exc);
            const double c0 = carry[0], c1 = carry[1];
            float rr = (float)(pre.m * c0 + pre.ar);
            float ri = (float)(pre.m * c1 + pre.ai);
#pragma unroll
            for (int k = 0; k < DECIM; k++) {
                if (k >= first && k < lastp1) {
                    rr = (x[k].x - rr) * alpha + rr;
                    ri = (x[k].y - ri) * alpha + ri;
                    float cr = fminf(fmaxf(rr, -0.01f), 0.01f);     // DCRlimit :429-442
                    float ci = fminf(fmaxf(ri, -0.01f), 0.01f);
                    x[k].x -= cr; x[k].y -= ci;
                }
            }
            __syncthreads();                          // everyone has read carry
            if (t == 255) {
                Aff tot = aff_then(pre, a);
                carry[0] = tot.m * c0 + tot.ar; carry[1] = tot.m * c1 + tot.ai;
            }
        }
        // ---- IQ balance + LO mix
        if (Lg != 1.0f || Rg != 1.0f) {
#pragma unroll
            for (int k = 0; k < DECIM; k++) { x[k].x *= Lg; x[k].y *= Rg; }
        }
        if (mix && nfresh > 0) {
            // LOPhase after sample i (0-based within the call) = (P0 - (i+1)*lo) mod R
            long long i1 = (long long)(base + first - g0) + 1;
            long long m = (i1 * (long long)lo) % (long long)R;
            int ph = (int)(((long long)lo_phase0 - m) % (long long)R);
            if (ph < 0) ph += R;
#pragma unroll
            for (int k = 0; k < DECIM; k++) {
                if (k >= first && k < lastp1) {
                    float2 w = T.lo_table[ph];
                    float2 v = x[k];
                    x[k] = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
                    ph -= lo;
                    if (ph < 0) ph += R; else if (ph >= R) ph -= R;
                }
            }
        }
        // ---- to LDS.  Rows of the first column that precede g0 keep their history values.
#pragma unroll
        for (int k = 0; k < DECIM; k++) {
            bool keep = (k < first) && (q == qa);
            if (!keep) X[k][cc] = (k >= first && k < lastp1) ? x[k] : make_float2(0.f, 0.f);
        }
        __syncthreads();

        // ---- polyphase FIR: out[j] = sum_d sum_r T[d][r] * X[r][cc - d]
        const int64_t j = q;
        if (j >= ja && j < jb) {
            float ar = 0.f, ai = 0.f;
            for (int d = 0; d < nd; d++) {
                const float *tp = taps + d * DECIM;
#pragma unroll
                for (int r = 0; r < DECIM; r++) {
                    const float w = tp[r];
                    const float2 v = X[r][cc - d];
                    ar = fmaf(w, v.x, ar); ai = fmaf(w, v.y, ai);
                }
            }
            float2 z = make_float2(ar * FS.gain_re - ai * FS.gain_im, ar * FS.gain_im + ai * FS.gain_re);
            zring[j & G.ring_mask] = z;
        }
        __syncthreads();
        // ---- slide: columns cc 256..279 -> 0..23 (only when another tile follows)
        if (qt + A_TILE_COLS <= qb) {
            float2 mv[2]; int cnt = 0;
            for (int i = t; i < DECIM * HL; i += 256) { int r = i / HL, c = i - r * HL; mv[cnt++] = X[r][A_TILE_COLS + c]; }
            __syncthreads();
            cnt = 0;
            for (int i = t; i < DECIM * HL; i += 256) { int r = i / HL, c = i - r * HL; X[r][c] = mv[cnt++]; }
            __syncthreads();
        } else {
            // ---- last tile: save history for the next call
            const int64_t qn = gend / 12;             // column of the next call's first sample
            const int64_t cbase = qn - HL - (qt - HL);   // LDS column of history slot 0
            for (int i = t; i < DECIM * A_HIST_COLS; i += 256) {
                int r = i / A_HIST_COLS, c = i - r * A_HIST_COLS;
                int64_t lc = cbase + c;
                float2 v = make_float2(0.f, 0.f);
                if (lc >= 0 && lc < XCOLS) v = X[r][lc];
                hist[i] = v;
            }
        }
    }
    if (t == 0) {
        if (dcr || (P.actions & ACT_DC_RESET)) { st->dc_re = (float)carry[0]; st->dc_im = (float)carry[1]; }
        if (lo != 0) {
            long long m = ((long long)n * (long long)lo) % (long long)R;
            int ph = (int)(((long long)lo_phase0 - m) % (long long)R);
            if (ph < 0) ph += R;
            st->lo_phase = ph;
        }
    }
}

void launch_front(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, const float2 *iq,
                  int channels, hipStream_t s) {
    hipLaunchKernelGGL(front_kernel, dim3(channels), dim3(256), 0, s, T, B, G, iq);
}

}  // namespace fmx
