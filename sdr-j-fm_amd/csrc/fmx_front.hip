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
constexpr int CPT = 2;                         // columns (= outputs) per thread
constexpr int TCOLS = 256 * CPT;               // 512 columns = 6144 input samples per tile
constexpr int XCOLS = HL + TCOLS;              // 536
constexpr int XS = XCOLS + 2;                  // row stride in float2 (even: rows stay 16-B aligned)
constexpr int SPT = DECIM * CPT;               // 24 samples per thread per tile

// The DC recurrence r <- r + alpha (x - r) over a run of samples is the affine map r -> r (1 - u) + a.
// (u, a) are kept instead of (m = 1 - u, a): u ~ count * alpha is tiny, so f32 holds it to 1e-7 relative,
// whereas 1 - alpha itself is not representable to better than 7 % of alpha in f32.
struct Aff { float u, ar, ai; };
__device__ __forceinline__ Aff aff_then(const Aff &f, const Aff &g) {      // apply f, then g
    Aff o;
    o.u = f.u + g.u - f.u * g.u;
    o.ar = f.ar + g.ar - f.ar * g.u;
    o.ai = f.ai + g.ai - f.ai * g.u;
    return o;
}

#define FMX_TICK(k) do { if (dbg_on) { unsigned long long now_ = clock64(); dbg_acc[k] += now_ - dbg_t; dbg_t = now_; } } while (0)

// (A variant that passed the tap set by value in the kernel arguments, to get the taps into SGPRs, ran 14x
//  slower on gfx950 -- the dynamically indexed kernarg array is not turned into scalar loads -- and was dropped.)
__global__ __launch_bounds__(256, 2) void front_kernel(DeviceTables T, DeviceBuffers B, CallGeom G,
                                                       const float2 *__restrict__ iq) {
    __shared__ __attribute__((aligned(16))) float2 X[DECIM][XS];
    __shared__ __attribute__((aligned(16))) float sT[A_TAPS_STRIDE];   // the channel's tap set (wave-uniform broadcast reads)
    __shared__ float wave_tot[4][3];
    __shared__ float carry[2][2];                    // double-buffered by tile parity

    const int ch = blockIdx.x;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const ChanParams P = B.params[ch];
    const FrontSet FS = T.front_sets[P.front_set];
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

    // ---- history -> LDS (columns qa-24 .. qa-1 at L 0..23, partial column qa at L 24)
    for (int i = t; i < DECIM * A_HIST_COLS; i += 256) {
        int r = i / A_HIST_COLS, c = i - r * A_HIST_COLS;
        float2 v = hist[i];
        if (c == HL && r >= r0) v = make_float2(0.f, 0.f);
        X[r][c] = v;
    }
    {
        const float *__restrict__ taps = T.front_taps + (size_t)P.front_set * A_TAPS_STRIDE;
        for (int i = t; i < A_TAPS_STRIDE; i += 256) sT[i] = taps[i];
    }
    if (t == 0) {
        const bool rst = (P.actions & ACT_DC_RESET) != 0;        // setDCRemove zeroes RfDC (:922-925)
        carry[0][0] = rst ? 0.f : st->dc_re; carry[0][1] = rst ? 0.f : st->dc_im;
    }

    const bool dcr = P.dc_remove != 0;
    const int lo = P.lo_freq;
    const bool mix = (lo != 0) && (T.lo_table != nullptr);
    const int R = G.input_rate;
    const float alpha = 1.0f / (float)R;              // rfDcAlpha fm-processor.cpp:379
    const float Lg = P.att_l, Rg = P.att_r;
    const bool aligned16 = ((g0 & 1) == 0) && ((G.stream_stride & 1) == 0) &&
                           ((reinterpret_cast<uintptr_t>(iq) & 15) == 0);
    const int lo_phase0 = st->lo_phase;
    const int pmin = (25 - nd) / 2;                  // first column pair that holds a non-zero tap

    // Each wave owns a quarter of the tile: 128 columns = 1536 consecutive samples.  It loads them with
    // fully coalesced float4 loads (lane l, step k -> sample pair l + 64 k of the quarter), scatters them into
    // its own X columns, and each lane then reads back "its" two columns (24 consecutive samples in time).
    constexpr int WCOLS = TCOLS / 4, WSAMP = WCOLS * DECIM;      // 128 columns, 1536 samples
    float4 raw[SPT / 2];
    auto load_tile = [&](int64_t qt) {
        const int64_t wbase = (qt + WCOLS * wave) * 12;          // global index of the wave's first sample
        if (aligned16 && wbase >= g0 && wbase + WSAMP <= gend) {
            const float4 *p4 = reinterpret_cast<const float4 *>(in + (wbase - g0));
#pragma unroll
            for (int k = 0; k < SPT / 2; k++) raw[k] = p4[lane + 64 * k];
        } else {
#pragma unroll
            for (int k = 0; k < SPT / 2; k++) {
                const int64_t i0 = wbase + 2 * (lane + 64 * k);
                const float2 a = (i0 >= g0 && i0 < gend) ? in[i0 - g0] : make_float2(0.f, 0.f);
                const float2 b = (i0 + 1 >= g0 && i0 + 1 < gend) ? in[i0 + 1 - g0] : make_float2(0.f, 0.f);
                raw[k] = make_float4(a.x, a.y, b.x, b.y);
            }
        }
    };
    const bool dbg_on = (B.dbg != nullptr) && (t == 0);
    unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long dbg_t = dbg_on ? clock64() : 0ull;
    load_tile(qa);
    float2 m0 = make_float2(0.f, 0.f), m1 = m0;      // history columns being slid to the front of the next tile
    bool slide = false;
    int it = 0;
    FMX_TICK(0);

    for (int64_t qt = qa; qt <= qb; qt += TCOLS, it++) {
        // ---- finish the slide of the previous tile (columns L 512..535 -> 0..23) and scatter the raw samples
        if (slide) {
            const int i0 = t, i1 = t + 256;                       // DECIM*HL = 288 elements
            const int ra = i0 / HL, ca = i0 - ra * HL, rb = i1 / HL, cb = i1 - rb * HL;
            X[ra][ca] = m0;
            if (i1 < DECIM * HL) X[rb][cb] = m1;
        }
        {
            const int64_t wbase = (qt + WCOLS * wave) * 12;
            const bool allfresh = (wbase >= g0) && (wbase + WSAMP <= gend);
#pragma unroll
            for (int k = 0; k < SPT / 2; k++) {
                const int e = 2 * (lane + 64 * k);                // sample index within the wave's quarter (even)
                const int c = e / 12, r = e - 12 * c;             // r is even: the pair stays inside one column
                const int L = HL + WCOLS * wave + c;
                if (allfresh) {
                    X[r][L] = make_float2(raw[k].x, raw[k].y);
                    X[r + 1][L] = make_float2(raw[k].z, raw[k].w);
                } else {
                    // samples before g0 keep their history value; samples from gend on are zero
                    const int64_t i0 = wbase + e;
                    if (i0 >= g0) X[r][L] = make_float2(raw[k].x, raw[k].y);
                    if (i0 + 1 >= g0) X[r + 1][L] = make_float2(raw[k].z, raw[k].w);
                }
            }
        }
        __syncthreads();
        FMX_TICK(1);
        const int64_t q = qt + CPT * t;               // this thread's first column
        const int64_t base = q * 12;
        const int L0 = HL + CPT * t;
        // fresh samples of this thread are rows [first, lastp1) of its 24
        int first = (base >= g0) ? 0 : (int)((g0 - base) < SPT ? (g0 - base) : SPT);
        int lastp1 = (base + SPT <= gend) ? SPT : (int)((gend - base) > 0 ? (gend - base) : 0);
        if (lastp1 < first) lastp1 = first;
        const bool wave_full = __all(first == 0 && lastp1 == SPT);

        if (dcr || mix || Lg != 1.0f || Rg != 1.0f) {
            float2 x[SPT];
#pragma unroll
            for (int r = 0; r < DECIM; r++) {
                const float4 v = *reinterpret_cast<const float4 *>(&X[r][L0]);
                x[r] = make_float2(v.x, v.y); x[r + DECIM] = make_float2(v.z, v.w);
            }
            // ---- RF DC removal (fm-processor.cpp:423-446): per-thread run, block scan of the affine maps, then
            //      the reference's own f32 recurrence RfDC = (x - RfDC)*alpha + RfDC from the scanned prefix.
            if (dcr) {
                Aff a; a.u = 0.f; a.ar = 0.f; a.ai = 0.f;
                if (wave_full) {
#pragma unroll
                    for (int k = 0; k < SPT; k++) {
                        a.u = (1.0f - a.u) * alpha + a.u;
                        a.ar = (x[k].x - a.ar) * alpha + a.ar;
                        a.ai = (x[k].y - a.ai) * alpha + a.ai;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < SPT; k++) {
                        if (k >= first && k < lastp1) {
                            a.u = (1.0f - a.u) * alpha + a.u;
                            a.ar = (x[k].x - a.ar) * alpha + a.ar;
                            a.ai = (x[k].y - a.ai) * alpha + a.ai;
                        }
                    }
                }
                Aff inc = a;                              // inclusive scan over the wave
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    Aff o; o.u = __shfl_up(inc.u, d, 64); o.ar = __shfl_up(inc.ar, d, 64); o.ai = __shfl_up(inc.ai, d, 64);
                    if (lane >= d) inc = aff_then(o, inc);
                }
                if (lane == 63) { wave_tot[wave][0] = inc.u; wave_tot[wave][1] = inc.ar; wave_tot[wave][2] = inc.ai; }
                Aff exc;                                  // exclusive prefix within the wave
                exc.u = __shfl_up(inc.u, 1, 64); exc.ar = __shfl_up(inc.ar, 1, 64); exc.ai = __shfl_up(inc.ai, 1, 64);
                if (lane == 0) { exc.u = 0.f; exc.ar = 0.f; exc.ai = 0.f; }
                __syncthreads();
                Aff pre; pre.u = 0.f; pre.ar = 0.f; pre.ai = 0.f;
                for (int w = 0; w < wave; w++) {
                    Aff wv; wv.u = wave_tot[w][0]; wv.ar = wave_tot[w][1]; wv.ai = wave_tot[w][2];
                    pre = aff_then(pre, wv);
                }
                pre = aff_then(pre, exc);
                const float c0 = carry[it & 1][0], c1 = carry[it & 1][1];
                float rr = c0 - c0 * pre.u + pre.ar;
                float ri = c1 - c1 * pre.u + pre.ai;
                if (wave_full) {
#pragma unroll
                    for (int k = 0; k < SPT; k++) {
                        rr = (x[k].x - rr) * alpha + rr;
                        ri = (x[k].y - ri) * alpha + ri;
                        x[k].x -= fminf(fmaxf(rr, -0.01f), 0.01f);       // DCRlimit :429-442
                        x[k].y -= fminf(fmaxf(ri, -0.01f), 0.01f);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < SPT; k++) {
                        if (k >= first && k < lastp1) {
                            rr = (x[k].x - rr) * alpha + rr;
                            ri = (x[k].y - ri) * alpha + ri;
                            x[k].x -= fminf(fmaxf(rr, -0.01f), 0.01f);
                            x[k].y -= fminf(fmaxf(ri, -0.01f), 0.01f);
                        }
                    }
                }
                if (t == 255) { carry[(it + 1) & 1][0] = rr; carry[(it + 1) & 1][1] = ri; }   // new state
            }
            // ---- IQ balance + LO mix (fm-processor.cpp:462-466, oscillator.cpp:49-58)
            if (Lg != 1.0f || Rg != 1.0f) {
#pragma unroll
                for (int k = 0; k < SPT; k++) if (k >= first && k < lastp1) { x[k].x *= Lg; x[k].y *= Rg; }
            }
            if (mix && lastp1 > first) {
                // LOPhase after sample i (0-based within the call) = (P0 - (i+1)*lo) mod R
                const long long i1 = (long long)(base + first - g0) + 1;
                const long long m = (i1 * (long long)lo) % (long long)R;
                int ph = (int)(((long long)lo_phase0 - m) % (long long)R);
                if (ph < 0) ph += R;
#pragma unroll
                for (int k = 0; k < SPT; k++) {
                    if (k >= first && k < lastp1) {
                        const float2 w = T.lo_table[ph];
                        const float2 v = x[k];
                        x[k] = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
                        ph -= lo;
                        if (ph < 0) ph += R; else if (ph >= R) ph -= R;
                    }
                }
            }
            // ---- back to LDS in place (entries that are not fresh pass through unchanged)
#pragma unroll
            for (int r = 0; r < DECIM; r++)
                *reinterpret_cast<float4 *>(&X[r][L0]) = make_float4(x[r].x, x[r].y, x[r + DECIM].x, x[r + DECIM].y);
            __syncthreads();
        }
        FMX_TICK(2);
        // ---- prefetch the next tile's raw samples; they land while the FIR runs
        const bool more = (qt + TCOLS <= qb);
        if (more) load_tile(qt + TCOLS);
        FMX_TICK(3);

        // ---- polyphase FIR for two adjacent outputs A (column L_A = HL+2t) and B (L_A + 1):
        //      out[j] = sum_d sum_r Tz[d+1][r] * X[r][L - d];  column pair p holds L = 2t+2p, 2t+2p+1
        // four partial sums per output (rows r mod 4): eight independent FMA chains hide the FMA latency
        float2 pA[4], pB[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { pA[i] = make_float2(0.f, 0.f); pB[i] = make_float2(0.f, 0.f); }
#pragma unroll 1
        for (int p = pmin; p <= HL / 2; p++) {
            // rows of Tz for d = 23-2p, 24-2p, 25-2p: 36 consecutive floats.  ONE ds_read_b32 (lane i < 36 reads
            // tap i), then v_readlane moves each tap into an SGPR that the packed FMAs take as a scalar operand.
            const int tv = __float_as_int(sT[(24 - 2 * p) * DECIM + (lane < 3 * DECIM ? lane : 0)]);
            float tw[3 * DECIM];
#pragma unroll
            for (int i = 0; i < 3 * DECIM; i++) tw[i] = __int_as_float(__builtin_amdgcn_readlane(tv, i));
#pragma unroll
            for (int r = 0; r < DECIM; r++) {
                const float w23 = tw[r], w24 = tw[DECIM + r], w25 = tw[2 * DECIM + r];
                const float4 v = *reinterpret_cast<const float4 *>(&X[r][CPT * t + 2 * p]);
                float2 &a = pA[r & 3], &b = pB[r & 3];
                a.x = fmaf(w24, v.x, a.x); a.y = fmaf(w24, v.y, a.y);
                a.x = fmaf(w23, v.z, a.x); a.y = fmaf(w23, v.w, a.y);
                b.x = fmaf(w25, v.x, b.x); b.y = fmaf(w25, v.y, b.y);
                b.x = fmaf(w24, v.z, b.x); b.y = fmaf(w24, v.w, b.y);
            }
        }
        const float2 aA = make_float2((pA[0].x + pA[1].x) + (pA[2].x + pA[3].x), (pA[0].y + pA[1].y) + (pA[2].y + pA[3].y));
        const float2 aB = make_float2((pB[0].x + pB[1].x) + (pB[2].x + pB[3].x), (pB[0].y + pB[1].y) + (pB[2].y + pB[3].y));
        FMX_TICK(4);
        if (q >= ja && q < jb)
            zring[q & G.ring_mask] = make_float2(aA.x * FS.gain_re - aA.y * FS.gain_im, aA.x * FS.gain_im + aA.y * FS.gain_re);
        if (q + 1 >= ja && q + 1 < jb)
            zring[(q + 1) & G.ring_mask] = make_float2(aB.x * FS.gain_re - aB.y * FS.gain_im, aB.x * FS.gain_im + aB.y * FS.gain_re);
        slide = more;
        if (more) {
            // ---- read the 24 columns that become the next tile's history (written back after the barrier)
            const int i0 = t, i1 = t + 256;
            const int ra = i0 / HL, ca = i0 - ra * HL, rb = i1 / HL, cb = i1 - rb * HL;
            m0 = X[ra][TCOLS + ca];
            m1 = (i1 < DECIM * HL) ? X[rb][TCOLS + cb] : make_float2(0.f, 0.f);
        }
        __syncthreads();
        FMX_TICK(5);
        if (!more) {
            // ---- last tile: save history for the next call
            const int64_t qn = gend / 12;             // column of the next call's first sample
            const int64_t cbase = qn - qt;            // LDS column of history slot 0 (= column qn-24)
            for (int i = t; i < DECIM * A_HIST_COLS; i += 256) {
                int r = i / A_HIST_COLS, c = i - r * A_HIST_COLS;
                int64_t lc = cbase + c;
                float2 v = make_float2(0.f, 0.f);
                if (lc >= 0 && lc < XCOLS) v = X[r][lc];
                hist[i] = v;
            }
        }
    }
    FMX_TICK(6);
    if (dbg_on) for (int k = 0; k < 8; k++) B.dbg[(size_t)ch * 16 + k] += dbg_acc[k];
    if (t == 0) {
        if (dcr || (P.actions & ACT_DC_RESET)) { st->dc_re = carry[it & 1][0]; st->dc_im = carry[it & 1][1]; }
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
