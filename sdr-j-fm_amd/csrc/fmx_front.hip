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
constexpr int CPT = 2;                         // columns per thread in the load / DC / mix / store phases
constexpr int TCOLS = 256 * CPT;               // 512 columns = 6144 input samples per tile
constexpr int XCOLS = HL + TCOLS;              // 536
constexpr int SPT = DECIM * CPT;               // 24 samples per thread per tile
constexpr int FCOLS = 8;                       // adjacent columns (= outputs) per lane in the FIR phase
constexpr int RPW = DECIM / 4;                 // polyphase rows per wave in the FIR phase

// LDS image of a tile: X[r][C], r = sample index mod 12, C = column (0..23 history, 24..535 fresh).  The unit of
// storage is the float4 holding the column pair (C even, C+1); unit index = r*XRS + ((C%8)/2)*XS4 + C/8, i.e. for a
// fixed row and pair slot the 8-column groups are contiguous.  That makes BOTH access patterns conflict-free
// ds_read_b128: the FIR phase (lane l reads groups l .. l+3 of one pair slot) and the DC phase (thread t reads the
// pair slot t%4 of group 3 + t/4; XS4 = 4 mod 16 spreads the four slots over the 16 sixteen-byte bank slots for the
// lane groups ds_read_b128 is serviced in).
constexpr int XS4 = 68;                        // >= 67 groups, = 4 (mod 16)
constexpr int XRS = 4 * XS4 + 1;               // row stride in units
__device__ __forceinline__ int xunit(int r, int C) { return r * XRS + ((C & 7) >> 1) * XS4 + (C >> 3); }
__device__ __forceinline__ int xidx(int r, int C) { return 2 * xunit(r, C) + (C & 1); }   // float2 index

typedef float v2f __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(4))) const float cfloat;    // constant address space: uniform loads become s_load

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

// FIR phase of one wave: rows r0 .. r0+2, eight adjacent outputs per lane.  For each row the lane holds the window
// W[0..31] = X[r][8 l .. 8 l + 31] (tile columns; output column 24 + 8 l + k uses W[24 + k - d]) in registers, the
// row's taps sit in SGPRs, and every tap feeds eight packed FMAs (re, im): 1 LDS byte per 1.6 flop.
template <int ND>
__device__ __forceinline__ void fir_rows(const float4 *__restrict__ X4, int lane, int r0, const float4 *__restrict__ tp, v2f acc[FCOLS]) {
    constexpr int JMIN = (HL - (ND - 1)) / 8;                       // first 8-column group the taps reach
#pragma unroll 1
    for (int rr = 0; rr < RPW; rr++) {
        float tw[(ND + 3) / 4 * 4];
#pragma unroll
        for (int d4 = 0; d4 < (ND + 3) / 4; d4++) {                  // wave-uniform address: one broadcast LDS cycle each
            const float4 v = tp[rr * (A_TAPS_ROW / 4) + d4];
            tw[4 * d4] = v.x; tw[4 * d4 + 1] = v.y; tw[4 * d4 + 2] = v.z; tw[4 * d4 + 3] = v.w;
        }
        v2f W[32];
#pragma unroll
        for (int j = 3; j >= JMIN; j--)                              // newest columns first: the taps d = 0.. use them first
#pragma unroll
            for (int kp = 0; kp < 4; kp++) {
                const float4 v = X4[(r0 + rr) * XRS + kp * XS4 + lane + j];
                W[8 * j + 2 * kp] = (v2f){v.x, v.y};
                W[8 * j + 2 * kp + 1] = (v2f){v.z, v.w};
            }
#pragma unroll
        for (int d = 0; d < ND; d++) {
            const v2f w = (v2f){tw[d], tw[d]};
#pragma unroll
            for (int k = 0; k < FCOLS; k++) acc[k] = __builtin_elementwise_fma(w, W[HL + k - d], acc[k]);
        }
    }
}

__global__ __launch_bounds__(256, 2) void front_kernel(DeviceTables T, DeviceBuffers B, CallGeom G,
                                                       const float2 *__restrict__ iq) {
    __shared__ __attribute__((aligned(16))) float4 X4[DECIM * XRS];
    __shared__ __attribute__((aligned(16))) float4 red[4][64][FCOLS / 2];   // per-wave partial sums, [wave][lane][pair]
    __shared__ __attribute__((aligned(16))) float sT[A_TAPS_DEV];          // the channel's tap set Trd[r][d]
    __shared__ float wave_tot[4][3];
    __shared__ float carry[2][2];                    // double-buffered by tile parity
    float2 *X2 = reinterpret_cast<float2 *>(X4);

    const int ch = blockIdx.x;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const ChanParams P = B.params[ch];
    const FrontSet FS = T.front_sets[P.front_set];
    const float2 *__restrict__ in = iq + (size_t)P.stream * G.stream_stride;
    ChanState *st = B.state + ch;
    float2 *hist = B.hist + (size_t)ch * DECIM * A_HIST_COLS;
    float2 *zring = B.zring + (size_t)ch * (G.ring_mask + 1);

    const int off = FS.off, nd = FS.nd;
    // Call-local 32-bit geometry: sample index s = global index - 12 qa, column index = global column - qa.
    const int64_t qa = G.g0 / 12;                     // column holding the first fresh sample
    const int r0 = (int)(G.g0 - qa * 12);             // the call's fresh samples are s in [g0, gend)
    const int g0 = r0, n = (int)G.n, gend = r0 + n;
    const int ja = (int)((G.g0 - off + 11) / 12 - qa);            // first output completed by this call
    const int jb = (int)((G.g0 + G.n - off + 11) / 12 - qa);      // one past the last
    const int qb = (gend - 1) / 12;                   // column holding the last fresh sample
    const int zr0 = (int)(qa & (int64_t)G.ring_mask);

    // ---- history -> LDS (columns qa-24 .. qa-1 at C 0..23, partial column qa at C 24)
    for (int i = t; i < DECIM * A_HIST_COLS; i += 256) {
        int r = i / A_HIST_COLS, c = i - r * A_HIST_COLS;
        float2 v = hist[i];
        if (c == HL && r >= r0) v = make_float2(0.f, 0.f);
        X2[xidx(r, c)] = v;
    }
    if (t == 0) {
        const bool rst = (P.actions & ACT_DC_RESET) != 0;        // setDCRemove zeroes RfDC (:922-925)
        carry[0][0] = rst ? 0.f : st->dc_re; carry[0][1] = rst ? 0.f : st->dc_im;
    }
    for (int i = t; i < A_TAPS_DEV; i += 256) sT[i] = T.front_taps[(size_t)P.front_set * A_TAPS_DEV + i];
    // this wave's three tap rows, d-contiguous: Trd[r][d] = G[12 d + off - r]
    const float4 *tp = reinterpret_cast<const float4 *>(sT + RPW * wave * A_TAPS_ROW);

    const bool dcr = P.dc_remove != 0;
    const int lo = P.lo_freq;
    const bool mix = (lo != 0) && (T.lo_table != nullptr);
    const int R = G.input_rate;
    const float alpha = 1.0f / (float)R;              // rfDcAlpha fm-processor.cpp:379
    const float Lg = P.att_l, Rg = P.att_r;
    const bool aligned16 = ((g0 & 1) == 0) && ((G.stream_stride & 1) == 0) &&
                           ((reinterpret_cast<uintptr_t>(iq) & 15) == 0);
    const int lo_phase0 = st->lo_phase;
    float u_full = 0.f;                               // u of a full 24-sample run (the same for every such thread)
    for (int k = 0; k < SPT; k++) u_full = (1.0f - u_full) * alpha + u_full;

    // Each wave owns a quarter of the tile: 128 columns = 1536 consecutive samples.  It loads them with
    // fully coalesced float4 loads (lane l, step k -> sample pair l + 64 k of the quarter), scatters them into
    // its own X columns, and each lane then reads back "its" two columns (24 consecutive samples in time).
    constexpr int WCOLS = TCOLS / 4, WSAMP = WCOLS * DECIM;      // 128 columns, 1536 samples
    float4 raw[SPT / 2];
    int sc_idx[SPT / 2];                                          // float2 index of sample pair k's first sample
#pragma unroll
    for (int k = 0; k < SPT / 2; k++) {
        const int e = 2 * (lane + 64 * k);                        // sample index within the wave's quarter (even)
        const int c = e / 12, r = e - 12 * c;                     // r is even: the pair stays inside one column
        sc_idx[k] = xidx(r, HL + WCOLS * wave + c);
    }
    auto load_tile = [&](int qt) {
        const int wbase = (qt + WCOLS * wave) * 12;              // index of the wave's first sample
        if (aligned16 && wbase >= g0 && wbase + WSAMP <= gend) {
            const float4 *p4 = reinterpret_cast<const float4 *>(in + (wbase - g0));
#pragma unroll
            for (int k = 0; k < SPT / 2; k++) raw[k] = p4[lane + 64 * k];
        } else {
#pragma unroll
            for (int k = 0; k < SPT / 2; k++) {
                const int i0 = wbase + 2 * (lane + 64 * k);
                const float2 a = (i0 >= g0 && i0 < gend) ? in[i0 - g0] : make_float2(0.f, 0.f);
                const float2 b = (i0 + 1 >= g0 && i0 + 1 < gend) ? in[i0 + 1 - g0] : make_float2(0.f, 0.f);
                raw[k] = make_float4(a.x, a.y, b.x, b.y);
            }
        }
    };
    const bool dbg_on = (B.dbg != nullptr) && (t == 0);
    unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long dbg_t = dbg_on ? clock64() : 0ull;
    load_tile(0);
    // history slide: the last 24 columns (groups 64..66) of every row / pair slot move to groups 0..2
    const int sl_r = t / 12, sl_rem = t - 12 * sl_r;
    const int sl_unit = sl_r * XRS + (sl_rem / 3) * XS4 + (sl_rem % 3);
    float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f);
    bool slide = false;
    int it = 0;
    const int dc_unit = (t & 3) * XS4 + 3 + (t >> 2);   // this thread's column pair (24 + 2t, 24 + 2t + 1), row 0
    FMX_TICK(0);

    for (int qt = 0; qt <= qb; qt += TCOLS, it++) {
        // ---- finish the slide of the previous tile and scatter the raw samples
        if (slide && t < DECIM * 12) X4[sl_unit] = m0;
        {
            const int wbase = (qt + WCOLS * wave) * 12;
            const bool allfresh = (wbase >= g0) && (wbase + WSAMP <= gend);
            if (allfresh) {
#pragma unroll
                for (int k = 0; k < SPT / 2; k++) {
                    X2[sc_idx[k]] = make_float2(raw[k].x, raw[k].y);
                    X2[sc_idx[k] + 2 * XRS] = make_float2(raw[k].z, raw[k].w);
                }
            } else {
#pragma unroll
                for (int k = 0; k < SPT / 2; k++) {
                    // samples before g0 keep their history value; samples from gend on are zero
                    const int i0 = wbase + 2 * (lane + 64 * k);
                    if (i0 >= g0) X2[sc_idx[k]] = make_float2(raw[k].x, raw[k].y);
                    if (i0 + 1 >= g0) X2[sc_idx[k] + 2 * XRS] = make_float2(raw[k].z, raw[k].w);
                }
            }
        }
        // no workgroup barrier here: a wave scatters exactly the columns its own lanes pick up below (wave w: columns
        // 128 w .. 128 w + 127), LDS operations of one wave complete in order, and the slid history columns are only read
        // by the FIR phase, two barriers further on
        __builtin_amdgcn_wave_barrier();
        FMX_TICK(1);
        const int q = qt + CPT * t;                   // this thread's first column
        const int base = q * 12;
        // fresh samples of this thread are rows [first, lastp1) of its 24
        int first = (base >= g0) ? 0 : ((g0 - base) < SPT ? (g0 - base) : SPT);
        int lastp1 = (base + SPT <= gend) ? SPT : ((gend - base) > 0 ? (gend - base) : 0);
        if (lastp1 < first) lastp1 = first;
        const bool wave_full = __all(first == 0 && lastp1 == SPT);

        if (dcr || mix || Lg != 1.0f || Rg != 1.0f) {
            v2f x[SPT];
#pragma unroll
            for (int r = 0; r < DECIM; r++) {
                const float4 v = X4[dc_unit + r * XRS];
                x[r] = (v2f){v.x, v.y}; x[r + DECIM] = (v2f){v.z, v.w};
            }
            // ---- RF DC removal (fm-processor.cpp:423-446): per-thread run, block scan of the affine maps, then
            //      the reference's own f32 recurrence RfDC = (x - RfDC)*alpha + RfDC from the scanned prefix.
            if (dcr) {
                Aff a; a.u = 0.f;
                v2f aa = (v2f){0.f, 0.f};
                const v2f al = (v2f){alpha, alpha};
                if (wave_full) {
#pragma unroll
                    for (int k = 0; k < SPT; k++) aa = __builtin_elementwise_fma(x[k] - aa, al, aa);
                    a.u = u_full;
                } else {
#pragma unroll
                    for (int k = 0; k < SPT; k++) {
                        if (k >= first && k < lastp1) {
                            a.u = (1.0f - a.u) * alpha + a.u;
                            aa = __builtin_elementwise_fma(x[k] - aa, al, aa);
                        }
                    }
                }
                a.ar = aa.x; a.ai = aa.y;
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
                v2f rr = (v2f){c0 - c0 * pre.u + pre.ar, c1 - c1 * pre.u + pre.ai};
                if (wave_full) {
#pragma unroll
                    for (int k = 0; k < SPT; k++) {
                        rr = __builtin_elementwise_fma(x[k] - rr, al, rr);
                        x[k] -= (v2f){fminf(fmaxf(rr.x, -0.01f), 0.01f), fminf(fmaxf(rr.y, -0.01f), 0.01f)};   // DCRlimit :429-442
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < SPT; k++) {
                        if (k >= first && k < lastp1) {
                            rr = __builtin_elementwise_fma(x[k] - rr, al, rr);
                            x[k] -= (v2f){fminf(fmaxf(rr.x, -0.01f), 0.01f), fminf(fmaxf(rr.y, -0.01f), 0.01f)};
                        }
                    }
                }
                if (t == 255) { carry[(it + 1) & 1][0] = rr.x; carry[(it + 1) & 1][1] = rr.y; }   // new state
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
                        const v2f v = x[k];
                        x[k] = (v2f){v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x};
                        ph -= lo;
                        if (ph < 0) ph += R; else if (ph >= R) ph -= R;
                    }
                }
            }
            // ---- back to LDS in place (entries that are not fresh pass through unchanged)
#pragma unroll
            for (int r = 0; r < DECIM; r++)
                X4[dc_unit + r * XRS] = make_float4(x[r].x, x[r].y, x[r + DECIM].x, x[r + DECIM].y);
            __syncthreads();
        }
        FMX_TICK(2);
        // ---- prefetch the next tile's raw samples; they land while the FIR runs
        const bool more = (qt + TCOLS <= qb);
        if (more) load_tile(qt + TCOLS);
        FMX_TICK(3);

        // ---- polyphase FIR  out[j] = sum_d sum_r Trd[r][d] * X[r][C_j - d]:  wave w sums rows 3w..3w+2 for all
        //      512 outputs (eight adjacent ones per lane), the four partial sums meet in LDS
        {
            v2f acc[FCOLS];
#pragma unroll
            for (int k = 0; k < FCOLS; k++) acc[k] = (v2f){0.f, 0.f};
            if (nd <= 4) fir_rows<4>(X4, lane, RPW * wave, tp, acc);
            else fir_rows<A_MAX_ND>(X4, lane, RPW * wave, tp, acc);
#pragma unroll
            for (int k = 0; k < FCOLS / 2; k++)
                red[wave][lane][(k + (lane >> 1)) & 3] = make_float4(acc[2 * k].x, acc[2 * k].y, acc[2 * k + 1].x, acc[2 * k + 1].y);
        }
        // the 24 columns that become the next tile's history (written back at the top of the next iteration)
        slide = more;
        if (more && t < DECIM * 12) m0 = X4[sl_unit + 64];
        __syncthreads();
        FMX_TICK(4);
        {
            // outputs 2t, 2t+1 = pair (t & 3) of FIR lane t >> 2
            const int fl = t >> 2, pr = ((t & 3) + (fl >> 1)) & 3;
            const float4 s0 = red[0][fl][pr], s1 = red[1][fl][pr], s2 = red[2][fl][pr], s3 = red[3][fl][pr];
            const float2 aA = make_float2((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y));
            const float2 aB = make_float2((s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
            if (q >= ja && q < jb)
                zring[(zr0 + q) & G.ring_mask] = make_float2(aA.x * FS.gain_re - aA.y * FS.gain_im, aA.x * FS.gain_im + aA.y * FS.gain_re);
            if (q + 1 >= ja && q + 1 < jb)
                zring[(zr0 + q + 1) & G.ring_mask] = make_float2(aB.x * FS.gain_re - aB.y * FS.gain_im, aB.x * FS.gain_im + aB.y * FS.gain_re);
        }
        FMX_TICK(5);
        if (!more) {
            // ---- last tile: save history for the next call
            const int qn = gend / 12;                 // column of the next call's first sample
            const int cbase = qn - qt;                // LDS column of history slot 0 (= column qn-24)
            for (int i = t; i < DECIM * A_HIST_COLS; i += 256) {
                int r = i / A_HIST_COLS, c = i - r * A_HIST_COLS;
                int lc = cbase + c;
                float2 v = make_float2(0.f, 0.f);
                if (lc >= 0 && lc < XCOLS) v = X2[xidx(r, lc)];
                hist[i] = v;
            }
        }
    }
    FMX_TICK(6);
    if (dbg_on) for (int k = 0; k < 8; k++) B.dbg[(size_t)ch * 16 + k] += dbg_acc[k];
    if (t == 0) {
        if (dcr || (P.actions & ACT_DC_RESET)) { st->dc_re = carry[it & 1][0]; st->dc_im = carry[it & 1][1]; }
        if (lo != 0) {
            long long m = ((long long)G.n * (long long)lo) % (long long)R;
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
