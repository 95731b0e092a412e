// fmx_front2.hip -- stage A as producer / consumer wave pairs with the FIR on the matrix pipe (the layout used when the batch
// fills the GPU; fmx_front.hip's four-waves-per-channel kernel serves smaller batches).
//
// Same arithmetic contract as fmx_front.hip (RF DC removal fm-processor.cpp:423-446, IQ balance + LO mix :462-466 /
// oscillator.cpp:49-58, inputFilter :469-470, fmBand_1 :472, fmBand_2 :474, folded into one real polyphase /12 FIR).
//
// What round 1's kernel could not overlap: its four waves per channel run load -> scatter -> DC -> hand-off -> FIR -> store one
// after the other, two waves to a SIMD, and its 600 packed FMAs per tile clock the chip down (a kernel of nothing but
// v_pk_fma_f32 sustains 92 TFLOP/s here, v_mfma_f32_16x16x4_f32 155.6: tools/ubench/mfma_f32.hip).  Here
//   * one 512-thread workgroup serves FOUR channels: wave p (producer) and wave p + 4 (consumer) land on the same SIMD and own
//     channel 4 blockIdx + p for the whole call -- no dependency between pairs, none on other workgroups;
//   * the producer streams the channel: coalesced tile loads two tiles ahead (24 KB in flight per producer), scatter into the
//     LDS image X[r][C] of the tile, DC removal (per-lane runs + DPP scan; the carry between tiles is a register now) and LO
//     mix, written back DE-INTERLEAVED (re C, re C+1, im C, im C+1) -- VALU / LDS / memory work only;
//   * the consumer runs the FIR of the previous tile meanwhile as 120 v_mfma_f32_16x16x4_f32 (Toeplitz form, see fir_mfma) and
//     stores the outputs -- matrix pipe only; the two meet through two sequence counters per pair in LDS;
//   * two images per pair (double buffer); the 24 history columns of a tile are written by the producer from its registers.
#include "fmx_internal.h"

namespace fmx {
namespace f2 {

constexpr int HL = A_HIST_COLS - 1;            // 24 history columns in front of a tile
constexpr int WCOLS = 128;                     // fresh columns (= outputs) per tile
constexpr int WSAMP = WCOLS * DECIM;           // 1536 input samples per tile
constexpr int XCOLS = HL + WCOLS;              // 152 columns in an image
constexpr int SPT = 2 * DECIM;                 // 24 samples per lane per tile (two adjacent columns)

// LDS image of a tile: X[r][C], r = sample index mod 12, C = column (0..23 history, 24..151 fresh).  Storage unit = the
// float4 of a column pair (C even, C + 1); unit index = r * XRS + ((C % 8) / 2) * XS4 + C / 8 (fmx_front.hip's layout: the DC
// phase's ds_read_b128 / ds_write_b128 -- lane l: pair slot l % 4 of group 3 + l / 4 -- are conflict-free, XS4 = 4 mod 16).
// Raw units (scatter) hold (re C, im C, re C+1, im C+1); the DC pass leaves (re C, re C+1, im C, im C+1).
constexpr int XS4 = 20;
constexpr int XRS = 4 * XS4 + 1;               // odd: rows r, r + 1 sit on odd / even 16-byte slots (matrix FIR reads)
constexpr int XUNITS = DECIM * XRS;            // 972 float4 = 15552 B per image
__device__ __forceinline__ int xunit(int r, int C) { return r * XRS + ((C & 7) >> 1) * XS4 + (C >> 3); }
__device__ __forceinline__ int xidx(int r, int C) { return 2 * xunit(r, C) + (C & 1); }                          // float2 index, raw unit
__device__ __forceinline__ int xidx_d(int r, int C, int comp) { return 4 * xunit(r, C) + 2 * comp + (C & 1); }  // float index, de-interleaved unit

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void seq_wait(int *p, int need) {
    while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void seq_post(int *p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// FIR on the matrix pipe.  The polyphase sum  out[c] = sum_r sum_d Trd[r][d] X[r][24 + c - d]  over a block of 16 adjacent
// outputs c = 16 b + i is the product of a 16 x 480 Toeplitz matrix of taps  A[i][(r, w)] = Trd[r][24 + i - w]  (zero outside
// 0 <= 24 + i - w <= 24) with the 40-column window  B[(r, w)][n] = X[r][16 b + w].comp,  n = (b, comp): the eight column
// blocks of a tile times (re, im) are exactly the 16 columns of v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate).  120 matrix
// instructions per tile; 38 % of their multiplies hit the zero corners of A, the price of running beside the producer's VALU
// work at a rate that does not throttle the clock.  K order: lane quarter kk = lane / 16 takes rows kk, kk + 4, kk + 8 and all 20
// column pairs of the window: one ds_read_b64 at unit + 8 comp of the de-interleaved image yields the lane's two B values, and
// the 32 lanes of an LDS service group (kk, kk + 1: odd / even 16-byte slots; 8 blocks; 2 halves) cover the 64 banks exactly.
//
// A operands: row r of the taps zero-padded to Tpad_r[e], e = d + 15 in [0, 55), sits in LDS at dword MF_ROW(r) + e (lanes with
// even i) and once more one dword lower at MF_TOFF + MF_ROW(r) + e - 1 (odd i), so that the lane's pair
// (Tpad[38 + i - 2 xp], Tpad[39 + i - 2 xp]) is one aligned 8-byte read either way; rows r, r + 1 are 32 banks apart and the two
// copies 16.  (120 registers per lane for them did not fit beside the accumulators: 99 spills.)
constexpr int MF_XP = 20;                      // column pairs in the 40-column window
__host__ __device__ constexpr int MF_ROW(int r) { return 160 * (r >> 1) + 96 * (r & 1); }
constexpr int MF_COPY = 960, MF_TOFF = 976, MF_ADW = MF_TOFF + MF_COPY;       // 1936 floats per tap image
constexpr int MF_G = 10;                       // steps per read group (register double buffer: reads run one group ahead)
constexpr int MF_NA = 3 * MF_XP * 2;           // A values per lane: registers of the consumer wave for the whole call
// B reads: the 20 unit addresses of a row (one per column pair of the window) are separate registers the compiler cannot see
// through, and the three rows of a lane are 4 XRS units = 5184 bytes apart -- no two reads of one address register can be paired
// into a ds_read2_b64 / ds_read2st64_b64, which move half the bytes per LDS cycle of a plain ds_read_b64.
typedef __attribute__((address_space(3))) const v2f lds_v2f;
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
template <int XP0>
__device__ __forceinline__ v4f fir_mfma(const uint32_t (&Bq)[MF_XP], const float (&A)[MF_NA]) {
    constexpr int NS = 3 * (MF_XP - XP0), NG = NS / MF_G;
    static_assert(NS % MF_G == 0, "steps per group");
    v4f acc0 = (v4f){0.f, 0.f, 0.f, 0.f}, acc1 = (v4f){0.f, 0.f, 0.f, 0.f};
    v2f vb[2][MF_G];
    auto load_group = [&](int g, int buf) {
#pragma unroll
        for (int k = 0; k < MF_G; k++) {
            const int st = g * MF_G + k, tt = st / (MF_XP - XP0), xp = XP0 + st % (MF_XP - XP0);
            vb[buf][k] = *reinterpret_cast<lds_v2f *>((uintptr_t)(Bq[xp] + 16 * (tt * 4 * XRS)));
        }
    };
    load_group(0, 0);
#pragma unroll
    for (int g = 0; g < NG; g++) {
        if (g + 1 < NG) load_group(g + 1, (g + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < MF_G; k++) {
            const int st = g * MF_G + k, tt = st / (MF_XP - XP0), xp = XP0 + st % (MF_XP - XP0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(tt * MF_XP + xp) * 2], vb[g & 1][k].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(tt * MF_XP + xp) * 2 + 1], vb[g & 1][k].y, acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc0 + acc1;
}

extern __shared__ __attribute__((aligned(16))) float4 f2_smem[];
#define F2_TICK(k) do { if (dbg_on) { unsigned long long now_ = clock64(); dbg_acc[k] += now_ - dbg_t; dbg_t = now_; } } while (0)
#ifndef F2_ABL
#define F2_ABL 0      /* diagnostic builds only (results WRONG): 1 no matrix FIR, 2 no DC / mix arithmetic, 4 no LDS work in the producer, 8 no loads */
#endif

// FMT: fmx_iq_format of the input (include/fmx.h); raw integer samples are converted while they are loaded, exactly as the
// reference's device handlers do (rtlsdr-handler.cpp:291, hackrf-handler.cpp:364, lime-handler.cpp:250).
// ntab: 1 = every channel of the batch uses the tap set of channel 0 (one A image per workgroup), 4 = one per pair.
// lo_cap: LDS entries per pair for one period of the LO (0: gather from the table in memory).
template <int FMT>
__global__ __launch_bounds__(512, 2) void front2_kernel(DeviceTables T, DeviceBuffers B, CallGeom G, const void *__restrict__ iq_raw,
                                                        int channels, int ntab, int lo_cap) {
    constexpr int BPS = (FMT == 0) ? 8 : (FMT == 3 ? 4 : 2);          // bytes per complex sample
    float4 *img_all = f2_smem;                                            // [4 pairs][2][XUNITS]
    float *sA_all = reinterpret_cast<float *>(f2_smem + 4 * 2 * XUNITS);  // [ntab][MF_ADW]
    float2 *sLO_all = reinterpret_cast<float2 *>(sA_all + ntab * MF_ADW);  // [4][lo_cap]
    int *seq = reinterpret_cast<int *>(sLO_all + 4 * lo_cap);              // prod_seq[4], cons_seq[4]

    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int pair = wave & 3;
    const bool consumer = wave >= 4;
    const int ch_raw = blockIdx.x * 4 + pair;
    const bool active = ch_raw < channels;
    const int ch = active ? ch_raw : channels - 1;                        // idle pairs read valid parameters and do nothing
    const ChanParams P = B.params[ch];
    const FrontSet FS = T.front_sets[P.front_set];
    ChanState *st = B.state + ch;
    float2 *hist = B.hist + (size_t)ch * DECIM * A_HIST_COLS;
    float2 *zring = B.zring + (size_t)ch * (G.ring_mask + 1);
    float4 *img0 = img_all + (size_t)pair * 2 * XUNITS;
    int *prod_seq = seq + pair, *cons_seq = seq + 4 + pair;
    const float *sA = sA_all + (ntab == 1 ? 0 : pair) * MF_ADW;
    float2 *sLO = sLO_all + (size_t)pair * lo_cap;

    const int off = FS.off, nd = FS.nd;
    // Call-local 32-bit geometry: sample index s = global index - 12 qa, column index = global column - qa.
    const int64_t qa = G.g0 / 12;                     // column holding the first fresh sample
    const int r0 = (int)(G.g0 - qa * 12);             // the call's fresh samples are s in [g0, gend)
    const int g0 = r0, gend = r0 + (int)G.n;
    const int ja = (int)((G.g0 - off + 11) / 12 - qa);            // first output completed by this call
    const int jb = (int)((G.g0 + G.n - off + 11) / 12 - qa);      // one past the last
    const int qb = (gend - 1) / 12;                   // column holding the last fresh sample
    const int NT = qb / WCOLS + 1;                    // tiles in this call
    const int zr0 = (int)(qa & (int64_t)G.ring_mask);

    // ---- tables: the tap images (every thread), the pair's LO period (its 128 threads), the counters
    for (int tb = 0; tb < ntab; tb++) {
        const int cht = (ntab == 1) ? 0 : min(blockIdx.x * 4 + tb, (unsigned)channels - 1);
        const size_t fs = (size_t)B.params[cht].front_set * A_TAPS_DEV;
        for (int i = t; i < 2 * DECIM * 56; i += 512) {
            const int cp = i / (DECIM * 56), r = (i - cp * DECIM * 56) / 56, pos = i - (cp * DECIM + r) * 56;
            const int d = pos + cp - 15;
            sA_all[tb * MF_ADW + cp * MF_TOFF + MF_ROW(r) + pos] = (d >= 0 && d <= HL) ? T.front_taps[fs + r * A_TAPS_ROW + d] : 0.f;
        }
    }
    if (t < 8) seq[t] = 0;
    // per-channel state is read BEFORE the barrier (the producer rewrites it when the call ends)
    const int lo_phase0 = st->lo_phase;
    const bool dc_rst = (P.actions & ACT_DC_RESET) != 0;          // setDCRemove zeroes RfDC (:922-925)
    const float dc0r = dc_rst ? 0.f : st->dc_re, dc0i = dc_rst ? 0.f : st->dc_im;
    // LO mix table: LOPhase after sample i of the call is (P0 - (i+1) lo) mod R; when lo / R has a short period p the p entries
    // the call will use sit in LDS, sLO[m] = T[(P0 - m lo) mod R] with m = (i + 1) mod p
    const int lo_per = (P.lo_freq != 0 && T.lo_table != nullptr && P.lo_period <= lo_cap) ? P.lo_period : 0;
    for (int m = lane + (consumer ? 64 : 0); m < lo_per; m += 128) {
        long long ph = ((long long)lo_phase0 - (long long)m * (long long)P.lo_freq) % (long long)G.input_rate;
        if (ph < 0) ph += G.input_rate;
        sLO[m] = T.lo_table[ph];
    }
    __syncthreads();                                  // the only workgroup barrier
    if (!active) return;

    if (consumer) {
        // =================================================================== consumer: matrix FIR + output store
        const int kk = lane >> 4, ni = lane & 15;                         // B / D lane: n = ni = 2 b + comp;  A lane: i = ni
        const bool odd = (lane & 1) != 0;
        // A[i][(r, w)] of this lane: rows kk + 4 tt, w = 2 xp + dc, from the zero-padded tap image (read once per call)
        float A[MF_NA];
        {
            const float2 *Ap = reinterpret_cast<const float2 *>(sA + (odd ? MF_TOFF + ni - 1 : ni) + MF_ROW(kk));
#pragma unroll
            for (int tt = 0; tt < 3; tt++)
#pragma unroll
                for (int xp = 0; xp < MF_XP; xp++) {
                    const float2 a = Ap[(320 * tt + 38 - 2 * xp) / 2];            // (tap of column 2 xp + 1, of 2 xp)
                    A[(tt * MF_XP + xp) * 2] = a.y; A[(tt * MF_XP + xp) * 2 + 1] = a.x;
                }
        }
        const int boff = 2 * (kk * XRS + (ni >> 1) * 2) + (lane & 1);     // float2 index of (row kk, column 16 b), half comp
        uint32_t Bq0[MF_XP], Bq1[MF_XP];                                  // LDS byte address per column pair of the window, image 0 / this tile's
#pragma unroll
        for (int xp = 0; xp < MF_XP; xp++) {
            Bq0[xp] = lds_addr(reinterpret_cast<const float2 *>(img0) + boff + 2 * ((xp & 3) * XS4 + (xp >> 2)));
            asm volatile("" : "+v"(Bq0[xp]));
        }
#ifndef F2_CONS_PRIO
#define F2_CONS_PRIO 0
#endif
        if (F2_CONS_PRIO) __builtin_amdgcn_s_setprio(F2_CONS_PRIO);
        const bool dbg_on = (B.dbg != nullptr) && (lane == 0);           // fmx_debug_phase_cycles: slots 4.. = consumer
        unsigned long long dbg_acc[3] = {0, 0, 0}, dbg_t = dbg_on ? clock64() : 0ull;
        for (int ti = 0; ti < NT; ti++) {
#pragma unroll
            for (int xp = 0; xp < MF_XP; xp++) Bq1[xp] = Bq0[xp] + ((ti & 1) ? 16 * XUNITS : 0);
            seq_wait(prod_seq, ti + 1);
            F2_TICK(0);
            v4f y;
            if (F2_ABL & 1) y = (v4f){(float)lane, 0.f, 1.f, 2.f};
            else if (nd <= 4) y = fir_mfma<MF_XP - 10>(Bq1, A);
            else y = fir_mfma<0>(Bq1, A);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) seq_post(cons_seq, ti + 1);                    // the image is read out
            F2_TICK(1);
            // lane (kk, n = 2 b + comp) holds component comp of outputs 16 b + 4 kk .. + 3; the other component sits in the
            // neighbouring lane: even lanes finish outputs 0, 1 of the four, odd lanes 2, 3
            v4f o;
#pragma unroll
            for (int k = 0; k < 4; k++)
                o[k] = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(y[k]), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
            const float re0 = odd ? o[2] : y[0], im0 = odd ? y[2] : o[0];
            const float re1 = odd ? o[3] : y[1], im1 = odd ? y[3] : o[1];
            const int qo = ti * WCOLS + 16 * (ni >> 1) + 4 * kk + 2 * (lane & 1);
            if (qo >= ja && qo < jb)
                zring[(zr0 + qo) & G.ring_mask] = make_float2(re0 * FS.gain_re - im0 * FS.gain_im, re0 * FS.gain_im + im0 * FS.gain_re);
            if (qo + 1 >= ja && qo + 1 < jb)
                zring[(zr0 + qo + 1) & G.ring_mask] = make_float2(re1 * FS.gain_re - im1 * FS.gain_im, re1 * FS.gain_im + im1 * FS.gain_re);
            F2_TICK(2);
        }
        if (dbg_on) for (int k = 0; k < 3; k++) B.dbg[(size_t)ch * DBG_SLOTS + 4 + k] += dbg_acc[k];
        return;
    }

    // ======================================================================= producer: load, scatter, DC removal, mix
    const char *__restrict__ inb = reinterpret_cast<const char *>(iq_raw) + (size_t)P.stream * G.stream_stride * BPS;
    const float2 *__restrict__ in = reinterpret_cast<const float2 *>(inb);       // FMT == 0
    const float qs = G.iq_scale;
    const bool dcr = P.dc_remove != 0;
    const int lo = P.lo_freq;
    const bool mix = (lo != 0) && (T.lo_table != nullptr);
    const int R = G.input_rate;
    const float alpha = 1.0f / (float)R;              // rfDcAlpha fm-processor.cpp:379
    const float Lg = P.att_l, Rg = P.att_r;
    const bool touch = (dcr || mix || Lg != 1.0f || Rg != 1.0f) && !(F2_ABL & 2);
    // a lane's sample PAIR is one 16 / 4 / 8 byte load when the buffer is aligned that far
    const bool aligned16 = ((g0 & 1) == 0) && ((G.stream_stride & 1) == 0) &&
                           ((reinterpret_cast<uintptr_t>(iq_raw) & (2 * BPS - 1)) == 0);
    auto cvt1 = [&](int i) -> float2 {                            // one sample at buffer index i (call-relative)
        if (FMT == 0) return in[i];
        if (FMT == 1) { const uint8_t *p = reinterpret_cast<const uint8_t *>(inb) + 2 * (size_t)i;
                        return make_float2((float)((int)p[0] - 127) * qs, (float)((int)p[1] - 127) * qs); }
        if (FMT == 2) { const int8_t *p = reinterpret_cast<const int8_t *>(inb) + 2 * (size_t)i;
                        return make_float2((float)p[0] * qs, (float)p[1] * qs); }
        const int16_t *p = reinterpret_cast<const int16_t *>(inb) + 2 * (size_t)i;
        return make_float2((float)p[0] * qs, (float)p[1] * qs);
    };
    // The DC recurrence r <- r + alpha (x - r) over a run of samples is the affine map r -> r (1 - u) + a; (u, a) are kept
    // instead of (1 - u, a): u ~ count * alpha is tiny, f32 holds it to 1e-7 relative (fmx_front.hip).  Full tiles: every lane's
    // run has the same u, so only the a parts are scanned (DPP: row_shr 1, 2, 4, 8, row_bcast 15, 31) with constant weights.
    float u_full = 0.f;
    for (int k = 0; k < SPT; k++) u_full = (1.0f - u_full) * alpha + u_full;
    const float m1 = 1.0f - u_full, m2 = m1 * m1, m4 = m2 * m2, m8 = m4 * m4;
    float u_exc = 0.f, u_tile = 0.f, mA = 1.f, mB = 1.f;
    for (int i = 0; i < 64; i++) {
        if (i < lane) u_exc = u_exc + u_full - u_exc * u_full;
        u_tile = u_tile + u_full - u_tile * u_full;
        if (i < (lane & 15) + 1) mA *= m1;
        if (i < (lane & 31) + 1) mB *= m1;
    }
    // Coalesced tile load: lane l, step k -> sample pair l + 64 k of the tile; after the scatter each lane reads back "its" two
    // columns (24 consecutive samples in time).
    int sc_idx[SPT / 2];                                          // float2 index of sample pair k's first sample
#pragma unroll
    for (int k = 0; k < SPT / 2; k++) {
        const int e = 2 * (lane + 64 * k);                        // sample index within the tile (even)
        const int c = e / 12, r = e - 12 * c;                     // r is even: the pair stays inside one column
        sc_idx[k] = xidx(r, HL + c);
    }
    auto load_tile = [&](int ti, float4 (&raw)[SPT / 2]) {
        const int wbase = ti * WSAMP;                             // index of the tile's first sample
        if (F2_ABL & 8) {
#pragma unroll
            for (int k = 0; k < SPT / 2; k++) raw[k] = make_float4(0.f, 1.f, 2.f, (float)ti);
        } else if (aligned16 && wbase >= g0 && wbase + WSAMP <= gend) {
            if (FMT == 0) {
                const v4f *p4 = reinterpret_cast<const v4f *>(in + (wbase - g0));
#pragma unroll
                for (int k = 0; k < SPT / 2; k++) {
                    const v4f v = __builtin_nontemporal_load(p4 + lane + 64 * k);     // streamed once: keep it out of the caches' way
                    raw[k] = make_float4(v.x, v.y, v.z, v.w);
                }
            } else if (FMT == 1 || FMT == 2) {
                const uint32_t *p1 = reinterpret_cast<const uint32_t *>(inb + (size_t)(wbase - g0) * BPS);
#pragma unroll
                for (int k = 0; k < SPT / 2; k++) {
                    const uint32_t w = p1[lane + 64 * k];             // I0 Q0 I1 Q1
                    if (FMT == 1)
                        raw[k] = make_float4((float)((int)(w & 255u) - 127) * qs, (float)((int)((w >> 8) & 255u) - 127) * qs,
                                             (float)((int)((w >> 16) & 255u) - 127) * qs, (float)((int)(w >> 24) - 127) * qs);
                    else
                        raw[k] = make_float4((float)(int8_t)(w & 255u) * qs, (float)(int8_t)((w >> 8) & 255u) * qs,
                                             (float)(int8_t)((w >> 16) & 255u) * qs, (float)(int8_t)(w >> 24) * qs);
                }
            } else {
                const uint2 *p2 = reinterpret_cast<const uint2 *>(inb + (size_t)(wbase - g0) * BPS);
#pragma unroll
                for (int k = 0; k < SPT / 2; k++) {
                    const uint2 w = p2[lane + 64 * k];                // (I0 Q0) (I1 Q1)
                    raw[k] = make_float4((float)(int16_t)(w.x & 0xffffu) * qs, (float)(int16_t)(w.x >> 16) * qs,
                                         (float)(int16_t)(w.y & 0xffffu) * qs, (float)(int16_t)(w.y >> 16) * qs);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < SPT / 2; k++) {
                const int i0 = wbase + 2 * (lane + 64 * k);
                const float2 a = (i0 >= g0 && i0 < gend) ? cvt1(i0 - g0) : make_float2(0.f, 0.f);
                const float2 b = (i0 + 1 >= g0 && i0 + 1 < gend) ? cvt1(i0 + 1 - g0) : make_float2(0.f, 0.f);
                raw[k] = make_float4(a.x, a.y, b.x, b.y);
            }
        }
    };
    const int dc_unit = (lane & 3) * XS4 + 3 + (lane >> 2);     // this lane's column pair (24 + 2 l, 24 + 2 l + 1), row 0
    const int h_unit = (lane & 3) * XS4 + (lane >> 2) - 13;     // the same pair as history of the next tile (lanes 52..63)

    // ---- history -> image 0: columns qa-24 .. qa-1 at C 0..23 (de-interleaved, as the DC pass leaves them), the partial
    //      column qa at C 24 raw like the fresh samples scattered next to it (its unit goes through the DC pass of lane 0)
    {
        float2 *X2 = reinterpret_cast<float2 *>(img0);
        float *Xf = reinterpret_cast<float *>(img0);
        for (int i = lane; i < DECIM * A_HIST_COLS; i += 64) {
            const int r = i / A_HIST_COLS, c = i - r * A_HIST_COLS;
            float2 v = hist[i];
            if (c == HL && r >= r0) v = make_float2(0.f, 0.f);
            if (c < HL) { Xf[xidx_d(r, c, 0)] = v.x; Xf[xidx_d(r, c, 1)] = v.y; }
            else X2[xidx(r, c)] = v;
        }
    }
#ifndef F2_PROD_PRIO
#define F2_PROD_PRIO 0
#endif
    if (F2_PROD_PRIO) __builtin_amdgcn_s_setprio(F2_PROD_PRIO);
    const bool dbg_on = (B.dbg != nullptr) && (lane == 0);               // fmx_debug_phase_cycles: slots 0..3 = producer
    unsigned long long dbg_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dbg_t = dbg_on ? clock64() : 0ull;
    float4 rawA[SPT / 2], rawB[SPT / 2];              // tiles ti (even / odd) in flight
    load_tile(0, rawA);
    if (NT > 1) load_tile(1, rawB);
    float c0 = dc0r, c1 = dc0i;                       // RfDC at the tile's first sample

    auto tile = [&](int ti, float4 (&raw)[SPT / 2]) {
        float4 *X4 = img0 + (ti & 1) * XUNITS, *Xo = img0 + ((ti + 1) & 1) * XUNITS;
        float2 *X2 = reinterpret_cast<float2 *>(X4);
        const int qt = ti * WCOLS;                    // first column of the tile
        const int wbase = qt * 12;
        // ---- scatter the raw samples into the image (free: the consumer finished tile ti - 2 before this tile's history
        //      was written, see below)
        if (F2_ABL & 4) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < SPT / 2; k++) acc += raw[k].x + raw[k].y + raw[k].z + raw[k].w;
            if (acc == 123.456f) X2[lane] = make_float2(acc, acc);
        } else {
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
        __builtin_amdgcn_wave_barrier();              // LDS operations of one wave complete in order
        F2_TICK(0);
        if (ti + 2 < NT) load_tile(ti + 2, raw);      // the registers are free: two tiles of loads stay in flight
        const int q = qt + 2 * lane;                  // this lane's first column
        const int base = q * 12;
        // fresh samples of this lane are rows [first, lastp1) of its 24
        int first = (base >= g0) ? 0 : ((g0 - base) < SPT ? (g0 - base) : SPT);
        int lastp1 = (base + SPT <= gend) ? SPT : ((gend - base) > 0 ? (gend - base) : 0);
        if (lastp1 < first) lastp1 = first;
        const bool wave_full = __all(first == 0 && lastp1 == SPT);

        // (scalar f32 throughout: packed f32 VALU operations issue badly beside the partner wave's matrix instructions)
        float xr[SPT], xi[SPT];
#pragma unroll
        for (int r = 0; r < DECIM; r++) {
            const float4 v = (F2_ABL & 4) ? make_float4(0.f, 0.f, 0.f, 0.f) : X4[dc_unit + r * XRS];
            xr[r] = v.x; xi[r] = v.y; xr[r + DECIM] = v.z; xi[r + DECIM] = v.w;
        }
#ifdef F2_SUBTICK
        if (dbg_on) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); F2_TICK(4); }
#endif
        if (touch) {
            // ---- RF DC removal (fm-processor.cpp:423-446): per-lane run, wave scan of the affine maps, then the reference's own
            //      f32 recurrence RfDC = (x - RfDC)*alpha + RfDC from the scanned prefix
            if (dcr) {
                // Full tiles take the recurrence to first order in alpha (= 4.3e-7; the terms dropped are alpha^2 k^2 |x| < 1e-10):
                // RfDC after sample k of the lane's run = d0 (1 - (k + 1) alpha) + alpha S_k with S_k the running sum of the run
                // and d0 the state in front of it -- sums instead of a 24-deep chain of dependent FMAs, which the one producer wave
                // of a SIMD has nobody to hide behind.  Partial tiles (a call's first / last) run the recurrence itself.
                float au = 0.f, ar = 0.f, ai = 0.f;
                float h0r = 0.f, h0i = 0.f;
                if (wave_full) {
                    float tr[4], tq[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        tr[j] = (xr[6 * j] + xr[6 * j + 1]) + (xr[6 * j + 2] + xr[6 * j + 3]) + (xr[6 * j + 4] + xr[6 * j + 5]);
                        tq[j] = (xi[6 * j] + xi[6 * j + 1]) + (xi[6 * j + 2] + xi[6 * j + 3]) + (xi[6 * j + 4] + xi[6 * j + 5]);
                    }
                    h0r = tr[0] + tr[1]; h0i = tq[0] + tq[1];
                    ar = alpha * (h0r + (tr[2] + tr[3])); ai = alpha * (h0i + (tq[2] + tq[3]));
                } else {
#pragma unroll
                    for (int k = 0; k < SPT; k++) {
                        if (k >= first && k < lastp1) {
                            au = (1.0f - au) * alpha + au;
                            ar = fmaf(xr[k] - ar, alpha, ar); ai = fmaf(xi[k] - ai, alpha, ai);
                        }
                    }
                }
                float pu, par, pai;                       // exclusive prefix within the tile
                float tu, tar, tai;                       // the whole tile's map
                if (wave_full) {
                    float sr = ar, si = ai;
#define FMX_SCAN_STEP(ctrl, rmask, mm)                                                                                       \
                    {                                                                                                        \
                        const float er = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sr), ctrl, rmask, 0xf, false)); \
                        const float ei = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(si), ctrl, rmask, 0xf, false)); \
                        sr = fmaf(er, mm, sr); si = fmaf(ei, mm, si);                                                        \
                    }
                    FMX_SCAN_STEP(0x111, 0xf, m1)
                    FMX_SCAN_STEP(0x112, 0xf, m2)
                    FMX_SCAN_STEP(0x114, 0xf, m4)
                    FMX_SCAN_STEP(0x118, 0xf, m8)
                    FMX_SCAN_STEP(0x142, 0xa, mA)
                    FMX_SCAN_STEP(0x143, 0xc, mB)
#undef FMX_SCAN_STEP
                    par = __shfl_up(sr, 1, 64); pai = __shfl_up(si, 1, 64);
                    if (lane == 0) { par = 0.f; pai = 0.f; }
                    pu = u_exc;
                    tu = u_tile;
                    tar = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sr), 63));
                    tai = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(si), 63));
                } else {
                    float iu = au, iar = ar, iai = ai;        // general inclusive scan (first / last tile of a call)
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const float ou = __shfl_up(iu, d, 64), oar = __shfl_up(iar, d, 64), oai = __shfl_up(iai, d, 64);
                        if (lane >= d) {                      // apply o, then inc
                            const float nu = ou + iu - ou * iu, nar = oar + iar - oar * iu, nai = oai + iai - oai * iu;
                            iu = nu; iar = nar; iai = nai;
                        }
                    }
                    pu = __shfl_up(iu, 1, 64); par = __shfl_up(iar, 1, 64); pai = __shfl_up(iai, 1, 64);
                    if (lane == 0) { pu = 0.f; par = 0.f; pai = 0.f; }
                    tu = __shfl(iu, 63, 64); tar = __shfl(iar, 63, 64); tai = __shfl(iai, 63, 64);
                }
                float rr = c0 - c0 * pu + par, ri = c1 - c1 * pu + pai;
#ifdef F2_SUBTICK
                if (dbg_on) { asm volatile("" :: "v"(rr), "v"(ri)); F2_TICK(5); }
#endif
                c0 = c0 - c0 * tu + tar; c1 = c1 - c1 * tu + tai;         // RfDC after this tile
                if (wave_full) {
                    float s0r = 0.f, s0i = 0.f, s1r = h0r, s1i = h0i;     // running sums of the two half runs
#pragma unroll
                    for (int k = 0; k < SPT / 2; k++) {
                        s0r += xr[k]; s0i += xi[k]; s1r += xr[k + 12]; s1i += xi[k + 12];
                        const float w0 = -(float)(k + 1) * alpha, w1 = -(float)(k + 13) * alpha;
                        const float d0r = fmaf(alpha, s0r, fmaf(w0, rr, rr)), d0i = fmaf(alpha, s0i, fmaf(w0, ri, ri));
                        const float d1r = fmaf(alpha, s1r, fmaf(w1, rr, rr)), d1i = fmaf(alpha, s1i, fmaf(w1, ri, ri));
                        xr[k] -= __builtin_amdgcn_fmed3f(d0r, -0.01f, 0.01f); xi[k] -= __builtin_amdgcn_fmed3f(d0i, -0.01f, 0.01f);   // DCRlimit :429-442
                        xr[k + 12] -= __builtin_amdgcn_fmed3f(d1r, -0.01f, 0.01f); xi[k + 12] -= __builtin_amdgcn_fmed3f(d1i, -0.01f, 0.01f);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < SPT; k++) {
                        if (k >= first && k < lastp1) {
                            rr = fmaf(xr[k] - rr, alpha, rr); ri = fmaf(xi[k] - ri, alpha, ri);
                            xr[k] -= __builtin_amdgcn_fmed3f(rr, -0.01f, 0.01f); xi[k] -= __builtin_amdgcn_fmed3f(ri, -0.01f, 0.01f);
                        }
                    }
                }
            }
            // ---- IQ balance + LO mix (fm-processor.cpp:462-466, oscillator.cpp:49-58)
            if (Lg != 1.0f || Rg != 1.0f) {
#pragma unroll
                for (int k = 0; k < SPT; k++) if (k >= first && k < lastp1) { xr[k] *= Lg; xi[k] *= Rg; }
            }
            if (mix && lastp1 > first) {
                // LOPhase after sample i (0-based within the call) = (P0 - (i+1)*lo) mod R
                if (lo_per > 0) {
                    int m = (int)((unsigned)(base + first - g0 + 1) % (unsigned)lo_per);
#pragma unroll
                    for (int k = 0; k < SPT; k++) {
                        if (k >= first && k < lastp1) {
                            const float2 w = sLO[m];
                            const float vr = xr[k], vi = xi[k];
                            xr[k] = vr * w.x - vi * w.y; xi[k] = vr * w.y + vi * w.x;
                            m = (m + 1 == lo_per) ? 0 : m + 1;
                        }
                    }
                } else {
                    const long long i1 = (long long)(base + first - g0) + 1;
                    const long long m = (i1 * (long long)lo) % (long long)R;
                    int ph = (int)(((long long)lo_phase0 - m) % (long long)R);
                    if (ph < 0) ph += R;
#pragma unroll
                    for (int k = 0; k < SPT; k++) {
                        if (k >= first && k < lastp1) {
                            const float2 w = T.lo_table[ph];
                            const float vr = xr[k], vi = xi[k];
                            xr[k] = vr * w.x - vi * w.y; xi[k] = vr * w.y + vi * w.x;
                            ph -= lo;
                            if (ph < 0) ph += R; else if (ph >= R) ph -= R;
                        }
                    }
                }
            }
        }
#ifdef F2_SUBTICK
        if (dbg_on) { asm volatile("" :: "v"(xr[23]), "v"(xi[11])); F2_TICK(6); }
#endif
        // ---- back to LDS in place, de-interleaved (entries that are not fresh pass through unchanged)
        if (!(F2_ABL & 4)) {
#pragma unroll
            for (int r = 0; r < DECIM; r++)
                X4[dc_unit + r * XRS] = make_float4(xr[r], xr[r + DECIM], xi[r], xi[r + DECIM]);
        }
        __builtin_amdgcn_wave_barrier();
        if (ti == NT - 1) {
            // ---- last tile: save history for the next call (columns qn-24 .. qn of the call, from this image)
            const float *Xf = reinterpret_cast<const float *>(X4);
            const int qn = gend / 12;                 // column of the next call's first sample
            const int cbase = qn - qt;                // image column of history slot 0 (= column qn-24)
            for (int i = lane; i < DECIM * A_HIST_COLS; i += 64) {
                const int r = i / A_HIST_COLS, c = i - r * A_HIST_COLS;
                const int lc = cbase + c;
                float2 v = make_float2(0.f, 0.f);
                if (lc >= 0 && lc < XCOLS) v = make_float2(Xf[xidx_d(r, lc, 0)], Xf[xidx_d(r, lc, 1)]);
                hist[i] = v;
            }
            if (lane == 0 && (dcr || dc_rst)) { st->dc_re = dcr ? c0 : dc0r; st->dc_im = dcr ? c1 : dc0i; }
            if (lane == 0) st->hist_fmt = 1;          // this layout keeps the history DC-corrected / mixed (fmx_front.hip converts)
            __builtin_amdgcn_wave_barrier();
        }
#ifdef F2_SUBTICK
        if (dbg_on) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); F2_TICK(7); }
#endif
        if (lane == 0) seq_post(prod_seq, ti + 1);
        F2_TICK(1);
        // ---- the 24 newest columns are the next tile's history: into the other image, once the consumer is done with tile
        //      ti - 1 there (which also frees that image for the next scatter)
        if (ti + 1 < NT) {
            if (ti >= 1) seq_wait(cons_seq, ti);
            F2_TICK(2);
            if (lane >= 52 && !(F2_ABL & 4)) {
#pragma unroll
                for (int r = 0; r < DECIM; r++)
                    Xo[h_unit + r * XRS] = make_float4(xr[r], xr[r + DECIM], xi[r], xi[r + DECIM]);
            }
            __builtin_amdgcn_wave_barrier();
            F2_TICK(3);
        }
    };
    for (int ti = 0; ti < NT; ti += 2) {
        tile(ti, rawA);
        if (ti + 1 < NT) tile(ti + 1, rawB);
    }
    if (dbg_on) { for (int k = 0; k < 4; k++) B.dbg[(size_t)ch * DBG_SLOTS + k] += dbg_acc[k]; for (int k = 4; k < 8; k++) B.dbg[(size_t)ch * DBG_SLOTS + 12 + k] += dbg_acc[k]; }
    if (lane == 0 && lo != 0) {
        long long m = ((long long)G.n * (long long)lo) % (long long)R;
        int ph = (int)(((long long)lo_phase0 - m) % (long long)R);
        if (ph < 0) ph += R;
        st->lo_phase = ph;
    }
}

}  // namespace f2

size_t front2_lds_bytes(int ntab, int lo_cap) {
    return sizeof(float4) * 4 * 2 * f2::XUNITS + sizeof(float) * (size_t)ntab * f2::MF_ADW + sizeof(float2) * 4 * (size_t)lo_cap + 8 * sizeof(int);
}

template <int FMT>
static hipError_t launch_front2_fmt(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, const void *iq, int channels,
                                    int ntab, int lo_cap, hipStream_t s) {
    const size_t lds = front2_lds_bytes(ntab, lo_cap);
    static size_t granted[64] = {};                               // per instantiation and device (function attributes are per device)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (lds > granted[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(f2::front2_kernel<FMT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        granted[dev] = lds;
    }
    hipLaunchKernelGGL(f2::front2_kernel<FMT>, dim3((channels + 3) / 4), dim3(512), lds, s, T, B, G, iq, channels, ntab, lo_cap);
    return hipSuccess;
}

hipError_t launch_front2(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, const void *iq, int channels,
                         int ntab, int lo_cap, hipStream_t s) {
    switch (G.iq_format) {
    case 1: return launch_front2_fmt<1>(T, B, G, iq, channels, ntab, lo_cap, s);
    case 2: return launch_front2_fmt<2>(T, B, G, iq, channels, ntab, lo_cap, s);
    case 3: return launch_front2_fmt<3>(T, B, G, iq, channels, ntab, lo_cap, s);
    default: return launch_front2_fmt<0>(T, B, G, iq, channels, ntab, lo_cap, s);
    }
}

}  // namespace fmx
