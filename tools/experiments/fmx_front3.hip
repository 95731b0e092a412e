// fmx_front3.hip -- stage A for the batches that fill the chip: the same arithmetic as front_kernel (fmx_front.hip), THREE waves per SIMD.
//
// Replaces, per channel and per call (as front_kernel does):
//   RF DC removal            fm-processor.cpp:423-446   (applied behind the filter, fmx_front.hip)
//   IQ balance               fm-processor.cpp:462-464
//   inputFilter (251 taps)   fm-processor.cpp:469-470, fft-filters.cpp:132-163
//   fmBand_1 (25 taps, /6)   fm-processor.cpp:472,  fir-filters.cpp:397-424
//   fmBand_2 (3 taps, /2)    fm-processor.cpp:474
//
// front_kernel is bound by the sum of a wave's serial phases at two waves per SIMD (253 VGPRs, 72 KB of LDS per workgroup).  This kernel is cut
// for three: <= 168 VGPRs and 12 KB of LDS per wave, 6 waves per channel, two channels per workgroup, one workgroup per CU (the same 512
// channel slots per chip).
//   * The LDS image of a wave holds the 128 fresh columns of its tile and nothing else (768 float4 units, no padding: the bank spread is a
//     rotation of the column group inside each 256-byte run).  The 24 history columns a tile's filter needs are READ IN PLACE from the image of
//     the wave that owns the previous tile: no history copy between images, no history columns in an image.
//   * FIR lane map: the four lanes of a quad take the four row quarters of eight adjacent outputs, so the quarter sums meet in two DPP
//     quad-permute additions instead of an LDS round trip, and the image is never overwritten by partial sums -- it stays readable for the
//     next tile's owner until its own next scatter.
//   * The window of a row slides through registers pair by pair (one ds_read_b128 per two taps, four taps ahead of its first use) instead of
//     sitting in 64 registers; tap values arrive the same way.
//   * Order within a tile: scatter, prefetch of the wave's next tile, FIR, THEN the RF DC pass (whose results only the outputs need), output.
//     With the DC pass behind the FIR a wave's image is free for its next scatter a quarter of a tile time after the successor has read its
//     history columns: the waves run staggered without waiting for each other.
//   * RfDC at the column boundaries stays in registers: the outputs take them from the lanes six to the left through ds_bpermute; the 13
//     boundaries in front of a tile travel with the DC carry in one LDS mailbox slot per tile.
// It handles what the headline runs -- float32 IQ, 16-byte aligned, no LO on any channel, the 287-tap fold (25 tap columns), calls that start
// on a column boundary -- in whole tiles of 1536 samples; launch_front gives a call's remainder, and every other case, to front_kernel.  The
// results are those of front_kernel bit for bit (same per-lane sums, same scan, same order of the filter's additions: tests/test_gpu_round5.py).
#include "fmx_internal.h"
#include "fmx_front_dc.h"
#include <utility>

namespace fmx {
namespace f3 {

constexpr int NW = 6;                          // waves (= tile images) per channel
constexpr int CPW = 2;                         // channels per workgroup: twelve waves, three per SIMD, ONE workgroup per CU -- two workgroups of six
                                               // waves do not reliably land on one CU together (measured: one resident, half the chip's wave slots empty)
constexpr int NTHR = 64 * NW * CPW;
constexpr int WCOLS = 128;                     // columns (= outputs) per tile
constexpr int WSAMP = WCOLS * DECIM;           // 1536 input samples per tile
constexpr int SPT = 2 * DECIM;                 // 24 samples per lane per tile (two adjacent columns)
constexpr int FCOLS = 8;                       // adjacent outputs per quad
constexpr int HL = A_HIST_COLS - 1;            // 24 history columns in front of a tile
constexpr int ND = A_MAX_ND;                   // 25 tap columns
constexpr int IMG_UNITS = DECIM * 64;          // 768 float4 = 12288 B per wave
constexpr int TROW = 28;                       // tap row stride (floats): rows 3 q + rr of the four quarters fall on distinct 16-byte bank slots (21 q mod 16)
constexpr int MB_N = 16;                       // mailbox slot: [0..12] RfDC in front of columns -13 .. -1 of the next tile, [13] in front of its column 0 (= the carry)
static_assert(ND == 25 && HL == 24 && FCOLS == 8, "the window schedule is worked out for this shape");

// LDS image of a tile: sample (row r = index mod 12, column C = 0..127).  The unit of storage is the float4 of the column pair (C even, C + 1) in
// row r; unit index = ((kp * 12 + r) * 16) + ((g + 4 (r / 3) + 4 kp) mod 16) with kp = (C mod 8) / 2 the pair slot and g = C / 8 the column group.
// ds_read_b128 is serviced in the 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32), i.e. quads {0,3,5,6} / {1,2,4,7} (+8):
//   FIR phase: lane = 4 cg + q reads (row 3 q + rr, fixed kp, group cg + const): slots cg + 4 q + const -- 16 distinct values per service group;
//   DC phase:  lane l reads (fixed r, kp = l mod 4, group l / 4): slots l / 4 + 4 (l mod 4) + const -- the same set.
__device__ __forceinline__ int unit3(int r, int C) {
    const int kp = (C & 7) >> 1, g = C >> 3;
    return ((kp * DECIM + r) << 4) + ((g + 4 * (r / 3) + 4 * kp) & 15);
}
__device__ __forceinline__ int idx3(int r, int C) { return 2 * unit3(r, C) + (C & 1); }   // float2 index

// ---- the FIR's load schedule, in taps (rows laid end to end: time = 25 rr + d).  Window pair m of a row = entries 2 m, 2 m + 1 (image columns
// 8 cg - 24 + entry); tap d reads entries 24 - d .. 31 - d, so pair m is first needed at tap max (0, 23 - 2 m) and dead behind tap 31 - 2 m.
constexpr int P_WIN = 4, P_TAP = 6;            // taps between a load and its first use
constexpr int T_PRO = -1;                      // everything due earlier is issued in front of the first tap
constexpr int pair_need(int rr, int m) { return 25 * rr + (23 - 2 * m > 0 ? 23 - 2 * m : 0); }
constexpr int pair_issue(int rr, int m) { const int t = pair_need(rr, m) - P_WIN - (m >= 12 ? m - 12 : 0); return t < T_PRO ? T_PRO : t; }
constexpr int tapq_issue(int rr, int d4) { const int t = 25 * rr + 4 * d4 - P_TAP; return t < T_PRO ? T_PRO : t; }

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ float dppq(float v, int ctrl_is_b1) {
    return ctrl_is_b1 ? __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false))      // quad_perm [1,0,3,2]
                      : __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]
}
__device__ __forceinline__ float bperm(int src_lane, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v))); }

#ifndef F3_ABL
#define F3_ABL 0      /* diagnostic builds only (tools/diag/f3_ablate.sh): bit 0 no scatter, 1 no DC pass, 2 no FIR, 3 no tile loads behind the first, 4 no waits
                         for other waves (the results are garbage) */
#endif
// (diagnostic builds, tools/build_variant.sh ... -DF3_TICKS: shader cycles of wave 0 of every channel by phase, into DeviceBuffers::dbg slots 48 ..)
#ifdef F3_TICKS
#define F3_TICK(k) do { if (dbg_on) { const unsigned long long now_ = __builtin_readcyclecounter(); dbg_acc[k] += now_ - dbg_t; dbg_t = now_; } } while (0)
#else
#define F3_TICK(k) do { } while (0)
#endif

template <bool NTL>
__global__ __launch_bounds__(NTHR, 3) __attribute__((amdgpu_waves_per_eu(3, 3))) void front3_kernel(DeviceTables T, DeviceBuffers B, CallGeom G,
                                                                                                       const float2 *__restrict__ iq) {
    struct ChanLds {
        float4 X[NW][IMG_UNITS];           // one image per wave (12288 B each)
        float sT[DECIM * TROW];            // the channel's tap set Trd[r][d], row stride 28
        float2 mb[8][MB_N];                // RfDC boundaries behind tile ti, slot = ti & 7
        int carry_seq;                     // tiles whose mailbox slot is published
        int scat_seq[NW], fir_seq[NW];     // per wave: tiles scattered / tiles whose filter has read everything, + 1
        int pad_[3];
    };
    __shared__ __attribute__((aligned(16))) ChanLds Lall[CPW];
    __shared__ __attribute__((aligned(16))) uint4 scT[SPT / 8][64];        // scatter indices (registers are short: read back once per tile)

    const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x / (64 * NW));
    const int ch_raw = (int)blockIdx.x * CPW + half;
    const bool active = ch_raw < G.channels;                               // (an odd channel count: the last workgroup's second half has nothing to do)
    const int ch = active ? ch_raw : G.channels - 1;
    const int t = (int)threadIdx.x - half * (64 * NW);                     // thread index within the channel's six waves
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    ChanLds &L = Lall[half];
    float4 (&Xall)[NW][IMG_UNITS] = L.X;
    float (&sT)[DECIM * TROW] = L.sT;
    float2 (&mb)[8][MB_N] = L.mb;
    int &carry_seq = L.carry_seq;
    int (&scat_seq)[NW] = L.scat_seq;
    int (&fir_seq)[NW] = L.fir_seq;
    float4 *X4 = Xall[wave];
    float2 *X2 = reinterpret_cast<float2 *>(X4);
    const ChanParams P = B.params[ch];
    const FrontSet FS = T.front_sets[P.front_set];
    const float2 *__restrict__ in = iq + (size_t)P.stream * G.stream_stride;
    ChanState *st = B.state + ch;
    float2 *hist = B.hist + (size_t)ch * DECIM * A_HIST_COLS;
    float2 *zring = B.zring + (size_t)ch * (G.ring_mask + 1);

    // Call-local geometry (front_kernel's, with the call starting on a column boundary and ending on a tile boundary)
    const int64_t qa = G.g0 / 12;
    const int NT = (int)(G.n / WSAMP);
    const int ja = (int)((G.g0 - FS.off + 11) / 12 - qa);           // first output completed by this call
    const int jb = (int)((G.g0 + G.n - FS.off + 11) / 12 - qa);     // one past the last
    const int zr0 = (int)((qa + FS.zshift) & (int64_t)G.ring_mask);

    for (int i = t; i < DECIM * TROW; i += 64 * NW) {
        const int r = i / TROW, d = i - r * TROW;
        sT[i] = T.front_taps[(size_t)P.front_set * A_TAPS_DEV + r * A_TAPS_ROW + d];
    }
    // scatter: sample pair k of the coalesced tile load (pair lane + 64 k of the tile) -> float2 indices of its two samples, packed
    if (threadIdx.x < 64) {
        unsigned sc[SPT / 2];
#pragma unroll
        for (int k = 0; k < SPT / 2; k++) {
            const int e = 2 * (t + 64 * k);
            const int c = e / 12, r = e - 12 * c;     // r is even: the pair stays inside one column
            sc[k] = (unsigned)idx3(r, c) | ((unsigned)idx3(r + 1, c) << 16);
        }
#pragma unroll
        for (int j = 0; j < SPT / 8; j++) scT[j][t] = make_uint4(sc[4 * j], sc[4 * j + 1], sc[4 * j + 2], sc[4 * j + 3]);
    }
    if (t == 0) { carry_seq = 0; for (int i = 0; i < NW; i++) { scat_seq[i] = 0; fir_seq[i] = 0; } }
    // ---- the call's history (raw samples, or what front_kernel's conversions make of them: see there) -> columns 104 .. 127 of the image of
    //      the wave in front of wave 0, where tile 0's filter looks for them
    const bool dc_rst = (P.actions & ACT_DC_RESET) != 0;           // setDCRemove zeroes RfDC (:922-925)
    const int hist_fmt0 = st->hist_fmt, lo_phase0 = st->lo_phase;
    const float st_dc_re = st->dc_re, st_dc_im = st->dc_im;
    const float2 R0 = (T.lo_table != nullptr && lo_phase0 != 0) ? T.lo_table[lo_phase0] : make_float2(1.f, 0.f);   // an oscillator set back to 0 Hz keeps its phase
    const bool hist_to_raw = (hist_fmt0 == 1);                     // the LO was switched off in front of this call
    const bool hist_rst = (hist_fmt0 == 0) && dc_rst;
    const bool dcr = P.dc_remove != 0;
    const float2 dc_now = (dc_rst || !dcr) ? make_float2(0.f, 0.f)
                                           : make_float2(__builtin_amdgcn_fmed3f(st_dc_re, -0.01f, 0.01f), __builtin_amdgcn_fmed3f(st_dc_im, -0.01f, 0.01f));
    const float2 *dcvR = B.dcv_hist + (size_t)ch * DCV_SAVE;
    if (wave == 0) {
        float2 *Xp = reinterpret_cast<float2 *>(Xall[NW - 1]);
        for (int i = lane; i < DECIM * A_HIST_COLS; i += 64) {
            const int r = i / A_HIST_COLS, c = i - r * A_HIST_COLS;
            if (c == HL) continue;                                 // (the partial column of a call that starts inside one: never here)
            float2 v = hist[i];
            if (hist_rst) {
                const int tb = c - HL + 13;
                const float2 d = dcvR[tb < 0 ? 0 : tb];
                v.x -= __builtin_amdgcn_fmed3f(d.x, -0.01f, 0.01f);
                v.y -= __builtin_amdgcn_fmed3f(d.y, -0.01f, 0.01f);
            } else if (hist_to_raw) {
                v = make_float2(v.x * R0.x + v.y * R0.y, v.y * R0.x - v.x * R0.y);
                v.x = (P.att_l != 0.f ? v.x / P.att_l : 0.f) + dc_now.x;
                v.y = (P.att_r != 0.f ? v.y / P.att_r : 0.f) + dc_now.y;
            }
            Xp[idx3(r, WCOLS - HL + c)] = v;
        }
    }
    // RfDC in front of the 13 columns before this call's first column and of that column itself
    if (t < 14) mb[7][t] = (hist_to_raw || hist_rst) ? make_float2(dc_rst ? 0.f : st_dc_re, dc_rst ? 0.f : st_dc_im) : dcvR[t];
    const float dc0r = dc_rst ? 0.f : st_dc_re, dc0i = dc_rst ? 0.f : st_dc_im;
    __syncthreads();                                  // the only workgroup barrier: tables, history and counters are set up

    const float cg_re = FS.gain_re * R0.x - FS.gain_im * R0.y, cg_im = FS.gain_re * R0.y + FS.gain_im * R0.x;     // complex output gain x R0
    const float alpha = 1.0f / (float)G.input_rate;   // rfDcAlpha fm-processor.cpp:379
    const float Lg = P.att_l, Rg = P.att_r;
    const float hsum = FS.hsum, dcw = FS.dc_w;
    const int dck = FS.dc_k;
    const DcK DK = dc_consts(alpha, lane);

    // ---- per-lane addresses
    const int cg = lane >> 2, q = lane & 3;           // FIR: outputs 8 cg .. 8 cg + 7, rows 3 q .. 3 q + 2
    const int pw = (wave + NW - 1) % NW, nw = (wave + 1) % NW;
    // window pair m = 4 jg + kp of row rr: column group cg + jg - 3 (of the previous tile's image when negative), pair slot kp; + rr * 16 units
    const float4 *vb[16];
#pragma unroll
    for (int jg = 0; jg < 4; jg++)
#pragma unroll
        for (int kp = 0; kp < 4; kp++) {
            const int g = cg + jg - 3;
            const float4 *img = (g < 0) ? Xall[pw] : X4;
            vb[4 * jg + kp] = img + (((kp * DECIM + 3 * q) << 4) + ((g + 4 * q + 4 * kp) & 15));
        }
    const float4 *tp = reinterpret_cast<const float4 *>(sT + 3 * q * TROW);
    // DC phase: the lane's column pair (2 l, 2 l + 1), rows 0 .. 11
    const int dcb = ((lane & 3) * DECIM) << 4, dcs = (lane >> 2) + 4 * (lane & 3);

    float4 raw[SPT / 2];
    auto load_tile = [&](int ti) {
        if (NTL) {
            typedef float v4f_ __attribute__((ext_vector_type(4)));
            const v4f_ *p4 = reinterpret_cast<const v4f_ *>(in + (size_t)ti * WSAMP);
#pragma unroll
            for (int k = 0; k < SPT / 2; k++) { const v4f_ v = __builtin_nontemporal_load(p4 + lane + 64 * k); raw[k] = make_float4(v.x, v.y, v.z, v.w); }
        } else {
            const float4 *p4 = reinterpret_cast<const float4 *>(in + (size_t)ti * WSAMP);
#pragma unroll
            for (int k = 0; k < SPT / 2; k++) raw[k] = p4[lane + 64 * k];
        }
    };
    const int NTa = active ? NT : 0;
#ifdef F3_TICKS
    const bool dbg_on = (B.dbg != nullptr) && (wave == 0);
    unsigned long long dbg_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long dbg_t = __builtin_readcyclecounter();
#endif
    if (wave < NTa) load_tile(wave);

    for (int ti = wave; ti < NTa; ti += NW) {
        const int qt = ti * WCOLS;                    // first column of the tile
        // ---- scatter the raw samples into the image, once the next tile's filter (the wave behind this one) has read its history columns from
        //      what the image held
        F3_TICK(9);
        if (!(F3_ABL & 16) && ti - NW + 1 >= 0) seq_wait(&fir_seq[nw], ti - NW + 2);
        F3_TICK(0);
        if (F3_ABL & 64) {
            // (diagnostic: the scatter of the f16-split form -- per sample pair two scalings, hi parts, remainders, lo parts, four 4-byte LDS writes)
            float *Xf = reinterpret_cast<float *>(X4);
#pragma unroll
            for (int k = 0; k < SPT / 2; k++) {
                const float4 r = raw[k];
                const float x0 = r.x * 4096.f, y0 = r.y * 4096.f, x1 = r.z * 4096.f, y1 = r.w * 4096.f;
                typedef __fp16 v2h __attribute__((ext_vector_type(2)));
                const v2h hr = __builtin_amdgcn_cvt_pkrtz(x0, x1), hi = __builtin_amdgcn_cvt_pkrtz(y0, y1);
                const float dx0 = x0 - (float)hr.x, dx1 = x1 - (float)hr.y, dy0 = y0 - (float)hi.x, dy1 = y1 - (float)hi.y;
                const v2h lr = __builtin_amdgcn_cvt_pkrtz(dx0, dx1), li = __builtin_amdgcn_cvt_pkrtz(dy0, dy1);
                Xf[lane + 64 * k] = __builtin_bit_cast(float, hr); Xf[768 + lane + 64 * k] = __builtin_bit_cast(float, hi);
                Xf[1536 + lane + 64 * k] = __builtin_bit_cast(float, lr); Xf[2304 + lane + 64 * k] = __builtin_bit_cast(float, li);
            }
        } else
        if (F3_ABL & 1) { asm volatile("" :: "v"(raw[0].x), "v"(raw[1].x), "v"(raw[2].x), "v"(raw[3].x), "v"(raw[4].x), "v"(raw[5].x), "v"(raw[6].x), "v"(raw[7].x), "v"(raw[8].x), "v"(raw[9].x), "v"(raw[10].x), "v"(raw[11].x)); }
        else
#pragma unroll
        for (int j = 0; j < SPT / 8; j++) {
            const uint4 s4 = scT[j][lane];
            const unsigned sc[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                X2[sc[k] & 0xffffu] = make_float2(raw[4 * j + k].x, raw[4 * j + k].y);
                X2[sc[k] >> 16] = make_float2(raw[4 * j + k].z, raw[4 * j + k].w);
            }
        }
        __builtin_amdgcn_wave_barrier();              // LDS operations of one wave complete in order
        if (lane == 0) seq_post(&scat_seq[wave], ti + 1);
        F3_TICK(1);
        // ---- prefetch this wave's next tile as soon as the registers are free: the loads are in flight for the whole iteration
        if (ti + NW < NT && !(F3_ABL & 8)) load_tile(ti + NW);
        // ---- the previous tile's newest 24 columns are this tile's history (tile 0: the call's, put there in front of the barrier)
        F3_TICK(2);
        if (!(F3_ABL & 16) && ti > 0) seq_wait(&scat_seq[pw], ti);
        F3_TICK(3);

        // ---- polyphase FIR  out[j] = sum_r sum_d Trd[r][d] * X[r][C_j - d]: the lane sums rows 3 q .. 3 q + 2 (in this order, taps 0 .. 24 each:
        //      front_kernel's order) for the quad's eight outputs
        v2f acc[FCOLS];
#pragma unroll
        for (int k = 0; k < FCOLS; k++) acc[k] = (v2f){0.f, 0.f};
        if (F3_ABL & 32) {
            // (diagnostic: the shape of an f16-split FIR on the matrix pipe -- 15 K-steps of two operand pairs and three v_mfma_f32_16x16x32_f16, then
            // 6 steps of column sums with two each; operands from the image as it is: the results are garbage)
            typedef _Float16 v8h __attribute__((ext_vector_type(8)));
            typedef float v4f_ __attribute__((ext_vector_type(4)));
            v4f_ am = (v4f_){0.f, 0.f, 0.f, 0.f}, as = am;
            const float4 *pa = reinterpret_cast<const float4 *>(sT) + (lane & 15);
            const float4 *pb = X4 + lane;
#pragma unroll
            for (int j = 0; j < 15; j++) {
                const float4 a0 = pa[(j & 3) * 16], a1 = pa[((j + 1) & 3) * 16], b0 = pb[j * 48], b1 = pb[j * 48 + 24];
                const v8h ah = __builtin_bit_cast(v8h, a0), al = __builtin_bit_cast(v8h, a1), bh = __builtin_bit_cast(v8h, b0), bl = __builtin_bit_cast(v8h, b1);
                am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, am, 0, 0, 0);
                am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, am, 0, 0, 0);
                am = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, am, 0, 0, 0);
                if (j >= 9) {
                    as = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, as, 0, 0, 0);
                    as = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bl, as, 0, 0, 0);
                }
            }
            acc[0] = (v2f){am.x + as.x, am.y + as.y}; acc[1] = (v2f){am.z + as.z, am.w + as.w};
        } else
        if (F3_ABL & 4) { acc[0] = (v2f){(float)cg, 1.f}; }
        else {
            float4 WP[3][16], TQ[3][7];
            static_for<3 * ND - T_PRO>([&](auto I) {
                constexpr int tt = decltype(I)::value + T_PRO;
                static_for<3>([&](auto RR) {
                    constexpr int rr = decltype(RR)::value;
                    static_for<16>([&](auto M) {
                        constexpr int m = 15 - decltype(M)::value;
                        if constexpr (pair_issue(rr, m) == tt) WP[rr][m] = vb[m][rr * 16];
                    });
                    static_for<7>([&](auto D4) {
                        constexpr int d4 = decltype(D4)::value;
                        if constexpr (tapq_issue(rr, d4) == tt) TQ[rr][d4] = tp[rr * (TROW / 4) + d4];
                    });
                });
                if constexpr (tt >= 0) {
                    constexpr int rr = tt / ND, d = tt - rr * ND;
                    const float4 tq = TQ[rr][d >> 2];
                    const float w1 = (d & 3) == 0 ? tq.x : ((d & 3) == 1 ? tq.y : ((d & 3) == 2 ? tq.z : tq.w));
                    const v2f w = (v2f){w1, w1};
                    static_for<FCOLS>([&](auto K) {
                        constexpr int k = decltype(K)::value;
                        constexpr int i = HL + k - d;
                        const float4 p = WP[rr][i >> 1];
                        const v2f x = (i & 1) ? (v2f){p.z, p.w} : (v2f){p.x, p.y};
                        acc[k] = __builtin_elementwise_fma(w, x, acc[k]);
                    });
                    // (pins the step here: its sums are used in another basic block, and the optimiser would sink the whole chain of FMAs there,
                    // behind all the loads)
                    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]));
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        __builtin_amdgcn_wave_barrier();
        F3_TICK(4);
        if (lane == 0) seq_post(&fir_seq[wave], ti + 1);           // the image in front of this one may take its owner's next tile
        // ---- the four quarter sums of an output: (s0 + s1) + (s2 + s3) in every lane of the quad
        float sel[4];
        {
            float s[2 * FCOLS];
#pragma unroll
            for (int k = 0; k < FCOLS; k++) { s[2 * k] = acc[k].x; s[2 * k + 1] = acc[k].y; }
#pragma unroll
            for (int k = 0; k < 2 * FCOLS; k++) s[k] += dppq(s[k], 1);
#pragma unroll
            for (int k = 0; k < 2 * FCOLS; k++) s[k] += dppq(s[k], 0);
            // outputs 2 l, 2 l + 1 of the tile = outputs 2 q, 2 q + 1 of the quad
#pragma unroll
            for (int k = 0; k < 4; k++) sel[k] = q == 0 ? s[k] : (q == 1 ? s[4 + k] : (q == 2 ? s[8 + k] : s[12 + k]));
        }
        float2 aA = make_float2(sel[0], sel[1]), aB = make_float2(sel[2], sel[3]);

        // ---- RF DC removal (fm-processor.cpp:423-446) behind the filter, as front_kernel does it for channels without an LO: the lane sums its
        //      two columns, the wave scans the affine maps, the state in front of the tile comes from the previous tile's mailbox slot
        float c_out_r = dc0r, c_out_i = dc0i;
        if (dcr && !(F3_ABL & 2)) {
            v2f x[SPT];
#pragma unroll
            for (int r = 0; r < DECIM; r++) {
                const float4 v = X4[dcb + (r << 4) + ((dcs + 4 * (r / 3)) & 15)];
                x[r] = (v2f){v.x, v.y}; x[r + DECIM] = (v2f){v.z, v.w};
            }
            const v2f al = (v2f){alpha, alpha};
            const DcMap DM = dc_tile_map(x, 0, SPT, true, true, DK, lane);
            // what the lanes at the tile's head need of the previous tile: RfDC in front of columns 2 lam, 2 lam + 1 for lam = A, A + 1 < 0
            const int col0 = 2 * lane - dck;                         // the outputs 2 l, 2 l + 1 take RfDC at columns col0 .. col0 + 2 (tile-relative)
            const int A = col0 >> 1;                                 // (arithmetic shift: floor)
            float c0 = dc0r, c1 = dc0i;
            const float2 *mp = mb[(ti - 1) & 7];
            F3_TICK(5);
            if (!(F3_ABL & 16) && ti > 0) seq_wait(&carry_seq, ti);
            F3_TICK(6);
            float2 pA0 = make_float2(0.f, 0.f), pA1 = pA0, pB0 = pA0, pB1 = pA0;
            if (A < 0) {
                const int iA = 13 + 2 * A, iB = 15 + 2 * A;          // entries of columns 2 A, 2 A + 1 and 2 A + 2, 2 A + 3
                pA0 = mp[iA < 0 ? 0 : iA]; pA1 = mp[iA + 1 < 0 ? 0 : iA + 1];
                if (A + 1 < 0) { pB0 = mp[iB]; pB1 = mp[iB + 1]; }
            }
            if (ti > 0) { const float2 cc = mp[13]; c0 = cc.x; c1 = cc.y; }
            c_out_r = dc_chain(c0, DM.tu, DM.tar); c_out_i = dc_chain(c1, DM.tu, DM.tai);
            const v2f rr = (v2f){c0 - c0 * DM.pre.u + DM.pre.ar, c1 - c1 * DM.pre.u + DM.pre.ai};      // RfDC in front of column 2 l ...
            const v2f b1 = __builtin_elementwise_fma(al, DM.sA, __builtin_elementwise_fma((v2f){-12.0f * alpha, -12.0f * alpha}, rr, rr));   // ... and of 2 l + 1
            // the next tile's slot: columns 115 .. 127 of this tile, then the state behind it
            float2 *mn = mb[ti & 7];
            if (lane >= 57) {
                const int e = 2 * (lane - 58) + 1;                   // entry of the lane's first column (116 + 2 (l - 58) - 115)
                if (e >= 0) mn[e] = make_float2(rr.x, rr.y);
                mn[e + 1] = make_float2(b1.x, b1.y);
            }
            if (lane == 0) mn[13] = make_float2(c_out_r, c_out_i);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) seq_post(&carry_seq, ti + 1);             // (every read of the previous slot is in front of this release)
            // RfDC at the three boundaries the lane's outputs interpolate between, from the lanes that own those columns
            const int la = A < 0 ? 0 : A, lb = A + 1 < 0 ? 0 : (A + 1 > 63 ? 63 : A + 1);
            float2 vA0 = make_float2(bperm(la, rr.x), bperm(la, rr.y)), vA1 = make_float2(bperm(la, b1.x), bperm(la, b1.y));
            float2 vB0 = make_float2(bperm(lb, rr.x), bperm(lb, rr.y)), vB1 = make_float2(bperm(lb, b1.x), bperm(lb, b1.y));
            if (A < 0) { vA0 = pA0; vA1 = pA1; }
            if (A + 1 < 0) { vB0 = pB0; vB1 = pB1; }
            const bool odd = (col0 & 1) != 0;
            const float2 e0 = odd ? vA1 : vA0, e1 = odd ? vB0 : vA1, e2 = odd ? vB1 : vB0;
            // what the FIR makes of the RfDC values the reference subtracts in front of it (limited to +-0.01, DCRlimit :429-442)
            const float dAr = fmaf(dcw, e1.x - e0.x, e0.x), dAi = fmaf(dcw, e1.y - e0.y, e0.y);
            const float dBr = fmaf(dcw, e2.x - e1.x, e1.x), dBi = fmaf(dcw, e2.y - e1.y, e1.y);
            aA.x = fmaf(-hsum, __builtin_amdgcn_fmed3f(dAr, -0.01f, 0.01f), aA.x); aA.y = fmaf(-hsum, __builtin_amdgcn_fmed3f(dAi, -0.01f, 0.01f), aA.y);
            aB.x = fmaf(-hsum, __builtin_amdgcn_fmed3f(dBr, -0.01f, 0.01f), aB.x); aB.y = fmaf(-hsum, __builtin_amdgcn_fmed3f(dBi, -0.01f, 0.01f), aB.y);
        }
        F3_TICK(7);
        // ---- IQ balance (:462-464), the decimators' complex gain, the fm-rate ring
        if (Lg != 1.0f || Rg != 1.0f) { aA.x *= Lg; aA.y *= Rg; aB.x *= Lg; aB.y *= Rg; }
        const float2 zA = make_float2(aA.x * cg_re - aA.y * cg_im, aA.x * cg_im + aA.y * cg_re);
        const float2 zB = make_float2(aB.x * cg_re - aB.y * cg_im, aB.x * cg_im + aB.y * cg_re);
        const int qc = qt + 2 * lane;                 // this lane's first output column
        const int zi = (zr0 + qc) & G.ring_mask;
        if ((zi & 1) == 0 && qc >= ja && qc + 1 < jb) {
            *reinterpret_cast<float4 *>(&zring[zi]) = make_float4(zA.x, zA.y, zB.x, zB.y);
        } else {
            if (qc >= ja && qc < jb) zring[(zr0 + qc) & G.ring_mask] = zA;
            if (qc + 1 >= ja && qc + 1 < jb) zring[(zr0 + qc + 1) & G.ring_mask] = zB;
        }
        F3_TICK(8);
        // ---- last tile: the state the next call finds (front_kernel's format: 24 raw columns and an empty partial one, the 14 newest boundaries)
        if (ti == NT - 1) {
            for (int i = lane; i < DECIM * A_HIST_COLS; i += 64) {
                const int r = i / A_HIST_COLS, c = i - r * A_HIST_COLS;
                hist[i] = (c < HL) ? X2[idx3(r, WCOLS - HL + c)] : make_float2(0.f, 0.f);
            }
            if (lane == 0 && (dcr || dc_rst)) { st->dc_re = c_out_r; st->dc_im = c_out_i; }
            if (lane == 0) st->hist_fmt = 0;
            __builtin_amdgcn_wave_barrier();
            if (lane < 14) B.dcv_hist[(size_t)ch * DCV_SAVE + lane] = dcr ? mb[ti & 7][lane] : make_float2(dc0r, dc0i);
        }
    }
#ifdef F3_TICKS
    if (dbg_on && active && lane == 0) for (int k = 0; k < 10; k++) B.dbg[(size_t)ch * DBG_SLOTS + 48 + k] += dbg_acc[k];
#endif
}

}  // namespace f3

// The calls front3_kernel takes (launch_front asks): whole tiles, on a column boundary, float32 samples 16-byte aligned.  The per-channel
// conditions -- no LO anywhere, every tap set the 25-column fold, one twin -- are the handle's (fmx_api.hip: front3_ok).
int front3_tiles(const CallGeom &G, const void *iq) {
    if (G.iq_format != 0 || G.twins != 1 || G.pre_processed || G.parts > 1) return 0;
    if ((G.g0 % DECIM) != 0 || (G.stream_stride & 1) != 0 || (reinterpret_cast<uintptr_t>(iq) & 15) != 0) return 0;
    return (int)(G.n / f3::WSAMP);
}
void launch_front3(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, const void *iq, int channels, hipStream_t s) {
    const dim3 grid((channels + f3::CPW - 1) / f3::CPW);
    if (G.streams_private) hipLaunchKernelGGL((f3::front3_kernel<true>), grid, dim3(f3::NTHR), 0, s, T, B, G, reinterpret_cast<const float2 *>(iq));
    else hipLaunchKernelGGL((f3::front3_kernel<false>), grid, dim3(f3::NTHR), 0, s, T, B, G, reinterpret_cast<const float2 *>(iq));
}

}  // namespace fmx
