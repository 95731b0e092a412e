// fmx_demod.hip -- the demodulators with a recurrence of their own, in front of the fused stage-B kernel (fmx_stageb.hip).
// COMPILED WITH -ffp-contract=off: the LUT index expressions and feedback loops below are evaluated in exactly the f32 / f64 types the
// reference's C++ uses, because a one-ulp difference can flip a table index.
//
// Replaces, for the channels that select them:
//   fm_Demodulator::demodulate  PLL decoder   fm-demodulator.cpp:145-148 with pllC::do_pll pllC.cpp:67-90
//   fm_Demodulator::decodeAM                   fm-demodulator.cpp:215-241 (carrier level IIR :130-131, pllC for the AFC)
//   squelch::do_noise_squelch / do_level_squelch   squelchClass.cpp:47-113 (fm-processor.cpp:499-509)
//   and the AFC / scaling behind them           fm-demodulator.cpp:197-198
// pllC's phase feeds the look-up that corrects it and carries two states through a clamp; the noise squelch is two order-20 recursive
// filters: no parallelism in time is used for them.  Their parallelism is the channel count -- ONE LANE PER CHANNEL, 64 channels per
// wavefront, every lane walking the whole call sample by sample (afc_kernel), behind a time-parallel kernel that limits the samples and
// takes |z| (disc_kernel).  Both are plain kernels on the caller's stream, launched by launch_demod_fused in front of stageb_kernel when a
// channel needs them (launch_demod_prepass): no side streams, no events, no waiting kernels.  The demodulator output lands in the
// 16-row tiles of w_osc, where the fused kernel picks it up; every other channel is skipped here (the fused kernel demodulates it).
// Round 1 / 2 ran ALL of stage B on kernels of this kind (a five-stream chunk pipeline, then one persistent recurrence kernel with
// progress words): removed in round 3 -- the fused kernel is 2.5x faster and has no co-residency assumption.
#include "fmx_internal.h"
#include <type_traits>
#include "fmx_demod_math.h"

namespace fmx {

thread_local hipError_t g_launch_err = hipSuccess;

// =================================================================================================
// disc_kernel: limiter, |z|, memoryless discriminator   (time-parallel; 64 samples x 16 channels per block)
// =================================================================================================
constexpr int DISC_ROWS = 64, DISC_CH = 16;             // one block: four work-array tile rows of sixteen channels
__device__ __forceinline__ void disc_body(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int64_t row0, int nrows, const int bid_x, const int bid_y) {
    // One block = 64 samples x 16 channels: the ring reads are 528-byte runs per channel (the ring is channel-major), the
    // work-array stores 1 KB runs (16 rows x 16 channels of a tile row are contiguous).
    const int CP = G.pitch;
    __shared__ float2 sLIM[DISC_CH][DISC_ROWS + 3];   // limited samples of rows r0-2 .. r0+63 (each is used by up to three outputs)
    __shared__ float sABS[DISC_CH][DISC_ROWS + 3];    // |z| of the same samples (AM decoder, level squelch)
    __shared__ int sDelay[DISC_CH], sDec[DISC_CH], sSpecial;
    const int tid = threadIdx.x;
    if (tid == 0) sSpecial = 0;
    __syncthreads();
    const int64_t nj = row0 + nrows;
    const int64_t r0 = row0 + (int64_t)bid_x * DISC_ROWS;
    const int c0 = bid_y * DISC_CH;
    const int ring = G.ring_mask + 1;
    const bool want_iq = B.w_iq != nullptr;
    if (tid < DISC_CH) {
        const int ch = c0 + tid;
        sDelay[tid] = ch < C ? T.front_sets[B.params[ch].front_set].delay_fm : 0;
        sDec[tid] = ch < C ? B.params[ch].decoder : 0;
        if (ch < C && (B.params[ch].decoder <= 2 || B.params[ch].squelch_mode != 0)) sSpecial = 1;
    }
    __syncthreads();
    if (B.prepass && !sSpecial) return;              // (pre-pass of the fused layout: none of these sixteen channels needs it)
    // ---- limiter (fm-demodulator.cpp:119-126), once per sample: 66 consecutive ring entries per channel
    for (int i = tid; i < DISC_CH * (DISC_ROWS + 2); i += 256) {
        const int cl = i / (DISC_ROWS + 2), rl = i - (DISC_ROWS + 2) * cl;      // rl 0..65 <-> row r0 - 2 + rl
        const int ch = c0 + cl;
        float2 v = make_float2(0.f, 0.f); float za = 0.f;
        if (ch < C) {
            const float2 *zr = B.zring + (size_t)ch * ring;
            const int64_t jj = G.J0 + r0 - 2 + rl;
            // z[j'] is 0 before the filter latency has elapsed; the demodulator's initial Imin/Qmin is 0.01
            // (fm-demodulator.cpp:79-82)
            const bool am = sDec[cl] == 1;                   // the AM decoder works on the unlimited sample (fm-demodulator.cpp:133-134)
            if (jj < 0) v = am ? make_float2(0.f, 0.f) : make_float2((float)0.01, (float)0.01);
            else {
                const int64_t s = jj - sDelay[cl];
                const float2 z = s >= 0 ? zr[s & G.ring_mask] : make_float2(0.f, 0.f);
                // |z| once per sample: the limiter divides by it; the AM / PLL decoders and the level squelch read it from sABS
                za = (float)sqrt((double)z.x * (double)z.x + (double)z.y * (double)z.y);
                v = am ? z : ((double)za <= 0.001 ? make_float2((float)0.001, (float)0.001) : make_float2(z.x / za, z.y / za));   // limiter :119-126
            }
        }
        sLIM[cl][rl] = v; sABS[cl][rl] = za;
    }
    __syncthreads();
    // ---- discriminator; threads as (sample in tile row, channel), one tile row per step
    const int rl16 = tid & 15, cl = tid >> 4;
    const int ch = c0 + cl;
    const int decoder = sDec[cl];
#pragma unroll
    for (int i = 0; i < DISC_ROWS / WT; i++) {
        const int rl = WT * i + rl16;
        const int64_t r = r0 + rl;
        if (ch < C && r < nj && !(B.prepass && !(decoder <= 2 || B.params[ch].squelch_mode != 0))) {
            float res = 0.f;
            const float2 cur = sLIM[cl][rl + 2], p1 = sLIM[cl][rl + 1];
            const float I = cur.x, Q = cur.y, I1 = p1.x, Q1 = p1.y;
            if (decoder == 1 || decoder == 2) {   // AM: |z| for decodeAM (:215-241); PLL decoder: |z| for the carrier IIR (the
                res = sABS[cl][rl + 2];           // demod value itself comes from pllC).  Both run in afc_kernel.
            } else if (decoder == 5) {     // REAL_BB fm-demodulator.cpp:174-182
                res = (float)((double)(I1 * Q - Q1 * I + 1) / 2.0);
                int index = (int)floorf(res * (float)ARCSINE_N);
                if (index < 0) index = 0;
                if (index >= ARCSINE_N) index = ARCSINE_N;
                res = T.arcsine[index];
            } else if (decoder == 6) {     // DIFF :184-189
                const float2 p2 = sLIM[cl][rl];
                const float Scaler = (float)1.4142135623730951;
                res = (I1 * (Q - p2.y) - Q1 * (I - p2.x));
                res /= (I1 * I1 + Q1 * Q1) * Scaler;
            } else if (decoder != 2) {     // MIXED :168-172 (COMPLEX_BB :174-177 is bitwise the same)
                res = lut_atan2(T.atan_ppy, Q * I1 - I * Q1, I * I1 + Q * Q1);
            }
            B.w_dem[widx(r, ch, CP)] = res;
            if (want_iq) B.w_iq[widx(r, ch, CP)] = decoder <= 2 ? cur : make_float2(sABS[cl][rl + 2], 0.f);   // (other decoders: |z| for the level squelch)
        }
    }
}
__global__ __launch_bounds__(256) void disc_kernel(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int64_t row0, int nrows) {
    disc_body(T, B, G, C, row0, nrows, (int)blockIdx.x, (int)blockIdx.y);
}

// =================================================================================================
// afc_kernel: pllC (PLL / AM decoders), carrier level, AFC + scaling, level squelch   [lane per channel, 64 channels per wave]
//   fm-demodulator.cpp:130-131,145-148,197-198,215-241; pllC.cpp:67-90; squelchClass.cpp:47-113
// A dependent chain per sample whose time is set by instruction latency; the work-array rows are read a batch ahead into registers
// (global latency off the chain).
// =================================================================================================
// Batch size of the register-prefetched work-array rows.  A wave can have at most 63 vector-memory operations in flight
// (6-bit vmcnt) and on gfx9 stores count too: the loads of the next batch are issued right behind the stores of the
// last one, so (loads + stores) per batch must stay below that or every batch stalls for a store round trip.
constexpr int SEQ_UB = WT;
// a few wavefronts whose run time is pure instruction latency: the issue arbiter must not make them wait behind other kernels' waves
#ifndef FMX_RECURRENCE_PRIO
#define FMX_RECURRENCE_PRIO() __builtin_amdgcn_s_setprio(3)
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
// one work-array tile of this lane's channel (16 consecutive rows, 64 or 128 contiguous bytes) <-> registers
__device__ __forceinline__ void wld(float x[WT], const float *tile) {
    const f32x4 *q = reinterpret_cast<const f32x4 *>(tile);
#pragma unroll
    for (int i = 0; i < WT / 4; i++) {
        const f32x4 v = q[i];
        x[4 * i] = v[0]; x[4 * i + 1] = v[1]; x[4 * i + 2] = v[2]; x[4 * i + 3] = v[3];
    }
}
__device__ __forceinline__ void wld(int x[WT], const int *tile) {
    const i32x4 *q = reinterpret_cast<const i32x4 *>(tile);
#pragma unroll
    for (int i = 0; i < WT / 4; i++) {
        const i32x4 v = q[i];
        x[4 * i] = v[0]; x[4 * i + 1] = v[1]; x[4 * i + 2] = v[2]; x[4 * i + 3] = v[3];
    }
}
__device__ __forceinline__ void wst(float *tile, const float x[WT]) {
    f32x4 *q = reinterpret_cast<f32x4 *>(tile);
#pragma unroll
    for (int i = 0; i < WT / 4; i++) { f32x4 v; v[0] = x[4 * i]; v[1] = x[4 * i + 1]; v[2] = x[4 * i + 2]; v[3] = x[4 * i + 3]; q[i] = v; }
}
__device__ __forceinline__ void wst(int *tile, const int x[WT]) {
    i32x4 *q = reinterpret_cast<i32x4 *>(tile);
#pragma unroll
    for (int i = 0; i < WT / 4; i++) { i32x4 v; v[0] = x[4 * i]; v[1] = x[4 * i + 1]; v[2] = x[4 * i + 2]; v[3] = x[4 * i + 3]; q[i] = v; }
}
__device__ __forceinline__ void wld2(float2 x[WT], const float2 *tile) {
    const f32x4 *q = reinterpret_cast<const f32x4 *>(tile);
#pragma unroll
    for (int i = 0; i < WT / 2; i++) { const f32x4 v = q[i]; x[2 * i] = make_float2(v[0], v[1]); x[2 * i + 1] = make_float2(v[2], v[3]); }
}
__device__ __forceinline__ void wst2(float2 *tile, const float2 x[WT]) {
    f32x4 *q = reinterpret_cast<f32x4 *>(tile);
#pragma unroll
    for (int i = 0; i < WT / 2; i++) { f32x4 v; v[0] = x[2 * i].x; v[1] = x[2 * i].y; v[2] = x[2 * i + 1].x; v[3] = x[2 * i + 1].y; q[i] = v; }
}
// Software pipeline over the `nfull` whole tiles of a chunk.  A work-array tile written by the previous kernel comes
// from HBM / Infinity Cache with ~1.4 us latency and a recurrence wave has nothing else to run meanwhile, so the loads
// run far ahead of the compute: two register sets of PD tiles each, set B's loads are all issued before set A's 64
// samples are consumed and vice versa (block form rather than a rotating window: the compiler's s_waitcnt placement
// then leaves the full distance).  `load(set, u, tile)` fills registers, `body(set, u, tile)` consumes them; set and u
// are compile-time after unrolling.  Loads are clamped, never conditional.
// (depth per kernel: 4 for the one-array recurrences, 3 for the two-array ones so that the persistent kernel keeps two
// waves per SIMD, 2 for the PLL-decoder AFC)
template <int PD, typename LoadF, typename BodyF>
__device__ __forceinline__ void tile_pipeline(int nfull, LoadF load, BodyF body) {
    if (nfull <= 0) return;
    const int nlast = nfull - 1;
#pragma unroll
    for (int u = 0; u < PD; u++) load(0, u, u < nlast ? u : nlast);
    for (int b0 = 0; b0 < nfull; b0 += 2 * PD) {
#pragma unroll
        for (int u = 0; u < PD; u++) load(1, u, (b0 + PD + u < nlast) ? b0 + PD + u : nlast);
#pragma unroll
        for (int u = 0; u < PD; u++) if (b0 + u < nfull) body(0, u, b0 + u);
#pragma unroll
        for (int u = 0; u < PD; u++) load(0, u, (b0 + 2 * PD + u < nlast) ? b0 + 2 * PD + u : nlast);
#pragma unroll
        for (int u = 0; u < PD; u++) if (b0 + PD + u < nfull) body(1, u, b0 + PD + u);
    }
}

// ---- B2
// n / d as the compiler's IEEE division computes it (v_div_scale, v_rcp, the Newton steps, v_div_fmas, v_div_fixup) WITHOUT the scaling and the
// fix-up: the same eight operations on the same values, bit for bit, wherever v_div_scale leaves its operands alone -- d and n / d well inside the
// normal range (tools/ubench/fdiv_check.hip compares the two over 4e9 operand pairs of the range used here).  Used on the PLL decoder's
// LIMITED samples only: |conj (nco) sig| is 1 (or 0.0014 for a silent input), the larger component the denominator.
__device__ __forceinline__ float fdiv_inrange(float n, float d) {
    float y = __builtin_amdgcn_rcpf(d);
    const float e = __fmaf_rn(-d, y, 1.0f);
    y = __fmaf_rn(e, y, y);
    float q = n * y;
    float r = __fmaf_rn(-d, q, n);
    q = __fmaf_rn(r, y, q);
    r = __fmaf_rn(-d, q, n);
    return __fmaf_rn(r, y, q);
}
// compAtan::atan2 (lut_atan2 of fmx_demod_math.h, the same operations) for the loop of pllC: the corner arguments Xtan2.cpp:56-68 answers
// without its table -- a NaN, an infinity, x = 0 (and a denormal x, for the division's sake) -- are looked for with one v_cmp_class per
// argument and sent to the general form for the WHOLE WAVE; everything else takes the arm without them.
// `shadow`: work of the caller's that does not depend on this arc-tangent, issued right behind the table read -- a lone wave waits ~64 cycles for the LDS
// and has nothing else to fill them with (called exactly once, on either path).
struct NoShadow { __device__ __forceinline__ void operator()() const {} };
template <bool LIMITED, typename ShadowF = NoShadow>
__device__ __forceinline__ float lut_atan2_chain(const float *__restrict__ ppy, float y, float x, ShadowF shadow = ShadowF()) {
    // (x and y are the two components of conj (nco) * sample with a finite nco: an infinity or a NaN in y comes from one in the sample, and then x -- a sum
    // of products with BOTH of the sample's components -- is not finite either: x's class answers for y's)
    const bool odd = __builtin_amdgcn_classf(x, 0x2f7);
    if (__builtin_expect(__any(odd), 0)) { shadow(); return lut_atan2(ppy, y, x); }
    asm volatile("" : "+v"(x), "+v"(y));          // (keeps the general form's comparisons, which it shares with this arm, out of the path in front of the test)
    const float St = (float)3.14159265358979323846, Sh = St * 0.5f;
    const bool xpos = x > 0.f, ypos = y >= 0.f;
    const bool swap = !(fabsf(x) >= fabsf(y));
    const bool same = xpos == ypos;
    const float size = same ? (float)ATAN_N : -(float)ATAN_N;
    const float num = swap ? x : y, den = swap ? y : x;
    // (LIMITED: the quotient with ONE correction step, fdiv_fast -- the IEEE quotient but for rare half-way cases, and what is used is its integer part:
    // over 4.3e9 operand pairs of this range 3 quotients and NO index differ, tools/ubench/fdiv_check.hip)
    const float q = LIMITED ? fdiv_fast(size * num, den) : size * num / den;
    // (int)((double) q + 0.5), Xtan2.cpp:70-90, in two f32 operations: q + (0.5 - 2^-25) rounds into the same integer interval for EVERY f32 q in
    // [-0, 8192] -- all 1 174 405 121 of them compared, tools/ubench/atan_round_check.py (0.5 itself fails at q = 0.5 - 2^-25: the sum is a tie that rounds to 1)
    const int idx = (int)(q + 0.49999997f);
    const float tv = ppy[idx];
    if (!std::is_same<ShadowF, NoShadow>::value) {      // (the caller's work BETWEEN the read and its use: left to itself the compiler packs it behind the wait)
        __builtin_amdgcn_sched_barrier(0);
        shadow();
        __builtin_amdgcn_sched_barrier(0);
    }
    // (Sh * +-1, St * +-1 are exact; flat selects: a nested conditional becomes divergent branches here)
    const float ah = ypos ? Sh : -Sh, at = ypos ? St : -St;
    const float a0 = xpos ? 0.f : at;
    const float A = swap ? ah : a0;
    return A + ((same == swap) ? -tv : tv);
}
// What a handle's pre-pass channels use (DeviceBuffers::prepass_var, from the host's copy of the parameters): the kernel is compiled for the
// combinations below and a handle runs the smallest one that covers it -- a lone wave pays four cycles for EVERY instruction, the scalar
// bookkeeping of a feature nobody uses included.
enum { AV_PLL = 1, AV_AM = 2, AV_LSQ = 4, AV_MIXED = 8, AV_ALL = 15 };     // AV_MIXED: some pre-pass channel is on neither decoder (a squelch behind another one)
template <int VAR>
__device__ __forceinline__ void afc_body(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int chunk_len, const int bid_x, float *atan_lds) {
    constexpr bool HAS_PLL = (VAR & AV_PLL) != 0, HAS_AM = (VAR & AV_AM) != 0, HAS_LSQ = (VAR & AV_LSQ) != 0;
    constexpr bool HAS_CHAIN = HAS_PLL || HAS_AM;            // pllC runs (the PLL decoder on the limited sample, the AM decoder on the sample as it is)
    constexpr bool HAS_IQ = HAS_CHAIN || HAS_LSQ;            // the second work array is read (the sample; |z| for the level squelch of the other decoders)
    constexpr bool MIXED = (VAR & AV_MIXED) != 0 || (HAS_PLL && HAS_AM);     // (not MIXED: every lane that gets here is on the variant's one decoder)
    constexpr int PDA = 2;
    const int CP = G.pitch;
    const int ch = bid_x * 64 + threadIdx.x;
    const int decoder0 = ch < C ? B.params[ch].decoder : 3;
    if (HAS_CHAIN) {
        // pllC's loop is one dependent chain per sample and a lone wave waits out every latency on it: the two table look-ups of a step must
        // not be trips to memory (they were 2 x ~350 ns of the step's 900).  The arc-tangent table (32 KB, Xtan2.cpp:28-31) is copied into LDS by
        // the wave that has a lane on the PLL or AM decoder; the NCO's table entry comes from the sine unit (sincos_idx_hw, as in stage B).
        if (__any(decoder0 == 1 || decoder0 == 2)) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(T.atan_ppy);
            f32x4 *dst = reinterpret_cast<f32x4 *>(atan_lds);
            for (int i = threadIdx.x; i < ATAN_N / 4; i += 64) dst[i] = src[i];
            if (threadIdx.x == 0) atan_lds[ATAN_N] = T.atan_ppy[ATAN_N];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0);          // (one wave per workgroup: its own LDS stores are in order; the wait is for the compiler's sake)
    }
    if (ch >= C) return;
    ChanState *st = B.state + ch;
    const int decoder = decoder0;
    if (!(decoder <= 2 || B.params[ch].squelch_mode != 0)) return;      // (the fused kernel does this channel's demodulator itself)
    const bool use_pll = HAS_PLL && (!MIXED || decoder == 2), use_am = HAS_AM && (!MIXED || decoder == 1);
    // the sample behind which the reference takes its metaData snapshot (++myCount > fmRate / 2, fm-processor.cpp:662-684), row of this call
    const int64_t snap_row = (int64_t)(SINCOS_N >> 1) - (G.host_count1 ? G.host_count1 - 1 : st->my_count);
    // level squelch (squelch::do_level_squelch squelchClass.cpp:89-113, fm-processor.cpp:504-506): the carrier amplitude IIR
    // of the demodulator (fm-demodulator.cpp:130-131) against a threshold, re-evaluated every fmRate / 20 samples
    const bool lsq = HAS_LSQ && (B.params[ch].squelch_mode == 2);
    const float sq_thr = B.params[ch].squelch_thr;
    int sq_cnt = st->sq_count; bool sq_sup = st->sq_suppress != 0;
    bool sq_mute = lsq && sq_sup;
    float am = st->am_carr;
    // (the noise squelch, squelchClass.cpp:47-87, is a pass of its own behind this kernel: nsq_kernel below)
    const float fmDcAlpha = 0.0001f, c1 = 1 - fmDcAlpha, K = T.K_FM, rK = T.K_FM_rcp;
    const double SC = T.sincos_C;
    float afc = st->fm_afc, nco_phase = st->nco_phase, incr = st->phase_incr;
    const size_t ro = widx(0, ch, CP);
    float *wd = B.w_dem + ro;
    const float2 *wiq = HAS_IQ ? B.w_iq + ro : nullptr;
    // One sample.  A lone wave issues an instruction every 4-5 cycles whatever it is, and a divergent branch is half a dozen of them: the
    // step is written with selects; what only some waves of a handle need sits behind wave-uniform flags, what the handle does not use at all
    // is not compiled (VAR), and what the loop never reaches in practice (an NCO phase outside [0, 2 pi]: the update below cannot leave one;
    // the arc-tangent's corner arguments) behind a wave-uniform test.
    const bool any_pll = HAS_CHAIN && (!MIXED || __any(use_pll || use_am)), any_am = HAS_AM && (!MIXED || __any(use_am)), any_lsq = HAS_LSQ && __any(lsq);
    const bool on_pll = !MIXED || use_pll || use_am;
    const float beta = T.pll_beta, omb = 1 - T.pll_beta, plo = T.pll_lo, phi = T.pll_hi;
    float pce = T.pll_center;
    asm volatile("" : "+v"(pce));       // (kept in a vector register: the compiler moved it there from its scalar one at every sample)
    // (decide: a level-squelch decision may fall due at this sample -- one tile in 600 has one; the others are walked without the look-out)
    auto step = [&](float res, float2 sig, auto decide) __attribute__((always_inline)) -> float {
        float r_am = 0.f;
        // |z| arrives in the demod array for the AM and PLL decoders, in the first half of the IQ array otherwise (disc_kernel)
        if ((HAS_AM || HAS_LSQ) && (any_am || any_lsq)) {
            const float am2 = (1.0f - 0.0010f) * am + 0.0010f * (decoder <= 2 ? res : sig.x);      // am_carr_ampl, carrierAlpha (fm-demodulator.cpp:117,130-131)
            am = (use_am || lsq) ? am2 : am;
        }
        if (HAS_CHAIN && any_pll) {                               // pllC::do_pll pllC.cpp:67-90 (AM: on the unlimited sample, :222)
            // The NCO phase is in [0, 2 pi] (as f32: the update below leaves nothing else, starting from the constructor's 0) and the
            // loop's increment is limited to +-0.95 pi (NcoLLimit / NcoHLimit): SinCos::getComplex's table entry is one multiplication
            // away, and the reference's wrap loops (pllC.cpp:84-89) are at most one turn either way.
            // (idx = SINCOS_N -- a phase of exactly 6.2831855f, what a tiny negative phase wraps to -- needs no wrap of its own: sincos_idx_hw_bits folds it
            // to the first entry, (1, -0.0) for the table's (1, +0.0); the products of a zero differ in the sign of a zero at most, and a zero argument of
            // the arc-tangent goes to the general form, which does not look at its sign)
            const int idx = (int)((double)nco_phase * SC);
            float2 nco;
            sincos_idx_hw_bits(idx, &nco.y, &nco.x);
            const float dre = nco.x * sig.x - (-nco.y) * sig.y;      // conj(nco) * signal
            const float dim = nco.x * sig.y + (-nco.y) * sig.x;
            const float perr = lut_atan2_chain<!HAS_AM>(atan_lds, dim, dre);
            float inc2 = omb * perr + beta * incr;
            inc2 = (inc2 < plo || inc2 > phi) ? pce : inc2;
            const float ph = nco_phase + inc2;
            // pllC.cpp:84-89: ph >= 2 pi ((double) ph >= 2 pi  <=>  ph >= 6.2831855f) goes down a turn (x - 2 pi is fmod (x, 2 pi) for 2 pi <= x < 4 pi), ph < 0
            // up.  One unsigned comparison finds both -- a negative float's bits lie above every positive one's; ph is never -0.0: the phase never is -- and the
            // turn takes ph's sign: ph - copysign (2 pi, ph).
            const unsigned phb = __float_as_uint(ph);
            const bool wrapped = phb >= 0x40c90fdbu;                               // (the bits of 6.2831855f)
            // (the comparison here, in front of the conversions: its mask is read by the select behind them, and a select straight behind its comparison waits
            // two issue slots for the mask)
            asm volatile("" :: "s"(__builtin_amdgcn_ballot_w64(wrapped)));
            __builtin_amdgcn_sched_barrier(0);
            const double turn = __hiloint2double((int)(((unsigned)__double2hiint(FMX_2PI) & 0x7fffffffu) | (phb & ~0x7fffffffu)), __double2loint(FMX_2PI));
            const float moved = (float)((double)ph - turn);
            const float phw = wrapped ? moved : ph;
            incr = on_pll ? inc2 : incr;
            nco_phase = on_pll ? phw : nco_phase;
            if (HAS_AM && any_am) {                      // decodeAM fm-demodulator.cpp:215-241
                const float gainLimit = 0.01f;
                float r = (res - am) / (am < gainLimit ? gainLimit : am);
                r = (r > 1.0f) ? 1.0f : (r < -1.0f ? -1.0f : r);
                r_am = r;
            }
            if (HAS_PLL) res = use_pll ? incr : res;
        }
        afc = c1 * afc + fmDcAlpha * ((HAS_AM && use_am) ? incr : res);     // fm-demodulator.cpp:197 (AM: of the loop's increment, :232)
        float r = fdiv_const(20.0f * (res - afc) * 1.0f, K, rK);            // :198
        if (HAS_AM) r = use_am ? r_am : r;
        if (HAS_LSQ && any_lsq) {
            // (squelch::do_level_squelch squelchClass.cpp:89-113; a lane's decision falls due once in fmRate / 20 samples: the wave looks for
            // one with a single test per sample and takes the decisions behind it)
            sq_cnt += lsq ? 1 : 0;
            if (decltype(decide)::value) {
                const bool hold = lsq && sq_cnt >= SINCOS_N / 20;      // holdPeriod = fmRate / 20 (fm-processor.cpp:87)
                if (__any(hold)) {
                    const bool sup2 = am < sq_thr - 0.000f ? true : (am >= sq_thr + 0.000f ? false : sq_sup);   // SQUELCH_HYSTERESIS_LSQ = 0
                    sq_sup = hold ? sup2 : sq_sup;
                    sq_cnt = hold ? 0 : sq_cnt;
                    sq_mute = lsq && sq_sup;
                }
            }
            r = sq_mute ? r * 0.000f : r;                    // LEVELREDUCTIONFACTOR = 0
        }
        return r;
    };
    // The PLL decoder, or the AM decoder, alone on every lane (VAR == AV_PLL, AV_AM: the headlines of the pre-pass populations) as a software pipeline: the step above in two parts --
    // `chain`, pllC's loop (the next sample waits for it), and `tail`, AFC and scaling of the loop's increment (nothing waits for it) -- with the tail of
    // sample k - 1 issued in the shadow of sample k's table read, and beta * incr in front of the sine unit's latency instead of behind the arc-tangent.  The
    // same operations on the same values in the same order per variable: bit-identical to `step`.
    auto tail_am = [&](float res, float inc) __attribute__((always_inline)) -> float {      // decodeAM fm-demodulator.cpp:215-241, as in `step`
        // (the additions as instructions of their own, as in `tail`: packed with the arc-tangent's last one they take the divisions behind the wait with them)
        { const float t1 = (1.0f - 0.0010f) * am, t2 = 0.0010f * res; asm("v_add_f32 %0, %1, %2" : "=v"(am) : "v"(t1), "v"(t2)); }
        const float gainLimit = 0.01f;
        float r = (res - am) / (am < gainLimit ? gainLimit : am);
        r = (r > 1.0f) ? 1.0f : (r < -1.0f ? -1.0f : r);
        { const float t1 = c1 * afc, t2 = fmDcAlpha * inc; asm("v_add_f32 %0, %1, %2" : "=v"(afc) : "v"(t1), "v"(t2)); }      // :232
        return r;
    };
    auto tail = [&](float res) __attribute__((always_inline)) -> float {
        // fm-demodulator.cpp:197.  (the addition as an instruction of its own: left to the compiler it is packed with the arc-tangent's last one into a
        // v_pk_add_f32 -- behind the wait for the table -- and the whole tail follows it there)
        { const float t1 = c1 * afc, t2 = fmDcAlpha * res; asm("v_add_f32 %0, %1, %2" : "=v"(afc) : "v"(t1), "v"(t2)); }
        return fdiv_const(20.0f * (res - afc) * 1.0f, K, rK);               // :198
    };
    auto chain = [&](float2 sig, auto shadow) __attribute__((always_inline)) -> float {      // pllC::do_pll pllC.cpp:67-90, as in `step`
        const float bi = beta * incr;
        const int idx = (int)((double)nco_phase * SC);
        float2 nco;
        sincos_idx_hw_bits(idx, &nco.y, &nco.x);
        const float dre = nco.x * sig.x - (-nco.y) * sig.y;
        const float dim = nco.x * sig.y + (-nco.y) * sig.x;
        const float perr = lut_atan2_chain<!HAS_AM>(atan_lds, dim, dre, shadow);
        float inc2 = omb * perr + bi;
        inc2 = (inc2 < plo || inc2 > phi) ? pce : inc2;
        const float ph = nco_phase + inc2;
        const unsigned phb = __float_as_uint(ph);
        const bool wrapped = phb >= 0x40c90fdbu;
        asm volatile("" :: "s"(__builtin_amdgcn_ballot_w64(wrapped)));
        __builtin_amdgcn_sched_barrier(0);
        const double turn = __hiloint2double((int)(((unsigned)__double2hiint(FMX_2PI) & 0x7fffffffu) | (phb & ~0x7fffffffu)), __double2loint(FMX_2PI));
        const float moved = (float)((double)ph - turn);
        incr = inc2;
        nco_phase = wrapped ? moved : ph;
        return inc2;
    };
    // a run of samples straight from the work arrays, with the look-out for the metaData snapshot (get_demodDcComponent () behind sample
    // snap_row): the ragged end of a call, and the one tile in 750 the snapshot falls into
    auto slow_rows = [&](int k0, int k1) __attribute__((always_inline)) {
#pragma unroll 1
        for (int k = k0; k < k1; k++) {
            const float r = step(wd[(k / WT) * (WT * CP) + (k % WT)], HAS_IQ ? wiq[(k / WT) * (WT * CP) + (k % WT)] : make_float2(0.f, 0.f), std::true_type{});
            if ((int64_t)k == snap_row) st->meta_dc_if = afc;
            wd[(k / WT) * (WT * CP) + (k % WT)] = r;
        }
    };
    constexpr int UB = SEQ_UB;
    const int nfull = chunk_len / UB;
    const int TS = UB * CP;                                       // elements from one tile of this channel to the next
    // (prefetch registers as plain float arrays: loop-carried arrays of HIP's float2 struct end up in scratch memory)
    float nx[2][PDA][UB]; float nqa[HAS_IQ ? 2 : 1][HAS_IQ ? PDA : 1][UB], nqb[HAS_IQ ? 2 : 1][HAS_IQ ? PDA : 1][UB];
    tile_pipeline<PDA>(nfull,
        [&](int s, int u, int tl) __attribute__((always_inline)) {
            wld(nx[s][u], wd + tl * TS);
            if (HAS_IQ) { const float *qp = reinterpret_cast<const float *>(wiq + tl * TS); wld(nqa[HAS_IQ ? s : 0][HAS_IQ ? u : 0], qp); wld(nqb[HAS_IQ ? s : 0][HAS_IQ ? u : 0], qp + UB); }
        },
        [&](int s, int u, int tb) __attribute__((always_inline)) {
            // (the metaData snapshot falls into one tile in 750: looked for once per tile, for the whole wave, and that tile walked row by row)
            if (__any(snap_row >= (int64_t)tb * UB && snap_row < (int64_t)(tb + 1) * UB)) { slow_rows(tb * UB, (tb + 1) * UB); return; }
            float x[UB];
            auto samples = [&](auto decide) __attribute__((always_inline)) {
#pragma unroll
                for (int k = 0; k < UB; k++) {
                    const int ss = HAS_IQ ? s : 0, uu = HAS_IQ ? u : 0;
                    const float2 xq = !HAS_IQ ? make_float2(0.f, 0.f)
                                      : (k < UB / 2 ? make_float2(nqa[ss][uu][2 * k], nqa[ss][uu][2 * k + 1])
                                                    : make_float2(nqb[ss][uu][(2 * k) % UB], nqb[ss][uu][(2 * k) % UB + 1]));
                    x[k] = step(nx[s][u][k], xq, decide);
                }
            };
            if constexpr (VAR == AV_PLL || VAR == AV_AM) {
                float pend = 0.f;
                auto tail_k = [&](int k) __attribute__((always_inline)) { x[k] = (VAR == AV_AM) ? tail_am(nx[s][u][k], pend) : tail(pend); };
#pragma unroll
                for (int k = 0; k < UB; k++) {
                    const float2 xq = k < UB / 2 ? make_float2(nqa[s][u][2 * k], nqa[s][u][2 * k + 1]) : make_float2(nqb[s][u][(2 * k) % UB], nqb[s][u][(2 * k) % UB + 1]);
                    const float inc = (k == 0) ? chain(xq, NoShadow()) : chain(xq, [&]() __attribute__((always_inline)) { tail_k(k - 1); });
                    pend = inc;
                }
                tail_k(UB - 1);
                wst(wd + tb * TS, x);
                return;
            }
            // (the variant with everything in it keeps one form of the tile: its loop is 120 KB of code as it is)
            if (HAS_LSQ && (VAR == AV_ALL || (any_lsq && __any(lsq && sq_cnt + UB >= SINCOS_N / 20)))) samples(std::true_type{});
            else samples(std::false_type{});
            wst(wd + tb * TS, x);
        });
    slow_rows(nfull * UB, chunk_len);                             // ragged end of a call: rows of the last, partial tile
    st->fm_afc = afc; st->am_carr = am;
    if (HAS_CHAIN) { st->nco_phase = nco_phase; st->phase_incr = incr; }
    if (HAS_LSQ && lsq) { st->sq_count = sq_cnt; st->sq_suppress = sq_sup ? 1 : 0; }
}
template <int VAR>
__global__ __launch_bounds__(64) void afc_kernel(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int chunk_len) {
    FMX_RECURRENCE_PRIO();
    __shared__ __attribute__((aligned(16))) float atan_lds[(VAR & (AV_PLL | AV_AM)) ? ATAN_N + 4 : 4];
    afc_body<VAR>(T, B, G, C, chunk_len, (int)blockIdx.x, atan_lds);
}

// =================================================================================================
// nsq_kernel: squelch::do_noise_squelch (squelchClass.cpp:47-87) behind afc_kernel -- |high-pass 69.9 kHz| against |low-pass 70 kHz| of the
// demodulator output, two order-20 Chebyshev cascades of ten biquads each (Basic_IIR::Pass iir-filters.h:89-103), decaying averages,
// a decision every fmRate / 20 samples, the output muted while the squelch is closed.
// One lane per channel (rounds 2-3) walked twenty dependent biquads per sample with their forty memories in LDS: 40 ms per step at 4096
// channels.  A cascade is a PIPELINE: here one lane is ONE BIQUAD -- lanes 20 k .. 20 k + 9 the high-pass of the wave's k-th channel, 20 k +
// 10 .. 20 k + 19 its low-pass, three channels per wave -- with its two memories and four coefficients in registers; at step s lane i of
// a cascade works on sample s - i and hands its output one lane up (DPP wave_shr:1).  Every biquad does the reference's own f32
// operations in the reference's order on the reference's inputs: the results are the sequential ones bit for bit, the dependent chain per
// step is one biquad instead of twenty.  The cascades' last lanes keep the decaying averages (f64 expressions as decayingAverage
// computes them), meet at the decision points, and the low-pass's last lane writes the muted output back in place, nine samples behind.
// =================================================================================================
constexpr int NSQ_CH_PER_WAVE = 3, NSQ_LANES = 2 * NSQ_QUADS;
__global__ __launch_bounds__(64) void nsq_kernel(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int nrows) {
    const int lane = threadIdx.x;
    const int k = lane / NSQ_LANES, li = lane - k * NSQ_LANES;           // channel of the wave, lane of its pair of cascades
    const int ch = blockIdx.x * NSQ_CH_PER_WAVE + k;
    const bool mine = k < NSQ_CH_PER_WAVE && ch < C && B.params[ch < C ? ch : 0].squelch_mode == 1 && T.nsq_coef != nullptr;
    if (!__any(mine)) return;
    const int chs = mine ? ch : 0;
    const int f = li >= NSQ_QUADS ? 1 : 0, q = li - f * NSQ_QUADS;        // which filter (0 high-pass, 1 low-pass), which biquad of it
    ChanState *st = B.state + chs;
    const float *cf = T.nsq_coef + (f * NSQ_QUADS + q) * 4;
    const float c0 = cf[0], c1 = cf[1], c2 = cf[2], c3 = cf[3];
    const float gain = T.nsq_coef[2 * NSQ_QUADS * 4 + f];
    float m1 = mine ? st->sq_m[f][q][0] : 0.f, m2 = mine ? st->sq_m[f][q][1] : 0.f;
    const bool first = q == 0, last = q == NSQ_QUADS - 1;
    float avg = (f == 0) ? st->sq_avg_hi : st->sq_avg_lo;                   // (kept by the cascade's last lane)
    int sq_cnt = st->sq_count; bool sq_sup = st->sq_suppress != 0;
    const float thr = B.params[chs].squelch_nthr;
    const int CP = G.pitch;
    float *wd = B.w_dem + widx(0, chs, CP);                                 // (prepass: w_dem = w_osc, 16-row tiles; tile t of this channel at + t * 16 * CP)
    const int TS = WT * CP;
    const double k1 = 1.0 / (double)(float)(SINCOS_N / 100), k2 = 1.0 - k1;   // decayingAverage squelchClass.cpp:40-45, weight = sampleRate / 100, in double
    // The tiles come from HBM / Infinity Cache with ~1.4 us of latency and a pipeline wave has nothing else to run meanwhile: every lane that reads a
    // tile at all (the cascades' first lanes: the input; the low-pass's last lane: the values it mutes, nine samples behind) takes tile t + 2 while
    // the wave steps through tile t -- four register slots in rotation, the loop unrolled by four so that the slots are compile-time names.
    float xs[4][WT], xout[WT];
#pragma unroll
    for (int i = 0; i < WT; i++) { xs[0][i] = 0.f; xs[1][i] = 0.f; xs[2][i] = 0.f; xs[3][i] = 0.f; xout[i] = 0.f; }
    float o = 0.f;                                                         // this lane's output of the previous step
    const int nsteps = nrows + NSQ_QUADS - 1;
    const int ntiles = (nrows + WT - 1) / WT;
    // (clamped, never conditional -- a load behind a branch makes the compiler wait for every load in flight at the next use; the lanes that read
    // nothing load their channel's tile as well, into registers nobody looks at)
    auto take = [&](float *dst, int t) __attribute__((always_inline)) { wld(dst, wd + (t < ntiles ? t : ntiles - 1) * TS); };
    take(xs[0], 0); take(xs[1], 1);
    constexpr int LAG = NSQ_QUADS - 1;
    constexpr int HOLD = SINCOS_N / 20;
    // A tile in the steady state -- every lane of a cascade on a sample of the call, no decision due in any of the wave's channels (one in 600 tiles
    // has one) -- is walked without a test: the operations of the general form below, on every lane, the results of the lanes they do not concern
    // never looked at.  (The general form costs 115 instructions per step, most of them the exec-mask bookkeeping of its conditions, and a lone wave
    // issues one every four to five cycles: 3.9 ms per call; this form has 25.)
    auto fast_tile = [&](int t, const float *cur, const float *prev) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < WT; i++) {
            const float up = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(o), 0x138, 0xf, 0xf, false));
            const float in = first ? cur[i] * gain : up;
            const float w = in - m1 * c2 - m2 * c3;
            o = w + m1 * c0 + m2 * c1;
            m2 = m1; m1 = w;
            avg = (float)((double)fabsf(o) * k1 + (double)avg * k2);
            const int ti = (i + WT - LAG) & (WT - 1);
            const float xv = i >= LAG ? cur[ti] : prev[ti];
            xout[ti] = sq_sup ? xv * 0.000f : xv;
            if (ti == WT - 1 && mine && last && f == 1) wst(wd + (t - 1) * TS, xout);      // (i = 8: sample t * 16 - 1, the previous tile's last)
        }
        sq_cnt += WT;
    };
    // the general form: the pipeline's fill and drain, the tiles with a decision point.  Every lane of a channel counts the samples its cascades' last lanes
    // have taken and takes the channel's decisions with them.
    auto tile = [&](int t, const float *cur, const float *prev) __attribute__((always_inline)) {
        const int s0 = t * WT;
#pragma unroll
        for (int i = 0; i < WT; i++) {
            const int s = s0 + i;
            const int n = s - q;                                           // the sample this lane works on in this step
            const float up = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(o), 0x138, 0xf, 0xf, false));     // the lane below, previous step
            const bool act = mine && n >= 0 && n < nrows;
            if (act) {
                const float in = first ? cur[i] * gain : up;                // Basic_IIR::Pass: o = in * gain, then the quads
                const float w = in - m1 * c2 - m2 * c3;
                o = w + m1 * c0 + m2 * c1;
                m2 = m1; m1 = w;
            }
            const bool tail = act && last;
            if (tail) avg = (float)((double)fabsf(o) * k1 + (double)avg * k2);
            // the two cascades' last lanes work on the same sample in the same step; at a decision point (squelchClass.cpp:60-75: every fmRate / 20
            // samples) the channel's lanes fetch the two averages -- only then -- and take the decision
            const bool counted = mine && s - LAG >= 0 && s - LAG < nrows;
            bool hit = false;
            if (counted) { hit = ++sq_cnt >= HOLD; sq_cnt = hit ? 0 : sq_cnt; }
            if (__any(hit)) {
                const float avg_hi = __shfl(avg, k * NSQ_LANES + NSQ_QUADS - 1, 64), avg_lo = __shfl(avg, k * NSQ_LANES + 2 * NSQ_QUADS - 1, 64);
                if (hit) {
                    if (thr < 0.001f) sq_sup = true;                       // SQUELCH_HYSTERESIS_NSQ = 0.001
                    else if (avg_hi < avg_lo * thr - 0.001f) sq_sup = false;
                    else if (avg_hi >= avg_lo * thr + 0.001f) sq_sup = true;
                }
            }
            if (tail) {
                if (f == 1) {
                    // (sample n = s - 9 sits at place (i + 7) mod 16 of tile n / 16 -- a constant of the unrolled step: the first nine steps of a
                    // tile mute the previous tile's last nine values, xout collects the muted values; rows of the last tile behind the call's last
                    // sample are nobody's)
                    const int ti = (i + WT - LAG) & (WT - 1);
                    const float xv = i >= LAG ? cur[ti] : prev[ti];
                    const float r = sq_sup ? xv * 0.000f : xv;
                    xout[ti] = r;
                    if (ti == WT - 1 || n == nrows - 1) wst(wd + (n / WT) * TS, xout);
                }
            }
        }
    };
    auto walk = [&](int t, const float *cur, const float *prev, float *ahead) __attribute__((always_inline)) {
        take(ahead, t + 2);
        const bool steady = t >= 1 && (t + 1) * WT <= nrows && !__any(mine && sq_cnt + WT >= HOLD);
        if (steady) fast_tile(t, cur, prev); else tile(t, cur, prev);
    };
    for (int t = 0; t * WT < nsteps + WT; t += 4) {
        walk(t, xs[0], xs[3], xs[2]); walk(t + 1, xs[1], xs[0], xs[3]); walk(t + 2, xs[2], xs[1], xs[0]); walk(t + 3, xs[3], xs[2], xs[1]);
    }
    if (mine) { st->sq_m[f][q][0] = m1; st->sq_m[f][q][1] = m2; }
    if (mine && last) {
        if (f == 0) st->sq_avg_hi = avg; else { st->sq_avg_lo = avg; st->sq_count = sq_cnt; st->sq_suppress = sq_sup ? 1 : 0; }
    }
}

void launch_demod_prepass(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, int C, hipStream_t s_main, const PrepassStreams *ps) {
    const int64_t nj = G.J1 - G.J0;
    if (nj <= 0) return;
    DeviceBuffers Bp = B;
    Bp.prepass = 1; Bp.lin_rows = 0; Bp.w_dem = B.w_osc;          // (the tiled work arrays of the two kernels: w_osc takes the demodulator output)
    // (whole calls: everything on the caller's stream.  Pieces of an overlapping call: see PrepassStreams)
    hipStream_t s = ps ? ps->s_disc : s_main;
    hipLaunchKernelGGL(disc_kernel, dim3((unsigned)((nj + DISC_ROWS - 1) / DISC_ROWS), (unsigned)((C + DISC_CH - 1) / DISC_CH)), dim3(256), 0, s, T, Bp, G, C, (int64_t)0, (int)nj); FMX_LAUNCHED();
    if (ps) { note_hip(hipEventRecord(ps->ev_disc, ps->s_disc)); note_hip(hipStreamWaitEvent(ps->s_afc, ps->ev_disc, 0)); s = ps->s_afc; }
    {
        const dim3 ga((unsigned)((C + 63) / 64));
        switch (B.prepass_var) {          // (what the handle's channels use: fmx_api.hip)
        case 0:      hipLaunchKernelGGL(afc_kernel<0>, ga, dim3(64), 0, s, T, Bp, G, C, (int)nj); break;
        case AV_MIXED: hipLaunchKernelGGL(afc_kernel<0>, ga, dim3(64), 0, s, T, Bp, G, C, (int)nj); break;
        case AV_PLL: hipLaunchKernelGGL(afc_kernel<AV_PLL>, ga, dim3(64), 0, s, T, Bp, G, C, (int)nj); break;
        case AV_PLL | AV_MIXED: hipLaunchKernelGGL((afc_kernel<AV_PLL | AV_MIXED>), ga, dim3(64), 0, s, T, Bp, G, C, (int)nj); break;
        case AV_AM:  hipLaunchKernelGGL(afc_kernel<AV_AM>, ga, dim3(64), 0, s, T, Bp, G, C, (int)nj); break;
        case AV_LSQ: case AV_LSQ | AV_MIXED: hipLaunchKernelGGL(afc_kernel<AV_LSQ>, ga, dim3(64), 0, s, T, Bp, G, C, (int)nj); break;
        default:     hipLaunchKernelGGL(afc_kernel<AV_ALL>, ga, dim3(64), 0, s, T, Bp, G, C, (int)nj); break;
        }
        FMX_LAUNCHED();
    }
    if (ps) { note_hip(hipEventRecord(ps->ev_afc, ps->s_afc)); note_hip(hipStreamWaitEvent(s_main, ps->ev_afc, 0)); s = s_main; }
    if (T.nsq_coef) {        // (some channel has, or had, the noise squelch on: fmx_api.hip uploads the coefficients then)
        hipLaunchKernelGGL(nsq_kernel, dim3((unsigned)((C + NSQ_CH_PER_WAVE - 1) / NSQ_CH_PER_WAVE)), dim3(64), 0, s, T, Bp, G, C, (int)nj); FMX_LAUNCHED();
    }
}

}  // namespace fmx
