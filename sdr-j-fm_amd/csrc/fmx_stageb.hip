// fmx_stageb.hip -- stage B time-parallel, one workgroup per channel, ONE kernel per call (or the same code as two, see stageb_kernel).
// COMPILED WITH -ffp-contract=off.
//
// Replaces per channel, like fmx_demod.hip:
//   fm_Demodulator::demodulate        fm-demodulator.cpp:111-205 (memoryless decoders: Mixed / ComplexBB / RealBB / Diff)
//   pilotRecovery::getPilotPhase      pilot-recover.cpp:54-83
//   process_signal_with_rds (stereo)  fm-processor.cpp:689-730
//   PerfectStereoSeparation           stereo-separation.cpp:60-109
//   L/R matrix, selector              fm-processor.cpp:517-549
//   de-emphasis                       fm-processor.cpp:594-595
//
// fmx_demod.hip runs the per-sample recurrences with one LANE per channel: their parallelism is the channel count, every
// sample costs a 180-cycle dependent chain, and five kinds of kernels hand chunks to each other through progress words.
// Here the parallelism is TIME: one 256-thread workgroup owns one channel and walks the call in segments of up to 1536 fm
// samples (six adjacent samples per thread); every recurrence becomes a scan over the workgroup (stageb_kernel below):
//   * linear recurrences with a constant decay (AFC, lock metric, PSS mean error, de-emphasis): a weighted wave scan in DPP
//     carries the state ACROSS threads, each thread then re-runs its own six samples in the reference's exact f32 / f64
//     expression.  The cross-thread carry differs from the sequential evaluation by rounding (1e-7 relative);
//   * the pilot PLL -- non-linear: the phase feeds the sine look-up that corrects it -- by Newton's method on the f32 trajectory of the
//     whole segment while the pilot is comfortably in lock, and on the reference's own sequential trajectory everywhere a lock decision
//     can fall (and everywhere for handles of few channels): found without a 1536-step chain by evaluating the segment's pilot periods
//     side by side from their anchors in [4, 8) -- where every f32 is a multiple of 2^-21, itself a multiple of every ulp of the phase --
//     and prefix-summing the integer misses of their end points until every run ends on the next run's start (see the kernel).  The step
//     evaluated is always the reference's own f32 step: its roundings are not noise but a pattern (adding the f32 omega to a phase in
//     [4, 8) adds omega rounded to that binade's grid), a frequency offset of ~1e-7 rad per sample that the loop turns into a standing
//     phase offset of ~5e-4 rad -- an evaluation in higher precision misses the reference by that much (tools/pll_fixed_point.py);
//   * the PSS phase integrator (whose increments are often smaller than half an ulp of the accumulator, so they must be
//     absorbed exactly as the reference absorbs them): increments evaluated at the segment's first value and summed in
//     f64, verified against the values found; a segment where that is not yet the trajectory iterates to the EXACT fixed
//     point, which IS the sequential f32 trajectory (induction over j), capped at PSS_MAX_ROUNDS;
//   * flags and counters (pilot lock, PSS call index, the "error minimised" state machine) follow from max / sum scans in
//     closed form; a segment in which a closed form does not apply (lock transitions inside a PSS segment, a counter next to
//     its 3 s threshold) is replayed sample by sample by one thread from LDS.
// The only feedback with a lag, the PSS error (low-pass of the 38 kHz mix, 1753 samples behind), bounds the segment: its
// errors come from s-ring entries that are at least one segment old.  Everything runs on the caller's stream: no side
// streams, no events, no waiting kernels.
#include "fmx_internal.h"
#include "fmx_demod_math.h"
#include "fmx_fftconv.h"
#include <type_traits>

namespace fmx {

#define SB_TICK(k) do { if (dbg_on) { const unsigned long long now_ = clock64(); dbg_acc[k] += now_ - dbg_t; dbg_t = now_; } } while (0)
// (diagnostic build SB_FINE_TICKS: cycles of thread 0 between ~40 points of a segment, accumulated in LDS; W = drain the memory counters first)
#ifdef SB_FINE_TICKS
#define SB_FT(k) do { if (ft_on) { const unsigned long long now_ = clock64(); atomicAdd(&ft_acc[k], now_ - ft_t); ft_t = now_; } } while (0)
#define SB_FTW(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); SB_FT(k); } while (0)
#else
#define SB_FT(k) do { } while (0)
#define SB_FTW(k) do { } while (0)
#endif

constexpr int FB_T = 256, FB_K = 6, FB_W = FB_T * FB_K;        // threads, samples per thread, segment length
static_assert(FB_T == fftc::T, "the convolution is written for the workgroup size");
static_assert(FB_W <= PSS_DELAY, "a segment's PSS errors must only need s-ring entries of earlier segments");
constexpr int PSS_MAX_ROUNDS = 64;
#ifndef SB_WG_PER_SIMD
#define SB_WG_PER_SIMD 3
#endif

// ---------------------------------------------------------------------------------------------------------------------
// wave scans in DPP: four steps inside the 16-lane rows (row_shr 1, 2, 4, 8), then the row totals ride row_bcast:15
// (into rows 1, 3) and row_bcast:31 (into rows 2, 3)
// ---------------------------------------------------------------------------------------------------------------------
template <int CTRL, int RM> __device__ __forceinline__ int dppi(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, RM, 0xf, false); }
// the same with ZERO where a lane has no source (bound_ctrl; every row enabled): no register to preset, and the move folds into the add / fma that follows
// (round 6: `old = 0` cost a v_mov_b32 per step and kept the compiler from folding -- 700 v_mov_b32_dpp and as many presets in the kernel's listing)
template <int CTRL> __device__ __forceinline__ int dppzi(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ float dppzf(float v) { return __int_as_float(dppzi<CTRL>(__float_as_int(v))); }
template <int CTRL> __device__ __forceinline__ double dppzd(double v) { return __hiloint2double(dppzi<CTRL>(__double2hiint(v)), dppzi<CTRL>(__double2loint(v))); }
template <int CTRL, int RM> __device__ __forceinline__ float dppf(float old, float v) { return __int_as_float(dppi<CTRL, RM>(__float_as_int(old), __float_as_int(v))); }
template <int CTRL, int RM> __device__ __forceinline__ double dppd(double old, double v) {
    const int lo = dppi<CTRL, RM>(__double2loint(old), __double2loint(v));
    const int hi = dppi<CTRL, RM>(__double2hiint(old), __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wscan_add_d(double v) {
    v += dppzd<0x111>(v); v += dppzd<0x112>(v); v += dppzd<0x114>(v); v += dppzd<0x118>(v);
    v += dppd<0x142, 0xa>(0.0, v); v += dppd<0x143, 0xc>(0.0, v);
    return v;
}
__device__ __forceinline__ float wscan_add_f(float v) {
    v += dppzf<0x111>(v); v += dppzf<0x112>(v); v += dppzf<0x114>(v); v += dppzf<0x118>(v);
    v += dppf<0x142, 0xa>(0.f, v); v += dppf<0x143, 0xc>(0.f, v);
    return v;
}
__device__ __forceinline__ int wscan_add_i(int v) {
    v += dppzi<0x111>(v); v += dppzi<0x112>(v); v += dppzi<0x114>(v); v += dppzi<0x118>(v);
    v += dppi<0x142, 0xa>(0, v); v += dppi<0x143, 0xc>(0, v);
    return v;
}
constexpr int IMIN = -0x7fffffff - 1;
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int wscan_max_i(int v) {
    v = imax(v, dppi<0x111, 0xf>(IMIN, v)); v = imax(v, dppi<0x112, 0xf>(IMIN, v)); v = imax(v, dppi<0x114, 0xf>(IMIN, v));
    v = imax(v, dppi<0x118, 0xf>(IMIN, v)); v = imax(v, dppi<0x142, 0xa>(IMIN, v)); v = imax(v, dppi<0x143, 0xc>(IMIN, v));
    return v;
}
// value of the previous lane (wave_shr:1), `id` for lane 0
__device__ __forceinline__ float lane_prev_f(float v, float id) { return dppf<0x138, 0xf>(id, v); }
__device__ __forceinline__ int lane_prev_i(int v, int id) { return dppi<0x138, 0xf>(id, v); }
__device__ __forceinline__ double lane_prev_d(double v, double id) { return dppd<0x138, 0xf>(id, v); }

__device__ __forceinline__ double powi(double b, int n) { double r = 1.0; while (n) { if (n & 1) r *= b; b *= b; n >>= 1; } return r; }

// weights of a scan of y[j] = d y[j-1] + u[j] over threads that own FB_K samples each (D = d^FB_K per thread)
struct DecayW { float m1, m2, m4, m8, mA, mB, dl, d64; };
// d^n = exp2 (n log2 d) with log2 d computed in f64 on the host from the reference's own f32 expression of d (the decay raised
// to 1536 must not carry the 3e-8 of a rounded base); v_exp_f32 is good to an ulp, like the f32 the weights are stored in
__device__ __forceinline__ DecayW make_decay(float l2, int lane) {
    DecayW w;
    const float L = l2 * (float)FB_K;                            // log2 of D = d^FB_K
    w.m1 = __builtin_amdgcn_exp2f(L); w.m2 = __builtin_amdgcn_exp2f(2 * L); w.m4 = __builtin_amdgcn_exp2f(4 * L); w.m8 = __builtin_amdgcn_exp2f(8 * L);
    w.mA = __builtin_amdgcn_exp2f((float)((lane & 15) + 1) * L); w.mB = __builtin_amdgcn_exp2f((float)((lane & 31) + 1) * L);
    w.dl = __builtin_amdgcn_exp2f((float)lane * L); w.d64 = __builtin_amdgcn_exp2f(64 * L);
    return w;
}
// The weights of the kernel's four one-pole scans (AFC, lock metric, PSS mean error, de-emphasis) per lane, in LDS: computed once per
// launch, read back with two 16-byte LDS reads where a scan needs them.  (Kept in registers they would occupy 8 VGPRs per scan for the
// whole kernel; recomputed per segment -- what was done before -- they cost eight quarter-rate v_exp_f32 and as many multiplies per scan
// and segment: 5 % of the first kernel's VALU time.)
enum { DEC_AFC = 0, DEC_LOCK = 1, DEC_PSSMEAN = 2, DEC_DEEMPH = 3 };
template <int NSCANS> struct DecayTab { float4 w[NSCANS][64][2]; };      // (a half of the kernel holds its own two scans: which & 1)
template <int NSCANS> __device__ __forceinline__ void store_decay(DecayTab<NSCANS> *tab, int which, float l2, int lane) {
    const DecayW w = make_decay(l2, lane);
    tab->w[which % NSCANS][lane][0] = make_float4(w.m1, w.m2, w.m4, w.m8);
    tab->w[which % NSCANS][lane][1] = make_float4(w.mA, w.mB, w.dl, w.d64);
}
template <int NSCANS> __device__ __forceinline__ DecayW load_decay(const DecayTab<NSCANS> *tab, int which, int lane) {
    const float4 a = tab->w[which % NSCANS][lane][0], b = tab->w[which % NSCANS][lane][1];
    DecayW w; w.m1 = a.x; w.m2 = a.y; w.m4 = a.z; w.m8 = a.w; w.mA = b.x; w.mB = b.y; w.dl = b.z; w.d64 = b.w;
    return w;
}
// inclusive over the wave: Z[t] = D Z[t-1] + L[t]
__device__ __forceinline__ float wscan_decay(float v, const DecayW &w) {
    v = fmaf(dppzf<0x111>(v), w.m1, v); v = fmaf(dppzf<0x112>(v), w.m2, v);
    v = fmaf(dppzf<0x114>(v), w.m4, v); v = fmaf(dppzf<0x118>(v), w.m8, v);
    v = fmaf(dppf<0x142, 0xa>(0.f, v), w.mA, v); v = fmaf(dppf<0x143, 0xc>(0.f, v), w.mB, v);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS of the segment kernel.  Every workgroup-wide step writes one slot of a two-slot scratch and meets at ONE barrier:
// a thread can only write slot k again after passing the barrier of the step in between, which every thread reaches only
// after it has read slot k.
// ---------------------------------------------------------------------------------------------------------------------
struct ScanLds {
    double wd[2][4][2];          // per wave: f64 totals
    float  wf[2][4][4];          // per wave: f32 totals
    int    wi[2][4][4];          // per wave: int totals / flags
    float  edge[FB_T];           // a value every thread hands to its right neighbour
};



// Workgroup-wide pieces.  `sl` is the scratch slot, toggled by every call; all threads of the workgroup make the same calls.
struct WG {
    ScanLds *L; int tid, lane, wv, sl;
    // y[j] = d y[j-1] + u[j]: the value in front of this thread's first sample, given this thread's run from zero `Lt`
    // (its contribution at its own last sample) and the state Y0 in front of the segment
    __device__ __forceinline__ float decay_incoming(float Lt, float Y0, const DecayW &w) {
        const float Z = wscan_decay(Lt, w);
        if (lane == 63) L->wf[sl][wv][0] = Z;
        __syncthreads();
        float C = Y0;                                     // value at the end of the previous wave, Y0 folded in
        for (int v = 0; v < wv; v++) C = fmaf(C, w.d64, L->wf[sl][v][0]);
        sl ^= 1;
        return fmaf(w.dl, C, lane_prev_f(Z, 0.f));
    }
    // the same, and the value behind a FULL segment (what the next segment of the same workgroup starts from)
    __device__ __forceinline__ float decay_incoming2(float Lt, float Y0, const DecayW &w, float *Yend) {
        const float Z = wscan_decay(Lt, w);
        if (lane == 63) L->wf[sl][wv][0] = Z;
        __syncthreads();
        float C = Y0, Cin = Y0;
#pragma unroll
        for (int v = 0; v < 4; v++) { C = fmaf(C, w.d64, L->wf[sl][v][0]); Cin = (v + 1 == wv) ? C : Cin; }
        sl ^= 1;
        *Yend = C;
        return fmaf(w.dl, Cin, lane_prev_f(Z, 0.f));
    }
    // exclusive prefix of an f64 sum over the threads (+ the workgroup total), with an OR-reduction of a flag riding along
    __device__ __forceinline__ double excl_add_d(double tot, double *total, bool flag, bool *any_flag) {
        const double inc = wscan_add_d(tot);
        const bool wany = __any(flag) != 0;
        if (lane == 63) L->wd[sl][wv][0] = inc;
        if (lane == 0) L->wi[sl][wv][0] = wany ? 1 : 0;
        __syncthreads();
        double pre = 0.0, all = 0.0; int fl = 0;
#pragma unroll
        for (int v = 0; v < 4; v++) { const double t = L->wd[sl][v][0]; all += t; pre += (v < wv) ? t : 0.0; fl |= L->wi[sl][v][0]; }
        sl ^= 1;
        *total = all; *any_flag = fl != 0;
        return pre + (inc - tot);
    }
    // exclusive prefix of an f32 sum over the threads (+ the workgroup total)
    __device__ __forceinline__ float excl_add_f(float tot, float *total) {
        const float inc = wscan_add_f(tot);
        if (lane == 63) L->wf[sl][wv][0] = inc;
        __syncthreads();
        float pre = 0.f, all = 0.f;
#pragma unroll
        for (int v = 0; v < 4; v++) { const float t = L->wf[sl][v][0]; all += t; pre += (v < wv) ? t : 0.f; }
        sl ^= 1;
        *total = all;
        return pre + lane_prev_f(inc, 0.f);
    }
    // exclusive prefixes of an int sum and an int max over the threads, with both totals
    __device__ __forceinline__ void excl_add_max_i(int a, int m, int *pre_a, int *tot_a, int *pre_m, int *tot_m) {
        const int ia = wscan_add_i(a), im = wscan_max_i(m);
        if (lane == 63) { L->wi[sl][wv][0] = ia; L->wi[sl][wv][1] = im; }
        __syncthreads();
        int pa = 0, ta = 0, pm = -0x7fffffff - 1, tm = -0x7fffffff - 1;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int xa = L->wi[sl][v][0], xm = L->wi[sl][v][1];
            ta += xa; tm = xm > tm ? xm : tm;
            if (v < wv) { pa += xa; pm = xm > pm ? xm : pm; }
        }
        sl ^= 1;
        const int ea = ia - a, em = lane_prev_i(im, -0x7fffffff - 1);
        *pre_a = pa + ea; *tot_a = ta; *pre_m = em > pm ? em : pm; *tot_m = tm;
    }
    // max-reduction of four ints over the workgroup
    __device__ __forceinline__ void reduce_max4(int &a, int &b, int &c, int &d) {
        // (the wave's maxima in lane 63 by DPP moves on the vector pipe -- 24 ds_bpermute per thread and segment before, a fifth of the second
        // kernel's LDS instructions)
        a = wscan_max_i(a); b = wscan_max_i(b); c = wscan_max_i(c); d = wscan_max_i(d);
        if (lane == 63) { L->wi[sl][wv][0] = a; L->wi[sl][wv][1] = b; L->wi[sl][wv][2] = c; L->wi[sl][wv][3] = d; }
        __syncthreads();
        a = b = c = d = IMIN;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int xa = L->wi[sl][v][0], xb = L->wi[sl][v][1], xc = L->wi[sl][v][2], xd = L->wi[sl][v][3];
            a = xa > a ? xa : a; b = xb > b ? xb : b; c = xc > c ? xc : c; d = xd > d ? xd : d;
        }
        sl ^= 1;
    }
    // the value the left neighbour hands over (`first` for thread 0) and the last thread's value
    __device__ __forceinline__ float from_left2(float mine, float first, float *last) {
        L->edge[tid] = mine;
        __syncthreads();
        const float v = tid ? L->edge[tid - 1] : first;
        *last = L->edge[FB_T - 1];
        __syncthreads();
        return v;
    }
    // the value the left neighbour hands over (`first` for thread 0)
    __device__ __forceinline__ float from_left(float mine, float first) {
        L->edge[tid] = mine;
        __syncthreads();
        const float v = tid ? L->edge[tid - 1] : first;
        __syncthreads();
        return v;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// one step of the PSS integrator and its state machines, every lane / thread in any state (fm-processor.cpp:699-718,
// stereo-separation.cpp:84-109); the replay path runs it sample by sample
// ---------------------------------------------------------------------------------------------------------------------
struct PssSt { float acc, mean, pdp; int lock_cnt, unlock_cnt; int minimized; };   // (minimized: 0 / 1; an int so that the struct has no padding bytes, which copies would move through scratch)
__device__ __forceinline__ float pss_step(PssSt &s, float alpha, float la, float keep, bool locked, int tag, float err) {
    const bool rst = !locked;                      // unlocked: pilotDelayPSS = 0; pPSS.reset() (fm-processor.cpp:699-702)
    s.pdp = rst ? 0.f : s.pdp; s.acc = rst ? 0.f : s.acc; s.mean = rst ? 0.f : s.mean;
    s.minimized = s.minimized && !rst; s.lock_cnt = rst ? 0 : s.lock_cnt; s.unlock_cnt = rst ? 0 : s.unlock_cnt;
    const float used = s.pdp;
    const bool call = tag >= 0;
    const float error = s.minimized ? err : err * 10.0f;
    const float c4 = 0.785398185253143310546875f;
    const float nacc = fminf(fmaxf(s.acc + alpha * error, -c4), c4);
    const float nmean = la * error + s.mean * keep;
    const bool small = fabsf(nmean) < 0.001f;
    const int lc1 = s.lock_cnt + ((small && !s.minimized) ? 1 : 0);
    const int uc1 = s.unlock_cnt + ((!small && s.minimized) ? 1 : 0);
    const bool nmin = small ? (s.minimized || lc1 > 3 * SINCOS_N) : (s.minimized && !(uc1 > 3 * SINCOS_N));
    s.acc = call ? nacc : s.acc; s.mean = call ? nmean : s.mean; s.minimized = call ? nmin : s.minimized;
    s.lock_cnt = call ? (small ? lc1 : 0) : s.lock_cnt; s.unlock_cnt = call ? (small ? 0 : uc1) : s.unlock_cnt;
    s.pdp = call ? nacc : ((tag == -1) ? 0.f : s.pdp);
    return used;
}

// the PSS part of the metaData snapshot (fm-processor.cpp:673-682) from the state behind the snapshot sample
__device__ __forceinline__ void meta_snapshot(ChanState *st, const ChanParams &P, float pdp, float mean, bool minimized, bool locked) {
    const bool lk = P.fm_mode != 2 && locked;
    st->meta_pss_deg = (float)((double)pdp / 3.14159265358979323846 * 180.0f);
    st->meta_pss_change = mean * 1000;
    st->meta_pss_state = (P.pss_active && lk) ? (minimized ? 2 : 1) : 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// The whole of stage B in ONE kernel per call: one workgroup = one channel, looping over the call's segments of up to 1536 fm
// samples.  Per segment: limiter + discriminator, AFC, pilot PLL, lock detector (nothing of which depends on the PSS feedback),
// then the PSS error of the calls the segment can make (fast convolution of the s ring, fmx_fftconv.h), the PSS integrator,
// 38 kHz mix, matrix, de-emphasis.  The PSS feedback lags by 1753 samples, so a segment's errors only need s-ring entries that
// EARLIER segments of the same workgroup wrote: the barriers between them order those stores and loads (workgroup scope: one
// CU, one L1).  demod / pilot phase / lock flags never leave the registers between the two halves; every recurrence's state
// rides from segment to segment in registers (each thread holds the same copy); the next segment's ring entries are loaded
// while the current one is computed.  Out: the d ring (stage C's input), the s ring, and the scope / RDS taps w_dem, w_cur,
// w_diff (channel-major rows of this call).
// ---------------------------------------------------------------------------------------------------------------------
// The kernel's arguments are read THROUGH the kernarg segment pointer, phase by phase (SB_ARGS_FRESH): taken by value the ~160
// dwords of tables / buffers / geometry are all loaded up front and then spilled to VGPR lanes -- a v_readlane per use, 12 % of
// the kernel's VALU instructions.  Scalar loads from the kernarg segment cost no VALU issue slot.
constexpr float AFC_EXACT_THR = 0.15f;     // rad per fm sample: a carrier 4.6 kHz off tune
struct StageBArgs { DeviceTables T; DeviceBuffers B; CallGeom G; int C; };
typedef const StageBArgs __attribute__((address_space(4))) *StageBArgsP;
#define SB_ARGS_FRESH() asm volatile("" : "+s"(ka))

// 0 <= v < limit for all of a thread's six values in one comparison: non-negative floats order like their bit patterns, and a negative
// value or a NaN has a larger pattern than any limit used here
__device__ __forceinline__ bool all_in(const float *v, float limit) {
    unsigned m = __float_as_uint(v[0]);
#pragma unroll
    for (int i = 1; i < FB_K; i++) m = max(m, __float_as_uint(v[i]));
    return m < __float_as_uint(limit);
}

// inclusive wave scan of affine maps d -> A d + Bv (composition: the later map after the earlier one)
struct Aff { float A, Bv; };
template <int CTRL, int RM> __device__ __forceinline__ Aff aff_step(Aff c) {
    const float pA = dppf<CTRL, RM>(1.f, c.A);                                      // lanes without a source compose with the identity
    float pB;
    if constexpr (RM == 0xf) pB = dppzf<CTRL>(c.Bv); else pB = dppf<CTRL, RM>(0.f, c.Bv);
    Aff r; r.Bv = fmaf(c.A, pB, c.Bv); r.A = c.A * pA;
    return r;
}
__device__ __forceinline__ Aff wscan_aff(Aff v) {
    v = aff_step<0x111, 0xf>(v); v = aff_step<0x112, 0xf>(v); v = aff_step<0x114, 0xf>(v); v = aff_step<0x118, 0xf>(v);
    v = aff_step<0x142, 0xa>(v); v = aff_step<0x143, 0xc>(v);
    return v;
}

#ifndef SB_ABLATE
#define SB_ABLATE 0      /* diagnostic builds only (tools/diag/valu_phases.sh): bit 0 no discriminator, 1 no AFC, 2 no pilot PLL, 3 no lock
                            detector -- for instruction counts per phase; the results are wrong by construction */
#endif
#ifndef SB_SEED_ROUNDS
#define SB_SEED_ROUNDS 2
#endif
#ifndef SB_NEWTON_TOL
#define SB_NEWTON_TOL 2e-3f
#endif
constexpr float PLL_NEWTON_TOL = SB_NEWTON_TOL;   // a Newton round whose largest update is below this ends the iteration: what it leaves is of second order,
                                          // 0.5 sum |g| d^2 < 2e-6 rad over a segment (g = 5 demod gain, sum |g| < 1 for programme material)
constexpr float PLL_GUARD = 0.12f;        // ChanParams::pll_seq == 0: Newton's method needs the lock metric above this through the whole previous segment (threshold 0.07;
                                          // a pilot of 8 % of the deviation settles at 0.28, one of 3 % at 0.105)
constexpr int PLL_NEWTON_MAX = 10;        // rounds before the segment is replayed sample by sample (ChanState::pll_replays counts those)

// PART 0: the whole of stage B in one kernel (handles of few channels: one launch, the channel's latency is what counts).
// PART 1 / 2: the same code as TWO kernels -- limiter .. lock detector, then PSS .. de-emphasis -- for large batches: each half needs
// at most 128 VGPRs without a spill (the whole needs 168), so FOUR workgroups share a CU instead of three, and the kernel is bound by the
// latency of its dependent chains, not by throughput.  The halves meet in the per-call rows they write anyway (w_dem, w_cur: the scope
// taps and the RDS path's inputs) plus one byte of lock flags per thread and segment (w_lockm).  Nothing in the first half depends on
// the second.
template <int PART>
__global__ __launch_bounds__(FB_T, PART == 0 ? SB_WG_PER_SIMD : 4) void stageb_kernel(StageBArgs by_value_never_touched) {
    StageBArgsP ka = (StageBArgsP)__builtin_amdgcn_kernarg_segment_ptr();
#define T (ka->T)
#define B (ka->B)
#define G (ka->G)
    const int C = ka->C;
    __shared__ ScanLds lds;
    // the recurrences' states in front of the next segment: the same for every thread, so they live in LDS, not in everybody's registers
    // (written by one thread behind a phase, read by all in front of the same phase of the next segment: barriers in between)
    __shared__ struct { float afc, x0, old, lock; int locked, stable; PssSt ps; float de_l, de_r; int calls; int newton_ok; } cy;
    // One block of LDS: the convolution's buffer X (afterwards er / pk), then the two rows in which demod and pilot phase of the segment
    // wait while the convolution has the registers (they are not needed in it; kept in registers they pushed the kernel over its budget of
    // 168: spills, i.e. scratch memory for every wave).  The sample-by-sample pass of the pilot PLL, which runs while all of that is
    // free, takes the whole block: FB_W candidate records of 16 bytes and FB_W phases (SpecRec below).
    constexpr int XF = 2 * fftc::LDS_N;                              // floats of X
    __shared__ __attribute__((aligned(16))) float big[XF + 2 * FB_W];
    float2 *const X = reinterpret_cast<float2 *>(big);
    float *er = big;                                                 // [FB_W] PSS error per call of the segment; replay paths: inputs in, results out
    int *pk = reinterpret_cast<int *>(big) + FB_W;                   // [FB_W] replay path: ((tag + 2) << 1) | locked per sample
    static_assert(2 * FB_W <= XF, "er and pk live in the convolution buffer");
    float *const park_dem = big + XF, *const park_cur = big + XF + FB_W;
    // the sample-by-sample pass of the pilot PLL: five rows of FB_W floats -- the corrections for the two table entries next to the guess, the
    // phase at which the entry changes, and the two addends of the wrap; the phases the pass finds replace the first row
    float *const sq_lo = big, *const sq_hi = big + FB_W, *const sq_b = big + 2 * FB_W, *const sq_wa = big + 3 * FB_W, *const sq_wb = big + 4 * FB_W;
    float *const pout = sq_lo;
    static_assert(5 * FB_W <= XF + 2 * FB_W && (FB_W % 4) == 0, "the rows fit the block");
    __shared__ __attribute__((aligned(16))) DecayTab<(PART == 0 ? 4 : 2)> dtab;
    const int ch = blockIdx.x + G.ch0;
    if (ch >= C) return;
#ifdef SB_POISON_LDS
    // (diagnostic build: every LDS word a NaN pattern before anything is written -- a read of a word nobody wrote shows in the results)
    for (int i = threadIdx.x; i < XF + 2 * FB_W; i += FB_T) big[i] = __int_as_float(0x7fc0dead);
    for (int i = threadIdx.x; i < (int)(sizeof(lds) / 4); i += FB_T) reinterpret_cast<int *>(&lds)[i] = 0x7fc0dead;
    __syncthreads();
#endif
    WG wg; wg.L = &lds; wg.tid = threadIdx.x; wg.lane = threadIdx.x & 63; wg.wv = threadIdx.x >> 6; wg.sl = 0;
    const ChanParams &P = B.params[ch];
    ChanState *st = B.state + ch;
    const int nj = (int)(G.J1 - G.J0);
#ifdef SB_PHASE_CYCLES                                           // (diagnostic build, tools/build_variant.sh: cycles per phase of thread 0 -- 20 VGPRs)
    const bool dbg_on = (B.dbg != nullptr) && (threadIdx.x == 0);
    unsigned long long dbg_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long dbg_t = dbg_on ? clock64() : 0ull;
#define SB_TICK1(k) SB_TICK(k)
#else
#define SB_TICK1(k) asm volatile("; SB_PHASE_END " #k " F%c0" :: "i"(SB_FASTFLAG))   /* (a marker in the assembly listing: tools/isa_phases.py) */
#endif
#ifdef SB_FINE_TICKS
    __shared__ unsigned long long ft_acc[64];
    const bool ft_on = (B.dbg != nullptr) && (threadIdx.x == 0);
    if (threadIdx.x < 64) ft_acc[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned long long ft_t = ft_on ? clock64() : 0ull;
#endif
    const bool stereo_possible = P.fm_mode != 2, auto_mono = P.auto_mono != 0, pss_active = P.pss_active != 0;
    // A demodulator with a recurrence of its own -- pllC (PLL and AM decoders, fm-demodulator.cpp:133-166, 215-241, pllC.cpp:67-90), the
    // squelches (squelchClass.cpp:47-113) -- has run before this kernel, one lane per channel (launch_demod_fused: disc_kernel and
    // afc_kernel<true> of fmx_demod.hip over the whole call): its output, AFC / scaling / squelch applied, waits in the 16-row tiles of
    // w_osc and this kernel starts behind the demodulator.
    const bool special = P.decoder <= 2 || P.squelch_mode != 0;
    const bool pss_on = stereo_possible && pss_active;
    // the sample behind which the reference takes its metaData snapshot (++myCount > fmRate / 2, fm-processor.cpp:662-684), call-relative
    const int my_count0 = st->my_count;
    const int jx = (SINCOS_N >> 1) - my_count0;
    // ring entries of the samples j0 - 2 .. j0 + K - 1 of the segment at seg0 (clamped to the call's last sample: never past what
    // stage A wrote; zero until the filter latency has elapsed; the marker NaN where the reference's start values 0.01 apply)
    // (loop-invariant scalars of the prefetch and of the segment dispatch, read once: through the kernarg pointer they would cost a chain
    // of three dependent scalar loads per segment, 1.5 k cycles of a wave's 58 k)
    const int zdelay = special ? 0 : T.front_sets[P.front_set].delay_fm;
    const float2 *const zr = B.zring + (size_t)ch * (G.ring_mask + 1);
    const int64_t callJ0 = G.J0;
    const int zmask = G.ring_mask;
    auto fetch = [&](int j0, int seg0, int w, float2 *z) {
        const int delay = zdelay;
        const int64_t base = callJ0 + seg0;
        if (w == FB_W && base - 2 - delay >= 0) {                // (the same for every thread) a full segment behind the filter latency:
            const int rmask = zmask;                             // no clamp, no marker, no zero fill; ring positions in 32 bits
            const int r0 = (int)((base - delay) & rmask) + j0 - 2;
            // (eight consecutive entries: where no lane's run crosses the ring's end -- the same for the whole wave -- one address
            // and eight immediate offsets instead of eight masked indices)
            if (!__any(r0 < 0 || r0 + FB_K + 1 > rmask)) {
                const float2 *p = zr + r0;
#pragma unroll
                for (int t = 0; t < FB_K + 2; t++) z[t] = p[t];
                return;
            }
#pragma unroll
            for (int t = 0; t < FB_K + 2; t++) z[t] = zr[(r0 + t) & rmask];
            return;
        }
#pragma unroll
        for (int t = 0; t < FB_K + 2; t++) {
            const int jr = j0 - 2 + t;
            const int64_t jj = base + (jr < w ? jr : w - 1);
            const int64_t s = jj - delay;
            z[t] = jj < 0 ? make_float2(__builtin_nanf(""), 0.f) : (s >= 0 ? zr[s & zmask] : make_float2(0.f, 0.f));
        }
    };
    if (threadIdx.x == 0) {
        cy.afc = st->fm_afc; cy.x0 = st->pil_phase; cy.old = st->pil_old; cy.lock = st->pil_lock;
        cy.locked = st->pil_locked; cy.stable = st->pil_stable; cy.newton_ok = st->pll_newton_ok;
        PssSt ps;
        ps.acc = st->pss_acc; ps.mean = st->pss_mean; ps.pdp = st->pilot_delay_pss;
        ps.lock_cnt = st->pss_lock_cnt; ps.unlock_cnt = st->pss_unlock_cnt; ps.minimized = st->pss_minimized != 0;
        if (P.actions & (ACT_TRIGGER_FREQ | ACT_RESTART_PSS)) {
            // triggerFrequencyChange / restartPssAnalyzer fm-processor.cpp:849-860
            ps.pdp = 0.f; ps.acc = 0.f; ps.minimized = false; ps.mean = 0.f; ps.lock_cnt = 0; ps.unlock_cnt = 0;
            if (P.actions & ACT_TRIGGER_FREQ) st->fade_start_frame = G.M0;
        }
        cy.ps = ps; cy.de_l = st->de_l; cy.de_r = st->de_r;
        cy.calls = 0;                                            // process_sample calls of this call's earlier segments
    }
    const int64_t pss_count0 = st->pss_count;
    if (threadIdx.x < 64) {
        if (PART != 2) { store_decay(&dtab, DEC_AFC, T.afc_l2, threadIdx.x); store_decay(&dtab, DEC_LOCK, T.lock_l2, threadIdx.x); }
        if (PART != 1) { store_decay(&dtab, DEC_PSSMEAN, T.pssmean_l2, threadIdx.x); store_decay(&dtab, DEC_DEEMPH, P.deemph_l2, threadIdx.x); }
    }
    float2 zn[FB_K + 2];
    if (PART != 2 && !special) fetch((int)threadIdx.x * FB_K, 0, nj < FB_W ? nj : FB_W, zn);
    __syncthreads();
    // One segment.  FAST = a full segment (every thread has its six samples) that neither holds the metaData snapshot sample nor the
    // call's first two samples: all the `i < nv` / `i == il` / `i == ix` guards of the general form fold away at compile time (they
    // were a third of the kernel's VALU instructions: v_cndmask, exec-mask bookkeeping, SGPR spills).  The general form runs the
    // ragged last segment of a call and the one segment in 62 that takes the snapshot.
    // EXACT = this segment's pilot PLL is evaluated on the sequential trajectory (see the pilot PLL below): an instantiation of its own,
    // so that the registers its solvers need do not press on the segments that run Newton's method (spill code sits where the pressure is).
    bool newton_ok_next = st->pll_newton_ok != 0;                     // (the same value in every thread: set by the lock detector from workgroup-wide results)
    // A MISTUNED channel of a batch (pll_seq 0): the AFC average of its demodulator output is large, and the forms a batch uses -- the average as a time-parallel
    // scan (6e-6 x |afc| of wander), the short forms of the limiter's and the table index's divisions -- show in its demodulator output (1e-4 of its scale at 1 rad).
    // While |afc| is above AFC_EXACT_THR (a station 4.6 kHz off tune; a tuned one sits below 0.03) its segments take the EXACT instantiation with the sequentially
    // walked average and the reference's divisions, as a small handle's do; tuned channels pay nothing.  (The same value in every thread: the scan's end value.)
    bool afc_big_next = P.pll_seq == 0 && !special && fabsf(__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(st->fm_afc)))) > AFC_EXACT_THR;
    auto segment = [&](auto fast_tag, auto exact_tag, const int seg0) {
        constexpr bool FAST = decltype(fast_tag)::value;
        constexpr bool EXACT = decltype(exact_tag)::value;
        constexpr int SB_FASTFLAG = FAST ? 1 : 0;
        const bool afc_big = EXACT && afc_big_next;                  // (what sent this segment here, among other things)
        // Everything below that only depends on the thread index (table addresses, twiddles, scan weights, the ramp) is
        // loop-invariant, and the compiler would keep it all in registers across the loop (346 VGPRs): the index is made opaque
        // per segment, so those values are recomputed / reloaded (L1 hits) where they are used.
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        wg.tid = tid; wg.lane = tid & 63; wg.wv = tid >> 6;
        const int lane = wg.lane;
        const int j0 = tid * FB_K;                                   // segment-relative index of this thread's first sample
        const int w = FAST ? FB_W : ((nj - seg0) < FB_W ? (nj - seg0) : FB_W);
        const bool lastseg = seg0 + FB_W >= nj;
        const int nv = FAST ? FB_K : w - j0;                         // sample i of this thread exists when i < nv
        const bool owner = FAST ? (tid == FB_T - 1) : (nv >= 1 && nv <= FB_K);   // this thread owns the segment's last sample
        const int il = FAST ? FB_K - 1 : nv - 1;                     // ... at this position
        const int ix = FAST ? -1 : jx - seg0 - j0;                   // this thread's index of the metaData snapshot sample, if 0 .. K-1

        SB_FT(0); SB_FTW(1);      // 0: loop top bookkeeping, 1: wait for the prefetched ring entries
        // what the second half takes over from the first
        float dem[FB_K], cur[FB_K];
        bool locked[FB_K];
        bool all_locked;                                         // every sample of the segment (the same in every thread)
        bool none_locked = false;                                // surely no sample of the segment (the same in every thread): the PSS filter has no call to serve
        uint8_t *const lockm = B.w_lockm + (size_t)ch * B.lockm_stride + (size_t)(seg0 / FB_W) * FB_T + tid;
        if constexpr (PART != 2) {
        if (!special) {
        // ================= limiter + discriminator (fm-demodulator.cpp:119-126, 168-189) =================
        float res[FB_K];
        if (SB_ABLATE & 1) {
#pragma unroll
            for (int i = 0; i < FB_K; i++) res[i] = zn[i + 2].x;
        } else {
            const int decoder = P.decoder;
            float2 lim[FB_K + 2];                                // limited samples j0-2 .. j0+K-1 (the two in front recomputed: cheaper
#pragma unroll                                                   // than an exchange through LDS with its two barriers)
            for (int t = 0; t < FB_K + 2; t++) lim[t] = (!FAST && zn[t].x != zn[t].x) ? make_float2((float)0.01, (float)0.01) : limiter_fast(zn[t]);
            // Handles that are asked for the reference's own values (pll_seq 1: up to 64 channels) take the limiter and the discriminators with
            // the reference's divisions and its f64 square root: the short forms above agree with them to an ulp, which moves a table index by
            // one entry in one sample of some ten thousand (7.9e-5 of the demodulator's scale at that sample).
            bool exact_disc = false;
            if constexpr (EXACT) {
                exact_disc = P.pll_seq == 1 || afc_big;
                if (exact_disc) {
#pragma unroll
                    for (int t = 0; t < FB_K + 2; t++) lim[t] = (!FAST && zn[t].x != zn[t].x) ? make_float2((float)0.01, (float)0.01) : limiter(zn[t]);
                }
            }
            SB_FT(2);
            // (one loop per decoder: the six table gathers of a thread are issued back to back, not one per branch arm)
            if (decoder == 5) {                                      // REAL_BB :174-182
                int index[FB_K];
#pragma unroll
                for (int i = 0; i < FB_K; i++) {
                    const float I = lim[i + 2].x, Q = lim[i + 2].y, I1 = lim[i + 1].x, Q1 = lim[i + 1].y;
                    const float r = (float)((double)(I1 * Q - Q1 * I + 1) / 2.0);
                    int ixx = (int)floorf(r * (float)ARCSINE_N);
                    ixx = ixx < 0 ? 0 : ixx;
                    index[i] = ixx >= ARCSINE_N ? ARCSINE_N : ixx;
                }
#pragma unroll
                for (int i = 0; i < FB_K; i++) res[i] = T.arcsine[index[i]];
            } else if (decoder == 6) {                               // DIFF :184-189
#pragma unroll
                for (int i = 0; i < FB_K; i++) {
                    const float I = lim[i + 2].x, Q = lim[i + 2].y, I1 = lim[i + 1].x, Q1 = lim[i + 1].y;
                    const float Scaler = (float)1.4142135623730951;
                    const float r = (I1 * (Q - lim[i].y) - Q1 * (I - lim[i].x));
                    res[i] = exact_disc ? r / ((I1 * I1 + Q1 * Q1) * Scaler) : fdiv_fast(r, (I1 * I1 + Q1 * Q1) * Scaler);
                }
            } else {                                                 // MIXED :168-172 (COMPLEX_BB is bitwise the same)
                AtanArmF arm[FB_K];
                bool odd = false;                                    // an argument pair compAtan::atan2 answers without its table
#pragma unroll
                for (int i = 0; i < FB_K; i++) {
                    const float I = lim[i + 2].x, Q = lim[i + 2].y, I1 = lim[i + 1].x, Q1 = lim[i + 1].y;
                    arm[i] = atan_arm_plain(Q * I1 - I * Q1, I * I1 + Q * Q1, &odd);
                }
                SB_FT(3);
                float tv[FB_K];
#pragma unroll
                for (int i = 0; i < FB_K; i++) tv[i] = T.atan_ppy[arm[i].idx];
                SB_FTW(4);
#pragma unroll
                for (int i = 0; i < FB_K; i++) res[i] = arm[i].A + (arm[i].neg ? -tv[i] : tv[i]);
                if (exact_disc) {                                    // (compAtan::atan2 with the reference's division, Xtan2.cpp:56-100)
#pragma unroll
                    for (int i = 0; i < FB_K; i++) {
                        const float I = lim[i + 2].x, Q = lim[i + 2].y, I1 = lim[i + 1].x, Q1 = lim[i + 1].y;
                        res[i] = lut_atan2(T.atan_ppy, Q * I1 - I * Q1, I * I1 + Q * Q1);
                    }
                } else
                if (__any(odd)) {                                    // (x = 0, an infinity, a NaN: the general form for the thread's samples)
#pragma unroll
                    for (int i = 0; i < FB_K; i++) {
                        const float I = lim[i + 2].x, Q = lim[i + 2].y, I1 = lim[i + 1].x, Q1 = lim[i + 1].y;
                        res[i] = lut_atan2_fast(T.atan_ppy, Q * I1 - I * Q1, I * I1 + Q * Q1);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < FB_K; i++) res[i] = (i < nv) ? res[i] : 0.f;
        }
        SB_FT(5);
        SB_ARGS_FRESH(); SB_TICK1(0);

        // ================= AFC + scaling (fm-demodulator.cpp:197-198) =================
        if (SB_ABLATE & 2) {
#pragma unroll
            for (int i = 0; i < FB_K; i++) dem[i] = res[i];
        } else {
            const float fmDcAlpha = 0.0001f, c1 = 1 - fmDcAlpha;
            float Lt = 0.f;
#pragma unroll
            for (int i = 0; i < FB_K; i++) Lt = c1 * Lt + fmDcAlpha * res[i];
            float afc_next;
            float afc = wg.decay_incoming2(Lt, cy.afc, load_decay(&dtab, DEC_AFC, lane), &afc_next);
            afc_big_next = P.pll_seq == 0 && fabsf(__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(afc_next)))) > AFC_EXACT_THR;     // (the same value in every lane: said so)
            bool afc_exact = false;
            if constexpr (EXACT) {
                // Handles that are asked for the reference's trajectories (pll_seq 1: up to 64 channels): the scan above gives the state in
                // front of this thread's six samples to ~1e-7 of it, summed in another order than the reference's sample-by-sample walk --
                // 6e-6 x |afc| of wander, 1e-4 of the demodulator's scale for a signal whose mean phase step is ~1 rad.
                // The recurrence forgets a shift of its state at 1e-4 per sample, i.e. a shift by a few ulps rides through a thread's six
                // samples unchanged (unless it flips a rounding: 6e-4 per ulp and run): every thread runs its samples from its incoming
                // value, the end values' misses against the next threads' incoming values are prefix-summed into corrections, and the
                // passes repeat until every run ends on the next run's start -- the chain from the exact cy.afc is then the sequential one.
                if (P.pll_seq == 1 || afc_big) {
                    const float afc0 = cy.afc;
                    for (int pass = 0; pass < 8; pass++) {
                        float o = afc;
#pragma unroll
                        for (int i = 0; i < FB_K; i++) if (i < nv) o = c1 * o + fmDcAlpha * res[i];
                        float po = dppf<0x138, 0xf>(0.f, o);
                        if (lane == 63) lds.wf[wg.sl][wg.wv][2] = o;
                        __syncthreads();
                        if (lane == 0) po = wg.wv ? lds.wf[wg.sl][(wg.wv + 3) & 3][2] : afc0;
                        wg.sl ^= 1;
                        const double d = (nv >= 1) ? (double)po - (double)afc : 0.0;
                        double total; bool any;
                        const double pre = wg.excl_add_d(d, &total, d != 0.0, &any);
                        if (!any) { afc_exact = true; break; }
                        afc = (float)((double)afc + (pre + d));
                    }
                }
            }
            float afc_end = 0.f;
            const float K_FM = T.K_FM, K_FM_rcp = T.K_FM_rcp;
#pragma unroll
            for (int i = 0; i < FB_K; i++) {
                afc = c1 * afc + fmDcAlpha * res[i];
                dem[i] = fdiv_const(20.0f * (res[i] - afc) * 1.0f, K_FM, K_FM_rcp);
                if (i == il) afc_end = afc;
                if (i == ix && i < nv) st->meta_dc_if = afc;                 // get_demodDcComponent () at the snapshot
            }
            if (lastseg && owner) st->fm_afc = afc_end;
            if (afc_exact) { if (owner) cy.afc = afc_end; }          // (the chain's own end value: read again behind the next segment's barriers)
            else if (tid == 0) cy.afc = afc_next;
        }
        } else {
            // (demodulator output of the pre-pass: row r of this call, channel ch, in the 16-row tiles of w_osc)
#pragma unroll
            for (int i = 0; i < FB_K; i++) dem[i] = (i < nv) ? B.w_osc[widx((int64_t)(seg0 + j0 + i), ch, G.pitch)] : 0.f;
        }
        SB_FT(6);
        SB_ARGS_FRESH(); SB_TICK1(1);

        // ================= pilot PLL (pilot-recover.cpp:54-61) =================
        // The loop is non-linear (the phase feeds the sine look-up that corrects it).  Two ways to its f32 trajectory
        // x[j+1] = step (x[j]), step = the reference's own f32 expression:
        //  * sequentially, by one thread from LDS (ChanParams::pll_seq, FMX_P_PLL_SOLVER): the reference's trajectory bit for bit
        //    (1e-7 rad against the oracle, from stage A's 1e-6 input differences) at ~45 us per segment -- what a handle with few
        //    channels (the drop-in receiver) uses;
        //  * by Newton's method on the whole segment at once.  With a guess x, P = the f64 prefix sum of the exact increments
        //    step (x) - x and d = x0 + P - x (what Picard's iteration would add), the update solves the linearised recurrence: the
        //    next guess is fl (x0 + P[j] + S[j]), S[j+1] = (1 + c[j]) S[j] + c[j] d[j], c = g cos x, g = 5 demod gain -- a scan of
        //    affine maps in f32.  The first guess is the free-running ramp improved by two rounds of x = ramp + sum g sin (x) in
        //    plain f32 with the hardware sine (within ~1e-4 rad), so ONE Newton round normally settles it and the trajectory is
        //    evaluated once more for its outputs.  What this cannot reproduce is the loop's own rounding noise: the step rounds
        //    twice per sample (half an ulp, 2.4e-7 rad), the loop integrates that over its time constant (~12000 samples), and a
        //    guess a few ulps off rounds differently at every sample -- the result wanders around the reference's trajectory by
        //    ~1.5e-5 rad RMS whatever the number of rounds (tools/pll_fixed_point.py; the Picard iteration of round 2 did the same).
        //    A segment that does not settle in PLL_NEWTON_MAX rounds is evaluated sequentially (ChanState::pll_replays counts them).
        float osc[FB_K];
        float osc_in;                                            // NCO sine of the sample in front of this thread's first
        if (SB_ABLATE & 4) {
#pragma unroll
            for (int i = 0; i < FB_K; i++) { cur[i] = dem[i]; osc[i] = dem[i]; }
            osc_in = dem[0];
        } else {
            const float gain = T.pil_gain, omega = T.pil_omega;
            const double SC64 = T.sincos_C;
            const float P32 = 6.2831855f, C32 = T.wrap32_c, INV2PI32 = 0.159154943f;
            const bool wrap_ok = T.wrap32_ok != 0;
            float x0 = cy.x0;
            if (!(x0 >= 0.f && x0 < P32)) x0 = pi_constrain(x0);
            float g[FB_K];
#pragma unroll
            for (int i = 0; i < FB_K; i++) g[i] = (i < nv) ? (5 * dem[i]) * gain : 0.f;
            float ph[FB_K];
            // the reference's step on one guess: table index in f64 as sincos.cpp:81-85 computes it, everything else in f32
            float nxl = 0.f;                                     // step result of this thread's last evaluated sample `il` (the owner's: the next segment's start)
            // Which segments are evaluated sample by sample (the same decision in every thread).  pll_seq 1: all of them.  pll_seq 0 (large
            // batches): the segments in which a lock decision can fall -- Newton's trajectory carries the rounding noise of ~1e-5 rad
            // described above, which moves the lock metric by ~1e-6, enough to shift the sample at which the metric crosses its threshold
            // (and, half a second later, the sample at which the stereo decoder switches) when it crosses slowly.  So Newton's method only
            // runs while the pilot is in lock AND the metric stayed above PLL_GUARD through the whole previous segment; the acquisition and
            // every approach of the threshold run on the reference's own trajectory (cy.newton_ok, set by the lock detector below).
            const int pll_mode = P.pll_seq;
            bool exact_pending = EXACT;                          // (the dispatch below decides: pll_mode 1, or pll_mode 0 and not newton_ok)
            const bool guard_seg = pll_mode == 0 && EXACT;
            bool seq = false;                                    // this pass evaluates the loop sample by sample (the same in every thread)
            bool failsafe = false;                               // ... because Newton's iteration did not settle
            bool force_plain = false;                            // ... again, without the predictions of the first pass, which did not hold
            auto eval = [&](int i, float phase, float *nx_out, float *val_out) {
                int idx = (int)((double)phase * SC64);           // SinCos::getSin sincos.cpp:81-85 for phase >= 0: entry (int)(phase * C) % Rate
                idx = (int)min((unsigned)idx, (unsigned)idx - (unsigned)SINCOS_N);       // (0 <= idx < 2 N: the wrap as an unsigned minimum)
                // (table value: from the hardware sine unit on the index folded into the first quadrant, within 1.2e-7 of the reference's entry
                // -- every evaluation of the step in this kernel, whichever solver asks: the runs of the exact solvers and this evaluation,
                // which verifies them, must read the same values.  Round 3's polynomial for the sample-by-sample solver was half as far
                // from the table and three times the instructions; the pilot phase agrees with the oracle's to 7e-7 rad rms either way.)
                const float o = sin_idx_hw(idx);
                const float perr = (5 * dem[i]) * o;             // pilot-recover.cpp:56-58 (pilot = 5 * demod fm-processor.cpp:696)
                const float t = phase + perr * gain;
                const float val = t + omega;
                const float wrapped = wrap_ok ? (val - P32) + C32 : (float)((double)val - FMX_2PI);
                // PI_Constrain fm-constants.h:148-158 for 0 <= val < 4 pi (anything else -- a correction of more than a turn: the DIFF
                // decoder's spike where the limiter output jumps from its 0.001 floor to the unit circle at signal onset -- is seen by
                // the caller's check of all six values at once and takes the general form there)
                cur[i] = t; osc[i] = o;
                *val_out = val;
                *nx_out = (val < P32) ? val : wrapped;
            };
            // a guess outside [0, 2 pi) (unfinished rounds only) is taken modulo 2 pi
            auto into_range = [&](float phase) {
                const double pd = (double)phase;
                const float pw = (float)(pd - floor(pd * (1.0 / FMX_2PI)) * FMX_2PI);
                return (phase >= 0.f && phase < P32) ? phase : ((pw >= 0.f && pw < P32) ? pw : 0.f);
            };
            {   // ---- the first guess
                float rv[FB_K];                                  // the ramp x0 + j omega in turns, fraction
                const double tb = ((double)x0 + (double)j0 * (double)omega) * (1.0 / FMX_2PI);
                const float tbf = (float)(tb - floor(tb));
#pragma unroll
                for (int i = 0; i < FB_K; i++) { const float v = tbf + (float)i * (omega * INV2PI32); rv[i] = v - floorf(v); }
                float cor[FB_K];                                 // sum of the corrections in front of each sample (rad)
#pragma unroll
                for (int i = 0; i < FB_K; i++) cor[i] = 0.f;
#pragma unroll
                for (int round = 0; round < SB_SEED_ROUNDS; round++) {
                    float c[FB_K], run;
#pragma unroll
                    for (int i = 0; i < FB_K; i++) {             // (x + 0 is not x to the compiler: the first round and the first term spelled out)
                        const float sv = g[i] * __builtin_amdgcn_sinf(round == 0 ? rv[i] : rv[i] + cor[i] * INV2PI32);
                        if (i == 0) { c[i] = 0.f; run = sv; } else { c[i] = run; run += sv; }
                    }
                    float total;
                    const float pre = wg.excl_add_f(run, &total);
#pragma unroll
                    for (int i = 0; i < FB_K; i++) cor[i] = pre + c[i];
                }
#pragma unroll
                for (int i = 0; i < FB_K; i++) {
                    float v = rv[i] + cor[i] * INV2PI32;
                    v = (v - floorf(v)) * P32;
                    ph[i] = (j0 + i == 0) ? x0 : (v < P32 ? v : 0.f);
                }
            }
            SB_FT(7);
            // Every guess is x[j] = fl (x0 + P[j] + S[j]): P = the f64 prefix sum of the exact increments step (x) - x of the previous
            // guess (what Picard's iteration would take), S = the Newton correction, S[j+1] = (1 + c[j]) S[j] + c[j] d[j] with
            // c = g cos x, d = x0 + P - x -- a scan of affine maps in f32 (S is small).  One rounding per sample, none accumulated:
            // an update d -> x + d in f32 would leave half an ulp of residual at EVERY step, which the loop integrates to 1e-5 rad.
            // (One loop for both ways, so that the step's evaluation exists once in the code.)
            bool open_ = true;                                   // this thread's last update was not small yet
            const double x0d = (double)x0;
            for (int it = 0; ; it++) {
                float ph_next = 0.f;                             // (sample-by-sample pass) the phase behind this thread's last sample
                if (seq) {
                    // The loop sample by sample: one thread, operands through LDS (the block is free here).  A lone wave issues an instruction
                    // -- of any kind -- every 4 to 5 cycles at best, and the reference's step with its table look-up is ~25 of them plus
                    // branches: 220-240 cycles per sample measured.  But the guess at hand (Newton's, good to a few 1e-6 rad; a table step is
                    // 3.3e-5) already tells which table entry each sample will read, give or take one, and whether its step will wrap.  So all
                    // threads tabulate, per sample: the correction 5 demod sin (.) gain for the two entries next to the guess (the f32
                    // expressions of the step, the same table values the final evaluation uses), the phase at which the entry changes, and the
                    // two addends of the wrap (0, 0 or -P32, C32).  The serial pass is then compare, select and four additions per sample --
                    // x' = (((x + c) + omega) + A) + B is the reference's  t = phase + corr; val = t + omega; PI_Constrain (val)  where the
                    // prediction holds.  Whether it held is checked by the evaluation that follows anyway: the exact step of every phase found
                    // must be the next phase found, bit for bit (a trajectory with that property that starts at x0 IS the sequential one); a
                    // segment that fails the check -- or whose Newton iteration failed -- takes the plain loop below.
                    __syncthreads();
                    const bool plain = !EXACT || failsafe || force_plain || !wrap_ok;
                    // ---- first: the pilot periods of the segment side by side.  Once per period (~10.1 samples) the phase enters [4, 8),
                    // where every f32 is a multiple of U = 2^-21 -- and U is a multiple of every ulp the phase has anywhere in [0, 2 pi).  Two
                    // trajectories that differ by k U at such an ANCHOR and make the same decisions afterwards (table entry of every sample,
                    // where the step wraps, where a binade is crossed) differ by exactly k U at every later sample: the rounding of an
                    // addition commutes with a shift by a multiple of its grid.  So the ~152 runs from anchor to anchor of the GUESS are
                    // evaluated by as many threads at once, each with the reference's own step, ten or eleven samples deep; a run's end misses
                    // the next run's start by an integer d_c (in U); the prefix sums of the d_c are the shifts K_c of the true trajectory
                    // against the guess -- provided no decision changes under the shift, which the next pass, started from the shifted
                    // anchors, shows: passes are repeated until every run ends on the next run's start.  A chain with that property that
                    // starts at x0 IS the sequential trajectory (tools/pll_cycle_sim.py: 3-5 passes, bit-identical on every segment tried).
                    // Anything outside the assumptions (a run that is too long, an end off the grid, no agreement in CYC_MAXPASS passes)
                    // leaves the segment to the single-thread pass below; the evaluation behind this block checks the result either way.
                    bool cyc_ok = false;
                    if constexpr (EXACT) {
                    if (!plain) {
                        constexpr int CYC_LMAX = 12, CYC_MAXPASS = 10, CYC_RUNS = FB_T;
                        float *const cg = big + 2 * FB_W, *const cd = big + 3 * FB_W;         // the guess, and 5 demod, per sample
                        int *const canc = reinterpret_cast<int *>(big + 4 * FB_W), *const cK = canc + CYC_RUNS + 8;   // anchors (sample indices), shifts
#pragma unroll
                        for (int i = 0; i < FB_K; i++) if (i < nv) { cg[j0 + i] = ph[i]; cd[j0 + i] = 5 * dem[i]; }
                        __syncthreads();
                        bool af[FB_K]; int cnt = 0;
                        {
                            float prev = (j0 > 0 && nv > 0) ? cg[j0 - 1] : 0.f;
#pragma unroll
                            for (int i = 0; i < FB_K; i++) {
                                af[i] = i < nv && (j0 + i == 0 || (ph[i] >= 4.0f && prev < 4.0f));
                                cnt += af[i] ? 1 : 0; prev = ph[i];
                            }
                        }
                        int pre, nc, dm1, dm2;
                        wg.excl_add_max_i(cnt, 0, &pre, &nc, &dm1, &dm2);
#pragma unroll
                        for (int i = 0; i < FB_K; i++) if (af[i]) { if (pre < CYC_RUNS) canc[pre] = j0 + i; pre++; }
                        if (tid == 0) canc[nc < CYC_RUNS ? nc : CYC_RUNS] = w;
                        cK[tid] = 0;
                        __syncthreads();
                        const int c = tid;
                        const bool runner = c < nc && nc <= CYC_RUNS;
                        const int s_ = runner ? canc[c] : 0, len = runner ? canc[c + 1] - s_ : 0;
                        const bool has_next = runner && c + 1 < nc;
                        const float gs = runner ? cg[s_] : 4.0f, ge = has_next ? cg[s_ + len] : 4.0f;
                        bool fail = nc > CYC_RUNS || (runner && (len > CYC_LMAX || len < 1));
                        constexpr float Uq = 4.76837158203125e-07f;                             // 2^-21
                        int passes = 0;
                        for (; passes < CYC_MAXPASS; passes++) {
                            const int Kc = cK[c], Kn = (c + 1 < CYC_RUNS) ? cK[c + 1] : 0;
                            float x = (c == 0) ? x0 : gs + (float)Kc * Uq;
                            if (runner && c > 0 && !(x >= 4.0f && x < 8.0f)) fail = true;
                            for (int i = 0; i < CYC_LMAX; i++) {
                                if (runner && i < len) {
                                    pout[s_ + i] = x;
                                    int idx = (int)((double)x * SC64);
                                    idx = idx >= SINCOS_N ? idx - SINCOS_N : idx;
                                    const float val = (x + (cd[s_ + i] * sin_idx_hw(idx)) * gain) + omega;
                                    // (PI_Constrain of [0, 4 pi) without a branch -- some run wraps in nearly every step; anything else ends this way)
                                    fail = fail || !(__float_as_uint(val) < __float_as_uint(2.f * P32));
                                    x = (val < P32) ? val : (val - P32) + C32;
                                }
                            }
                            int d = 0;
                            if (has_next && !fail) {
                                const float dq = (x - (ge + (float)Kn * Uq)) * 2097152.0f;          // in U (exact: both on the grid of [4, 8))
                                const float dr = rintf(dq);
                                if (!(dq == dr) || !(fabsf(dr) < 4096.f)) fail = true; else d = (int)dr;
                            }
                            int preD, totD, preM, totM;
                            wg.excl_add_max_i(d, (d != 0 ? 1 : 0) | (fail ? 2 : 0), &preD, &totD, &preM, &totM);
                            if (totM >= 2) break;
                            if (totM == 0) { cyc_ok = true; break; }
                            cK[c] = Kc + preD;
                            __syncthreads();
                        }
                        if (B.dbg && tid == 0) { B.dbg[(size_t)ch * DBG_SLOTS + 28] += passes + 1; B.dbg[(size_t)ch * DBG_SLOTS + 29] += 1; B.dbg[(size_t)ch * DBG_SLOTS + 30] += cyc_ok ? 0 : 1; }
                        __syncthreads();
                    }
                    }
                    const bool spec = EXACT && !plain && !cyc_ok;    // ---- second: one thread, its table arithmetic prepared by all
                    if constexpr (EXACT) {
                    if (spec) {
                        constexpr double INVC = FMX_2PI / SINCOS_N;
#pragma unroll
                        for (int i = 0; i < FB_K; i++) {
                            float clo = 0.f, chi = 0.f, bb = 0.f, wa = 0.f, wb = 0.f;
                            if (i < nv) {
                                int idx = (int)((double)ph[i] * SC64);
                                idx = (int)min((unsigned)idx, (unsigned)idx - (unsigned)SINCOS_N);
                                idx = (unsigned)idx < (unsigned)SINCOS_N ? idx : 0;              // (a guess outside [0, 2 pi): the check will see what comes of it)
                                // the two entries: the guess's own and the one on the side of the cell the guess sits in
                                const bool lower = (double)ph[i] < ((double)idx + 0.5) * INVC;
                                int klo = lower ? idx - 1 : idx;
                                klo = klo < 0 ? 0 : (klo > SINCOS_N - 2 ? SINCOS_N - 2 : klo);
                                const float d5 = 5 * dem[i];
                                clo = (d5 * sin_idx_hw(klo)) * gain; chi = (d5 * sin_idx_hw(klo + 1)) * gain;
                                // the smallest f32 phase whose index (int) ((double) phase * C) reaches klo + 1: the f32 nearest (klo + 1) / C, or the one above it
                                const float bh = (float)((double)(klo + 1) * INVC);
                                bb = ((int)((double)bh * SC64) >= klo + 1) ? bh : __int_as_float(__float_as_int(bh) + 1);
                                const float vg = (ph[i] + (idx == klo ? clo : chi)) + omega;
                                const bool wr = !(vg < P32);
                                wa = wr ? -P32 : 0.f; wb = wr ? C32 : 0.f;
                            }
                            sq_lo[j0 + i] = clo; sq_hi[j0 + i] = chi; sq_b[j0 + i] = bb; sq_wa[j0 + i] = wa; sq_wb[j0 + i] = wb;
                        }
                    }
                    }
                    if (plain) {
#pragma unroll
                        for (int i = 0; i < FB_K; i++) if (i < nv) sq_hi[j0 + i] = dem[i];
                    }
                    SB_FT(33);
                    __syncthreads();
                    SB_FT(34);
                    if constexpr (EXACT)
                    if (tid == 0 && spec) {
                        float x = x0;
                        auto one = [&](float clo, float chi, float bb, float wa, float wb) {
                            const float xin = x;
                            // (both candidates side by side and the choice last: the dependent chain is four additions and a select)
                            const float xl = (((xin + clo) + omega) + wa) + wb, xh = (((xin + chi) + omega) + wa) + wb;
                            x = (xin < bb) ? xl : xh;
                            return xin;
                        };
                        const float4 *qlo = reinterpret_cast<const float4 *>(sq_lo), *qhi = reinterpret_cast<const float4 *>(sq_hi), *qb = reinterpret_cast<const float4 *>(sq_b),
                                     *qwa = reinterpret_cast<const float4 *>(sq_wa), *qwb = reinterpret_cast<const float4 *>(sq_wb);
                        float4 r0 = qlo[0], r1 = qhi[0], r2 = qb[0], r3 = qwa[0], r4 = qwb[0];
                        const int nq = (w + 3) >> 2;
                        for (int q = 0; q < nq; q++) {
                            const int qn = (q + 1 < FB_W / 4) ? q + 1 : q;                    // (the next four are requested before these four are stored)
                            const float4 n0 = qlo[qn], n1 = qhi[qn], n2 = qb[qn], n3 = qwa[qn], n4 = qwb[qn];
                            float4 o;
                            o.x = one(r0.x, r1.x, r2.x, r3.x, r4.x); o.y = one(r0.y, r1.y, r2.y, r3.y, r4.y);
                            o.z = one(r0.z, r1.z, r2.z, r3.z, r4.z); o.w = one(r0.w, r1.w, r2.w, r3.w, r4.w);
                            reinterpret_cast<float4 *>(pout)[q] = o;
                            r0 = n0; r1 = n1; r2 = n2; r3 = n3; r4 = n4;
                        }
                    }
                    if (tid == 0 && plain) {                     // the plain loop: every sample looks its table value up itself
                        float phase = x0;
                        for (int j = 0; j < w; j++) {
                            const float d5 = 5 * sq_hi[j];
                            pout[j] = phase;
                            int idx = (int)((double)phase * SC64);
                            idx = idx >= SINCOS_N ? idx - SINCOS_N : idx;
                            const float val = (phase + (d5 * sin_idx_hw(idx)) * gain) + omega;
                            phase = (val >= 0.f && val < P32) ? val : pi_constrain(val);
                        }
                    }
                    if (tid == 0) {
                        if (failsafe) st->pll_replays += 1;      // (a Newton iteration that did not settle)
                        else if (guard_seg && !force_plain) st->pll_exact_segs += 1;
                        if (B.dbg && force_plain) B.dbg[(size_t)ch * DBG_SLOTS + 15] += 1;
                    }
                    SB_FT(35);
                    __syncthreads();
#pragma unroll
                    for (int i = 0; i < FB_K; i++) ph[i] = (i < nv) ? pout[j0 + i] : 0.f;
                    ph_next = (j0 + FB_K < w) ? pout[j0 + FB_K] : 0.f;
                    __syncthreads();
                    SB_FT(36);
                }
                float nx[FB_K], val[FB_K];
                if (__any(!all_in(ph, P32))) {
#pragma unroll
                    for (int i = 0; i < FB_K; i++) ph[i] = into_range(ph[i]);
                }
#pragma unroll
                for (int i = 0; i < FB_K; i++) eval(i, ph[i], &nx[i], &val[i]);
                if (__any(!all_in(val, 2.f * P32))) {
#pragma unroll
                    for (int i = 0; i < FB_K; i++) nx[i] = pi_constrain(val[i]);
                }
                SB_FT(8);
                if (seq || it > 0) {
                    // This evaluation is the final one unless the update in front of it was not small somewhere: that flag rides on the
                    // barrier of the NCO-sine hand-off the lock detector needs anyway (the sine in front of each thread's first sample;
                    // the last one behind a full segment is the next segment's `old`).
                    int wopen = (!seq && __any(open_)) ? 1 : 0;
                    if (seq && !failsafe && !force_plain) {
                        // the pass that relied on predictions: every phase's exact step must be the next phase
                        bool defect = false;
#pragma unroll
                        for (int i = 0; i < FB_K; i++)
                            defect = defect || (j0 + i + 1 < w && __float_as_int(nx[i]) != __float_as_int(i + 1 < FB_K ? ph[i + 1 < FB_K ? i + 1 : i] : ph_next));
                        wopen |= __any(defect) ? 2 : 0;
                    }
                    if (lane == 63) lds.wf[wg.sl][wg.wv][1] = osc[FB_K - 1];
                    if (lane == 0) lds.wi[wg.sl][wg.wv][1] = wopen;
                    const float cold = cy.old;
                    __syncthreads();
                    const int anyopen = lds.wi[wg.sl][0][1] | lds.wi[wg.sl][1][1] | lds.wi[wg.sl][2][1] | lds.wi[wg.sl][3][1];
                    const float from_prev_wave = lds.wf[wg.sl][(wg.wv + 3) & 3][1], old_next = lds.wf[wg.sl][3][1];
                    wg.sl ^= 1;
                    if (anyopen & 2) { force_plain = true; continue; }
                    if (!anyopen) {
                        osc_in = dppf<0x138, 0xf>(0.f, osc[FB_K - 1]);
                        if (lane == 0) osc_in = wg.wv ? from_prev_wave : cold;
                        if (tid == 0) cy.old = old_next;             // (read again behind the next segment's barriers)
#pragma unroll
                        for (int i = 0; i < FB_K; i++) if (i == il) nxl = nx[i];
                        if (B.dbg && tid == 0 && !seq) { B.dbg[(size_t)ch * DBG_SLOTS + 8] += it; B.dbg[(size_t)ch * DBG_SLOTS + 11] += 1; }
                        break;
                    }
                }
                SB_FT(9);
                // the increments' prefix sums in f64
                double e[FB_K], tot = 0.0;
#pragma unroll
                for (int i = 0; i < FB_K; i++) { e[i] = tot; tot += (i < nv) ? (double)nx[i] - (double)ph[i] : 0.0; }
                double pre;
                {
                    const double inc = wscan_add_d(tot);
                    if (lane == 63) lds.wd[wg.sl][wg.wv][0] = inc;
                    __syncthreads();
                    pre = inc - tot;
#pragma unroll
                    for (int v = 0; v < 3; v++) pre += (v < wg.wv) ? lds.wd[wg.sl][v][0] : 0.0;
                    wg.sl ^= 1;
                }
                if (it == PLL_NEWTON_MAX - 1) { seq = true; failsafe = true; continue; }       // not settled: sample by sample
                // ---- the Newton correction
                float d[FB_K], c[FB_K];
#pragma unroll
                for (int i = 0; i < FB_K; i++) {
                    d[i] = (i < nv) ? (float)((x0d + (pre + e[i])) - (double)ph[i]) : 0.f;
                    c[i] = g[i] * __builtin_amdgcn_cosf(ph[i] * INV2PI32);
                }
                {   // (a wrap that sits one sample apart in guess and step shows as a residual of a whole turn: rare, looked for once)
                    unsigned m = __float_as_uint(d[0]) & 0x7fffffffu;
#pragma unroll
                    for (int i = 1; i < FB_K; i++) m = max(m, __float_as_uint(d[i]) & 0x7fffffffu);
                    if (__any(!(m < __float_as_uint(3.0f)))) {
#pragma unroll
                        for (int i = 0; i < FB_K; i++) {
                            double dd = (x0d + (pre + e[i])) - (double)ph[i];
                            dd = dd > 3.14159265358979323846 ? dd - FMX_2PI : (dd < -3.14159265358979323846 ? dd + FMX_2PI : dd);
                            d[i] = (i < nv) ? (float)dd : 0.f;
                        }
                    }
                }
                Aff m; m.A = 1.f; m.Bv = 0.f;
#pragma unroll
                for (int i = 0; i < FB_K; i++) { m.Bv = fmaf(1.f + c[i], m.Bv, c[i] * d[i]); m.A *= 1.f + c[i]; }
                const Aff ain = wscan_aff(m);
                if (lane == 63) { lds.wf[wg.sl][wg.wv][2] = ain.A; lds.wf[wg.sl][wg.wv][3] = ain.Bv; }
                __syncthreads();
                float S = 0.f;                                   // the correction at the wave's first sample
#pragma unroll
                for (int v = 0; v < 3; v++) if (v < wg.wv) S = fmaf(lds.wf[wg.sl][v][2], S, lds.wf[wg.sl][v][3]);
                wg.sl ^= 1;
                {   // ... at this thread's first sample: the maps of the lanes in front applied to it
                    const float eA = dppf<0x138, 0xf>(1.f, ain.A), eB = dppf<0x138, 0xf>(0.f, ain.Bv);
                    S = fmaf(eA, S, eB);
                }
                float dmax = 0.f;
#pragma unroll
                for (int i = 0; i < FB_K; i++) {
                    if (i < nv) {
                        dmax = fmaxf(dmax, fabsf(d[i] + S));
                        ph[i] = (j0 + i == 0) ? x0 : (float)((x0d + (pre + e[i])) + (double)S);
                    }
                    S = fmaf(1.f + c[i], S, c[i] * d[i]);
                }
                SB_FT(10);
                open_ = !(dmax < PLL_NEWTON_TOL);
                // a segment that is to be evaluated sample by sample takes the corrected guess as it is: the serial pass checks it
                if (exact_pending) { seq = true; exact_pending = false; }
            }
            SB_FT(11);
            // (cur / osc are those of the last evaluation: of the trajectory the iteration ended on)
            if (owner) { const float xe = (nxl >= 0.f && nxl < P32) ? nxl : 0.f; cy.x0 = xe; if (lastseg) st->pil_phase = xe; }
            if (lastseg && owner) {
                float oe = 0.f;
#pragma unroll
                for (int i = 0; i < FB_K; i++) if (i == il) oe = osc[i];
                st->pil_old = oe;
            }
        }
        SB_FT(12);
        SB_ARGS_FRESH(); SB_TICK1(2);

        // ================= lock detector (pilot-recover.cpp:62-80) =================
        if (SB_ABLATE & 8) {
#pragma unroll
            for (int i = 0; i < FB_K; i++) locked[i] = osc[i] > 0.f;
            all_locked = osc_in > 0.f;
        } else {
            const float lockA = 1.0f / 3000.0f;
            const double keep = 1.0 - (double)lockA;
            const float keepf = (float)keep;
            const float omega = T.pil_omega, romega = T.pil_omega_rcp;
            float xq[FB_K];
            {
                float old = osc_in;
#pragma unroll
                for (int i = 0; i < FB_K; i++) {
                    const float quadRef = fdiv_const(osc[i] - old, omega, romega);
                    old = osc[i];
                    xq[i] = (i < nv) ? lockA * (-quadRef * (5 * dem[i])) : 0.f;
                }
            }
            float Lt = 0.f;
#pragma unroll
            for (int i = 0; i < FB_K; i++) Lt = xq[i] + Lt * keepf;
            SB_FT(13);
            float lock_next;
            const int locked0 = cy.locked, stable0 = cy.stable;
            float lock = wg.decay_incoming2(Lt, cy.lock, load_decay(&dtab, DEC_LOCK, lane), &lock_next);
            bool lock_exact = false;
            if constexpr (EXACT) {
                // Handles that are asked for the reference's trajectories (pll_seq 1): the lock metric as the AFC above -- every thread runs its
                // samples from its incoming value in the reference's own expression (pilot-recover.cpp:62-66: f64 product and sum, rounded to f32
                // per sample), the misses of the end values against the next threads' incoming values are prefix-summed into corrections, until
                // every run ends on the next run's start: the chain from the exact cy.lock is the sequential one (the metric forgets a shift at
                // 1 / 3000 per sample).  The lock decisions then sit on the reference's metric, not within 1e-7 of it.
                if (P.pll_seq == 1) {
                    const float lock0 = cy.lock;
                    for (int pass = 0; pass < 8; pass++) {
                        float o = lock;
#pragma unroll
                        for (int i = 0; i < FB_K; i++) if (i < nv) o = (float)((double)xq[i] + (double)o * keep);
                        float po = dppf<0x138, 0xf>(0.f, o);
                        if (lane == 63) lds.wf[wg.sl][wg.wv][2] = o;
                        __syncthreads();
                        if (lane == 0) po = wg.wv ? lds.wf[wg.sl][(wg.wv + 3) & 3][2] : lock0;
                        wg.sl ^= 1;
                        const double d = (nv >= 1) ? (double)po - (double)lock : 0.0;
                        double total; bool any;
                        const double pre = wg.excl_add_d(d, &total, d != 0.0, &any);
                        if (!any) { lock_exact = true; break; }
                        lock = (float)((double)lock + (pre + d));
                    }
                }
            }
            bool hi[FB_K]; int lastf = -1; float lock_end = 0.f, lock_x = 0.f;
            float lock_min = 1.0f;
#pragma unroll
            for (int i = 0; i < FB_K; i++) {
                lock = (float)((double)xq[i] + (double)lock * keep);
                hi[i] = lock > 0.07f;
                lock_min = (i < nv) ? fminf(lock_min, lock) : lock_min;
                if (i < nv && !hi[i]) lastf = j0 + i;
                if (i == il) lock_end = lock;
                if (i == ix) lock_x = lock;
            }
            SB_FT(14);
            // locked[j] = no sample <= j below the threshold AND (locked before, or the run has lasted long enough)
            // (a pilot in lock has no such sample: one flag per wave through the barrier instead of the maximum's scan, which follows
            // behind a second barrier where a sample did fall below)
            int preF = -1, totF = -1;
            bool anynear;                                        // some sample's metric not above PLL_GUARD (NaN counts)
            {
                const int wlow = (__any(lastf >= 0) ? 1 : 0) | (__any(!(lock_min > PLL_GUARD)) ? 2 : 0);
                if (lane == 0) lds.wi[wg.sl][wg.wv][2] = wlow;
                __syncthreads();
                const int anyw = lds.wi[wg.sl][0][2] | lds.wi[wg.sl][1][2] | lds.wi[wg.sl][2][2] | lds.wi[wg.sl][3][2];
                const int anylow = anyw & 1;
                anynear = (anyw & 2) != 0;
                wg.sl ^= 1;
                if (anylow) { int cnt_dummy, tot_dummy; wg.excl_add_max_i(0, lastf, &cnt_dummy, &tot_dummy, &preF, &totF); }
            }
            {
                int F = preF;
#pragma unroll
                for (int i = 0; i < FB_K; i++) {
                    if (i < nv && !hi[i]) F = j0 + i;
                    const bool lk = (F < 0) && (locked0 != 0 || stable0 + (j0 + i) + 1 > (SINCOS_N >> 1));
                    locked[i] = lk;
                    if (i == ix && i < nv) {                                 // isPilotLocked (PilotPllLockStrength) :870-880
                        st->meta_locked = (stereo_possible && lk) ? 1 : 0;
                        st->meta_lock_strength = stereo_possible ? lock_x : 0.f;
                    }
                }
            }
            all_locked = totF < 0 && (locked0 != 0 || stable0 + 1 > (SINCOS_N >> 1));
            none_locked = locked0 == 0 && !(stable0 + w > (SINCOS_N >> 1));      // (not in lock in front of the segment and the run cannot get long enough inside it)
            int nl, ns;
            if (totF < 0) { nl = (locked0 != 0 || stable0 + w > (SINCOS_N >> 1)) ? 1 : 0;
                            ns = locked0 ? stable0 : (stable0 + w < (SINCOS_N >> 1) + 1 ? stable0 + w : (SINCOS_N >> 1) + 1); }
            else { nl = 0; ns = w - 1 - totF; }
            const int nok = (nl && !anynear) ? 1 : 0;
            newton_ok_next = nok != 0;
            if (lastseg && owner) { st->pil_lock = lock_end; st->pil_locked = nl; st->pil_stable = ns; st->pll_newton_ok = nok; }
            if (tid == 0) { cy.locked = nl; cy.stable = ns; cy.newton_ok = nok; if (!lock_exact) cy.lock = lock_next; }
            if (lock_exact && owner) cy.lock = lock_end;             // (the chain's own end value: read again behind the next segment's barriers)
        }
        SB_FT(15);
        if (PART == 1 || B.rows_on) {   // the second kernel's input; scope taps and the inputs of the RDS path (the whole kernel: where somebody wants them): channel-major rows of this call
            const size_t lrow = (size_t)ch * B.lin_rows + seg0 + j0;
            float *wd = B.w_dem + lrow, *wc = B.w_cur + lrow;
#pragma unroll
            for (int i = 0; i < FB_K; i++) if (i < nv) { wd[i] = dem[i]; wc[i] = cur[i]; }
        }
        SB_FT(16);
        SB_ARGS_FRESH(); SB_TICK1(3);
        if constexpr (PART == 1) {
            // hand-over to the second kernel: lock flags of this thread's samples (bits 0 .. 5) and of the whole segment (bit 7); the next
            // segment's ring entries are requested here (the whole kernel does that under its de-emphasis)
            unsigned m = (all_locked ? 0x80u : 0u) | (none_locked ? 0x40u : 0u);
#pragma unroll
            for (int i = 0; i < FB_K; i++) m |= locked[i] ? (1u << i) : 0u;
            *lockm = (uint8_t)m;
            if (!lastseg && !special) { const int wn = (nj - seg0 - FB_W) < FB_W ? (nj - seg0 - FB_W) : FB_W; fetch(j0, seg0 + FB_W, wn, zn); }
        }
        } else {
            // second kernel: demodulator output, pilot phase and lock flags as the first one left them
            const size_t lrow = (size_t)ch * B.lin_rows + seg0 + j0;
            const float *wd = B.w_dem + lrow, *wc = B.w_cur + lrow;
            const unsigned m = *lockm;
#pragma unroll
            for (int i = 0; i < FB_K; i++) { dem[i] = (i < nv) ? wd[i] : 0.f; cur[i] = (i < nv) ? wc[i] : 0.f; locked[i] = ((m >> i) & 1u) != 0; }
            all_locked = (m & 0x80u) != 0; none_locked = (m & 0x40u) != 0;
        }
        if constexpr (PART != 1) {

        // ================= PSS errors of the calls this segment can make: err[m] = Re (y) Im (y), y = low-pass of the s ring
        // (stereo-separation.cpp:60-83), m = call index within the segment, into er =================
        const int calls_before = cy.calls;
        float2 *sring = B.sring + (size_t)ch * (G.sring_mask + 1);
        const int smask = G.sring_mask;
        // (a segment without a sample in lock makes no process_sample call when autoMono is on, :704: the low-pass has nobody to serve --
        // what unlocked channels, which pay for the sample-by-sample PLL, save here)
        if (pss_on && !(auto_mono && none_locked)) {
            const int64_t i0 = pss_count0 + calls_before;                            // call index of the segment's first output
            float2 a[8];
            const int64_t first = i0 - (PSS_DELAY + PSS_TAPS - 1);                   // s index of window entry 0
            if (FAST && first >= 0) {          // (the usual case: ring positions in 32 bits, only the last of a thread's eight entries can be padding)
                const int r0 = (int)(first & smask) + tid;
#pragma unroll
                for (int p = 0; p < 7; p++) a[p] = sring[(r0 + FB_T * p) & smask];
                static_assert(FB_T * 7 < FB_W + PSS_TAPS - 1 && FB_W + PSS_TAPS - 1 <= FB_T * 8, "which window entries are padding");
                a[7] = (tid < FB_W + PSS_TAPS - 1 - FB_T * 7) ? sring[(r0 + FB_T * 7) & smask] : make_float2(0.f, 0.f);
            } else {
#pragma unroll
                for (int p = 0; p < 8; p++) {      // window entry n <-> s index i0 - (1753 + 294) + n; entries past the segment's need are padding
                    const int n = tid + FB_T * p;
                    const int64_t idx = first + n;
                    a[p] = (n < w + PSS_TAPS - 1 && idx >= 0) ? sring[idx & smask] : make_float2(0.f, 0.f);
                }
            }
            // (the window's loads are in flight before the first use of what the second kernel loaded at the segment's top -- one trip to
            // memory for both instead of one after the other)
#pragma unroll
            for (int i = 0; i < FB_K; i += 2) {
                *reinterpret_cast<float2 *>(&park_dem[j0 + i]) = make_float2(dem[i], dem[i + 1]);
                *reinterpret_cast<float2 *>(&park_cur[j0 + i]) = make_float2(cur[i], cur[i + 1]);
            }
            SB_FT(17); SB_FTW(18);
            fftc::convolve(tid, a, X, T.fft_w, T.pss_hs);
            SB_FT(19);
            __syncthreads();                   // (er overlays the buffer the last stage was read from)
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int m = tid + FB_T * p - (PSS_TAPS - 1);
                if (m >= 0 && m < FB_W) er[m] = a[p].x * a[p].y;
            }
#pragma unroll
            for (int i = 0; i < FB_K; i += 2) {
                const float2 d2 = *reinterpret_cast<const float2 *>(&park_dem[j0 + i]), c2 = *reinterpret_cast<const float2 *>(&park_cur[j0 + i]);
                dem[i] = d2.x; dem[i + 1] = d2.y; cur[i] = c2.x; cur[i + 1] = c2.y;
            }
        }
        __syncthreads();
        SB_FT(20);
        SB_ARGS_FRESH(); SB_TICK1(4);

        // ================= the PSS call index of every sample (fm-processor.cpp:704-718) =================
        int tag[FB_K];
        int ncalls;
        int firstU = -0x7fffffff - 1, firstZ = -0x7fffffff - 1;              // first unlocked sample / first stereo sample without PSS, as maxima of negated indices
        if (all_locked) {
            // (the usual case: no scan, no reduction)
#pragma unroll
            for (int i = 0; i < FB_K; i++) tag[i] = stereo_possible ? (pss_active ? ((i < nv) ? j0 + i : -2) : -1) : -2;
            ncalls = pss_on ? w : 0;
            if (stereo_possible && !pss_active) firstZ = 0;
        } else {
            int ncall_t = 0;
#pragma unroll
            for (int i = 0; i < FB_K; i++) {
                const bool branch = stereo_possible && (locked[i] || !auto_mono);
                tag[i] = branch ? (pss_active ? 0 : -1) : -2;
                ncall_t += (i < nv && branch && pss_active) ? 1 : 0;
            }
            int preC, dm1, dm2;
            wg.excl_add_max_i(ncall_t, 0, &preC, &ncalls, &dm1, &dm2);
            int c = preC;
#pragma unroll
            for (int i = 0; i < FB_K; i++) if (tag[i] == 0) { tag[i] = (i < nv) ? c : -2; c += (i < nv) ? 1 : 0; }    // index of the call within the segment
            int anyl = 0, alll = 0;
#pragma unroll
            for (int i = 0; i < FB_K; i++) if (i < nv) {
                if (!locked[i]) { const int v = -(j0 + i); firstU = v > firstU ? v : firstU; }
                if (tag[i] == -1) { const int v = -(j0 + i); firstZ = v > firstZ ? v : firstZ; }
            }
            wg.reduce_max4(firstU, firstZ, anyl, alll);
        }
        // (every thread read cy.calls in front of the barrier above; the next segment reads it again at its very top -- in the second
        // kernel with no barrier in front of that read, so the count is stored HERE, with the integrator's barriers still to come, and
        // not behind the segment's last barrier, where a wave that runs ahead into the next segment could still see the old one)
        if (tid == 0) cy.calls = calls_before + ncalls;

        SB_FT(21);
        // ================= PSS integrator (stereo-separation.cpp:84-109, fm-processor.cpp:699-718) =================
        float used[FB_K];                                            // pilotDelayPSS as used by each sample
        {
            const PssSt s = cy.ps;
            const float alpha = T.pss_alpha, la = T.pss_lock_alpha, keep = 1.0f - la;
            const float c4 = 0.785398185253143310546875f;
            float err[FB_K];
#pragma unroll
            for (int i = 0; i < FB_K; i++) err[i] = (pss_on && tag[i] >= 0) ? er[tag[i]] : 0.f;
            const bool steady = pss_on && all_locked && ((s.minimized ? s.unlock_cnt : s.lock_cnt) + w <= 3 * SINCOS_N);
            const bool nocall = ncalls == 0;
            PssSt e = s;                                             // state behind the segment
            if (steady) {
                const bool mz = s.minimized;
                float xa[FB_K], er10[FB_K];
#pragma unroll
                for (int i = 0; i < FB_K; i++) { er10[i] = mz ? err[i] : err[i] * 10.0f; xa[i] = alpha * er10[i]; }
                // accPhaseShift: the exact f32 trajectory a[j] = value in front of sample j.  The increments are often below half an
                // ulp of the accumulator and must be absorbed as the reference absorbs them: d[j] = fl (a[j] + xa[j]) - a[j] depends
                // on a[j] only through its binade, the clamp and ties, so d evaluated at the segment's first value, summed in f64,
                // is normally already the trajectory -- verified by evaluating d again at the values found (no second scan; the flag
                // rides with the next scan's barrier); a segment where it is not iterates to the fixed point as before.
                float a[FB_K];
                const double a0 = (double)s.acc;
                float aend;
                bool changed = false;
                {
                    double d1[FB_K], ex[FB_K], tot = 0.0;
#pragma unroll
                    for (int i = 0; i < FB_K; i++) {
                        const float na = fminf(fmaxf(s.acc + xa[i], -c4), c4);
                        d1[i] = (i < nv) ? (double)na - (double)s.acc : 0.0;
                        ex[i] = tot; tot += d1[i];
                    }
                    double total; bool any;
                    const double pre = wg.excl_add_d(tot, &total, false, &any);
#pragma unroll
                    for (int i = 0; i < FB_K; i++) {
                        const double exact = a0 + (pre + ex[i]);
                        a[i] = (float)exact;
                        const float na = fminf(fmaxf(a[i] + xa[i], -c4), c4);
                        // the value found must reproduce itself: representable, and the same increment from it
                        changed = changed || (i < nv && ((double)a[i] != exact || (double)na - (double)a[i] != d1[i]));
                    }
                    aend = (float)(a0 + total);
                    changed = changed || (double)aend != a0 + total;
                }
                SB_FT(22);
                // mean_error (1 / rate smoothing) and the "minimised" bookkeeping in closed form
                const DecayW dw = load_decay(&dtab, DEC_PSSMEAN, lane);
                float Lt = 0.f;
#pragma unroll
                for (int i = 0; i < FB_K; i++) Lt = ((i < nv) ? la * er10[i] : 0.f) + Lt * keep;
                float mean_next;
                float mean;
                {   // (decay_incoming2 with the integrator's flag riding along)
                    const float Z = wscan_decay(Lt, dw);
                    const int wch = __any(changed) ? 1 : 0;
                    if (lane == 63) lds.wf[wg.sl][wg.wv][0] = Z;
                    if (lane == 0) lds.wi[wg.sl][wg.wv][0] = wch;
                    __syncthreads();
                    float Cc = s.mean, Cin = s.mean;
#pragma unroll
                    for (int v = 0; v < 4; v++) { Cc = fmaf(Cc, dw.d64, lds.wf[wg.sl][v][0]); Cin = (v + 1 == wg.wv) ? Cc : Cin; }
                    changed = (lds.wi[wg.sl][0][0] | lds.wi[wg.sl][1][0] | lds.wi[wg.sl][2][0] | lds.wi[wg.sl][3][0]) != 0;
                    wg.sl ^= 1;
                    mean_next = Cc;
                    mean = fmaf(dw.dl, Cin, lane_prev_f(Z, 0.f));
                }
                SB_FT(23);
                int rounds = 1;
                if (changed) {
                    for (int it = 0; ; it++) {
                        double d[FB_K], ex[FB_K], tot = 0.0;
#pragma unroll
                        for (int i = 0; i < FB_K; i++) {
                            const float na = fminf(fmaxf(a[i] + xa[i], -c4), c4);
                            d[i] = (i < nv) ? (double)na - (double)a[i] : 0.0;
                            ex[i] = tot; tot += d[i];
                        }
                        double total; bool any;
                        const double pre = wg.excl_add_d(tot, &total, false, &any);
                        bool ch2 = false;
#pragma unroll
                        for (int i = 0; i < FB_K; i++) {
                            const float na = (float)(a0 + (pre + ex[i]));
                            ch2 = ch2 || (i < nv && __float_as_int(na) != __float_as_int(a[i]));
                            a[i] = na;
                        }
                        aend = (float)(a0 + total);
                        if (lane == 0) lds.wi[wg.sl][wg.wv][0] = 0;
                        if (__any(ch2) && lane == 0) lds.wi[wg.sl][wg.wv][0] = 1;
                        __syncthreads();
                        const int anych = lds.wi[wg.sl][0][0] | lds.wi[wg.sl][1][0] | lds.wi[wg.sl][2][0] | lds.wi[wg.sl][3][0];
                        wg.sl ^= 1;
                        rounds++;
                        if (!anych || it == PSS_MAX_ROUNDS - 1) break;
                    }
                }
                SB_FT(24);
                if (B.dbg && tid == 0) { B.dbg[(size_t)ch * DBG_SLOTS + 9] += rounds; B.dbg[(size_t)ch * DBG_SLOTS + 12] += 1; }
#pragma unroll
                for (int i = 0; i < FB_K; i++) used[i] = (j0 + i == 0) ? s.pdp : a[i];
                int lastS = -1, lastN = -1, d3 = 0, d4 = 0; float mean_end = 0.f;
#pragma unroll
                for (int i = 0; i < FB_K; i++) {
                    mean = la * er10[i] + mean * keep;
                    if (i < nv) { if (fabsf(mean) < 0.001f) lastS = j0 + i; else lastN = j0 + i; }
                    if (i == il) mean_end = mean;
                    if (i == ix && i < nv) meta_snapshot(st, P, fminf(fmaxf(a[i] + xa[i], -c4), c4), mean, mz, true);   // (mz: no flip inside a steady segment)
                }
                SB_FT(25);
                wg.reduce_max4(lastS, lastN, d3, d4);
                // (reduce_max4's barrier: everybody has its copy of the state in front of the segment; the two counters are only carried
                // by thread 0, which reads them again here instead of every thread keeping them in registers through the phase)
                if (tid == 0) {
                    const int lc0 = cy.ps.lock_cnt, uc0 = cy.ps.unlock_cnt;
                    const bool all_small = lastN < 0, any_small = lastS >= 0;
                    int lc, uc;
                    if (mz) { lc = all_small ? lc0 : 0; uc = any_small ? (w - 1 - lastS) : uc0 + w; }
                    else { lc = all_small ? lc0 + w : (w - 1 - lastN); uc = any_small ? 0 : uc0; }
                    cy.ps.acc = aend; cy.ps.pdp = aend; cy.ps.minimized = mz ? 1 : 0; cy.ps.lock_cnt = lc; cy.ps.unlock_cnt = uc;
                }
                if (owner) cy.ps.mean = mean_end;
            } else if (nocall) {
                // nobody calls process_sample: an unlocked sample clears everything, a stereo sample without PSS clears pilotDelayPSS
                const int fu = (firstU == -0x7fffffff - 1) ? 0x7fffffff : -firstU, fz = (firstZ == -0x7fffffff - 1) ? 0x7fffffff : -firstZ;
#pragma unroll
                for (int i = 0; i < FB_K; i++) used[i] = (j0 + i >= fu || j0 + i > fz) ? 0.f : s.pdp;
#pragma unroll
                for (int i = 0; i < FB_K; i++) if (i == ix && i < nv) {
                    const bool cleared = j0 + i >= fu;
                    meta_snapshot(st, P, (cleared || j0 + i >= fz) ? 0.f : s.pdp, cleared ? 0.f : s.mean, cleared ? false : s.minimized, locked[i]);
                }
                if (fu != 0x7fffffff) { e.pdp = 0.f; e.acc = 0.f; e.mean = 0.f; e.minimized = false; e.lock_cnt = 0; e.unlock_cnt = 0; }
                else if (fz != 0x7fffffff) e.pdp = 0.f;
                __syncthreads();                                     // (everybody has its copy of the state in front of the segment)
                if (tid == 0) cy.ps = e;
            } else {
                // replay (lock transitions inside a PSS segment, a counter within a segment of its 3 s threshold)
                __syncthreads();                                     // (every thread has taken its errors out of er)
#pragma unroll
                for (int i = 0; i < FB_K; i++) if (i < nv) { pk[j0 + i] = ((tag[i] + 2) << 1) | (locked[i] ? 1 : 0); er[j0 + i] = err[i]; }
                __syncthreads();
                if (B.dbg && tid == 0) B.dbg[(size_t)ch * DBG_SLOTS + 10] += 1;
                if (tid == 0) {
                    PssSt r = s;
                    const int jxs = jx - seg0;
                    for (int j = 0; j < w; j++) {
                        const int p = pk[j];
                        er[j] = pss_step(r, alpha, la, keep, (p & 1) != 0, (p >> 1) - 2, er[j]);
                        if (j == jxs) meta_snapshot(st, P, r.pdp, r.mean, r.minimized, (p & 1) != 0);
                    }
                    cy.ps = r;
                }
                __syncthreads();
#pragma unroll
                for (int i = 0; i < FB_K; i++) used[i] = (i < nv) ? er[j0 + i] : 0.f;
            }
        }
        SB_FT(26);
        SB_ARGS_FRESH(); SB_TICK1(5);

        // ================= 38 kHz mix, PSS input, stereo matrix (fm-processor.cpp:707-730, 517-549) =================
        float2 x[FB_K];
        {
            constexpr double INV2PI = 1.0 / FMX_2PI;
            const int ssel = P.sound_sel; const float pano = (P.fm_mode == 1) ? P.panorama : 1.0f;
            const int icl = (int)((pss_count0 + calls_before) & smask);   // ring position of the segment's first call
            const float P32 = 6.2831855f, C32 = T.wrap32_c;
            const bool wrap_ok = T.wrap32_ok != 0;
            float sumv[FB_K], diffv[FB_K];
            const bool wide = __any(!all_in(cur, P32 + 0.5f)) || !wrap_ok;     // (some phase of the wave outside [0, 2 pi + 0.5): the general PI_Constrain)
#pragma unroll
            for (int i = 0; i < FB_K; i++) {
                // phaseforLRDiff fm-processor.cpp:707-714: 2 (currentPilotPhase + pi/4) - pilotDelayPSS lies in (0, 4 pi + 2.4), so the
                // "< -2 pi" branch never runs and fmod (., 2 pi) is the fraction of the turn count
                float cc = (cur[i] < P32) ? cur[i] : (cur[i] - P32) + C32;                   // PI_Constrain of [0, 2 pi + 0.7), see the pilot PLL
                if (wide) cc = pi_constrain(cur[i]);
                const float p = (float)(2 * ((double)cc + FMX_PI_4) - (double)used[i]);      // (+ PILOTTESTDELAY = 0, fm-processor.cpp:42: x + 0 = x for x > 0)
                const double u = __builtin_amdgcn_fract((double)p * INV2PI);
                // SinCos::getComplex sincos.cpp:93-97 (the fraction of a turn is at most 1 - 2^-53, v_fract_f64's own bound: its product with 192000 rounds
                // below 192000, the index needs no clamp)
                const int idx = (int)(u * (double)SINCOS_N);
                float2 e;
                sincos_idx_hw_bits(idx, &e.y, &e.x);
                float dif = 0.f;
                if (tag[i] != -2) {
                    if (tag[i] >= 0 && i < nv) sring[(icl + tag[i]) & smask] = make_float2(e.x * dem[i], e.y * dem[i]);
                    const float lut = (ssel == 6) ? e.y : e.x;       // S_LEFTminusRIGHT_Test mixes with the sine
                    dif = 2.0f * (lut * dem[i]);                     // (float)(2.0 * lut * demod): one rounding of the exact product either way
                }
                sumv[i] = dem[i]; diffv[i] = dif;
            }
            SB_FT(27);
            if (B.w_diff) {   // scope tap (fmx_get_tap; FMX_P_SCOPE_TAPS: a display feed that large batches do not keep): channel-major rows of this call
                float *wf = B.w_diff + (size_t)ch * B.lin_rows + seg0 + j0;
#pragma unroll
                for (int i = 0; i < FB_K; i++) if (i < nv) wf[i] = diffv[i];
            }
#pragma unroll
            for (int i = 0; i < FB_K; i++) {
                const float sumLR = sumv[i];
                const float dw = diffv[i] * pano;
                const float left = sumLR + dw, right = sumLR - dw;
                float2 o;
                switch (ssel) {
                default:
                case 0: o = make_float2(left, right); break;
                case 1: o = make_float2(right, left); break;
                case 2: o = make_float2(left, left); break;
                case 3: o = make_float2(right, right); break;
                case 4: o = make_float2(sumLR, sumLR); break;
                case 5: case 6: o = make_float2(dw, dw); break;
                }
                x[i] = (i < nv) ? o : make_float2(0.f, 0.f);
            }
        }
        SB_FT(28);
        SB_ARGS_FRESH(); SB_TICK1(6);

        // ================= de-emphasis (fm-processor.cpp:594-595) into the d ring =================
        // (the next segment's ring entries are requested here: they land under the de-emphasis, and are not in the way of the
        // register-hungry phases above)
        if (PART == 0 && !lastseg && !special) { const int wn = (nj - seg0 - FB_W) < FB_W ? (nj - seg0 - FB_W) : FB_W; fetch(j0, seg0 + FB_W, wn, zn); }
        SB_FT(29);
        if (G.no_deemph) {
            // (handles whose audio low-pass runs as the reference's block machine, fmx_ola.hip: the filter comes first, :589-595 -- the pair goes
            // out as it is, deemph_kernel owns the de-emphasis state)
            const int dmask = G.dring_mask;
            float2 *dr = B.dring + (size_t)ch * (dmask + 1);
            const int jb = (int)((G.J0 + seg0 + j0) & dmask);
#pragma unroll
            for (int i = 0; i < FB_K; i++) if (i < nv) dr[(jb + i) & dmask] = x[i];
        } else
        {
            const float a = P.deemph_alpha;
            const DecayW dw = load_decay(&dtab, DEC_DEEMPH, lane);
            float Ll = 0.f, Lr = 0.f;
#pragma unroll
            for (int i = 0; i < FB_K; i++) { Ll = (x[i].x - Ll) * a + Ll; Lr = (x[i].y - Lr) * a + Lr; }
            SB_FT(30);
            float yl, yr;
            {   // (both channels behind one barrier)
                const float Zl = wscan_decay(Ll, dw), Zr = wscan_decay(Lr, dw);
                if (lane == 63) { lds.wf[wg.sl][wg.wv][0] = Zl; lds.wf[wg.sl][wg.wv][1] = Zr; }
                const float cl0 = cy.de_l, cr0 = cy.de_r;
                __syncthreads();
                float Cl = cl0, Cr = cr0;
                for (int v = 0; v < wg.wv; v++) { Cl = fmaf(Cl, dw.d64, lds.wf[wg.sl][v][0]); Cr = fmaf(Cr, dw.d64, lds.wf[wg.sl][v][1]); }
                wg.sl ^= 1;
                yl = fmaf(dw.dl, Cl, lane_prev_f(Zl, 0.f)); yr = fmaf(dw.dl, Cr, lane_prev_f(Zr, 0.f));
            }
            SB_FT(31);
            const int dmask = G.dring_mask;
            float2 *dr = B.dring + (size_t)ch * (dmask + 1);
            const int jb = (int)((G.J0 + seg0 + j0) & dmask);
            float el = 0.f, er_ = 0.f;
#pragma unroll
            for (int i = 0; i < FB_K; i++) {
                yl = (x[i].x - yl) * a + yl;
                yr = (x[i].y - yr) * a + yr;
                if (i < nv) dr[(jb + i) & dmask] = make_float2(yl, yr);
                if (i == il) { el = yl; er_ = yr; }
            }
            // the state behind the segment's last sample, as its owner computed it (read again behind the next segment's barriers)
            if (owner) { cy.de_l = el; cy.de_r = er_; }
        }
        SB_FT(32);
        SB_ARGS_FRESH(); SB_TICK1(7);
        }   // PART != 1
    };
    for (int seg0 = 0; seg0 < nj; seg0 += FB_W) {
        const bool fast = (nj - seg0 >= FB_W) && !((unsigned)(jx - seg0) < (unsigned)FB_W) && (callJ0 + seg0 >= 2);
        // (the pilot PLL of this segment on the sequential trajectory?  pll_seq 1: always; 0: unless the pilot is comfortably in lock, PLL_GUARD)
        const bool exact = PART != 2 && (P.pll_seq == 1 || (P.pll_seq == 0 && stereo_possible && !newton_ok_next) || afc_big_next);
        if (fast) { if (exact) segment(std::true_type{}, std::true_type{}, seg0); else segment(std::true_type{}, std::false_type{}, seg0); }
        else { if (exact) segment(std::false_type{}, std::true_type{}, seg0); else segment(std::false_type{}, std::false_type{}, seg0); }
    }
    // ================= bookkeeping behind the call =================
    __syncthreads();
    if (PART != 1 && threadIdx.x == 0) {
        const PssSt ps = cy.ps;
        st->pss_acc = ps.acc; st->pss_mean = ps.mean; st->pilot_delay_pss = ps.pdp;
        st->pss_lock_cnt = ps.lock_cnt; st->pss_unlock_cnt = ps.unlock_cnt; st->pss_minimized = ps.minimized ? 1 : 0;
        if (!G.no_deemph) { st->de_l = cy.de_l; st->de_r = cy.de_r; }
        // metaData: the snapshot behind sample fmRate / 2 - myCount of the call was stored sample-exactly above; the RF DC level moves
        // by 1e-7 of its distance per input sample and is taken here, at the end of that call
        int cnt = my_count0 + nj;
        if (cnt > (SINCOS_N >> 1)) {
            const float dcabs = (float)sqrt((double)st->dc_re * (double)st->dc_re + (double)st->dc_im * (double)st->dc_im);
            st->meta_dc_rf = P.dc_remove ? 20 * log10f(dcabs + 1.0f / 32768) : (float)-99.99;
            cnt -= (SINCOS_N >> 1) + 1;
        }
        st->my_count = cnt;
        st->pss_count = pss_count0 + cy.calls;                   // the PSS filter time base advances by this call's process_sample calls
        st->pss_call_total = 0;
    }
    constexpr int SB_FASTFLAG = 2;
    SB_TICK1(8);
#ifdef SB_PHASE_CYCLES
    if (dbg_on) for (int k = 0; k < 9; k++) B.dbg[(size_t)ch * DBG_SLOTS + 16 + k] += dbg_acc[k];
#endif
#ifdef SB_FINE_TICKS
    __syncthreads();
    if (ft_on) for (int k = 0; k < 64; k++) B.dbg[(size_t)ch * DBG_SLOTS + 32 + k] += ft_acc[k];
#endif
    if (B.dbg && threadIdx.x == 0) {
        B.dbg[(size_t)ch * DBG_SLOTS + 13] = __float_as_uint(cy.x0); B.dbg[(size_t)ch * DBG_SLOTS + 14] = __float_as_uint(cy.lock);   // (diagnostics: state behind the call)
    }
}
#undef T
#undef B
#undef G

void launch_demod_fused(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, int C, hipStream_t s, const PrepassStreams *ps) {
    if (G.J1 - G.J0 <= 0) return;
    if (B.w_iq) launch_demod_prepass(T, B, G, C, s, ps);     // some channel has (had) a PLL / AM decoder or a squelch: fmx_demod.hip
    StageBArgs A; A.T = T; A.B = B; A.G = G; A.C = C;
    // One kernel or two?  Per channel both forms cost the same (measured, 256 ... 4096 channels: the kernel is bound by instruction
    // issue, a fourth workgroup per CU adds nothing), what differs is the tail: the whole kernel runs 3 workgroups per CU, its halves 4,
    // and a batch that does not fill the last round of either leaves CUs idle.  The form whose rounds waste less wins; a tie goes to
    // the single launch.  (4096 channels on 256 CUs: 5.33 rounds of 768 against 4 of 1024 -- 2.05 against 1.91 ms.)
    // FMX_P_STAGEB_FORM (tests) or the environment (FMX_STAGEB_SPLIT=0 / 1: A/B runs of the bench) force either form.
    // (the handle's own CU count rides in CallGeom: handles on different devices -- another SKU, another partition mode -- each do their own
    // round arithmetic)
    const int env = env_switches().stageb_split;
    const int force = G.stageb_form ? G.stageb_form - 1 : env;
    const int cus = G.n_cus > 0 ? G.n_cus : 256;
    const long whole = (long)((C + SB_WG_PER_SIMD * cus - 1) / (SB_WG_PER_SIMD * cus)) * SB_WG_PER_SIMD * 100;
    const long halves = (long)((C + 4 * cus - 1) / (4 * cus)) * 4 * 102;          // (two launches, the hand-over through HBM: 2 %)
    // (a batch that keeps no scope taps and decodes no RDS: the whole kernel leaves the rows unwritten -- 0.63 GB per call at 4096 channels, and writes
    // are the expensive direction on this GPU -- and is then the faster form where the halves were: 1.60 against 1.69 ms)
    const bool split = force >= 0 ? force != 0 : (B.rows_on ? halves < whole : (env_switches().rows_off_split ? halves < whole : false));
    const unsigned grid = (unsigned)(G.ch_count > 0 ? G.ch_count : C);       // (a launch for some of the channels: CallGeom::ch0)
    if (split) {
        hipLaunchKernelGGL(stageb_kernel<1>, dim3(grid), dim3(FB_T), 0, s, A); FMX_LAUNCHED();
        hipLaunchKernelGGL(stageb_kernel<2>, dim3(grid), dim3(FB_T), 0, s, A); FMX_LAUNCHED();
    } else {
        hipLaunchKernelGGL(stageb_kernel<0>, dim3(grid), dim3(FB_T), 0, s, A); FMX_LAUNCHED();
    }
}

}  // namespace fmx
