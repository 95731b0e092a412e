// fmx_demod.hip -- stage B: everything that runs at fmRate (192 kS/s) between the decimators and
// the audio low-pass.  COMPILED WITH -ffp-contract=off: the LUT index expressions and feedback
// loops below are evaluated in exactly the f32/f64 types the reference's C++ uses, because a
// one-ulp difference can flip a table index.
//
// Replaces per channel:
//   fm_Demodulator::demodulate        fm-demodulator.cpp:111-205 (+ compAtan Xtan2.cpp:56-100, pllC.cpp:67-90)
//   pilotRecovery::getPilotPhase      pilot-recover.cpp:54-83
//   process_signal_with_rds (stereo)  fm-processor.cpp:689-730
//   PerfectStereoSeparation           stereo-separation.cpp:60-109 (its overlap-add low-pass is a
//                                     295-tap direct FIR with the same 1753-sample latency)
//   L/R matrix, selector              fm-processor.cpp:517-549
//   de-emphasis, gain                 fm-processor.cpp:594-595, 303-306
//
// MI355X design.  The stage is a chain of small kernels over a call's fm-rate samples, of two kinds:
//   * time-parallel kernels (one thread per (channel, sample)): limiter + LUT discriminator,
//     the PSS low-pass, the 38 kHz mix + matrix;
//   * recurrence kernels (AFC, pilot PLL, lock detector, PSS integrator, de-emphasis): ONE LANE PER
//     CHANNEL, 64 channels per wavefront, all lanes stepping through time together.  These loops
//     cannot be parallelised in time (non-linear feedback through LUT indices); their parallelism
//     is the channel count, and every lane is busy.  Work arrays between the kernels are
//     sample-major [sample][channel] so both kinds of kernel access them coalesced.
// The only feedback path with a lag is the PSS loop (error -> integrator -> 38 kHz phase -> mix ->
// 1753-sample low-pass -> error), which is why a call is processed in chunks.  Two schedules exist (launch_demod): the
// persistent one -- all recurrences of a call in ONE kernel on its own CUs, progress words in device memory -- and the
// event-driven one (a five-stream software pipeline of per-chunk kernels).
#include "fmx_internal.h"
#include "fmx_demod_math.h"

namespace fmx {

thread_local hipError_t g_launch_err = hipSuccess;

// How a time-parallel kernel meets the persistent recurrence kernel (launch_demod_persistent); all null / zero on the
// event-driven path.  `sig`: completion word of the kernel in FRONT of this one on the stream -- stored by this kernel's
// first block, because the kernel boundary in between has made that kernel's stores visible device-wide (a release
// fence per block would write the XCD's L2 back thousands of times).  `gate`: progress words [group] of the recurrence
// role this kernel consumes; a block starts once its own group has reached `need`.
struct TSync { const int *gate; int need; int *sig; int sigv; int *abort_flag; int skip; };
__device__ __forceinline__ bool tsync_enter(const TSync &Y, int group) {
    if (Y.sig != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        __hip_atomic_store(Y.sig, Y.sigv, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (Y.gate == nullptr) return true;
    __shared__ int ok;
    if (threadIdx.x == 0) {
        int spins = 0, good = 1;
        // (the words and the rows behind them were never in this XCD's L2 before the kernel started: no invalidate needed)
        while (__hip_atomic_load(Y.gate + group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < Y.need) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1 << 22) || __hip_atomic_load(Y.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                __hip_atomic_store(Y.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); good = 0; break;
            }
        }
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

// =================================================================================================
// B1  limiter + memoryless discriminator   (time-parallel; 64 samples x 64 channels per block)
// =================================================================================================
constexpr int DISC_ROWS = 64, DISC_CH = 16;             // one block: four work-array tile rows of sixteen channels
__device__ __forceinline__ void disc_body(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int64_t row0, int nrows, const int bid_x, const int bid_y) {
    // One block = 64 samples x 16 channels: the ring reads are 528-byte runs per channel (the ring is channel-major), the
    // work-array stores 1 KB runs (16 rows x 16 channels of a tile row are contiguous).
    const int CP = G.pitch;
    __shared__ float2 sLIM[DISC_CH][DISC_ROWS + 3];   // limited samples of rows r0-2 .. r0+63 (each is used by up to three outputs)
    __shared__ float sABS[DISC_CH][DISC_ROWS + 3];    // |z| of the same samples (AM decoder, level squelch)
    __shared__ int sDelay[DISC_CH], sDec[DISC_CH];
    const int tid = threadIdx.x;
    const int64_t nj = row0 + nrows;
    const int64_t r0 = row0 + (int64_t)bid_x * DISC_ROWS;
    const int c0 = bid_y * DISC_CH;
    const int ring = G.ring_mask + 1;
    const bool want_iq = B.w_iq != nullptr;
    if (tid < DISC_CH) {
        const int ch = c0 + tid;
        sDelay[tid] = ch < C ? T.front_sets[B.params[ch].front_set].delay_fm : 0;
        sDec[tid] = ch < C ? B.params[ch].decoder : 0;
    }
    __syncthreads();
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
        if (ch < C && r < nj) {
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
__global__ __launch_bounds__(256) void disc_kernel(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int64_t row0, int nrows, TSync Y) {
    if (!tsync_enter(Y, (int)blockIdx.y >> 2) || Y.skip) return;
    disc_body(T, B, G, C, row0, nrows, (int)blockIdx.x, (int)blockIdx.y);
}

// =================================================================================================
// B2..B4  the per-sample recurrences up to the pilot lock   [lane per channel, 64 channels per wave]
//   B2 afc_kernel   AFC + scaling            fm-demodulator.cpp:197-198  (pllC.cpp:67-90 when decoder == PLL)
//   B3 pll_kernel   pilot PLL                pilot-recover.cpp:54-61
//   B4 lock_kernel  lock detector            pilot-recover.cpp:62-80
//                   PSS call index (tag)     fm-processor.cpp:704-705,716-718 (which samples call process_sample)
// Three kernels rather than one: each is a dependent chain whose time is set by instruction latency, and as
// separate stages of the chunk pipeline (launch_demod) they run concurrently on different wavefronts.
// Work arrays are read a batch ahead into registers (global latency off the dependent chain).
// =================================================================================================
// Batch size of the register-prefetched work-array rows.  A wave can have at most 63 vector-memory operations in flight
// (6-bit vmcnt) and on gfx9 stores count too: the loads of the next batch are issued right behind the stores of the
// last one, so (loads + stores) per batch must stay below that or every batch stalls for a store round trip.
constexpr int SEQ_UB = WT;
// LDS of the recurrence bodies: a workgroup runs exactly one of them (also in the persistent kernel, where the role is
// fixed per workgroup), so they share one buffer -- the pilot PLL's factor tables, or the replay staging of the lock
// detector / PSS integrator.
constexpr int REC_LDS_BYTES = (TRIG2_A + TRIG2_APAD + TRIG2_B) * 16 > 2 * SEQ_UB * 64 * 4 ? (TRIG2_A + TRIG2_APAD + TRIG2_B) * 16 : 2 * SEQ_UB * 64 * 4;
__shared__ __attribute__((aligned(16))) char g_rec_lds[REC_LDS_BYTES];
// The recurrence kernels are a few wavefronts whose run time is pure instruction latency; when the time-parallel kernels
// of the other streams fill the same SIMDs, the issue arbiter must not make them wait: raise their wave priority.
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
template <bool PLLDEC>
__device__ __forceinline__ void afc_body(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int64_t rc0, int chunk_len, const int bid_x, const int bid_y) {
    constexpr int PDA = PLLDEC ? 2 : 4;
    const int CP = G.pitch;
    const int ch = bid_x * 64 + threadIdx.x;
    if (ch >= C) return;
    ChanState *st = B.state + ch;
    const int decoder = B.params[ch].decoder;
    const bool use_pll = PLLDEC && (decoder == 2), use_am = PLLDEC && (decoder == 1);
    // level squelch (squelch::do_level_squelch squelchClass.cpp:89-113, fm-processor.cpp:504-506): the carrier amplitude IIR
    // of the demodulator (fm-demodulator.cpp:130-131) against a threshold, re-evaluated every fmRate / 20 samples
    const bool lsq = PLLDEC && (B.params[ch].squelch_mode == 2);
    const float sq_thr = B.params[ch].squelch_thr;
    int sq_cnt = st->sq_count; bool sq_sup = st->sq_suppress != 0;
    float am = st->am_carr;
    // noise squelch (squelch::do_noise_squelch squelchClass.cpp:47-87): |high-pass 69.9 kHz| against |low-pass 70 kHz| of the
    // demodulator output, two order-20 Chebyshev cascades (Basic_IIR::Pass iir-filters.h:89-103, same f32 operation order)
    const bool nsq = PLLDEC && (B.params[ch].squelch_mode == 1) && T.nsq_coef != nullptr;
    const float nsq_thr = B.params[ch].squelch_nthr;
    // (the forty filter memories of a lane live in LDS, [memory][lane]: the AFC role has the wave's LDS buffer to itself, and
    // forty more live registers would push the PLL-decoder / AM / level-squelch paths of this body into scratch)
    float *nmem = reinterpret_cast<float *>(g_rec_lds) + threadIdx.x;
    float avg_hi = 0.f, avg_lo = 0.f;
    if (PLLDEC && nsq) {
        for (int k = 0; k < 4 * NSQ_QUADS; k++) nmem[64 * k] = (&st->sq_m[0][0][0])[k];
        avg_hi = st->sq_avg_hi; avg_lo = st->sq_avg_lo;
    }
    const float fmDcAlpha = 0.0001f, c1 = 1 - fmDcAlpha, K = T.K_FM, rK = T.K_FM_rcp;
    const double SC = T.sincos_C;
    float afc = st->fm_afc, nco_phase = st->nco_phase, incr = st->phase_incr;
    const size_t ro = widx(rc0, ch, CP);              // rc0 is a multiple of the tile height
    float *wd = B.w_dem + ro;
    const float2 *wiq = PLLDEC ? B.w_iq + ro : nullptr;
    auto demod1 = [&](float res, float2 sig) -> float {
        if (PLLDEC) {
            // |z| arrives in the demod array for the AM and PLL decoders, in the first half of the IQ array otherwise (disc_kernel)
            if (use_am || lsq) am = (1.0f - 0.0010f) * am + 0.0010f * (decoder <= 2 ? res : sig.x);      // am_carr_ampl, carrierAlpha (fm-demodulator.cpp:117,130-131)
            if (use_pll || use_am) {                 // pllC::do_pll pllC.cpp:67-90 (AM: on the unlimited sample, :222)
                const float2 nco = sc_complex(T.sincos, SC, nco_phase);
                const float dre = nco.x * sig.x - (-nco.y) * sig.y;      // conj(nco) * signal
                const float dim = nco.x * sig.y + (-nco.y) * sig.x;
                const float perr = lut_atan2(T.atan_ppy, dim, dre);
                incr = (1 - T.pll_beta) * perr + T.pll_beta * incr;
                if (incr < T.pll_lo || incr > T.pll_hi) incr = T.pll_center;
                nco_phase += incr;
                if ((double)nco_phase >= FMX_2PI) nco_phase = (float)fmod_2pi((double)nco_phase);
                else while (nco_phase < 0) nco_phase = (float)((double)nco_phase + FMX_2PI);
                if (use_am) {                        // decodeAM fm-demodulator.cpp:215-241
                    afc = c1 * afc + fmDcAlpha * incr;
                    const float gainLimit = 0.01f;
                    float r = (res - am) / (am < gainLimit ? gainLimit : am);
                    if (r > 1.0f) r = 1.0f; else if (r < -1.0f) r = -1.0f;
                    return r;
                }
                res = incr;
            }
        }
        afc = c1 * afc + fmDcAlpha * res;            // fm-demodulator.cpp:197
        return fdiv_const(20.0f * (res - afc) * 1.0f, K, rK);      // :198
    };
    auto step = [&](float res, float2 sig) -> float {
        float r = demod1(res, sig);
        if (PLLDEC && lsq) {
            if (++sq_cnt >= SINCOS_N / 20) {         // holdPeriod = fmRate / 20 (fm-processor.cpp:87)
                sq_cnt = 0;
                if (am < sq_thr - 0.000f) sq_sup = true;             // SQUELCH_HYSTERESIS_LSQ = 0
                else if (am >= sq_thr + 0.000f) sq_sup = false;
            }
            r = sq_sup ? r * 0.000f : r;             // LEVELREDUCTIONFACTOR = 0
        }
        return r;
    };
    // the noise squelch of one demodulator output (a pass of its own behind `step`, in rolled loops: inlined sixteen times
    // into the unrolled tile body it pushed every path of this body into scratch memory)
    auto nsq1 = [&](float r) __attribute__((always_inline)) -> float {
        float val[2];
#pragma unroll
        for (int f = 0; f < 2; f++) {
            const float *cf = T.nsq_coef + f * NSQ_QUADS * 4;
            float o = r * T.nsq_coef[2 * NSQ_QUADS * 4 + f];
#pragma unroll 1
            for (int i = 0; i < NSQ_QUADS; i++) {
                float *m = nmem + 64 * 2 * (f * NSQ_QUADS + i);               // (m1, m2) of this biquad
                const float rm1 = m[0], rm2 = m[64];
                const float w = o - rm1 * cf[4 * i + 2] - rm2 * cf[4 * i + 3];
                o = w + rm1 * cf[4 * i] + rm2 * cf[4 * i + 1];
                m[64] = rm1; m[0] = w;
            }
            val[f] = fabsf(o);
        }
        // decayingAverage squelchClass.cpp:40-45, weight = sampleRate / 100, evaluated in double
        const double k1 = 1.0 / (double)(float)(SINCOS_N / 100), k2 = 1.0 - k1;
        avg_hi = (float)((double)val[0] * k1 + (double)avg_hi * k2);
        avg_lo = (float)((double)val[1] * k1 + (double)avg_lo * k2);
        if (++sq_cnt >= SINCOS_N / 20) {
            sq_cnt = 0;
            if (nsq_thr < 0.001f) sq_sup = true;                                   // SQUELCH_HYSTERESIS_NSQ = 0.001
            else if (avg_hi < avg_lo * nsq_thr - 0.001f) sq_sup = false;
            else if (avg_hi >= avg_lo * nsq_thr + 0.001f) sq_sup = true;
        }
        return sq_sup ? r * 0.000f : r;
    };
    const bool nsq_wave = PLLDEC && __any(nsq);
    float *nsx = reinterpret_cast<float *>(g_rec_lds) + 64 * 4 * NSQ_QUADS + threadIdx.x;      // [16][lane] staging of a tile's outputs
    constexpr int UB = SEQ_UB;
    const int nfull = chunk_len / UB;
    const int TS = UB * CP;                                       // elements from one tile of this channel to the next
    // (prefetch registers as plain float arrays: loop-carried arrays of HIP's float2 struct end up in scratch memory)
    float nx[2][PDA][UB]; float nqa[PLLDEC ? 2 : 1][PLLDEC ? PDA : 1][UB], nqb[PLLDEC ? 2 : 1][PLLDEC ? PDA : 1][UB];
    tile_pipeline<PDA>(nfull,
        [&](int s, int u, int tl) __attribute__((always_inline)) {
            wld(nx[s][u], wd + tl * TS);
            if (PLLDEC) { const float *qp = reinterpret_cast<const float *>(wiq + tl * TS); wld(nqa[PLLDEC ? s : 0][PLLDEC ? u : 0], qp); wld(nqb[PLLDEC ? s : 0][PLLDEC ? u : 0], qp + UB); }
        },
        [&](int s, int u, int tb) __attribute__((always_inline)) {
            float x[UB]; float2 xq[UB];
#pragma unroll
            for (int k = 0; k < UB; k++) {
                x[k] = nx[s][u][k];
                const int ss = PLLDEC ? s : 0, uu = PLLDEC ? u : 0;
                xq[k] = !PLLDEC ? make_float2(0.f, 0.f)
                        : (k < UB / 2 ? make_float2(nqa[ss][uu][2 * k], nqa[ss][uu][2 * k + 1])
                                      : make_float2(nqb[ss][uu][(2 * k) % UB], nqb[ss][uu][(2 * k) % UB + 1]));
            }
#pragma unroll
            for (int k = 0; k < UB; k++) x[k] = step(x[k], xq[k]);
            if (PLLDEC && nsq_wave) {
#pragma unroll
                for (int k = 0; k < UB; k++) nsx[64 * k] = x[k];
                if (nsq) {
#pragma unroll 1
                    for (int k = 0; k < UB; k++) nsx[64 * k] = nsq1(nsx[64 * k]);
                }
#pragma unroll
                for (int k = 0; k < UB; k++) x[k] = nsx[64 * k];
            }
            wst(wd + tb * TS, x);
        });
    {
        float *wdt = wd + nfull * TS; const float2 *wiqt = PLLDEC ? wiq + nfull * TS : nullptr;
        for (int k = 0; k < chunk_len - nfull * UB; k++)          // ragged end of a call: rows of the last, partial tile
        {
            float r = step(wdt[k], PLLDEC ? wiqt[k] : make_float2(0.f, 0.f));
            if (PLLDEC && nsq) r = nsq1(r);
            wdt[k] = r;
        }
    }
    st->fm_afc = afc; st->nco_phase = nco_phase; st->phase_incr = incr; st->am_carr = am;
    if (PLLDEC) { st->sq_count = sq_cnt; st->sq_suppress = sq_sup ? 1 : 0; }
    if (PLLDEC && nsq) {
        st->sq_avg_hi = avg_hi; st->sq_avg_lo = avg_lo;
        for (int k = 0; k < 4 * NSQ_QUADS; k++) (&st->sq_m[0][0][0])[k] = nmem[64 * k];
    }
}
template <bool PLLDEC>
__global__ __launch_bounds__(64) void afc_kernel(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int64_t rc0, int chunk_len) {
    FMX_RECURRENCE_PRIO();
    afc_body<PLLDEC>(T, B, G, C, rc0, chunk_len, (int)blockIdx.x, (int)blockIdx.y);
}

// ---- B3.  The pilot PLL: the longest dependent chain of the path (phase -> LUT index -> sine -> phase).  One wave
// issues a dependent VALU operation every ~8.5 cycles and an LDS read returns after ~52, so the kernel is written for
// a SHORT CHAIN rather than for few instructions:
//   * the NCO sine (float)sin(2*pi*idx/192000) is rebuilt from two f64 factor tables held in LDS (idx = 256 a + b);
//     the host proved the expression rounds to the reference's f32 table entry for every idx (fmx_api.hip), else
//     T.trig2 is null and the global table is used.  The A table carries two wrap-around entries, so idx needs no
//     reduction modulo N (phase < fl32(2 pi) gives idx <= N);
//   * the phase never goes negative (it is PI_Constrain'ed, the correction 5*demod*gain is < 0.01 < omega), so the
//     odd-symmetry branch of SinCos::getSin and the negative branch of PI_Constrain drop out;
//   * the wrap itself, (float)fmod((double)val, 2 pi) for val in [2 pi, 2 pi + 0.7), is two exact-by-construction f32
//     operations (val - P32 is exact by Sterbenz, P32 - 2 pi is added as a constant) when the host verified that
//     identity for every float in the interval (T.wrap32_ok), else the f64 form.
template <bool T2, bool W32>
__device__ __forceinline__ void pll_body(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int64_t rc0, int chunk_len, const int bid_x, const int bid_y, const bool load_tables = true) {
    constexpr int PDP = 4;
    const int CP = G.pitch;
    double2 *sA = reinterpret_cast<double2 *>(g_rec_lds), *sB = sA + (TRIG2_A + TRIG2_APAD);
    if (T2 && load_tables) {                          // (the persistent kernel loads them once per call)
        for (int i = threadIdx.x; i < TRIG2_A + TRIG2_APAD; i += 64) sA[i] = T.trig2[i];
        for (int i = threadIdx.x; i < TRIG2_B; i += 64) sB[i] = T.trig2[TRIG2_A + TRIG2_APAD + i];
    }
    __syncthreads();
    const int ch = bid_x * 64 + threadIdx.x;
    if (ch >= C) return;
    ChanState *st = B.state + ch;
    const float gain = T.pil_gain, omega = T.pil_omega;
    const double SC = T.sincos_C, SC256 = T.sincos_C * (1.0 / 256.0);
    const float P32 = 6.2831855f, C32 = T.wrap32_c;              // fl32 just above 2 pi; fl32(P32 - 2 pi)
    float phase = st->pil_phase;
    if (!(phase >= 0.f)) phase = pi_constrain(phase);             // cannot happen (see above); keeps the invariant anyway
    const size_t ro = widx(rc0, ch, CP);
    const float *wd = B.w_dem + ro; float *wc = B.w_cur + ro; float *wo = B.w_osc + ro;
    auto step = [&](float demod, float &o_cur, float &o_osc) {
        // SinCos::getSin sincos.cpp:81-85 with phase >= 0
        const double pd = (double)phase;
        int idx = (int)(pd * SC);
        float osc;
        if (T2) {
            // idx >> 8 straight from the product: p * (SC/256) == (p * SC) / 256 exactly (power-of-two scaling), and
            // trunc(x / 256) == trunc(x) >> 8 for x >= 0 -- one operation less on the dependent chain than shift + shift
            const int ia = (int)(pd * SC256);
            const double2 ea = sA[ia], eb = sB[idx & 255];
            osc = (float)(ea.y * eb.x + ea.x * eb.y);
        } else {
            idx = (idx >= SINCOS_N) ? idx - SINCOS_N : idx;
            osc = T.sincos[idx].y;
        }
        const float perr = (5 * demod) * osc;                    // pilot-recover.cpp:56-58
        const float t = phase + perr * gain;
        o_cur = t;                                               // PI_Constrain of it is applied in pss_mix_kernel
        const float val = t + omega;                             // in (0.6, 2 pi + 0.64)
        const float wrapped = W32 ? (val - P32) + C32 : (float)((double)val - FMX_2PI);
        phase = (val < P32) ? val : wrapped;                     // PI_Constrain fm-constants.h:148-158
        o_osc = osc;
    };
    const int nfull = chunk_len / SEQ_UB;
    const int TS = SEQ_UB * CP;
    float nx[2][PDP][SEQ_UB];
    tile_pipeline<PDP>(nfull,
        [&](int s, int u, int tl) __attribute__((always_inline)) { wld(nx[s][u], wd + tl * TS); },
        [&](int s, int u, int tb) __attribute__((always_inline)) {
            float x[SEQ_UB], oc[SEQ_UB], oo[SEQ_UB];
#pragma unroll
            for (int k = 0; k < SEQ_UB; k++) x[k] = nx[s][u][k];
#pragma unroll
            for (int k = 0; k < SEQ_UB; k++) step(x[k], oc[k], oo[k]);
            wst(wc + tb * TS, oc); wst(wo + tb * TS, oo);
        });
    for (int k = 0; k < chunk_len - nfull * SEQ_UB; k++) {
        float oc, oo;
        step(wd[nfull * TS + k], oc, oo);
        wc[nfull * TS + k] = oc; wo[nfull * TS + k] = oo;
    }
    st->pil_phase = phase;
}
template <bool T2, bool W32>
__global__ __launch_bounds__(64) void pll_kernel(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int64_t rc0, int chunk_len) {
    FMX_RECURRENCE_PRIO();
    pll_body<T2, W32>(T, B, G, C, rc0, chunk_len, (int)blockIdx.x, (int)blockIdx.y);
}

// ---- B4
__device__ __forceinline__ void lock_body(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int64_t rc0, int chunk_len, const int bid_x, const int bid_y) {
    constexpr int PDL = 3;
    const int CP = G.pitch;
    const int ch = bid_x * 64 + threadIdx.x;
    if (ch >= C) return;
    ChanState *st = B.state + ch;
    const ChanParams &P = B.params[ch];
    const bool stereo_possible = P.fm_mode != 2, auto_mono = P.auto_mono != 0, pss_active = P.pss_active != 0;
    const float lockA = 1.0f / 3000.0f;
    const double keep = 1.0 - (double)lockA;
    const float omega = T.pil_omega, romega = T.pil_omega_rcp;
    float lock = st->pil_lock, old = st->pil_old;
    int stable = st->pil_stable, locked = st->pil_locked;
    int tagn = (rc0 == 0) ? 0 : st->pss_call_total;
    const size_t ro = widx(rc0, ch, CP);
    const float *wd = B.w_dem + ro, *wo = B.w_osc + ro;
    int *wt = B.w_tag + ro;                        // packed: ((tag + 2) << 1) | locked
    auto step = [&](float demod, float osc, int &o_lk, int &o_tag) {
        const float pilot = 5 * demod;
        const float quadRef = fdiv_const(osc - old, omega, romega);
        old = osc;
        lock = (float)((double)(lockA * (-quadRef * pilot)) + (double)lock * keep);
        const bool tmp = lock > 0.07f;
        // if (tmp) { if (locked || ++stable > N/2) locked = 1; } else { locked = 0; stable = 0; }
        const int stable_inc = stable + ((tmp && !locked) ? 1 : 0);
        locked = tmp ? ((locked || stable_inc > (SINCOS_N >> 1)) ? 1 : 0) : 0;
        stable = tmp ? stable_inc : 0;
        o_lk = locked;
        const bool branch = stereo_possible && (locked || !auto_mono);
        o_tag = branch ? (pss_active ? tagn : -1) : -2;
        tagn += (branch && pss_active) ? 1 : 0;
    };
    // The lock VALUE is a linear recurrence of the inputs; only the flags and counters derived from it feed the
    // outputs.  Fast path per block of UB samples: run the value chain alone, note whether `lock > 0.07` held for
    // all / none of the block's samples; if so (and no counter threshold can be crossed inside the block) the state
    // machine has a closed form.  Anything else (a wave-uniform decision) replays the block sample by sample.
    constexpr int UB = SEQ_UB;
    const int nfull = chunk_len / UB;
    const int TS = UB * CP;
    float nd[2][PDL][UB], no[2][PDL][UB];
    float (*sLd)[64] = reinterpret_cast<float (*)[64]>(g_rec_lds), (*sLo)[64] = sLd + UB;
    tile_pipeline<PDL>(nfull,
        [&](int s, int u, int tl) __attribute__((always_inline)) { wld(nd[s][u], wd + tl * TS); wld(no[s][u], wo + tl * TS); },
        [&](int s, int u, int tb) __attribute__((always_inline)) {
            float d[UB], o[UB]; int pk[UB];
#pragma unroll
            for (int k = 0; k < UB; k++) { d[k] = nd[s][u][k]; o[k] = no[s][u][k]; }
            const float lock0 = lock, old0 = old;
            bool all_hi = true, all_lo = true;
#pragma unroll
            for (int k = 0; k < UB; k++) {
                const float quadRef = fdiv_const(o[k] - old, omega, romega);
                old = o[k];
                lock = (float)((double)(lockA * (-quadRef * (5 * d[k]))) + (double)lock * keep);
                const bool tmp = lock > 0.07f;
                all_hi = all_hi && tmp; all_lo = all_lo && !tmp;
            }
            // closed forms:  all_hi & locked -> unchanged;  all_hi & !locked & stable + UB <= N/2 -> stable += UB;
            //                all_lo -> locked = 0, stable = 0
            const bool easy = all_lo || (all_hi && (locked != 0 || stable + UB <= (SINCOS_N >> 1)));
            const bool fast = __all(easy) != 0;
            if (B.dbg && threadIdx.x == 0) B.dbg[(size_t)ch * DBG_SLOTS + (fast ? 11 : 12)] += 1;
            if (fast) {
                if (all_lo) { locked = 0; stable = 0; }
                else if (!locked) stable += UB;
                const bool branch = stereo_possible && (locked || !auto_mono);
                const bool counts = branch && pss_active;
                const int base = branch ? (pss_active ? tagn : -1) : -2;
                const int inc = counts ? 1 : 0;
#pragma unroll
                for (int k = 0; k < UB; k++) pk[k] = ((base + inc * k + 2) << 1) | locked;
                tagn += inc * UB;
                wst(wt + tb * TS, pk);
            } else {
                // rare (lock acquisition / loss): replay the block sample by sample in a rolled loop, operands through LDS
                // (no vector-memory operations on this path, see pss_acc_kernel)
                lock = lock0; old = old0;
#pragma unroll
                for (int k = 0; k < UB; k++) { sLd[k][threadIdx.x] = d[k]; sLo[k][threadIdx.x] = o[k]; }
#pragma unroll 1
                for (int k = 0; k < UB; k++) {
                    int lk, tg;
                    step(sLd[k][threadIdx.x], sLo[k][threadIdx.x], lk, tg);
                    sLo[k][threadIdx.x] = __int_as_float(((tg + 2) << 1) | lk);
                }
#pragma unroll
                for (int k = 0; k < UB; k++) pk[k] = __float_as_int(sLo[k][threadIdx.x]);
                wst(wt + tb * TS, pk);
            }
        });
    for (int k = 0; k < chunk_len - nfull * UB; k++) {
        int lk, tg;
        step(wd[nfull * TS + k], wo[nfull * TS + k], lk, tg);
        wt[nfull * TS + k] = ((tg + 2) << 1) | lk;
    }
    st->pil_lock = lock; st->pil_old = old; st->pil_stable = stable; st->pil_locked = locked;
    st->pss_call_total = tagn;
}
__global__ __launch_bounds__(64) void lock_kernel(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int64_t rc0, int chunk_len) {
    FMX_RECURRENCE_PRIO();
    lock_body(T, B, G, C, rc0, chunk_len, (int)blockIdx.x, (int)blockIdx.y);
}

// =================================================================================================
// B5  PSS low-pass + error for every sample of the chunk   (time-parallel)
//     stereo-separation.cpp:60-83 : err = Re(y)*Im(y), y = sum_k h[k] s[i - 1753 - k], i = PSS call index
// =================================================================================================
constexpr int PSS_TILE = 512;
constexpr int PSS_FPT = 8;                                  // adjacent samples per thread
constexpr int PSS_TG = (PSS_TAPS + 11) / 12 * 3;            // tap groups of four, a multiple of three (75 groups = 300 taps, zero padded)
constexpr int PSS_WU = (PSS_TILE + 4 * PSS_TG + 8) / 8 + 2; // window units (a unit = two s samples = one float4) per plane
// One wave per (512-sample tile, channel); a thread computes EIGHT adjacent samples.  When their PSS call indices are
// consecutive (the steady case: every sample calls process_sample) the s window slides through a twelve-entry register
// ring -- per four taps two ds_read_b128 and 32 packed FMAs (the PSS input is complex, the taps real), the taps come in
// as scalars (wave-uniform loads).  The window sits in LDS in four planes, unit u (samples 2u, 2u+1) in plane u % 4 at
// offset u / 4: a lane's samples start 4 units after its neighbour's, so every ds_read_b128 of the wave reads consecutive
// 16-byte slots of one plane.  Otherwise (pilot lock coming or going inside the tile) each output reads its own window.
__device__ __forceinline__ void pss_fir_body(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int64_t rc0, int chunk_len, const int bid_x, const int bid_y) {
    const int64_t CP = G.pitch;   // padded row pitch of the sample-major work arrays
    __shared__ __attribute__((aligned(16))) float4 sW[4][PSS_WU];
    __shared__ __attribute__((aligned(16))) float sH[4 * PSS_TG];     // sH[w] = h[PSS_TAPS - 1 - w] (0 beyond)
    const int ch = bid_y;
    const int lane = threadIdx.x;
    const int q0 = bid_x * PSS_TILE;
    const ChanParams &P = B.params[ch];
    if (P.fm_mode == 2 || !P.pss_active) return;
    const int64_t ic = B.state[ch].pss_count;            // PSS call index at the start of this CALL
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    int tg[PSS_FPT];
    const int q = q0 + PSS_FPT * lane;                   // eight adjacent rows: half a work-array tile, two dwordx4
    {
        i32x4_t pa = {0, 0, 0, 0}, pb = {0, 0, 0, 0};
        if (q < chunk_len) {
            const i32x4_t *src = reinterpret_cast<const i32x4_t *>(&B.w_tag[widx(rc0 + q, ch, (int)CP)]);
            pa = src[0]; pb = src[1];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { tg[j] = (q + j < chunk_len) ? (pa[j] >> 1) - 2 : -2; tg[4 + j] = (q + 4 + j < chunk_len) ? (pb[j] >> 1) - 2 : -2; }
    }
    int tmin = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < PSS_FPT; j++) if (tg[j] >= 0) tmin = tg[j] < tmin ? tg[j] : tmin;
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(tmin, d, 64); tmin = o < tmin ? o : tmin; }
    if (tmin == 0x7fffffff) return;                      // no PSS call in this tile
    const float2 *sring = B.sring + (size_t)ch * (G.sring_mask + 1);
    float2 *sW2 = reinterpret_cast<float2 *>(&sW[0][0]);
    auto wslot = [](int w) { const int u = w >> 1; return (((u & 3) * PSS_WU + (u >> 2)) << 1) | (w & 1); };   // float2 index of window entry w
    // window entry w <-> s index (ic + tmin) - (1753 + 294) + w ; tags in a tile span < PSS_TILE
    // (fills in batches: the loads of a batch are in flight together -- the fill is pure memory latency for the wave)
    {
        constexpr int NW = 8 * (PSS_WU - 1), FB = 7;
#pragma unroll
        for (int w0 = 0; w0 < NW; w0 += 64 * FB) {
            float2 v[FB];
#pragma unroll
            for (int k = 0; k < FB; k++) {
                const int w = w0 + 64 * k + lane;
                const int64_t idx = ic + tmin - (PSS_DELAY + PSS_TAPS - 1) + w;
                v[k] = (w < NW && idx >= 0 && w < PSS_TILE + PSS_TAPS - 1) ? sring[idx & G.sring_mask] : make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < FB; k++) {
                const int w = w0 + 64 * k + lane;
                if (w < NW) sW2[wslot(w)] = v[k];
            }
        }
        constexpr int HB = (4 * PSS_TG + 63) / 64;
        float hv[HB];
#pragma unroll
        for (int k = 0; k < HB; k++) { const int w = 64 * k + lane; hv[k] = (w < PSS_TAPS) ? T.pss_taps[PSS_TAPS - 1 - w] : 0.f; }
#pragma unroll
        for (int k = 0; k < HB; k++) { const int w = 64 * k + lane; if (w < 4 * PSS_TG) sH[w] = hv[k]; }
    }
    __syncthreads();
    // y(sample) = sum_w sH[w] * s[w + off], off = tag - tmin
    typedef float v2f_t __attribute__((ext_vector_type(2)));
    float err8[PSS_FPT];
    const bool beyond = q >= chunk_len;                  // lanes past the end of the chunk compute nothing that is stored
    bool consecutive = (tg[0] >= 0) && (((tg[0] - tmin) & 7) == 0);
#pragma unroll
    for (int j = 1; j < PSS_FPT; j++) consecutive = consecutive && (tg[j] == tg[0] + j);
    consecutive = consecutive || beyond;
    if (__all(consecutive)) {
        const int m = beyond ? 0 : (tg[0] - tmin) >> 3;  // this thread's first window unit is 4 m
        const float4 *hr = reinterpret_cast<const float4 *>(sH);
        v2f_t acc[PSS_FPT], c[12];
#pragma unroll
        for (int j = 0; j < PSS_FPT; j++) acc[j] = (v2f_t){0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 6; k++) {                    // ring = window entries 8 m .. 8 m + 11 (units 4 m .. 4 m + 5)
            const float4 v = sW[k & 3][m + (k >> 2)];
            c[2 * k] = (v2f_t){v.x, v.y}; c[2 * k + 1] = (v2f_t){v.z, v.w};
        }
        for (int g3 = 0; g3 < PSS_TG; g3 += 3) {
#pragma unroll
            for (int gg = 0; gg < 3; gg++) {             // group g: taps 4 g .. 4 g + 3 on ring entries (4 gg + q + j) % 12
                const int g = g3 + gg;
                const float4 h4 = hr[g];
                const float hq[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
                for (int qq = 0; qq < 4; qq++) {
                    const v2f_t w = (v2f_t){hq[qq], hq[qq]};
#pragma unroll
                    for (int j = 0; j < PSS_FPT; j++) acc[j] = __builtin_elementwise_fma(w, c[(4 * gg + qq + j) % 12], acc[j]);
                }
                // entries 4 g + 12 .. 4 g + 15 replace the four oldest: units 4 m + 2 g + 6, + 7
                const int u0 = 2 * g + 6;
                const float4 va = sW[u0 & 3][m + (u0 >> 2)], vb = sW[(u0 + 1) & 3][m + ((u0 + 1) >> 2)];
                c[(4 * gg) % 12] = (v2f_t){va.x, va.y}; c[(4 * gg + 1) % 12] = (v2f_t){va.z, va.w};
                c[(4 * gg + 2) % 12] = (v2f_t){vb.x, vb.y}; c[(4 * gg + 3) % 12] = (v2f_t){vb.z, vb.w};
            }
        }
#pragma unroll
        for (int j = 0; j < PSS_FPT; j++) err8[j] = acc[j].x * acc[j].y;
    } else {
#pragma unroll 1
        for (int j = 0; j < PSS_FPT; j++) {
            float ar = 0.f, ai = 0.f;
            if (tg[j] >= 0) {
                const int off = tg[j] - tmin;
                for (int w = 0; w < PSS_TAPS; w++) { const float2 v = sW2[wslot(w + off)]; ar = fmaf(sH[w], v.x, ar); ai = fmaf(sH[w], v.y, ai); }
            }
            err8[j] = ar * ai;
        }
    }
    {
        float *dst = &B.w_err[widx(q, ch, (int)CP)];
        if (q + PSS_FPT - 1 < chunk_len) {
            reinterpret_cast<f32x4_t *>(dst)[0] = (f32x4_t){err8[0], err8[1], err8[2], err8[3]};
            reinterpret_cast<f32x4_t *>(dst)[1] = (f32x4_t){err8[4], err8[5], err8[6], err8[7]};
        } else for (int j = 0; j < PSS_FPT; j++) if (q + j < chunk_len) dst[j] = err8[j];
    }
}
__global__ __launch_bounds__(64) void pss_fir_kernel(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int64_t rc0, int chunk_len, TSync Y) {
    if (!tsync_enter(Y, (int)blockIdx.y >> 6) || Y.skip) return;
    pss_fir_body(T, B, G, C, rc0, chunk_len, (int)blockIdx.x, (int)blockIdx.y);
}

// =================================================================================================
// B6  PSS integrator + state machines   [lane per channel]
//     fm-processor.cpp:699-718, stereo-separation.cpp:84-109
// =================================================================================================
constexpr int ACC_UB = SEQ_UB;
struct AccState { float acc, mean, pdp; int lock_cnt, unlock_cnt; bool minimized; };
__device__ __forceinline__ float pss_acc_step(AccState &s, float alpha, float la, float keep, bool locked, int tag, float err) {
    // branch-free: every lane (channel) may be in a different state
    const bool rst = !locked;                      // unlocked: pilotDelayPSS = 0; pPSS.reset() (fm-processor.cpp:699-702)
    s.pdp = rst ? 0.f : s.pdp; s.acc = rst ? 0.f : s.acc; s.mean = rst ? 0.f : s.mean;
    s.minimized = s.minimized && !rst; s.lock_cnt = rst ? 0 : s.lock_cnt; s.unlock_cnt = rst ? 0 : s.unlock_cnt;
    const float used = s.pdp;                      // the value phaseforLRDiff is built from (:707-709)
    const bool call = tag >= 0;                    // PerfectStereoSeparation::process_sample :60-109
    const float error = s.minimized ? err : err * 10.0f;
    // clamp to +-M_PI_4: `acc < -M_PI_4` (f64 compare) <=> acc <= -fl32(pi/4), and the assigned value is fl32(pi/4)
    const float c4 = 0.785398185253143310546875f;
    const float nacc = fminf(fmaxf(s.acc + alpha * error, -c4), c4);
    const float nmean = la * error + s.mean * keep;
    const bool small = fabsf(nmean) < 0.001f;
    // small: if (minimized || ++lock_cnt > 3N) minimized = 1; unlock_cnt = 0;
    // else : if (!minimized || ++unlock_cnt > 3N) minimized = 0; lock_cnt = 0;
    const int lc1 = s.lock_cnt + ((small && !s.minimized) ? 1 : 0);
    const int uc1 = s.unlock_cnt + ((!small && s.minimized) ? 1 : 0);
    const bool nmin = small ? (s.minimized || lc1 > 3 * SINCOS_N) : (s.minimized && !(uc1 > 3 * SINCOS_N));
    s.acc = call ? nacc : s.acc; s.mean = call ? nmean : s.mean; s.minimized = call ? nmin : s.minimized;
    s.lock_cnt = call ? (small ? lc1 : 0) : s.lock_cnt; s.unlock_cnt = call ? (small ? 0 : uc1) : s.unlock_cnt;
    s.pdp = call ? nacc : ((tag == -1) ? 0.f : s.pdp);
    return used;
}
__device__ __forceinline__ void pss_acc_body(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int64_t rc0, int chunk_len, const int bid_x, const int bid_y) {
    constexpr int PDC = 3;
    const int CP = G.pitch;       // padded row pitch of the sample-major work arrays
    const int ch = bid_x * 64 + threadIdx.x;
    if (ch >= C) return;
    ChanState *st = B.state + ch;
    const ChanParams &P = B.params[ch];
    AccState s;
    s.acc = st->pss_acc; s.mean = st->pss_mean; s.pdp = st->pilot_delay_pss;
    s.lock_cnt = st->pss_lock_cnt; s.unlock_cnt = st->pss_unlock_cnt; s.minimized = st->pss_minimized != 0;
    if ((P.actions & (ACT_TRIGGER_FREQ | ACT_RESTART_PSS)) && rc0 == 0) {
        // triggerFrequencyChange / restartPssAnalyzer fm-processor.cpp:849-860
        s.pdp = 0.f; s.acc = 0.f; s.minimized = false; s.mean = 0.f; s.lock_cnt = 0; s.unlock_cnt = 0;
        if (P.actions & ACT_TRIGGER_FREQ) st->fade_start_frame = G.M0;
    }
    const int *tg = B.w_tag + widx(rc0, ch, CP);              // packed ((tag + 2) << 1) | locked
    const float *err = B.w_err + widx(0, ch, CP);
    float *pdpw = B.w_pdp + widx(rc0, ch, CP);
    const float alpha = T.pss_alpha, la = T.pss_lock_alpha, keep = 1.0f - la;
    const bool pss_on = (P.fm_mode != 2) && (P.pss_active != 0);
    // Fast paths per block of ACC_UB samples (wave-uniform decisions, identical arithmetic):
    //  * steady: every channel of the wave locked and calling process_sample for the whole block, and no lock /
    //    unlock counter within a block of its 3 s threshold, so `minimized` cannot change inside the block: the float
    //    recurrences run alone (`small` is collected as a bit per sample) and the counters follow in closed form;
    //  * idle: nobody locked, nobody calling (mono or no pilot with autoMono): the state is all zeros.
    const int nfull = chunk_len / ACC_UB;
    const int TS = ACC_UB * CP;
    int nt[2][PDC][ACC_UB]; float ne[2][PDC][ACC_UB];
    int (*sTg)[64] = reinterpret_cast<int (*)[64]>(g_rec_lds);
    float (*sEr)[64] = reinterpret_cast<float (*)[64]>(g_rec_lds) + ACC_UB;
    const float c4 = 0.785398185253143310546875f;
    tile_pipeline<PDC>(nfull,
        [&](int g, int u, int tl) __attribute__((always_inline)) { wld(nt[g][u], tg + tl * TS); wld(ne[g][u], err + tl * TS); },
        [&](int g, int u, int tb) __attribute__((always_inline)) {
            float e[ACC_UB]; float o[ACC_UB];
            unsigned andv = ~0u, orv = 0u; int minv = 0x7fffffff;
#pragma unroll
            for (int k = 0; k < ACC_UB; k++) {
                e[k] = pss_on ? ne[g][u][k] : 0.f;
                andv &= (unsigned)nt[g][u][k]; orv |= (unsigned)nt[g][u][k]; minv = nt[g][u][k] < minv ? nt[g][u][k] : minv;
            }
            // (only the counter that can flip `minimized` matters: lock_cnt while it is 0, unlock_cnt while it is 1)
            const bool steady = ((andv & 1u) != 0) && (minv >= 4) &&                 // locked, tag >= 0 throughout
                                ((s.minimized ? s.unlock_cnt : s.lock_cnt) + ACC_UB <= 3 * SINCOS_N);
            const bool idle = (orv & ~2u) == 0;                                      // unlocked, tag < 0 throughout
            const bool f_steady = __all(steady) != 0, f_idle = __all(idle) != 0;
            if (B.dbg && threadIdx.x == 0) B.dbg[(size_t)ch * DBG_SLOTS + (f_steady ? 8 : (f_idle ? 9 : 10))] += 1;
            if (f_steady) {
                const float scale = s.minimized ? 1.0f : 10.0f;                      // error = minimized ? err : err * 10
                constexpr unsigned ALL = (ACC_UB == 32) ? ~0u : ((1u << (ACC_UB & 31)) - 1u);
                unsigned bits = 0;
                float prev = s.pdp;
#pragma unroll
                for (int k = 0; k < ACC_UB; k++) {
                    o[k] = prev;
                    const float error = e[k] * scale;
                    s.acc = __builtin_amdgcn_fmed3f(s.acc + alpha * error, -c4, c4);
                    s.mean = la * error + s.mean * keep;
                    bits = (bits << 1) | (fabsf(s.mean) < 0.001f ? 1u : 0u);
                    prev = s.acc;
                }
                s.pdp = s.acc;
                if (s.minimized) {
                    s.lock_cnt = (bits == ALL) ? s.lock_cnt : 0;
                    s.unlock_cnt = (bits == 0u) ? s.unlock_cnt + ACC_UB : __builtin_ctz(bits);
                } else {
                    s.lock_cnt = (bits == ALL) ? s.lock_cnt + ACC_UB : __builtin_ctz(~bits);
                    s.unlock_cnt = (bits != 0u) ? 0 : s.unlock_cnt;
                }
                wst(pdpw + tb * TS, o);
            } else if (f_idle) {
                s.pdp = 0.f; s.acc = 0.f; s.mean = 0.f; s.minimized = false; s.lock_cnt = 0; s.unlock_cnt = 0;
#pragma unroll
                for (int k = 0; k < ACC_UB; k++) o[k] = 0.f;
                wst(pdpw + tb * TS, o);
            } else {
                // rare (lock transitions, a counter near its threshold, mixed modes in one wave): sample by sample in a
                // rolled loop.  Its operands go through LDS, NOT through memory: vector-memory operations on this path
                // would make the compiler drain every prefetched load (s_waitcnt vmcnt(0)) where the paths join, on
                // every tile, fast path included (measured: 124 us per chunk instead of 56).
#pragma unroll
                for (int k = 0; k < ACC_UB; k++) { sTg[k][threadIdx.x] = nt[g][u][k]; sEr[k][threadIdx.x] = e[k]; }
#pragma unroll 1
                for (int k = 0; k < ACC_UB; k++) {
                    const int p = sTg[k][threadIdx.x];
                    sEr[k][threadIdx.x] = pss_acc_step(s, alpha, la, keep, (p & 1) != 0, (p >> 1) - 2, sEr[k][threadIdx.x]);
                }
#pragma unroll
                for (int k = 0; k < ACC_UB; k++) o[k] = sEr[k][threadIdx.x];
                wst(pdpw + tb * TS, o);
            }
        });
    for (int k = 0; k < chunk_len - nfull * ACC_UB; k++) {
        const int p = tg[nfull * TS + k];
        pdpw[nfull * TS + k] = pss_acc_step(s, alpha, la, keep, (p & 1) != 0, (p >> 1) - 2, pss_on ? err[nfull * TS + k] : 0.f);
    }
    st->pss_acc = s.acc; st->pss_mean = s.mean; st->pilot_delay_pss = s.pdp;
    st->pss_lock_cnt = s.lock_cnt; st->pss_unlock_cnt = s.unlock_cnt; st->pss_minimized = s.minimized ? 1 : 0;
}
__global__ __launch_bounds__(64) void pss_acc_kernel(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int64_t rc0, int chunk_len) {
    FMX_RECURRENCE_PRIO();
    pss_acc_body(T, B, G, C, rc0, chunk_len, (int)blockIdx.x, (int)blockIdx.y);
}

// =================================================================================================
// B7  38 kHz mix, PSS input, stereo matrix   (time-parallel; transposing like B1)
//     fm-processor.cpp:707-730, 517-549
// =================================================================================================
constexpr int MIX_ROWS = 64, MIX_CH = 16;               // one block: four work-array tile rows of sixteen channels
__device__ __forceinline__ void pss_mix_body(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int64_t rc0, int chunk_len, const int bid_x, const int bid_y) {
    // One block = 64 samples x 16 channels, threads as (sample in tile row, channel); a thread's four elements are four
    // tile rows of ONE channel, so the channel's settings are read once, the tiled work arrays are accessed in 1 KB runs
    // and the channel-major s ring in 128-byte runs that a block extends to 512 bytes.  The loads of all four elements are
    // issued before anything is computed, and the SinCos gathers of all four before they are used.
    const int CP = G.pitch;
    const int tid = threadIdx.x;
    const int q0 = bid_x * MIX_ROWS;
    const float2 *__restrict__ sct = T.sincos;
    const double SC = T.sincos_C;
    constexpr int EPT = MIX_ROWS / WT;
    const int chx = bid_y * MIX_CH + (tid >> 4);
    const int chc = chx < C ? chx : C - 1;
    const ChanParams &P = B.params[chc];
    const int ssel1 = P.sound_sel, fmode1 = P.fm_mode; const float pano1 = P.panorama;
    const int64_t ic1 = B.state[chc].pss_count;
    int ch[EPT]; bool ok[EPT]; size_t wi[EPT]; int qq[EPT];
    float demod[EPT], cur[EPT], pdp[EPT]; int tag[EPT];
    int ssel[EPT], fmode[EPT]; float pano[EPT]; int64_t ic[EPT];
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        const int q = q0 + WT * i + (tid & 15);
        qq[i] = q;
        const bool qok = q < chunk_len;
        const int64_t r = rc0 + (qok ? q : chunk_len - 1);            // clamped: loads stay unconditional
        ch[i] = chx; ok[i] = qok && chx < C;
        wi[i] = widx(r, chc, CP);
        demod[i] = B.w_dem[wi[i]]; tag[i] = (B.w_tag[wi[i]] >> 1) - 2; cur[i] = B.w_cur[wi[i]]; pdp[i] = B.w_pdp[wi[i]];
        ssel[i] = ssel1; fmode[i] = fmode1; pano[i] = pano1; ic[i] = ic1;
    }
    float ph[EPT]; float2 e[EPT]; float sn[EPT];
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        // phaseforLRDiff fm-processor.cpp:707-714
        float p = (float)(2 * ((double)pi_constrain(cur[i]) + FMX_PI_4 + 0) - (double)pdp[i]);
        if ((double)p < -FMX_2PI) p = (float)((double)p + 2 * FMX_2PI);
        ph[i] = (tag[i] != -2) ? (float)fmod_2pi((double)p) : 0.f;     // (mono: the LUT entry is not used)
    }
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        e[i] = sct[sc_index(sc_wrap(ph[i]), SC)];
        sn[i] = (ssel[i] == 6) ? sc_sin(sct, SC, ph[i]) : 0.f;    // S_LEFTminusRIGHT_Test only
    }
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        if (!ok[i]) continue;
        float2 audio = make_float2(demod[i], 0.f);
        if (tag[i] != -2) {
            if (tag[i] >= 0)
                B.sring[(size_t)ch[i] * (G.sring_mask + 1) + ((ic[i] + tag[i]) & G.sring_mask)] = make_float2(e[i].x * demod[i], e[i].y * demod[i]);
            const float lut = (ssel[i] == 6) ? sn[i] : e[i].x;
            audio.y = (float)(2.0 * (double)lut * (double)demod[i]);
        }
        const float sumLR = audio.x, diffLR = audio.y;
        const float dw = diffLR * (fmode[i] == 1 ? pano[i] : 1.0f);
        const float left = sumLR + dw, right = sumLR - dw;
        float2 o;
        switch (ssel[i]) {
        default:
        case 0: o = make_float2(left, right); break;
        case 1: o = make_float2(right, left); break;
        case 2: o = make_float2(left, left); break;
        case 3: o = make_float2(right, right); break;
        case 4: o = make_float2(sumLR, sumLR); break;
        case 5: case 6: o = make_float2(dw, dw); break;
        }
        B.w_x[wi[i]] = o;
        B.w_diff[wi[i]] = audio.y;                  // (sum, diff) scope tap = (w_dem, w_diff): read back by fmx_get_tap
    }
}
__global__ __launch_bounds__(256) void pss_mix_kernel(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int64_t rc0, int chunk_len, TSync Y) {
    if (!tsync_enter(Y, (int)blockIdx.y >> 2) || Y.skip) return;
    pss_mix_body(T, B, G, C, rc0, chunk_len, (int)blockIdx.x, (int)blockIdx.y);
}
// =================================================================================================
// B8  de-emphasis   [lane per channel]   fm-processor.cpp:594-595 (the gain of :303-306 is applied by the audio kernel)
//     plus the 0.5 s meta snapshot (:662-684)
// =================================================================================================
__device__ __forceinline__ void deemph_body(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int64_t rc0, int chunk_len, int last_chunk, const int bid_x, const int bid_y) {
    constexpr int PDD = 3;
    const int CP = G.pitch;       // padded row pitch of the sample-major work arrays
    const int ch = bid_x * 64 + threadIdx.x;
    if (ch >= C) return;
    const int64_t nj = chunk_len;
    ChanState *st = B.state + ch;
    const ChanParams &P = B.params[ch];
    const float a = P.deemph_alpha;
    float yl = st->de_l, yr = st->de_r;
    const float2 *x = B.w_x + widx(rc0, ch, CP);
    const int64_t dmask = G.dring_mask, dcap = G.dring_mask + 1;
    float2 *dr = B.dring + (size_t)ch * dcap;
    constexpr int UB = SEQ_UB;
    const int nfull = (int)(nj / UB);
    const int TS = UB * CP;
    float2 nx[2][PDD][UB];
    tile_pipeline<PDD>(nfull,
        [&](int s, int u, int tl) __attribute__((always_inline)) { wld2(nx[s][u], x + tl * TS); },
        [&](int s, int u, int tb) __attribute__((always_inline)) {
            float2 v[UB];
#pragma unroll
            for (int k = 0; k < UB; k++) v[k] = nx[s][u][k];
#pragma unroll
            for (int k = 0; k < UB; k++) {
                yl = (v[k].x - yl) * a + yl;
                yr = (v[k].y - yr) * a + yr;
                v[k] = make_float2(yl, yr);
            }
            // straight into the channel-major d ring the audio kernel reads: 16 rows = 128 contiguous bytes per lane
            const int64_t p0 = (G.J0 + rc0 + (int64_t)tb * UB) & dmask;
            if (p0 + UB <= dcap && (p0 & 1) == 0) wst2(dr + p0, v);
            else for (int k = 0; k < UB; k++) dr[(p0 + k) & dmask] = v[k];
        });
    for (int k = 0; k < (int)(nj - (int64_t)nfull * UB); k++) {
        const float2 v = x[nfull * TS + k];
        yl = (v.x - yl) * a + yl;
        yr = (v.y - yr) * a + yr;
        dr[(G.J0 + rc0 + (int64_t)nfull * UB + k) & dmask] = make_float2(yl, yr);
    }
    st->de_l = yl; st->de_r = yr;
    if (!last_chunk) return;
    // meta snapshot: emitted by the reference every fmRate/2 samples; taken at the end of the call
    // in which that count is crossed (values of the call end)
    int cnt = st->my_count + (int)(G.J1 - G.J0);
    if (cnt > (SINCOS_N >> 1)) {
        const bool stereo_possible = P.fm_mode != 2;
        const bool lk = stereo_possible && st->pil_locked;
        st->meta_locked = lk ? 1 : 0;
        st->meta_lock_strength = stereo_possible ? st->pil_lock : 0.f;
        const float dcabs = (float)sqrt((double)st->dc_re * (double)st->dc_re + (double)st->dc_im * (double)st->dc_im);
        st->meta_dc_rf = P.dc_remove ? 20 * log10f(dcabs + 1.0f / 32768) : (float)-99.99;
        st->meta_dc_if = st->fm_afc;
        st->meta_pss_deg = (float)((double)st->pilot_delay_pss / 3.14159265358979323846 * 180.0f);
        st->meta_pss_change = st->pss_mean * 1000;
        st->meta_pss_state = (P.pss_active && lk) ? (st->pss_minimized ? 2 : 1) : 0;
        cnt -= (SINCOS_N >> 1) + 1;
    }
    st->my_count = cnt;
    st->pss_count += st->pss_call_total;           // advance the PSS filter time base by this call's process_sample calls
    st->pss_call_total = 0;
}
__global__ __launch_bounds__(64) void deemph_kernel(DeviceTables T, DeviceBuffers B, CallGeom G, int C, int64_t rc0, int chunk_len, int last_chunk) {
    FMX_RECURRENCE_PRIO();
    deemph_body(T, B, G, C, rc0, chunk_len, last_chunk, (int)blockIdx.x, (int)blockIdx.y);
}

// =================================================================================================
// Persistent layout of stage B (see launch_demod_persistent)
// =================================================================================================
struct ChunkPlan { int n; int role_mask; int rc0[PB_MAX_CHUNKS]; int len[PB_MAX_CHUNKS]; int nb_disc[PB_MAX_CHUNKS], nb_fir[PB_MAX_CHUNKS], nb_mix[PB_MAX_CHUNKS]; };
constexpr int PB_SPIN_LIMIT = 1 << 22;            // x ~0.5 us: a wait longer than ~2 s gives up and raises DemodSync::abort

// wave-uniform wait until *p >= need; false when the pipeline was aborted.  The polls are relaxed loads (an acquire load
// per poll would invalidate the XCD's L2 every time); one acquire fence follows once the word has arrived.
__device__ __forceinline__ bool pb_wait(const int *p, int need, int *abort_flag, int who) {
    int spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > PB_SPIN_LIMIT || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            if (__hip_atomic_exchange(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                DemodSync *S = reinterpret_cast<DemodSync *>(abort_flag);       // diagnostics: who gave up first, and the state then
                if (S->host_flag) __hip_atomic_store(S->host_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                S->info[0] = who; S->info[1] = need; S->info[2] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int g = S->groups;
                for (int i = 0; i < 16; i++) {
                    S->snap[i] = __hip_atomic_load(&S->cnt_disc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    S->snap[16 + i] = __hip_atomic_load(&S->cnt_fir[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    S->snap[32 + i] = __hip_atomic_load(&S->cnt_mix[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                for (int r = 0; r < PB_ROLES; r++) {
                    S->snap[48 + r] = __hip_atomic_load(&S->prog[0][0] + r * g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    S->snap[56 + r] = __hip_atomic_load(&S->prog[0][0] + r * g + g - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            return false;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}

// One wave per (64-channel group, role); every role walks the call's chunks in order.  A role waits for the progress word
// of the role in front of it (same group), or for the completion count of the time-parallel kernel that feeds it, then
// runs the same chunk body as the event-driven layout and publishes its own progress.
template <bool PLLDEC, bool T2, bool W32>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void recurrences_kernel(DeviceTables T, DeviceBuffers B, CallGeom G, int C, ChunkPlan P,
                                                         DemodSync *S, int groups) {
    FMX_RECURRENCE_PRIO();
    const int grp = blockIdx.x;
    int *prog = &S->prog[0][0];
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&S->started, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int role = blockIdx.y;
    bool first = true;
    for (int c = 0; c < P.n; c++) {
        bool ok = true;
        switch (role) {
        case 0: ok = pb_wait(&S->cnt_disc[c], P.nb_disc[c], &S->abort, 100 * c + 0); break;
        case 1: ok = pb_wait(&prog[0 * groups + grp], c + 1, &S->abort, 100 * c + 1); break;
        case 2: ok = pb_wait(&prog[1 * groups + grp], c + 1, &S->abort, 100 * c + 2); break;
        case 3: ok = pb_wait(&S->cnt_fir[c], P.nb_fir[c], &S->abort, 100 * c + 3); break;
        default: ok = pb_wait(&S->cnt_mix[c], P.nb_mix[c], &S->abort, 100 * c + 4); break;
        }
        if (!ok) return;
        const int64_t rc0 = P.rc0[c]; const int len = P.len[c];
        if ((P.role_mask >> role) & 1)                    // (diagnostics: roles outside FMX_DEBUG_ROLE_MASK only pass the word on)
        switch (role) {
        case 0: afc_body<PLLDEC>(T, B, G, C, rc0, len, grp, 0); break;
        case 1: pll_body<T2, W32>(T, B, G, C, rc0, len, grp, 0, first); break;
        case 2: lock_body(T, B, G, C, rc0, len, grp, 0); break;
        case 3: {                                            // chunk c's errors sit in half (c & 1) of the error array
            DeviceBuffers Bc = B;
            Bc.w_err = B.w_err + (size_t)(c & 1) * (PB_CHUNK / WT) * G.pitch * WT;
            pss_acc_body(T, Bc, G, C, rc0, len, grp, 0);
            break; }
        default: deemph_body(T, B, G, C, rc0, len, (c == P.n - 1) ? 1 : 0, grp, 0); break;
        }
        first = false;
        __builtin_amdgcn_wave_barrier();
        __threadfence();                                      // this chunk's stores (work arrays, channel state) before the word
        if (threadIdx.x == 0) __hip_atomic_store(&prog[role * groups + grp], c + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Head of the time-parallel stream: returns once every workgroup of the recurrence kernel is running.  The time-parallel
// kernels' blocks wait in place for their group's progress word; where they share CUs with the recurrence kernel they
// must not be allowed to fill those CUs before the workgroups they wait for are resident.
__global__ __launch_bounds__(64) void start_gate_kernel(DemodSync *S, int need) {
    (void)pb_wait(&S->started, need, &S->abort, 9999);
}

// Completion word of a time-parallel kernel: launched right behind it on the same stream, so the kernel boundary has
// already made that kernel's stores visible device-wide (a release fence per BLOCK would write the XCD's L2 back
// thousands of times per kernel).
__global__ __launch_bounds__(64) void signal_kernel(int *p, int v) {
    if (threadIdx.x == 0) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(64) void sync_init_kernel(DemodSync *S, int groups, int *host_flag) {
    if (threadIdx.x == 0) { S->groups = groups; S->host_flag = host_flag; }
}

template <bool PLLDEC, bool T2, bool W32>
static void launch_recurrences(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, int C, const ChunkPlan &P,
                               DemodSync *S, int groups, hipStream_t s) {
    hipLaunchKernelGGL((recurrences_kernel<PLLDEC, T2, W32>), dim3((unsigned)groups, PB_ROLES), dim3(64), 0, s, T, B, G, C, P, S, groups); FMX_LAUNCHED();
}

// Occupancy of the persistent kernel (blocks per CU), for the co-residency check the host makes before choosing this layout.
int recurrences_blocks_per_cu() {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, recurrences_kernel<false, true, true>, 64, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int m = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&m, recurrences_kernel<true, true, true>, 64, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n < m ? n : m;
}

// The persistent schedule of stage B.  Measured on this GPU: a time-parallel kernel at thousands of channels fills every CU
// for tens of microseconds and a recurrence wave that shares a CU with it runs 2-3x slower; stream events between many
// queues cost 60-160 us each (they are cheap only between two or three queues).  So the recurrences of the WHOLE call are
// one persistent kernel on a CU set of their own, the time-parallel kernels run on one other stream, and the two sides
// meet through progress words in device memory --
//   ts   : start gate | disc(0..2) | per chunk c: low-pass(c) [lock(c)], disc(c+3), mix(c-1) [integrator(c-1)]
//   rs   : AFC(c) <- disc(c) word;  PLL <- AFC;  lock <- PLL;  integrator(c) <- low-pass(c) word;  de-emphasis(c) <- mix(c) word
// Stream events remain only at the two ends of the call.
static void launch_demod_persistent(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, int C, hipStream_t s,
                                    const DemodStreams &DS) {
    const int64_t nj = G.J1 - G.J0;
    constexpr int FIRST_CHUNK = 256;
    const int groups = (C + 63) / 64;
    ChunkPlan P{};
    for (int64_t rc0 = 0; rc0 < nj;) {
        const int c = P.n;
        const int cap = (c == 0 && nj > 2 * FIRST_CHUNK) ? FIRST_CHUNK : PB_CHUNK;
        const int len = (int)((nj - rc0) < cap ? (nj - rc0) : cap);
        P.rc0[c] = (int)rc0; P.len[c] = len;
        P.nb_disc[c] = ((len + DISC_ROWS - 1) / DISC_ROWS) * ((C + DISC_CH - 1) / DISC_CH); P.nb_mix[c] = ((len + MIX_ROWS - 1) / MIX_ROWS) * ((C + MIX_CH - 1) / MIX_CH);
        P.nb_fir[c] = ((len + PSS_TILE - 1) / PSS_TILE) * C;
        P.n++; rc0 += len;
    }
    { static const int rm = getenv("FMX_DEBUG_ROLE_MASK") ? atoi(getenv("FMX_DEBUG_ROLE_MASK")) : 31; P.role_mask = rm; }
    DemodSync *S = DS.sync;
    note_hip(hipMemsetAsync(S, 0, sizeof(DemodSync) + sizeof(int) * PB_ROLES * groups, s));
    hipLaunchKernelGGL(sync_init_kernel, dim3(1), dim3(64), 0, s, S, groups, DS.host_flag); FMX_LAUNCHED();
    hipEvent_t e0 = DS.ev[(*DS.ev_next)++ % DS.nev];
    note_hip(hipEventRecord(e0, s));                                   // the front kernel's output and the cleared words
    note_hip(hipStreamWaitEvent(DS.rs, e0, 0)); note_hip(hipStreamWaitEvent(DS.ts, e0, 0));
    const bool plldec = B.w_iq != nullptr;
    if (T.trig2 && T.wrap32_ok) { if (plldec) launch_recurrences<true, true, true>(T, B, G, C, P, S, groups, DS.rs); else launch_recurrences<false, true, true>(T, B, G, C, P, S, groups, DS.rs); }
    else if (T.trig2) { if (plldec) launch_recurrences<true, true, false>(T, B, G, C, P, S, groups, DS.rs); else launch_recurrences<false, true, false>(T, B, G, C, P, S, groups, DS.rs); }
    else { if (plldec) launch_recurrences<true, false, false>(T, B, G, C, P, S, groups, DS.rs); else launch_recurrences<false, false, false>(T, B, G, C, P, S, groups, DS.rs); }
    hipStream_t tq = DS.ts;
    int *prog = &S->prog[0][0];
    // the completion word of a kernel travels with the NEXT kernel on the stream (TSync::sig)
    int *pend_p = nullptr; int pend_v = 0;
    static const bool nogate = getenv("FMX_DEBUG_NO_GATES") != nullptr;   // diagnostics (timing only, results invalid): the time-parallel kernels do not wait for the recurrences
    auto ysync = [&](int role, int need) {
        TSync Y{};
        static const bool t_empty = getenv("FMX_DEBUG_T_EMPTY") != nullptr;   // diagnostics: the time-parallel kernels do nothing (the recurrences' own speed)
        Y.skip = t_empty ? 1 : 0;
        Y.gate = (role >= 0 && !nogate) ? prog + role * groups : nullptr; Y.need = need; Y.sig = pend_p; Y.sigv = pend_v; Y.abort_flag = &S->abort;
        pend_p = nullptr;
        return Y;
    };
    auto disc = [&](int c) {
        hipLaunchKernelGGL(disc_kernel, dim3((unsigned)((P.len[c] + DISC_ROWS - 1) / DISC_ROWS), (unsigned)((C + DISC_CH - 1) / DISC_CH)), dim3(256), 0, tq, T, B, G, C, (int64_t)P.rc0[c], P.len[c], ysync(-1, 0)); FMX_LAUNCHED();
        pend_p = &S->cnt_disc[c]; pend_v = P.nb_disc[c];
    };
    // One stream for all the time-parallel kernels.  Chunks are at most HALF the PSS feedback lag long, so the low-pass of
    // chunk c reads s-ring entries the mix wrote no later than chunk c - 2: it does not wait for the integrator of chunk
    // c - 1, the integrator runs chunk after chunk without a gap, and the order below only has to keep every kernel behind
    // its producers:  low-pass(c), disc(c + 3), mix(c - 1).  (The de-emphasis role writes the d ring itself.)
    static const int LEAD = getenv("FMX_DISC_LEAD") ? atoi(getenv("FMX_DISC_LEAD")) : 3;
    auto fir = [&](int c) {
        DeviceBuffers Bc = B;
        Bc.w_err = B.w_err + (size_t)(c & 1) * (PB_CHUNK / WT) * G.pitch * WT;
        hipLaunchKernelGGL(pss_fir_kernel, dim3((unsigned)((P.len[c] + PSS_TILE - 1) / PSS_TILE), (unsigned)C), dim3(64), 0, tq, T, Bc, G, C, (int64_t)P.rc0[c], P.len[c], ysync(2, c + 1)); FMX_LAUNCHED();
        pend_p = &S->cnt_fir[c]; pend_v = P.nb_fir[c];
    };
    auto mix = [&](int c) {
        hipLaunchKernelGGL(pss_mix_kernel, dim3((unsigned)((P.len[c] + MIX_ROWS - 1) / MIX_ROWS), (unsigned)((C + MIX_CH - 1) / MIX_CH)), dim3(256), 0, tq, T, B, G, C, (int64_t)P.rc0[c], P.len[c], ysync(3, c + 1)); FMX_LAUNCHED();
        pend_p = &S->cnt_mix[c]; pend_v = P.nb_mix[c];
    };
    // FMX_DEBUG_FORCE_STALL (tests): the gate asks for one workgroup more than exist, so the pipeline gives up after ~2 s
    const int force_stall = getenv("FMX_DEBUG_FORCE_STALL") ? atoi(getenv("FMX_DEBUG_FORCE_STALL")) : 0;
    hipLaunchKernelGGL(start_gate_kernel, dim3(1), dim3(64), 0, tq, S, PB_ROLES * groups + (force_stall ? 1 : 0)); FMX_LAUNCHED();
    for (int c = 0; c < LEAD && c < P.n; c++) disc(c);
    for (int c = 0; c < P.n + 1; c++) {
        if (c < P.n) fir(c);
        if (c + LEAD < P.n) disc(c + LEAD);
        if (c >= 1 && c - 1 < P.n) mix(c - 1);
    }
    if (pend_p) hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(64), 0, tq, pend_p, pend_v); FMX_LAUNCHED();
    hipStream_t ends[2] = { DS.rs, DS.ts };
    for (hipStream_t q : ends) {
        hipEvent_t e = DS.ev[(*DS.ev_next)++ % DS.nev];
        note_hip(hipEventRecord(e, q));
        note_hip(hipStreamWaitEvent(s, e, 0));
    }
}

void launch_demod(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, int C, hipStream_t s,
                  const DemodStreams &DS) {
    const int64_t nj = G.J1 - G.J0;
    if (nj <= 0) return;
    const dim3 tiles((unsigned)((nj + 63) / 64), (unsigned)((C + 63) / 64));
    const dim3 lanes((unsigned)((C + 63) / 64));
    // The recurrences are latency-bound and use a handful of wavefronts each, so the chunked stages run as a
    // software pipeline on five streams:  AFC(c+3) || PLL(c+2) || lock(c+1) || PSS loop(c) || de-emphasis(c-1).
    // Chunks are rows of the same work arrays, so nothing is double-buffered.
    hipStream_t st[5] = { s, DS.side[0] ? DS.side[0] : s, DS.side[1] ? DS.side[1] : s, DS.side[2] ? DS.side[2] : s,
                          DS.side[3] ? DS.side[3] : s };
    hipStream_t s3 = st[4];
    auto hand_over = [&](int from, int to, int c) {
        if (st[from] == st[to]) return;
        hipEvent_t e = DS.ev[(4 * c + from) % DS.nev];
        note_hip(hipEventRecord(e, st[from]));
        note_hip(hipStreamWaitEvent(st[to], e, 0));
    };
    // The first chunk is short: the five-stage pipeline fills in the time of 256 samples instead of 1744 (the call's
    // first PSS kernel starts ~0.2 ms earlier); any chunk length <= PSS_CHUNK that is a multiple of the tile is valid.
    constexpr int FIRST_CHUNK = 256;
    int c = 0;
    if (DS.partitioned && nj <= (int64_t)(PB_MAX_CHUNKS - 2) * PB_CHUNK) { launch_demod_persistent(T, B, G, C, s, DS); return; }
    for (int64_t rc0 = 0; rc0 < nj; c++) {
        const int cap = (c == 0 && nj > 2 * FIRST_CHUNK) ? FIRST_CHUNK : PSS_CHUNK;
        const int len = (int)((nj - rc0) < cap ? (nj - rc0) : cap);
        const int last = (rc0 + len >= nj) ? 1 : 0;
        // the discriminator runs per chunk in front of the AFC (a stage with time to spare), so the first PSS kernel
        // starts after 256 rows of it instead of after the whole call's
        hipLaunchKernelGGL(disc_kernel, dim3((unsigned)((len + DISC_ROWS - 1) / DISC_ROWS), (unsigned)((C + DISC_CH - 1) / DISC_CH)), dim3(256), 0, st[0], T, B, G, C, rc0, len, TSync{}); FMX_LAUNCHED();
        if (B.w_iq) hipLaunchKernelGGL(afc_kernel<true>, lanes, dim3(64), 0, st[0], T, B, G, C, rc0, len);
        else hipLaunchKernelGGL(afc_kernel<false>, lanes, dim3(64), 0, st[0], T, B, G, C, rc0, len); FMX_LAUNCHED();
        hand_over(0, 1, c);
        if (T.trig2 && T.wrap32_ok) hipLaunchKernelGGL((pll_kernel<true, true>), lanes, dim3(64), 0, st[1], T, B, G, C, rc0, len);
        else if (T.trig2) hipLaunchKernelGGL((pll_kernel<true, false>), lanes, dim3(64), 0, st[1], T, B, G, C, rc0, len);
        else hipLaunchKernelGGL((pll_kernel<false, false>), lanes, dim3(64), 0, st[1], T, B, G, C, rc0, len); FMX_LAUNCHED();
        hand_over(1, 2, c);
        hipLaunchKernelGGL(lock_kernel, lanes, dim3(64), 0, st[2], T, B, G, C, rc0, len); FMX_LAUNCHED();
        hand_over(2, 3, c);
        hipLaunchKernelGGL(pss_fir_kernel, dim3((unsigned)((len + PSS_TILE - 1) / PSS_TILE), (unsigned)C), dim3(64), 0, st[3], T, B, G, C, rc0, len, TSync{}); FMX_LAUNCHED();
        hipLaunchKernelGGL(pss_acc_kernel, lanes, dim3(64), 0, st[3], T, B, G, C, rc0, len); FMX_LAUNCHED();
        hipLaunchKernelGGL(pss_mix_kernel, dim3((unsigned)((len + MIX_ROWS - 1) / MIX_ROWS), (unsigned)((C + MIX_CH - 1) / MIX_CH)), dim3(256), 0, st[3], T, B, G, C, rc0, len, TSync{}); FMX_LAUNCHED();
        hand_over(3, 4, c);
        hipLaunchKernelGGL(deemph_kernel, lanes, dim3(64), 0, st[4], T, B, G, C, rc0, len, last); FMX_LAUNCHED();
        rc0 += len;
    }
    if (s3 != s) { note_hip(hipEventRecord(DS.join, s3)); note_hip(hipStreamWaitEvent(s, DS.join, 0)); }
}

}  // namespace fmx
