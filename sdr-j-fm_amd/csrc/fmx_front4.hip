// fmx_front4.hip -- stage A for the batches that fill the chip, the input FIR on the MATRIX pipe.
//
// Replaces, per channel and per call (as front_kernel does):
//   RF DC removal            fm-processor.cpp:423-446   (applied behind the filter, fmx_front.hip)
//   IQ balance               fm-processor.cpp:462-464
//   inputFilter (251 taps)   fm-processor.cpp:469-470, fft-filters.cpp:132-163
//   fmBand_1 (25 taps, /6)   fm-processor.cpp:472,  fir-filters.cpp:397-424
//   fmBand_2 (3 taps, /2)    fm-processor.cpp:474
//
// Why.  front_kernel (and its six-wave sibling, tools/experiments/fmx_front3.hip) spend 600 v_pk_fma_f32 per 1536-sample tile on the folded 287-tap filter, and a
// kernel of packed FMAs runs this GPU into its power limit: 93 TFLOP/s sustained of the nominal 157 (tools/ubench/pkfma_clock.hip).  With
// every load compiled out front3_kernel takes 1.67 ms per launch at 4096 channels, with everything BUT the loads compiled out 1.49 ms
// (tools/diag/f3_ablate.sh): the stage sits at the vector ALU's power limit, not at the HBM stream's.  The f32 matrix instruction is no way
// out (round 2, fmx_front2.hip: the Toeplitz form wastes 38 % of its multiplies, and it runs on the same f32 data path).  The f16 matrix
// instruction is: sixteen times the rate, so the filter can afford both the Toeplitz zeros and a SPLIT of every operand into two halves --
//     x * 2^e = xh + xl,   t * 2^14 = th + tl       (xh, th: the value rounded to f16; xl, tl: the remainder rounded to f16)
//     sum t x  ~  2^-(e + 14) sum (th xh + th xl + tl xh)  (+ tl xl with FMX_F4_TERMS = 4)
// -- products of f16 values are exact in the f32 accumulator, so what is lost is the rounding of the remainders and the dropped tl xl: 2^-21
// of a product, the same order as the f32 filter's own rounding (measured against front_kernel: tests/test_gpu_round5.py).
// BLOCK FLOATING POINT (round 6): e is the TILE's -- 2^e max |x| in [2^14, 2^15) over the tile's 1536 balanced samples, found by the wave that
// holds them (24 v_max3, one DPP reduction) -- so the filter is as linear as the reference's f32 one at any level a device or a file delivers:
// nothing is clamped, a sample keeps 22 bits whatever the stream's amplitude, and a remainder that falls into f16's subnormals is below
// 2^-39 of the tile's largest sample.  A window that begins in the previous tile (the first two column blocks: K-steps 0 .. 8 / 0 .. 2) sums
// that part in the previous tile's unit and is rescaled by the exact power of two where the tiles' exponents differ (wave-uniform test; the
// exponents ride with the scatter's sequence counters).  The IQ balance (fm-processor.cpp:462-464) is applied IN FRONT of the split, where the
// reference has it: x (att 2^e) = RN (x att) 2^e, the reference's own product.
//
// Layout.  K of the matrix product is TIME: the Toeplitz matrix A[i][k] = G[12 i + off + 288 - k] of the folded taps G (16 adjacent outputs i
// against the 480 samples k of their 40-column window) times B[k][n] = the window of column block b, component comp, n = 2 b + comp.  So the
// LDS image is simply the de-interleaved stream -- four linear planes of f16 (hi re, hi im, lo re, lo im) over a RING of the channel's last six
// tiles: a tile's filter reads its 288 history samples in place in front of its own, the scatter is four linear 4-byte writes per sample pair
// (one address register), a lane's operand is 16 consecutive bytes of a plane (one address register for all of a tile's 30 reads).  The column
// sums the RF DC recurrence needs come off the same operands: two more matrix instructions per K-step with a boxcar of ones for A.
//   per tile and wave: 45 + 12 v_mfma_f32_16x16x32_f16, 66 LDS operand reads, ~300 plain VALU instructions (conversion, DC scan, epilogue)
// instead of 600 packed FMAs and ~370 others.
// Six waves per channel, two channels per workgroup, one workgroup per CU, the waves of a channel staggered through its tiles as in
// tools/experiments/fmx_front3.hip (scatter, prefetch of the wave's next tile, filter, DC recurrence through a mailbox, output).
// Where the time goes (tools/diag/f4_ablate.sh, 4096 channels, ms per launch on a box that runs the kernel in 1.50): no loads and no stores 0.97 --
// no stores 1.25 -- no matrix instructions 1.50 -- everything 1.50.  The kernel runs at the rate this GPU streams its traffic mix at: twelve bytes
// read for one written is 5.4-5.5 TB/s in a kernel that does nothing else, whatever the pattern (linear or 512 streams), the waves per CU (8 .. 64),
// the form of the store (tools/ubench/stream_shape.hip: reads alone 6.8-7.0); bench.py's `frac_of_measured` is 1.0.  Tried on that evidence and
// without effect, so not kept: the outputs in lane order through the LDS crossbar (whole 128-byte lines per quarter-wave), the store an iteration
// late (in front of the next loads, so that no wait covers a fresh store), nontemporal stores (slower), a tile-major ring, twelve waves on one channel.
#include "fmx_internal.h"
#include "fmx_front_dc.h"

// This source is compiled TWICE (round 6): as it stands -- namespace f4, six waves per channel and two channels per workgroup, real taps: the headline's
// kernel -- and from fmx_front4lo.hip with F4_LO = 1 -- namespace f4lo, TWELVE waves on ONE channel per workgroup (the same LDS, room for two more tap
// tables; 256 channels are one workgroup per CU), for handles with LOCAL OSCILLATORS: BASELINE configs[2], 256 carriers in 24 wide-band streams.  There the mix
// v[n] = x'[n] LO[n] (fm-processor.cpp:466, oscillator.cpp:49-58: LO[n] = table[(P0 - (n + 1) lo) mod R], so LO[n - m] = LO[n] table[(m lo) mod R]) leaves
// the samples alone and goes into the TAPS:
//     z[q] = cg sum_m G[m] v[n - m]  =  cg LO[n] sum_m (G[m] table[(m lo) mod R]) x'[n - m],     n = 12 q + off,
// a complex tap set per channel (built here, once per launch, from the real one and the reference's own oscillator table) against the UNROTATED
// samples -- twice the matrix instructions, which the stage has to spare, and nothing per sample: the scatter, the f16 split, the RF DC column sums are the
// real-tap kernel's.  The output takes the oscillator's value at its newest sample (four table entries per lane and tile); the RF DC term the reference
// subtracts in front of the mix becomes (c att) sum_m Gc[m] = (c att) Hlo, Hlo a complex constant of the channel (ChanParams::hlo_*, host side).
#ifndef F4_NS
#define F4_NS f4
#define F4_NW 6        /* (12 waves on ONE channel per workgroup, F4_NW = 12, F4_CPW = 1: the same time per launch) */
#define F4_CPW 2
#define F4_LO 0
#define F4_FN(name) name
#endif
namespace fmx {
namespace F4_NS {

constexpr bool LO = F4_LO != 0;                // complex taps for channels with a local oscillator
constexpr int NW = F4_NW;                      // waves (= ring slots) per channel
constexpr int CPW = F4_CPW;                    // channels per workgroup
constexpr int NTHR = 64 * NW * CPW;
constexpr int WCOLS = 128;                     // columns (= outputs) per tile
constexpr int WSAMP = WCOLS * DECIM;           // 1536 input samples per tile
constexpr int SPT = 2 * DECIM;                 // 24 samples per lane per tile
constexpr int HL = A_HIST_COLS - 1;            // 24 history columns in front of a tile
constexpr int HS = HL * DECIM;                 // = 288 samples
constexpr int PL = NW * WSAMP + HS;            // samples per plane: the ring, and in front of it a mirror of its last 288 samples (slot 0's history)
constexpr int PLB = PL * 2;                    // bytes per plane (19008)
constexpr int KSTEPS = (HL + 16) * DECIM / 32; // 15 K-steps of 32 samples over the 480-sample window
constexpr int KSUM0 = HS / 32;                 // the block's own 16 columns begin at K-step 9
constexpr int TA_N = 664;                      // entries of a reversed tap table: u = 32 j + 8 kg + e + 180 - 12 i in [0, 660)
constexpr int MB_N = 16;                       // mailbox slot: [0..12] RfDC in front of columns -13 .. -1 of the next tile, [13] in front of its column 0 (= the carry)
#ifndef FMX_F4_TERMS
#define FMX_F4_TERMS 3
#endif
constexpr float TSC = 16384.f;                 // pre-scale of the taps (2^14); the samples' is the tile's own 2^e (block floating point)
// a tile's scale from the biased exponent E of its largest balanced sample (clamped to [29, 254]: every factor below stays a normal float):
// 2^e with e = 141 - E puts that sample into [2^14, 2^15); 2^-e; 2^-(e + 14) takes the accumulator back
__device__ __forceinline__ int   bfp_E(float m) { const int E = (int)((__float_as_uint(m) >> 23) & 255u); return E < 29 ? 29 : (E > 254 ? 254 : E); }
__device__ __forceinline__ float bfp_scale(int E) { return __uint_as_float((uint32_t)(268 - E) << 23); }
__device__ __forceinline__ float bfp_inv(int E) { return __uint_as_float((uint32_t)(E - 14) << 23); }
__device__ __forceinline__ float bfp_out(int E) { return __uint_as_float((uint32_t)(E - 28) << 23); }
// 2^(e_cur - e_prev) = 2^(E_prev - E_cur): what takes a sum in the previous tile's unit into this tile's (the exponent limited to a float's)
__device__ __forceinline__ float bfp_ratio(int E_prev, int E_cur) { int d = E_prev - E_cur; d = d < -126 ? -126 : (d > 127 ? 127 : d); return __uint_as_float((uint32_t)(127 + d) << 23); }
// largest |.| of a wave's values, in every lane's SGPR copy
__device__ __forceinline__ float wave_max(float v) {
#define F4_MAX_STEP(ctrl, rmask) v = __builtin_fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, false)));
    F4_MAX_STEP(0x111, 0xf) F4_MAX_STEP(0x112, 0xf) F4_MAX_STEP(0x114, 0xf) F4_MAX_STEP(0x118, 0xf) F4_MAX_STEP(0x142, 0xa) F4_MAX_STEP(0x143, 0xc)
#undef F4_MAX_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
static_assert(HS % 32 == 0 && KSTEPS == 15 && PLB % 16 == 0 && (TA_N * 2) % 16 == 0, "geometry");

typedef _Float16 h16;
typedef h16 v8h __attribute__((ext_vector_type(8)));
typedef h16 v2h __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bperm(int src_lane, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v))); }
__device__ __forceinline__ float dpp_swap1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)); }   // quad_perm [1,0,3,2]
template <int SH> __device__ __forceinline__ float dpp_row_shr(float v) {        // lanes shifted right by SH within their row of 16, zeros shifted in
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + SH, 0xf, 0xf, true));
}
// a sample pair of one component times the tile's factor (balance x 2^e): the two f16 roundings and the two f16 remainders, packed.
// The product is OPAQUE to the compiler behind its f32 rounding: with a factor that is not a power of two it is inexact, and hipcc (ROCm 7.2) otherwise
// folds it into one of the two conversions -- the stored half RN16 (RN32 (x s)) by v_cvt_pk_f16_f32, the half it subtracts RN16 (x s) by
// v_fma_mixlo_f16 -- one f16 ulp apart at every double-rounding tie: one sample in 2^13 lost its low half, 1e-4 of an output (found by
// tools/diag/f4_bal.py; `#pragma clang fp contract(off)` does not stop the fold).  So: rounded to f32 once -- the reference's own x * Lgain
// (fm-processor.cpp:462-464) -- and both halves taken from that value.
__device__ __forceinline__ void split2(float x0, float x1, float sc, uint32_t *hi, uint32_t *lo) {
    float a = x0 * sc, b = x1 * sc;
#ifndef F4_SPLIT_PLAIN
    // five instructions per pair: the two products, ONE packed conversion for the stored halves, and the remainders straight from the f32 products and the
    // packed halves by the mixed-precision fma (a * 1.0 - (f32) half, rounded to f16: the difference itself is exact in f32)
    uint32_t h, l;
    asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
                 "v_fma_mixlo_f16 %1, %2, 1.0, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
                 "v_fma_mixhi_f16 %1, %3, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                 : "=&v"(h), "=&v"(l) : "v"(a), "v"(b));
    *hi = h; *lo = l;
#else
    asm volatile("" : "+v"(a), "+v"(b));
    const h16 ha = (h16)a, hb = (h16)b;
    const h16 la = (h16)(a - (float)ha), lb = (h16)(b - (float)hb);
    *hi = __builtin_bit_cast(uint32_t, (v2h){ha, hb});
    *lo = __builtin_bit_cast(uint32_t, (v2h){la, lb});
#endif
}

#ifndef F4_ABL
#define F4_ABL 0      /* diagnostic builds only: bit 0 no scatter, 1 no DC pass, 2 no matrix FIR, 3 no tile loads behind the first, 4 no output stores (garbage results) */
#endif

// FMT: fmx_iq_format of the input (include/fmx.h).  Raw integer samples are converted while they are loaded -- (u8 - 127) / 128, s8 / 128,
// s16 / denominator, all exact as in the reference's device handlers (rtlsdr-handler.cpp:291, hackrf-handler.cpp:364, lime-handler.cpp:250).
template <int FMT, bool NTL>
__global__ __launch_bounds__(NTHR, 3) __attribute__((amdgpu_waves_per_eu(3, 3))) void front4_kernel(DeviceTables T, DeviceBuffers B, CallGeom G,
                                                                                                       const void *__restrict__ iq_raw) {
    constexpr int BPS = (FMT == 0) ? 8 : (FMT == 3 ? 4 : 2);          // bytes per complex sample
    struct __attribute__((aligned(16))) ChanLds {      // (a multiple of 16 bytes: the second channel's planes are read 16 bytes at a time -- an 8-byte
                                                       // member added in round 6 cost the stage 25 % until this said so)
        h16 pl[4][PL];                     // hi re, hi im, lo re, lo im of the channel's newest six tiles (and the mirror)
        h16 ta[LO ? 5 : 3][TA_N];          // reversed tap table: hi, lo, and the boxcar of ones that sums a column; LO: the taps' imaginary parts, hi and lo
        float2 mb[8][MB_N];                // RfDC boundaries behind tile ti, slot = ti & 7
        int carry_seq;                     // tiles whose mailbox slot is published
        int scat_seq[NW], fir_seq[NW];     // per wave: tiles scattered / tiles whose filter has read everything, + 1
        int texp[NW + 2];                     // biased exponent (bfp_E) of the scale of the tile in ring slot w (written with the slot, read by the next tile's filter
                                           // like the slot's last 288 samples: the same counters order both); [NW]: of the call's history
        int pad_[3];
        float2 hb[LO ? 28 : 2];            // LO, a call's last tile: RfDC in front of its columns 103 .. 128 (the processed history the next call finds)
    };
    __shared__ __attribute__((aligned(16))) ChanLds Lall[CPW];
    static_assert(sizeof(ChanLds) % 16 == 0, "every channel's planes on a 16-byte boundary");

    const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x / (64 * NW));
    const int ch_raw = (int)blockIdx.x * CPW + half;
    const bool active = ch_raw < G.channels;                               // (an odd channel count: the last workgroup's second half has nothing to do)
    const int ch = active ? ch_raw : G.channels - 1;
    const int t = (int)threadIdx.x - half * (64 * NW);                     // thread index within the channel's six waves
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    ChanLds &L = Lall[half];
    float2 (&mb)[8][MB_N] = L.mb;
    int &carry_seq = L.carry_seq;
    int (&scat_seq)[NW] = L.scat_seq;
    int (&fir_seq)[NW] = L.fir_seq;
    char *const plc = reinterpret_cast<char *>(&L.pl[0][0]);
    const char *const tac = reinterpret_cast<const char *>(&L.ta[0][0]);
    const ChanParams P = B.params[ch];
    const FrontSet FS = T.front_sets[P.front_set];
    const char *__restrict__ inb = reinterpret_cast<const char *>(iq_raw) + (size_t)P.stream * G.stream_stride * BPS;
    const float qs = G.iq_scale;
    ChanState *st = B.state + ch;
    float2 *hist = B.hist + (size_t)ch * DECIM * A_HIST_COLS;
    float2 *zring = B.zring + (size_t)ch * (G.ring_mask + 1);

    // Call-local geometry (front_kernel's, with the call starting on a column boundary and ending on a tile boundary)
    const int64_t qa = G.g0 / 12;
    const int NT = (int)(G.n / WSAMP);
    const int off = FS.off;
    const int ja = (int)((G.g0 - off + 11) / 12 - qa);              // first output completed by this call
    const int jb = (int)((G.g0 + G.n - off + 11) / 12 - qa);        // one past the last
    const int zr0 = (int)((qa + FS.zshift) & (int64_t)G.ring_mask);

    // ---- tap tables: entry u holds G[M0 - u], G the folded filter in time order (G[12 d + off - r] = Trd[r][d])
    {
        const int M0 = off + 468;
        for (int u = t; u < TA_N; u += 64 * NW) {
            const int m = M0 - u;
            float g = 0.f;
            if (m >= 0) {
                const int r = ((off - m) % DECIM + DECIM) % DECIM, d = (m - off + r) / DECIM;
                if (d < A_MAX_ND) g = T.front_taps[(size_t)P.front_set * A_TAPS_DEV + r * A_TAPS_ROW + d];
            }
            float gs = g * TSC;
            if constexpr (LO) {
                // Gc[m] = G[m] table[(m lo) mod R]: the oscillator's value m samples back, relative to its value at the newest sample
                float gi = 0.f;
                if (P.lo_freq != 0 && T.lo_table != nullptr && m >= 0) {
                    long long ph = ((long long)m * (long long)P.lo_freq) % (long long)G.input_rate;
                    if (ph < 0) ph += G.input_rate;
                    const float2 w = T.lo_table[ph];
                    gi = gs * w.y; gs = gs * w.x;
                }
                const h16 ih = (h16)gi;
                L.ta[3][u] = ih;
                L.ta[4][u] = (h16)(gi - (float)ih);
            }
            const h16 gh = (h16)gs;
            L.ta[0][u] = gh;
            L.ta[1][u] = (h16)(gs - (float)gh);
            L.ta[2][u] = (m >= off - (DECIM - 1) && m <= off) ? (h16)1.0f : (h16)0.0f;      // the 12 samples of the output's own column
        }
    }
    const int lo = LO ? P.lo_freq : 0;
    const bool mix = LO && lo != 0 && T.lo_table != nullptr;       // this channel's oscillator runs (the same for the whole workgroup half)
    const int Rr = G.input_rate;
    // oscillator table index at call-relative sample s: (P0 - (s + 1) lo) mod R
    auto lo_idx = [&](long long s_rel, int P0) -> int {
        long long ph = ((long long)P0 - (s_rel + 1) * (long long)lo) % (long long)Rr;
        if (ph < 0) ph += Rr;
        return (int)ph;
    };
    if (t == 0) { carry_seq = 0; for (int i = 0; i < NW; i++) { scat_seq[i] = 0; fir_seq[i] = 0; } }
    // ---- the call's history (raw samples, or what front_kernel's conversions make of them: see there) -> the mirror in front of ring slot 0
    const bool dc_rst = (P.actions & ACT_DC_RESET) != 0;           // setDCRemove zeroes RfDC (:922-925)
    const int hist_fmt0 = st->hist_fmt, lo_phase0 = st->lo_phase;
    const float st_dc_re = st->dc_re, st_dc_im = st->dc_im;
    const float2 R0 = (T.lo_table != nullptr && lo_phase0 != 0) ? T.lo_table[lo_phase0] : make_float2(1.f, 0.f);   // an oscillator set back to 0 Hz keeps its phase
    const bool hist_to_raw = (hist_fmt0 == 1) && !mix;             // the LO was switched off in front of this call
    const bool hist_rst = (hist_fmt0 == 0) && dc_rst;
    const bool dcr = P.dc_remove != 0;
    const float2 dc_now = (dc_rst || !dcr) ? make_float2(0.f, 0.f)
                                           : make_float2(__builtin_amdgcn_fmed3f(st_dc_re, -0.01f, 0.01f), __builtin_amdgcn_fmed3f(st_dc_im, -0.01f, 0.01f));
    const float2 *dcvR = B.dcv_hist + (size_t)ch * DCV_SAVE;
    if (wave == 0) {
        constexpr int HQ = (DECIM * A_HIST_COLS + 63) / 64;
        float2 hv[HQ];
        float hm = 0.f;
#pragma unroll
        for (int q = 0; q < HQ; q++) {
            const int i = lane + 64 * q;
            const int r = i / A_HIST_COLS, c = i - r * A_HIST_COLS;
            float2 v = make_float2(0.f, 0.f);
            if (i < DECIM * A_HIST_COLS && c != HL) {              // (c == HL: the partial column of a call that starts inside one: never here)
                v = hist[i];
                if (mix) {
                    // A running oscillator: the reference's filter memory holds p = ((x - RfDC) att) LO [s] of every history sample -- what front_kernel keeps for
                    // such a channel (hist_fmt 1), and what a history of raw samples (hist_fmt 0: the oscillator stood at R0 while they came) amounts to with
                    // R0 in LO's place.  The complex taps turn every sample of the window by the oscillator's own steps, so the sample that stands for p here is
                    // x_eq = (p conj (LO [s])) / att + RfDC now.
                    float2 p = v;
                    if (hist_fmt0 != 1) {
                        const int tb = c - HL + 13;
                        const float2 d = dcvR[tb < 0 ? 0 : tb];
                        const bool sub = dcr || dc_rst;           // (as front_kernel's conversion: the boundaries saved with the history)
                        const float qx = (v.x - (sub ? __builtin_amdgcn_fmed3f(d.x, -0.01f, 0.01f) : 0.f)) * P.att_l;
                        const float qy = (v.y - (sub ? __builtin_amdgcn_fmed3f(d.y, -0.01f, 0.01f) : 0.f)) * P.att_r;
                        p = make_float2(qx * R0.x - qy * R0.y, qx * R0.y + qy * R0.x);
                    }
                    const float2 w = T.lo_table[lo_idx((long long)(DECIM * (c - HL) + r), lo_phase0)];
                    v = make_float2((p.x * w.x + p.y * w.y) / P.att_l + dc_now.x, (p.y * w.x - p.x * w.y) / P.att_r + dc_now.y);
                } else
                if (hist_rst) {
                    const int tb = c - HL + 13;
                    const float2 d = dcvR[tb < 0 ? 0 : tb];
                    v.x -= __builtin_amdgcn_fmed3f(d.x, -0.01f, 0.01f);
                    v.y -= __builtin_amdgcn_fmed3f(d.y, -0.01f, 0.01f);
                } else if (hist_to_raw) {
                    v = make_float2(v.x * R0.x + v.y * R0.y, v.y * R0.x - v.x * R0.y);
                    v.x = v.x / P.att_l + dc_now.x;                // (the balance is never 0 here: fmx_api.hip, front4_ok)
                    v.y = v.y / P.att_r + dc_now.y;
                }
            }
            hv[q] = v;
            hm = __builtin_fmaxf(hm, __builtin_fmaxf(__builtin_fabsf(v.x * P.att_l), __builtin_fabsf(v.y * P.att_r)));
        }
        const int Eh = bfp_E(wave_max(hm));
        const float hsl = P.att_l * bfp_scale(Eh), hsr = P.att_r * bfp_scale(Eh);     // IQ balance :462-464 and the block's scale, one exact factor
#pragma unroll
        for (int q = 0; q < HQ; q++) {
            const int i = lane + 64 * q;
            const int r = i / A_HIST_COLS, c = i - r * A_HIST_COLS;
            if (i < DECIM * A_HIST_COLS && c != HL) {
                const int s = DECIM * c + r;                       // sample of the 288 in front of the call
                uint32_t hr2, lr2, hi2, li2;
                split2(hv[q].x, 0.f, hsl, &hr2, &lr2);
                split2(hv[q].y, 0.f, hsr, &hi2, &li2);
                reinterpret_cast<uint16_t *>(&L.pl[0][0])[s] = (uint16_t)(hr2 & 0xffffu); reinterpret_cast<uint16_t *>(&L.pl[1][0])[s] = (uint16_t)(hi2 & 0xffffu);
                reinterpret_cast<uint16_t *>(&L.pl[2][0])[s] = (uint16_t)(lr2 & 0xffffu); reinterpret_cast<uint16_t *>(&L.pl[3][0])[s] = (uint16_t)(li2 & 0xffffu);
            }
        }
        if (lane == 0) L.texp[NW] = Eh;
    }
    // RfDC in front of the 13 columns before this call's first column and of that column itself
    if (t < 14) mb[7][t] = (hist_to_raw || hist_rst || mix) ? make_float2(dc_rst ? 0.f : st_dc_re, dc_rst ? 0.f : st_dc_im) : dcvR[t];
    const float dc0r = dc_rst ? 0.f : st_dc_re, dc0i = dc_rst ? 0.f : st_dc_im;
    __syncthreads();                                  // the only workgroup barrier: tables, history and counters are set up

    // ---- per-lane constants.  Accumulator layout of the 16 x 16 matrix instruction: the lane holds rows 4 kg + v (v = 0..3) of column n.
    const int kg = lane >> 4, n = lane & 15, blk = n >> 1, comp = n & 1;
    const int c0col = 16 * blk + 4 * kg;              // the lane's first output column in the tile
    const float alpha = 1.0f / (float)G.input_rate;   // rfDcAlpha fm-processor.cpp:379
    const float2 R0g = mix ? make_float2(1.f, 0.f) : R0;          // (a running oscillator's value rides with every output, below)
    const float cg_re = (FS.gain_re * R0g.x - FS.gain_im * R0g.y), cg_im = (FS.gain_re * R0g.y + FS.gain_im * R0g.x);     // complex output gain x R0
    const float kown = cg_re, kpar = comp ? cg_im : -cg_im;       // z = a_own kown + a_partner kpar
    const float sgn = comp ? 1.0f : -1.0f;
    // the oscillator's table index at the newest sample of the lane's first output of the wave's first tile, and its steps per column and per round of tiles
    int ph_q0 = 0, ph_step12 = 0, ph_stepT = 0;
    if (LO && mix) {
        ph_q0 = lo_idx((long long)(DECIM * (wave * WCOLS + c0col) + off), lo_phase0);
        ph_step12 = (int)((((long long)DECIM * lo) % Rr + Rr) % Rr);
        ph_stepT = (int)((((long long)NW * WSAMP * lo) % Rr + Rr) % Rr);
    }
    const float bal = comp ? P.att_r : P.att_l;                   // IQ balance :462-464: applied with the tile's scale, in front of the filter
    const float ibal = 1.0f / bal;                                // (the column sums of the RF DC recurrence are wanted without it)
    const float hlo_re = (LO && mix) ? P.hlo_re * bal : 0.f, hlo_im = (LO && mix) ? P.hlo_im : 0.f;      // (sum of the complex taps; the own component's balance folded in)
    const float hsum = FS.hsum * bal, dcw = FS.dc_w;              // what the filter makes of a constant that is balanced like the samples
    // the RF DC recurrence over a tile (first order in alpha inside it, as front_kernel's fast path; the decay of the state over the tile and of
    // the samples' weights towards its end to third / second order: relative errors below 1e-10)
    const float ut = 1536.0f * alpha, u_tile = ut - 0.5f * ut * ut + (1.0f / 6.0f) * ut * ut * ut;       // 1 - (1 - alpha)^1536
    float wv[4];
#pragma unroll
    for (int v = 0; v < 4; v++) { const float x = (float)(DECIM * (WCOLS - 1 - (c0col + v)) + 6) * alpha; wv[v] = 1.0f - x + 0.5f * x * x; }   // (1 - alpha)^(samples behind the column's middle)
    const int pw = (wave + NW - 1) % NW, nw = (wave + 1) % NW;
    // operand addresses: B = 16 bytes of a plane at sample 1536 wave - 288 + 192 blk + 32 j + 8 kg; A = 16 bytes (8-byte aligned) of a tap table
    const char *const bB = plc + comp * PLB + 3072 * wave + 384 * blk + 16 * kg;
    const char *const aB = tac + 16 * kg + 360 - 24 * (lane & 15);
    char *const scW = plc + 2 * HS + 3072 * wave + 4 * lane;          // the lane's dword of sample pair lane (+ 256 k) of plane 0
    char *const scM = plc + 4 * lane - 192;                            // ... of the mirror, for the pairs 624 + (lane - 48) (+ 256 (k - 9))
    // RfDC boundaries the lane's outputs take: columns c0col - dck + (0..4), dck = 12: four values of lane srcA, one of srcB, or the previous tile's
    const bool prevA = (blk == 0 && kg < 3), prevB = (blk == 0 && kg < 2);
    const int srcA = (kg == 3) ? n : 16 * (kg + 1) + n - 2;
    const int srcB = (kg >= 2) ? 16 * (kg - 2) + n : 16 * (kg + 2) + n - 2;
    const int mbe = 4 * kg + 1;                                        // mailbox entry of column 4 kg - 12 of the tile

    float4 raw[SPT / 2];
    auto load_tile = [&](int ti) {
        const char *tb = inb + (size_t)ti * WSAMP * BPS;
        if (FMT == 0 && NTL) {
            typedef float v4f_ __attribute__((ext_vector_type(4)));
            const v4f_ *p4 = reinterpret_cast<const v4f_ *>(tb);
#pragma unroll
            for (int k = 0; k < SPT / 2; k++) { const v4f_ v = __builtin_nontemporal_load(p4 + lane + 64 * k); raw[k] = make_float4(v.x, v.y, v.z, v.w); }
        } else if (FMT == 0) {
            const float4 *p4 = reinterpret_cast<const float4 *>(tb);
#pragma unroll
            for (int k = 0; k < SPT / 2; k++) raw[k] = p4[lane + 64 * k];
        } else if (FMT == 1 || FMT == 2) {
            const uint32_t *p1 = reinterpret_cast<const uint32_t *>(tb);
#pragma unroll
            for (int k = 0; k < SPT / 2; k++) {
                const uint32_t w = p1[lane + 64 * k];                 // I0 Q0 I1 Q1
                if (FMT == 1)
                    raw[k] = make_float4((float)((int)(w & 255u) - 127) * qs, (float)((int)((w >> 8) & 255u) - 127) * qs,
                                         (float)((int)((w >> 16) & 255u) - 127) * qs, (float)((int)(w >> 24) - 127) * qs);
                else
                    raw[k] = make_float4((float)(int8_t)(w & 255u) * qs, (float)(int8_t)((w >> 8) & 255u) * qs,
                                         (float)(int8_t)((w >> 16) & 255u) * qs, (float)(int8_t)(w >> 24) * qs);
            }
        } else {
            const uint2 *p2 = reinterpret_cast<const uint2 *>(tb);
#pragma unroll
            for (int k = 0; k < SPT / 2; k++) {
                const uint2 w = p2[lane + 64 * k];                    // (I0 Q0) (I1 Q1)
                raw[k] = make_float4((float)(int16_t)(w.x & 0xffffu) * qs, (float)(int16_t)(w.x >> 16) * qs,
                                     (float)(int16_t)(w.y & 0xffffu) * qs, (float)(int16_t)(w.y >> 16) * qs);
            }
        }
    };
    const int NTa = active ? NT : 0;
    if (wave < NTa) load_tile(wave);

    for (int ti = wave; ti < NTa; ti += NW) {
        const int qt = ti * WCOLS;                    // first column of the tile
        // ---- split the samples and put them into the ring, once the next tile's filter (the wave behind this one) has read its history from
        //      what the slot held
        // the tile's scale: its largest balanced sample goes to [2^14, 2^15) (an infinite or NaN sample spreads over the filter's length as it does
        // in the reference: nothing limits)
        int Et;
        {
            float mr = 0.f, mi = 0.f;
#pragma unroll
            for (int k = 0; k < SPT / 2; k++) {
                mr = __builtin_fmaxf(mr, __builtin_fmaxf(__builtin_fabsf(raw[k].x), __builtin_fabsf(raw[k].z)));
                mi = __builtin_fmaxf(mi, __builtin_fmaxf(__builtin_fabsf(raw[k].y), __builtin_fabsf(raw[k].w)));
            }
            Et = bfp_E(wave_max(__builtin_fmaxf(mr * __builtin_fabsf(P.att_l), mi * __builtin_fabsf(P.att_r))));
        }
        const float sc_l = P.att_l * bfp_scale(Et), sc_r = P.att_r * bfp_scale(Et);      // x (att 2^e) = RN (x att) 2^e: the reference's product :462-464
        if (ti - NW + 1 >= 0) seq_wait(&fir_seq[nw], ti - NW + 2);
        if (!(F4_ABL & 1)) {
#pragma unroll
            for (int k = 0; k < SPT / 2; k++) {
                uint32_t hr, lr, hi, li;
                split2(raw[k].x, raw[k].z, sc_l, &hr, &lr);
                split2(raw[k].y, raw[k].w, sc_r, &hi, &li);
                *reinterpret_cast<uint32_t *>(scW + 256 * k) = hr;
                *reinterpret_cast<uint32_t *>(scW + 256 * k + PLB) = hi;
                *reinterpret_cast<uint32_t *>(scW + 256 * k + 2 * PLB) = lr;
                *reinterpret_cast<uint32_t *>(scW + 256 * k + 3 * PLB) = li;
                if (wave == NW - 1 && k >= 9 && (k > 9 || lane >= 48)) {      // the ring's last 288 samples once more in front of slot 0
                    *reinterpret_cast<uint32_t *>(scM + 256 * (k - 9)) = hr;
                    *reinterpret_cast<uint32_t *>(scM + 256 * (k - 9) + PLB) = hi;
                    *reinterpret_cast<uint32_t *>(scM + 256 * (k - 9) + 2 * PLB) = lr;
                    *reinterpret_cast<uint32_t *>(scM + 256 * (k - 9) + 3 * PLB) = li;
                }
            }
        }
        // last tile: the 24 newest columns are the next call's history -- the raw samples as they came (front_kernel's format: [r][c], an empty
        // partial column behind them)
        if (ti == NT - 1) {
#pragma unroll
            for (int k = 9; k < SPT / 2; k++) {
                const int s = 2 * (lane + 64 * k) - (WSAMP - HS);          // sample of the 288, even
                if (s >= 0) {
                    const int c = s / DECIM, r = s - DECIM * c;
                    hist[r * A_HIST_COLS + c] = make_float2(raw[k].x, raw[k].y);
                    hist[(r + 1) * A_HIST_COLS + c] = make_float2(raw[k].z, raw[k].w);
                }
            }
            if (lane < DECIM) hist[lane * A_HIST_COLS + HL] = make_float2(0.f, 0.f);
        }
        if (lane == 0) L.texp[wave] = Et;
        __builtin_amdgcn_wave_barrier();              // LDS operations of one wave complete in order
        if (lane == 0) seq_post(&scat_seq[wave], ti + 1);
        // ---- prefetch this wave's next tile as soon as the registers are free: the loads are in flight for the whole iteration
        if (!LO && ti + NW < NT && !(F4_ABL & 8)) load_tile(ti + NW);        // (LO: behind the filter, whose second set of accumulators needs the registers)
        // ---- the previous tile's newest 288 samples are this tile's history (tile 0: the call's, put there in front of the barrier)
        if (ti > 0) seq_wait(&scat_seq[pw], ti);
        // the first two column blocks' windows begin in the previous tile (K-steps 0 .. 8 of block 0, 0 .. 2 of block 1): where its scale is
        // another one, their sums change unit behind those steps
        const int Ep = __builtin_amdgcn_readfirstlane(L.texp[ti == 0 ? NW : pw]);
        const bool rescale = Ep != Et;
        const float rt = bfp_ratio(Ep, Et);
        const float rt2 = (blk == 1) ? rt : 1.0f, rt8 = (blk == 0) ? rt : 1.0f;

        // ---- the filter: D[i][n] += A[i][k] B[k][n] over the 480 samples of the window, three (four) f16 terms; the column sums beside it
        v4f ahh = (v4f){0.f, 0.f, 0.f, 0.f}, ahl = ahh, alh = ahh, asum = ahh;
        v4f bhh = ahh, bhl = ahh, blh = ahh;          // LO: the same against the taps' imaginary parts

        if (!(F4_ABL & 4)) {
#pragma unroll
            for (int j = 0; j < KSTEPS; j++) {
                const u32x4 bh = *reinterpret_cast<const u32x4 *>(bB + 64 * j), bl = *reinterpret_cast<const u32x4 *>(bB + 64 * j + 2 * PLB);
                const u32x2 a0 = *reinterpret_cast<const u32x2 *>(aB + 64 * j), a1 = *reinterpret_cast<const u32x2 *>(aB + 64 * j + 8);
                const u32x2 l0 = *reinterpret_cast<const u32x2 *>(aB + 64 * j + 2 * TA_N), l1 = *reinterpret_cast<const u32x2 *>(aB + 64 * j + 2 * TA_N + 8);
                const v8h Bh = __builtin_bit_cast(v8h, bh), Bl = __builtin_bit_cast(v8h, bl);
                const v8h Ah = __builtin_bit_cast(v8h, (u32x4){a0.x, a0.y, a1.x, a1.y}), Al = __builtin_bit_cast(v8h, (u32x4){l0.x, l0.y, l1.x, l1.y});
                ahh = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bh, ahh, 0, 0, 0);
                ahl = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, Bl, ahl, 0, 0, 0);
                alh = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, Bh, alh, 0, 0, 0);
#if FMX_F4_TERMS >= 4
                ahl = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, Bl, ahl, 0, 0, 0);
#endif
                if (dcr && j >= KSUM0 && !(F4_ABL & 2)) {
                    const u32x2 o0 = *reinterpret_cast<const u32x2 *>(aB + 64 * j + 4 * TA_N), o1 = *reinterpret_cast<const u32x2 *>(aB + 64 * j + 4 * TA_N + 8);
                    const v8h Ao = __builtin_bit_cast(v8h, (u32x4){o0.x, o0.y, o1.x, o1.y});
                    asum = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ao, Bh, asum, 0, 0, 0);
                    asum = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ao, Bl, asum, 0, 0, 0);
                }
                if constexpr (LO) {
                    if (mix) {
                        const u32x2 i0 = *reinterpret_cast<const u32x2 *>(aB + 64 * j + 6 * TA_N), i1 = *reinterpret_cast<const u32x2 *>(aB + 64 * j + 6 * TA_N + 8);
                        const u32x2 m0 = *reinterpret_cast<const u32x2 *>(aB + 64 * j + 8 * TA_N), m1 = *reinterpret_cast<const u32x2 *>(aB + 64 * j + 8 * TA_N + 8);
                        const v8h Ih = __builtin_bit_cast(v8h, (u32x4){i0.x, i0.y, i1.x, i1.y}), Il = __builtin_bit_cast(v8h, (u32x4){m0.x, m0.y, m1.x, m1.y});
                        bhh = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ih, Bh, bhh, 0, 0, 0);
                        bhl = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ih, Bl, bhl, 0, 0, 0);
                        blh = __builtin_amdgcn_mfma_f32_16x16x32_f16(Il, Bh, blh, 0, 0, 0);
                    }
                }
                if (rescale && j == 2) { ahh *= rt2; ahl *= rt2; alh *= rt2; if (LO) { bhh *= rt2; bhl *= rt2; blh *= rt2; } }
                if (rescale && j == 8) { ahh *= rt8; ahl *= rt8; alh *= rt8; if (LO) { bhh *= rt8; bhl *= rt8; blh *= rt8; } }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) seq_post(&fir_seq[wave], ti + 1);           // this slot's predecessor may take its owner's next tile
        if (LO && ti + NW < NT && !(F4_ABL & 8)) load_tile(ti + NW);
        // LO: the oscillator's values at the newest samples of the lane's four outputs (requested behind the filter, whose registers they would crowd;
        // wanted behind the RF DC recurrence)
        float2 Wq[4];
        if constexpr (LO) {
            if (mix) {
                int ph = ph_q0;
#pragma unroll
                for (int v = 0; v < 4; v++) { Wq[v] = T.lo_table[ph]; ph -= ph_step12; ph += ph < 0 ? Rr : 0; }
                ph_q0 -= ph_stepT; ph_q0 += ph_q0 < 0 ? Rr : 0;
            }
        }
        float a[4];
        const float osc = bfp_out(Et);
#pragma unroll
        for (int v = 0; v < 4; v++) a[v] = (ahh[v] + (ahl[v] + alh[v])) * osc;
        if constexpr (LO) {
            if (mix) {
                // complex taps: the real part's lane has Gr * xr and Gi * xr, its partner Gr * xi and Gi * xi:  S.re = Gr xr - Gi xi,  S.im = Gr xi + Gi xr
#pragma unroll
                for (int v = 0; v < 4; v++) a[v] = fmaf(sgn, dpp_swap1((bhh[v] + (bhl[v] + blh[v])) * osc), a[v]);
            }
        }

        // ---- RF DC removal (fm-processor.cpp:423-446) behind the filter, as front_kernel does it for channels without an LO -- here in the
        //      accumulator layout: the lane has the sums of its four columns (of its component), the exclusive prefix over the tile's 128 columns
        //      comes from two cross-row exchanges and a three-step row scan, the state in front of the tile from the previous tile's mailbox slot
        float c_out_r = dc0r, c_out_i = dc0i;
        float rr[4] = {0.f, 0.f, 0.f, 0.f};           // RfDC in front of the lane's four columns
        if (dcr && !(F4_ABL & 2)) {
            const float isc = bfp_inv(Et) * ibal;                 // (the sums of the raw samples: RfDC runs in front of the balance)
            const float S0 = asum[0] * isc, S1 = asum[1] * isc, S2 = asum[2] * isc, S3 = asum[3] * isc;
            const float e2 = S0 + S1, e3 = e2 + S2, tot = e3 + S3;
            const float p16 = bperm(lane ^ 16, tot), x1 = tot + p16;
            const float p32 = bperm(lane ^ 32, x1), bt = x1 + p32;                                  // the block's 16 columns
            const float exk = ((kg & 1) ? p16 : 0.f) + ((kg & 2) ? p32 : 0.f);                      // the block's columns in front of the lane's
            float inc = bt;
            inc += dpp_row_shr<2>(inc); inc += dpp_row_shr<4>(inc); inc += dpp_row_shr<8>(inc);   // blocks 0 .. blk (same component)
            const float pre0 = (inc - bt) + exk;                                                    // columns 0 .. c0col - 1
            // the tile's weighted sum (what is left of each column's samples at the tile's end), both components to every lane
            float ws = (S0 * wv[0] + S1 * wv[1]) + (S2 * wv[2] + S3 * wv[3]);
            ws += dpp_row_shr<2>(ws); ws += dpp_row_shr<4>(ws); ws += dpp_row_shr<8>(ws);
            const float w16 = bperm(lane ^ 16, ws), ws2 = ws + w16;
            const float w32 = bperm(lane ^ 32, ws2), ws3 = ws2 + w32;
            const float W_re = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ws3), 14));
            const float W_im = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ws3), 15));
            // the state in front of the tile, and what the lanes at the tile's head need of the previous tile's boundaries
            const float2 *mp = mb[(ti - 1) & 7];
            if (ti > 0) seq_wait(&carry_seq, ti);
            float pv[5];
#pragma unroll
            for (int k = 0; k < 5; k++) { const float2 e = mp[(mbe + k) > 13 ? 13 : mbe + k]; pv[k] = comp ? e.y : e.x; }
            float c0r = dc0r, c0i = dc0i;
            if (ti > 0) { const float2 cc = mp[13]; c0r = cc.x; c0i = cc.y; }
            c_out_r = dc_chain(c0r, u_tile, alpha * W_re); c_out_i = dc_chain(c0i, u_tile, alpha * W_im);
            const float c0 = comp ? c0i : c0r;
            {
                const float P0 = pre0, P1 = pre0 + S0, P2 = pre0 + e2, P3 = pre0 + e3;
                const float base = (float)(DECIM * c0col) * alpha;
                rr[0] = fmaf(alpha, P0, fmaf(-c0, base, c0));
                rr[1] = fmaf(alpha, P1, fmaf(-c0, base + 12.0f * alpha, c0));
                rr[2] = fmaf(alpha, P2, fmaf(-c0, base + 24.0f * alpha, c0));
                rr[3] = fmaf(alpha, P3, fmaf(-c0, base + 36.0f * alpha, c0));
            }
            // the next tile's slot: columns 115 .. 127 of this tile, then the state behind it
            float *mn = reinterpret_cast<float *>(mb[ti & 7]);
            if (blk == 7) {
#pragma unroll
                for (int v = 0; v < 4; v++) { const int e = 4 * kg + v - 3; if (e >= 0) mn[2 * e + comp] = rr[v]; }
            }
            if (lane == 0) mb[ti & 7][13] = make_float2(c_out_r, c_out_i);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) seq_post(&carry_seq, ti + 1);             // (every read of the previous slot is in front of this release)
            // RfDC at the five boundaries the lane's outputs interpolate between (columns c0col - 12 .. c0col - 8)
            float E[5];
#pragma unroll
            for (int v = 0; v < 4; v++) { const float f = bperm(srcA, rr[v]); E[v] = prevA ? pv[v] : f; }
            { const float f = bperm(srcB, rr[0]); E[4] = prevB ? pv[4] : f; }
            // what the FIR makes of the RfDC values the reference subtracts in front of it (limited to +-0.01, DCRlimit :429-442)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const float d = fmaf(dcw, E[v + 1] - E[v], E[v]);
                const float dl = __builtin_amdgcn_fmed3f(d, -0.01f, 0.01f);
                if (LO && mix) {
                    // (c att) Hlo, complex: own component D Hr -+ partner's D Hi (hlo_re carries the own component's balance; the partner's its own)
                    const float dp = dpp_swap1(dl * bal);
                    a[v] = a[v] - (dl * hlo_re + sgn * (dp * hlo_im));
                } else a[v] = fmaf(-hsum, dl, a[v]);
            }
        }
        // ---- the decimators' complex gain, the fm-rate ring: the lane of the real part stores the quad's first two outputs,
        //      the lane of the imaginary part the other two
        float z[4];
#pragma unroll
        for (int v = 0; v < 4; v++) {
            if (LO && mix) {
                // z = (cg LO [n]) S: the output gain turned by the oscillator's value at the output's newest sample
                const float wr = cg_re * Wq[v].x - cg_im * Wq[v].y, wi = cg_re * Wq[v].y + cg_im * Wq[v].x;
                z[v] = fmaf(dpp_swap1(a[v]), sgn * wi, a[v] * wr);
            } else z[v] = fmaf(dpp_swap1(a[v]), kpar, a[v] * kown);
        }
        const float k0 = comp ? z[2] : z[0], k1 = comp ? z[3] : z[1];       // what the lane keeps ...
        const float g0 = comp ? z[0] : z[2], g1 = comp ? z[1] : z[3];       // ... and what its partner stores
        const float r0 = dpp_swap1(g0), r1 = dpp_swap1(g1);
        const float4 o4 = comp ? make_float4(r0, k0, r1, k1) : make_float4(k0, r0, k1, r1);
        const int qc = qt + c0col + 2 * comp;         // the lane's first output column
        const int zi = (zr0 + qc) & G.ring_mask;
        if (F4_ABL & 16) {
            if (o4.x == 1.2345e33f) zring[0] = make_float2(o4.y, o4.z);       // (diagnostic: the results stay alive, nothing is stored)
        } else if ((zi & 1) == 0 && qc >= ja && qc + 1 < jb) {
            *reinterpret_cast<float4 *>(&zring[zi]) = o4;
        } else {
            if (qc >= ja && qc < jb) zring[(zr0 + qc) & G.ring_mask] = make_float2(o4.x, o4.y);
            if (qc + 1 >= ja && qc + 1 < jb) zring[(zr0 + qc + 1) & G.ring_mask] = make_float2(o4.z, o4.w);
        }
        // ---- last tile: the state the next call finds (front_kernel's format)
        if (ti == NT - 1) {
            if (lane == 0 && (dcr || dc_rst)) { st->dc_re = c_out_r; st->dc_im = c_out_i; }
            if (lane == 0) st->hist_fmt = mix ? 1 : 0;
            __builtin_amdgcn_wave_barrier();
            if (lane < 14) B.dcv_hist[(size_t)ch * DCV_SAVE + lane] = dcr ? mb[ti & 7][lane] : make_float2(dc0r, dc0i);
            if constexpr (LO) {
                if (mix) {
                    // A running oscillator: the next call finds the history PROCESSED, p = ((x - RfDC) att) LO [s] per sample, as front_kernel keeps it for such a
                    // channel (hist_fmt 1) -- whichever kernel takes the next call reads the same thing.  RfDC behind sample 12 c + r from the boundaries of
                    // columns c and c + 1 (it moves by 1e-6 of |x| within a column: linear to 1e-9); the samples are the raw ones this wave stored behind
                    // its scatter, converted in place.
                    float *hbf = reinterpret_cast<float *>(&L.hb[0]);
                    if (blk >= 6) {
#pragma unroll
                        for (int v = 0; v < 4; v++) { const int c = c0col + v; if (c >= 103) hbf[2 * (c - 103) + comp] = rr[v]; }
                    }
                    if (lane == 0) L.hb[25] = make_float2(c_out_r, c_out_i);
                    __builtin_amdgcn_wave_barrier();
                    __threadfence_block();            // (the raw history this wave stored behind its scatter is read back here, converted in place)
                    constexpr int HQ2 = (DECIM * A_HIST_COLS + 63) / 64;
#pragma unroll
                    for (int q = 0; q < HQ2; q++) {
                        const int i = lane + 64 * q;
                        const int r = i / A_HIST_COLS, c = i - r * A_HIST_COLS;          // history column c = the tile's column 104 + c
                        if (i < DECIM * A_HIST_COLS && c != HL) {
                            const float2 x = hist[i];
                            const float2 b0 = L.hb[c + 1], b1 = L.hb[c + 2];
                            const float fr = (float)(r + 1) * (1.0f / 12.0f);
                            const float cr_ = dcr ? __builtin_amdgcn_fmed3f(fmaf(fr, b1.x - b0.x, b0.x), -0.01f, 0.01f) : 0.f;
                            const float ci_ = dcr ? __builtin_amdgcn_fmed3f(fmaf(fr, b1.y - b0.y, b0.y), -0.01f, 0.01f) : 0.f;
                            const float xr = (x.x - cr_) * P.att_l, xi = (x.y - ci_) * P.att_r;
                            const float2 w = T.lo_table[lo_idx((long long)G.n - HS + DECIM * c + r, lo_phase0)];
                            hist[i] = make_float2(xr * w.x - xi * w.y, xr * w.y + xi * w.x);
                        }
                    }
                    if (lane == 0) st->lo_phase = lo_idx((long long)G.n - 1, lo_phase0);       // LOPhase behind the call's last sample: (P0 - n lo) mod R
                }
            }
        }
    }
}

}  // namespace F4_NS

// The calls front4_kernel takes (launch_front asks): whole tiles, on a column boundary, float32 samples 16-byte aligned.  The per-channel
// conditions -- no LO anywhere, every tap set the long fold with its RfDC taken 12 columns back, one twin -- are the handle's (fmx_api.hip).
#if !F4_LO
int front4_tiles(const CallGeom &G, const void *iq) {
    if (G.iq_format < 0 || G.iq_format > 3 || G.twins != 1 || G.pre_processed || G.parts > 1) return 0;
    const int bps = (G.iq_format == 0) ? 8 : (G.iq_format == 3 ? 4 : 2);
    // (a lane's sample PAIR is one 16 / 4 / 8 byte load)
    if ((G.g0 % DECIM) != 0 || (G.stream_stride & 1) != 0 || (reinterpret_cast<uintptr_t>(iq) & (uintptr_t)(2 * bps - 1)) != 0) return 0;
    return (int)(G.n / f4::WSAMP);
}
#endif
void F4_FN(launch_front4)(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, const void *iq, int channels, hipStream_t s) {
    namespace f4 = F4_NS;
    const dim3 grid((channels + f4::CPW - 1) / f4::CPW);
    switch (G.iq_format) {
    case 1: hipLaunchKernelGGL((f4::front4_kernel<1, false>), grid, dim3(f4::NTHR), 0, s, T, B, G, iq); break;
    case 2: hipLaunchKernelGGL((f4::front4_kernel<2, false>), grid, dim3(f4::NTHR), 0, s, T, B, G, iq); break;
    case 3: hipLaunchKernelGGL((f4::front4_kernel<3, false>), grid, dim3(f4::NTHR), 0, s, T, B, G, iq); break;
    default:
        if (G.streams_private) hipLaunchKernelGGL((f4::front4_kernel<0, true>), grid, dim3(f4::NTHR), 0, s, T, B, G, iq);
        else hipLaunchKernelGGL((f4::front4_kernel<0, false>), grid, dim3(f4::NTHR), 0, s, T, B, G, iq);
        break;
    }
}

}  // namespace fmx
