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
// the tap alignment `off`, 5440 fm-rate samples applied by the consumer of the ring).
//
// One persistent 256-thread workgroup per channel streams the call; its four waves take the call's
// 1536-sample tiles (128 output columns of 12 samples) round-robin and NEVER meet at a workgroup barrier:
//   * a wave loads its tile with coalesced dwordx4 loads, scatters it into its private LDS image X[r][C],
//     removes DC / mixes (each lane owns 24 consecutive samples; wave scan of the affine DC maps), runs the
//     polyphase FIR for its own 128 outputs and stores them;
//   * the only things a tile needs from its predecessor -- the DC state at its first sample and the 24 newest
//     processed columns (the FIR history) -- travel through small LDS mailboxes with sequence counters
//     (decoupled look-back: the DC carry is published before the wave's own second pass).
#include "fmx_internal.h"
#include "fmx_front_dc.h"

namespace fmx {

constexpr int HL = A_HIST_COLS - 1;            // 24 history columns in front of a tile
constexpr int WCOLS = 128;                     // fresh columns (= outputs) per wave tile
constexpr int WSAMP = WCOLS * DECIM;           // 1536 input samples per wave tile
constexpr int XCOLS = HL + WCOLS;              // 152 columns in a wave's LDS image
constexpr int SPT = 2 * DECIM;                 // 24 samples per lane per tile (two adjacent columns)
constexpr int FCOLS = 8;                       // adjacent outputs per lane in the FIR phase
#ifndef FMX_NW
#define FMX_NW 4                               /* waves (= tile images) per workgroup */
#endif
constexpr int NW = FMX_NW, NTHR = 64 * NW;
#ifndef FMX_AUX_N
#define FMX_AUX_N LO_LDS_MAX
#endif
constexpr int AUX_N = FMX_AUX_N;               // entries of the LO period table / the RfDC ring (one LDS array, see the kernel)
static_assert(AUX_N <= LO_LDS_MAX && (AUX_N & (AUX_N - 1)) == 0, "power of two, at most what the host tabulates");
constexpr int DCV_N = AUX_N;                   // ring of RfDC column-boundary values: several tiles deep (the waves of a workgroup are never that far apart)
constexpr int RPQ = DECIM / 4;                 // polyphase rows per lane quarter in the FIR phase

// LDS image of a wave tile: X[r][C], r = sample index mod 12, C = column (0..23 history, 24..151 fresh).  The unit of
// storage is the float4 holding the column pair (C even, C+1); unit index = r*XRS + ((C%8)/2)*XS4 + C/8, i.e. for a
// fixed row and pair slot the 8-column groups are contiguous.  Both consumers read it with conflict-free
// ds_read_b128: the DC phase (lane l: pair slot l%4 of group 3 + l/4; XS4 = 4 mod 16 spreads the four slots over the
// sixteen 16-byte bank slots of the lane groups ds_read_b128 is serviced in) and the FIR phase (each of those
// 16-lane service groups reads 16 consecutive groups of ONE row, see fir_lane_map).
constexpr int XS4 = 20;                        // >= 19 groups, = 4 (mod 16)
constexpr int XRS = 4 * XS4 + 1;               // row stride in units (odd: even rows spread over the banks on scatter)
constexpr int XUNITS = DECIM * XRS;            // 972 float4 = 15552 B per wave
__device__ __forceinline__ int xunit(int r, int C) { return r * XRS + ((C & 7) >> 1) * XS4 + (C >> 3); }
__device__ __forceinline__ int xidx(int r, int C) { return 2 * xunit(r, C) + (C & 1); }   // float2 index


#ifndef FMX_EARLY_PREFETCH
#define FMX_EARLY_PREFETCH 1
#endif
#ifndef FMX_WAVES_ATTR
/* LDS allows two workgroups per CU = two waves per SIMD: say so, or the register allocator aims at four (128 VGPRs, spills) */
#define FMX_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
#ifndef FMX_WG_PER_CU
#define FMX_WG_PER_CU 2
#endif
#ifndef FMX_FIR_ROT
#define FMX_FIR_ROT 1   /* 0: the plain row loop (A/B builds) */
#endif
#ifndef FMX_FIR_PRIO
#define FMX_FIR_PRIO -1   /* diagnostic builds: s_setprio inside the FIR phase (FMX_REST_PRIO outside); -1: none */
#endif
#ifndef FMX_REST_PRIO
#define FMX_REST_PRIO 0
#endif
#ifndef FMX_ABL
#define FMX_ABL 0      /* diagnostic builds only (tools/ablate_front.sh): bit 0 no scatter, 1 no DC/mix pass, 2 no FIR */
#endif
#define FMX_TICK(k) do { if (dbg_on) { unsigned long long now_ = clock64(); dbg_acc[k] += now_ - dbg_t; dbg_t = now_; } } while (0)
/* (diagnostics: the cycles wave 0 spends waiting for another wave's sequence counter, by counter, into slot 7 and the stage-B slots 13-15 nobody uses here) */
#ifdef FMX_WAIT_TICKS   /* (a diagnostic build, tools/build_variant.sh: the three counters cost registers the kernel does not have) */
#define FMX_WAIT(k, call) do { if (dbg_on) { const unsigned long long w0_ = clock64(); call; dbg_wait[k] += clock64() - w0_; } else { call; } } while (0)
#else
#define FMX_WAIT(k, call) do { call; } while (0)
#endif


// FIR phase: the lane computes eight adjacent outputs (columns 8 cg .. 8 cg + 7 of the tile) over polyphase rows
// r0 .. r0+2.  For each row it holds the window W[0..31] = X[r][8 cg .. 8 cg + 31] (image columns; output column
// 24 + 8 cg + k uses W[24 + k - d]) in registers, the row's taps arrive by wave-uniform-per-quarter LDS reads, and
// every tap feeds eight packed FMAs (re, im): 1 LDS byte per 1.6 flop.
template <int ND>
__device__ __forceinline__ void fir_rows(const float4 *__restrict__ X4, int cg, int r0, const float4 *__restrict__ tp, v2f acc[FCOLS]) {
    constexpr int JMIN = (HL - (ND - 1)) / 8;                       // first 8-column group the taps reach
#pragma unroll 1
    for (int rr = 0; rr < RPQ; rr++) {
        float tw[(ND + 3) / 4 * 4];
#pragma unroll
        for (int d4 = 0; d4 < (ND + 3) / 4; d4++) {                  // same address for the 16 lanes of a quarter: broadcast
            const float4 v = tp[rr * (A_TAPS_ROW / 4) + d4];
            tw[4 * d4] = v.x; tw[4 * d4 + 1] = v.y; tw[4 * d4 + 2] = v.z; tw[4 * d4 + 3] = v.w;
        }
        v2f W[32];
#pragma unroll
        for (int j = 3; j >= JMIN; j--)                              // newest columns first: the taps d = 0.. use them first
#pragma unroll
            for (int kp = 0; kp < 4; kp++) {
                const float4 v = X4[(r0 + rr) * XRS + kp * XS4 + cg + j];
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

// The same sums with the LDS latency under the FMAs (25 tap columns only).  Tap d of a row reads window entries 24 - d .. 31 - d: taps 0 .. 8 touch
// only the window's newer half (entries 16 .. 31), taps 16 .. 24 only its older half -- and tap d is used at step d alone.  So while taps 0 .. 15
// run, the older half and the taps 16 .. 27 of the row arrive; while taps 16 .. 24 run, the NEXT row's newer half and taps 0 .. 15 arrive in the
// registers that have just gone dead.  Only the first row's first loads are waited for (a wave has the SIMD to itself for much of a tile -- two
// waves per SIMD --, so a stall of the FIR is a stall of the tile: 1600 of the FIR's 4000 cycles were such stalls).
__device__ __forceinline__ void fir_rows_rot(const float4 *__restrict__ X4, int cg, int r0, const float4 *__restrict__ tp, v2f acc[FCOLS]) {
    constexpr int ND = A_MAX_ND;
    static_assert(ND == 25 && HL == 24 && FCOLS == 8, "the halves of the window are worked out for this shape");
    v2f Wn[16], Wo[16];                  // window entries 16 .. 31 and 0 .. 15
    float tlo[16], thi[12];              // taps 0 .. 15 and 16 .. 27
    auto load_new = [&](int row) {
#pragma unroll
        for (int j = 3; j >= 2; j--)
#pragma unroll
            for (int kp = 0; kp < 4; kp++) {
                const float4 v = X4[row * XRS + kp * XS4 + cg + j];
                Wn[8 * (j - 2) + 2 * kp] = (v2f){v.x, v.y}; Wn[8 * (j - 2) + 2 * kp + 1] = (v2f){v.z, v.w};
            }
    };
    auto load_old = [&](int row) {
#pragma unroll
        for (int j = 1; j >= 0; j--)
#pragma unroll
            for (int kp = 0; kp < 4; kp++) {
                const float4 v = X4[row * XRS + kp * XS4 + cg + j];
                Wo[8 * j + 2 * kp] = (v2f){v.x, v.y}; Wo[8 * j + 2 * kp + 1] = (v2f){v.z, v.w};
            }
    };
    auto taps_lo = [&](int rr) {
#pragma unroll
        for (int d4 = 0; d4 < 4; d4++) { const float4 v = tp[rr * (A_TAPS_ROW / 4) + d4]; tlo[4 * d4] = v.x; tlo[4 * d4 + 1] = v.y; tlo[4 * d4 + 2] = v.z; tlo[4 * d4 + 3] = v.w; }
    };
    auto taps_hi = [&](int rr) {
#pragma unroll
        for (int d4 = 0; d4 < 3; d4++) { const float4 v = tp[rr * (A_TAPS_ROW / 4) + 4 + d4]; thi[4 * d4] = v.x; thi[4 * d4 + 1] = v.y; thi[4 * d4 + 2] = v.z; thi[4 * d4 + 3] = v.w; }
    };
    auto W = [&](int i) -> v2f { return i >= 16 ? Wn[i - 16] : Wo[i]; };
    taps_lo(0); load_new(r0);
#pragma unroll
    for (int rr = 0; rr < RPQ; rr++) {
        taps_hi(rr); load_old(r0 + rr);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < 16; d++) {
            const v2f w = (v2f){tlo[d], tlo[d]};
#pragma unroll
            for (int k = 0; k < FCOLS; k++) acc[k] = __builtin_elementwise_fma(w, W(HL + k - d), acc[k]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (rr + 1 < RPQ) { taps_lo(rr + 1); load_new(r0 + rr + 1); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 16; d < ND; d++) {
            const v2f w = (v2f){thi[d - 16], thi[d - 16]};
#pragma unroll
            for (int k = 0; k < FCOLS; k++) acc[k] = __builtin_elementwise_fma(w, W(HL + k - d), acc[k]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// FMT: fmx_iq_format of the input (include/fmx.h).  Raw integer samples are converted while they are loaded --
// (u8 - 127) / 128, s8 / 128, s16 / denominator, all exact as in the reference's device handlers
// (rtlsdr-handler.cpp:291, hackrf-handler.cpp:364, lime-handler.cpp:250) -- into the same register layout.
// NTL: the tile loads are nontemporal -- every sample of a stream that only this channel listens to is read exactly once, so it
// should not displace the tables and rings in L2 / MALL (1 % of the launch time at 4096 channels); shared streams keep the cache.
template <int FMT, bool NTL>
__global__ __launch_bounds__(NTHR, FMX_WG_PER_CU) FMX_WAVES_ATTR void front_kernel(DeviceTables T, DeviceBuffers B, CallGeom G,
                                                       const void *__restrict__ iq_raw) {
    constexpr int BPS = (FMT == 0) ? 8 : (FMT == 3 ? 4 : 2);          // bytes per complex sample
    __shared__ __attribute__((aligned(16))) float4 Xall[NW][XUNITS];      // one image per wave (15552 B each)
    __shared__ __attribute__((aligned(16))) float sT[A_TAPS_DEV];          // the channel's tap set Trd[r][d]
    // a channel either mixes with an LO (one period of it here, when it has a short one) or takes its RF DC removal behind the FIR
    // (RfDC in front of call-relative column q at [q & (DCV_N - 1)]): never both, one array
    __shared__ float2 aux[AUX_N];
    float2 *const sLO = aux, *const dcv = aux;
    __shared__ float carry[8][2];                                          // DC state after tile ti, slot = ti & 7
    __shared__ int carry_seq;                                              // tiles whose carry is published
    __shared__ int hist_seq[NW], free_seq[NW];                               // per wave image: history of tile (n-1) is in / tile (n-1) is done

    // (twins: G.twins workgroups per channel, twin tw computes the outputs of phase tw -- see CallGeom::twins)
    const int vc = blockIdx.x;
    // (a channel split in time, CallGeom::parts: workgroup `part` of the channel stores the outputs of tiles tA .. t_end - 1; a part behind the
    // first begins one tile early -- a warm-up tile whose outputs nobody stores: its DC-corrected / mixed columns are the history of tile tA,
    // its RfDC boundaries the ones tile tA's outputs look back to --, with the DC state the maps of front_pre_kernel give it)
    const int NP = (G.parts > 1 && G.twins == 1) ? G.parts : 1, part = NP > 1 ? (int)blockIdx.y : 0;
    const int TW = G.twins;
    const int ch = TW == 1 ? vc : vc / TW;
    const int tw = vc - ch * TW;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    float4 *X4 = Xall[wave];
    float2 *X2 = reinterpret_cast<float2 *>(X4);
    ChanParams P = B.params[ch];
    // (the stream pre_kernel and the overlap-add machine of fmx_ola.hip have made: one per channel, RF DC removal, IQ balance and LO mix done)
    const bool pp = G.pre_processed != 0;
    if (G.cont) P.actions &= ~ACT_DC_RESET;          // (the head of this call has been made by another launch: setDCRemove's reset with it)
    if (pp) { P.stream = ch; P.dc_remove = 0; P.lo_freq = 0; P.lo_period = 0; P.att_l = 1.0f; P.att_r = 1.0f; P.actions &= ~ACT_DC_RESET; }
    const FrontSet FS = T.front_sets[P.front_set + tw];
    const char *__restrict__ inb = reinterpret_cast<const char *>(iq_raw) + (size_t)P.stream * G.stream_stride * BPS;
    const float2 *__restrict__ in = reinterpret_cast<const float2 *>(inb);       // FMT == 0
    const float qs = G.iq_scale;
    ChanState *st = tw == 0 ? B.state + ch : B.state_tw + (size_t)(tw - 1) * G.channels + ch;
    float2 *hist = B.hist + (size_t)vc * DECIM * A_HIST_COLS;
    // the state in front of the call: the live one, or -- when the channel's last part may rewrite it before this part has started -- its snapshot
    FrontSnap sn;
    if (NP > 1) sn = B.fsnap[vc]; else { sn.lo_phase = st->lo_phase; sn.hist_fmt = st->hist_fmt; sn.dc_re = st->dc_re; sn.dc_im = st->dc_im; }
    const float2 *histR = NP > 1 ? B.hist_snap + (size_t)vc * DECIM * A_HIST_COLS : hist;
    const float2 *dcvR = (NP > 1 ? B.dcv_snap : B.dcv_hist) + (size_t)vc * DCV_SAVE;
    float2 *zring = B.zring + (size_t)ch * (G.ring_mask + 1);

    const int off = FS.off, nd = FS.nd;
    // Call-local 32-bit geometry: sample index s = global index - 12 qa, column index = global column - qa.
    const int64_t qa = G.g0 / 12;                     // column holding the first fresh sample
    const int r0 = (int)(G.g0 - qa * 12);             // the call's fresh samples are s in [g0, gend)
    const int g0 = r0, gend = r0 + (int)G.n;
    const int ja = (int)((G.g0 - off + 11) / 12 - qa);            // first output completed by this call
    const int jb = (int)((G.g0 + G.n - off + 11) / 12 - qa);      // one past the last
    const int qb = (gend - 1) / 12;                   // column holding the last fresh sample
    const int NT = qb / WCOLS + 1;                    // wave tiles in this call
    const int tA = part * G.part_tiles, t_first = part > 0 ? tA - 1 : 0;
    const int t_end = (part + 1 == NP) ? NT : (tA + G.part_tiles < NT ? tA + G.part_tiles : NT);
    const int zr0 = (int)(((qa + FS.zshift) * TW + tw) & (int64_t)G.ring_mask);      // ring position of this twin's output column qa

    for (int i = t; i < A_TAPS_DEV; i += NTHR) sT[i] = T.front_taps[(size_t)(P.front_set + tw) * A_TAPS_DEV + i];
    if (t == 0) { carry_seq = 0; for (int i = 0; i < NW; i++) { hist_seq[i] = 0; free_seq[i] = 0; } }
    // ---- history -> the image of tile 0 (wave 0): columns qa-24 .. qa-1 at C 0..23, partial column qa at C 24
    // The history a call finds is in the format the last call left (ChanState::hist_fmt): raw for a channel without an LO (RfDC and IQ
    // balance are applied behind the FIR), DC-corrected, balanced and mixed for a channel with one.  Where the format -- or the RfDC value
    // the raw format is read with -- changes between two calls, the history is converted on load, so that the filter memory holds what
    // the reference's holds (its memory always has the processed samples of their own time):
    //  * LO switched on (raw -> processed): RfDC of each column from the saved boundaries, then the balance, then R0;
    //  * LO switched off (processed -> raw): the sample the output-side correction (sum h) clamp (RfDC), the balance and R0 take back to
    //    the processed one, r = conj (R0) v / att + clamp (RfDC now) (RfDC moves by < 1e-6 over the history's 288 samples);
    //  * setDCRemove (raw, RfDC zeroed at this call, fm-processor.cpp:922-925): the memory holds samples corrected with the OLD value,
    //    the output-side correction will use the new one: r = raw - clamp (old RfDC of the column) + clamp (0).
    // In the last two cases the boundaries in front of the call's first column are those of the new value.
    // R0: an oscillator set back to 0 Hz keeps its phase (Oscillator::nextValue oscillator.cpp:49-58 goes on reading the table entry it
    // stopped at), so the reference multiplies every sample by that constant -- which commutes with the real-tap filters and rides with
    // the complex output gain here.
    // (round 6: a channel whose IQ balance is not 1 takes the per-sample pass as well -- RfDC, balance, mix in the reference's order in front of the
    // filter, :423-466 -- so that the filter sees the reference's own products x att.  `lo_on` reads "the channel's samples are processed in front of
    // the filter", `lo_real` "its oscillator runs")
    const bool lo_real = (P.lo_freq != 0) && (T.lo_table != nullptr);
    const bool lo_on = lo_real || P.att_l != 1.0f || P.att_r != 1.0f;
    const float2 R0 = (!pp && !lo_real && T.lo_table != nullptr && sn.lo_phase != 0) ? T.lo_table[sn.lo_phase] : make_float2(1.f, 0.f);
    const bool dc_rst0 = (P.actions & ACT_DC_RESET) != 0;
    const float2 R0h = (lo_real && sn.hist_fmt == 0 && sn.lo_phase != 0) ? T.lo_table[sn.lo_phase] : make_float2(1.f, 0.f);   // (what the raw history was read with)
    // processed history of a channel whose oscillator has just been set back to 0 Hz and whose balance keeps it on the per-sample pass: the constant the
    // oscillator stopped at rides with the output gain from here on, the history's samples carry the oscillator's own values already
    const bool hist_derot = !pp && lo_on && !lo_real && sn.hist_fmt == 1 && T.lo_table != nullptr && sn.lo_phase != 0;
    const bool hist_convert = lo_on && (sn.hist_fmt == 0) && (P.dc_remove != 0 || P.att_l != 1.0f || P.att_r != 1.0f || dc_rst0 || sn.lo_phase != 0);
    const bool hist_to_raw = !pp && !lo_on && (sn.hist_fmt == 1);
    const bool hist_rst = !lo_on && (sn.hist_fmt == 0) && dc_rst0;
    const float2 dc_now = (dc_rst0 || P.dc_remove == 0) ? make_float2(0.f, 0.f)
                                                        : make_float2(__builtin_amdgcn_fmed3f(sn.dc_re, -0.01f, 0.01f), __builtin_amdgcn_fmed3f(sn.dc_im, -0.01f, 0.01f));
    if (wave == 0 && part == 0) {
        for (int i = lane; i < DECIM * A_HIST_COLS; i += 64) {
            int r = i / A_HIST_COLS, c = i - r * A_HIST_COLS;
            float2 v = histR[i];
            if (c == HL && r >= r0) v = make_float2(0.f, 0.f);
            else if (hist_convert || hist_rst) {
                const int tb = c - HL + 13;
                const float2 d = dcvR[tb < 0 ? 0 : tb];
                v.x -= __builtin_amdgcn_fmed3f(d.x, -0.01f, 0.01f);
                v.y -= __builtin_amdgcn_fmed3f(d.y, -0.01f, 0.01f);
                if (hist_convert) { v.x *= P.att_l; v.y *= P.att_r; v = make_float2(v.x * R0h.x - v.y * R0h.y, v.x * R0h.y + v.y * R0h.x); }
            } else if (hist_to_raw) {
                v = make_float2(v.x * R0.x + v.y * R0.y, v.y * R0.x - v.x * R0.y);
                v.x = (P.att_l != 0.f ? v.x / P.att_l : 0.f) + dc_now.x;
                v.y = (P.att_r != 0.f ? v.y / P.att_r : 0.f) + dc_now.y;
            } else if (hist_derot) v = make_float2(v.x * R0.x + v.y * R0.y, v.y * R0.x - v.x * R0.y);
            X2[xidx(r, c)] = v;
        }
    }
    // RfDC in front of the 13 columns before this call's first column and of that column itself (ring slots -13 .. 0)
    if (t < 14 && !lo_on && part == 0) dcv[(t - 13) & (DCV_N - 1)] = (hist_to_raw || hist_rst) ? make_float2(dc_rst0 ? 0.f : sn.dc_re, dc_rst0 ? 0.f : sn.dc_im)
                                                                                              : dcvR[t];
    // per-channel state is read by every wave BEFORE the barrier (the wave that ends the call rewrites it)
    const int lo_phase0 = sn.lo_phase;
    const bool dc_rst = (P.actions & ACT_DC_RESET) != 0;          // setDCRemove zeroes RfDC (:922-925)
    const float dc0r = dc_rst ? 0.f : sn.dc_re, dc0i = dc_rst ? 0.f : sn.dc_im;
    // a later part: the DC state in front of its warm-up tile = the maps of the stream's tiles 0 .. t_first - 1 applied to the call's, one
    // after the other as the workgroup of one part walks them (the values are the same bit for bit)
    if (part > 0 && P.dc_remove != 0 && wave == 0) {
        const float4 *mp = B.dc_tiles + (size_t)P.stream * B.dc_pitch * 2 + (lo_on ? 1 : 0);
        float cr = dc0r, ci = dc0i;
        for (int b = 0; b < t_first; b += 64) {
            const float4 m = (b + lane < t_first) ? mp[2 * (b + lane)] : make_float4(0.f, 0.f, 0.f, 0.f);
            const int cnt = t_first - b < 64 ? t_first - b : 64;
            for (int k = 0; k < cnt; k++) {
                const float tu = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m.x), k));
                cr = dc_chain(cr, tu, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m.y), k)));
                ci = dc_chain(ci, tu, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m.z), k)));
            }
        }
        if (lane == 0) { carry[(t_first - 1) & 7][0] = cr; carry[(t_first - 1) & 7][1] = ci; carry_seq = t_first; }
    }
    // ---- LO mix table: LOPhase after sample i of the call is (P0 - (i+1) lo) mod R; when lo / R has a short period p
    //      (channels on a raster: 200 kHz at 2.304 MS/s gives p = 288) the p table entries the call will use sit in LDS,
    //      sLO[m] = T[(P0 - m lo) mod R] with m = (i + 1) mod p, instead of 24 scattered reads of the 18 MB table per lane
    //      and tile
    const int lo_per = (P.lo_freq != 0 && T.lo_table != nullptr && P.lo_period <= AUX_N) ? P.lo_period : 0;
    for (int m = t; m < lo_per; m += NTHR) {
        long long ph = ((long long)lo_phase0 - (long long)m * (long long)P.lo_freq) % (long long)G.input_rate;
        if (ph < 0) ph += G.input_rate;
        sLO[m] = T.lo_table[ph];
    }
    __syncthreads();                                  // the only workgroup barrier: tables and counters are set up

    const float cg_re = FS.gain_re * R0.x - FS.gain_im * R0.y, cg_im = FS.gain_re * R0.y + FS.gain_im * R0.x;     // complex output gain x R0
    const bool dcr = P.dc_remove != 0;
    const int lo = P.lo_freq;
    const bool mix = (lo != 0) && (T.lo_table != nullptr);
    const int R = G.input_rate;
    const float alpha = 1.0f / (float)R;              // rfDcAlpha fm-processor.cpp:379
    const float Lg = P.att_l, Rg = P.att_r;
    const bool touch = dcr || mix || Lg != 1.0f || Rg != 1.0f;
    // Channels without an LO keep their samples RAW in the image: the RfDC value moves by at most alpha |x| = 4e-7 |x| per
    // sample, so  sum_i h_i clamp (RfDC[n - i])  =  (sum h) clamp (RfDC at the taps' centre of mass)  to ~1e-7, and the
    // subtraction -- and the IQ balance, a gain per component -- happens on the 128 outputs of a tile instead of its 1536 inputs.
    // The DC pass then only SUMS the lane's samples (first order in alpha: the terms dropped are alpha^2 k^2 |x| < 5e-7 |RfDC| over
    // a tile; the decay between tiles stays exact), scans, and leaves the RfDC value at every column boundary in `dcv`.
    const bool fast = !mix && Lg == 1.0f && Rg == 1.0f;
    const bool fastdc = fast && dcr;
    const bool dc_phase = !fast ? touch : dcr;
    const float hsum = FS.hsum, dcw = FS.dc_w;
    const int dck = FS.dc_k;
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
    const DcK DK = dc_consts(alpha, lane);
    // history hand-off: the 24 newest columns of this image go straight into the next wave's image (144 float4 units)
    float4 *Xn = Xall[(wave + 1) % NW];
    int ho_src[3], ho_dst[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int i = lane + 64 * k, r = (i < DECIM * 12) ? i / 12 : 0, cp = (i < DECIM * 12) ? i - 12 * r : 0;
        ho_src[k] = xunit(r, WCOLS + 2 * cp); ho_dst[k] = xunit(r, 2 * cp);
    }

    // FIR lane map: ds_read_b128 is serviced in the 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32); each
    // group takes ONE row quarter and 16 consecutive column groups, so its 16 reads fall on 16 distinct bank slots.
    const int lg = lane & 31;
    const bool inA = (lg < 4) || (lg >= 12 && lg < 16) || (lg >= 20 && lg < 28);
    const int rq = (lane >> 5) * 2 + (inA ? 0 : 1);
    const int cg = inA ? (lg < 4 ? lg : (lg < 16 ? lg - 8 : lg - 12)) : (lg < 12 ? lg - 4 : (lg < 20 ? lg - 8 : lg - 16));
    const float4 *tp = reinterpret_cast<const float4 *>(sT + RPQ * rq * A_TAPS_ROW);

    // Coalesced tile load: lane l, step k -> sample pair l + 64 k of the tile; after the scatter each lane reads back
    // "its" two columns (24 consecutive samples in time).
    float4 raw[SPT / 2];
    int sc_idx[SPT / 2];                                          // float2 index of sample pair k's first sample
#pragma unroll
    for (int k = 0; k < SPT / 2; k++) {
        const int e = 2 * (lane + 64 * k);                        // sample index within the tile (even)
        const int c = e / 12, r = e - 12 * c;                     // r is even: the pair stays inside one column
        sc_idx[k] = xidx(r, HL + c);
    }
    auto load_tile = [&](int ti) {
        const int wbase = ti * WSAMP;                             // index of the tile's first sample
        if (aligned16 && wbase >= g0 && wbase + WSAMP <= gend) {
            if (FMT == 0 && NTL) {
                typedef float v4f_ __attribute__((ext_vector_type(4)));
                const v4f_ *p4 = reinterpret_cast<const v4f_ *>(in + (wbase - g0));
#pragma unroll
                for (int k = 0; k < SPT / 2; k++) { const v4f_ v = __builtin_nontemporal_load(p4 + lane + 64 * k); raw[k] = make_float4(v.x, v.y, v.z, v.w); }
            } else if (FMT == 0) {
                const float4 *p4 = reinterpret_cast<const float4 *>(in + (wbase - g0));
#pragma unroll
                for (int k = 0; k < SPT / 2; k++) raw[k] = p4[lane + 64 * k];
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
    const bool dbg_on = (B.dbg != nullptr) && (t == 0);
    unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#ifdef FMX_WAIT_TICKS
    unsigned long long dbg_wait[3] = {0, 0, 0};
#endif
    unsigned long long dbg_t = dbg_on ? clock64() : 0ull;
    if (t_first + wave < t_end) load_tile(t_first + wave);
    const int dc_unit = (lane & 3) * XS4 + 3 + (lane >> 2);     // this lane's column pair (24 + 2 l, 24 + 2 l + 1), row 0
    FMX_TICK(0);

    for (int ti = t_first + wave; ti < t_end; ti += NW) {
        const int qt = ti * WCOLS;                    // first column of the tile
        const int wbase = qt * 12;
        // ---- scatter the raw samples into the image
        {
            const bool allfresh = (wbase >= g0) && (wbase + WSAMP <= gend);
            if (allfresh && (FMX_ABL & 8)) {              // (diagnostic: what a conflict-free scatter would cost -- the image is garbage)
#pragma unroll
                for (int k = 0; k < SPT / 2; k++) {
                    X2[lane + 128 * k] = make_float2(raw[k].x, raw[k].y);
                    X2[lane + 128 * k + 64] = make_float2(raw[k].z, raw[k].w);
                }
            } else
            if (allfresh && !(FMX_ABL & 1)) {
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
        // ---- prefetch this wave's next tile as soon as the registers are free: the loads are in flight for the whole
        //      iteration (DC pass, hand-off, FIR), so every wave keeps 12 KB of HBM reads outstanding all the time
        const bool more = (ti + NW < t_end);
#if FMX_EARLY_PREFETCH
        if (more) load_tile(ti + NW);
#endif
        FMX_TICK(1);
        const int q = qt + 2 * lane;                  // this lane's first column
        const int base = q * 12;
        // fresh samples of this lane are rows [first, lastp1) of its 24
        int first = (base >= g0) ? 0 : ((g0 - base) < SPT ? (g0 - base) : SPT);
        int lastp1 = (base + SPT <= gend) ? SPT : ((gend - base) > 0 ? (gend - base) : 0);
        if (lastp1 < first) lastp1 = first;
        const bool wave_full = __all(first == 0 && lastp1 == SPT);
        float c_out_r = 0.f, c_out_i = 0.f;           // DC state after this tile

        if (dc_phase && !(FMX_ABL & 2)) {
            v2f x[SPT];
#pragma unroll
            for (int r = 0; r < DECIM; r++) {
                const float4 v = X4[dc_unit + r * XRS];
                x[r] = (v2f){v.x, v.y}; x[r + DECIM] = (v2f){v.z, v.w};
            }
            // ---- RF DC removal (fm-processor.cpp:423-446): per-lane run, wave scan of the affine maps, the carry of
            //      the previous tile from the mailbox, then the reference's own f32 recurrence
            //      RfDC = (x - RfDC)*alpha + RfDC from the scanned prefix.
            if (dcr) {
                const v2f al = (v2f){alpha, alpha};
                const DcMap DM = dc_tile_map(x, first, lastp1, wave_full, fast, DK, lane);
                const Aff pre = DM.pre;
                const float tu = DM.tu, tar = DM.tar, tai = DM.tai;
                const v2f sA = DM.sA;
                // carry in: the DC state at the tile's first sample
                float c0 = dc0r, c1 = dc0i;
                if (ti > 0) {
                    FMX_WAIT(0, seq_wait(&carry_seq, ti));
                    c0 = carry[(ti - 1) & 7][0]; c1 = carry[(ti - 1) & 7][1];
                }
                c_out_r = dc_chain(c0, tu, tar); c_out_i = dc_chain(c1, tu, tai);
                if (lane == 0) { carry[ti & 7][0] = c_out_r; carry[ti & 7][1] = c_out_i; }
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) seq_post(&carry_seq, ti + 1);              // look-back hand-off, before this wave's pass 2
                v2f rr = (v2f){c0 - c0 * pre.u + pre.ar, c1 - c1 * pre.u + pre.ai};
                if (fast) {
                    // RfDC in front of the lane's two columns (= behind the sample in front of each) into the ring; the boundary in
                    // front of column 0 of a call that starts inside it belongs to the previous call (kept from dcv_hist)
                    v2f b1;
                    if (wave_full) b1 = __builtin_elementwise_fma(al, sA, __builtin_elementwise_fma((v2f){-12.0f * alpha, -12.0f * alpha}, rr, rr));
                    else {
                        b1 = rr;
#pragma unroll
                        for (int k = 0; k < DECIM; k++) if (k >= first && k < lastp1) b1 = __builtin_elementwise_fma(x[k] - b1, al, b1);
                    }
                    if (q * 12 >= g0) dcv[q & (DCV_N - 1)] = make_float2(rr.x, rr.y);
                    dcv[(q + 1) & (DCV_N - 1)] = make_float2(b1.x, b1.y);
                } else
                if (wave_full) {
#pragma unroll
                    for (int k = 0; k < SPT; k++) {
                        rr = __builtin_elementwise_fma(x[k] - rr, al, rr);
                        x[k] -= (v2f){__builtin_amdgcn_fmed3f(rr.x, -0.01f, 0.01f), __builtin_amdgcn_fmed3f(rr.y, -0.01f, 0.01f)};   // DCRlimit :429-442
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < SPT; k++) {
                        if (k >= first && k < lastp1) {
                            rr = __builtin_elementwise_fma(x[k] - rr, al, rr);
                            x[k] -= (v2f){__builtin_amdgcn_fmed3f(rr.x, -0.01f, 0.01f), __builtin_amdgcn_fmed3f(rr.y, -0.01f, 0.01f)};
                        }
                    }
                }
            }
            // ---- IQ balance + LO mix (fm-processor.cpp:462-466, oscillator.cpp:49-58); without an LO the balance is applied to the
            //      outputs and the image stays as it is
            if (!fast) {
            if (Lg != 1.0f || Rg != 1.0f) {
#pragma unroll
                for (int k = 0; k < SPT; k++) if (k >= first && k < lastp1) { x[k].x *= Lg; x[k].y *= Rg; }
            }
            if (mix && lastp1 > first) {
                // LOPhase after sample i (0-based within the call) = (P0 - (i+1)*lo) mod R
                if (lo_per > 0) {
                    int m = (int)((unsigned)(base + first - g0 + 1) % (unsigned)lo_per);
#pragma unroll
                    for (int k = 0; k < SPT; k++) {
                        if (k >= first && k < lastp1) {
                            const float2 w = sLO[m];
                            const v2f v = x[k];
                            x[k] = (v2f){v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x};
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
                            const v2f v = x[k];
                            x[k] = (v2f){v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x};
                            ph -= lo;
                            if (ph < 0) ph += R; else if (ph >= R) ph -= R;
                        }
                    }
                }
            }
            // ---- back to LDS in place (entries that are not fresh pass through unchanged)
#pragma unroll
            for (int r = 0; r < DECIM; r++)
                X4[dc_unit + r * XRS] = make_float4(x[r].x, x[r].y, x[r + DECIM].x, x[r + DECIM].y);
            }
            __builtin_amdgcn_wave_barrier();
        }
        FMX_TICK(2);
        // ---- hand the 24 newest processed columns to the next tile: written straight into the next wave's image, once
        //      that wave is done with its previous tile (ti - 3), whose history / partial sums live there
        if (ti + 1 < t_end) {
            const int nw = (wave + 1) % NW;
            if (ti + 1 - NW >= t_first) FMX_WAIT(1, seq_wait(&free_seq[nw], ti + 2 - NW));
#pragma unroll
            for (int k = 0; k < 3; k++)
                if (k < 2 || lane < DECIM * 12 - 128) Xn[ho_dst[k]] = X4[ho_src[k]];
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) seq_post(&hist_seq[nw], ti + 1);
        }
        // ---- and wait for the previous tile's (tile 0 got the call's history from HBM)
        if (ti > t_first) FMX_WAIT(2, seq_wait(&hist_seq[wave], ti));
        FMX_TICK(3);
#if !FMX_EARLY_PREFETCH
        if (more) load_tile(ti + NW);                 // (A/B build) prefetch only in front of the FIR
#endif

#if FMX_FIR_PRIO >= 0
        __builtin_amdgcn_s_setprio(FMX_FIR_PRIO);      // (A/B builds: the FIR's packed FMAs against the other wave's loads, LDS traffic and stores)
#endif
        // ---- polyphase FIR  out[j] = sum_d sum_r Trd[r][d] * X[r][C_j - d]:  lane quarter rq sums rows 3 rq .. 3 rq + 2
        //      for eight adjacent outputs per lane; the four partial sums meet in LDS (on top of the image, which is
        //      dead by then)
        {
            v2f acc[FCOLS];
#pragma unroll
            for (int k = 0; k < FCOLS; k++) acc[k] = (v2f){0.f, 0.f};
            if (FMX_ABL & 4) { acc[0] = (v2f){(float)cg, 1.f}; }
            else if (nd <= 4) fir_rows<4>(X4, cg, RPQ * rq, tp, acc);
#if FMX_FIR_ROT
            else fir_rows_rot(X4, cg, RPQ * rq, tp, acc);
#else
            else fir_rows<A_MAX_ND>(X4, cg, RPQ * rq, tp, acc);
#endif
            __builtin_amdgcn_wave_barrier();
            // save the columns the call-end history needs before the image is overwritten (last tile only: below)
            if (ti == NT - 1) {
                // ---- last tile: save history for the next call (columns qn-24 .. qn of the call, from this image)
                const int qn = gend / 12;                 // column of the next call's first sample
                const int cbase = qn - qt;                // image column of history slot 0 (= column qn-24)
                for (int i = lane; i < DECIM * A_HIST_COLS; i += 64) {
                    int r = i / A_HIST_COLS, c = i - r * A_HIST_COLS;
                    int lc = cbase + c;
                    float2 v = make_float2(0.f, 0.f);
                    if (lc >= 0 && lc < XCOLS) v = X2[xidx(r, lc)];
                    hist[i] = v;
                }
                if (lane == 0 && (dcr || dc_rst)) { st->dc_re = dcr ? c_out_r : dc0r; st->dc_im = dcr ? c_out_i : dc0i; }
                if (lane == 0 && !pp) st->hist_fmt = !fast ? 1 : 0;
                if (fast) {
                    // RfDC in front of the 13 columns before the next call's first column qn and of qn itself (the state behind the
                    // call when the call ends on a column boundary; zero history when DC removal is off)
                    if ((gend % 12) == 0 && lane == 0) dcv[qn & (DCV_N - 1)] = make_float2(dcr ? c_out_r : dc0r, dcr ? c_out_i : dc0i);
                    __builtin_amdgcn_wave_barrier();
                    if (lane < 14) B.dcv_hist[(size_t)vc * DCV_SAVE + lane] = dcr ? dcv[(qn - 13 + lane) & (DCV_N - 1)] : make_float2(dc0r, dc0i);
                }
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int k = 0; k < FCOLS / 2; k++)
                X4[(rq * 16 + cg) * 4 + ((k + (cg >> 1)) & 3)] = make_float4(acc[2 * k].x, acc[2 * k].y, acc[2 * k + 1].x, acc[2 * k + 1].y);
        }
        __builtin_amdgcn_wave_barrier();
#if FMX_FIR_PRIO >= 0
        __builtin_amdgcn_s_setprio(FMX_REST_PRIO);
#endif
        FMX_TICK(4);
        {
            // outputs 2 l, 2 l + 1 = pair (l & 3) of column group l >> 2
            const int fg = lane >> 2, pr = ((lane & 3) + (fg >> 1)) & 3;
            const float4 s0 = X4[(0 * 16 + fg) * 4 + pr], s1 = X4[(1 * 16 + fg) * 4 + pr];
            const float4 s2 = X4[(2 * 16 + fg) * 4 + pr], s3 = X4[(3 * 16 + fg) * 4 + pr];
            float2 aA = make_float2((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y));
            float2 aB = make_float2((s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
            if (fastdc) {
                // what the FIR makes of the RfDC values the reference subtracts in front of it (limited to +-0.01, DCRlimit :429-442)
                const float2 e0 = dcv[(q - dck) & (DCV_N - 1)], e1 = dcv[(q - dck + 1) & (DCV_N - 1)], e2 = dcv[(q - dck + 2) & (DCV_N - 1)];
                const float dAr = fmaf(dcw, e1.x - e0.x, e0.x), dAi = fmaf(dcw, e1.y - e0.y, e0.y);
                const float dBr = fmaf(dcw, e2.x - e1.x, e1.x), dBi = fmaf(dcw, e2.y - e1.y, e1.y);
                aA.x = fmaf(-hsum, __builtin_amdgcn_fmed3f(dAr, -0.01f, 0.01f), aA.x); aA.y = fmaf(-hsum, __builtin_amdgcn_fmed3f(dAi, -0.01f, 0.01f), aA.y);
                aB.x = fmaf(-hsum, __builtin_amdgcn_fmed3f(dBr, -0.01f, 0.01f), aB.x); aB.y = fmaf(-hsum, __builtin_amdgcn_fmed3f(dBi, -0.01f, 0.01f), aB.y);
            }
            const float2 zA = make_float2(aA.x * cg_re - aA.y * cg_im, aA.x * cg_im + aA.y * cg_re);
            const float2 zB = make_float2(aB.x * cg_re - aB.y * cg_im, aB.x * cg_im + aB.y * cg_re);
            const int zi = (zr0 + q) & G.ring_mask;
            if (ti < tA) { /* the warm-up tile of a later part: its outputs are the previous part's */ }
            else if (TW == 1 && (zi & 1) == 0 && q >= ja && q + 1 < jb) {
                // the lane's two outputs are neighbours in the ring and start on a 16-byte boundary (the ring's size is even): one store,
                // 1 KB contiguous per wave instead of two interleaved 8-byte streams
#ifdef FMX_ZNT
                { typedef float v4f_ __attribute__((ext_vector_type(4))); __builtin_nontemporal_store((v4f_){zA.x, zA.y, zB.x, zB.y}, reinterpret_cast<v4f_ *>(&zring[zi])); }
#else
                *reinterpret_cast<float4 *>(&zring[zi]) = make_float4(zA.x, zA.y, zB.x, zB.y);
#endif
            } else {
                if (q >= ja && q < jb) zring[(zr0 + q * TW) & G.ring_mask] = zA;
                if (q + 1 >= ja && q + 1 < jb) zring[(zr0 + (q + 1) * TW) & G.ring_mask] = zB;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) seq_post(&free_seq[wave], ti + 1);          // this image may receive the history of tile ti + 3
        FMX_TICK(5);
    }
    FMX_TICK(6);
    if (dbg_on && tw == 0 && part == 0) {
        for (int k = 0; k < 8; k++) B.dbg[(size_t)ch * DBG_SLOTS + k] += dbg_acc[k];
#ifdef FMX_WAIT_TICKS
        for (int k = 0; k < 3; k++) B.dbg[(size_t)ch * DBG_SLOTS + 40 + k] += dbg_wait[k];
#endif
    }
    if (t == 0 && lo != 0 && part + 1 == NP) {
        long long m = ((long long)G.n * (long long)lo) % (long long)R;
        int ph = (int)(((long long)lo_phase0 - m) % (long long)R);
        if (ph < 0) ph += R;
        st->lo_phase = ph;
    }
}

// In front of a launch that splits channels in time (CallGeom::parts > 1).  Workgroups 0 .. streams * n_tiles - 1: the RF DC recurrence over tile
// (x % n_tiles) of stream (x / n_tiles) as an affine map, computed exactly as the owner of the tile computes it in front_kernel (dc_tile_map:
// the same runs, the same scan), in both forms -- [0] channels without an LO (full tiles: sums), [1] channels with one (the recurrence).  The
// workgroups behind them: one per channel, the snapshot of what front_kernel reads of the channel's state, history and saved boundaries.
template <int FMT>
__global__ __launch_bounds__(64) void front_pre_kernel(DeviceBuffers B, CallGeom G, const void *__restrict__ iq_raw, int n_tiles, int streams) {
    constexpr int BPS = (FMT == 0) ? 8 : (FMT == 3 ? 4 : 2);
    const int lane = threadIdx.x;
    const int x = blockIdx.x;
    if (x >= streams * n_tiles) {
        const int vc = x - streams * n_tiles;
        const ChanState *st = B.state + vc;
        if (lane == 0) { FrontSnap sn; sn.lo_phase = st->lo_phase; sn.hist_fmt = st->hist_fmt; sn.dc_re = st->dc_re; sn.dc_im = st->dc_im; B.fsnap[vc] = sn; }
        for (int i = lane; i < DECIM * A_HIST_COLS; i += 64) B.hist_snap[(size_t)vc * DECIM * A_HIST_COLS + i] = B.hist[(size_t)vc * DECIM * A_HIST_COLS + i];
        if (lane < DCV_SAVE) B.dcv_snap[(size_t)vc * DCV_SAVE + lane] = B.dcv_hist[(size_t)vc * DCV_SAVE + lane];
        return;
    }
    const int sidx = x / n_tiles, ti = x - sidx * n_tiles;
    const char *__restrict__ inb = reinterpret_cast<const char *>(iq_raw) + (size_t)sidx * G.stream_stride * BPS;
    const float qs = G.iq_scale;
    const int64_t qa = G.g0 / 12;
    const int g0 = (int)(G.g0 - qa * 12), gend = g0 + (int)G.n;
    const float alpha = 1.0f / (float)G.input_rate;
    const DcK DK = dc_consts(alpha, lane);
    const int base = (ti * WCOLS + 2 * lane) * 12;
    int first = (base >= g0) ? 0 : ((g0 - base) < SPT ? (g0 - base) : SPT);
    int lastp1 = (base + SPT <= gend) ? SPT : ((gend - base) > 0 ? (gend - base) : 0);
    if (lastp1 < first) lastp1 = first;
    const bool wave_full = __all(first == 0 && lastp1 == SPT);
    v2f xs[SPT];
#pragma unroll
    for (int k = 0; k < SPT; k++) {
        float2 v = make_float2(0.f, 0.f);
        if (k >= first && k < lastp1) {
            const size_t i = (size_t)(base + k - g0);
            if (FMT == 0) v = reinterpret_cast<const float2 *>(inb)[i];
            else if (FMT == 1) { const uint8_t *p = reinterpret_cast<const uint8_t *>(inb) + 2 * i; v = make_float2((float)((int)p[0] - 127) * qs, (float)((int)p[1] - 127) * qs); }
            else if (FMT == 2) { const int8_t *p = reinterpret_cast<const int8_t *>(inb) + 2 * i; v = make_float2((float)p[0] * qs, (float)p[1] * qs); }
            else { const int16_t *p = reinterpret_cast<const int16_t *>(inb) + 2 * i; v = make_float2((float)p[0] * qs, (float)p[1] * qs); }
        }
        xs[k] = (v2f){v.x, v.y};
    }
    float4 *out = B.dc_tiles + ((size_t)sidx * B.dc_pitch + ti) * 2;
    const DcMap M1 = dc_tile_map(xs, first, lastp1, wave_full, false, DK, lane);
    if (wave_full) {
        const DcMap M0 = dc_tile_map(xs, first, lastp1, true, true, DK, lane);
        if (lane == 0) out[0] = make_float4(M0.tu, M0.tar, M0.tai, 0.f);
    } else if (lane == 0) out[0] = make_float4(M1.tu, M1.tar, M1.tai, 0.f);      // (a tile that is not full: one form for both)
    if (lane == 0) out[1] = make_float4(M1.tu, M1.tar, M1.tai, 0.f);
}

void launch_front(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, const void *iq,
                  int channels, hipStream_t s) {
    if (G.front4) {
        // batches that fill the chip: the call's whole tiles on the matrix pipe, what is left of it (less than a tile) behind them as a
        // call of its own -- the chain is invariant to how a stream is cut into calls
        const int k = front4_tiles(G, iq);
        if (k > 0) {
            CallGeom G3 = G; G3.n = (int64_t)k * WSAMP; G3.parts = 1;
            if (G.front4 == 2) launch_front4_lo(T, B, G3, iq, channels, s); else launch_front4(T, B, G3, iq, channels, s);
            if (G3.n == G.n) return;
            CallGeom G2 = G; G2.front4 = 0; G2.cont = 1; G2.parts = 1; G2.g0 = G.g0 + G3.n; G2.n = G.n - G3.n;
            const size_t bps = (G.iq_format == 0) ? 8 : (G.iq_format == 3 ? 4 : 2);
            launch_front(T, B, G2, reinterpret_cast<const char *>(iq) + (size_t)G3.n * bps, channels, s);
            return;
        }
    }
    const int parts = (G.parts > 1 && G.twins == 1) ? G.parts : 1;
    if (parts > 1) {
        // (the maps of the tiles in front of the last part's warm-up tile; a stream that pre_kernel has made carries no RF DC removal here)
        const int n_tiles = G.pre_processed ? 0 : (parts - 1) * G.part_tiles - 1;
        const int streams = n_tiles > 0 ? G.streams : 0;
        const dim3 pg(streams * n_tiles + channels);
        switch (G.iq_format) {
        case 1: hipLaunchKernelGGL((front_pre_kernel<1>), pg, dim3(64), 0, s, B, G, iq, n_tiles, streams); break;
        case 2: hipLaunchKernelGGL((front_pre_kernel<2>), pg, dim3(64), 0, s, B, G, iq, n_tiles, streams); break;
        case 3: hipLaunchKernelGGL((front_pre_kernel<3>), pg, dim3(64), 0, s, B, G, iq, n_tiles, streams); break;
        default: hipLaunchKernelGGL((front_pre_kernel<0>), pg, dim3(64), 0, s, B, G, iq, n_tiles, streams); break;
        }
    }
    const dim3 grid(channels * G.twins, parts);
    switch (G.iq_format) {
    case 1: hipLaunchKernelGGL((front_kernel<1, false>), grid, dim3(NTHR), 0, s, T, B, G, iq); break;
    case 2: hipLaunchKernelGGL((front_kernel<2, false>), grid, dim3(NTHR), 0, s, T, B, G, iq); break;
    case 3: hipLaunchKernelGGL((front_kernel<3, false>), grid, dim3(NTHR), 0, s, T, B, G, iq); break;
    default:
        if (G.streams_private) hipLaunchKernelGGL((front_kernel<0, true>), grid, dim3(NTHR), 0, s, T, B, G, iq);
        else hipLaunchKernelGGL((front_kernel<0, false>), grid, dim3(NTHR), 0, s, T, B, G, iq);
        break;
    }
}

}  // namespace fmx
