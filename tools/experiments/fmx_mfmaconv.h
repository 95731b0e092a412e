// fmx_mfmaconv.h -- the PSS low-pass of a stage-B segment (stereo-separation.cpp:60-83: 295 taps on the complex s ring) on the MATRIX pipe,
// the recipe of fmx_front4.hip: samples and taps split into two f16 halves each (s * 2^9 = sh + sl, h * 2^14 = hh + hl; products of f16
// values are exact in the f32 accumulator, the remainders' roundings are 2^-22 of a product),
//     y[m0 + i] = sum_k A[i][k] B[k][n],   A[i][k] = h[i + 294 - k],   B[k][n] = win[m0 + k].comp,   n = 2 b + comp,
// 16 adjacent outputs i against the 310 entries of their block's window (K = 320: ten steps of v_mfma_f32_16x16x32_f16), eight blocks x (re, im)
// = the 16 columns; a wave takes three groups of 128 outputs of the segment's 1536.  The window sits in LDS as four linear f16 planes (hi re,
// hi im, lo re, lo im; 3712 bytes each: eight 16-byte bank slots apart mod 16, which with a block stride of two slots makes the operand reads
// conflict-free); the taps reversed in a table per half, in four copies one entry apart (an output's taps begin one entry behind its
// neighbour's, and an operand read wants 8-byte alignment).  Replaces the fast convolution (fmx_fftconv.h: two 2048-point transforms of packed
// f32 FMAs and six workgroup barriers per segment -- a third of the second kernel's VALU instructions, the kind that runs this GPU into its
// power limit) by 90 matrix instructions per wave, 32 + 4 LDS writes and ~60 plain VALU instructions per thread and two barriers.
// STATUS (round 5): an experiment, compiled out by default (SB_PSS_MFMA in fmx_stageb.hip) -- see the note there.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

namespace fmx {
namespace mconv {

constexpr int TAPS = 295;                      // PSS_TAPS
constexpr int OUT = 1536;                      // outputs per segment (FB_W)
constexpr int NPL = 1856;                      // entries per plane: the last block's window ends at 1408 + 112 + 320 = 1840
constexpr int PLB = NPL * 2;                   // 3712 bytes = 232 slots = 8 (mod 16)
constexpr int TABN = 336;                      // entries per table copy: v = 32 j + 8 kg + ((15 - i) & ~3) + e in [0, 332)
constexpr int TABB = 4 * TABN * 2;             // bytes of the four copies of one half (2688)
constexpr float SSC = 512.f, HSC = 16384.f;
constexpr float OSC2 = (1.0f / (512.f * 16384.f)) * (1.0f / (512.f * 16384.f));
static_assert((PLB / 16) % 16 == 8 && TABB % 16 == 0, "bank spread / vector copies");

// host: [half][copy c][v] = f16 half of h[309 - (v + c)] * 2^14 (zero outside the filter), as bit patterns
inline void make_tables(const float *h, uint16_t *out /*[2][4][TABN]*/) {
    for (int c = 0; c < 4; c++)
        for (int v = 0; v < TABN; v++) {
            const int m = 309 - (v + c);
            const float gs = (m >= 0 && m < TAPS) ? h[m] * HSC : 0.f;
            const _Float16 gh = (_Float16)gs, gl = (_Float16)(gs - (float)gh);
            memcpy(&out[(0 * 4 + c) * TABN + v], &gh, 2);
            memcpy(&out[(1 * 4 + c) * TABN + v], &gl, 2);
        }
}

#if defined(__HIPCC__)
typedef _Float16 h16;
typedef h16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 256 threads.  a[p] = window entry tid + 256 p (entries from OUT + 294 on: zeros).  planes: >= 4 PLB + TABB bytes of LDS, 16-byte aligned (the hi
// table goes behind the planes); lo_tab: TABB bytes of LDS; gtab: make_tables' output in global memory.  On return (behind the function's last
// barrier) er[m] = Re (y[m]) Im (y[m]) for m = 0 .. OUT - 1; er may overlay the planes.
__device__ __forceinline__ void pss_errors(int tid, const float2 (&a)[8], char *planes, char *lo_tab, const uint16_t *__restrict__ gtab, float *er) {
    h16 *pl = reinterpret_cast<h16 *>(planes);
#ifdef MCONV_ENTRY_BARRIER
    __syncthreads();
#endif
#ifndef MCONV_DRAIN
#define MCONV_DRAIN "s_nop 7\n\ts_nop 4"    /* wait states between the last matrix instruction and the first VALU read of the sums */
#endif
#ifndef MCONV_TAIL
#define MCONV_TAIL ""      /* (diagnostic builds: wait states behind a step's last matrix instruction, e.g. "\n\ts_nop 7") */
#endif
#ifndef MCONV_SKIP
#define MCONV_SKIP 0   /* diagnostic builds: bit 0 no plane writes, 1 no table copy, 2 no matrix phase, 3 no er stores */
#endif
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const int n = tid + 256 * p;
        if (!(MCONV_SKIP & 1) && (p < 7 || n < NPL)) {
            const float xr = a[p].x * SSC, xi = a[p].y * SSC;
            const h16 hr = (h16)xr, hi = (h16)xi;
            pl[n] = hr; pl[NPL + n] = hi;
            pl[2 * NPL + n] = (h16)(xr - (float)hr); pl[3 * NPL + n] = (h16)(xi - (float)hi);
        }
    }
    if (!(MCONV_SKIP & 2)) {
        const uint4 *src = reinterpret_cast<const uint4 *>(gtab);
        uint4 *dh = reinterpret_cast<uint4 *>(planes + 4 * PLB), *dl = reinterpret_cast<uint4 *>(lo_tab);
#ifdef MCONV_NO_LO_WRITE
        for (int i = tid; i < TABB / 16; i += 256) dh[i] = src[i];
        (void)dl;
#else
        for (int i = tid; i < 2 * TABB / 16; i += 256) { if (i < TABB / 16) dh[i] = src[i]; else dl[i - TABB / 16] = src[i]; }
#endif
    }
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6;
    const int kg = lane >> 4, n = lane & 15, blk = n >> 1, comp = n & 1;
    const int sft = 15 - n;                                             // (row i = lane & 15 of A: its taps begin 15 - i entries into the table)
    const int aoff = (sft & 3) * (TABN * 2) + 16 * kg + 2 * (sft & ~3);
    const char *const aH = planes + 4 * PLB + aoff, *const aL = lo_tab + aoff;
    float pr[3][4];
#pragma unroll
    for (int gi = 0; gi < 3; gi++) {
        __builtin_amdgcn_sched_barrier(0);                              // (one group after the other: three at once cost the kernel its register budget)
        const int g = wv + 4 * gi;
        const char *const bB = planes + comp * PLB + 256 * g + 32 * blk + 16 * kg;
        // The matrix instructions as inline assembly with the accumulator TIED (destination = addend).  As a builtin, under the second kernel's
        // register budget, the compiler re-bases an accumulator between two steps -- v_mfma ... v[18:21], .., .., v[20:23]: destination and addend
        // overlapping PARTLY, which an eight-pass matrix instruction does not survive (it reads the addend row by row while it writes the
        // destination row by row).  The sums came out wrong by what the timing made of it: channels with the same input came apart
        // (tools/diag/flake_hunt.py; tests/test_isa_hazards.py scans every kernel's listing for such operand pairs).  Inline assembly is opaque
        // to the hazard recognizer: the first step takes 0 as its addend (no VALU-written register feeds a matrix instruction), and the wait
        // states between the last step and the first VALU read of the sums are put in by hand (eleven for eight passes).
        v4f ahh, ahl, alh;
#pragma unroll
        for (int j = 0; j < ((MCONV_SKIP & 4) ? 0 : 10); j++) {
            const u32x4 bh = *reinterpret_cast<const u32x4 *>(bB + 64 * j), bl = *reinterpret_cast<const u32x4 *>(bB + 64 * j + 2 * PLB);
            const u32x2 a0 = *reinterpret_cast<const u32x2 *>(aH + 64 * j), a1 = *reinterpret_cast<const u32x2 *>(aH + 64 * j + 8);
            const u32x2 l0 = *reinterpret_cast<const u32x2 *>(aL + 64 * j), l1 = *reinterpret_cast<const u32x2 *>(aL + 64 * j + 8);
            const v8h Bh = __builtin_bit_cast(v8h, bh), Bl = __builtin_bit_cast(v8h, bl);
            const v8h Ah = __builtin_bit_cast(v8h, (u32x4){a0.x, a0.y, a1.x, a1.y}), Al = __builtin_bit_cast(v8h, (u32x4){l0.x, l0.y, l1.x, l1.y});
            if (j == 0) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(ahh) : "v"(Ah), "v"(Bh));
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(ahl) : "v"(Ah), "v"(Bl));
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" MCONV_TAIL : "=&v"(alh) : "v"(Al), "v"(Bh));
            } else {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(ahh) : "v"(Ah), "v"(Bh));
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(ahl) : "v"(Ah), "v"(Bl));
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" MCONV_TAIL : "+v"(alh) : "v"(Al), "v"(Bh));
            }
        }
        if (MCONV_SKIP & 4) { ahh = (v4f){0.f, 0.f, 0.f, 0.f}; ahl = ahh; alh = ahh; }
        else asm volatile(MCONV_DRAIN : "+v"(ahh), "+v"(ahl), "+v"(alh));
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const float y = ahh[v] + (ahl[v] + alh[v]);
            const float yo = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(y), 0xB1, 0xf, 0xf, false));      // the other component (quad_perm [1,0,3,2])
            pr[gi][v] = (y * yo) * OSC2;
        }
    }
    __syncthreads();                                                    // (er may lie on top of the planes)
    if (!comp && !(MCONV_SKIP & 8)) {
#pragma unroll
        for (int gi = 0; gi < 3; gi++)
            *reinterpret_cast<float4 *>(&er[128 * (wv + 4 * gi) + 16 * blk + 4 * kg]) = make_float4(pr[gi][0], pr[gi][1], pr[gi][2], pr[gi][3]);
    }
#ifdef MCONV_EXIT_BARRIER
    __syncthreads();
#endif
}
#endif

}  // namespace mconv
}  // namespace fmx
