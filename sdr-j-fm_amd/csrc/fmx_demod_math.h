// fmx_demod_math.h -- the per-sample arithmetic of stage B shared by the chunked kernels (fmx_demod.hip) and the fused
// per-channel kernel (fmx_stageb.hip): phase wraps, the SinCos / atan2 look-up expressions and the limiter, each in
// exactly the f32 / f64 types the reference's C++ evaluates them in.  Only for translation units compiled with
// -ffp-contract=off.
#pragma once
#include "fmx_internal.h"

namespace fmx {

#define FMX_2PI 6.283185307179586476925286766559   /* 2 * M_PI as the double the reference uses */
#define FMX_PI_4 0.78539816339744830962

// exact fmod(x, 2*pi) for |x| < 8*pi by Sterbenz-exact subtractions (the generic ocml fmod is a
// long loop; every use here is within a few turns).  Falls back to fmod outside that range.
__device__ __attribute__((noinline)) double fmod_2pi_slow(double x) { return fmod(x, FMX_2PI); }
__device__ __forceinline__ double fmod_2pi(double x) {
    double ax = fabs(x);
    if (__builtin_expect(!(ax < 4 * FMX_2PI), 0)) return fmod_2pi_slow(x);   // also NaN/inf
    ax = (ax >= 2 * FMX_2PI) ? ax - 2 * FMX_2PI : ax;
    ax = (ax >= FMX_2PI) ? ax - FMX_2PI : ax;
    return copysign(ax, x);
}
// ---- PI_Constrain fm-constants.h:148-158; the in-range test is done in f32:
//      val < 2*M_PI (double)  <=>  val < 6.2831855f (the float just above 2*pi)
__device__ __forceinline__ float pi_constrain(float val) {
    if (val >= 0.f && val < 6.2831855f) return val;
    const double v = (double)val;
    if (v >= FMX_2PI) return (float)fmod_2pi(v);
    if (v > -FMX_2PI) return (float)(v + FMX_2PI);
    return (float)(FMX_2PI - fmod_2pi(-v));
}
// PI_Constrain for arguments known to lie in (-2*pi, 4*pi) (the pilot phase after one update):
// branch-free selects; v - 2*pi is exact for v in [2*pi, 4*pi) (Sterbenz), as is fmod there.
__device__ __forceinline__ float pi_constrain_near(float val) {
    // (the pilot phase is [0, 2pi) +- 5*|demod|*gain + omega: |5*demod*gain| < 0.01 since |demod| < 4.1)
    const double v = (double)val;
    const float hi = (float)(v - FMX_2PI), lo = (float)(v + FMX_2PI);
    return (val < 0.f) ? lo : ((val < 6.2831855f) ? val : hi);
}
// x / c for a CONSTANT c with rc = RN(1/c): q0 = x*rc; r = fma(-q0, c, x); q = fma(r, rc, q0).
// Markstein's correction step gives the correctly rounded IEEE quotient; verified exhaustively on the
// host for K_FM and the pilot omega over |x| in [2^-60, 2^60] (DESIGN.md "exact division by constants").
__device__ __forceinline__ float fdiv_const(float x, float c, float rc) {
    const float q0 = x * rc;
    const float r = __fmaf_rn(-q0, c, x);
    return __fmaf_rn(r, rc, q0);
}
// ---- SinCos sincos.cpp:63-97
__device__ __forceinline__ int sc_index(float phase, double C) {    // phase >= 0
    return ((int)((double)phase * C)) % SINCOS_N;
}
__device__ __forceinline__ float sc_sin(const float2 *__restrict__ tab, double C, float phase) {
    if (phase < 0) return -tab[sc_index(-phase, C)].y;
    return tab[sc_index(phase, C)].y;
}
__device__ __forceinline__ float sc_wrap(float phase) {
    while (phase < 0) phase = (float)((double)phase + FMX_2PI);
    return (float)fmod_2pi((double)phase);
}
__device__ __forceinline__ float2 sc_complex(const float2 *__restrict__ tab, double C, float phase) {
    return tab[sc_index(sc_wrap(phase), C)];
}
// table entry `idx` of SinCos (cos, sin) without the memory access (SinPoly in fmx_internal.h); bit-identical to tab[idx]
__device__ __forceinline__ float2 sc_entry(const DeviceTables &T, int idx) {
    if (!T.sp.ok) return T.sincos[idx];
    float sn, cs;
    sincos_poly(idx, T.sp.step, &sn, &cs);
    for (int k = 0; k < T.sp.ns; k++) sn = (idx == T.sp.s_idx[k]) ? T.sp.s_val[k] : sn;
    for (int k = 0; k < T.sp.nc; k++) cs = (idx == T.sp.c_idx[k]) ? T.sp.c_val[k] : cs;
    return make_float2(cs, sn);
}
// ---- compAtan::atan2 Xtan2.cpp:56-100.  Only the PPY table is stored; the other seven tables are
// the reference's own f32 expressions of it (Xtan2.cpp:31-38), evaluated here with the same ops.
__device__ __forceinline__ int at_idx(float size, float num, float den) {
    return (int)((double)(size * num / den) + 0.5);
}
// Branch-free: a wavefront's lanes fall into all eight octants (the FM phase step reaches +-2.4 rad), so the
// reference's if-tree would execute every arm one after the other.  Octant -> (table sign `size`, numerator /
// denominator, offset A, sign of the table value); every arm is  A + (+-table[idx])  with one f32 addition, which
// is the reference's own expression (a - t == a + (-t) etc. in IEEE arithmetic).
__device__ __forceinline__ float lut_atan2(const float *__restrict__ ppy, float y, float x) {
    const float St = (float)3.14159265358979323846, Sh = St * 0.5f;
    const bool special = isinf(x) || isinf(y) || isnan(x) || isnan(y) || x == 0.f;
    const bool xpos = x > 0.f, ypos = y >= 0.f;
    const bool swap = !(fabsf(x) >= fabsf(y));                 // the ..X arms: |y| > |x|
    const bool same = xpos == ypos;
    const float size = same ? (float)ATAN_N : -(float)ATAN_N;  // PPY PPX NNY NNX use +Size, the others -Size
    const float num = swap ? x : y, den = swap ? y : x;
    int idx = at_idx(size, num, special ? 1.f : den);
    idx = special ? 0 : idx;
    const float tv = ppy[idx];
    const float ys = ypos ? 1.0f : -1.0f;                      // (+-1 times a constant is exact: selects, not a tree of branches)
    const float A = swap ? Sh * ys : (xpos ? 0.f : St * ys);
    const float r = A + ((same == swap) ? -tv : tv);
    // (selects, not branches: the callers walk recurrences on lone waves, where a divergent branch costs more than both arms)
    const bool axis = x == 0.f && y != 0.f && !isnan(y) && !isinf(y);
    const float rs = axis ? (y > 0.f ? (float)(3.14159265358979323846 / 2) : (float)(-3.14159265358979323846 / 2)) : 0.f;
    return special ? rs : r;
}
// ---- limiter fm-demodulator.cpp:119-126 (std::abs(complex<float>) == hypotf == f64 sqrt of f64 sum)
__device__ __forceinline__ float2 limiter(float2 z) {
    const float zAbs = (float)sqrt((double)z.x * (double)z.x + (double)z.y * (double)z.y);
    if ((double)zAbs <= 0.001) return make_float2((float)0.001, (float)0.001);
    return make_float2(z.x / zAbs, z.y / zAbs);
}

// ---- f32 forms for the fused stage-B kernel (fmx_stageb.hip): the same expressions with the IEEE divisions, the f64 square
// root and the table gathers replaced by short sequences that agree with them to an ulp.
// n / d: reciprocal + one Markstein correction (the correctly rounded quotient except in rare half-way cases)
__device__ __forceinline__ float fdiv_fast(float n, float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    const float q = n * r;
    return __fmaf_rn(__fmaf_rn(-q, d, n), r, q);
}
// limiter fm-demodulator.cpp:119-126 with |z| from v_sqrt_f32 (1 ulp) and the two divisions by one reciprocal
__device__ __forceinline__ float2 limiter_fast(float2 z) {
    const float a = __fmaf_rn(z.x, z.x, z.y * z.y);
    const float h = __builtin_amdgcn_sqrtf(a);
    const float r = __builtin_amdgcn_rcpf(h);
    const float qx = z.x * r, qy = z.y * r;
    const float lx = __fmaf_rn(__fmaf_rn(-qx, h, z.x), r, qx), ly = __fmaf_rn(__fmaf_rn(-qy, h, z.y), r, qy);
    const bool tiny = !(h >= 0.001f);                     // (double) zAbs <= 0.001 <=> zAbs < 0.001f; NaN goes the same way
    return make_float2(tiny ? 0.001f : lx, tiny ? 0.001f : ly);
}
// compAtan::atan2 as lut_atan2 above, split around the table access so that a thread can issue the gathers of all its
// samples together; the index (int)(size * num / den + 0.5) in f32 (exact there: the quotient is <= 8192)
struct AtanArm { int idx; float A; float special; unsigned flags; };   // flags: bit 0 negate the table value, bit 1 special
__device__ __forceinline__ AtanArm atan_arm(float y, float x) {
    const float St = (float)3.14159265358979323846, Sh = St * 0.5f;
    const bool special = !(fabsf(x) > 0.f) || !(fabsf(x) < __builtin_inff()) || !(fabsf(y) < __builtin_inff());
    const bool xpos = x > 0.f, ypos = y >= 0.f;
    const bool swap = !(fabsf(x) >= fabsf(y));                 // the ..X arms: |y| > |x|
    const bool same = xpos == ypos;
    const float size = same ? (float)ATAN_N : -(float)ATAN_N;  // PPY PPX NNY NNX use +Size, the others -Size
    const float num = swap ? x : y, den = swap ? y : x;
    int idx = (int)(fdiv_fast(size * num, den) + 0.5f);
    idx = idx < 0 ? 0 : (idx > ATAN_N ? ATAN_N : idx);
    AtanArm a;
    a.idx = special ? 0 : idx;
    a.A = swap ? (ypos ? Sh : -Sh) : (xpos ? 0.f : (ypos ? St : -St));
    a.special = (x == 0.f && fabsf(y) > 0.f && fabsf(y) < __builtin_inff()) ? (y > 0.f ? (float)(3.14159265358979323846 / 2) : (float)(-3.14159265358979323846 / 2)) : 0.f;
    a.flags = ((same == swap) ? 1u : 0u) | (special ? 2u : 0u);
    return a;
}
__device__ __forceinline__ float atan_finish(const AtanArm &a, float tv) {
    const float r = a.A + ((a.flags & 1u) ? -tv : tv);
    return (a.flags & 2u) ? a.special : r;
}
// The same for finite arguments with x != 0 (everything but the corners Xtan2.cpp:56-68 handles first): no special value to carry.
// `odd` collects the arguments that are not of that kind (one v_cmp_class each); the caller looks at it once for all its samples and
// takes the general form above when it is set anywhere in the wave.
struct AtanArmF { int idx; float A; bool neg; };
__device__ __forceinline__ AtanArmF atan_arm_plain(float y, float x, bool *odd) {
    const float St = (float)3.14159265358979323846, Sh = St * 0.5f;
    // v_cmp_class masks: 0x001 sNaN 0x002 qNaN 0x004 -inf 0x010 -denormal 0x020 -0 0x040 +0 0x080 +denormal 0x200 +inf (denormal x: to the general form, whatever the denormal mode makes of it)
    // (the caller's x = I I1 + Q Q1 and y = Q I1 - I Q1 are sums of products that hold all four of its inputs: an infinity or a NaN in y comes from one of them,
    // and then x is not finite either -- x's class answers for y's)
    *odd = *odd || __builtin_amdgcn_classf(x, 0x2f7);
    const bool xpos = x > 0.f, ypos = y >= 0.f;
    const bool swap = !(fabsf(x) >= fabsf(y));
    const bool same = xpos == ypos;
    const float size = same ? (float)ATAN_N : -(float)ATAN_N;
    const float num = swap ? x : y, den = swap ? y : x;
    int idx = (int)(fdiv_fast(size * num, den) + 0.5f);
    AtanArmF a;
    a.idx = idx < 0 ? 0 : (idx > ATAN_N ? ATAN_N : idx);
    // (Sh * +-1, St * +-1 are exact; FLAT selects: the nested conditional  swap ? (ypos ? Sh : -Sh) : (xpos ? 0 : (ypos ? St : -St))  became divergent branches --
    // a dozen exec-mask instructions per sample in stage B's discriminator)
    const float ah = ypos ? Sh : -Sh, at = ypos ? St : -St;
    const float a0 = xpos ? 0.f : at;
    a.A = swap ? ah : a0;
    a.neg = same == swap;
    return a;
}
__device__ __forceinline__ float lut_atan2_fast(const float *__restrict__ ppy, float y, float x) {
    const AtanArm a = atan_arm(y, x);
    return atan_finish(a, ppy[a.idx]);
}
// SinCos table entries (sincos.cpp:45-54: (float) sin / cos of 2 pi idx / Rate) evaluated in f32 instead of gathered:
// quadrant / octant reduction of the INDEX in integers (exact), Taylor polynomials on the reduced angle.  |error| < 1.5e-7
// against the table for every idx (fmx_create checks all 192000 entries with this very code).
__host__ __device__ __forceinline__ float sin_idx_f32(int idx) {
    constexpr int QUAD = SINCOS_N / 4;                    // 48000 = 2^7 * 375
    const int q = (int)(((unsigned)(idx >> 7) * 2797u) >> 20);          // idx / 48000 for idx < 384000
    const int r = idx - q * QUAD;
    const int rr = (q & 1) ? QUAD - r : r;
    const float x = (float)rr * (float)(6.283185307179586476925286766559 / SINCOS_N);
    const float z = x * x;
    float p = -1.0f / 39916800.0f;                        // x - x^3/6 + ... - x^11/11! + x^13/13!: truncation < 1e-8 on [0, pi/2]
    p = __builtin_fmaf(z, 1.0f / 6227020800.0f, p);
    p = __builtin_fmaf(p, z, 1.0f / 362880.0f);
    p = __builtin_fmaf(p, z, -1.0f / 5040.0f);
    p = __builtin_fmaf(p, z, 1.0f / 120.0f);
    p = __builtin_fmaf(p, z, -1.0f / 6.0f);
    const float s = __builtin_fmaf(x * z, p, x);
    return (q & 2) ? -s : s;
}
__host__ __device__ __forceinline__ void sincos_idx_f32(int idx, float *sn, float *cs) {
    constexpr int OCT = SINCOS_N / 8;                     // 24000 = 2^6 * 375
    const int q = (int)(((unsigned)(idx >> 6) * 2797u) >> 20);          // idx / 24000 for idx < 192000
    const int r = idx - q * OCT;
    const int rr = (q & 1) ? OCT - r : r;
    const float x = (float)rr * (float)(6.283185307179586476925286766559 / SINCOS_N);
    const float z = x * x;
    float ps = 1.0f / 362880.0f;                          // sin to x^9, cos to x^8 on [0, pi/4]: truncation < 3e-8
    ps = __builtin_fmaf(ps, z, -1.0f / 5040.0f);
    ps = __builtin_fmaf(ps, z, 1.0f / 120.0f);
    ps = __builtin_fmaf(ps, z, -1.0f / 6.0f);
    const float sv = __builtin_fmaf(x * z, ps, x);
    float pc = 1.0f / 40320.0f;
    pc = __builtin_fmaf(pc, z, -1.0f / 720.0f);
    pc = __builtin_fmaf(pc, z, 1.0f / 24.0f);
    pc = __builtin_fmaf(pc, z, -0.5f);
    const float cv = __builtin_fmaf(z, pc, 1.0f);
    // angle = k pi/4 + a (k even) or (k + 1) pi/4 - b (k odd), k = q & 3; q >= 4 negates both
    const bool sw = ((q + 1) & 2) != 0;                   // k = 1, 2: sine from the cosine polynomial
    const float s = sw ? cv : sv, c = sw ? sv : cv;
    *sn = (q & 4) ? -s : s;
    *cs = (((q >> 1) ^ (q >> 2)) & 1) ? -c : c;
}

// The same table entries from the hardware sine unit (v_sin_f32 / v_cos_f32 take turns): the index is folded into the first quadrant in
// integers (exact), so the argument is at most a quarter turn and carries 2^-26 turns of rounding; measured against the table for all
// 192000 entries (tools/ubench/hwsin.hip on gfx950): worst |error| 1.19e-7 (one ulp at 1), rms 4.1e-8 -- inside the polynomials'
// 1.5e-7 -- at a third of their instructions.
// (the fold without a division: h = idx mod N/2 by an unsigned minimum, then its mirror image about the quarter turn; the same rr and
// the same signs as q = idx / (N/4), r = idx - q N/4, rr = q odd ? N/4 - r : r, sine negative for q >= 2, cosine for q = 1, 2)
__device__ __forceinline__ float sin_idx_hw(int idx) {            // 0 <= idx < 192000
    constexpr unsigned HALF = SINCOS_N / 2;
    const unsigned u = (unsigned)idx;
    const unsigned uh = u - HALF;                                  // (wraps to a huge value -- bit 31 set -- exactly when u < HALF)
    const unsigned h = min(u, uh);
    const unsigned rr = min(h, HALF - h);
    const float s = __builtin_amdgcn_sinf((float)rr * (1.0f / (float)SINCOS_N));
    return __uint_as_float(__float_as_uint(s) ^ (~uh & 0x80000000u));      // -s for u >= HALF: the sign put on as a bit (one operation, no comparison and select)
}
__device__ __forceinline__ void sincos_idx_hw(int idx, float *sn, float *cs) {      // 0 <= idx < 192000
    constexpr unsigned HALF = SINCOS_N / 2, QUAD = SINCOS_N / 4;
    const unsigned u = (unsigned)idx;
    const unsigned h = min(u, u - HALF);
    const unsigned rr = min(h, HALF - h);
    const float t = (float)rr * (1.0f / (float)SINCOS_N);
    const float s = __builtin_amdgcn_sinf(t), c = __builtin_amdgcn_cosf(t);
    *sn = (u >= HALF) ? -s : s;
    *cs = (u - QUAD < HALF) ? -c : c;                              // N/4 <= idx < 3N/4
}
// the same values with the signs put on as bits (one three-input bit operation each instead of a comparison and a select -- for the lone waves of
// fmx_demod.hip, where every instruction is four cycles): u - HALF wraps (bit 31 set) exactly when u < HALF, h - QUAD exactly when h < QUAD;
// the sine is negative for u >= HALF, the cosine for QUAD <= u < 3 QUAD, i.e. when exactly one of the two wraps
__device__ __forceinline__ void sincos_idx_hw_bits(int idx, float *sn, float *cs) {      // 0 <= idx < 192000
    constexpr unsigned HALF = SINCOS_N / 2, QUAD = SINCOS_N / 4;
    const unsigned u = (unsigned)idx;
    const unsigned uh = u - HALF;
    const unsigned h = min(u, uh);
    const unsigned rr = min(h, HALF - h);
    const float t = (float)rr * (1.0f / (float)SINCOS_N);
    const float s = __builtin_amdgcn_sinf(t), c = __builtin_amdgcn_cosf(t);
    *sn = __uint_as_float(__float_as_uint(s) ^ (~uh & 0x80000000u));
    *cs = __uint_as_float(__float_as_uint(c) ^ ((uh ^ (h - QUAD)) & 0x80000000u));
}

}  // namespace fmx
