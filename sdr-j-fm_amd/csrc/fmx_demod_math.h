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
    const float A = swap ? (ypos ? Sh : -Sh) : (xpos ? 0.f : (ypos ? St : -St));
    const float r = A + ((same == swap) ? -tv : tv);
    if (special) {
        if (x == 0.f && y != 0.f && !isnan(y) && !isinf(y)) return y > 0.f ? (float)(3.14159265358979323846 / 2) : (float)(-3.14159265358979323846 / 2);
        return 0.f;
    }
    return r;
}
// ---- limiter fm-demodulator.cpp:119-126 (std::abs(complex<float>) == hypotf == f64 sqrt of f64 sum)
__device__ __forceinline__ float2 limiter(float2 z) {
    const float zAbs = (float)sqrt((double)z.x * (double)z.x + (double)z.y * (double)z.y);
    if ((double)zAbs <= 0.001) return make_float2((float)0.001, (float)0.001);
    return make_float2(z.x / zAbs, z.y / zAbs);
}


}  // namespace fmx
