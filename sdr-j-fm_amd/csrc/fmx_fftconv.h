// fmx_fftconv.h -- a 2048-point circular convolution for one 256-thread workgroup (eight points per thread), written so that
// the very same arithmetic runs on the host: fmx_create pushes the PSS low-pass taps through the forward half (which yields
// their spectrum in exactly the order the device's forward transform leaves its output in, whatever that order is), and
// tests/test_fftconv_cpu.py drives the whole thing thread by thread against a direct convolution.
//
// The reference filters with fftFilter (fft-filters.cpp:132-163: 2048-point overlap-add, 295 taps); stage B needs the same
// linear convolution for 1536 fresh outputs per segment: a window of 1536 + 294 inputs, zero-padded to 2048, forward
// transform, times the taps' spectrum (1 / N folded in), backward transform; outputs 294 .. 294 + 1535 are free of wrap-around.
//
// 2048 = 8 * 8 * 8 * 4, decimation in frequency forward (natural order in, digit-reversed out), the mirror image backward,
// so no reordering pass exists: thread t starts with the natural-order points t + 256 p straight from global memory and ends
// with the natural-order outputs t + 256 p in registers.  The radix-4 stage in the middle works on four points that sit in
// the same register of four adjacent lanes, so it runs across the quad in DPP (forward, times the spectrum, backward) and the
// data meets LDS only four times.  Stage L = 256 and L = 32 read and write the same addresses from the same half-wave / quad:
// in place, no barrier between load and store; four barriers per convolution.
// LDS index i -> i + 4 (i >> 5): every access pattern below is conflict-free for 8-byte elements.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <vector>

namespace fmx {
namespace fftc {

constexpr int N = 2048, T = 256, LDS_N = N + 4 * (N >> 5);
__host__ __device__ __forceinline__ int pad(int i) { return i + 4 * (i >> 5); }

// Complex arithmetic on the native two-float vector, so that every operation is ONE packed instruction (v_pk_add_f32,
// v_pk_mul_f32, v_pk_fma_f32; the (re, im) swaps and sign flips ride on their op_sel / neg modifiers).  The products are fused
// multiply-adds -- on the host too (make_spectrum runs this very code), whatever -ffp-contract says.
typedef float cf __attribute__((ext_vector_type(2)));
__host__ __device__ __forceinline__ cf V(float2 a) { cf r; r.x = a.x; r.y = a.y; return r; }
__host__ __device__ __forceinline__ float2 F(cf a) { return make_float2(a.x, a.y); }
__host__ __device__ __forceinline__ cf vfma(cf a, cf b, cf c) { return __builtin_elementwise_fma(a, b, c); }
// SIGN j t as t.yx times (-SIGN, SIGN): the swap rides on the instruction's op_sel, so "base + SIGN j t" is one packed fma
template <int SIGN> __host__ __device__ __forceinline__ cf jsign() { cf m; m.x = (float)-SIGN; m.y = (float)SIGN; return m; }
template <int SIGN> __host__ __device__ __forceinline__ cf vmulj(cf a) { return a.yx * jsign<SIGN>(); }
template <int SIGN> __host__ __device__ __forceinline__ cf add_j(cf base, cf t) { return vfma(t.yx, jsign<SIGN>(), base); }    // base + SIGN j t
template <int SIGN> __host__ __device__ __forceinline__ cf sub_j(cf base, cf t) { return vfma(t.yx, jsign<-SIGN>(), base); }   // base - SIGN j t
__host__ __device__ __forceinline__ cf vcmul(cf a, cf b) {                  // a b = a.x (b.x, b.y) + a.y (-b.y, b.x)
    return vfma(a.yy, vmulj<+1>(b), a.xx * b);
}
__host__ __device__ __forceinline__ cf vcmulc(cf a, cf b) {                 // a conj (b) = b.x (a.x, a.y) + b.y (a.y, -a.x)
    return vfma(b.yy, vmulj<-1>(a), b.xx * a);
}
__host__ __device__ __forceinline__ float2 cadd(float2 a, float2 b) { return F(V(a) + V(b)); }
__host__ __device__ __forceinline__ float2 csub(float2 a, float2 b) { return F(V(a) - V(b)); }
__host__ __device__ __forceinline__ float2 cmul(float2 a, float2 b) { return F(vcmul(V(a), V(b))); }
__host__ __device__ __forceinline__ float2 cmulc(float2 a, float2 b) { return F(vcmulc(V(a), V(b))); }   // a * conj (b)
// multiplication by SIGN * j
template <int SIGN> __host__ __device__ __forceinline__ float2 mulj(float2 a) { return F(vmulj<SIGN>(V(a))); }

// X[q] = sum_p a[p] w^(p q), w = exp (SIGN * 2 pi i / 4)
template <int SIGN> __host__ __device__ __forceinline__ void dft4(cf &a0, cf &a1, cf &a2, cf &a3) {
    const cf s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
    a0 = s02 + s13; a2 = s02 - s13; a1 = add_j<SIGN>(d02, d13); a3 = sub_j<SIGN>(d02, d13);
}
// the same with a2 standing for SIGN j a2 (the caller's rotation folded in)
template <int SIGN> __host__ __device__ __forceinline__ void dft4_j2(cf &a0, cf &a1, cf &a2, cf &a3) {
    const cf s02 = add_j<SIGN>(a0, a2), d02 = sub_j<SIGN>(a0, a2), s13 = a1 + a3, d13 = a1 - a3;
    a0 = s02 + s13; a2 = s02 - s13; a1 = add_j<SIGN>(d02, d13); a3 = sub_j<SIGN>(d02, d13);
}
// X[q] = sum_p a[p] w^(p q), w = exp (SIGN * 2 pi i / 8), in place
template <int SIGN> __host__ __device__ __forceinline__ void dft8(float2 *a) {
    const cf R = {0.70710678118654752440f, 0.70710678118654752440f};
    cf e0 = V(a[0]) + V(a[4]), e1 = V(a[1]) + V(a[5]), e2 = V(a[2]) + V(a[6]), e3 = V(a[3]) + V(a[7]);
    cf o0 = V(a[0]) - V(a[4]), o1 = V(a[1]) - V(a[5]), o2 = V(a[2]) - V(a[6]), o3 = V(a[3]) - V(a[7]);
    // o_p *= w^p: w = (1 + SIGN j) / sqrt 2, w^2 = SIGN j (inside dft4_j2), w^3 = (-1 + SIGN j) / sqrt 2
    o1 = add_j<SIGN>(o1, o1) * R;
    o3 = add_j<SIGN>(-o3, o3) * R;
    dft4<SIGN>(e0, e1, e2, e3);          // X[0], X[2], X[4], X[6]
    dft4_j2<SIGN>(o0, o1, o2, o3);       // X[1], X[3], X[5], X[7]
    a[0] = F(e0); a[2] = F(e1); a[4] = F(e2); a[6] = F(e3); a[1] = F(o0); a[3] = F(o1); a[5] = F(o2); a[7] = F(o3);
}

// A radix-8 stage works on blocks of length L: the butterfly j of a block (j < S = L / 8) takes the points base + j + p S,
// forward: DFT8, output q times exp (-2 pi i j q / L), stored at base + q S + j; backward: the inverse.
template <int L> __host__ __device__ __forceinline__ void geom8(int t, int &base, int &j) {
    constexpr int S = L / 8;
    j = t % S; base = (t / S) * L;       // (L = 32: the four butterflies of a block are the four lanes of a quad)
}
// The twiddles a stage needs, stored per stage as [q - 1][j]: a wave's loads are contiguous (the plain table exp (-2 pi i k / N)
// indexed by (j q) (N / L) costs a cache line per lane).
constexpr int W_OFF_2048 = 0, W_OFF_256 = 7 * 256, W_OFF_32 = W_OFF_256 + 7 * 32, W_COUNT = W_OFF_32 + 7 * 4;
template <int L> __host__ __device__ __forceinline__ int w_index(int q, int j) {
    return (L == 2048 ? W_OFF_2048 : (L == 256 ? W_OFF_256 : W_OFF_32)) + (q - 1) * (L / 8) + j;
}
// tw[q - 1] = twiddle of output q
template <int L> __host__ __device__ __forceinline__ void load_tw(float2 *tw, int j, const float2 *__restrict__ W) {
#pragma unroll
    for (int q = 1; q < 8; q++) tw[q - 1] = W[w_index<L>(q, j)];
}
__host__ __device__ __forceinline__ void fwd8(float2 *a, const float2 *tw) {
    dft8<-1>(a);
#pragma unroll
    for (int q = 1; q < 8; q++) a[q] = cmul(a[q], tw[q - 1]);
}
__host__ __device__ __forceinline__ void inv8(float2 *a, const float2 *tw) {
#pragma unroll
    for (int q = 1; q < 8; q++) a[q] = cmulc(a[q], tw[q - 1]);
    dft8<+1>(a);
}
template <int L> __host__ __device__ __forceinline__ void fwd8(float2 *a, int j, const float2 *__restrict__ W) { float2 tw[7]; load_tw<L>(tw, j, W); fwd8(a, tw); }
template <int L> __host__ __device__ __forceinline__ void inv8(float2 *a, int j, const float2 *__restrict__ W) { float2 tw[7]; load_tw<L>(tw, j, W); inv8(a, tw); }
// points base + j + p S (the inputs of a forward / outputs of a backward butterfly)
template <int L> __host__ __device__ __forceinline__ void load_p(float2 *a, const float2 *X, int base, int j) {
#pragma unroll
    for (int p = 0; p < 8; p++) a[p] = X[pad(base + j + p * (L / 8))];
}
template <int L> __host__ __device__ __forceinline__ void store_p(const float2 *a, float2 *X, int base, int j) {
#pragma unroll
    for (int p = 0; p < 8; p++) X[pad(base + j + p * (L / 8))] = a[p];
}
// points base + q S + j (the outputs of a forward / inputs of a backward butterfly)
template <int L> __host__ __device__ __forceinline__ void load_q(float2 *a, const float2 *X, int base, int j) {
#pragma unroll
    for (int q = 0; q < 8; q++) a[q] = X[pad(base + q * (L / 8) + j)];
}
template <int L> __host__ __device__ __forceinline__ void store_q(const float2 *a, float2 *X, int base, int j) {
#pragma unroll
    for (int q = 0; q < 8; q++) X[pad(base + q * (L / 8) + j)] = a[q];
}

// The radix-4 stage across a quad: lane j (= t & 3) holds point j of a block of four.  Two exchanges (partner = lane ^ 2, then
// lane ^ 1) leave (Y0, Y2, Y1, Y3) in lanes (0, 1, 2, 3); the backward pair undoes them (times 4).  `own` / `other` = this
// lane's and the partner's value; the lane-dependent sign is a multiplier (own * (+-1) + other: one packed fma), not a select.
template <int SIGN> __host__ __device__ __forceinline__ float2 quad_f1(int j, float2 own, float2 other) {
    const float sg = (j & 2) ? -1.f : 1.f;
    const cf sv = {sg, sg};
    const cf s = vfma(V(own), sv, V(other));
    const float k = (j == 3) ? 0.f : 1.f, r = (j == 3) ? (float)SIGN : 0.f;      // lane 3: times SIGN j, as s k + s.yx (-r, r)
    const cf kv = {k, k}, rv = {-r, r};
    return F(vfma(s.yx, rv, s * kv));
}
__host__ __device__ __forceinline__ float2 quad_f2(int j, float2 own, float2 other) {
    const float sg = (j & 1) ? -1.f : 1.f;
    const cf sv = {sg, sg};
    return F(vfma(V(own), sv, V(other)));
}
template <int SIGN> __host__ __device__ __forceinline__ float2 quad_b1(int j, float2 own, float2 other) {
    const float sg = (j & 1) ? -1.f : 1.f;
    const cf sv = {sg, sg};
    const cf s = vfma(V(own), sv, V(other));
    const float k = (j == 3) ? 0.f : 1.f, r = (j == 3) ? (float)-SIGN : 0.f;     // lane 3: times -SIGN j
    const cf kv = {k, k}, rv = {-r, r};
    return F(vfma(s.yx, rv, s * kv));
}
__host__ __device__ __forceinline__ float2 quad_b2(int j, float2 own, float2 other) {
    const float sg = (j & 2) ? -1.f : 1.f;
    const cf sv = {sg, sg};
    return F(vfma(V(own), sv, V(other)));
}

template <int CTRL> __device__ __forceinline__ float2 quad_get(float2 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return make_float2(__int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v.x), CTRL, 0xf, 0xf, true)),
                       __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v.y), CTRL, 0xf, 0xf, true)));
#else
    return v;
#endif
}
// thread t of the workgroup: `a` = the natural-order inputs t + 256 p on entry, the natural-order outputs t + 256 p on return.
// X: LDS_N complex of LDS; W: twiddles; Hs: the spectrum in "slot" order (entry 8 t + q = what thread t holds in register q in
// front of the multiplication, make_spectrum below), 1 / N included.
// forward half: `a` = natural-order inputs t + 256 p on entry, slot values (register q of thread t) on return
// (hs / Hs: when given, the thread's eight spectrum entries Hs[8 t ..] are requested in front of the second barrier)
__device__ __forceinline__ void forward_slots(int t, float2 *a, float2 *X, const float2 *__restrict__ W, float4 *hs = nullptr, const float2 *__restrict__ Hs = nullptr) {
    // (every stage's twiddles are requested a stage ahead: their latency lands under the butterflies and the barrier in front)
    int base, j;
    float2 twa[7], twb[7];
    load_tw<2048>(twa, t, W);
    geom8<256>(t, base, j);
    load_tw<256>(twb, j, W);
    fwd8(a, twa); store_q<2048>(a, X, 0, t);
    {
        int b32, j32;
        geom8<32>(t, b32, j32);
        load_tw<32>(twa, j32, W);
    }
    __syncthreads();
    load_p<256>(a, X, base, j); fwd8(a, twb); store_q<256>(a, X, base, j);
    if (hs) {
        const float4 *H4 = reinterpret_cast<const float4 *>(Hs + 8 * t);
#pragma unroll
        for (int q = 0; q < 4; q++) hs[q] = H4[q];
    }
    __syncthreads();
    geom8<32>(t, base, j);
    load_p<32>(a, X, base, j); fwd8(a, twa);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        float2 v = a[q];
        v = quad_f1<-1>(j, v, quad_get<0x4E>(v));                // quad_perm [2, 3, 0, 1]
        a[q] = quad_f2(j, v, quad_get<0xB1>(v));                 // quad_perm [1, 0, 3, 2]
    }
}
// backward half: slot values in, natural-order outputs t + 256 p out (times N)
__device__ __forceinline__ void backward_slots(int t, float2 *a, float2 *X, const float2 *__restrict__ W) {
    int base, j;
    float2 twa[7], twb[7];
    geom8<32>(t, base, j);
    load_tw<32>(twa, j, W);
    {
        int b256, j256;
        geom8<256>(t, b256, j256);
        load_tw<256>(twb, j256, W);
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
        float2 v = a[q];
        v = quad_b1<-1>(j, v, quad_get<0xB1>(v));
        a[q] = quad_b2(j, v, quad_get<0x4E>(v));
    }
    inv8(a, twa); store_p<32>(a, X, base, j);
    load_tw<2048>(twa, t, W);
    __syncthreads();
    geom8<256>(t, base, j);
    load_q<256>(a, X, base, j); inv8(a, twb); store_p<256>(a, X, base, j);
    __syncthreads();
    load_q<2048>(a, X, 0, t); inv8(a, twa);
}
// a[q] *= Hs[8 t + q]
__device__ __forceinline__ void times_spectrum(int t, float2 *a, const float2 *__restrict__ Hs) {
    const float4 *H4 = reinterpret_cast<const float4 *>(Hs + 8 * t);
#pragma unroll
    for (int q = 0; q < 4; q++) { const float4 v = H4[q]; a[2 * q] = cmul(a[2 * q], make_float2(v.x, v.y)); a[2 * q + 1] = cmul(a[2 * q + 1], make_float2(v.z, v.w)); }
}
// (the caller must put a barrier between the last LDS read of one transform and the first LDS write of the next: the end of
// forward_slots and of backward_slots reads X)
__device__ __forceinline__ void convolve(int t, float2 *a, float2 *X, const float2 *__restrict__ W, const float2 *__restrict__ Hs) {
    float4 hs[4];
    forward_slots(t, a, X, W, hs, Hs);
#pragma unroll
    for (int q = 0; q < 4; q++) { a[2 * q] = cmul(a[2 * q], make_float2(hs[q].x, hs[q].y)); a[2 * q + 1] = cmul(a[2 * q + 1], make_float2(hs[q].z, hs[q].w)); }
    backward_slots(t, a, X, W);
}

// ---- host side: the same arithmetic, all 256 threads in turn between the barriers --------------------------------------
template <int L> inline void make_twiddles_stage(float2 *W) {
    for (int q = 1; q < 8; q++)
        for (int j = 0; j < L / 8; j++) {
            const double ang = -2.0 * 3.14159265358979323846 * (double)(j * q) / (double)L;
            W[w_index<L>(q, j)] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        }
}
inline void make_twiddles(float2 *W /*[W_COUNT]*/) { make_twiddles_stage<2048>(W); make_twiddles_stage<256>(W); make_twiddles_stage<32>(W); }
// forward half for all threads: in[n] natural order -> slot[8 t + q] (what thread t holds in register q in front of the multiplication)
inline void host_forward(const float2 *in, float2 *slot, const float2 *W) {
    std::vector<float2> X(LDS_N), R((size_t)T * 8);
    float2 a[8];
    int base, j;
    for (int t = 0; t < T; t++) { for (int p = 0; p < 8; p++) a[p] = in[t + 256 * p]; fwd8<2048>(a, t, W); store_q<2048>(a, X.data(), 0, t); }
    for (int t = 0; t < T; t++) { geom8<256>(t, base, j); load_p<256>(a, X.data(), base, j); fwd8<256>(a, j, W); for (int q = 0; q < 8; q++) R[(size_t)t * 8 + q] = a[q]; }
    for (int t = 0; t < T; t++) { geom8<256>(t, base, j); store_q<256>(&R[(size_t)t * 8], X.data(), base, j); }
    for (int t = 0; t < T; t++) { geom8<32>(t, base, j); load_p<32>(a, X.data(), base, j); fwd8<32>(a, j, W); for (int q = 0; q < 8; q++) R[(size_t)t * 8 + q] = a[q]; }
    for (int t0 = 0; t0 < T; t0 += 4)
        for (int q = 0; q < 8; q++) {
            float2 v[4], s[4];
            for (int k = 0; k < 4; k++) v[k] = R[(size_t)(t0 + k) * 8 + q];
            for (int k = 0; k < 4; k++) s[k] = quad_f1<-1>(k, v[k], v[k ^ 2]);
            for (int k = 0; k < 4; k++) slot[(size_t)(t0 + k) * 8 + q] = quad_f2(k, s[k], s[k ^ 1]);
        }
}
// backward half for all threads: slot order in -> natural order out (times N)
inline void host_backward(const float2 *slot, float2 *out, const float2 *W) {
    std::vector<float2> X(LDS_N), R((size_t)T * 8);
    float2 a[8];
    int base, j;
    for (int t0 = 0; t0 < T; t0 += 4)
        for (int q = 0; q < 8; q++) {
            float2 v[4], s[4];
            for (int k = 0; k < 4; k++) v[k] = slot[(size_t)(t0 + k) * 8 + q];
            for (int k = 0; k < 4; k++) s[k] = quad_b1<-1>(k, v[k], v[k ^ 1]);
            for (int k = 0; k < 4; k++) R[(size_t)(t0 + k) * 8 + q] = quad_b2(k, s[k], s[k ^ 2]);
        }
    for (int t = 0; t < T; t++) { geom8<32>(t, base, j); for (int q = 0; q < 8; q++) a[q] = R[(size_t)t * 8 + q]; inv8<32>(a, j, W); store_p<32>(a, X.data(), base, j); }
    for (int t = 0; t < T; t++) { geom8<256>(t, base, j); load_q<256>(a, X.data(), base, j); inv8<256>(a, j, W); for (int q = 0; q < 8; q++) R[(size_t)t * 8 + q] = a[q]; }
    for (int t = 0; t < T; t++) { geom8<256>(t, base, j); store_p<256>(&R[(size_t)t * 8], X.data(), base, j); }
    for (int t = 0; t < T; t++) { load_q<2048>(a, X.data(), 0, t); inv8<2048>(a, t, W); for (int p = 0; p < 8; p++) out[t + 256 * p] = a[p]; }
}
// spectrum of `ntaps` real taps in slot order, 1 / N included
inline void make_spectrum(const float *taps, int ntaps, float2 *Hs, const float2 *W) {
    std::vector<float2> in(N, make_float2(0.f, 0.f));
    for (int k = 0; k < ntaps; k++) in[k] = make_float2(taps[k], 0.f);
    host_forward(in.data(), Hs, W);
    for (int k = 0; k < N; k++) { Hs[k].x *= 1.0f / N; Hs[k].y *= 1.0f / N; }
}

}  // namespace fftc
}  // namespace fmx
