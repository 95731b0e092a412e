// fdiv_inrange (fmx_demod.hip: the compiler's IEEE f32 division without v_div_scale / v_div_fixup) against n / d, bit for bit, over the operand range
// the PLL decoder's arc-tangent uses it on: |d| in [0.0009, 1.5] (the larger component of a limited sample times a unit vector), |n| <= 8192 |d|.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off fdiv_check.hip -o fdiv_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
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
// n / d with one Markstein correction (fmx_demod_math.h fdiv_fast): four operations; the quotient is the IEEE one except in rare half-way cases -- counted
// here, with how often the arc-tangent's table index (int)(q + 0.5) differs for it
__device__ __forceinline__ float fdiv_fast(float n, float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    const float q = n * r;
    return __fmaf_rn(__fmaf_rn(-q, d, n), r, q);
}
__device__ __forceinline__ uint32_t rnd(uint64_t &s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 32); }
__global__ void k(unsigned long long *bad, float *ex, int iters, unsigned long long *bad2) {
    uint64_t s = (uint64_t)(blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    unsigned long long b = 0, bq = 0, bi = 0;
    for (int it = 0; it < iters; it++) {
        const float u = (float)rnd(s) * (1.0f / 4294967296.0f), v = (float)rnd(s) * (1.0f / 4294967296.0f);
        const uint32_t w = rnd(s);
        float d = (w & 4) ? 0.0009f + u * 1.5f : ((w & 8) ? 0.70710678f + u * 0.3f : 0.0009f + u * 0.02f);
        if (w & 1) d = -d;
        float n = 8192.0f * d * ((w & 16) ? v : v * v * v);           // |n| <= 8192 |d|, small ratios as likely as large ones
        if (w & 2) n = -n;
        const float a = n / d, c = fdiv_inrange(n, d);
        const float f = fdiv_fast(n, d);
        if (__float_as_uint(a) != __float_as_uint(f)) bq++;
        if ((int)(fabsf(a) + 0.49999997f) != (int)(fabsf(f) + 0.49999997f)) bi++;
        if (__float_as_uint(a) != __float_as_uint(c)) { if (!b) { ex[0] = n; ex[1] = d; ex[2] = a; ex[3] = c; } b++; }
    }
    if (b) atomicAdd(bad, b);
    if (bq) atomicAdd(bad2, bq);
    if (bi) atomicAdd(bad2 + 1, bi);
}
int main() {
    unsigned long long *bad; float *ex;
    (void)hipMalloc(&bad, 8); (void)hipMemset(bad, 0, 8); (void)hipMalloc(&ex, 16); (void)hipMemset(ex, 0, 16);
    unsigned long long *bad2; (void)hipMalloc(&bad2, 16); (void)hipMemset(bad2, 0, 16);
    const int blocks = 4096, iters = 4096;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, bad, ex, iters, bad2);
    unsigned long long hb = 0; float hex[4];
    (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hex, ex, 16, hipMemcpyDeviceToHost);
    printf("%llu operand pairs, %llu differ", (unsigned long long)blocks * 256 * iters, hb);
    if (hb) printf("  (e.g. %a / %a: %a against %a)", hex[0], hex[1], hex[2], hex[3]);
    printf("\n");
    unsigned long long h2[2]; (void)hipMemcpy(h2, bad2, 16, hipMemcpyDeviceToHost);
    printf("fdiv_fast (one correction step): %llu quotients differ from n / d, %llu table indices (int)(|q| + 0.5)\n", h2[0], h2[1]);
    return hb != 0;
}
