// micro-benchmark: what does a lane-per-channel sequential kernel cost on MI355X?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void alu_chain(float *out, int n, float a) {
    float y = threadIdx.x * 1e-3f;
    for (int i = 0; i < n; i++) y = (0.5f - y) * a + y;      // 3 dependent f32 ops
    out[blockIdx.x * 64 + threadIdx.x] = y;
}
__global__ void alu_chain_clk(float *out, long long *clk, int n, float a) {
    float y = threadIdx.x * 1e-3f;
    long long t0 = clock64(); long long w0 = wall_clock64();
    for (int i = 0; i < n; i++) y = (0.5f - y) * a + y;
    long long t1 = clock64(); long long w1 = wall_clock64();
    out[blockIdx.x * 64 + threadIdx.x] = y;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}
template <int UB>
__global__ void strided_rw(float2 *x, int nj, int pitch, float a) {
    const int ch = blockIdx.x * 64 + threadIdx.x;
    float2 *p = x + ch;
    float yl = 0, yr = 0;
    float2 nx[UB];
#pragma unroll
    for (int k = 0; k < UB; k++) nx[k] = p[(long long)k * pitch];
    for (int r0 = 0; r0 < nj; r0 += UB) {
        float2 v[UB];
#pragma unroll
        for (int k = 0; k < UB; k++) v[k] = nx[k];
#pragma unroll
        for (int k = 0; k < UB; k++) { int r = r0 + UB + k; r = r < nj - 1 ? r : nj - 1; nx[k] = p[(long long)r * pitch]; }
#pragma unroll
        for (int k = 0; k < UB; k++) {
            yl = (v[k].x - yl) * a + yl; yr = (v[k].y - yr) * a + yr;
            p[(long long)(r0 + k) * pitch] = make_float2(yl, yr);
        }
    }
}
template <int UB>
__global__ void strided_r_only(const float2 *x, float2 *out, int nj, int pitch, float a) {
    const int ch = blockIdx.x * 64 + threadIdx.x;
    const float2 *p = x + ch;
    float yl = 0, yr = 0;
    for (int r0 = 0; r0 < nj; r0 += UB) {
        float2 v[UB];
#pragma unroll
        for (int k = 0; k < UB; k++) v[k] = p[(long long)(r0 + k) * pitch];
#pragma unroll
        for (int k = 0; k < UB; k++) { yl = (v[k].x - yl) * a + yl; yr = (v[k].y - yr) * a + yr; }
    }
    out[ch] = make_float2(yl, yr);
}
int main() {
    const int C = 512, NJ = 19200, pitch = 576;
    float2 *x; float *o; long long *clk; float2 *o2;
    CK(hipMalloc(&x, sizeof(float2) * (size_t)NJ * pitch)); CK(hipMemset(x, 0, sizeof(float2) * (size_t)NJ * pitch));
    CK(hipMalloc(&o, 4 * 65536)); CK(hipMalloc(&clk, 16 * 1024)); CK(hipMalloc(&o2, 8 * 65536));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto fn, double per) {
        fn(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int i = 0; i < 3; i++) fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
        printf("%-40s %8.3f ms  %8.1f ns/iter\n", name, ms, ms * 1e6 / per);
        return 0;
    };
    for (int blocks : {8, 64, 1024}) {
        char nm[64]; snprintf(nm, 64, "alu_chain blocks=%d", blocks);
        timeit(nm, [&] { hipLaunchKernelGGL(alu_chain, dim3(blocks), dim3(64), 0, 0, o, 200000, 0.1f); }, 200000);
    }
    hipLaunchKernelGGL(alu_chain_clk, dim3(8), dim3(64), 0, 0, o, clk, 200000, 0.1f);
    CK(hipDeviceSynchronize());
    long long h[16]; CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    printf("clock64 ticks/iter %.2f  wall_clock64 ticks/iter %.2f (100 MHz?)\n", h[0] / 200000.0, h[1] / 200000.0);
    timeit("strided_rw UB=16 (deemph-like)", [&] { hipLaunchKernelGGL(strided_rw<16>, dim3(C / 64), dim3(64), 0, 0, x, NJ, pitch, 0.1f); }, NJ);
    timeit("strided_rw UB=64", [&] { hipLaunchKernelGGL(strided_rw<64>, dim3(C / 64), dim3(64), 0, 0, x, NJ, pitch, 0.1f); }, NJ);
    timeit("strided_r_only UB=16", [&] { hipLaunchKernelGGL(strided_r_only<16>, dim3(C / 64), dim3(64), 0, 0, x, o2, NJ, pitch, 0.1f); }, NJ);
    timeit("strided_r_only UB=64", [&] { hipLaunchKernelGGL(strided_r_only<64>, dim3(C / 64), dim3(64), 0, 0, x, o2, NJ, pitch, 0.1f); }, NJ);
    return 0;
}
