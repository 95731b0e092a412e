// micro-benchmark: issue rate of v_pk_fma_f32 vs v_fma_f32 on gfx950 (per wave, 1 or 2 waves per SIMD),
// with the multiplier in an SGPR (as the front kernel's FIR uses it).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, long long *clk, int n, float a, float b) {
    v2f acc[8];
    for (int i = 0; i < 8; i++) acc[i] = (v2f){threadIdx.x * 1e-3f + i, 1.f};
    v2f x[8];
    for (int i = 0; i < 8; i++) x[i] = (v2f){threadIdx.x * 1e-4f + i, 0.5f};
    long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) acc[i] = __builtin_elementwise_fma((v2f){a, a}, x[(i + u) & 7], acc[i]);           // pk, sgpr
                else if (MODE == 1) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "s"(a), "v"(x[(i + u) & 7].x)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "s"(a), "v"(x[(i + u) & 7].y)); }
                else if (MODE == 2) acc[i] = __builtin_elementwise_fma(x[(i + u + 1) & 7], x[(i + u) & 7], acc[i]);   // pk, all vgpr
                else asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(acc[i]) : "v"(x[(i + u) & 7]), "v"(acc[i]));
            }
        }
    }
    long long t1 = clock64(), w1 = wall_clock64();
    float s = 0; for (int i = 0; i < 8; i++) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}
int main() {
    float *out; long long *clk; const int n = 4000;
    CK(hipMalloc(&out, 4 * 256 * 2048)); CK(hipMalloc(&clk, 16 * 2048));
    const char *names[] = {"v_pk_fma_f32 sgpr*vgpr", "2 x v_fma_f32 sgpr*vgpr", "v_pk_fma_f32 vgpr*vgpr", "v_pk_mul_f32"};
    for (int mode = 0; mode < 4; mode++)
        for (int blocks : {256, 512, 1024, 2048}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, clk, n, 1.0001f, 0.5f);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, clk, n, 1.0001f, 0.5f);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, clk, n, 1.0001f, 0.5f);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, clk, n, 1.0001f, 0.5f);
                hipEventRecord(e1, 0);
                CK(hipDeviceSynchronize()); hipEventElapsedTime(&ms, e0, e1);
            }
            long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            const double fl = (double)n * 64;          // complex (2-float) fma per lane per wave
            printf("%-26s %d WG/CU: %6.2f clk64 cycles per 2-float fma per wave, %6.2f ns wall; kernel %.3f ms -> %.1f TFLOP/s\n", names[mode],
                   blocks / 256, (double)h[0] / fl, (double)h[1] * 10.0 / fl, ms, (double)blocks * 4 * fl * 256 / (ms * 1e9));
        }
    return 0;
}
