// micro-benchmark: dependent-issue latency of VALU / LDS operations for a single wave per SIMD on gfx950
// (what bounds the lane-per-channel recurrence kernels of stage B)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, long long *clk, int n, float a, double da) {
    __shared__ double tab[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) tab[i] = (double)((i * 7 + 1) & 1023);
    __syncthreads();
    float y = threadIdx.x * 1e-3f; double d = threadIdx.x * 1e-3; int idx = threadIdx.x;
    long long t0 = clock64();
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int u = 0; u < 32; u++) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(y) : "v"(a));
            else if (MODE == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(y) : "v"(a));
            else if (MODE == 2) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d) : "v"(da));
            else if (MODE == 3) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(y)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(y) : "v"(d)); }
            else if (MODE == 4) { idx = (int)tab[idx & 1023]; }                       // LDS pointer chase (b64 read + cvt)
            else if (MODE == 5) { asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(y) : "v"(a), "v"(y) : "vcc"); }
            else if (MODE == 6) { asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(y) : "v"(a), "v"(-a)); }
            else if (MODE == 7) { asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(idx) : "v"(d)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d) : "v"(idx)); }
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = y + (float)d + idx;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
int main() {
    float *out; long long *clk; const int n = 2000;
    CK(hipMalloc(&out, 4 * 64 * 64)); CK(hipMalloc(&clk, 8 * 64));
    const char *names[] = {"v_fma_f32 dependent", "v_add_f32 dependent", "v_fma_f64 dependent", "cvt f32->f64->f32 (2 ops)",
                           "LDS read b64 + cvt chase", "v_cmp + v_cndmask (2 ops)", "v_med3_f32 dependent", "cvt f64->i32->f64 (2 ops)"};
    for (int mode = 0; mode < 8; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            switch (mode) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(8), dim3(64), 0, 0, out, clk, n, 0.999f, 0.999); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(8), dim3(64), 0, 0, out, clk, n, 0.999f, 0.999); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(8), dim3(64), 0, 0, out, clk, n, 0.999f, 0.999); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(8), dim3(64), 0, 0, out, clk, n, 0.999f, 0.999); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(8), dim3(64), 0, 0, out, clk, n, 0.999f, 0.999); break;
            case 5: hipLaunchKernelGGL(k<5>, dim3(8), dim3(64), 0, 0, out, clk, n, 0.999f, 0.999); break;
            case 6: hipLaunchKernelGGL(k<6>, dim3(8), dim3(64), 0, 0, out, clk, n, 0.999f, 0.999); break;
            case 7: hipLaunchKernelGGL(k<7>, dim3(8), dim3(64), 0, 0, out, clk, n, 0.999f, 0.999); break;
            }
            CK(hipDeviceSynchronize());
        }
        long long h; CK(hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost));
        printf("%-32s %7.2f cycles per unrolled step\n", names[mode], (double)h / (n * 32.0));
    }
    return 0;
}
