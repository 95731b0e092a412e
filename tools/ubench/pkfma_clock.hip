// Effective engine clock under packed-FMA load: every SIMD issues one v_pk_fma_f32 per 4 cycles at best, so
// clock >= (pk_fma per wave x 4 x waves per SIMD) / time.  hipcc --offload-arch=gfx950 -O3 pkfma_clock.hip -o pkfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k(float *out, int iters, float s) {
    v2f a[16];
    for (int i = 0; i < 16; i++) a[i] = (v2f){(float)threadIdx.x + i, 1.f};
    const v2f w = (v2f){s, s}, b = (v2f){1e-9f, 1e-9f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) a[i] = __builtin_elementwise_fma(a[i], w, b);
    }
    float r = 0; for (int i = 0; i < 16; i++) r += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount, wps = 2;                 // waves per SIMD
    const int blocks = ncu * wps; const int iters = 200000;
    float *out; hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double cyc = (double)iters * 16 * 4 * wps;            // SIMD cycles at one pk_fma per 4 cycles
        printf("CUs %d  %.3f ms  -> effective clock %.3f GHz (nominal %.3f)  %.1f TFLOP/s fp32\n", ncu, ms, cyc / (ms * 1e6), p.clockRate / 1e6,
               (double)blocks * 256 * iters * 16 * 4 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
