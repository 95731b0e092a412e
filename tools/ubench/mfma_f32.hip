// Sustained rate of v_mfma_f32_16x16x4_f32 (the f32-input matrix instruction stage A's FIR uses), alone and next to
// packed-FMA waves on the same SIMDs.  hipcc --offload-arch=gfx950 -O3 mfma_f32.hip -o mfma_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
// mode 0: every wave MFMA only; mode 1: odd waves of a SIMD pair run packed FMAs instead
__global__ __launch_bounds__(256) void k(float *out, int iters, float s, int mode) {
    const int wave = threadIdx.x >> 6;
    float r = 0;
    if (mode == 0 || (blockIdx.x & 1) == 0) {
        v4f acc[4];
        for (int i = 0; i < 4; i++) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
        float a = (float)threadIdx.x * s, b = s;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 3], 0, 0, 0);
        }
        for (int i = 0; i < 4; i++) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    } else {
        v2f a[16];
        for (int i = 0; i < 16; i++) a[i] = (v2f){(float)threadIdx.x + i, 1.f};
        const v2f w = (v2f){s, s}, b = (v2f){1e-9f, 1e-9f};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = __builtin_elementwise_fma(a[i], w, b);
        }
        for (int i = 0; i < 16; i++) r += a[i].x + a[i].y;
    }
    out[blockIdx.x * 256 + threadIdx.x] = r + wave;
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount, wps = 2;
    const int blocks = ncu * wps;
    float *out; hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; mode++)
        for (int iters : {4000, 40000, 400000}) {
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f, mode); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double mf_waves = (mode == 0) ? blocks * 4.0 : blocks * 2.0;
                const double tf = mf_waves * iters * 16 * 2048.0 / (ms * 1e-3) / 1e12;
                const double pk = (mode == 0) ? 0 : blocks * 2.0 * 64 * iters * 16 * 4 / (ms * 1e-3) / 1e12;
                printf("mode %d iters %6d  %.3f ms  mfma %.1f TFLOP/s  pk_fma beside it %.1f TFLOP/s\n", mode, iters, ms, tf, pk);
            }
        }
    return 0;
}
