// What does a VALU-only wave get on a SIMD whose other wave streams v_mfma_f32_16x16x4_f32?  One 512-thread block per CU: waves 0-3
// run the matrix loop, waves 4-7 (same SIMDs) a loop of independent scalar v_fma_f32 / of LDS reads; each role alone and together.
// hipcc --offload-arch=gfx950 -O3 mfma_valu_share.hip -o mfma_valu_share
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int it_m, int it_v, int vmode, float s) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = (float)i * s;
    __syncthreads();
    float r = 0;
    const unsigned long long t0 = clock64();
    if (wave < 4) {
        v4f acc[2];
        acc[0] = acc[1] = (v4f){0.f, 0.f, 0.f, 0.f};
#ifdef BF16
        typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
        bf8 a, b;
        for (int i = 0; i < 8; i++) { a[i] = (__bf16)((float)threadIdx.x * s + i); b[i] = (__bf16)(s * i); }
        for (int it = 0; it < it_m; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 1], 0, 0, 0);
        }
#else
        float a = (float)threadIdx.x * s, b = s;
        for (int it = 0; it < it_m; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 1], 0, 0, 0);
        }
#endif
        r = acc[0].x + acc[1].y;
    } else if (vmode == 0) {
        float a[16];
        for (int i = 0; i < 16; i++) a[i] = (float)threadIdx.x + i;
        for (int it = 0; it < it_v; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) a[i] = fmaf(a[i], s, 1e-9f);
        }
        for (int i = 0; i < 16; i++) r += a[i];
    } else if (vmode == 1) {             // dependent chain of scalar fma
        float a = (float)threadIdx.x;
        for (int it = 0; it < it_v; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) a = fmaf(a, s, 1e-9f);
        }
        r = a;
    } else {                             // LDS reads (b128, conflict-free), 16 per iteration
        const float4 *p = reinterpret_cast<const float4 *>(lds) + lane;
        float4 acc = make_float4(0, 0, 0, 0);
        for (int it = 0; it < it_v; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) { const float4 v = p[64 * (i & 7) + ((it & 3) << 9)]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        }
        r = acc.x + acc.y + acc.z + acc.w;
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount;
    float *out; unsigned long long *cyc, h[8 * 512];
    hipMalloc(&out, sizeof(float) * blocks * 512); hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int IM = 20000, IV = 20000;
    for (int vmode = 0; vmode < 3; vmode++)
        for (int cfg = 0; cfg < 3; cfg++) {            // 0: matrix only, 1: the other role only, 2: both
            const int im = cfg == 1 ? 0 : IM, iv = cfg == 0 ? 0 : IV;
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, cyc, im, iv, vmode, 0.999f); hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            hipMemcpy(h, cyc, sizeof(unsigned long long) * 8 * 16, hipMemcpyDeviceToHost);
            double cm = 0, cv = 0; for (int b = 0; b < 16; b++) for (int w = 0; w < 4; w++) { cm += h[b * 8 + w]; cv += h[b * 8 + 4 + w]; }
            cm /= 64; cv /= 64;
            printf("other role %s, %s: %.3f ms; matrix wave %.1f cycles per MFMA, other wave %.2f cycles per op\n",
                   vmode == 0 ? "16 independent v_fma_f32" : vmode == 1 ? "dependent v_fma_f32 chain" : "ds_read_b128 + 4 adds",
                   cfg == 0 ? "matrix only" : cfg == 1 ? "other only" : "both", ms, im ? cm / (im * 16.0) : 0.0, iv ? cv / (iv * 16.0) : 0.0);
        }
    return 0;
}
