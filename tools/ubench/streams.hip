// micro-benchmark: HBM read bandwidth of the front kernel's ACCESS PATTERN -- S independent streams (one persistent
// workgroup each, 1.8 MB apart), W waves per workgroup taking TILE-byte tiles round-robin, each wave keeping DEPTH tiles
// of loads in flight -- with no compute at all.  Compare with the grid-stride probe (fmx_debug_stream_bandwidth).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// KB16: 16-byte units per lane per tile (12 -> 12 KB per wave tile)
template <int K16, int DEPTH>
__global__ __launch_bounds__(256) void k(const f4 *__restrict__ src, float2 *__restrict__ dst, size_t stream16, int ntiles, int waves) {
    extern __shared__ float pad[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f4 *p = src + (size_t)blockIdx.x * stream16;
    f4 buf[DEPTH][K16];
    f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
        const int ti = wave + d * waves;
        if (ti < ntiles)
#pragma unroll
            for (int k = 0; k < K16; k++) buf[d][k] = __builtin_nontemporal_load(p + (size_t)ti * K16 * 64 + k * 64 + lane);
    }
    for (int ti = wave; ti < ntiles; ti += waves * DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const int tj = ti + d * waves;
            if (tj >= ntiles) break;
#pragma unroll
            for (int k = 0; k < K16; k++) acc += buf[d][k];
            const int tn = tj + DEPTH * waves;
            if (tn < ntiles)
#pragma unroll
                for (int k = 0; k < K16; k++) buf[d][k] = __builtin_nontemporal_load(p + (size_t)tn * K16 * 64 + k * 64 + lane);
            dst[((size_t)blockIdx.x * ntiles + tj) * 64 + lane] = make_float2(acc.x + acc.z, acc.y + acc.w);
        }
    }
    if (pad[0] == 123.f) dst[0] = make_float2(pad[1], 0);
}

// lane-contiguous variant: lane l of a wave tile reads ITS OWN K16 consecutive 16-byte units (what a per-lane run of 24
// samples needs), i.e. every load instruction touches 64 different 128-byte lines, each line serving 8 instructions
template <int K16>
__global__ __launch_bounds__(256) void klane(const f4 *__restrict__ src, float2 *__restrict__ dst, size_t stream16, int ntiles, int waves) {
    extern __shared__ float pad[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f4 *p = src + (size_t)blockIdx.x * stream16;
    f4 buf[K16];
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    if (wave < ntiles)
#pragma unroll
        for (int k = 0; k < K16; k++) buf[k] = p[(size_t)wave * K16 * 64 + lane * K16 + k];
    for (int ti = wave; ti < ntiles; ti += waves) {
#pragma unroll
        for (int k = 0; k < K16; k++) acc += buf[k];
        const int tn = ti + waves;
        if (tn < ntiles)
#pragma unroll
            for (int k = 0; k < K16; k++) buf[k] = p[(size_t)tn * K16 * 64 + lane * K16 + k];
        dst[((size_t)blockIdx.x * ntiles + ti) * 64 + lane] = make_float2(acc.x + acc.z, acc.y + acc.w);
    }
    if (pad[0] == 123.f) dst[0] = make_float2(pad[1], 0);
}
template <int K16>
int runlane(const f4 *src, float2 *dst, int streams, size_t stream_bytes, int lds, const char *name) {
    const int ntiles = (int)(stream_bytes / (K16 * 1024));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void *)klane<K16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((klane<K16>), dim3(streams), dim3(256), lds, 0, src, dst, stream_bytes / 16, ntiles, 4);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double bytes = (double)streams * ntiles * K16 * 1024;
    printf("%-44s streams %5d threads 256 lds %6d: %.3f ms  %.0f GB/s read\n", name, streams, lds, best, bytes / best * 1e-6);
    return 0;
}

template <int K16, int DEPTH>
int run(const f4 *src, float2 *dst, int streams, size_t stream_bytes, int threads, int lds, const char *name) {
    const int waves = threads / 64;
    const int ntiles = (int)(stream_bytes / (K16 * 1024));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)k<K16, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<K16, DEPTH>), dim3(streams), dim3(threads), lds, 0, src, dst, stream_bytes / 16, ntiles, waves);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double bytes = (double)streams * ntiles * K16 * 1024;
    printf("%-44s streams %5d threads %3d lds %6d: %.3f ms  %.0f GB/s read\n", name, streams, threads, lds, best, bytes / best * 1e-6);
    return 0;
}

int main(int argc, char **argv) {
    const size_t stream_bytes = 230400 * 8;
    const int maxs = 4096;
    f4 *src; float2 *dst;
    CK(hipMalloc(&src, stream_bytes * maxs)); CK(hipMalloc(&dst, (size_t)maxs * 160 * 64 * 8 * 4));
    CK(hipMemset(src, 0, stream_bytes * maxs));
    for (int streams : {512, 1024, 4096}) {
        run<12, 1>(src, dst, streams, stream_bytes, 256, 72 * 1024, "12 KB tiles, 1 in flight, 2 wg/CU (front)");
        run<12, 1>(src, dst, streams, stream_bytes, 256, 36 * 1024, "12 KB tiles, 1 in flight, 4 wg/CU");
        run<12, 2>(src, dst, streams, stream_bytes, 256, 72 * 1024, "12 KB tiles, 2 in flight, 2 wg/CU");
        run<6, 1>(src, dst, streams, stream_bytes, 256, 72 * 1024, "6 KB tiles, 1 in flight, 2 wg/CU");
        run<6, 2>(src, dst, streams, stream_bytes, 256, 72 * 1024, "6 KB tiles, 2 in flight, 2 wg/CU");
        run<6, 4>(src, dst, streams, stream_bytes, 256, 72 * 1024, "6 KB tiles, 4 in flight, 2 wg/CU");
        runlane<12>(src, dst, streams, stream_bytes, 72 * 1024, "12 KB tiles, lane-contiguous 192 B, 2 wg/CU");
        run<4, 3>(src, dst, streams, stream_bytes, 256, 72 * 1024, "4 KB tiles, 3 in flight, 2 wg/CU");
    }
    return 0;
}
