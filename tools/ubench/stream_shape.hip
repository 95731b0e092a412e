// What does a store cost a streaming kernel with few waves per CU?  One workgroup per CU (the LDS request sees to it), WPC waves each, every wave a loop
// over 12 KB chunks: 12 x 1 KB loads (16 B per lane, nontemporal), wait, sum, and 1 KB stored per chunk (stage A's read-12-write-1 shape).  Patterns:
//   linear   : the chip's waves sweep the buffer side by side (wave w of W takes chunks w, w + W, ...)
//   streams  : every group of `gw` waves walks a stream of its own (230400 float2 apart), its waves taking the stream's chunks in turn -- stage A's shape
// Store forms: none / every chunk / the results of K chunks kept in registers and stored together every K-th chunk (same bytes, 1 / K of the waits that cover
// a store) / AHEAD: the next chunk's loads are issued BEFORE this chunk's store, so that the wait for them need not cover it.
// Vector memory operations share one in-order counter (vmcnt): a wave that waits for its newest loads also waits for every store issued in front of them.
// hipcc --offload-arch=gfx950 -O3 stream_shape.hip -o stream_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int LOADS = 12;
// K = 0: no stores; K >= 1: store every K-th chunk; AHEAD: two chunks per trip, the second one's loads in flight over the first one's store
template <int K, bool AHEAD>
__global__ void k(const v4f *__restrict__ src, float2 *__restrict__ dst, int pattern, int gw, size_t stream_v4, int iters, int wpc) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t W = (size_t)gridDim.x * wpc, w = (size_t)blockIdx.x * wpc + wave;
    // (addresses: a base and a step per wave, no divisions in the loop)
    const unsigned g = (unsigned)w / (unsigned)gw, m = (unsigned)w % (unsigned)gw;
    const v4f *const in0 = pattern == 0 ? src + w * LOADS * 64 + lane : src + (size_t)g * stream_v4 + (size_t)m * LOADS * 64 + lane;
    const size_t in_step = (pattern == 0 ? W : (size_t)gw) * LOADS * 64;
    float2 *const out0 = pattern == 0 ? dst + w * 128 + 2 * lane : dst + (size_t)g * (stream_v4 / 6) + (size_t)m * 128 + 2 * lane;
    const size_t out_step = (pattern == 0 ? W : (size_t)gw) * 128;
    auto in_at = [&](int it) -> const v4f * { return in0 + (size_t)it * in_step; };
    auto out_at = [&](int it) -> v4f * { return reinterpret_cast<v4f *>(out0 + (size_t)it * out_step); };
    if (AHEAD) {
        v4f b0[LOADS], b1[LOADS];
#pragma unroll
        for (int l = 0; l < LOADS; l++) b0[l] = __builtin_nontemporal_load(in_at(0) + l * 64);
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int l = 0; l < LOADS; l++) b1[l] = __builtin_nontemporal_load(in_at(it + 1 < iters ? it + 1 : it) + l * 64);
            v4f a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int l = 0; l < LOADS; l++) a += b0[l];
            *out_at(it) = a;
#pragma unroll
            for (int l = 0; l < LOADS; l++) b0[l] = __builtin_nontemporal_load(in_at(it + 2 < iters ? it + 2 : it) + l * 64);
            a = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int l = 0; l < LOADS; l++) a += b1[l];
            if (it + 1 < iters) *out_at(it + 1) = a;
        }
        return;
    }
    v4f keep[K > 0 ? K : 1];
    for (int it0 = 0; it0 < iters; it0 += (K > 0 ? K : 1)) {
#pragma unroll
        for (int j = 0; j < (K > 0 ? K : 1); j++) {
            const int it = it0 + j < iters ? it0 + j : iters - 1;
            v4f a = {0.f, 0.f, 0.f, 0.f};
            const v4f *p = in_at(it);
#pragma unroll
            for (int l = 0; l < LOADS; l++) a += __builtin_nontemporal_load(p + l * 64);
            keep[j] = a;
            if (K == 0 && a.x == 1.2345e33f) dst[0] = make_float2(a.y, a.z);
        }
        if (K > 0) {
#pragma unroll
            for (int j = 0; j < K; j++) if (it0 + j < iters) *out_at(it0 + j) = keep[j];
        }
    }
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int ncu0 = p.multiProcessorCount;
    const size_t stream_bytes = 230400ull * 8, total = 4096ull * stream_bytes;
    v4f *src; float2 *dst; (void)hipMalloc(&src, total); (void)hipMalloc(&dst, total / 6 + (1 << 20));
    (void)hipMemset(src, 0, total);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    typedef void (*kfn)(const v4f *, float2 *, int, int, size_t, int, int);
    auto run = [&](const char *name, kfn kf, int wpc, int pattern, int gw, int wgs = 1) {
        const int ncu = ncu0 * wgs; const size_t ldsb = wgs == 1 ? 100 * 1024 : 0;
        const size_t chunk_b = (size_t)LOADS * 1024;
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0);
            size_t done = 0;
            if (pattern == 0) {
                const int iters = (int)(total / chunk_b / ((size_t)ncu * wpc));
                hipLaunchKernelGGL(kf, dim3(ncu), dim3(64 * wpc), ldsb, 0, src, dst, 0, gw, stream_bytes / 16, iters, wpc);
                done = (size_t)iters * chunk_b * ncu * wpc;
            } else {
                const int groups = ncu * wpc / gw, iters = (int)(stream_bytes / chunk_b / gw);
                for (int r = 0; r < 4096 / groups; r++) {
                    hipLaunchKernelGGL(kf, dim3(ncu), dim3(64 * wpc), ldsb, 0, src + (size_t)r * groups * (stream_bytes / 16), dst + (size_t)r * groups * (stream_bytes / 16 / 6) * 2, 1, gw, stream_bytes / 16, iters, wpc);
                    done += (size_t)iters * chunk_b * gw * groups;
                }
            }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const float per = ms * (float)((double)total / (double)done);
            if (per < best) best = per;
        }
        printf("waves/CU %3d  %-7s  %-34s : %.3f ms per 7.55 GB read  (%.2f TB/s read + write)\n", wpc * wgs, pattern ? "streams" : "linear", name, best, total * (kf == (kfn)k<0, false> ? 1.0 : 13.0 / 12) / (best * 1e-3) / 1e12);
    };
    for (int wpc = 8; wpc <= 16; wpc += 4)
        for (int pat = 0; pat < 2; pat++) {
            const int gw = wpc / 2;
            run("no stores", k<0, false>, wpc, pat, gw);
            run("a store per chunk", k<1, false>, wpc, pat, gw);
            run("4 chunks' stores together", k<4, false>, wpc, pat, gw);
            run("next chunk's loads ahead of the store", k<1, true>, wpc, pat, gw);
        }
    // many small workgroups per CU (256 threads, no LDS): 16 .. 64 waves per CU
    for (int wgs = 4; wgs <= 16; wgs *= 2) {
        run("no stores (256-thread workgroups)", k<0, false>, 4, 0, 4, wgs);
        run("a store per chunk (256-thread workgroups)", k<1, false>, 4, 0, 4, wgs);
    }
    return 0;
}
