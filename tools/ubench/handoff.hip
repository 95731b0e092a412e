// micro-benchmark: latency of an event hand-off between two streams (record on A, wait on B, tiny kernel), for plain
// non-blocking streams and for CU-masked streams (hipExtStreamCreateWithCUMask), with 2 or 8 streams in the ring.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(int *p) { if (threadIdx.x == 0) atomicAdd(p, 1); }
int run(std::vector<hipStream_t> &st, const char *name, int *d) {
    const int n = 400;
    std::vector<hipEvent_t> ev((size_t)n);
    for (size_t i = 0; i < ev.size(); i++) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    for (int rep = 0; rep < 2; rep++) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; i++) {
            hipStream_t a = st[i % st.size()], b = st[(i + 1) % st.size()];
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d);
            CK(hipEventRecord(ev[i], a));
            CK(hipStreamWaitEvent(b, ev[i], 0));
        }
        auto t1 = std::chrono::steady_clock::now();
        CK(hipDeviceSynchronize());
        auto t2 = std::chrono::steady_clock::now();
        if (rep == 1) printf("%-46s %zu streams: enqueue %.1f us, end-to-end %.1f us per hand-off\n", name, st.size(),
                             std::chrono::duration<double, std::micro>(t1 - t0).count() / n, std::chrono::duration<double, std::micro>(t2 - t0).count() / n);
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return 0;
}
int main() {
    int *d; CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
    for (int ns : {2, 8}) {
        std::vector<hipStream_t> plain(ns), masked(ns), maskedsame(ns);
        for (auto &s : plain) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        run(plain, "plain non-blocking streams", d);
        std::vector<uint32_t> m1(8, 0u), m2(8, 0xffffffffu); m1[0] = 0xffffffffu; m1[1] = 0xffffffffu; m2[0] = 0; m2[1] = 0;
        for (int i = 0; i < ns; i++) CK(hipExtStreamCreateWithCUMask(&masked[i], 8, (i & 1) ? m2.data() : m1.data()));
        run(masked, "CU-masked streams (alternating masks)", d);
        for (int i = 0; i < ns; i++) CK(hipExtStreamCreateWithCUMask(&maskedsame[i], 8, m1.data()));
        run(maskedsame, "CU-masked streams (same mask)", d);
        std::vector<hipStream_t> mixed;
        for (int i = 0; i < ns; i++) mixed.push_back((i & 1) ? masked[i] : plain[i]);
        run(mixed, "plain <-> CU-masked", d);
        for (auto &s : plain) (void)hipStreamDestroy(s);
        for (auto &s : masked) (void)hipStreamDestroy(s);
        for (auto &s : maskedsame) (void)hipStreamDestroy(s);
    }
    return 0;
}
