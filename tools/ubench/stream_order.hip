// Does a kernel of a stream start before the previous kernel of the same stream has finished?  (It must not: libfmx relies on stream
// order between its stages.)  Kernel A: workgroup i spins i microseconds, then writes flag[i]; kernel B: workgroup i reads the flag of
// workgroup (i + N / 2) % N.  Also timed: A, B back to back against A alone + B alone.   hipcc --offload-arch=gfx950 -O2 stream_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void ka(int *flag, int n) {
    const unsigned long long t0 = wall_clock64();
    const unsigned long long ticks = (unsigned long long)blockIdx.x * 100 * 200 / n;      // wall clock = 100 MHz: up to 200 us for the last workgroup
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) flag[blockIdx.x] = 1;
}
__global__ void kb(const int *flag, int *seen, int n) {
    if (threadIdx.x == 0) seen[blockIdx.x] = flag[(blockIdx.x + n / 2) % n];
}
int main() {
    const int n = 4096;
    int *flag, *seen, h[4096];
    hipMalloc(&flag, sizeof(int) * n); hipMalloc(&seen, sizeof(int) * n);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int bad = 0;
    for (int rep = 0; rep < 20; rep++) {
        hipMemsetAsync(flag, 0, sizeof(int) * n, s); hipMemsetAsync(seen, 0xff, sizeof(int) * n, s);
        hipLaunchKernelGGL(ka, dim3(n), dim3(256), 0, s, flag, n);
        hipLaunchKernelGGL(kb, dim3(n), dim3(256), 0, s, flag, seen, n);
        hipStreamSynchronize(s);
        hipMemcpy(h, seen, sizeof(int) * n, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; i++) bad += h[i] != 1;
    }
    printf("workgroups of the second kernel that ran before their producer in the first had finished: %d of %d\n", bad, 20 * n);
    return bad != 0;
}
