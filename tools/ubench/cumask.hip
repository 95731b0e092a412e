// micro-benchmark: does hipExtStreamCreateWithCUMask work here, and which physical CUs (XCC, SE, CU) does mask bit i select?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
#include <map>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void who(unsigned *out) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the block alive a little so that blocks spread over every CU the mask allows
    long long t0 = wall_clock64(); while (wall_clock64() - t0 < 20000) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("CUs %d\n", p.multiProcessorCount);
    const int nb = 4096; unsigned *d; CK(hipMalloc(&d, nb * 8)); std::vector<unsigned> h(nb * 2);
    std::vector<std::vector<unsigned>> masks;
    { std::vector<unsigned> m(8, 0); m[0] = 0xff; masks.push_back(m); }              // bits 0-7
    { std::vector<unsigned> m(8, 0); m[0] = 0xff00; masks.push_back(m); }            // bits 8-15
    { std::vector<unsigned> m(8, 0); m[0] = 0xffffffffu; masks.push_back(m); }       // bits 0-31
    { std::vector<unsigned> m(8, 0); m[7] = 0xffffffffu; masks.push_back(m); }       // bits 224-255
    { std::vector<unsigned> m(8, 0xffffffffu); m[0] = 0; masks.push_back(m); }       // all but 0-31
    for (auto &m : masks) {
        hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data());
        if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e)); return 0; }
        hipLaunchKernelGGL(who, dim3(nb), dim3(64), 0, s, d);
        CK(hipStreamSynchronize(s)); CK(hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost));
        std::map<unsigned, std::set<unsigned>> per_xcc;
        for (int b = 0; b < nb; b++) {
            const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
        }
        printf("mask %08x %08x .. %08x:", m[0], m[1], m[7]);
        int tot = 0;
        for (auto &kv : per_xcc) { printf("  xcc%u:%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
        printf("  -> %d distinct CUs\n", tot);
        CK(hipStreamDestroy(s));
    }
    return 0;
}
