// accuracy of the hardware sine / cosine (v_sin_f32 / v_cos_f32, input in turns) on the 192000 SinCos table angles (sincos.cpp:45-54)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(float *s, float *c, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float t = (float)i * (1.0f / 192000.0f);
    s[i] = __builtin_amdgcn_sinf(t); c[i] = __builtin_amdgcn_cosf(t);
    if (n < 0) {   // (variant: quadrant folded)
    }
}
__global__ void kf(float *s, float *c, int n) {
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int q = (int)(((unsigned)(idx >> 7) * 2797u) >> 20);          // idx / 48000
    const int r = idx - q * 48000;
    const int rr = (q & 1) ? 48000 - r : r;
    const float t = (float)rr * (1.0f / 192000.0f);                      // <= 0.25 turns
    const float sv = __builtin_amdgcn_sinf(t), cv = __builtin_amdgcn_cosf(t);
    // angle = q * 90 deg + r: sin = {sv, cv', ...}: with the folding rr the sine of the angle is +-sin(rr) , the cosine +-cos(rr)
    s[idx] = (q & 2) ? -sv : sv;
    c[idx] = ((q + 1) & 2) ? -cv : cv;
}
int main() {
    const int n = 192000;
    float *ds, *dc; hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, ds, dc, n);
    std::vector<float> s(n), c(n);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double ws = 0, wc = 0, rs = 0; int is = 0, ic = 0;
    for (int i = 0; i < n; i++) {
        const double a = 2.0 * M_PI * i / 192000.0;
        const double es = std::fabs((double)s[i] - (double)(float)std::sin(a)), ec = std::fabs((double)c[i] - (double)(float)std::cos(a));
        if (es > ws) { ws = es; is = i; }
        if (ec > wc) { wc = ec; ic = i; }
        rs += es * es;
    }
    printf("v_sin_f32 worst |err| %.3e at idx %d, v_cos_f32 worst %.3e at idx %d, rms %.3e\n", ws, is, wc, ic, std::sqrt(rs / n));
    hipLaunchKernelGGL(kf, dim3((n + 255) / 256), dim3(256), 0, 0, ds, dc, n);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    ws = wc = rs = 0;
    for (int i = 0; i < n; i++) {
        const double a = 2.0 * M_PI * i / 192000.0;
        const double es = std::fabs((double)s[i] - (double)(float)std::sin(a)), ec = std::fabs((double)c[i] - (double)(float)std::cos(a));
        if (es > ws) { ws = es; is = i; }
        if (ec > wc) { wc = ec; ic = i; }
        rs += es * es;
    }
    printf("quadrant folded: sin worst %.3e at idx %d, cos worst %.3e at idx %d, rms %.3e\n", ws, is, wc, ic, std::sqrt(rs / n));
    return 0;
}
