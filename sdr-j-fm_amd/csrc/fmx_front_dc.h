// fmx_front_dc.h -- pieces the input-FIR kernels (fmx_front.hip, fmx_front4.hip) and front_pre_kernel share: the RF DC recurrence of
// fm-processor.cpp:423-446 over one tile as an affine map (per-lane run, DPP wave scan), and the mailbox counters between the waves of a
// workgroup.  One definition, so that every kernel that walks a stream's tiles gets the same values bit for bit.
#pragma once
#include "fmx_internal.h"

namespace fmx {

typedef float v2f __attribute__((ext_vector_type(2)));

// The DC recurrence r <- r + alpha (x - r) over a run of samples is the affine map r -> r (1 - u) + a.
// (u, a) are kept instead of (m = 1 - u, a): u ~ count * alpha is tiny, so f32 holds it to 1e-7 relative,
// whereas 1 - alpha itself is not representable to better than 7 % of alpha in f32.
struct Aff { float u, ar, ai; };
__device__ __forceinline__ Aff aff_then(const Aff &f, const Aff &g) {      // apply f, then g
    Aff o;
    o.u = f.u + g.u - f.u * g.u;
    o.ar = f.ar + g.ar - f.ar * g.u;
    o.ai = f.ai + g.ai - f.ai * g.u;
    return o;
}

// one tile's map applied to the state in front of it (the same expression wherever the chain is walked: the workgroup that owns the tile,
// and the workgroups of later parts of a channel that is split in time, which walk the maps of the tiles in front of their own)
__device__ __forceinline__ float dc_chain(float c, float tu, float ta) { return fmaf(-c, tu, c) + ta; }

#ifndef FMX_WAVE_SHR
#define FMX_WAVE_SHR 1   /* 0: ds_bpermute (__shfl_up) for the one-lane shift of the scan (A/B builds) */
#endif
// constants of the wave scan of full tiles: every lane's run of 24 samples has the same u, so the scan of u is known in advance -- u_exc = u of
// `lane` runs, u_tile = u of 64 runs -- and only the `a` parts are scanned: a <- a + a_earlier * m with m = (1 - u)^(runs the lane's partial
// result covers), a constant per scan step (m1, m2, m4, m8) or per lane (mA, mB)
struct DcK { float alpha, u_full, m1, m2, m4, m8, u_exc, u_tile, mA, mB; };
__device__ __forceinline__ DcK dc_consts(float alpha, int lane) {
    DcK K; K.alpha = alpha;
    float u_full = 0.f;                               // u of a full 24-sample run (the same for every such lane)
    for (int k = 0; k < 2 * DECIM; k++) u_full = (1.0f - u_full) * alpha + u_full;
    K.u_full = u_full;
    K.m1 = 1.0f - u_full; K.m2 = K.m1 * K.m1; K.m4 = K.m2 * K.m2; K.m8 = K.m4 * K.m4;
    float u_exc = 0.f, u_tile = 0.f, mA = 1.f, mB = 1.f;
    for (int i = 0; i < 64; i++) {
        if (i < lane) u_exc = u_exc + u_full - u_exc * u_full;
        u_tile = u_tile + u_full - u_tile * u_full;
        if (i < (lane & 15) + 1) mA *= K.m1;
        if (i < (lane & 31) + 1) mB *= K.m1;
    }
    K.u_exc = u_exc; K.u_tile = u_tile; K.mA = mA; K.mB = mB;
    return K;
}
// The RF DC recurrence (fm-processor.cpp:423-446) over one tile as an affine map: the lane's run over its 24 samples x[first .. lastp1), the wave
// scan of the runs; pre = the map of the lanes in front of this one, (tu, tar, tai) = the whole tile's, sA = the sum of the lane's first column
// (channels without an LO, full tiles).  One function for the kernel and for front_pre_kernel, which tabulates the maps of a stream's tiles.
struct DcMap { Aff pre; float tu, tar, tai; v2f sA; };
__device__ __forceinline__ DcMap dc_tile_map(const v2f *x, int first, int lastp1, bool wave_full, bool fast, const DcK &K, int lane) {
    constexpr int SPT_ = 2 * DECIM;
    DcMap M;
    const float alpha = K.alpha, u_full = K.u_full;
    Aff a; a.u = 0.f;
    v2f aa = (v2f){0.f, 0.f};
    const v2f al = (v2f){alpha, alpha};
    v2f sA = (v2f){0.f, 0.f};                 // fast path: sum of the lane's first column
    if (wave_full && fast) {
        v2f t4[4];
#pragma unroll
        for (int j = 0; j < 4; j++) t4[j] = (x[6 * j] + x[6 * j + 1]) + (x[6 * j + 2] + x[6 * j + 3]) + (x[6 * j + 4] + x[6 * j + 5]);
        sA = t4[0] + t4[1];
        aa = al * (sA + (t4[2] + t4[3]));
        a.u = u_full;
    } else if (wave_full) {
#pragma unroll
        for (int k = 0; k < SPT_; k++) aa = __builtin_elementwise_fma(x[k] - aa, al, aa);
        a.u = u_full;
    } else {
#pragma unroll
        for (int k = 0; k < SPT_; k++) {
            if (k >= first && k < lastp1) {
                a.u = (1.0f - a.u) * alpha + a.u;
                aa = __builtin_elementwise_fma(x[k] - aa, al, aa);
            }
        }
    }
    Aff pre;                                  // exclusive prefix within the tile
    float tu, tar, tai;                       // the whole tile's map
    if (wave_full) {
        // inclusive scan of the a parts with DPP: four steps inside the 16-lane rows, then the row totals
        // ride row_bcast:15 (into rows 1, 3) and row_bcast:31 (into rows 2, 3)
        float sr = aa.x, si = aa.y;
#define FMX_SCAN_STEP(ctrl, rmask, mm)                                                                                       \
        {                                                                                                        \
            const float er = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sr), ctrl, rmask, 0xf, false)); \
            const float ei = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(si), ctrl, rmask, 0xf, false)); \
            sr = fmaf(er, mm, sr); si = fmaf(ei, mm, si);                                                        \
        }
        FMX_SCAN_STEP(0x111, 0xf, K.m1)
        FMX_SCAN_STEP(0x112, 0xf, K.m2)
        FMX_SCAN_STEP(0x114, 0xf, K.m4)
        FMX_SCAN_STEP(0x118, 0xf, K.m8)
        FMX_SCAN_STEP(0x142, 0xa, K.mA)
        FMX_SCAN_STEP(0x143, 0xc, K.mB)
#undef FMX_SCAN_STEP
#if FMX_WAVE_SHR
        // exclusive prefix = the inclusive one of the lane to the left: wave_shr:1 (DPP, lane 0 gets the zero of `old`)
        pre.ar = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sr), 0x138, 0xf, 0xf, false));
        pre.ai = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(si), 0x138, 0xf, 0xf, false));
#else
        pre.ar = __shfl_up(sr, 1, 64); pre.ai = __shfl_up(si, 1, 64);
        if (lane == 0) { pre.ar = 0.f; pre.ai = 0.f; }
#endif
        pre.u = K.u_exc;
        tu = K.u_tile;
        tar = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sr), 63));
        tai = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(si), 63));
    } else {
        a.ar = aa.x; a.ai = aa.y;
        Aff inc = a;                          // general inclusive scan (first / last tile of a call)
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            Aff o; o.u = __shfl_up(inc.u, d, 64); o.ar = __shfl_up(inc.ar, d, 64); o.ai = __shfl_up(inc.ai, d, 64);
            if (lane >= d) inc = aff_then(o, inc);
        }
        pre.u = __shfl_up(inc.u, 1, 64); pre.ar = __shfl_up(inc.ar, 1, 64); pre.ai = __shfl_up(inc.ai, 1, 64);
        if (lane == 0) { pre.u = 0.f; pre.ar = 0.f; pre.ai = 0.f; }
        tu = __shfl(inc.u, 63, 64); tar = __shfl(inc.ar, 63, 64); tai = __shfl(inc.ai, 63, 64);
    }
    M.pre = pre; M.tu = tu; M.tar = tar; M.tai = tai; M.sA = sA;
    return M;
}

// mailbox counters between the waves of a workgroup (LDS, workgroup scope)
__device__ __forceinline__ void seq_wait(int *p, int need) {
    while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void seq_post(int *p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace fmx
