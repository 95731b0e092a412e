// audio_mfma_kernel -- stage C's folded 883-tap FIR on the matrix pipe (round 5).  Measured no faster than audio_fft_kernel (0.425 against 0.410 ms at
// 4096 channels) and taken out of the product in round 6; this is the kernel as it stood in csrc/fmx_audio.hip (it used that file's helpers and the tap
// tables fmx_api.hip built into DeviceTables::audio_mtab: [sets][2][2][AM_TAB] f16 halves of g * 2^14, reversed, shifted by the window's parity).  Known
// defect when it left: `ch = blockIdx.y` lacks `+ G.ch0`, so it must not run on the second stage-B / C channel group.
// The same FIR on the MATRIX pipe (round 5; the recipe of fmx_front4.hip).  The fast convolution above is five 2048-point transforms of packed
// f32 FMAs per 1792 frames, and packed FMAs are what runs this GPU into its power limit (93 of the nominal 157 TFLOP/s, tools/ubench/
// pkfma_clock.hip).  Here the folded 883-tap FIR is a Toeplitz product on v_mfma_f32_32x32x16_f16: samples and taps split into two f16 halves
// each (d * 2^10 = dh + dl, g * 2^14 = gh + gl; products of f16 values are exact in the f32 accumulator; what is lost is the rounding of the
// remainders, 2^-22 of a product),
//     out[m0 + i] = sum_k A[i][k] B[k][n],   A[i][k] = g[4 i + nt - 1 - k],   B[k][n] = d[w0 + 128 b + k].comp,   n = 2 b + comp,
// 32 adjacent frames i against the (124 + nt)-sample window of their block b, sixteen blocks x (L, R) = the 32 columns: 512 frames per wave.
// K is TIME, so the LDS image is the de-interleaved d ring of the workgroup's 1024 frames -- four linear f16 planes (hi L, hi R, lo L, lo R) --
// and a lane's operand 16 consecutive bytes of a plane; the taps sit reversed in a table per half.  The 32 x 32 tile because the operands come
// from LDS: 12 multiply-adds per operand byte (the 16 x 16 tile: 6, and the LDS, not the matrix pipe, set the time -- measured, 0.43 ms against
// 0.41 for the fast convolution).  Two waves share a group's 63 K-steps and exchange halves of their sums through LDS (four waves per workgroup).  A block's window begins 128 samples = 256 bytes behind its neighbour's -- sixteen blocks on ONE of the
// sixteen 16-byte bank slots --, so every 256 bytes of a plane are followed by 16 bytes of padding (block stride 17 slots) and the planes of L
// and R sit 8 slots apart: the operand reads are conflict-free.  One 128-thread workgroup per 1024 frames and channel; 63 K-steps of three
// matrix instructions per wave; the per-frame epilogue (gain, fade-in, test tone, peak maxima) is the fast convolution's.
namespace am {
constexpr int WV = 4;                            // waves per workgroup: two per group of 512 frames, each with half of the K-steps
constexpr int FRW = 512;                         // frames per group: 16 blocks of 32
constexpr int FR = 2 * FRW;                      // frames per workgroup
constexpr int KMAX = 1024;                       // >= 124 + 883 + 1, a multiple of 16
constexpr int TAB = AM_TAB;                      // entries of a tap table: u = k - 4 i + 124 in [0, KMAX + 124)
constexpr int PLN = 4 * FR - 128 + KMAX;         // samples per plane: the last block's window ends here (4992)
constexpr int PLB = (PLN * 2 / 256) * 272 + 16;  // bytes per padded plane: 664 slots = 8 (mod 16)
constexpr float DSC = 1024.f, GSC = 16384.f, OSC = 1.0f / (1024.f * 16384.f);
typedef _Float16 h16;
typedef h16 v8h __attribute__((ext_vector_type(8)));
typedef h16 v2h __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
static_assert(C_MAX_TAPS + 125 <= KMAX && KMAX % 16 == 0 && (PLN * 2) % 256 == 0 && (PLB / 16) % 16 == 8, "geometry");
__device__ __forceinline__ int pad256(int byte) { return byte + ((byte >> 8) << 4); }
__device__ __forceinline__ void split2(float a, float b, uint32_t *hi, uint32_t *lo) {
    const h16 ha = (h16)a, hb = (h16)b;
    const h16 la = (h16)(a - (float)ha), lb = (h16)(b - (float)hb);
    *hi = __builtin_bit_cast(uint32_t, (v2h){ha, hb});
    *lo = __builtin_bit_cast(uint32_t, (v2h){la, lb});
}
__device__ __forceinline__ float dpp_swap1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)); }   // quad_perm [1,0,3,2]
}  // namespace am

__global__ __launch_bounds__(64 * am::WV) void audio_mfma_kernel(DeviceTables T, DeviceBuffers B, CallGeom G, float2 *__restrict__ pcm) {
    using namespace am;
    __shared__ __attribute__((aligned(16))) char pl[4 * PLB];             // hi L, hi R, lo L, lo R (padded: pad256)
    __shared__ __attribute__((aligned(16))) h16 ta[2][TAB];
    __shared__ int pkt[FR / C_TILE][4];
    const int ch = blockIdx.y, t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wave = wv >> 1, kh = wv & 1;            // group of 512 frames, half of the K-steps
    const int64_t mb = G.M0 + (int64_t)blockIdx.x * FR;
    if (mb >= G.M1) return;
    const ChanParams P = B.params[ch];
    const AudioSet AS = T.audio_sets[P.audio_set];
    const int nt = AS.ntaps;
    const float2 *__restrict__ dring = B.dring + (size_t)ch * (G.dring_mask + 1);
    if (t < 4 * (FR / C_TILE)) pkt[t >> 2][t & 3] = 0;
    // window: plane entry p <-> d sample w0e + p; w0 = the oldest sample of the workgroup's first frame, w0e = w0 rounded down to even (a lane
    // moves sample PAIRS: 16-byte loads of the ring, 4-byte stores of a plane)
    const int64_t w0 = 4 * mb + 3 - AS.delay - (nt - 1);
    const int sh = (int)(w0 & 1);
    const int64_t w0e = w0 - sh;
    // taps: A[i][k] = rev[k - sh - 4 i]; table entry u = k - 4 i + 124: made on the host (fmx_api.hip ensure_sets), copied here
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(T.audio_mtab + ((size_t)P.audio_set * 2 + sh) * 2 * TAB);
        uint4 *dst = reinterpret_cast<uint4 *>(&ta[0][0]);
        for (int i = t; i < 2 * TAB * 2 / 16; i += 64 * WV) dst[i] = src[i];
    }
    const int nsteps = (((nt + 124 + sh + 15) >> 4) + 3) & ~3;           // K-steps, a multiple of four (the taps behind the filter's last are zeros)
    const int need = 4 * FR - 128 + 16 * nsteps;                          // plane entries the matrix products read
    constexpr int NB = 10;                                                // pairs a thread has in flight (two rounds fill the planes)
    for (int q0 = t; 2 * q0 < need; q0 += NB * 64 * WV) {
        float4 v[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int q = q0 + k * 64 * WV;
            const int64_t s0 = w0e + 2 * q;
            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (s0 >= 0 && 2 * q < need) v[k] = *reinterpret_cast<const float4 *>(&dring[s0 & G.dring_mask]);   // (the ring's size is even, s0 is: the pair does not wrap)
        }
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const int q = q0 + k * 64 * WV;
            if (2 * q < need) {
                uint32_t hl, ll, hr, lr;
                split2(v[k].x * DSC, v[k].z * DSC, &hl, &ll);
                split2(v[k].y * DSC, v[k].w * DSC, &hr, &lr);
                char *w = pl + pad256(4 * q);
                *reinterpret_cast<uint32_t *>(w) = hl; *reinterpret_cast<uint32_t *>(w + PLB) = hr;
                *reinterpret_cast<uint32_t *>(w + 2 * PLB) = ll; *reinterpret_cast<uint32_t *>(w + 3 * PLB) = lr;
            }
        }
    }
    __syncthreads();
    // operands of the 32 x 32 x 16 matrix instruction: the lane has 8 consecutive k (k group kgr = lane / 32) of row / column lane % 32
    const int kgr = lane >> 5, n = lane & 31, blk = n >> 1, comp = n & 1;
    // (a K-step's 32 bytes of a lane begin at 256 (16 wave + blk) + 32 j + 16 kgr: inside one 256-byte run, run 16 wave + blk + j / 8)
    const char *const bB = pl + comp * PLB + 272 * (16 * wave + blk) + 16 * kgr;
    const char *const aB = reinterpret_cast<const char *>(&ta[0][0]) + 16 * kgr + 248 - 8 * n;
    v16f ahh, ahl, alh;
#pragma unroll
    for (int v = 0; v < 16; v++) { ahh[v] = 0.f; ahl[v] = 0.f; alh[v] = 0.f; }
    auto ldB = [&](int j, u32x4 *bh, u32x4 *bl) {
        const int bo = 32 * (j & 7) + 272 * (j >> 3);
        *bh = *reinterpret_cast<const u32x4 *>(bB + bo); *bl = *reinterpret_cast<const u32x4 *>(bB + bo + 2 * PLB);
    };
    auto ldA = [&](int j, u32x4 *ah, u32x4 *al) {
        const u32x2 a0 = *reinterpret_cast<const u32x2 *>(aB + 32 * j), a1 = *reinterpret_cast<const u32x2 *>(aB + 32 * j + 8);
        const u32x2 l0 = *reinterpret_cast<const u32x2 *>(aB + 32 * j + 2 * TAB), l1 = *reinterpret_cast<const u32x2 *>(aB + 32 * j + 2 * TAB + 8);
        *ah = (u32x4){a0.x, a0.y, a1.x, a1.y}; *al = (u32x4){l0.x, l0.y, l1.x, l1.y};
    };
    // operand sets in rotation: a step's operands are requested two steps ahead, pinned there against the scheduler, which would sink the
    // loads to their use
    u32x4 bh[4], bl[4], ah[4], al[4];
    auto mm = [&](int k) {
        const v8h Bh = __builtin_bit_cast(v8h, bh[k]), Bl = __builtin_bit_cast(v8h, bl[k]), Ah = __builtin_bit_cast(v8h, ah[k]), Al = __builtin_bit_cast(v8h, al[k]);
        ahh = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bh, ahh, 0, 0, 0);
        ahl = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, Bl, ahl, 0, 0, 0);
        alh = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, Bh, alh, 0, 0, 0);
    };
    const int jA = kh * (nsteps >> 1), jE = jA + (nsteps >> 1);           // this wave's K-steps (nsteps is a multiple of four: an even half)
    ldB(jA, &bh[0], &bl[0]); ldA(jA, &ah[0], &al[0]);
    ldB(jA + 1, &bh[1], &bl[1]); ldA(jA + 1, &ah[1], &al[1]);
    for (int j = jA; j < jE; j += 2) {
        const int j2 = j + 2 < jE ? j + 2 : j, j3 = j + 3 < jE ? j + 3 : j;           // (the last round's look-ahead: any valid step)
        __builtin_amdgcn_sched_barrier(0);
        ldB(j2, &bh[2], &bl[2]); ldA(j2, &ah[2], &al[2]);
        __builtin_amdgcn_sched_barrier(0);
        mm(0);
        __builtin_amdgcn_sched_barrier(0);
        ldB(j3, &bh[3], &bl[3]); ldA(j3, &ah[3], &al[3]);
        __builtin_amdgcn_sched_barrier(0);
        mm(1);
        __builtin_amdgcn_sched_barrier(0);
        bh[0] = bh[2]; bl[0] = bl[2]; ah[0] = ah[2]; al[0] = al[2];
        bh[1] = bh[3]; bl[1] = bl[3]; ah[1] = ah[3]; al[1] = al[3];
    }
    // the two waves of a group exchange halves of their sums through LDS (on top of the planes, which nobody reads any more): the wave of the
    // first K-half finishes accumulator registers 0 .. 7 (frames 8 q + 4 kgr + r of a block, q = 0, 1), the other one 8 .. 15 (q = 2, 3)
    float acc[8];
    {
        __syncthreads();
        float *xs = reinterpret_cast<float *>(pl) + (size_t)(2 * wave + (kh ^ 1)) * 8 * 64;       // what the partner will read
#pragma unroll
        for (int v = 0; v < 8; v++) { const int r = 8 * (kh ^ 1) + v; xs[v * 64 + lane] = ahh[r] + (ahl[r] + alh[r]); }
        __syncthreads();
        const float *xr = reinterpret_cast<const float *>(pl) + (size_t)(2 * wave + kh) * 8 * 64;
#pragma unroll
        for (int v = 0; v < 8; v++) { const int r = 8 * kh + v; acc[v] = ((ahh[r] + (ahl[r] + alh[r])) + xr[v * 64 + lane]) * OSC; }
    }
    // ---- per frame: gain, fade, test tone, peaks, store (as audio_fft_kernel).  Accumulator register v = 4 q + r of the lane is frame
    //      8 q + 4 kgr + r of the block (component comp); the lane of the left channel takes r = 0, 1 with the right channel's values from its
    //      neighbour, the lane of the right channel r = 2, 3
    const float gl = P.volume * P.left_ch, gr = P.volume * P.right_ch;
    const ChanState *__restrict__ st = &B.state[ch];
    const int64_t F = st->fade_start_frame;
    const int Max = 24000;
    const int cnt0 = st->pk_cnt, tt0 = st->tt_pos;
    const int tile = 2 * wave + (blk >> 3);           // the 256-frame tile of the lane's block (the same for a row of 16 lanes)
    const int i_tile = (int)(mb - G.M0) + tile * C_TILE;
    float pv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qq = 0; qq < 2; qq++) {
        const int q = 2 * kh + qq;
        const float a0 = acc[4 * qq], a1 = acc[4 * qq + 1], a2 = acc[4 * qq + 2], a3 = acc[4 * qq + 3];
        const float k0 = comp ? a2 : a0, k1 = comp ? a3 : a1;
        const float r0 = dpp_swap1(comp ? a0 : a2), r1 = dpp_swap1(comp ? a1 : a3);
        const float fl[2] = {comp ? r0 : k0, comp ? r1 : k1}, fr[2] = {comp ? k0 : r0, comp ? k1 : r1};
        const int rel0 = FRW * wave + 32 * blk + 8 * q + 4 * kgr + 2 * comp;      // the lane's first frame of this q
        float2 o[2];
        bool lv[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int64_t m = mb + rel0 + k;
            const bool live = m < G.M1;
            lv[k] = live;
            float al_ = fl[k] * gl, ar_ = fr[k] * gr;
            if (live) {
                if (G.gain_fix && m - G.M0 < GAIN_FIX_FRAMES) {
                    const float2 c = B.gfix[(size_t)ch * GAIN_FIX_FRAMES + (m - G.M0)];
                    al_ += c.x; ar_ += c.y;
                }
                const int64_t since = m - F;                  // start-up fade fm-processor.cpp:638-642
                if (since >= 0 && since < Max) {
                    const float cnt = (float)(Max - (int)since);
                    const float f = ((float)Max - cnt) / (float)Max;
                    al_ *= f; ar_ *= f;
                }
                const int i = (int)(m - G.M0);
                if (P.test_tone) {                            // insertTestTone fm-processor.cpp:800-823
#pragma clang fp contract(off)
                    const float level = 0.9f;
                    al_ = al_ * (1.0f - level); ar_ = ar_ * (1.0f - level);
                    const int pos = (int)(((int64_t)tt0 + i) % TT_CYCLE);
                    if (pos >= TT_SILENT) {
                        const float smpl = level * B.tone[pos - TT_SILENT];
                        al_ = al_ + smpl; ar_ = ar_ + smpl;
                    }
                }
                const bool second = (cnt0 + i) / PK_WIN != (cnt0 + i_tile) / PK_WIN;
                pv[second ? 2 : 0] = fmaxf(pv[second ? 2 : 0], fabsf(al_)); pv[second ? 3 : 1] = fmaxf(pv[second ? 3 : 1], fabsf(ar_));
            }
            o[k] = make_float2(al_, ar_);
        }
        float2 *dst = pcm + (size_t)ch * G.pcm_stride + (mb + rel0 - G.M0);
        if (lv[1] && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) *reinterpret_cast<float4 *>(dst) = make_float4(o[0].x, o[0].y, o[1].x, o[1].y);
        else { if (lv[0]) dst[0] = o[0]; if (lv[1]) dst[1] = o[1]; }
    }
    // peak maxima per 256-frame tile: the 16 lanes of a row share their tile (non-negative floats order like their bit patterns)
#pragma unroll
    for (int q = 0; q < 4; q++) {
        int v = __float_as_int(pv[q]);
#define AM_MAX_STEP(ctrl) v = max(v, __builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, false))
        AM_MAX_STEP(0x111); AM_MAX_STEP(0x112); AM_MAX_STEP(0x114); AM_MAX_STEP(0x118);       // row_shr 1, 2, 4, 8: lane 15 of a row has the row's
#undef AM_MAX_STEP
        if ((lane & 15) == 15) atomicMax(&pkt[tile][q], v);
    }
    __syncthreads();
    const int tiles = (int)(((G.M1 - mb) < FR ? (G.M1 - mb) : FR) + C_TILE - 1) / C_TILE;
    if (t < tiles) B.pk_part[(size_t)ch * B.pk_tiles + (FR / C_TILE) * blockIdx.x + t] = make_float4(__int_as_float(pkt[t][0]), __int_as_float(pkt[t][1]), __int_as_float(pkt[t][2]), __int_as_float(pkt[t][3]));
}
