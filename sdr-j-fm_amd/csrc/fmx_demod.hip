// fmx_demod.hip -- stage B: everything that runs at fmRate (192 kS/s) between the decimators and
// the audio low-pass.  COMPILED WITH -ffp-contract=off: the LUT index expressions and feedback
// loops below are evaluated in exactly the f32/f64 types the reference's C++ uses, because a
// one-ulp difference can flip a table index.
//
// Replaces per channel:
//   fm_Demodulator::demodulate        fm-demodulator.cpp:111-205 (+ compAtan Xtan2.cpp:56-100, pllC.cpp:67-90)
//   pilotRecovery::getPilotPhase      pilot-recover.cpp:54-83
//   process_signal_with_rds (stereo)  fm-processor.cpp:689-730
//   PerfectStereoSeparation           stereo-separation.cpp:60-109 (its overlap-add low-pass is a
//                                     295-tap direct FIR with the same 1753-sample latency)
//   L/R matrix, selector              fm-processor.cpp:517-549
//   de-emphasis, gain                 fm-processor.cpp:594-595, 303-306
//
// MI355X design: one wavefront per channel, persistent over the call.  The work is cut into
// chunks of 256 fm samples; inside a chunk the time-parallel parts (limiter + atan2 LUT
// discriminator, the PSS low-pass, the 38 kHz mix and the matrix) use all 64 lanes, while the
// genuinely sequential feedback loops (AFC, pilot PLL, PSS integrator, lock state machines,
// de-emphasis) run in lane 0 from LDS.  Parallelism across the chip comes from the channels.
#include "fmx_internal.h"

namespace fmx {

#define FMX_2PI 6.283185307179586476925286766559   /* 2 * M_PI as the double the reference uses */
#define FMX_PI_4 0.78539816339744830962

// ---- PI_Constrain fm-constants.h:148-158
__device__ __forceinline__ float pi_constrain(float val) {
    const double v = (double)val;
    if (0.0 <= v && v < FMX_2PI) return val;
    if (v >= FMX_2PI) return (float)fmod(v, FMX_2PI);
    if (v > -FMX_2PI) return (float)(v + FMX_2PI);
    return (float)(FMX_2PI - fmod(-v, FMX_2PI));
}
// ---- SinCos sincos.cpp:63-97
__device__ __forceinline__ int sc_index(float phase, double C) {    // phase >= 0
    return ((int)((double)phase * C)) % SINCOS_N;
}
__device__ __forceinline__ float sc_sin(const float2 *__restrict__ tab, double C, float phase) {
    if (phase < 0) return -tab[sc_index(-phase, C)].y;
    return tab[sc_index(phase, C)].y;
}
__device__ __forceinline__ float sc_wrap(float phase) {
    while (phase < 0) phase = (float)((double)phase + FMX_2PI);
    return (float)fmod((double)phase, FMX_2PI);
}
__device__ __forceinline__ float2 sc_complex(const float2 *__restrict__ tab, double C, float phase) {
    return tab[sc_index(sc_wrap(phase), C)];
}
// ---- compAtan::atan2 Xtan2.cpp:56-100.  Only the PPY table is stored; the other seven tables are
// the reference's own f32 expressions of it (Xtan2.cpp:31-38), evaluated here with the same ops.
__device__ __forceinline__ int at_idx(float size, float num, float den) {
    return (int)((double)(size * num / den) + 0.5);
}
__device__ __forceinline__ float lut_atan2(const float *__restrict__ ppy, float y, float x) {
    const float St = (float)3.14159265358979323846;
    if (isinf(x) || isinf(y)) return 0.f;
    if (isnan(x) || isnan(y)) return 0.f;
    if (x == 0.f) {
        if (y == 0.f) return 0.f;
        return y > 0.f ? (float)(3.14159265358979323846 / 2) : (float)(-3.14159265358979323846 / 2);
    }
    const float S = (float)ATAN_N, E = -(float)ATAN_N;
    if (x > 0.f) {
        if (y >= 0.f) {
            if (x >= y) return ppy[at_idx(S, y, x)];                     // PPY
            return St * 0.5f - ppy[at_idx(S, x, y)];                     // PPX
        }
        if (x >= -y) return -ppy[at_idx(E, y, x)];                       // PNY
        return ppy[at_idx(E, x, y)] - St * 0.5f;                         // PNX
    }
    if (y >= 0.f) {
        if (-x >= y) return St - ppy[at_idx(E, y, x)];                   // NPY
        return ppy[at_idx(E, x, y)] + St * 0.5f;                         // NPX
    }
    if (x <= y) return ppy[at_idx(S, y, x)] - St;                        // NNY
    return -St * 0.5f - ppy[at_idx(S, x, y)];                            // NNX
}

constexpr int WIN = B_CHUNK + PSS_TAPS - 1;      // 550 PSS low-pass window entries

__global__ __launch_bounds__(64) void demod_kernel(DeviceTables T, DeviceBuffers B, CallGeom G) {
    __shared__ float2 sIQ[B_CHUNK + 2];          // limiter outputs; [0],[1] = two previous samples
    __shared__ float  sRES[B_CHUNK];             // discriminator output before AFC
    __shared__ float  sERR[B_CHUNK];             // PSS error Re*Im for call index i0 + r
    __shared__ float  sDEM[B_CHUNK];
    __shared__ float  sPH[B_CHUNK];              // 38 kHz mixing phase (phaseforLRDiff)
    __shared__ int    sIDX[B_CHUNK];             // -2 mono branch, -1 stereo without PSS, >=0 s-ring index offset
    __shared__ float2 sWIN[WIN];                 // PSS window, later reused for the matrix output
    __shared__ float2 sY[B_CHUNK];               // de-emphasised stereo

    const int ch = blockIdx.x;
    const int lane = threadIdx.x;
    const ChanParams P = B.params[ch];
    const FrontSet FS = T.front_sets[P.front_set];
    ChanState *stp = B.state + ch;
    const int ring = G.ring_mask + 1;
    const float2 *__restrict__ zring = B.zring + (size_t)ch * ring;
    float *demod_ring = B.demod_ring + (size_t)ch * ring;
    float2 *lr_ring = B.lr_ring + (size_t)ch * ring;
    float2 *sring = B.sring + (size_t)ch * (G.sring_mask + 1);
    float2 *dring = B.dring + (size_t)ch * (G.dring_mask + 1);
    const float2 *__restrict__ sct = T.sincos;
    const double SC = T.sincos_C;
    const int decoder = P.decoder;
    const bool stereo_possible = (P.fm_mode != 2);
    const bool want_pss = stereo_possible && (P.pss_active != 0);

    ChanState st = *stp;                         // every lane holds a copy; lane 0 is authoritative
    int64_t fade_start = st.fade_start_frame;
    if (P.actions & (ACT_TRIGGER_FREQ | ACT_RESTART_PSS)) {
        // triggerFrequencyChange / restartPssAnalyzer fm-processor.cpp:849-860
        st.pilot_delay_pss = 0.f;
        st.pss_acc = 0.f; st.pss_minimized = 0; st.pss_mean = 0.f; st.pss_lock_cnt = 0; st.pss_unlock_cnt = 0;
        if (P.actions & ACT_TRIGGER_FREQ) fade_start = G.M0;
    }

    for (int64_t jc = G.J0; jc < G.J1; jc += B_CHUNK) {
        const int cnt = (int)((G.J1 - jc) < B_CHUNK ? (G.J1 - jc) : B_CHUNK);
        // ================= phase 1: limiter (fm-demodulator.cpp:119-126) =================
        if (lane == 0) { sIQ[0] = make_float2(st.Imin2, st.Qmin2); sIQ[1] = make_float2(st.Imin1, st.Qmin1); }
        for (int r = lane; r < cnt; r += 64) {
            const int64_t j = jc + r;
            const int64_t jv = j - FS.delay_fm;                // overlap-add latency of the input filter
            float2 z = make_float2(0.f, 0.f);
            if (jv >= 0) z = zring[jv & G.ring_mask];
            const float zAbs = (float)sqrt((double)z.x * (double)z.x + (double)z.y * (double)z.y);  // hypotf
            float I, Q;
            if ((double)zAbs <= 0.001) { I = Q = (float)0.001; }
            else { I = z.x / zAbs; Q = z.y / zAbs; }
            sIQ[r + 2] = make_float2(I, Q);
        }
        __syncthreads();
        // ================= phase 1b: memoryless discriminators =================
        if (decoder != 2) {
            for (int r = lane; r < cnt; r += 64) {
                const float2 c = sIQ[r + 2], p1 = sIQ[r + 1], p2 = sIQ[r];
                const float I = c.x, Q = c.y, I1 = p1.x, Q1 = p1.y;
                float res;
                if (decoder == 5) {            // REAL_BB :174-182
                    res = (float)((double)(I1 * Q - Q1 * I + 1) / 2.0);
                    int index = (int)floorf(res * (float)ARCSINE_N);
                    if (index < 0) index = 0;
                    if (index >= ARCSINE_N) index = ARCSINE_N;
                    res = T.arcsine[index];
                } else if (decoder == 6) {     // DIFF :184-189
                    const float Scaler = (float)1.4142135623730951;
                    res = (I1 * (Q - p2.y) - Q1 * (I - p2.x));
                    res /= (I1 * I1 + Q1 * Q1) * Scaler;
                } else {                       // MIXED :168-172 (COMPLEX_BB :174-177 is bitwise the same)
                    res = lut_atan2(T.atan_ppy, Q * I1 - I * Q1, I * I1 + Q * Q1);
                }
                sRES[r] = res;
            }
        }
        // ================= phase 1c: PSS low-pass -> error for this chunk's call indices ==========
        const int64_t i0 = st.pss_count;
        if (want_pss) {
            for (int w = lane; w < WIN; w += 64) {
                const int64_t idx = i0 - (PSS_DELAY + PSS_TAPS - 1) + w;
                sWIN[w] = (idx >= 0) ? sring[idx & G.sring_mask] : make_float2(0.f, 0.f);
            }
            __syncthreads();
            float ar[4] = {0.f, 0.f, 0.f, 0.f}, ai[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < PSS_TAPS; k++) {
                const float h = T.pss_taps[k];
                const int w = lane + (PSS_TAPS - 1) - k;
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const float2 v = sWIN[w + 64 * m];
                    ar[m] = fmaf(h, v.x, ar[m]); ai[m] = fmaf(h, v.y, ai[m]);
                }
            }
#pragma unroll
            for (int m = 0; m < 4; m++) sERR[lane + 64 * m] = ar[m] * ai[m];
        }
        __syncthreads();
        // ================= phase 2: the sequential loops (lane 0) =================
        if (lane == 0) {
            const float fmDcAlpha = 0.0001f;
            const float lockA = 1.0f / 3000.0f;
            int64_t ipss = i0;
            for (int r = 0; r < cnt; r++) {
                float res;
                if (decoder == 2) {            // pllC::do_pll pllC.cpp:67-90
                    const float2 sig = sIQ[r + 2];
                    const float2 nco = sc_complex(sct, SC, st.nco_phase);
                    // conj(nco) * signal
                    const float dre = nco.x * sig.x - (-nco.y) * sig.y;
                    const float dim = nco.x * sig.y + (-nco.y) * sig.x;
                    const float perr = lut_atan2(T.atan_ppy, dim, dre);
                    st.phase_incr = (1 - T.pll_beta) * perr + T.pll_beta * st.phase_incr;
                    if (st.phase_incr < T.pll_lo || st.phase_incr > T.pll_hi) st.phase_incr = T.pll_center;
                    st.nco_phase += st.phase_incr;
                    if ((double)st.nco_phase >= FMX_2PI) st.nco_phase = (float)fmod((double)st.nco_phase, FMX_2PI);
                    else while (st.nco_phase < 0) st.nco_phase = (float)((double)st.nco_phase + FMX_2PI);
                    res = st.phase_incr;
                } else res = sRES[r];
                // AFC + scaling fm-demodulator.cpp:197-198
                st.fm_afc = (1 - fmDcAlpha) * st.fm_afc + fmDcAlpha * res;
                const float demod = 20.0f * (res - st.fm_afc) * 1.0f / T.K_FM;
                sDEM[r] = demod;
                // pilot PLL pilot-recover.cpp:54-83
                const float pilot = 5 * demod;
                const float osc = sc_sin(sct, SC, st.pil_phase);
                const float perr = pilot * osc;
                st.pil_phase += perr * T.pil_gain;
                const float cur = pi_constrain(st.pil_phase);
                st.pil_phase = pi_constrain(st.pil_phase + T.pil_omega);
                const float quadRef = (osc - st.pil_old) / T.pil_omega;
                st.pil_old = osc;
                st.pil_lock = (float)((double)(lockA * (-quadRef * pilot)) + (double)st.pil_lock * (1.0 - (double)lockA));
                if (st.pil_lock > 0.07f) {
                    if (st.pil_locked || ++st.pil_stable > (SINCOS_N >> 1)) st.pil_locked = 1;
                } else { st.pil_locked = 0; st.pil_stable = 0; }
                // process_signal_with_rds fm-processor.cpp:699-730
                if (!st.pil_locked) {
                    st.pilot_delay_pss = 0.f;
                    st.pss_acc = 0.f; st.pss_minimized = 0; st.pss_mean = 0.f; st.pss_lock_cnt = 0; st.pss_unlock_cnt = 0;
                }
                int tag = -2;
                float ph = 0.f;
                if (stereo_possible && (st.pil_locked || !P.auto_mono)) {
                    ph = (float)(2 * ((double)cur + FMX_PI_4 + 0) - (double)st.pilot_delay_pss);
                    if ((double)ph < -FMX_2PI) ph = (float)((double)ph + 2 * FMX_2PI);
                    ph = (float)fmod((double)ph, FMX_2PI);
                    if (P.pss_active) {        // PerfectStereoSeparation::process_sample :60-109
                        tag = (int)(ipss - i0);
                        float error = sERR[tag];
                        ipss++;
                        if (!st.pss_minimized) error *= 10.0f;
                        st.pss_acc += T.pss_alpha * error;
                        st.pss_mean = T.pss_lock_alpha * error + st.pss_mean * (1.0f - T.pss_lock_alpha);
                        if (fabsf(st.pss_mean) < 0.001f) {
                            if (st.pss_minimized || (++st.pss_lock_cnt > 3 * SINCOS_N)) st.pss_minimized = 1;
                            st.pss_unlock_cnt = 0;
                        } else {
                            if (!st.pss_minimized || (++st.pss_unlock_cnt > 3 * SINCOS_N)) st.pss_minimized = 0;
                            st.pss_lock_cnt = 0;
                        }
                        if ((double)st.pss_acc < -FMX_PI_4) st.pss_acc = (float)-FMX_PI_4;
                        else if ((double)st.pss_acc > FMX_PI_4) st.pss_acc = (float)FMX_PI_4;
                        st.pilot_delay_pss = st.pss_acc;
                    } else { tag = -1; st.pilot_delay_pss = 0.f; }
                }
                sPH[r] = ph; sIDX[r] = tag;
                // meta snapshot fm-processor.cpp:662-684
                if (++st.my_count > (SINCOS_N >> 1)) {
                    const bool lk = stereo_possible && st.pil_locked;
                    st.meta_locked = lk ? 1 : 0;
                    st.meta_lock_strength = stereo_possible ? st.pil_lock : 0.f;
                    const float dcabs = (float)sqrt((double)st.dc_re * (double)st.dc_re + (double)st.dc_im * (double)st.dc_im);
                    st.meta_dc_rf = P.dc_remove ? 20 * log10f(dcabs + 1.0f / 32768) : (float)-99.99;
                    st.meta_dc_if = st.fm_afc;
                    st.meta_pss_deg = (float)((double)st.pilot_delay_pss / 3.14159265358979323846 * 180.0f);
                    st.meta_pss_change = st.pss_mean * 1000;
                    st.meta_pss_state = (P.pss_active && lk) ? (st.pss_minimized ? 2 : 1) : 0;
                    st.my_count = 0;
                }
            }
            st.pss_count = ipss;
            const float2 l1 = sIQ[cnt + 1], l2 = sIQ[cnt];
            st.Imin1 = l1.x; st.Qmin1 = l1.y; st.Imin2 = l2.x; st.Qmin2 = l2.y;
        }
        __syncthreads();
        // ================= phase 3: 38 kHz mix, PSS input, matrix (all lanes) =================
        for (int r = lane; r < cnt; r += 64) {
            const int64_t j = jc + r;
            const float demod = sDEM[r];
            const int tag = sIDX[r];
            float2 audio = make_float2(demod, 0.f);
            if (tag != -2) {
                const float ph = sPH[r];
                if (tag >= 0) {
                    const float2 e = sc_complex(sct, SC, ph);
                    sring[(i0 + tag) & G.sring_mask] = make_float2(e.x * demod, e.y * demod);
                }
                float lut;
                if (P.sound_sel == 6) lut = sc_sin(sct, SC, ph);
                else lut = sct[sc_index(sc_wrap(ph), SC)].x;
                audio.y = (float)(2.0 * (double)lut * (double)demod);
            }
            demod_ring[j & G.ring_mask] = demod;
            lr_ring[j & G.ring_mask] = audio;
            const float sumLR = audio.x, diffLR = audio.y;
            const float dw = diffLR * (P.fm_mode == 1 ? P.panorama : 1.0f);
            const float left = sumLR + dw, right = sumLR - dw;
            float2 o;
            switch (P.sound_sel) {
            default:
            case 0: o = make_float2(left, right); break;
            case 1: o = make_float2(right, left); break;
            case 2: o = make_float2(left, left); break;
            case 3: o = make_float2(right, right); break;
            case 4: o = make_float2(sumLR, sumLR); break;
            case 5: case 6: o = make_float2(dw, dw); break;
            }
            sWIN[r] = o;
        }
        __syncthreads();
        // ================= phase 4: de-emphasis (lanes 0/1 = L/R), fm-processor.cpp:594-595 =======
        if (lane < 2) {
            const float a = P.deemph_alpha;
            float y = lane == 0 ? st.de_l : st.de_r;
            const float *xin = reinterpret_cast<const float *>(sWIN) + lane;
            float *yout = reinterpret_cast<float *>(sY) + lane;
            for (int r = 0; r < cnt; r++) {
                y = (xin[2 * r] - y) * a + y;
                yout[2 * r] = y;
            }
            if (lane == 0) st.de_l = y; else st.de_r = y;
        }
        // lane 1's de_r -> lane 0 (the authoritative copy)
        {
            const float der = __shfl(st.de_r, 1, 64);
            if (lane == 0) st.de_r = der;
        }
        __syncthreads();
        // ================= phase 5: gain (audioGainCorrection :303-306) -> d ring =================
        for (int r = lane; r < cnt; r += 64) {
            const int64_t j = jc + r;
            const float2 y = sY[r];
            dring[j & G.dring_mask] = make_float2(P.volume * P.left_ch * y.x, P.volume * P.right_ch * y.y);
        }
        // broadcast lane 0's state so every lane starts the next chunk consistently
        {
            int *w = reinterpret_cast<int *>(&st);
#pragma unroll
            for (unsigned i = 0; i < sizeof(ChanState) / 4; i++) w[i] = __shfl(w[i], 0, 64);
        }
        __syncthreads();
    }
    if (lane == 0) {
        // the front end owns dc/lo state: do not overwrite what it stored this call
        ChanState out = st;
        out.dc_re = stp->dc_re; out.dc_im = stp->dc_im; out.lo_phase = stp->lo_phase;
        out.fade_start_frame = fade_start;
        *stp = out;
    }
}

void launch_demod(const DeviceTables &T, const DeviceBuffers &B, const CallGeom &G, int channels, hipStream_t s) {
    if (G.J1 <= G.J0) return;
    hipLaunchKernelGGL(demod_kernel, dim3(channels), dim3(64), 0, s, T, B, G);
}

}  // namespace fmx
