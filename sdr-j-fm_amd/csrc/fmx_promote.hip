// fmx_promote.hip -- a batch handle leaves its folded filters for the block machines (fmx_ola.hip) behind a mid-stream setBandwidth / setlfcutoff.
//
// The reference's overlap-add filters are block machines (fft-filters.cpp:84-95,132-163): a setLowPass in mid-stream replays the last output block, drops
// the block in progress and carries the old tail over.  Handles of up to 64 channels run the machines from their first call.  A larger batch runs the
// filters folded into stage A's / stage C's polyphase FIRs (one eighth of the cost) -- until a filter setter arrives in mid-stream.  The change then stays
// PENDING while the library keeps a copy of what the streams deliver (capture_kernel; 3 blocks of the input filter: 85 ms), and at the first call boundary
// behind that the handle is PROMOTED (fmx_api.hip promote): the machines' state at that sample -- block in progress, last block result, Overloop -- is
// what the machines leave when they are run over the kept samples from the channel state snapshotted at their start (the same kernels a small handle
// runs every call); the audio machine's likewise from the d ring's last three blocks with the de-emphasis taken back out; the decimators' history and the
// fm-rate ring's newest entries are moved to where the machines' stage A looks for them.  From that call on the handle is a block-machine handle and the
// pending setter restarts its filter exactly as the reference's does (ola_take_settings).  The kernels here are the small glue around that.
#include "fmx_internal.h"
#include <algorithm>

namespace fmx {

// samples [0, n) of every stream of a call, any input format, appended to the handle's tail buffer as float32 IQ (the conversions are front_kernel's:
// exact, as the reference's device handlers)
template <int FMT>
__global__ __launch_bounds__(256) void capture_kernel(const void *__restrict__ iq, int64_t stream_stride, int64_t n, float qs, float2 *__restrict__ tail, int64_t tail_cap, int64_t pos) {
    constexpr int BPS = (FMT == 0) ? 8 : (FMT == 3 ? 4 : 2);
    const int sidx = blockIdx.y;
    const char *inb = reinterpret_cast<const char *>(iq) + (size_t)sidx * stream_stride * BPS;
    float2 *out = tail + (size_t)sidx * tail_cap + pos;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float2 v;
        if (FMT == 0) v = reinterpret_cast<const float2 *>(inb)[i];
        else if (FMT == 1) { const uint8_t *p = reinterpret_cast<const uint8_t *>(inb) + 2 * (size_t)i; v = make_float2((float)((int)p[0] - 127) * qs, (float)((int)p[1] - 127) * qs); }
        else if (FMT == 2) { const int8_t *p = reinterpret_cast<const int8_t *>(inb) + 2 * (size_t)i; v = make_float2((float)p[0] * qs, (float)p[1] * qs); }
        else { const int16_t *p = reinterpret_cast<const int16_t *>(inb) + 2 * (size_t)i; v = make_float2((float)p[0] * qs, (float)p[1] * qs); }
        out[i] = v;
    }
}
void launch_capture(const void *iq, int fmt, float qs, int64_t stream_stride, int64_t n, int streams, float2 *tail, int64_t tail_cap, int64_t pos, hipStream_t s) {
    const dim3 grid((unsigned)std::min<int64_t>((n + 255) / 256, 256), (unsigned)streams);
    switch (fmt) {
    case 1: hipLaunchKernelGGL(capture_kernel<1>, grid, dim3(256), 0, s, iq, stream_stride, n, qs, tail, tail_cap, pos); break;
    case 2: hipLaunchKernelGGL(capture_kernel<2>, grid, dim3(256), 0, s, iq, stream_stride, n, qs, tail, tail_cap, pos); break;
    case 3: hipLaunchKernelGGL(capture_kernel<3>, grid, dim3(256), 0, s, iq, stream_stride, n, qs, tail, tail_cap, pos); break;
    default: hipLaunchKernelGGL(capture_kernel<0>, grid, dim3(256), 0, s, iq, stream_stride, n, qs, tail, tail_cap, pos); break;
    }
}

// what pre_kernel reads of a channel's state, in front of the first kept sample (mode 0: state -> snapshot) / back in front of the run over the kept
// samples (mode 1: snapshot -> state; the de-emphasis state, which deemph_kernel owns from here on, starts its run over the d ring's tail from zero)
__global__ __launch_bounds__(256) void promo_state_kernel(ChanState *__restrict__ st, FrontSnap *__restrict__ snap, int channels, int mode) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= channels) return;
    if (mode == 0) { FrontSnap sn; sn.lo_phase = st[c].lo_phase; sn.hist_fmt = st[c].hist_fmt; sn.dc_re = st[c].dc_re; sn.dc_im = st[c].dc_im; snap[c] = sn; }
    else { const FrontSnap sn = snap[c]; st[c].lo_phase = sn.lo_phase; st[c].dc_re = sn.dc_re; st[c].dc_im = sn.dc_im; st[c].hist_fmt = 0; st[c].de_l = 0.f; st[c].de_r = 0.f; }
}
void launch_promo_state(ChanState *st, FrontSnap *snap, int channels, int mode, hipStream_t s) {
    hipLaunchKernelGGL(promo_state_kernel, dim3((unsigned)((channels + 255) / 256)), dim3(256), 0, s, st, snap, channels, mode);
}

// The decimators' history (stage A behind the machines reads the machines' output stream, CallGeom::pre_processed: 24 whole columns of 12 samples and
// the column in progress, raw) from the last samples of the machines' output over the kept samples -- u[ch][0 .. len) ends at the promotion sample g0 --
// and the fm-rate ring's newest entries where a stage B that reads the ring WITHOUT the folded filter's delay looks for the two samples in front of the
// call's first (the folded stage A kept the ring `delay` entries ahead of what stage B read).
__global__ __launch_bounds__(320) void promo_hist_kernel(float2 *__restrict__ hist, const float2 *__restrict__ u, int64_t u_stride, int64_t len, int r0, int twins,
                                                           float2 *__restrict__ zring, int ring_mask, int64_t J0, const ChanParams *__restrict__ params,
                                                           const FrontSet *__restrict__ old_sets, const int32_t *__restrict__ old_set_of) {
    const int ch = blockIdx.x, t = threadIdx.x;
    const float2 *uc = u + (size_t)ch * u_stride;
    if (t < DECIM * A_HIST_COLS) {
        const int r = t / A_HIST_COLS, c = t - r * A_HIST_COLS;
        // sample of the stream relative to the promotion sample: column qa - 24 + c, row r; qa's column holds the r0 samples in front of g0
        const int64_t rel = (int64_t)DECIM * (c - (A_HIST_COLS - 1)) + r - r0;
        float2 v = make_float2(0.f, 0.f);
        if (rel < 0 && len + rel >= 0) v = uc[len + rel];
        for (int tw = 0; tw < twins; tw++) hist[((size_t)ch * twins + tw) * DECIM * A_HIST_COLS + t] = v;
    }
    const int delay = old_sets[old_set_of[ch]].delay_fm;
    float2 *zr = zring + (size_t)ch * (ring_mask + 1);
    float2 v = make_float2(0.f, 0.f);
    if (t < 16) v = zr[(J0 - 1 - t - delay) & ring_mask];
    __syncthreads();
    if (t < 16 && delay != 0) zr[(J0 - 1 - t) & ring_mask] = v;
}
void launch_promo_hist(float2 *hist, const float2 *u, int64_t u_stride, int64_t len, int r0, int twins, float2 *zring, int ring_mask, int64_t J0,
                       const ChanParams *params, const FrontSet *old_sets, const int32_t *old_set_of, int channels, hipStream_t s) {
    hipLaunchKernelGGL(promo_hist_kernel, dim3((unsigned)channels), dim3(320), 0, s, hist, u, u_stride, len, r0, twins, zring, ring_mask, J0, params, old_sets, old_set_of);
}

// The stereo pair in front of the de-emphasis (what the reference's audio filter sees, fm-processor.cpp:589-595) from the d ring of a folded handle,
// which holds it de-emphasised, y[n] = (x[n] - y[n-1]) a + y[n-1]:  x[n] = (y[n] - y[n-1]) / a + y[n-1]  for fm samples J0 - NA .. J0 - 1 (zeros in front
// of the stream's first sample)
__global__ __launch_bounds__(256) void promo_inv_deemph_kernel(const float2 *__restrict__ dring, int dmask, int64_t J0, int NA, const ChanParams *__restrict__ params,
                                                                 float2 *__restrict__ out) {
    const int ch = blockIdx.y;
    const float a = params[ch].deemph_alpha, ra = 1.0f / a;
    const float2 *dr = dring + (size_t)ch * (dmask + 1);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < NA; i += gridDim.x * 256) {
        const int64_t j = J0 - NA + i;
        float2 v = make_float2(0.f, 0.f);
        if (j >= 0) {
            const float2 y = dr[j & dmask], yp = j >= 1 ? dr[(j - 1) & dmask] : make_float2(0.f, 0.f);
            v = make_float2((y.x - yp.x) * ra + yp.x, (y.y - yp.y) * ra + yp.y);
        }
        out[(size_t)ch * NA + i] = v;
    }
}
void launch_promo_inv_deemph(const float2 *dring, int dmask, int64_t J0, int NA, const ChanParams *params, float2 *out, int channels, hipStream_t s) {
    hipLaunchKernelGGL(promo_inv_deemph_kernel, dim3(32, (unsigned)channels), dim3(256), 0, s, dring, dmask, J0, NA, params, out);
}

}  // namespace fmx
