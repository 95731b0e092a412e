// multi_gpu_host.cpp -- the C++ host of the N-GPU deployment (BASELINE north_star: "host side stays C++ ... one rank per GPU with
// RCCL over xGMI only for the fan-out / gather").  One process, one host thread per GPU; each thread owns ONE fmx handle for its shard of
// the channels (static block partition, what sdr-j-fm_amd/shard.py does in Python) and one RCCL communicator rank.  There is NO collective
// on the data path -- the channels are independent --; RCCL appears either side of it:
//   fan-out   ncclBroadcast of a shared wide-band IQ stream from rank 0 (BASELINE configs[2]: every GPU demodulates its own carriers out of
//             the same samples)
//   gather    the PCM of a step from every rank to rank 0 (ncclSend / ncclRecv inside one group), where an audio sink would take it
//   clock     ncclAllReduce (max) of the per-rank times: the job is as fast as its slowest rank
// Everything goes through the C ABI (include/fmx.h): what this file needs from the library is what any C++ / Qt host needs.
//
//   hipcc -O2 -std=c++17 multi_gpu_host.cpp -I../../include -L../lib -lfmx -lrccl -o multi_gpu_host
//   ./multi_gpu_host --gpus 8 --channels 4096 --steps 20 [--block 230400] [--shared-streams 24]
// prints one JSON line (whole-job MS/s, per-rank values, the RCCL legs).  The bench driver of this repository is bench.py (Python,
// torch.distributed = the same RCCL); this is the host a C++ application would start from.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "fmx.h"

#define CK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #expr, hipGetErrorString(e_)); std::exit(3); } } while (0)
#define NK(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { std::fprintf(stderr, "%s: %s\n", #expr, ncclGetErrorString(r_)); std::exit(4); } } while (0)
#define FK(expr) do { int r_ = (expr); if (r_ != FMX_OK) { std::fprintf(stderr, "%s: %s\n", #expr, fmx_last_error()); std::exit(5); } } while (0)

namespace {

constexpr int kInputRate = 2304000;

// one block of stereo FM (two tones, pilot, L-R on 38 kHz), periodic in n: every frequency a multiple of inputRate / n
std::vector<float> stereo_fm_block(int n, int tone_l, int tone_r, double offset_hz) {
    std::vector<double> inc((size_t)n);
    double mean = 0;
    for (int i = 0; i < n; i++) {
        const double t = (double)i / kInputRate;
        const double L = 0.5 * std::sin(2 * M_PI * tone_l * t), R = 0.5 * std::sin(2 * M_PI * tone_r * t + 1.0);
        const double p19 = 2 * M_PI * 19000.0 * t;
        const double mpx = 0.45 * (L + R) + 0.10 * std::sin(p19) + 0.45 * (L - R) * std::sin(2 * p19);
        inc[(size_t)i] = 2 * M_PI * (75000.0 * mpx + offset_hz) / kInputRate;
        mean += inc[(size_t)i];
    }
    mean = offset_hz != 0 ? 0.0 : mean / n;
    std::vector<float> iq((size_t)2 * n);
    double ph = 0;
    for (int i = 0; i < n; i++) {
        ph += inc[(size_t)i] - mean;
        iq[(size_t)2 * i] = (float)(0.5 * std::cos(ph)); iq[(size_t)2 * i + 1] = (float)(0.5 * std::sin(ph));
    }
    return iq;
}

struct Options { int gpus = 1, channels = 512, steps = 20, warmup = 44, block = 230400, shared_streams = 0; };

struct RankResult { double seconds = 0, gather_ms = 0, bcast_ms = 0; int channels = 0; long frames = 0; bool self_check = true; };

// the shard of `total` channels rank r of `world` owns: a block partition, the remainder spread over the first ranks (shard.py)
void shard_channels(int total, int world, int r, int *first, int *count) {
    const int base = total / world, rem = total % world;
    *count = base + (r < rem ? 1 : 0);
    *first = r * base + std::min(r, rem);
}

void rank_main(int rank, const Options &opt, ncclComm_t comm, std::vector<RankResult> *results) {
    CK(hipSetDevice(rank));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int first = 0, nch = 0;
    shard_channels(opt.channels, opt.gpus, rank, &first, &nch);
    const int n = opt.block;
    const int streams = opt.shared_streams > 0 ? opt.shared_streams : nch;
    RankResult &res = (*results)[(size_t)rank];
    res.channels = nch;

    // ---- the handle of this rank's shard -----------------------------------------------------------------------------------------
    std::vector<int32_t> smap((size_t)nch);
    for (int c = 0; c < nch; c++) smap[(size_t)c] = opt.shared_streams > 0 ? (first + c) % streams : c;
    fmx_config cfg{};
    cfg.struct_size = (int32_t)sizeof cfg; cfg.device = rank; cfg.channels = nch; cfg.streams = streams; cfg.stream_of_channel = smap.data();
    cfg.inputRate = kInputRate; cfg.fmRate = 192000; cfg.workingRate = 48000; cfg.audioRate = 48000; cfg.max_block = n;
    if (fmx_abi_version() != FMX_ABI_VERSION) { std::fprintf(stderr, "libfmx ABI %d, built for %d\n", fmx_abi_version(), FMX_ABI_VERSION); std::exit(6); }
    fmx_handle h = nullptr;
    FK(fmx_create(&cfg, &h));
    FK(fmx_set_param(h, -1, FMX_P_BANDWIDTH, 165000)); FK(fmx_set_param(h, -1, FMX_P_LF_CUTOFF, 15000));
    FK(fmx_set_param(h, -1, FMX_P_DEEMPHASIS, 50)); FK(fmx_set_param(h, -1, FMX_P_VOLUME_DB, -6.0)); FK(fmx_set_param(h, -1, FMX_P_FM_MODE, 0));
    if (opt.shared_streams > 0)
        for (int c = 0; c < nch; c++) FK(fmx_set_param(h, c, FMX_P_LOCAL_OSCILLATOR, (double)((((first + c) / streams) % 11) - 5) * 200000.0));

    // ---- input: resident in HBM before the clock starts ---------------------------------------------------------------------------
    float *d_iq = nullptr; CK(hipMalloc(&d_iq, sizeof(float) * 2 * (size_t)streams * n));
    if (opt.shared_streams > 0) {
        // fan-out: rank 0 holds the wide-band streams (11 carriers on a 200 kHz raster each), RCCL broadcasts them
        if (rank == 0) {
            std::vector<float> sum((size_t)2 * n, 0.f);
            for (int k = 0; k < 11; k++) {
                const std::vector<float> one = stereo_fm_block(n, 300 + 70 * k, 500 + 90 * k, (k - 5) * 200000.0);
                for (size_t i = 0; i < sum.size(); i++) sum[i] += one[i] / 3.5f;
            }
            for (int st = 0; st < streams; st++) CK(hipMemcpy(d_iq + (size_t)2 * st * n, sum.data(), sizeof(float) * sum.size(), hipMemcpyHostToDevice));
        }
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        NK(ncclBroadcast(d_iq, d_iq, (size_t)2 * streams * n, ncclFloat, 0, comm, s));
        CK(hipStreamSynchronize(s));
        res.bcast_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    } else {
        // four programmes, channel c of the whole job receives programme c % 4 (the same on every rank that holds such a channel)
        for (int k = 0; k < 4 && k < nch; k++) {
            const std::vector<float> one = stereo_fm_block(n, 300 + 110 * k, 500 + 130 * k, 0.0);
            for (int c = 0; c < nch; c++)
                if ((first + c) % 4 == k) CK(hipMemcpy(d_iq + (size_t)2 * c * n, one.data(), sizeof(float) * one.size(), hipMemcpyHostToDevice));
        }
    }
    const int64_t cap = n / 48 + 96;
    float *d_pcm = nullptr; CK(hipMalloc(&d_pcm, sizeof(float) * 2 * (size_t)nch * cap));
    CK(hipMemset(d_pcm, 0, sizeof(float) * 2 * (size_t)nch * cap));

    auto step = [&]() { int64_t fr = 0; FK(fmx_process_device(h, d_iq, n, n, d_pcm, cap, &fr, s)); return fr; };
    for (int i = 0; i < std::max(opt.warmup, 44); i++) step();        // (pilot lock and the PSS state machine settle: bench.py's rule)
    CK(hipStreamSynchronize(s));

    // ---- the timed region: a barrier over RCCL, K steps, a barrier; the job's time is the slowest rank's -----------------------------
    float *d_t = nullptr; CK(hipMalloc(&d_t, sizeof(float)));
    auto barrier_max = [&](float v) {
        CK(hipMemcpyAsync(d_t, &v, sizeof v, hipMemcpyHostToDevice, s));
        NK(ncclAllReduce(d_t, d_t, 1, ncclFloat, ncclMax, comm, s));
        float out = 0; CK(hipMemcpyAsync(&out, d_t, sizeof out, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        return out;
    };
    barrier_max(0.f);
    const auto t0 = std::chrono::steady_clock::now();
    long frames = 0;
    for (int i = 0; i < opt.steps; i++) frames += (long)step();
    CK(hipStreamSynchronize(s));
    const float mine = (float)std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    res.seconds = barrier_max(mine);
    res.frames = frames;
    FK(fmx_synchronize(h));

    // ---- gather: one step's PCM of every rank on rank 0 -----------------------------------------------------------------------------
    const int64_t fr = frames / std::max(opt.steps, 1);
    float *d_all = nullptr;
    if (rank == 0) CK(hipMalloc(&d_all, sizeof(float) * 2 * (size_t)opt.channels * cap));
    CK(hipStreamSynchronize(s));
    const auto g0 = std::chrono::steady_clock::now();
    NK(ncclGroupStart());
    NK(ncclSend(d_pcm, (size_t)2 * nch * cap, ncclFloat, 0, comm, s));
    if (rank == 0)
        for (int r = 0; r < opt.gpus; r++) {
            int f2 = 0, c2 = 0; shard_channels(opt.channels, opt.gpus, r, &f2, &c2);
            NK(ncclRecv(d_all + (size_t)2 * f2 * cap, (size_t)2 * c2 * cap, ncclFloat, r, comm, s));
        }
    NK(ncclGroupEnd());
    CK(hipStreamSynchronize(s));
    res.gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g0).count();

    // ---- self check on rank 0: its own shard came back bit for bit; channels of one programme are identical wherever they ran -------
    if (rank == 0 && opt.shared_streams == 0) {
        std::vector<float> all((size_t)2 * opt.channels * cap), own((size_t)2 * nch * cap);
        CK(hipMemcpy(all.data(), d_all, sizeof(float) * all.size(), hipMemcpyDeviceToHost));
        CK(hipMemcpy(own.data(), d_pcm, sizeof(float) * own.size(), hipMemcpyDeviceToHost));
        bool ok = std::memcmp(all.data(), own.data(), sizeof(float) * own.size()) == 0;
        double energy = 0;
        for (int64_t i = 0; i < 2 * fr; i++) energy += (double)own[(size_t)i] * own[(size_t)i];
        ok = ok && energy > 1e-3;
        for (int c = 4; c < opt.channels && ok; c++)
            ok = std::memcmp(&all[(size_t)2 * c * cap], &all[(size_t)2 * (c % 4) * cap], sizeof(float) * 2 * (size_t)fr) == 0;
        res.self_check = ok;
    }
    if (d_all) CK(hipFree(d_all));
    CK(hipFree(d_t)); CK(hipFree(d_pcm)); CK(hipFree(d_iq));
    FK(fmx_destroy(h));
    CK(hipStreamDestroy(s));
}

}  // namespace

int main(int argc, char **argv) {
    Options opt;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&]() { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", a.c_str()); std::exit(2); } return std::atoi(argv[++i]); };
        if (a == "--gpus") opt.gpus = val(); else if (a == "--channels") opt.channels = val(); else if (a == "--steps") opt.steps = val();
        else if (a == "--warmup") opt.warmup = val(); else if (a == "--block") opt.block = val(); else if (a == "--shared-streams") opt.shared_streams = val();
        else { std::fprintf(stderr, "usage: multi_gpu_host [--gpus N] [--channels C] [--steps K] [--warmup W] [--block n] [--shared-streams S]\n"); return 2; }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { std::fprintf(stderr, "no HIP device: libfmx has no CPU fallback\n"); return 3; }
    if (opt.gpus < 1 || opt.gpus > ndev) { std::fprintf(stderr, "--gpus %d but %d device(s) visible\n", opt.gpus, ndev); return 2; }
    if (opt.channels < opt.gpus || opt.block < 12 || opt.steps < 1) { std::fprintf(stderr, "fewer channels than ranks, or an empty run\n"); return 2; }

    // one communicator per GPU of this process (ncclCommInitAll: the single-process form of one rank per GPU)
    std::vector<ncclComm_t> comms((size_t)opt.gpus);
    std::vector<int> devs((size_t)opt.gpus);
    for (int d = 0; d < opt.gpus; d++) devs[(size_t)d] = d;
    NK(ncclCommInitAll(comms.data(), opt.gpus, devs.data()));
    int nranks = 0; NK(ncclCommCount(comms[0], &nranks));

    std::vector<RankResult> results((size_t)opt.gpus);
    std::vector<std::thread> th;
    for (int r = 0; r < opt.gpus; r++) th.emplace_back(rank_main, r, std::cref(opt), comms[(size_t)r], &results);
    for (auto &t : th) t.join();
    for (auto &c : comms) NK(ncclCommDestroy(c));

    const double dt = results[0].seconds;        // (the all-reduced maximum: the same on every rank)
    const double total = (double)opt.channels * opt.block * opt.steps;
    std::printf("{\"host\": \"multi_gpu_host.cpp\", \"metric\": \"IQ MSamples/s demodulated to 48 kHz stereo\", \"value\": %.3f, \"unit\": \"MS/s\", "
                "\"n_gpus\": %d, \"rccl_ranks\": %d, \"steps\": %d, \"channels_total\": %d, \"block\": %d, \"ms_per_step\": %.4f, \"scaling\": \"strong\", "
                "\"gather_ms\": %.3f, \"broadcast_ms\": %.3f, \"self_check\": %s, \"channels_per_rank\": [",
                total / dt / 1e6, opt.gpus, nranks, opt.steps, opt.channels, opt.block, dt / opt.steps * 1e3, results[0].gather_ms, results[0].bcast_ms,
                results[0].self_check ? "true" : "false");
    for (int r = 0; r < opt.gpus; r++) std::printf("%s%d", r ? ", " : "", results[(size_t)r].channels);
    std::printf("]}\n");
    return results[0].self_check ? 0 : 1;
}
