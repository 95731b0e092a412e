#!/usr/bin/env python3
"""bench.py -- throughput of the FM demodulation hot path on MI355X (BASELINE.json metric:
"IQ MSamples/s demodulated to 48 kHz stereo").

A "step" = one pass of the whole chain (front-end FIR -> discriminator/pilot PLL/PSS -> audio FIR +
resampler) over one batch: `channels` independent FM channels x `block` complex samples each,
IQ already resident in HBM.  Default workload = BASELINE configs[3] -- 4096 independent channels with
configs[1]'s per-channel settings (stereo + PSS + de-emphasis + input FIR ON) -- which fits one GPU
(7.5 GB of IQ per step), so every rank runs all of it (weak scaling: 4096 channels per GPU); `--workload
shard512` is the same config split over eight GPUs.  Multi-GPU: one rank per GPU, channels sharded, no
data-path collective.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config4|shard512|config2|config3|config5]
"""
import argparse
import ctypes
import importlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# libfmx's event-driven stage-B layout uses five HIP streams; ROCm maps streams onto 4 hardware queues by default, so two
# of them would share one (the persistent layout has its own CU-masked queues).  Must be set before the HIP runtime starts (i.e. before torch is imported).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

INPUT_RATE = 2304000
BLOCK = 230400            # 0.1 s per channel per step; every synthetic tone is periodic in it
ALG_BYTES_STAGE_A = 8.0 + 8.0 / 12.0      # 8 B read + 8/12 B written per input sample
HBM_PEAK_GBPS = 8000.0                    # MI355X_MICROARCH.md: 8 TB/s spec

WORKLOADS = {
    # name: (channels per GPU, streams per GPU (0 = one per channel), description)
    "config4": (4096, 0, "configs[3] whole on each GPU: 4096 independent 2.304 MS/s channels, configs[1] settings "
                         "(stereo + PSS + de-emphasis 50us + input FIR 165 kHz + audio LPF 15 kHz)"),
    "shard512": (512, 0, "configs[3] split over 8 GPUs, one GPU's shard: 512 independent channels, configs[1] settings"),
    "config2": (1, 0, "configs[1]: 1 channel, stereo + PSS + de-emphasis + input FIR ON"),
    "config3": (256, 24, "configs[2]: 256 carriers in 24 wide-band IQ streams (11 per stream, 200 kHz raster)"),
    "config5": (2048, 0, "configs[4] per-GPU shard: 2048 channels (16384 over 8 GPUs), full chain incl. the RDS front end and "
                        "RDS_2 bit slicer (the synthetic MPX carries no 57 kHz sub-carrier: the slicer runs on noise, same work)"),
}


def synth_device(torch, channels, n, device, offsets_hz=None, seed=0):
    """[channels, n, 2] float32 stereo-FM IQ on the GPU, periodic in n (all tones multiples of 10 Hz)."""
    out = torch.empty((channels, n, 2), dtype=torch.float32, device=device)
    t = torch.arange(n, dtype=torch.float64, device=device) / INPUT_RATE
    g = torch.Generator(device="cpu").manual_seed(1234 + seed)
    step = 32
    for c0 in range(0, channels, step):
        c1 = min(channels, c0 + step)
        k = torch.arange(c0, c1, dtype=torch.float64)
        fl = (300 + 10 * ((37 * k) % 400)).to(device)[:, None]          # left tone, Hz
        fr = (500 + 10 * ((53 * k) % 400)).to(device)[:, None]
        ph = (torch.rand((c1 - c0, 3), generator=g, dtype=torch.float64) * 2 * np.pi).to(device)
        L = 0.5 * torch.sin(2 * np.pi * fl * t + ph[:, 0:1])
        R = 0.5 * torch.sin(2 * np.pi * fr * t + ph[:, 1:2])
        p19 = 2 * np.pi * 19000.0 * t + ph[:, 2:3]
        mpx = 0.45 * (L + R) + 0.10 * torch.sin(p19) + 0.45 * (L - R) * torch.sin(2 * p19)
        off = 0.0 if offsets_hz is None else torch.as_tensor(offsets_hz[c0:c1], dtype=torch.float64, device=device)[:, None]
        inc = 2 * np.pi * (75000.0 * mpx + off) / INPUT_RATE
        inc = inc - inc.mean(dim=1, keepdim=True) * (0.0 if offsets_hz is not None else 1.0)   # exact periodicity
        phase = torch.cumsum(inc, dim=1)
        out[c0:c1, :, 0] = (0.5 * torch.cos(phase)).float()
        out[c0:c1, :, 1] = (0.5 * torch.sin(phase)).float()
    return out


def usable_cores():
    """Host cores this process may really use: the affinity mask, cut down by a cgroup CPU quota if one is set."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return n


def cpu_baseline(seconds_budget=12.0):
    """The oracle (a port of the reference chain, oracle/fm_oracle.c) timed on this box's host cores:
    one channel per core, all usable cores busy, configs[1] settings -- the reference is single-threaded per
    channel.  Bounded sample: every worker demodulates 0.1 s blocks until the time budget is spent."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    cores = usable_cores()
    n = 16384 * 14                                    # ~0.1 s block, multiple of the reference's 16384
    iq = ol.synth_iq(n)
    chains = [ol.OracleChain(inputFilterBw=165000) for _ in range(cores)]
    L = ol.oracle()
    pcm = [np.zeros((n // 48 + 64, 2), np.float32) for _ in range(cores)]
    done = [0] * cores
    t0 = time.perf_counter()

    def work(i):
        while True:
            L.fmo_chain_process(chains[i].h, ol.fptr(iq), n, ol.fptr(pcm[i]), pcm[i].shape[0])
            done[i] += 1
            if time.perf_counter() - t0 >= seconds_budget:
                break

    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t0
    total = sum(done) * n
    return {"value": round(total / dt / 1e6, 3), "unit": "MS/s", "cores": cores, "kind": "port",
            "sample": "%d channels (one per usable core), %d blocks of %d samples in all within a %.0f s budget, "
                      "configs[1] settings, oracle/fm_oracle.c -O2" % (cores, sum(done), n, seconds_budget),
            "per_core_MSps": round(total / dt / 1e6 / cores, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=12)   # past the pilot-lock / PSS transition of the synthetic signal (calls 5-8)
    ap.add_argument("--workload", default="config4", choices=sorted(WORKLOADS))
    ap.add_argument("--channels", type=int, default=0, help="override channels per GPU")
    ap.add_argument("--block", type=int, default=BLOCK)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stride-pad", type=int, default=0,
                    help="complex samples of padding between consecutive streams in the IQ buffer (even)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libfmx has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)

    fmx_amd = importlib.import_module("sdr-j-fm_amd")
    m = fmx_amd.fmx
    channels, streams, desc = WORKLOADS[args.workload]
    if args.channels > 0:
        channels = args.channels
    n = args.block
    smap, offsets = None, None
    if streams:
        # configs[2]: channel c listens to carrier (c % 11) of stream c // 11 via set_localOscillator
        smap = [min(c // 11, streams - 1) for c in range(channels)]
    nstreams = streams if streams else channels
    f = fmx_amd.Fmx(channels, streams=streams, stream_of_channel=smap, device=local_rank, max_block=n)
    f.set_param(m.P_BANDWIDTH, 165000)
    f.set_param(m.P_LF_CUTOFF, 15000)
    f.set_param(m.P_DEEMPHASIS, 50)
    f.set_param(m.P_VOLUME_DB, -6.0)
    f.set_param(m.P_FM_MODE, 0)
    if args.workload == "config5":
        f.set_param(m.P_RDS_MODE, 2)
    if streams:
        for c in range(channels):
            f.set_param(m.P_LOCAL_OSCILLATOR, ((c % 11) - 5) * 200000, channel=c)

    if streams:
        # 11 carriers per wide-band stream on a 200 kHz raster
        iq = torch.zeros((nstreams, n, 2), dtype=torch.float32, device=device)
        for k in range(11):
            offs = [((k - 5) * 200000.0)] * nstreams
            iq += synth_device(torch, nstreams, n, device, offsets_hz=offs, seed=rank * 100 + k) * (1.0 / 3.5)
    else:
        iq = synth_device(torch, channels, n, device, seed=rank)
    stride = n + args.stride_pad
    if args.stride_pad:
        padded = torch.zeros((nstreams, stride, 2), dtype=torch.float32, device=device)
        padded[:, :n] = iq
        iq = padded
    frames_cap = n // 48 + 96
    pcm = torch.zeros((channels, frames_cap, 2), dtype=torch.float32, device=device)
    # The call runs on a stream of the caller's own (what a host application passes).  torch's current stream here is HIP's
    # default stream, i.e. handle 0 = NULL, which the C ABI reads as "the handle's own stream, ordered behind the default
    # stream" -- correct, but the default stream's implicit synchronisation with other streams costs ~0.4 ms per call.
    call_stream = torch.cuda.Stream(device=device)
    call_stream.wait_stream(torch.cuda.current_stream())
    stream = call_stream.cuda_stream

    def step():
        return f.process_device(iq.data_ptr(), stride, n, pcm.data_ptr(), frames_cap, hip_stream=stream)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    f.profile_enable(True)
    f.profile_read(reset=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frames = 0
    for _ in range(args.steps):
        frames += step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    prof = f.profile_read(reset=True)
    f.profile_enable(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        total = float(world) * channels * n * args.steps
        value = total / dt / 1e6
        launches = max(prof["launches"][0], 1)
        ms_a = prof["ms"][0] / launches
        # algorithmic bytes of ONE front-end launch: every channel reads its n samples (8 B) and writes n/12 (8 B)
        alg_bytes = ALG_BYTES_STAGE_A * channels * n
        achieved = alg_bytes / (ms_a * 1e-3) / 1e9 if ms_a > 0 else 0.0
        # HBM bytes per launch from the rocprofv3 PMC passes (FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE),
        # committed under profiles/; only valid for the configuration it was collected on.
        traffic = None
        try:
            pmcs = sorted(x for x in os.listdir(os.path.join(ROOT, "profiles")) if x.endswith("_front_pmc.json"))
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmcs[-1])))
            if pmc.get("channels") == channels and pmc.get("block") == n:
                traffic = pmc["front_kernel_hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        # practical ceiling next to the nominal peak: this box's streaming bandwidth, measured after the timed
        # region with the library's probe kernels (float2 copy; and stage A's own shape: read 12, write 1)
        measured = {}
        try:
            import ctypes as C
            L = fmx_amd.load_library()
            L.fmx_debug_stream_bandwidth.argtypes = [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_double)]
            for mode, key in ((0, "copy_GBps"), (1, "read12_write1_GBps")):
                g = C.c_double()
                if L.fmx_debug_stream_bandwidth(local_rank, mode, 1 << 30, 10, C.byref(g)) == 0:
                    measured[key] = round(g.value, 1)
        except Exception as e:      # the probe is a diagnostic; the bench line does not depend on it
            measured = {"error": str(e)}
        out = {
            "metric": "IQ MSamples/s demodulated to 48 kHz stereo",
            "value": round(value, 3), "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "description": desc, "channels_per_gpu": channels,
                       "streams_per_gpu": nstreams, "block_samples_per_channel": n,
                       "realtime_channels_equiv": round(value / 2.304, 1),
                       "pcm_frames_per_channel_per_step": frames // max(args.steps, 1), "parallelism": "channels sharded, 1 rank/GPU"},
            "roofline": {"bound": "hbm", "kernel": "fmx::front_kernel (input FIR stage)", "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": traffic, "avg_launch_ms": round(ms_a, 4),
                         "algorithmic_bytes_per_launch": alg_bytes, "measured_stream_bandwidth": measured,
                         "frac_of_measured": (round(achieved / measured["read12_write1_GBps"], 4)
                                              if measured.get("read12_write1_GBps") else None)},
            "kernels_ms_per_step": {"front_fir": round(prof["ms"][0] / launches, 4),
                                    "demod_pilot_pss": round(prof["ms"][1] / launches, 4),
                                    "audio_fir_resample": round(prof["ms"][2] / launches, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
