"""Run a few calls at a given channel count and synchronise through the library (reports a stalled stage-B pipeline)."""
import importlib, sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
pkg = importlib.import_module("sdr-j-fm_amd"); m = pkg.fmx
ch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = 230400
f = pkg.Fmx(ch, max_block=n)
for p, v in ((m.P_BANDWIDTH, 165000), (m.P_LF_CUTOFF, 15000), (m.P_DEEMPHASIS, 50), (m.P_VOLUME_DB, -6.0)): f.set_param(p, v)
dev = torch.device('cuda', 0)
iq = bench.synth_device(torch, ch, n, dev)
pcm = torch.zeros((ch, n // 48 + 96, 2), dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream
for k in range(K):
    t0 = time.perf_counter()
    f.process_device(iq.data_ptr(), n, n, pcm.data_ptr(), n // 48 + 96, hip_stream=s)
    try:
        f.synchronize()
    except Exception as e:
        print("call", k, "FAILED:", e)
        import ctypes as C, numpy as np
        L = pkg.load_library()
        buf = (C.c_int32 * 4096)(); nn = C.c_int32()
        L.fmx_debug_sync_dump.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32)]
        L.fmx_debug_sync_dump(f.h, buf, 4096, C.byref(nn))
        a = np.frombuffer(buf, np.int32, nn.value)
        print("abort/info", a[:5])
        sn = a[16:80]
        print("at abort: cnt_disc", sn[0:16]); print("          cnt_fir ", sn[16:32]); print("          cnt_mix ", sn[32:48])
        print("          prog[role][0]", sn[48:53], " prog[role][last]", sn[56:61])
        continue
    print(f"call {k}: {1e3 * (time.perf_counter() - t0):.3f} ms")
# async: enqueue K calls back to back, synchronise once
import time as _t
torch.cuda.synchronize(); t0 = _t.perf_counter()
for k in range(K): f.process_device(iq.data_ptr(), n, n, pcm.data_ptr(), n // 48 + 96, hip_stream=s)
t1 = _t.perf_counter(); f.synchronize(); t2 = _t.perf_counter()
print(f"async: enqueue {1e3*(t1-t0)/K:.3f} ms/call, end-to-end {1e3*(t2-t0)/K:.3f} ms/call")
