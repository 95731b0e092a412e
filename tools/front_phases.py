"""Per-phase shader cycles of fmx::front_kernel (diagnostic hook fmx_debug_phase_cycles)."""
import importlib, ctypes as C, sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = importlib.import_module("sdr-j-fm_amd"); m = pkg.fmx
ch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = 230400
f = pkg.Fmx(ch, max_block=n)
for p, v in ((m.P_BANDWIDTH, 165000), (m.P_LF_CUTOFF, 15000), (m.P_DEEMPHASIS, 50), (m.P_VOLUME_DB, -6.0)): f.set_param(p, v)
dev = torch.device('cuda', 0)
iq = bench.synth_device(torch, ch, n, dev)
pcm = torch.zeros((ch, n // 48 + 96, 2), dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3): f.process_device(iq.data_ptr(), n, n, pcm.data_ptr(), n // 48 + 96, hip_stream=s)
torch.cuda.synchronize()
L = pkg.load_library()
L.fmx_debug_phase_cycles.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_ulonglong)]
L.fmx_debug_phase_cycles(f.h, 1, None)
K = 3
for _ in range(K): f.process_device(iq.data_ptr(), n, n, pcm.data_ptr(), n // 48 + 96, hip_stream=s)
out = (C.c_ulonglong * 96)()
L.fmx_debug_phase_cycles(f.h, 0, out)
names = ["prologue+first load", "scatter to LDS", "DC removal (scan, carry wait) + mix", "history hand-off (waits)", "prefetch + FIR + partial sums", "reduce + store", "epilogue"]
tiles = ch * K * (n / 1536.0 / 4)     # wave 0 of every workgroup handles a quarter of the 1536-sample tiles
tot = sum(out[:8])
for k, nm in enumerate(names):
    print(f"{nm:24s} {out[k]/tiles:9.0f} cycles/tile  {100*out[k]/tot:5.1f}%")
print("total cycles/tile", tot / tiles)
print("of which waiting for another wave: RF DC carry %.0f, next image free %.0f, history in %.0f cycles/tile" % (out[40] / tiles, out[41] / tiles, out[42] / tiles))
print("stage-B block paths: pss_acc steady/idle/replay =", out[8], out[9], out[10], " lock closed-form/replay =", out[11], out[12])
