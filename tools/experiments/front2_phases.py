"""Per-phase shader cycles of fmx::f2::front2_kernel (diagnostic hook fmx_debug_phase_cycles): producer and consumer wave of a pair."""
import importlib, ctypes as C, sys, torch
sys.path.insert(0, '/root/repo')
import bench
pkg = importlib.import_module("sdr-j-fm_amd"); m = pkg.fmx
ch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = 230400
f = pkg.Fmx(ch, max_block=n)
for p, v in ((m.P_BANDWIDTH, 165000), (m.P_LF_CUTOFF, 15000), (m.P_DEEMPHASIS, 50), (m.P_VOLUME_DB, -6.0)): f.set_param(p, v)
dev = torch.device('cuda', 0)
iq = bench.synth_device(torch, ch, n, dev)
pcm = torch.zeros((ch, n // 48 + 96, 2), dtype=torch.float32, device=dev)
s = torch.cuda.current_stream().cuda_stream
for _ in range(3): f.process_device(iq.data_ptr(), n, n, pcm.data_ptr(), n // 48 + 96, hip_stream=s)
torch.cuda.synchronize()
L = pkg.load_library()
L.fmx_debug_phase_cycles.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_ulonglong)]
L.fmx_debug_phase_cycles(f.h, 1, None)
K = 3
for _ in range(K): f.process_device(iq.data_ptr(), n, n, pcm.data_ptr(), n // 48 + 96, hip_stream=s)
out = (C.c_ulonglong * 96)()
L.fmx_debug_phase_cycles(f.h, 0, out)
names = ["P: wait loads + scatter", "P: DC pass + write-back + post", "P: wait for consumer", "P: history write",
         "C: wait for producer", "C: matrix FIR + post", "C: exchange + store"]
tiles = ch * K * (n / 1536.0)
for k, nm in enumerate(names):
    print(f"{nm:34s} {out[k]/tiles:9.0f} cycles/tile")
if sum(out[16:20]):
    for k, nm in zip(range(16, 20), ["  DC: LDS read-back arrived", "  DC: sums + scan + carry", "  DC: correction + mix", "  DC: write-back drained"]): print(f"{nm:34s} {out[k]/tiles:9.0f} cycles/tile (of the DC pass)")
print("producer total", sum(out[:4]) / tiles, " consumer total", sum(out[4:7]) / tiles)
