"""Per-phase shader cycles of fmx::f3::front3_kernel, wave 0 of every channel (a -DF3_TICKS build: tools/build_variant.sh ticks fmx_front3 -DF3_TICKS,
run with FMX_LIB=sdr-j-fm_amd/lib/ab/libfmx_ticks.so)."""
import importlib, ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
names = ["wait: next wave's FIR done", "scatter (incl. tile loads)", "prefetch issue", "wait: previous scatter", "FIR", "quad sums + DC sums + scan", "wait: carry",
         "DC rest + bpermute", "output", "loop"]
tiles = ch * K * (n / 1536.0 / 6)
tot = sum(out[48:58])
for k, nm in enumerate(names):
    print(f"{nm:30s} {out[48+k]/tiles:9.0f} cycles/tile  {100*out[48+k]/max(tot,1):5.1f}%")
print("total cycles/tile", tot / tiles)
