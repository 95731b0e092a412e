"""Host-side cost of enqueueing one call (no synchronisation inside the timed region) vs the GPU time of the call."""
import importlib, sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
pkg = importlib.import_module("sdr-j-fm_amd"); m = pkg.fmx
ch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
use_null = (len(sys.argv) > 2 and sys.argv[2] == "null")
n = 230400
f = pkg.Fmx(ch, max_block=n)
for p, v in ((m.P_BANDWIDTH, 165000), (m.P_LF_CUTOFF, 15000), (m.P_DEEMPHASIS, 50), (m.P_VOLUME_DB, -6.0)): f.set_param(p, v)
if len(sys.argv) > 3: f.set_param(m.P_FM_DECODER, int(sys.argv[3]))
dev = torch.device('cuda', 0)
iq = bench.synth_device(torch, ch, n, dev)
pcm = torch.zeros((ch, n // 48 + 96, 2), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream() if use_null else torch.cuda.Stream()
s = st.cuda_stream
for _ in range(3): f.process_device(iq.data_ptr(), n, n, pcm.data_ptr(), n // 48 + 96, hip_stream=s)
torch.cuda.synchronize()
K = 6
t0 = time.perf_counter()
for _ in range(K): f.process_device(iq.data_ptr(), n, n, pcm.data_ptr(), n // 48 + 96, hip_stream=s)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"channels {ch} stream {'null' if use_null else 'own'}: enqueue {1e3*(t1-t0)/K:.3f} ms/call, total {1e3*(t2-t0)/K:.3f} ms/call")
