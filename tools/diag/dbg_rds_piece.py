import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import oracle_lib as ol
import test_gpu_round4 as T
pkg = importlib.import_module("sdr-j-fm_amd"); M = pkg.fmx
n = 921600
iq = np.stack([ol.synth_iq(n, rds=1, rdsLevel=0.05, rdsBitsSeed=sd) for sd in (5, 6)])
res = []
for blocks in ([n], [230400] * 4, [383988, 383988, 153624], [300000, 300000, 321600]):
    f = T._handle(pkg, 2, 2, n, 0, 2, [0, 0]); f.set_param(M.P_RDS_MODE, 2)
    pos = 0
    for b in blocks:
        f.process_host(iq[:, pos:pos + b]); pos += b
    bits = [f.rds_bits(c, 8192) for c in range(2)]
    res.append(bits); print(blocks, [len(b) for b in bits])
for k in range(1, len(res)):
    for c in range(2):
        a, b = res[0][c], res[k][c]
        m = min(len(a), len(b))
        d = np.flatnonzero(a[:m] != b[:m])
        print(k, c, len(a), len(b), "mismatches", len(d), d[:10])
