import numpy as np, time
# (int)((double)q + 0.5)  ==  (int)(q + 0.49999997f) in f32, for every f32 q in [-0, 8192]?
c = np.float32(0.49999997)
assert c == np.float32(0.5) - np.float32(2.0 ** -25)
hi = int(np.float32(8192.0).view(np.uint32))
bad = 0; t0 = time.time()
step = 1 << 24
for b0 in range(0, hi + 1, step):
    bits = np.arange(b0, min(b0 + step, hi + 1), dtype=np.uint32)
    q = bits.view(np.float32)
    ref = (q.astype(np.float64) + 0.5).astype(np.int32)          # C cast: truncation
    new = (q + c).astype(np.int32)
    nb = int((ref != new).sum()); bad += nb
    if nb: print("mismatch at", q[ref != new][:5], ref[ref != new][:5], new[ref != new][:5])
q = np.array([-0.0], np.float32)
assert int((q.astype(np.float64) + 0.5).astype(np.int32)[0]) == int((q + c).astype(np.int32)[0]) == 0
print("checked %d values, mismatches %d, %.0f s" % (hi + 1, bad, time.time() - t0))
