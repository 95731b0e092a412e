#!/usr/bin/env python3
"""CPU model (numpy) of the pilot-PLL fixed-point iteration of the fused stage-B kernel (csrc/fmx_stageb.hip): rounds per
1536-sample segment and the error against the sequential f32 loop of pilot-recover.cpp:54-61, for
  * the iteration the kernel runs: the reference's own f32 step evaluated on a guess, d = step(x) - x summed in f64;
  * the same run to the exact fixed point (bit-identical to the sequential loop, hundreds of rounds);
  * an evaluation of the phase in higher precision (x0 + j omega + sum of corrections): it misses the reference by the
    standing phase offset that the f32 roundings of `phase + omega` produce in the loop (~5e-4 rad).
usage: python tools/pll_fixed_point.py [noise sigma] [tolerance]"""
import sys

import numpy as np

N = 192000
TAB = np.sin(2 * np.pi * np.arange(N) / N).astype(np.float32)          # SinCos table, sine column (sincos.cpp:45-54)
C = N / (2 * np.pi)
f32 = np.float32
OMEGA = f32(f32(f32(19000) / f32(192000)) * (2 * np.pi))                # OMEGA_PILOT fm-processor.cpp:34
GAIN = f32(10 * (2 * np.pi) / 192000)                                   # fm-processor.cpp:79
P32 = f32(6.2831855)
TWO_PI = 2 * np.pi
W = 1536


def constrain(v):
    v = np.asarray(v, np.float32)
    out = v.copy()
    bad = ~((v >= 0) & (v < P32))
    out[bad] = np.mod(v[bad].astype(np.float64), TWO_PI).astype(np.float32)
    return out


def step(ph, p):
    """one sample of getPilotPhase on an array of states: returns the next state"""
    idx = (ph.astype(np.float64) * C).astype(np.int64) % N
    perr = (p * TAB[idx]).astype(np.float32)
    t = (ph + (perr * GAIN).astype(np.float32)).astype(np.float32)
    return constrain((t + OMEGA).astype(np.float32))


def main():
    noise = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
    tol = float(sys.argv[2]) if len(sys.argv) > 2 else 3e-5
    n = W * 24
    t = np.arange(n) / 192000.0
    rng = np.random.default_rng(1)
    dem = (0.4 * np.sin(2 * np.pi * 1000 * t) + 0.2 * np.sin(2 * np.pi * 3100 * t) + 0.1 * np.sin(2 * np.pi * 19000 * t + 0.3)
           + 0.3 * np.sin(2 * np.pi * 700 * t) * np.sin(2 * (2 * np.pi * 19000 * t + 0.3)) + noise * rng.standard_normal(n)).astype(np.float32)
    p5 = (f32(5) * dem).astype(np.float32)
    x = np.zeros(n + 1, np.float32)
    for j in range(n):
        x[j + 1] = step(x[j:j + 1], p5[j:j + 1])[0]

    def iterate(tolerance, max_rounds, carry):
        xs, rounds, traj = f32(0), [], []
        for s in range(0, n, W):
            x0 = xs if carry else x[s]
            r = np.float64(x0) + np.arange(W) * np.float64(OMEGA)
            ph = (r - np.floor(r / TWO_PI) * TWO_PI).astype(np.float32)
            ph[0] = x0
            for it in range(max_rounds):
                g = constrain(ph)
                d = step(g, p5[s:s + W]).astype(np.float64) - g.astype(np.float64)
                cs = np.cumsum(d)
                nph = (np.float64(x0) + np.concatenate([[0.0], cs[:-1]])).astype(np.float32)
                dd = np.abs(nph - ph)
                dd = np.minimum(dd, np.abs(dd - P32))
                same = np.array_equal(nph.view(np.int32), ph.view(np.int32))
                ph = nph
                if (tolerance > 0 and dd.max() < tolerance) or (tolerance == 0 and same):
                    break
            rounds.append(it + 1)
            traj.append(g)
            xe = np.float64(x0) + cs[-1]
            xs = f32(xe - np.floor(xe / TWO_PI) * TWO_PI)
        return rounds, np.concatenate(traj)

    def err(tr):
        e = np.abs(tr.astype(np.float64) - x[:n])
        return np.minimum(e, np.abs(e - TWO_PI))

    r, tr = iterate(tol, 64, True)
    e = err(tr)
    print("kernel's iteration (tolerance %.0e): rounds per segment mean %.2f max %d; against the sequential loop max %.2e rms %.2e rad"
          % (tol, np.mean(r), max(r), e.max(), np.sqrt(np.mean(e * e))))
    r, tr = iterate(0.0, W + 8, False)
    print("run to the exact fixed point: rounds per segment mean %.0f; bit-identical to the sequential loop: %s"
          % (np.mean(r[:4]), np.array_equal(tr[:4 * W].view(np.int32), x[:4 * W].view(np.int32))))
    # higher-precision evaluation: phase = x0 + j omega + sum of the corrections, no f32 rounding of the running phase
    S = np.zeros(n)
    for _ in range(8):
        ph = np.mod(np.arange(n) * np.float64(OMEGA) + S, TWO_PI)
        c = (p5 * TAB[(ph * C).astype(np.int64) % N]).astype(np.float32).astype(np.float64) * np.float64(GAIN)
        S = np.concatenate([[0.0], np.cumsum(c)[:-1]])
    e = err(np.mod(np.arange(n) * np.float64(OMEGA) + S, TWO_PI).astype(np.float32))
    print("phase evaluated in f64 instead of stepped in f32: against the sequential loop max %.2e rms %.2e rad (the loop's standing offset)"
          % (e.max(), np.sqrt(np.mean(e[n // 2:] ** 2))))


if __name__ == "__main__":
    main()
