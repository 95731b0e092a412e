#!/usr/bin/env python3
"""CPU model (numpy) of the pilot-PLL segment solver of csrc/fmx_stageb.hip on a recorded demodulator output (gpurun_out/creep.npz,
tools/diag/dbg_creep.py): Newton's method on 1536-sample segments as the kernel runs it, followed by K polishing rounds
x <- fl (x0 + prefix sum of step (x) - x) (no Newton correction: the fixed point of that map is the sequential f32 loop), against the
sequential loop of pilot-recover.cpp:54-61.  Prints, per K: the phase error (rad), the lock-metric error, the last sample below the
0.07 threshold in front of the first lock, and how many segments still hold a defect (a sample whose successor is not its step).
usage: python tools/pll_polish_sim.py [creep.npz] [max K]"""
import sys

import numpy as np

N = 192000
TAB = np.sin(2 * np.pi * np.arange(N) / N).astype(np.float32)
C = N / (2 * np.pi)
f32 = np.float32
OMEGA = f32(f32(f32(19000) / f32(192000)) * (2 * np.pi))
GAIN = f32(10 * (2 * np.pi) / 192000)
P32 = f32(6.2831855)
TWO_PI = 2 * np.pi
W = 1536


def constrain(v):
    v = np.asarray(v, np.float32)
    out = v.copy()
    bad = ~((v >= 0) & (v < P32))
    out[bad] = np.mod(v[bad].astype(np.float64), TWO_PI).astype(np.float32)
    return out


def step(ph, p5):
    idx = (ph.astype(np.float64) * C).astype(np.int64) % N
    perr = (p5 * TAB[idx]).astype(np.float32)
    t = (ph + (perr * GAIN).astype(np.float32)).astype(np.float32)
    return constrain((t + OMEGA).astype(np.float32)), t, TAB[idx]


def sequential(p5):
    n = len(p5)
    x = np.zeros(n + 1, np.float32); cur = np.zeros(n, np.float32); osc = np.zeros(n, np.float32)
    for j in range(n):
        nx, t, o = step(x[j:j + 1], p5[j:j + 1])
        x[j + 1] = nx[0]; cur[j] = t[0]; osc[j] = o[0]
    return x[:n], cur, osc


def wrap_diff(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return d - np.round(d / TWO_PI) * TWO_PI


def solve(p5, polish, tol=2e-3):
    n = len(p5)
    xs = f32(0)
    phs, curs, oscs = [], [], []
    defects = 0
    rounds_n = []
    for s in range(0, n - W + 1, W):
        q = p5[s:s + W]
        g = (q * GAIN).astype(np.float32)
        x0 = xs
        # seed: ramp + two rounds of x = ramp + sum g sin x (f32, in turns)
        ramp = np.float64(x0) + np.arange(W) * np.float64(OMEGA)
        cor = np.zeros(W)
        for _ in range(2):
            sv = g.astype(np.float64) * np.sin(ramp + cor)
            cor = np.concatenate([[0.0], np.cumsum(sv)[:-1]])
        ph = np.mod(ramp + cor, TWO_PI).astype(np.float32); ph[0] = x0
        ph = constrain(ph)
        it = 0
        while True:
            nx, t, o = step(ph, q)
            d64 = wrap_diff(nx, ph)                         # increments incl. the wrap taken out
            Pn = np.concatenate([[0.0], np.cumsum(d64)[:-1]])
            d = wrap_diff(np.mod(np.float64(x0) + Pn, TWO_PI), ph)
            c = g.astype(np.float64) * np.cos(ph.astype(np.float64))
            S = np.zeros(W)
            acc = 0.0
            for j in range(W):                               # S[j+1] = (1 + c) S + c d
                S[j] = acc
                acc = (1 + c[j]) * acc + c[j] * d[j]
            upd = np.abs(d + S).max()
            ph = constrain(np.mod(np.float64(x0) + Pn + S, TWO_PI).astype(np.float32)); ph[0] = x0
            it += 1
            if upd < tol or it >= 10:
                break
        rounds_n.append(it)
        for _ in range(polish):
            nx, t, o = step(ph, q)
            d64 = wrap_diff(nx, ph)
            Pn = np.concatenate([[0.0], np.cumsum(d64)[:-1]])
            ph = constrain(np.mod(np.float64(x0) + Pn, TWO_PI).astype(np.float32)); ph[0] = x0
        nx, t, o = step(ph, q)
        if np.any(nx[:-1].view(np.int32) != ph[1:].view(np.int32)):
            defects += 1
        phs.append(ph); curs.append(t); oscs.append(o)
        xs = nx[-1]
    return np.concatenate(phs), np.concatenate(curs), np.concatenate(oscs), defects, np.mean(rounds_n)


def lock_metric(osc, p5):
    from scipy.signal import lfilter
    quad = np.diff(np.concatenate([[0.0], osc.astype(np.float64)])) / np.float64(OMEGA)
    x = (1.0 / 3000.0) * (-quad * p5.astype(np.float64))
    return lfilter([1.0], [1.0, -(1.0 - 1.0 / 3000.0)], x)


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/creep.npz"
    kmax = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    z = np.load(path)
    dem = z["dem_o"][:W * 120]                               # 0.96 s: the first lock of the creeping-pilot signal sits at sample 23751
    p5 = (f32(5) * dem).astype(np.float32)
    xseq, cseq, oseq = sequential(p5)
    mseq = lock_metric(oseq, p5)
    n = len(p5) // W * W
    below = np.flatnonzero(mseq[:n] <= 0.07)
    print("sequential: last sample below 0.07 within the first 60000: %d" % below[below < 60000][-1])
    for k in range(kmax + 1):
        ph, cur, osc, defects, rn = solve(p5, k)
        e = wrap_diff(ph, xseq[:n])
        m = lock_metric(osc, p5[:n])
        below = np.flatnonzero(m <= 0.07)
        print("polish %d: Newton rounds %.2f; phase error rms %.2e max %.2e rad; metric error rms %.2e max %.2e; last below %d; segments with a defect %d of %d"
              % (k, rn, np.sqrt(np.mean(e * e)), np.abs(e).max(), np.sqrt(np.mean((m - mseq[:n]) ** 2)), np.abs(m - mseq[:n]).max(),
                 below[below < 60000][-1], defects, n // W))


if __name__ == "__main__":
    main()
