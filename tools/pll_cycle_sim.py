#!/usr/bin/env python3
"""CPU model (numpy) of the cycle-parallel exact solver of the pilot PLL (csrc/fmx_stageb.hip): the sequential f32 trajectory of
pilot-recover.cpp:54-61 over a 1536-sample segment WITHOUT a 1536-step dependent chain.

Once per pilot period (~10.1 samples) the phase enters [4, 8), where every f32 is a multiple of U = 2^-21 -- and U is a multiple of every
ulp the phase has anywhere else in [0, 2 pi).  So if two trajectories differ by k U at such an "anchor" sample and make the same decisions
(table entry of every sample, where the step wraps, where a binade is crossed), they differ by exactly k U at every later sample: the
rounding of every addition commutes with a shift by a multiple of its own grid.  A guess good to a few U (Newton's) therefore splits the
segment into ~152 runs from anchor to anchor that can be evaluated side by side -- the reference's own step, sample by sample, ten steps
deep --; the runs' end points miss the next anchors of the guess by integers d_c (in U), whose prefix sums K_c are the shifts of the TRUE
trajectory against the guess -- provided no decision changes under the shift, which is checked by evaluating the runs again from the
shifted anchors and repeating until every run ends on the next run's start (a chain with that property that starts at x0 IS the sequential
trajectory).  Prints passes per segment and checks bit-identity with the sequential loop.
usage: python tools/pll_cycle_sim.py [creep.npz] [segments]"""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pll_polish_sim import C, GAIN, N, OMEGA, P32, TAB, TWO_PI, W, constrain, f32, sequential, wrap_diff  # noqa: E402

U = 2.0 ** -21


def step1(x, p5):
    """one exact step on arrays (the reference's f32 expression)"""
    idx = (x.astype(np.float64) * C).astype(np.int64) % N
    perr = (p5 * TAB[idx]).astype(np.float32)
    t = (x + (perr * GAIN).astype(np.float32)).astype(np.float32)
    return constrain((t + OMEGA).astype(np.float32))


def newton_guess(x0, q):
    g = (q * GAIN).astype(np.float32)
    ramp = np.float64(x0) + np.arange(W) * np.float64(OMEGA)
    cor = np.zeros(W)
    for _ in range(2):
        sv = g.astype(np.float64) * np.sin(ramp + cor)
        cor = np.concatenate([[0.0], np.cumsum(sv)[:-1]])
    ph = constrain(np.mod(ramp + cor, TWO_PI).astype(np.float32)); ph[0] = x0
    nx = step1(ph, q)
    d64 = wrap_diff(nx, ph)
    Pn = np.concatenate([[0.0], np.cumsum(d64)[:-1]])
    d = wrap_diff(np.mod(np.float64(x0) + Pn, TWO_PI), ph)
    c = g.astype(np.float64) * np.cos(ph.astype(np.float64))
    S = np.zeros(W); acc = 0.0
    for j in range(W):
        S[j] = acc; acc = (1 + c[j]) * acc + c[j] * d[j]
    ph = constrain(np.mod(np.float64(x0) + Pn + S, TWO_PI).astype(np.float32)); ph[0] = x0
    return ph


def cycle_solve(x0, q, guess, max_pass=8):
    w = len(q)
    anchor = np.zeros(w, bool)
    anchor[0] = True
    anchor[1:] = (guess[1:] >= 4.0) & (guess[:-1] < 4.0)
    anc = np.flatnonzero(anchor)
    nc = len(anc)
    ends = np.concatenate([anc[1:], [w]])
    lens = ends - anc
    L = lens.max()
    K = np.zeros(nc, np.int64)                      # shift of run c's start against the guess, in U (run 0 starts at x0 itself)
    out = np.zeros(w, np.float32)
    for p in range(max_pass):
        x = guess[anc].astype(np.float64) + K * U
        x = x.astype(np.float32); x[0] = x0
        for i in range(L):
            act = i < lens
            pos = np.minimum(anc + i, w - 1)
            out[pos[act]] = x[act]
            nx = step1(x, q[pos])
            x = np.where(act, nx, x)
        # x = value at the next anchor (or behind the segment); the defect against where the next run started
        nxt = np.concatenate([guess[anc[1:]].astype(np.float64) + K[1:] * U, [0.0]])
        dd = (x.astype(np.float64) - nxt) / U
        dd[-1] = 0
        if np.any(dd != np.round(dd)):
            return None, p + 1, "off grid"
        di = np.round(dd).astype(np.int64)
        if not di[:-1].any():
            return out, p + 1, x[-1]
        K[1:] += np.cumsum(di[:-1])
    return None, max_pass, "no convergence"


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/creep.npz"
    nseg = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    z = np.load(path)
    dem = z["dem_o"][:W * nseg]
    p5 = (f32(5) * dem).astype(np.float32)
    xseq, _, _ = sequential(p5)
    passes, bad = [], 0
    x0 = f32(0)
    for s in range(0, W * nseg, W):
        q = p5[s:s + W]
        guess = newton_guess(x0, q)
        out, np_, end = cycle_solve(x0, q, guess)
        passes.append(np_)
        ref = xseq[s:s + W]
        if out is None or not np.array_equal(out.view(np.int32), ref.view(np.int32)):
            bad += 1
            print("segment %d: %s" % (s // W, end if out is None else "differs from the sequential loop"))
        x0 = step1(xseq[s + W - 1:s + W], p5[s + W - 1:s + W])[0]          # (the exact state carries on)
        ge = wrap_diff(guess, ref)
    print("segments %d: passes mean %.2f max %d (each = one ten-step run of all cycles side by side); not bit-identical or not converged: %d"
          % (nseg, np.mean(passes), max(passes), bad))
    print("guess error of the last segment: rms %.2e max %.2e rad = %.1f U" % (np.sqrt(np.mean(ge ** 2)), np.abs(ge).max(), np.abs(ge).max() / U))


if __name__ == "__main__":
    main()
