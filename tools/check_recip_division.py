#!/usr/bin/env python3
"""Container tool (CPU, numpy): how often does  q = num*r; q += (num - den*q)*r  (exact residual, fused multiply-adds) differ
from IEEE fp32 division, with r the rounded reciprocal of den or one ulp beside it (v_rcp_f32 is a 1-ulp instruction), after
0, 1, 2 and 3 corrections?  (csrc/s2d_raster.hip div_by_recip applies ONE.)

Operands as in the backward blend (main.cpp:627-628): den = fl(fl(1 - alpha) + 1e-15) with alpha = G * o in [0, 1] -- so den is
1e-15 (alpha == 1) or a multiple of 2^-24 in [2^-24, 1] -- and num = S = final - colour, |num| <~ 1.  Three populations:
  uniform  : alpha uniform in [0, 1), num uniform in (-1, 1)                      (round 3's check, 2e7 trials then)
  log      : 1 - alpha log-uniform over [2^-24, 1] plus alpha == 1 exactly; num log-uniform in magnitude over [1e-8, 1]
  midpoint : DIRECTED -- for a random den, num is chosen so that num / den lies next to the midpoint of two neighbouring
             floats (the quotients a correction step is most likely to round the other way)
The fused operations are evaluated in 64-bit-mantissa arithmetic (the product of two fp32 values is exact there, the nearly
cancelling sum as well) and rounded once to fp32.
usage: check_recip_division.py [trials per population, default 4e7] [processes, default 8]"""
import multiprocessing as mp
import sys

import numpy as np

f32, ld = np.float32, np.longdouble
NAMES = ("r = RN(1/den)", "r = RN(1/den) + 1 ulp", "r = RN(1/den) - 1 ulp")


def fma(a, b, c):
    return (a.astype(ld) * b.astype(ld) + c.astype(ld)).astype(f32)


def operands(kind, n, rng):
    if kind == "uniform":
        alpha = rng.random(n).astype(f32)
        num = ((rng.random(n) - 0.5) * 2).astype(f32)
    elif kind == "log":
        one_minus = np.exp(rng.uniform(np.log(2.0 ** -24), 0.0, n)).astype(f32)
        alpha = (f32(1.0) - one_minus).astype(f32)
        alpha[rng.random(n) < 0.02] = f32(1.0)                       # the pixel-saturating case: den == 1e-15
        num = (np.exp(rng.uniform(np.log(1e-8), 0.0, n)) * rng.choice([-1.0, 1.0], n)).astype(f32)
    else:
        alpha = rng.random(n).astype(f32)
        num = None
    den = ((f32(1.0) - alpha).astype(f32) + f32(1e-15)).astype(f32)
    if num is None:  # num / den next to a rounding midpoint: q0 a random float in [2^-6, 1), mid = q0 + ulp/2, num = RN(mid * den)
        q0 = np.exp(rng.uniform(np.log(2.0 ** -6), 0.0, n)).astype(f32)
        mid = (q0.astype(np.float64) + np.nextafter(q0, f32(2.0)).astype(np.float64)) * 0.5
        num = (mid * den.astype(np.float64)).astype(f32)
        num *= rng.choice([-1.0, 1.0], n).astype(f32)
    return num, den


def work(job):
    kind, n, seed = job
    rng = np.random.default_rng(seed)
    num, den = operands(kind, n, rng)
    q_ieee = (num / den).astype(f32)  # numpy's fp32 division is correctly rounded
    r0 = (f32(1.0) / den).astype(f32)
    out = {}
    for name, r in zip(NAMES, (r0, np.nextafter(r0, f32(np.inf)), np.nextafter(r0, f32(-np.inf)))):
        q = (num * r).astype(f32)
        res = [int((q != q_ieee).sum())]
        first = None
        for k in range(3):
            q = fma(fma(-den, q, num), r, q)
            res.append(int((q != q_ieee).sum()))
            if k == 0:
                first = q.copy()
            if k == 1:
                res.append(int((q != first).sum()))  # quotients the SECOND correction still changes
        out[name] = res
    return kind, n, out


def run(total=40_000_000, procs=8, batch=2_000_000):
    """{population: {"n": trials, name of r: [differ after 0, 1, 2 corrections, changed by the second, differ after 3]}}"""
    if np.finfo(ld).nmant < 63:
        raise SystemExit("numpy.longdouble has no 64-bit mantissa on this machine")
    batch = min(batch, total)
    jobs = [(kind, batch, 1000 * i + k) for i, kind in enumerate(("uniform", "log", "midpoint")) for k in range(max(1, total // batch))]
    tot = {}
    pool = mp.Pool(procs) if procs > 1 else None
    try:
        for kind, n, out in (pool.imap_unordered(work, jobs) if pool else map(work, jobs)):
            t = tot.setdefault(kind, {"n": 0})
            t["n"] += n
            for name, res in out.items():
                acc = t.setdefault(name, [0] * 5)
                for i in range(5):
                    acc[i] += res[i]
    finally:
        if pool:
            pool.close()
    return tot


def main():
    total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 40_000_000
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    tot = run(total, procs)
    for kind in ("uniform", "log", "midpoint"):
        t = tot[kind]
        print("population %-8s  %d trials" % (kind, t["n"]))
        for name in NAMES:
            a = t[name]
            print("  %-22s differ from IEEE division after 0 / 1 / 2 / 3 corrections: %d / %d / %d / %d;  changed by the second correction: %d"
                  % (name, a[0], a[1], a[2], a[4], a[3]))
    return tot


if __name__ == "__main__":
    main()
