#!/usr/bin/env python3
"""Container tool (CPU, numpy): how often does  q = num*r; q += (num - den*q)*r  (exact residual, fused multiply-adds) differ
from IEEE fp32 division, with r the rounded reciprocal of den or one ulp beside it (v_rcp_f32 is a 1-ulp instruction), after
0, 1, 2 and 3 corrections?  Operands as in the backward blend: den = 1 - alpha + 1e-15 in (0.01, 1], |num| < 1
(csrc/s2d_raster.hip, div_by_recip).  The fused operations are evaluated in 64-bit-mantissa arithmetic (the product of two
fp32 values is exact there, and the nearly cancelling sum as well) and rounded once to fp32."""
import numpy as np

f32, ld = np.float32, np.longdouble


def fma(a, b, c):
    return (a.astype(ld) * b.astype(ld) + c.astype(ld)).astype(f32)


def main(n=4_000_000, reps=5):
    if np.finfo(ld).nmant < 63:
        raise SystemExit("numpy.longdouble has no 64-bit mantissa on this machine")
    rng = np.random.default_rng(1)
    tot = {}
    for _ in range(reps):
        den = ((1.0 - rng.random(n) * 0.99).astype(f32) + f32(1e-15)).astype(f32)
        num = ((rng.random(n) - 0.5) * 2).astype(f32)
        q_ieee = (num / den).astype(f32)  # numpy's fp32 division is correctly rounded
        r0 = (f32(1.0) / den).astype(f32)
        for name, r in (("r = RN(1/den)", r0), ("r = RN(1/den) + 1 ulp", np.nextafter(r0, f32(np.inf))),
                        ("r = RN(1/den) - 1 ulp", np.nextafter(r0, f32(-np.inf)))):
            q = (num * r).astype(f32)
            res = [int((q != q_ieee).sum())]
            for _k in range(3):
                q = fma(fma(-den, q, num), r, q)
                res.append(int((q != q_ieee).sum()))
            t = tot.setdefault(name, [0, 0, 0, 0])
            for i in range(4):
                t[i] += res[i]
    for name, t in tot.items():
        print("%-24s quotients differing from IEEE division after 0/1/2/3 corrections: %s of %d" % (name, t, reps * n))
    return tot


if __name__ == "__main__":
    main()
