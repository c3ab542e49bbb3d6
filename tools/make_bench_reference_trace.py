#!/usr/bin/env python3
"""Container tool: the CPU oracle's MSE trajectory on bench.py's workload (BASELINE.json configs[3]: 4096x4096
synthetic target, 1 000 000 Gaussians from init(), optimizeOpacity off), iterations 0..199, committed as
tests/golden/bench_reference_trace.json.  bench.py quotes its PSNR against this trace ("PSNR vs ref"), and
tests/test_gpu_fullsize.py asserts the band.

The oracle is oracle/s2d_oracle.c (`s2do_step_mt`: forward/backward over row-slab threads, per-slab partial
gradients added in slab order -- the reference's arithmetic with a different fp32 summation order of the
gradients only; the trajectory is chaotic, so only a PSNR band is meaningful beyond the first iterations).
About 6-8 s per iteration on 8 container cores.

  python tools/make_bench_reference_trace.py [--iters 200] [--threads 6] [--width 4096 --height 4096 --splats 1000000]
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--height", type=int, default=4096)
    ap.add_argument("--splats", type=int, default=1_000_000)
    ap.add_argument("--out", default=os.path.join(O.GOLDEN, "bench_reference_trace.json"))
    args = ap.parse_args()
    W, H, n = args.width, args.height, args.splats
    o = O.OracleTrainer(O.synthetic_target(W, H), n)
    trace = []
    t0 = time.time()
    for k in range(args.iters):
        st, mse = o.step(threads=args.threads)
        assert st == 0, "the oracle hit the reference's abort() at iteration %d" % k
        trace.append(mse)
        if k % 10 == 0 or k == args.iters - 1:
            print("%d itr, mse %.4f   (%.0f s)" % (k, mse, time.time() - t0), flush=True)
            json.dump({"partial": True, "mse": trace}, open(args.out + ".partial", "w"))
    out = {
        "what": "MSE the reference's loop prints (main.cpp:807) per iteration on bench.py's workload, from the CPU oracle",
        "generator": "tools/make_bench_reference_trace.py",
        "oracle": "oracle/s2d_oracle.c s2do_step_mt, gcc -O2 -ffp-contract=off, %d row-slab threads" % args.threads,
        "width": W, "height": H, "n_splats": n, "optimize_opacity": False, "training_rate": 0.05,
        "target": "ref(x,y) = (x/W, 1 - x/W, y/H)", "splats": "init() seeds (main.cpp:280-305)",
        "mse": trace,
        "psnr_db": [10.0 * math.log10(255.0 ** 2 / m) for m in trace],
    }
    json.dump(out, open(args.out, "w"), indent=0)
    if os.path.exists(args.out + ".partial"):
        os.remove(args.out + ".partial")
    print("wrote", args.out)


if __name__ == "__main__":
    main()
