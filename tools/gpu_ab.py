#!/usr/bin/env python3
"""GPU-box tool: A/B timing of library builds on ONE box in ONE call (boxes differ by several per cent, so numbers from
different gpurun calls do not compare).  Every build runs the same s2d_step(iters) on the same workloads, alternating,
and the best of `reps` wall-clock timings per build is reported.

  python tools/gpu_ab.py lib/libsplat2d_hip_base.so lib/libsplat2d_hip.so [--reps 3] [--small]
"""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S2D = importlib.import_module("2dgaussiansplatting_amd")

WORKLOADS = [("4096^2/1M", 4096, 4096, 1_000_000, 200, None), ("2048^2/250k", 2048, 2048, 250_000, 400, None),
             ("slab 1/8 of 4096^2/1M", 4096, 4096, 1_000_000, 400, (2048, 2560)),
             ("535x426/50k", 535, 426, 50_000, 2000, None), ("268x213/1024 (as shipped)", 268, 213, 1024, 5000, None)]


MODE = {}


def run(path, W, H, n, iters, rows):
    S2D._lib = None
    L = S2D.load_library(path)
    S2D._lib = L
    kw = dict(MODE)
    if rows is not None:
        kw.update(row_begin=rows[0], row_end=rows[1])
    with S2D.Trainer(W, H, n, **kw) as t:
        t.set_target_synthetic()
        t.init()
        t.step(30, want_mse=False)
        t.synchronize()
        t0 = time.perf_counter()
        t.step(iters, want_mse=False)
        t.synchronize()
        dt = time.perf_counter() - t0
    S2D._lib = None
    return iters / dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", type=int, default=-1, help="index into the workload list")
    ap.add_argument("--deterministic", action="store_true", help="contexts with S2D_CFG_DETERMINISTIC")
    ap.add_argument("--fp16-images", action="store_true")
    args = ap.parse_args()
    if args.deterministic:
        MODE["deterministic"] = True
    if args.fp16_images:
        MODE["fp16_images"] = True
    libs = [os.path.abspath(p) for p in args.libs]
    for k, (name, W, H, n, iters, rows) in enumerate(WORKLOADS):
        if args.only >= 0 and k != args.only:
            continue
        best = {p: 0.0 for p in libs}
        for _ in range(args.reps):
            for p in libs:
                best[p] = max(best[p], run(p, W, H, n, iters, rows))
        ref = best[libs[0]]
        print("%-28s " % name + "  ".join("%s %9.1f it/s (%+5.1f%%)" % (os.path.basename(p), v, 100 * (v / ref - 1)) for p, v in best.items()), flush=True)


if __name__ == "__main__":
    main()
