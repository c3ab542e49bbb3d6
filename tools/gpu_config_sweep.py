#!/usr/bin/env python3
"""GPU-box tool: iterations/s of the BASELINE.json configurations (single GPU), for DESIGN.md."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
S2D = importlib.import_module("2dgaussiansplatting_amd")

def run(name, W, H, n, target=None, iters=200, **kw):
    with S2D.Trainer(W, H, n, **kw) as t:
        if target is None: t.set_target_synthetic()
        else: t.set_target(target)
        t.init()
        t.step(20, want_mse=False); t.synchronize()
        t0 = time.perf_counter(); tr = t.step(iters); t.synchronize(); dt = time.perf_counter() - t0
        st = t.stats()
    print("%-44s %9.1f it/s  %7.3f ms/it  pairs %9d  rebuilds %3d  mse %.2f -> psnr %.2f dB" % (
        name, iters / dt, 1e3 * dt / iters, st["pairs_binned"], st["rebins"], tr[-1], 10 * np.log10(255 ** 2 / tr[-1])))

mini = O.target_rgba32f(O.load_s2di(os.path.join(O.GOLDEN, "squirrel_cls_mini_268x213.s2di")))
full = O.target_rgba32f(O.load_s2di(os.path.join(O.GOLDEN, "squirrel_cls_535x426.s2di")))
run("cfg0 mini 268x213, N=2000", 268, 213, 2000, mini, 500)
run("reference as shipped 268x213, N=1024", 268, 213, 1024, mini, 500)
run("cfg1 native 535x426, N=50000", 535, 426, 50000, full, 300)
run("cfg2 2048x2048 synthetic, N=250k", 2048, 2048, 250000)
run("cfg3 4096x4096 synthetic, N=1M", 4096, 4096, 1000000)
run("cfg3 ... deterministic", 4096, 4096, 1000000, deterministic=True)
run("cfg3 ... fp16 images", 4096, 4096, 1000000, fp16_images=True)
run("cfg4 8192x8192 synthetic, N=4M (fp32 images)", 8192, 8192, 4000000, iters=60)
run("cfg4 8192x8192 synthetic, N=4M, fp16 images", 8192, 8192, 4000000, iters=60, fp16_images=True)
