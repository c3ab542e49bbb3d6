#!/usr/bin/env python3
"""GPU-box tool: atomic-gradient mode against S2D_CFG_DETERMINISTIC on one box, alternating (boxes differ by several per cent):
  python3 tools/gpu_det_ab.py [reps=2]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S2D = importlib.import_module("2dgaussiansplatting_amd")


def run(W, H, n, iters, **kw):
    with S2D.Trainer(W, H, n, **kw) as t:
        t.set_target_synthetic()
        t.init()
        t.step(20, want_mse=False)
        t.synchronize()
        t0 = time.perf_counter()
        t.step(iters, want_mse=False)
        t.synchronize()
        return iters / (time.perf_counter() - t0)


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for name, W, H, n, iters in (("4096^2/1M", 4096, 4096, 1000000, 200), ("2048^2/250k", 2048, 2048, 250000, 400), ("535x426/50k", 535, 426, 50000, 2000)):
    best = {False: 0.0, True: 0.0}
    for _ in range(reps):
        for det in (False, True):
            best[det] = max(best[det], run(W, H, n, iters, deterministic=det))
    print("%-14s atomics %9.1f it/s   deterministic %9.1f it/s  (%+.1f %%)" % (name, best[False], best[True], 100 * (best[True] / best[False] - 1)), flush=True)
