#!/usr/bin/env python3
"""GPU-box tool: A/B of two library builds on the WHOLE iteration with all nine gradients (forward_backward without the
skip flag + adam_step, what bench.py's `value` times), alternating on one box, best of `reps`.
  python tools/gpu_ab_full.py build/libsplat2d_hip_base.so 2dgaussiansplatting_amd/lib/libsplat2d_hip.so [reps]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S2D = importlib.import_module("2dgaussiansplatting_amd")
libs = [os.path.abspath(p) for p in sys.argv[1:3]]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
WORK = [("4096^2/1M", 4096, 4096, 1_000_000, 200), ("2048^2/250k", 2048, 2048, 250_000, 400)]


def run(path, W, H, n, iters, full):
    S2D._lib = None
    S2D._lib = S2D.load_library(path)
    with S2D.Trainer(W, H, n) as t:
        t.lean_backward = not full
        t.set_target_synthetic(); t.init()
        for _ in range(30):
            t.forward_backward(skip_image=True); t.adam_step()
        t.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            t.forward_backward(skip_image=True); t.adam_step()
        t.synchronize()
        dt = time.perf_counter() - t0
    S2D._lib = None
    return iters / dt


for name, W, H, n, iters in WORK:
    for full in (True, False):
        best = [0.0, 0.0]
        for r in range(reps):
            for k, p in enumerate(libs):
                best[k] = max(best[k], run(p, W, H, n, iters, full))
        print("%-12s %-14s  A %8.1f it/s   B %8.1f it/s   B/A %+.2f %%" % (name, "nine gradients" if full else "lean (8 of 9)", best[0], best[1], 100.0 * (best[1] / best[0] - 1.0)), flush=True)
