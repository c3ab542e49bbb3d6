#!/usr/bin/env python3
"""GPU-box tool: iterations/s at 4096x4096 / 1M against the tile-list re-use margin (s2d_config.rebin_margin)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S2D = importlib.import_module("2dgaussiansplatting_amd")
W = H = 4096
n = 1000000
for margin in (1.0, 1.5, 2.0, 2.5, 3.0, 4.0):
    with S2D.Trainer(W, H, n, rebin_margin=margin) as t:
        t.lean_backward = True
        t.set_target_synthetic(); t.init()
        t.step(20, want_mse=False); t.synchronize()
        r0 = t.stats()["rebins"]
        t0 = time.perf_counter()
        t.step(200, want_mse=False); t.synchronize()
        dt = time.perf_counter() - t0
        st = t.stats()
        print("margin %.1f px: %.1f it/s  pairs %d  rebuilds %d in 200 iterations" % (margin, 200 / dt, st["pairs_binned"], st["rebins"] - r0), flush=True)
