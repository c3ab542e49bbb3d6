#!/usr/bin/env python3
"""GPU-box tool: long runs (thousands of iterations) at several sizes; reports rate, rebuilds, final PSNR and that the
status word stayed clean.  usage: gpu_soak.py"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S2D = importlib.import_module("2dgaussiansplatting_amd")
for W, H, n, steps, kw in ((268, 213, 2000, 20000, {}), (4096, 4096, 1000000, 3000, {}), (4096, 4096, 1000000, 2000, {"opacity": True}),
                            (4096, 4096, 1000000, 1000, {"deterministic": True}), (8192, 8192, 4000000, 300, {"fp16_images": True})):
    opacity = kw.pop("opacity", False)   # "Optimize opacity" on: the nine-gradient kernel, opacity trained
    with S2D.Trainer(W, H, n, **kw) as t:
        t.lean_backward = True
        t.optimize_opacity = opacity
        if opacity:
            kw = dict(kw, optimize_opacity=True)
        t.set_target_synthetic(); t.init()
        t0 = time.perf_counter()
        done = 0
        while done < steps:
            k = min(500, steps - done)
            t.step(k, want_mse=False)
            done += k
        t.synchronize()   # raises on a non-finite parameter
        dt = time.perf_counter() - t0
        st = t.stats()
        sq = t.sqerr_trace(steps - 1, 1)[0] / (H * W * 3)
        sp = t.get_splats().view(np.float32).reshape(n, 9)
        print("%dx%d n=%d %s: %d iterations, %.1f it/s, rebuilds %d, pairs %d (capacity %d), final psnr %.2f dB, finite %s, "
              "sx range [%.2f, %.2f]" % (W, H, n, kw or "", st["iterations"], steps / dt, st["rebins"], st["pairs_binned"],
              st["pairs_capacity"], 10 * np.log10(255.0 ** 2 / sq), bool(np.isfinite(sp).all()), sp[:, 2].min(), sp[:, 2].max()), flush=True)
