#!/usr/bin/env python3
"""GPU-box tool (EXPERIMENT build only: build/libsplat2d_hip_exp.so, whose counting backward kernel adds, per wave and batch,
the number of steps a walk with FOUR list cursors per wave -- one per 4x4-pixel quadrant of the wave's 8x8 block -- would
take: max over the quadrants of the entries that touch a live pixel of the quadrant; it lands in bwd_lane_hist[0], which is
otherwise always zero).  Prints that against the executions of today's one-cursor walk.
usage: S2D_LIBRARY=build/libsplat2d_hip_exp.so gpu_quadrant_steps.py [W H n] [warm iterations ...]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S2D = importlib.import_module("2dgaussiansplatting_amd")
W, H, n = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4096, 4096, 1000000)
warms = [int(v) for v in sys.argv[4:]] or [0, 100]
for warm in warms:
    with S2D.Trainer(W, H, n) as t0:
        t0.set_target_synthetic(); t0.init()
        if warm: t0.step(warm, want_mse=False)
        sp = t0.get_splats()
    with S2D.Trainer(W, H, n, count_pairs=True) as t:
        t.set_target_synthetic(); t.set_splats(sp)
        t.forward(); t.backward(); t.synchronize()
        st = t.stats()
    execs, steps, quads = st["bwd_wave_execs"], st["bwd_lane_hist"][0], st["bwd_quadrant_execs"]
    print("%dx%d n=%d after %d iterations: one cursor per wave %d executions; four cursors %d steps = %.3f of them "
          "(perfectly balanced quadrants would need %.3f); live lanes per execution %.1f of 64 -> per step %.1f of 64" % (
              W, H, n, warm, execs, steps, steps / execs, quads / 4.0 / execs, st["bwd_active"] / execs, st["bwd_active"] / max(steps, 1)), flush=True)
