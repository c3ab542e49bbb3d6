"""GPU-box tool: long bit-identity run of the multi-device handle's two schemes (s2d_multi_*, all ranks on one GPU).
Slab ownership (hold sets, refreshes with state hand-over, peer copies of shared gradient rows) against replicated
state (host-staged sum of all gradients in rank order), deterministic gradients: MSE trace, parameters and Adam
moments must agree bit for bit.   python3 tools/gpu_multi_long_check.py [W H N ranks iters]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
S2D = importlib.import_module("2dgaussiansplatting_amd")

W, H, n, world, iters = (int(a) for a in (sys.argv[1:6] if len(sys.argv) >= 6 else (2048, 2048, 250000, 8, 600)))
out = {}
for replicated in (False, True):
    t0 = time.perf_counter()
    with S2D.MultiTrainer(W, H, n, [0] * world, share_gpu=True, deterministic=True, replicated=replicated) as m:
        m.set_target_synthetic()
        m.init()
        tr = np.concatenate([m.step(min(100, iters - k)) for k in range(0, iters, 100)])
        info = m.exchange_info()
        out[replicated] = (tr, m.get_splats(), m.get_adam()[0])
    print("%-10s %d ranks, %dx%d, %d splats, %d iterations in %.1f s: mse %.6f -> %.6f  %s" % (
        info["scheme"], world, W, H, n, iters, time.perf_counter() - t0, tr[0], tr[-1], info), flush=True)
same = [out[False][k].tobytes() == out[True][k].tobytes() for k in range(3)]
print("bit-identical: trace %s, parameters %s, Adam moments %s" % tuple(same))
sys.exit(0 if all(same) else 1)
