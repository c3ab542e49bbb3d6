"""GPU-box tool: a short run of the multi-device handle for a kernel trace (rocprofv3 --kernel-trace -- python3 this W H N ranks iters)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
S2D = importlib.import_module("2dgaussiansplatting_amd")
W, H, n, world, iters = (int(a) for a in sys.argv[1:6])
with (S2D.MultiTrainer(W, H, n, [0] * world, share_gpu=True) if world > 0 else S2D.Trainer(W, H, n)) as t:
    t.set_target_synthetic()
    t.init()
    t.step(16)
    t.step(iters)
