"""GPU-box tool: deterministic mode (S2D_CFG_DETERMINISTIC) at the bench workload, for a kernel trace:
rocprofv3 --kernel-trace --stats -d gpurun_out/det -- python3 tools/gpu_det_trace.py"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
S2D = importlib.import_module("2dgaussiansplatting_amd")
with S2D.Trainer(4096, 4096, 1000000, deterministic=True) as t:
    t.set_target_synthetic()
    t.init()
    t.step(20)
    t0 = time.perf_counter()
    t.step(100)
    dt = time.perf_counter() - t0
    print("%.1f it/s" % (100 / dt))
