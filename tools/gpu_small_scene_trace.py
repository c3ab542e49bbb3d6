"""GPU-box tool: the reference's as-shipped scene (268x213 squirrel, 1024 Gaussians, main.cpp:257,271) in batches,
for a kernel trace:  rocprofv3 --kernel-trace --stats -d gpurun_out/small -- python3 tools/gpu_small_scene_trace.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import importlib

S2D = importlib.import_module("2dgaussiansplatting_amd")
import struct

import numpy as np


def load_target(path):
    """.s2di fixture -> RGBA32F as Image2DRGBA8_to_Image2DRGBA32 makes it (main.cpp:258)."""
    with open(path, "rb") as fh:
        assert fh.read(4) == b"S2DI"
        w, h, c = struct.unpack("<III", fh.read(12))
        rgb = np.frombuffer(fh.read(w * h * c), dtype=np.uint8).reshape(h, w, c)
    out = np.ones((h, w, 4), dtype=np.float32)
    out[..., :3] = rgb.astype(np.float32) / np.float32(255.0)
    return out


n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
img = load_target(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "squirrel_cls_mini_268x213.s2di"))
with S2D.Trainer(img.shape[1], img.shape[0], n) as t:
    t.set_target(img)
    t.init()
    t.step(200)
    t0 = time.perf_counter()
    mse = t.step(iters)
    dt = time.perf_counter() - t0
    print(f"{iters / dt:.0f} it/s  {dt / iters * 1e6:.1f} us/it  mse {mse[-1]:.3f}  rebuilds {t.rebuild_count()}")
