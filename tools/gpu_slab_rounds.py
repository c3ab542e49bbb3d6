#!/usr/bin/env python3
"""GPU-box tool: duration of the fused raster launch against the number of tiles it covers (row slabs of the 4096^2 / 1 M
scene around the middle of the image, lists built once): what a launch of few rounds of workgroups pays in ramp and tail.
usage: gpu_slab_rounds.py [launches]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S2D = importlib.import_module("2dgaussiansplatting_amd")
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 60
W = H = 4096
n = 1_000_000
print("rows  tiles  rounds(2048 slots)  us/launch  us/round  us/launch - rows/4096 * full")
res = []
for rows in (64, 128, 256, 512, 1024, 2048, 4096):
    r0 = (H - rows) // 2 // 16 * 16
    with S2D.Trainer(W, H, n, row_begin=r0, row_end=r0 + rows) as t:
        t.lean_backward = True
        t.set_target_synthetic()
        t.init()
        for _ in range(5):
            t.forward_backward(skip_image=True)
        t.synchronize()
        t0 = time.perf_counter()
        for _ in range(launches):
            t.forward_backward(skip_image=True)
        t.synchronize()
        us = 1e6 * (time.perf_counter() - t0) / launches
    res.append((rows, us))
full = res[-1][1]
for rows, us in res:
    tiles = rows // 16 * 256
    print("%4d  %5d  %6.2f  %9.1f  %8.1f  %8.1f" % (rows, tiles, tiles / 2048.0, us, us / (tiles / 2048.0), us - full * rows / 4096.0), flush=True)
