#!/usr/bin/env python3
"""GPU-box tool: the whole backward pass (all nine gradients, main.cpp:595-710) against the lean one (dL/dopacity skipped),
alternating blocks on one context at 4096^2 / 1 M, so that neither is always the first (coldest) block of a process.
Prints iterations/s per block and per-step medians; usage: gpu_full_vs_lean.py [blocks] [steps]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S2D = importlib.import_module("2dgaussiansplatting_amd")
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
with S2D.Trainer(4096, 4096, 1_000_000) as t:
    t.set_target_synthetic()
    t.init()
    for b in range(blocks):
        full = b % 2 == 0
        t.lean_backward = not full
        per = []
        t.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            t1 = time.perf_counter()
            t.forward_backward()
            t.adam_step()
            if k % 10 == 9:
                t.synchronize()
                per.append((time.perf_counter() - t1))
        t.synchronize()
        dt = time.perf_counter() - t0
        print("block %d %-4s: %.1f it/s over %d steps (%.3f ms/step)" % (b, "full" if full else "lean", steps / dt, steps, 1e3 * dt / steps), flush=True)
