"""GPU-box tool: what the multi-device handle (s2d_multi_*) costs around the kernels, on ONE GPU.
N ranks sharing the GPU rasterise the same pixels as one context (their kernels overlap on the device), so the
iterations/s of an N-rank handle against the single context's shows the handle's own overheads: worker threads, the
per-iteration barrier and peer copies of slab ownership, the refresh of the hold sets every 64 iterations.  It says
nothing about xGMI.   python3 tools/gpu_multi_handle_timing.py [W H N iters]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
S2D = importlib.import_module("2dgaussiansplatting_amd")

W, H, n, iters = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (4096, 4096, 1000000, 192)))


def run(make, label):
    with make() as t:
        t.set_target_synthetic()
        t.init()
        t.step(16)
        t0 = time.perf_counter()
        tr = t.step(iters)
        dt = time.perf_counter() - t0
        extra = ""
        if hasattr(t, "exchange_info"):
            i = t.exchange_info()
            extra = "  %s, %d rows/iteration, %d state rows handed over, held %.3f of n per rank" % (
                i["scheme"], i["rows_per_iteration"], i["state_handovers"], i["held"] / (n * len(label_devs[label])))
        print("%-34s %8.1f it/s  %7.3f ms/it  mse %.5f%s" % (label, iters / dt, 1e3 * dt / iters, tr[-1], extra), flush=True)


label_devs = {}
run(lambda: S2D.Trainer(W, H, n), "single context")
for world in (1, 2, 4, 8):
    label = "handle, %d rank(s) on one GPU" % world
    label_devs[label] = [0] * world
    run(lambda: S2D.MultiTrainer(W, H, n, [0] * world, share_gpu=True), label)
