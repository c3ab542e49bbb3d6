#!/usr/bin/env python3
"""GPU-box tool: the host's own cost of one HaloStep / SlabStep iteration -- the same Python + ctypes + torch calls as in
bench.py --gpus N, on a scene so small (268x213, 2 000 Gaussians, the middle slab of 4) that the device finishes long
before the host has queued the next one.  Loopback collectives: a real all_to_all_single adds torch.distributed's own
per-call cost on top.  What it bounds: at 8 ranks of 4096^2 / 1 M the device needs ~0.35 ms per iteration."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
S2D = importlib.import_module("2dgaussiansplatting_amd")
D = importlib.import_module("2dgaussiansplatting_amd.distributed")
from gpu_halo_rank_timing import Loopback  # noqa: E402

W, H, n, world, rank = 268, 213, 2000, 4, 1
for scheme in ("halo", "dense", "single"):
    stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
    r0, r1 = D.slab_rows(H, rank, world) if scheme != "single" else (0, H)
    grads = torch.zeros(n * 9, dtype=torch.float32, device="cuda")
    with S2D.Trainer(W, H, n, row_begin=r0, row_end=r1, stream=stream.cuda_stream) as t:
        t.bind_grads(grads.data_ptr()); t.set_target_synthetic(); t.init()
        dist = Loopback() if scheme != "single" else None
        step = D.HaloStep(t, D.HipHaloOps(t, n, "cuda"), dist, rank, world, H) if scheme == "halo" else D.SlabStep(t, grads, dist)
        for _ in range(50):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(1000):
            step()
        torch.cuda.synchronize()
        print("%-6s host + tiny device work: %.1f us per iteration" % (scheme, 1e6 * (time.perf_counter() - t0) / 1000), flush=True)
