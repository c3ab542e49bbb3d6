#!/usr/bin/env python3
"""GPU-box tool: the multi-process path (torchrun, 2 ranks sharing one GPU, gloo all-reduce) must reproduce the
single-process result: same MSE trace to fp32-summation noise, identical replicas.  Run as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 tools/gpu_two_rank_check.py
"""
import importlib, os, sys
import numpy as np
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S2D = importlib.import_module("2dgaussiansplatting_amd")
D = importlib.import_module("2dgaussiansplatting_amd.distributed")

W, H, n, steps = 1024, 768, 60000, 12
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
r0, r1 = D.slab_rows(H, rank, world)
grads = torch.zeros(n * 9, dtype=torch.float32, device="cuda")
t = S2D.Trainer(W, H, n, row_begin=r0, row_end=r1, stream=stream.cuda_stream)
t.bind_grads(grads.data_ptr()); t.lean_backward = True
t.set_target_synthetic(); t.init()
step = D.SlabStep(t, grads, dist)
for _ in range(steps):
    step()
torch.cuda.synchronize()
sq = D.reduce_sqerr(torch.from_numpy(t.sqerr_trace(0, steps)), dist).numpy() / (H * W * 3)
mine = torch.from_numpy(t.get_splats().view(np.float32).reshape(-1).copy())
allp = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(allp, mine)
if rank == 0:
    with S2D.Trainer(W, H, n) as s:
        s.set_target_synthetic(); s.init()
        ref = s.step(steps)
        refp = s.get_splats().view(np.float32).reshape(-1)
    same = all(p.numpy().tobytes() == allp[0].numpy().tobytes() for p in allp)
    print("replicas identical:", same)
    print("mse %d-rank :" % world, " ".join("%.4f" % v for v in sq))
    print("mse 1-rank :", " ".join("%.4f" % v for v in ref))
    print("max rel mse diff %.2e ; max abs param diff %.2e" % (np.max(np.abs(sq - ref) / ref), np.max(np.abs(allp[0].numpy() - refp))))
    assert same and np.max(np.abs(sq - ref) / ref) < 1e-4
    print("two-rank check ok")
t.close()
dist.destroy_process_group()
