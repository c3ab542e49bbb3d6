"""The N > 1 path on CPU: slab partition, and a world_size-2 gloo run of the same SlabStep that bench.py
drives on GPUs, with an oracle-backed stand-in for the HIP backend (so the orchestration, the all-reduce of
the N x 9 gradients and the replica-identity property are covered without a GPU)."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O

D = importlib.import_module("2dgaussiansplatting_amd.distributed")


@pytest.mark.parametrize("H", [1, 15, 16, 17, 213, 426, 4096, 8192])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_slab_rows_partition(H, world):
    rows = [D.slab_rows(H, r, world) for r in range(world)]
    assert rows[0][0] == 0 and rows[-1][1] == H
    for (a0, a1), (b0, b1) in zip(rows, rows[1:]):
        assert a1 == b0 or (a0 == a1 == H and b0 >= H)  # contiguous
    for r0, r1 in rows:
        assert r0 % 16 == 0 and r0 <= r1 <= H
    assert sum(r1 - r0 for r0, r1 in rows) == H
    if H >= 16 * world:
        sizes = [r1 - r0 for r0, r1 in rows]
        assert max(sizes) - min(sizes) <= 16 + (16 - H % 16) % 16


class OracleSlabBackend:
    """forward()/backward()/adam_step() of one rank's slab, computed by the CPU oracle."""

    def __init__(self, target, n, r0, r1):
        self.o = O.OracleTrainer(target, n)
        self.r0, self.r1 = r0, r1
        self.grads = torch.from_numpy(self.o.dsplats.view(np.float32).reshape(-1))  # aliases dsplats
        self.sqerr = []

    def forward(self):
        self.o.forward(self.r0, self.r1)

    def backward(self):
        self.o.backward(self.r0, self.r1, zero=True)
        self.sqerr.append(self.o.L.s2do_sqerr_rows(self.o.image0.ctypes.data, self.o.ref.ctypes.data, self.o.W,
                                                   self.o.H, self.r0, self.r1))

    def adam_step(self):
        assert self.o.adam() == 0


def _worker(rank, world, port, steps, n, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tgt = O.target_rgba32f(O.load_s2di(os.path.join(O.GOLDEN, "squirrel_cls_mini_268x213.s2di")))
        r0, r1 = D.slab_rows(tgt.shape[0], rank, world)
        be = OracleSlabBackend(tgt, n, r0, r1)
        step = D.SlabStep(be, be.grads, dist)
        for _ in range(steps):
            step()
        sq = D.reduce_sqerr(torch.tensor(be.sqerr, dtype=torch.float64), dist)
        gathered = [torch.zeros(n * 9) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(be.o.splats.view(np.float32).reshape(-1).copy()))
        if rank == 0:
            np.save(out + ".splats.npy", np.stack([g.numpy() for g in gathered]))
            np.save(out + ".sq.npy", sq.numpy())
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_row_slabs_match_single_process(tmp_path, world):
    steps, n = 4, 600
    out = str(tmp_path / "res")
    mp.spawn(_worker, args=(world, _free_port(), steps, n, out), nprocs=world, join=True)
    splats = np.load(out + ".splats.npy")
    sq = np.load(out + ".sq.npy")
    # replicas stay bit-identical: every rank applied Adam to the same all-reduced gradients
    for r in range(1, world):
        assert splats[r].tobytes() == splats[0].tobytes()
    # and agree with the single-process reference loop up to the fp32 summation order of the gradients
    tgt = O.target_rgba32f(O.load_s2di(os.path.join(O.GOLDEN, "squirrel_cls_mini_268x213.s2di")))
    o = O.OracleTrainer(tgt, n)
    mses = [o.step()[1] for _ in range(steps)]
    want = o.splats.view(np.float32).reshape(-1)
    np.testing.assert_allclose(splats[0], want, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(sq / (o.H * o.W * 3), mses, rtol=1e-6)
    assert abs(sq[0] / (o.H * o.W * 3) - mses[0]) <= 1e-12 * mses[0]  # iteration 0: same framebuffer, only the row split of the double sum differs
