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
        step = D.SlabStep(be, be.grads, dist, params=lambda: torch.from_numpy(be.o.splats.view(np.float32).reshape(-1, 9)),
                          check_interval=2)
        for _ in range(steps):
            step()
        assert step.checks == steps // 2       # the replica checksum was exchanged and agreed
        if rank == 1:                          # ... and a replica that drifts by one bit is caught (on every rank)
            be.o.splats["pos"][0, 0] = np.nextafter(be.o.splats["pos"][0, 0], np.float32(1e9))
        try:
            step.assert_replicas_identical()
            raise AssertionError("a diverged replica went unnoticed")
        except RuntimeError:
            pass
        if rank == 1:
            be.o.splats["pos"][0, 0] = np.nextafter(be.o.splats["pos"][0, 0], np.float32(-1e9))
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


# ---------------------------------------------------------------------------------------------------------
# Slab ownership (distributed.HaloStep): same orchestration as on GPUs, row operations done by numpy on the oracle's
# arrays.  A small margin and a refresh every iteration make splats change hands within a few steps.
# ---------------------------------------------------------------------------------------------------------
class OracleHaloBackend(OracleSlabBackend):
    """OracleSlabBackend whose Adam step leaves the splats this rank does not hold untouched."""

    held = None

    def adam_step(self):
        if self.held is None:
            return super().adam_step()
        keep = ~self.held
        sp, ad = self.o.splats.copy(), self.o.adams.copy()
        assert self.o.adam() == 0
        self.o.splats[keep] = sp[keep]
        self.o.adams[keep] = ad[keep]


class OracleHaloOps:
    def __init__(self, be):
        self.be, self.n = be, be.o.n
        self.arr = {D.ROWS_GRADS: be.o.dsplats.view(np.float32).reshape(self.n, 9),
                    D.ROWS_SPLATS: be.o.splats.view(np.float32).reshape(self.n, 9),
                    D.ROWS_ADAM: be.o.adams.view(np.float32).reshape(self.n, 18)}

    def halo_masks(self, bounds, margin):
        sp = self.arr[D.ROWS_SPLATS]
        y = sp[:, 1]
        reach = np.float32(3.0) * np.maximum(sp[:, 2], sp[:, 3]) + np.float32(2.0) + np.float32(margin)
        m = np.zeros(self.n, dtype=np.int32)
        for q in range(len(bounds) - 1):
            m |= ((y + reach >= np.float32(bounds[q])) & (y - reach <= np.float32(bounds[q + 1]))).astype(np.int32) << q
        if self.be.held is not None:
            m[~self.be.held] = 0
        return torch.from_numpy(m)

    def halo_commit(self, mask, rank, added=True):
        self.be.held = ((mask.numpy() >> rank) & 1).astype(bool)

    def rows_gather(self, what, ids, out=None):
        v = torch.from_numpy(self.arr[what][ids.numpy().astype(np.int64)].copy())
        if out is not None:
            out.copy_(v)
            return out
        return v

    def rows_scatter(self, what, ids, values):
        self.arr[what][ids.numpy().astype(np.int64)] = values.numpy().reshape(len(ids), -1)

    def grads_combine(self, rows, src, recv):
        g, rows, src, recv = self.arr[D.ROWS_GRADS], rows.numpy(), src.numpy(), recv.numpy()
        for u, i in enumerate(rows):
            acc = None
            for q in range(src.shape[1]):
                s = src[u, q]
                if s == -1:
                    continue
                v = g[i].copy() if s == -2 else recv[s]
                acc = v if acc is None else (acc + v).astype(np.float32)
            g[i] = acc


def _halo_worker(rank, world, port, steps, n, interval, margin, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tgt = O.target_rgba32f(O.load_s2di(os.path.join(O.GOLDEN, "squirrel_cls_mini_268x213.s2di")))
        r0, r1 = D.slab_rows(tgt.shape[0], rank, world)
        be = OracleHaloBackend(tgt, n, r0, r1)
        step = D.HaloStep(be, OracleHaloOps(be), dist, rank, world, tgt.shape[0], rehalo_interval=interval,
                          margin_rows=margin)
        held_frac = [float(be.held.mean())]
        for _ in range(steps):
            step()
            held_frac.append(float(be.held.mean()))
        sq = D.reduce_sqerr(torch.tensor(be.sqerr, dtype=torch.float64), dist)
        full = step.gather_full(D.ROWS_SPLATS)
        # every rank's raw copy + mask, to check that all holders agree bit for bit
        raw = [torch.zeros(n * 9) for _ in range(world)]
        dist.all_gather(raw, torch.from_numpy(be.o.splats.view(np.float32).reshape(-1).copy()))
        masks = [torch.zeros(n, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(masks, step.mask)
        moved = torch.tensor([step.handed_over], dtype=torch.int64)
        dist.all_reduce(moved)
        if rank == 0:
            np.save(out + ".full.npy", full.numpy())
            np.save(out + ".raw.npy", np.stack([g.numpy() for g in raw]))
            np.save(out + ".masks.npy", np.stack([m.numpy() for m in masks]))
            np.save(out + ".sq.npy", sq.numpy())
            np.save(out + ".meta.npy", np.array([int(moved.item()), max(held_frac) < 1.0]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,interval,margin", [(2, 1, 1.0), (3, 2, 2.0), (2, 16, 8.0)])
def test_gloo_slab_ownership_matches_single_process(tmp_path, world, interval, margin):
    steps, n = 12, 600
    out = str(tmp_path / "res")
    mp.spawn(_halo_worker, args=(world, _free_port(), steps, n, interval, margin, out), nprocs=world, join=True)
    full, raw, masks, sq = (np.load(out + s) for s in (".full.npy", ".raw.npy", ".masks.npy", ".sq.npy"))
    moved, _ = np.load(out + ".meta.npy")
    raw = raw.reshape(world, n, 9)
    # invariants: every splat has a holder; all holders carry the same mask and bit-identical parameters
    union = np.zeros(n, dtype=np.int64)
    for q in range(world):
        held_q = ((masks[q] >> q) & 1).astype(bool)
        union |= np.where(held_q, 1 << q, 0)
    assert (union != 0).all()
    for q in range(world):
        held_q = ((masks[q] >> q) & 1).astype(bool)
        assert (masks[q][held_q] == union[held_q]).all()       # a holder's mask lists exactly the holders
        assert (masks[q][~held_q] == 0).all()
        first = np.array([int(u & -u).bit_length() - 1 for u in union])
        ref_rows = raw[first, np.arange(n)]
        assert raw[q][held_q].tobytes() == ref_rows[held_q].tobytes()
    if interval <= 2:
        assert moved > 0, "the test is meant to exercise hand-overs"
    # agreement with the single-process reference loop up to the fp32 summation order of the gradients
    tgt = O.target_rgba32f(O.load_s2di(os.path.join(O.GOLDEN, "squirrel_cls_mini_268x213.s2di")))
    o = O.OracleTrainer(tgt, n)
    mses = [o.step()[1] for _ in range(steps)]
    want = o.splats.view(np.float32).reshape(n, 9)
    np.testing.assert_allclose(full, want, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(sq / (o.H * o.W * 3), mses, rtol=1e-6)


def _selftest_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = D.all_to_all_selftest(dist, "cpu")
        if rank == 0:
            np.save(out, np.array([int(ok)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_all_to_all_selftest_passes_on_gloo(tmp_path, world):
    out = str(tmp_path / "ok.npy")
    mp.spawn(_selftest_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert np.load(out)[0] == 1


def test_gloo_slab_ownership_without_margin_fails_on_every_rank(tmp_path):
    """With no halo margin a splat reaches a neighbour's rows before it is handed over: every rank must raise (none
    may be left waiting in a collective)."""
    out = str(tmp_path / "res")
    with pytest.raises(Exception) as ei:
        mp.spawn(_halo_worker, args=(2, _free_port(), 12, 600, 3, 0.0, out), nprocs=2, join=True)
    assert "handed over" in str(ei.value)


def test_bench_spawns_its_ranks_from_the_plain_command():
    """`python bench.py --gpus 2` with no launcher environment must start its two ranks itself (as a child
    torch.distributed.run), relay rank 0's JSON line and return non-zero when a rank fails.  --launch-selftest stops
    each rank after the rendezvous + an all-reduce, so this runs without a GPU; the GPU suite runs the real thing
    (tests/test_gpu_fullsize.py::test_bench_plain_command_two_ranks_gloo)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-selftest"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # ONE JSON line on stdout
    out = json.loads(lines[0])
    assert out == {"selftest": "launch", "world": 2, "sum": 3, "gpus_arg": 2}
    env["S2D_BENCH_SELFTEST_FAIL_RANK"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-selftest"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0


def test_bench_watchdog_reports_a_rank_that_stops_answering():
    """A rank that never comes back (here: it sleeps for ever right after the rendezvous; on hardware: a first-contact
    RCCL or peer-to-peer stall) must not end as a silent kill at somebody else's time limit: the PARENT of `python
    bench.py --gpus 2` -- which never touches the GPU -- stops the child group when its wall-clock budget runs out,
    prints ONE JSON line with "error", the stage the ranks had reached and the last stderr lines, and exits non-zero."""
    import json
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["S2D_BENCH_SELFTEST_HANG_RANK"] = "1"
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-selftest", "--watchdog-seconds", "25"],
                       env=env, capture_output=True, text=True, timeout=300)
    dt = time.perf_counter() - t0
    assert p.returncode != 0
    assert dt < 25 + 30, dt                              # budget + the grace of the group kill, not torch's 10 minutes
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                        # still ONE JSON line
    out = json.loads(lines[0])
    assert out["value"] is None and "stopped answering" in out["error"], out
    assert out["stage"] in ("init", "selftest"), out     # the furthest stage any rank reported
    assert out["stage_per_rank"].get("1") == "init", out # the sleeping rank got as far as the rendezvous
    assert out["n_gpus"] == 2 and out["attempts"][0]["rc"] is None and out["watchdog_budget_s"] == 25
    # nothing of the job is left behind
    left = subprocess.run(["ps", "-eo", "pid,args"], capture_output=True, text=True).stdout
    assert not any("--launch-selftest" in ln and "--watchdog-seconds 25" in ln for ln in left.splitlines()), left


def test_bench_rank_watchdog_when_somebody_elses_launcher_starts_the_ranks():
    """The driver starts the ranks itself (`python -m torch.distributed.run ... bench.py --gpus N`): no parent of ours watches
    them.  Each rank then carries its own watchdog timer: with one rank asleep for ever the job still ends inside the budget,
    rank 0 prints the ONE JSON line -- an error record naming the stage it was at -- and the launcher returns non-zero."""
    import json
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "S2D_BENCH_PARENT")}
    env["S2D_BENCH_SELFTEST_HANG_RANK"] = "1"
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--launch-selftest",
                        "--watchdog-seconds", "20"], env=env, capture_output=True, text=True, timeout=300)
    dt = time.perf_counter() - t0
    assert p.returncode != 0
    assert dt < 20 + 40, dt
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["value"] is None and "no result within 20 s" in out["error"] and out["stage"] in ("init", "selftest"), out
    assert "bench.py watchdog: rank" in p.stderr
