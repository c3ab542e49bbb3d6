"""A stand-in for the few torch.distributed calls SlabStep / HaloStep make, with the ranks as THREADS of one
process: lets the GPU tests drive several slab contexts on the single GPU of a test box through the real host
logic (one process on the card, no rendezvous).  Test infrastructure only."""
import threading

import torch


class _ReduceOp:
    SUM, MAX, MIN = "sum", "max", "min"


class ThreadDist:
    def __init__(self, world):
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world

    def rank_view(self, rank):
        return _RankView(self, rank)

    def run(self, fn):
        """fn(rank, dist_view) on `world` threads; re-raises the first exception."""
        errs = [None] * self.world

        def body(r):
            try:
                fn(r, self.rank_view(r))
            except BaseException as e:  # noqa: BLE001 - reported below
                errs[r] = e
                self.bar.abort()

        th = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for e in errs:
            if e is not None and not isinstance(e, threading.BrokenBarrierError):
                raise e
        for e in errs:
            if e is not None:
                raise e


class _RankView:
    ReduceOp = _ReduceOp

    def __init__(self, group, rank):
        self.g, self.rank = group, rank

    def get_backend(self):
        return "threads"

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.g.world

    def _sync(self, t):
        if t.is_cuda:
            torch.cuda.synchronize()

    def barrier(self):
        self.g.bar.wait()

    def all_reduce(self, t, op=_ReduceOp.SUM):
        c = t.clone()
        self._sync(c)  # ranks may work on different streams: complete before another thread reads it
        self.g.slots[self.rank] = c
        self.g.bar.wait()
        acc = self.g.slots[0].clone()
        for q in range(1, self.g.world):  # fixed order: every rank computes the same bits
            if op == _ReduceOp.MAX:
                acc = torch.maximum(acc, self.g.slots[q])
            elif op == _ReduceOp.MIN:
                acc = torch.minimum(acc, self.g.slots[q])
            else:
                acc = acc + self.g.slots[q]
        self._sync(acc)
        self.g.bar.wait()
        t.copy_(acc)

    def all_to_all_single(self, recv, send, recv_rows=None, send_rows=None):
        w = self.g.world
        if send_rows is None:
            send_rows = [send.shape[0] // w] * w
        if recv_rows is None:
            recv_rows = [recv.shape[0] // w] * w
        self._sync(send)
        self.g.slots[self.rank] = (send, list(send_rows))
        self.g.bar.wait()
        off = 0
        for p in range(w):
            s, rows = self.g.slots[p]
            a = sum(rows[:self.rank])
            k = rows[self.rank]
            assert k == recv_rows[p], "rank %d expects %d rows from %d, which sends %d" % (self.rank, recv_rows[p], p, k)
            if k:
                recv[off:off + k].copy_(s[a:a + k])
            off += k
        self._sync(recv)
        self.g.bar.wait()
