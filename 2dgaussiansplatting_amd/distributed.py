"""Row-slab data parallelism: one process per GPU, gradient rows are the only per-iteration exchange.

Partition: the image is cut into `world` contiguous slabs of whole 16-pixel tile rows; per iteration
every rank rasterises its slab forward and backward (partial gradients, partial squared error).  Framebuffers and
parameters are never exchanged.  Two exchange schemes sit on that partition:
  * HaloStep  -- slab OWNERSHIP (bench.py's default for N > 1): a rank holds, lists and updates only the splats that
                 can reach its rows (plus a halo margin) and exchanges gradient rows only for splats held by more than
                 one rank -- one all_to_all per iteration;
  * SlabStep  -- splats and Adam state replicated on every rank, the N x 9 fp32 gradient array all-reduced (sum; 36 MB
                 at N = 10^6) -- RCCL over xGMI on GPUs, gloo in the CPU tests -- and every rank applies the identical
                 Adam step, so the replicas stay bit-identical (bench.py --exchange dense: north_star's scheme).

`backend` is anything with forward() / backward() / adam_step() (and optionally forward_backward(), the fused form):
the HIP Trainer in bench.py, or the oracle-backed stand-in of tests/test_distributed_cpu.py.
"""


def _raster(backend, before, after):
    """Forward + backward of one iteration: the fused launch where the backend has one."""
    if before is not None:
        before()
    fused = getattr(backend, "forward_backward", None)
    if fused is not None:
        fused()
    else:
        backend.forward()
        backend.backward()
    if after is not None:
        after()


def slab_rows(height, rank, world, tile=16):
    """Rows [r0, r1) of rank `rank`: whole tile rows, as even as possible, r0 % tile == 0."""
    tile_rows = (height + tile - 1) // tile
    t0 = tile_rows * rank // world
    t1 = tile_rows * (rank + 1) // world
    return t0 * tile, min(t1 * tile, height)


class SlabStep:
    """One training iteration of one rank.  The collective and the backend must work on the same stream (bench.py
    creates the Trainer on torch's current stream), or the caller must synchronise between them.

    The replicas stay bit-identical only because every rank applies the same Adam step to the same all-reduced
    gradients; `params` (a callable returning the rank's parameters as a tensor) + `check_interval` assert that every so
    many iterations with a checksum exchanged over the same collective."""

    def __init__(self, backend, grads, dist=None, params=None, check_interval=0):
        self.backend = backend
        self.grads = grads  # torch tensor aliasing the backend's gradient buffer (n * 9 fp32)
        self.dist = dist    # torch.distributed module, or None for a single process
        self.params, self.check_interval, self.it, self.checks = params, int(check_interval), 0, 0

    def __call__(self, before_raster=None, after_raster=None):
        _raster(self.backend, before_raster, after_raster)
        if self.dist is not None:
            self.dist.all_reduce(self.grads)  # sum of the slabs' partial gradients, in place
        self.backend.adam_step()
        self.it += 1
        if self.dist is not None and self.params is not None and self.check_interval > 0 and self.it % self.check_interval == 0:
            self.assert_replicas_identical()

    def assert_replicas_identical(self):
        """Every rank's parameters have the same bits: a 64-bit checksum of them (sum of the words, and of the words
        times their position) must come back from a MIN and a MAX all-reduce unchanged."""
        import torch
        w = self.params().contiguous().view(torch.int32).flatten().to(torch.int64)
        pos = torch.arange(1, w.numel() + 1, dtype=torch.int64, device=w.device)
        mine = torch.stack([w.sum(), (w * (pos % 65521)).sum()])
        lo, hi = mine.clone(), mine.clone()
        if self.dist.get_backend() == "gloo" and lo.is_cuda:
            lo, hi = lo.cpu(), hi.cpu()
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN)
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX)
        self.checks += 1
        if not bool((lo == hi).all()):
            raise RuntimeError("row-slab replicas diverged: parameter checksums differ between ranks at iteration %d" % self.it)


def reduce_sqerr(sqerr, dist=None):
    """Sum the per-iteration partial squared errors of the slabs (one collective after many steps)."""
    if dist is not None:
        dist.all_reduce(sqerr)
    return sqerr


# ---------------------------------------------------------------------------------------------------------
# Slab ownership with a halo (include/splat2d.h "Slab ownership", csrc/s2d_halo.hip)
#
# Rank q HOLDS splat i while [pos.y - reach - margin, pos.y + reach + margin] meets its rows (reach = 3*max(sx,sy)+2).
# Invariants kept by HaloStep:
#   (1) every rank whose rows a splat touches holds it (the margin is sized for Adam's worst-case step, see
#       __init__, and hold sets are refreshed every `rehalo_interval` = 64 iterations; a splat that
#       arrives already touching the receiver's rows raises);
#   (2) all holders of a splat have bit-identical parameters and Adam state: they add the holders' partial
#       gradients in ascending rank order (ops.grads_combine) and run the same Adam kernel;
#   (3) mask[i] (bit q = rank q holds i) is identical on all holders of i, 0 elsewhere.
# Between refreshes the exchange lists are frozen, so an ordinary iteration costs one gather kernel, one
# all_to_all_single with fixed split sizes and one combine kernel -- no host synchronisation.
# ---------------------------------------------------------------------------------------------------------

ROWS_GRADS, ROWS_SPLATS, ROWS_ADAM = 0, 1, 2
_ROW_WIDTH = {ROWS_GRADS: 9, ROWS_SPLATS: 9, ROWS_ADAM: 18}


class HipHaloOps:
    """The row operations of HaloStep on the HIP Trainer: torch tensors in, C-ABI calls with their device pointers.

    The library's kernels and torch's (index building, the collective) touch the same buffers, so they must be
    ordered: create the Trainer on torch's current stream (`Trainer(..., stream=torch.cuda.current_stream().cuda_stream)`
    on a non-default stream, as bench.py does) and everything is stream-ordered for free.  If the Trainer works on
    another stream (e.g. the one the library creates when none is given) every call here synchronises both sides
    instead -- correct, but it stalls the host; meant for small tests."""

    def __init__(self, trainer, n, device):
        import torch
        self.torch, self.t, self.n, self.device = torch, trainer, n, device
        cur = torch.cuda.current_stream().cuda_stream
        self.shared_stream = trainer.stream is not None and trainer.stream != 0 and trainer.stream == cur

    def _enter(self):  # torch-produced inputs must be complete before the library's stream reads them
        if not self.shared_stream:
            self.torch.cuda.current_stream().synchronize()

    def _exit(self):   # ... and the library's results before torch reads them
        if not self.shared_stream:
            self.t.synchronize()

    def halo_masks(self, bounds, margin):
        m = self.torch.empty(self.n, dtype=self.torch.int32, device=self.device)
        self._enter()
        self.t.halo_masks(list(bounds), float(margin), m.data_ptr())
        self._exit()
        return m

    def halo_commit(self, mask, rank, added=True):
        assert mask.dtype == self.torch.int32 and mask.is_contiguous() and mask.numel() == self.n
        self._enter()
        self.t.halo_commit(mask.data_ptr(), rank, 1 if added else 0)
        self._exit()

    def rows_gather(self, what, ids, out=None):
        k = ids.numel()
        if out is None:
            out = self.torch.empty((k, _ROW_WIDTH[what]), dtype=self.torch.float32, device=self.device)
        assert ids.dtype == self.torch.int32 and ids.is_contiguous() and out.is_contiguous()
        if k:
            self._enter()
            self.t.rows_gather(what, ids.data_ptr(), k, out.data_ptr())
            self._exit()
        return out

    def rows_scatter(self, what, ids, values):
        k = ids.numel()
        assert ids.dtype == self.torch.int32 and ids.is_contiguous() and values.is_contiguous()
        assert values.dtype == self.torch.float32 and values.numel() == k * _ROW_WIDTH[what]
        if k:
            self._enter()
            self.t.rows_scatter(what, ids.data_ptr(), k, values.data_ptr())
            self._exit()

    def grads_combine(self, rows, src, recv):
        k = rows.numel()
        assert rows.dtype == self.torch.int32 and src.dtype == self.torch.int32 and src.is_contiguous() and recv.is_contiguous()
        if k:
            self._enter()
            self.t.grads_combine(rows.data_ptr(), k, src.data_ptr(), src.shape[1], recv.data_ptr())
            self._exit()


def _all_to_all_rows(dist, recv, send, recv_rows, send_rows):
    """all_to_all_single along dim 0 with per-rank row counts.  gloo cannot move device tensors (CPU rehearsals of
    the GPU path only): stage through host memory there."""
    if dist.get_backend() == "gloo" and send.is_cuda:
        r, s = recv.cpu(), send.cpu()
        dist.all_to_all_single(r, s, recv_rows, send_rows)
        recv.copy_(r)
    else:
        dist.all_to_all_single(recv, send, recv_rows, send_rows)


def all_to_all_selftest(dist, device):
    """The collective pattern HaloStep relies on, on tiny tensors: all_to_all_single with UNEVEN row counts, zero-size
    segments for non-neighbours, float32 and int32 payloads, an exchange in which nobody sends anything, plus the
    equal-split int64 count exchange.  Returns the
    same verdict on every rank (an all-reduce of the local results), so callers can fall back together."""
    import torch
    ok = 1
    try:
        w, r = dist.get_world_size(), dist.get_rank()
        near = [p for p in range(w) if p != r and abs(p - r) == 1]
        send_rows = [(p + 1) if p in near else 0 for p in range(w)]   # rank r sends p+1 rows to neighbour p
        recv_rows = [(r + 1) if p in near else 0 for p in range(w)]   # ... and so receives r+1 rows from each
        for dtype in (torch.float32, torch.int32):
            send = torch.cat([torch.full((k, 9), r * 100 + p, dtype=dtype, device=device) for p, k in enumerate(send_rows)])
            recv = torch.full((sum(recv_rows), 9), -1, dtype=dtype, device=device)
            _all_to_all_rows(dist, recv, send, recv_rows, send_rows)
            want = torch.cat([torch.full((k, 9), p * 100 + r, dtype=dtype, device=device) for p, k in enumerate(recv_rows)])
            ok &= int(torch.equal(recv, want))
        # a refresh in which no splat changes hands: nothing to send or receive on any rank (zero-size tensors)
        none = [0] * w
        _all_to_all_rows(dist, torch.empty((0, 29), dtype=torch.int32, device=device),
                         torch.empty((0, 29), dtype=torch.int32, device=device), none, none)
        cnt_out = torch.arange(w, dtype=torch.int64) + 10 * r
        cnt_in = torch.empty_like(cnt_out)
        if dist.get_backend() == "gloo":
            dist.all_to_all_single(cnt_in, cnt_out)
        else:
            co, ci = cnt_out.to(device), cnt_in.to(device)
            dist.all_to_all_single(ci, co)
            cnt_in = ci.cpu()
        ok &= int(cnt_in.tolist() == [r + 10 * p for p in range(w)])
    except Exception:  # noqa: BLE001 - any failure means "do not use this path"
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int64)
    f = flag if dist.get_backend() == "gloo" else flag.to(device)
    dist.all_reduce(f, op=dist.ReduceOp.MIN)
    return int(f.cpu().item()) == 1


class HaloStep:
    """One training iteration of one rank under slab ownership.  `backend`: forward() / backward() / adam_step();
    `ops`: HipHaloOps or an equivalent (tests use an oracle-backed one).  Construct it while every rank still has the
    same, complete set of splats (after init / set_splats / a checkpoint load on all ranks)."""

    def __init__(self, backend, ops, dist, rank, world, height, rehalo_interval=64, margin_rows=None, lr=0.05):
        import torch
        self.torch = torch
        self.backend, self.ops, self.dist = backend, ops, dist
        self.rank, self.world = rank, world
        self.bounds = [slab_rows(height, q, world)[0] for q in range(world)] + [height]
        self.r0, self.r1 = self.bounds[rank], self.bounds[rank + 1]
        self.interval = int(rehalo_interval)
        # Adam moves a parameter by lr*|m^|/sqrt(v^) <= 2.35*lr per step for beta = (0.9, 0.99)
        # ((1-b1)/sqrt(1-b2) / sqrt(1 - b1^2/b2)); pos.y moves by that and reach = 3*max(sx,sy)+2 by three times
        # that: the margin must outlast one refresh interval in the worst case (33 rows at 64 iterations)
        self.margin = float(margin_rows) if margin_rows is not None else max(8.0, 1.1 * 2.35 * 4.0 * lr * self.interval)
        self.it = 0
        self.handed_over = 0  # splats whose state this rank received or sent since the start (diagnostic)
        assert 1 <= world <= 32 and self.interval >= 1 and self.margin >= 0.0
        # every rank starts from the same full set of splats: no hand-over needed for the first hold sets
        m = ops.halo_masks(self.bounds, self.margin)
        self.mask = torch.where(((m >> rank) & 1) == 1, m, torch.zeros_like(m)).contiguous()  # invariant (3)
        ops.halo_commit(self.mask, rank)
        self._plan()
        if world > 1:
            # a refresh on unchanged parameters moves nothing, but it walks every collective of the steady state
            # once (the all-pairs count exchange opens the peer-to-peer connections) outside any timed region
            self._rehalo()

    def _bit_counts(self, words):
        """Per bit q < world: how many entries of the int32 tensor have it set (one device reduction, one sync)."""
        t = self.torch
        if not hasattr(self, "_shifts"):
            self._shifts = t.arange(self.world, dtype=t.int32, device=words.device)
        return [int(v) for v in ((words[:, None] >> self._shifts[None, :]) & 1).sum(0).tolist()]

    # -- exchange lists for the frozen hold sets
    def _plan(self):
        t, m, r = self.torch, self.mask, self.rank
        # m is 0 outside this rank's hold set, so bit p of m counts the rows shared with rank p: one reduction and
        # one host read for all peers, then a nonzero() only for the (usually two) peers that share anything
        shared = self._bit_counts(m)
        self.peer_ids = []
        for p in range(self.world):
            if p == r or shared[p] == 0:
                self.peer_ids.append(m.new_empty(0))
            else:
                self.peer_ids.append(t.nonzero((m >> p) & 1).flatten().to(t.int32))
        self.splits = [int(x.numel()) for x in self.peer_ids]
        total = sum(self.splits)
        self.send_ids = t.cat(self.peer_ids) if total else m.new_empty(0)
        dev = m.device
        self.send_buf = t.empty((total, 9), dtype=t.float32, device=dev)
        self.recv_buf = t.empty((total, 9), dtype=t.float32, device=dev)
        if total:
            rows = t.unique(self.send_ids.long(), sorted=True)
            src = t.full((rows.numel(), self.world), -1, dtype=t.int32, device=dev)
            src[:, r] = -2
            base = 0
            for p, k in enumerate(self.splits):
                if k:
                    pos = t.searchsorted(rows, self.peer_ids[p].long())
                    src[pos, p] = t.arange(base, base + k, dtype=t.int32, device=dev)
                    base += k
            self.rows, self.src = rows.to(t.int32).contiguous(), src.contiguous()
        else:
            self.rows, self.src = m.new_empty(0), t.empty((0, self.world), dtype=t.int32, device=dev)
        # skip the collective only if NO rank shares anything (all ranks must agree on calling it)
        flag = t.tensor([total], dtype=t.int64)
        if self.dist is not None and self.world > 1:
            f = flag if self.dist.get_backend() == "gloo" else flag.to(dev)
            self.dist.all_reduce(f, op=self.dist.ReduceOp.MAX)
            flag = f.cpu()
        self.any_exchange = int(flag.item()) > 0

    def _exchange_grads(self):
        if not self.any_exchange:
            return
        send = self.ops.rows_gather(ROWS_GRADS, self.send_ids, out=self.send_buf)
        _all_to_all_rows(self.dist, self.recv_buf, send, self.splits, self.splits)
        self.ops.grads_combine(self.rows, self.src, self.recv_buf)

    # -- refresh the hold sets from the current parameters; hand the state of arriving splats over
    def _rehalo(self):
        t, r, ops, dist = self.torch, self.rank, self.ops, self.dist
        old = self.mask
        new = ops.halo_masks(self.bounds, self.margin)  # 0 where this rank holds nothing
        held = ((old >> r) & 1) == 1
        sender = held & ((old & (-old)) == (1 << r))    # the lowest-ranked old holder hands a splat over
        arriving = t.where(sender, new & ~old, t.zeros_like(new))  # bit q: this rank hands the splat to rank q
        n_out = self._bit_counts(arriving)
        out_ids = []
        for q in range(self.world):
            if q == r or n_out[q] == 0:
                out_ids.append(old.new_empty(0))
            else:
                out_ids.append(t.nonzero((arriving >> q) & 1).flatten().to(t.int32))
        out_rows = [int(x.numel()) for x in out_ids]
        dev = old.device
        cnt_out = t.tensor(out_rows, dtype=t.int64)
        cnt_in = t.empty_like(cnt_out)
        if dist.get_backend() == "gloo":
            dist.all_to_all_single(cnt_in, cnt_out)
        else:
            co, ci = cnt_out.to(dev), cnt_in.to(dev)
            dist.all_to_all_single(ci, co)
            cnt_in = ci.cpu()
        in_rows = [int(v) for v in cnt_in.tolist()]
        ids = t.cat(out_ids) if sum(out_rows) else old.new_empty(0)
        k = ids.numel()
        # one row per splat: id, new mask, 9 parameters, 18 Adam moments -- as raw 32-bit words
        pay = t.empty((k, 29), dtype=t.int32, device=dev)
        if k:
            pay[:, 0] = ids
            pay[:, 1] = new[ids.long()]
            pay[:, 2:11] = ops.rows_gather(ROWS_SPLATS, ids).view(t.int32)
            pay[:, 11:29] = ops.rows_gather(ROWS_ADAM, ids).view(t.int32)
        got = t.empty((sum(in_rows), 29), dtype=t.int32, device=dev)
        _all_to_all_rows(dist, got, pay, in_rows, out_rows)
        mask = t.where(held & (((new >> r) & 1) == 1), new, t.zeros_like(new))  # 0 for splats dropped or never held
        late = 0
        if got.shape[0]:
            rid = got[:, 0].contiguous()
            sp = got[:, 2:11].contiguous().view(t.float32)
            ad = got[:, 11:29].contiguous().view(t.float32)
            reach = 3.0 * t.maximum(sp[:, 2], sp[:, 3]) + 2.0
            touching = (sp[:, 1] + reach >= float(self.r0)) & (sp[:, 1] - reach <= float(self.r1))
            late = int(touching.sum().item())
        # invariant (1) broken anywhere is fatal everywhere: agree on it, so that no rank is left in a collective
        worst = t.tensor([late], dtype=t.int64)
        w = worst if dist.get_backend() == "gloo" else worst.to(dev)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        if int(w.cpu().item()) > 0:
            raise RuntimeError("slab ownership: a splat reached the rows of a rank before its state was handed over "
                               "(%d on rank %d); raise margin_rows or lower rehalo_interval" % (late, r))
        if got.shape[0]:
            ops.rows_scatter(ROWS_SPLATS, rid, sp)
            ops.rows_scatter(ROWS_ADAM, rid, ad)
            mask[rid.long()] = got[:, 1]
        self.handed_over += k + got.shape[0]
        self.mask = mask.contiguous()
        ops.halo_commit(self.mask, r, added=got.shape[0] > 0)  # departures alone leave the tile lists valid
        self._plan()

    def __call__(self, before_raster=None, after_raster=None):
        _raster(self.backend, before_raster, after_raster)
        if self.world > 1:
            self._exchange_grads()
        self.backend.adam_step()
        self.it += 1
        if self.world > 1 and self.it % self.interval == 0:
            self._rehalo()

    # -- test / checkpoint path: the full array, every row taken from its lowest-ranked holder
    def gather_full(self, what):
        t, r = self.torch, self.rank
        n = self.mask.numel()
        vals = self.ops.rows_gather(what, t.arange(n, dtype=t.int32, device=self.mask.device))
        own = (((self.mask >> r) & 1) == 1) & ((self.mask & (-self.mask)) == (1 << r))
        vals = t.where(own[:, None], vals, t.zeros_like(vals))
        if self.world > 1:
            if self.dist.get_backend() == "gloo" and vals.is_cuda:
                v = vals.cpu()
                self.dist.all_reduce(v)
                vals = v.to(vals.device)
            else:
                self.dist.all_reduce(vals)
        return vals
