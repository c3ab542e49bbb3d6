"""Row-slab data parallelism: one process per GPU, the per-splat gradients are the only exchange.

Partition: the image is cut into `world` contiguous slabs of whole 16-pixel tile rows; splats
and Adam state are replicated.  Per iteration every rank rasterises its slab forward and backward (partial
gradients, partial squared error), the N x 9 fp32 gradient array is all-reduced (sum) -- RCCL over xGMI on
GPUs, gloo in the CPU tests -- and every rank applies the identical Adam step, so the replicas stay bit-identical
without ever exchanging parameters or framebuffers.

`backend` is anything with forward() / backward() / adam_step(): the HIP Trainer bound to a torch gradient
tensor in bench.py, or the oracle-backed stand-in of tests/test_distributed_gloo.py.
"""


def slab_rows(height, rank, world, tile=16):
    """Rows [r0, r1) of rank `rank`: whole tile rows, as even as possible, r0 % tile == 0."""
    tile_rows = (height + tile - 1) // tile
    t0 = tile_rows * rank // world
    t1 = tile_rows * (rank + 1) // world
    return t0 * tile, min(t1 * tile, height)


class SlabStep:
    """One training iteration of one rank."""

    def __init__(self, backend, grads, dist=None):
        self.backend = backend
        self.grads = grads  # torch tensor aliasing the backend's gradient buffer (n * 9 fp32)
        self.dist = dist    # torch.distributed module, or None for a single process

    def __call__(self, after_forward=None, after_backward=None):
        self.backend.forward()
        if after_forward is not None:
            after_forward()
        self.backend.backward()
        if after_backward is not None:
            after_backward()
        if self.dist is not None:
            self.dist.all_reduce(self.grads)  # sum of the slabs' partial gradients, in place
        self.backend.adam_step()


def reduce_sqerr(sqerr, dist=None):
    """Sum the per-iteration partial squared errors of the slabs (one collective after many steps)."""
    if dist is not None:
        dist.all_reduce(sqerr)
    return sqerr
