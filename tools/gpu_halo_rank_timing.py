#!/usr/bin/env python3
"""GPU-box tool: what ONE rank of an N-GPU run costs per iteration, without the wire.

Runs the middle slab of an N-way split of the 4096x4096 / 1M workload alone on the GPU, through the real HaloStep /
SlabStep host code, with a loopback object in place of torch.distributed (all_to_all returns the rows this rank
sent, as a symmetric neighbour would; all_reduce is a no-op).  The gradients are therefore wrong -- this measures
time only: raster on 1/N of the rows + Adam on the held splats + gather/combine kernels + periodic refresh and
list rebuild.  The collective's latency comes on top on real hardware.
"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S2D = importlib.import_module("2dgaussiansplatting_amd")
D = importlib.import_module("2dgaussiansplatting_amd.distributed")


class Loopback:
    class ReduceOp:
        SUM, MAX = "sum", "max"

    def get_backend(self):
        return "loopback"

    def all_reduce(self, t, op=None):
        pass

    def all_to_all_single(self, recv, send, recv_rows=None, send_rows=None):
        if recv.dim() == 1:      # the count exchange of a refresh: nobody hands anything to this rank
            recv.zero_()
        elif recv.shape == send.shape:
            recv.copy_(send)


def main():
    W = H = 4096
    n = 1000000
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    worlds = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8]
    schemes = sys.argv[3].split(",") if len(sys.argv) > 3 else None
    for world in worlds:
        rank = world // 2
        for scheme in (schemes or (("halo", "dense") if world > 1 else ("single",))):
            stream = torch.cuda.Stream()
            torch.cuda.set_stream(stream)
            r0, r1 = D.slab_rows(H, rank, world)
            grads = torch.zeros(n * 9, dtype=torch.float32, device="cuda")
            with S2D.Trainer(W, H, n, row_begin=r0, row_end=r1, stream=stream.cuda_stream) as t:
                t.bind_grads(grads.data_ptr())
                t.lean_backward = True
                t.set_target_synthetic()
                t.init()
                dist = Loopback() if world > 1 else None
                if scheme == "halo":
                    step = D.HaloStep(t, D.HipHaloOps(t, n, "cuda"), dist, rank, world, H,
                                          rehalo_interval=int(os.environ.get("S2D_REHALO", "32")))
                else:
                    step = D.SlabStep(t, grads, dist)
                for _ in range(16):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    step()
                host = (time.perf_counter() - t0) / steps   # the host's own time to QUEUE an iteration (it runs ahead of the device)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / steps
                extra = ""
                if scheme == "halo":
                    extra = "  held %.1f%%  rows exchanged/iter %d" % (100.0 * float(((step.mask >> rank) & 1).float().mean().item()), sum(step.splits))
                print("N=%d rank %d rows %4d..%4d  %-6s  %.3f ms/iteration (%.0f it/s if the wire were free; host queues one in %.3f ms)%s" % (
                    world, rank, r0, r1, scheme, 1e3 * dt, 1.0 / dt, 1e3 * host, extra), flush=True)


if __name__ == "__main__":
    main()
