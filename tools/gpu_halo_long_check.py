#!/usr/bin/env python3
"""GPU-box tool: long runs of slab ownership (HaloStep) against replicated state (SlabStep) on the same slabs, ranks as
threads of this process, deterministic gradients: the assembled parameters must be bit-identical after hundreds of
iterations and many refreshes / hand-overs.  usage: gpu_halo_long_check.py W H N world steps"""
import importlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from thread_dist import ThreadDist  # noqa: E402
S2D = importlib.import_module("2dgaussiansplatting_amd")
D = importlib.import_module("2dgaussiansplatting_amd.distributed")

W, H, n, world, steps = (int(v) for v in sys.argv[1:6])


def run(scheme):
    res = [None] * world

    def body(rank, dist):
        r0, r1 = D.slab_rows(H, rank, world)
        stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)  # shared by torch and the library, per rank
        grads = torch.zeros(n * 9, dtype=torch.float32, device="cuda")
        with S2D.Trainer(W, H, n, row_begin=r0, row_end=r1, deterministic=True, stream=stream.cuda_stream) as t:
            t.bind_grads(grads.data_ptr()); t.lean_backward = True
            t.set_target_synthetic(); t.init()
            step = D.HaloStep(t, D.HipHaloOps(t, n, "cuda"), dist, rank, world, H) if scheme == "halo" else D.SlabStep(t, grads, dist)
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            out = {"sq": t.sqerr_trace(max(steps - 256, 0), min(steps, 256))}
            if scheme == "halo":
                out["full"] = step.gather_full(D.ROWS_SPLATS).cpu().numpy()
                out["moved"] = step.handed_over
                out["held"] = float(((step.mask >> rank) & 1).float().mean().item())
            else:
                out["full"] = t.get_splats().view(np.float32).reshape(n, 9).copy()
            res[rank] = out

    t0 = time.time()
    ThreadDist(world).run(body)
    return res, time.time() - t0


mode = sys.argv[6] if len(sys.argv) > 6 else "halo-vs-dense"
a_name, b_name = {"halo-vs-dense": ("halo", "dense"), "dense-twice": ("dense", "dense"), "halo-twice": ("halo", "halo")}[mode]
a, ta = run(a_name)
b, tb = run(b_name)
same = a[0]["full"].tobytes() == b[0]["full"].tobytes()
sq_a, sq_b = sum(h["sq"] for h in a), sum(d["sq"] for d in b)
if a_name == "halo":
    print("state hand-overs %d; held per rank %s" % (sum(h["moved"] for h in a), " ".join("%.3f" % h["held"] for h in a)))
ndiff = int((a[0]["full"] != b[0]["full"]).any(axis=1).sum())
print("%dx%d n=%d world=%d steps=%d %s: %.1fs / %.1fs; parameters bit-identical: %s (%d splats differ, max |diff| %.3g); "
      "squared-error trace bit-identical: %s; last mse %.6f" % (W, H, n, world, steps, mode, ta, tb, same, ndiff,
      float(np.abs(a[0]["full"] - b[0]["full"]).max()), sq_a.tobytes() == sq_b.tobytes(), sq_a[-1] / (H * W * 3)), flush=True)
if not same:
    first = int(np.argmax(sq_a != sq_b)) if (sq_a != sq_b).any() else -1
    print("first differing iteration of the trace window:", first)
    sys.exit(1)
print("long check ok")
