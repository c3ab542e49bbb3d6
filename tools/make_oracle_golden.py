#!/usr/bin/env python3
"""Generate tests/golden/oracle_cfg1_it5.npz from the CPU oracle (which tests/test_oracle_kat.py pins to the
survey's known-answer vectors): BASELINE configs[0] (mini image, N = 2000) advanced 5 iterations, then one
forward + backward.  Inputs and expected outputs only; run in the build container:  python tools/make_oracle_golden.py
"""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O

tgt = O.target_rgba32f(O.load_s2di(os.path.join(O.GOLDEN, "squirrel_cls_mini_268x213.s2di")))
o = O.OracleTrainer(tgt, 2000)
trace = [o.step()[1] for _ in range(5)]
state = dict(splats=o.splats.copy(), adams=o.adams.copy(), beta1t=o.beta1t.copy(), beta2t=o.beta2t.copy(),
             iterations=np.int32(o.iterations))
img = o.forward().copy()
c = O.Counters()
o.backward(counters=c)
g32, dsum, dabs = o.backward_stats()
g32 = g32.copy()
mse = o.mse()
st, mse_step = o.step()
out = os.path.join(O.GOLDEN, "oracle_cfg1_it5.npz")
np.savez_compressed(out, **state, mse_trace_0_4=np.array(trace), image_sha256=np.array(hashlib.sha256(img.tobytes()).hexdigest()),
                    image_rows_100_103=img[100:104].copy(), grads_fp32=g32.view(np.float32).reshape(-1, 9),
                    grads_exact_sum=dsum, grads_abs_sum=dabs.astype(np.float32), active_pairs=np.uint64(c.active), mse=np.float64(mse),
                    splats_after_step=o.splats.copy(), mse_of_step=np.float64(mse_step))
print(out, os.path.getsize(out), "bytes; mse", mse, "active", c.active)

# The overlay's vertex list (main.cpp:419-477, oracle s2do_overlay_vertices) of the scene as shipped (N = 1024) at init() and
# after 10 oracle iterations: the first three splats' 46 vertices verbatim, and a sha256 of all of them.
o = O.OracleTrainer(tgt, 1024)
ov = {}
for tag in ("it0", "it10"):
    xyz, rgb = O.overlay_vertices(o.splats)
    ov["xyz_first3_" + tag] = xyz[:3 * O.OVERLAY_VERTICES].copy()
    ov["rgb_first3_" + tag] = rgb[:3 * O.OVERLAY_VERTICES].copy()
    ov["sha256_" + tag] = np.array(hashlib.sha256(xyz.tobytes() + rgb.tobytes()).hexdigest())
    for _ in range(10):
        o.step()
out = os.path.join(O.GOLDEN, "overlay_vertices_mini_1024.npz")
np.savez_compressed(out, **ov)
print(out, os.path.getsize(out), "bytes")
