#!/usr/bin/env python3
"""Container tool: committed oracle outputs for gradient parity at the big BASELINE configurations.

For each (W, H, N, window) the CPU oracle runs forward + backward on every init() splat that can reach the window
(tests/oracle_lib.py WindowOracle) and the nine gradient components of the splats whose whole footprint lies inside
the window are stored -- the oracle's fp32 sums, the same terms summed in double (dsum) and the sum of their
magnitudes (dabs) -- together with the window's framebuffer.  tests/test_gpu_fullsize.py compares the HIP path's
gradients of the FULL scene against them (bars (a)-(c) of tests/test_gpu_parity.py).

  python tools/make_window_golden.py      ->  tests/golden/window_*.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

# name, W, H, N, window (x0, y0, w, h)
CASES = [
    ("cfg3_2048_250k", 2048, 2048, 250_000, (896, 1200, 256, 256)),
    ("cfg4_4096_1m", 4096, 4096, 1_000_000, (1000, 2000, 256, 256)),
    ("cfg4_4096_1m_corner", 4096, 4096, 1_000_000, (3840, 0, 256, 192)),   # image corner: clipped footprints nearby
]


def main():
    for name, W, H, n, win in CASES:
        tgt = O.synthetic_target(W, H)
        s = np.zeros(n, dtype=O.SPLAT_DTYPE)
        O.lib().s2do_init(s.ctypes.data, None, n, W, H)
        wo = O.WindowOracle(tgt, s, win)
        img, w32, dsum, dabs = wo.run()
        x0, y0, w, h = win
        out = os.path.join(O.GOLDEN, "window_%s.npz" % name)
        np.savez_compressed(out, width=W, height=H, n_splats=n, window=np.array(win, dtype=np.int32),
                            inside_ids=wo.inside.astype(np.int32), n_candidates=len(wo.cand),
                            grads_f32=w32, dsum=dsum, dabs=dabs.astype(np.float32),
                            image=img[y0:y0 + h, x0:x0 + w].copy())
        print("%s: %d candidates, %d inside, %d KB" % (name, len(wo.cand), len(wo.inside), os.path.getsize(out) // 1024))


if __name__ == "__main__":
    main()
