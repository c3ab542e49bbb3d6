#!/usr/bin/env python3
"""GPU-box tool: where the waves of the raster kernels spend their cycles (needs tools/build_timing_lib.py's
diagnostic library).  Prints, per kernel, the share of summed wave lifetime per phase and cycles per executed
(wave, entry)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "2dgaussiansplatting_amd", "lib", "libsplat2d_hip_timing.so")
os.environ["S2D_LIBRARY"] = lib
S2D = importlib.import_module("2dgaussiansplatting_amd")


def run(W, H, n, warm=20, rows=None):
    kw = {} if rows is None else {"row_begin": rows[0], "row_end": rows[1]}
    with S2D.Trainer(W, H, n, **kw) as t:
        t.lean_backward = True
        t.set_target_synthetic()
        t.init()
        t.step(warm, want_mse=False)
        a = t.stats()["phase_cycles"]
        iters = 10
        t.step(iters, want_mse=False)
        b = t.stats()["phase_cycles"]
    d = [(y - x) / iters for x, y in zip(a, b)]
    print("== %dx%d / %d splats%s (per iteration, cycles summed over waves)" % (W, H, n, "" if rows is None else " rows %s" % (rows,)))
    for name, base, phases in (("forward", 0, ["staging", "barrier1", "blend", "barrier2"]),
                               ("backward", 8, ["staging", "barrier1", "blend", "barrier2", "flush"])):
        total, execs = d[base + 6], d[base + 7]
        print("  %-8s wave-cycles %.3e  execs %.3e" % (name, total, execs))
        acc = 0.0
        for k, ph in enumerate(phases):
            acc += d[base + k]
            print("     %-9s %5.1f %%   (%.0f cycles per executed (wave, entry))" % (ph, 100 * d[base + k] / total, d[base + k] / max(execs, 1)))
        print("     %-9s %5.1f %%" % ("other", 100 * (total - acc) / total))


if __name__ == "__main__":
    run(4096, 4096, 1_000_000)
    run(4096, 4096, 1_000_000, rows=(2048, 2560))
    run(2048, 2048, 250_000)
    run(268, 213, 1024)
