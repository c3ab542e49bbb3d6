#!/usr/bin/env python3
"""GPU-box tool (diagnostic build only: -DS2D_EXP_CLOCK, S2D_LIBRARY pointing at it): where a tile's workgroup spends its
life in raster_fused_kernel.  Four s_memrealtime stamps (100 MHz) per tile: kernel entry, end of the forward walk, end of the
backward walk, all its stores and atomics acknowledged.  usage: gpu_tile_clock.py [W H N [warm]]"""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S2D = importlib.import_module("2dgaussiansplatting_amd")
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 40
tiles = ((W + 15) // 16) * ((H + 15) // 16)
with S2D.Trainer(W, H, n) as t:
    t.lean_backward = True
    t.set_target_synthetic(); t.init()
    t.step(warm, want_mse=False); t.synchronize()
    fn = t.L.s2d_exp_read_probe
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]; fn.restype = C.c_int
    buf = np.zeros((tiles, 4), dtype=np.uint64)
    assert fn(t._h, buf.ctypes.data, tiles) == 0
s = buf.astype(np.int64) * 10e-3  # -> microseconds
t0, t1, t2, t3 = s[:, 0], s[:, 1], s[:, 2], s[:, 3]
start, end = t0.min(), t3.max()
def q(x): return "mean %.1f  median %.1f  5%%..95%% %.1f..%.1f  max %.1f" % (x.mean(), np.median(x), np.percentile(x, 5), np.percentile(x, 95), x.max())
print("%dx%d n=%d: launch spans %.1f us from the first workgroup's entry to the last one's drain" % (W, H, n, end - start))
print("forward walk   [us]: " + q(t1 - t0))
print("backward walk  [us]: " + q(t2 - t1))
print("drain (vmcnt 0)[us]: " + q(t3 - t2))
print("whole workgroup[us]: " + q(t3 - t0))
print("resident workgroups by their own clocks: sum of lives / span = %.0f" % ((t3 - t0).sum() / (end - start)))
# occupancy over time: how many workgroups are between entry and drain at each instant
ev = np.concatenate([np.stack([t0, np.ones_like(t0)], 1), np.stack([t3, -np.ones_like(t3)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
occ = np.cumsum(ev[:, 1])
for a, b in ((0.0, 0.05), (0.05, 0.5), (0.5, 0.95), (0.95, 1.0)):
    lo, hi = start + a * (end - start), start + b * (end - start)
    m = (ev[:, 0] >= lo) & (ev[:, 0] < hi)
    print("  %3.0f%%..%3.0f%% of the span: %.0f workgroups in flight on average" % (100 * a, 100 * b, occ[m].mean() if m.any() else 0.0))
order = np.argsort(t0)
gap = []
print("entries per 10 us in the middle of the launch: %.0f" % (((t0 > start + 0.4 * (end - start)) & (t0 < start + 0.6 * (end - start))).sum() / (0.2 * (end - start) / 10)))
