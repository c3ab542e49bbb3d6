#!/usr/bin/env python3
"""GPU-box tool: pair/lane statistics of one iteration at a given size (diagnostic counters on)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S2D = importlib.import_module("2dgaussiansplatting_amd")
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 0
with S2D.Trainer(W, H, n) as t0:
    t0.set_target_synthetic(); t0.init()
    if warm: t0.step(warm, want_mse=False)
    sp = t0.get_splats()
with S2D.Trainer(W, H, n, count_pairs=True) as t:
    t.set_target_synthetic(); t.set_splats(sp)
    t.forward(); t.backward(); t.synchronize()
    st = t.stats()
tiles = ((W + 15) // 16) * ((H + 15) // 16)
px = W * H
print("size %dx%d n=%d after %d iterations" % (W, H, n, warm))
print("pairs binned %d  (%.1f per tile, %.1f tiles per splat)" % (st["pairs_binned"], st["pairs_binned"] / tiles, st["pairs_binned"] / n))
for d in ("fwd", "bwd"):
    print("%s: staged %.1f/tile (%.0f%% of list)  visited %.1f/px  active %.1f/px  wave-execs %.1f/wave  lanes/exec %.1f" % (
        d, st[d + "_staged"] / tiles, 100.0 * st[d + "_staged"] / max(st["pairs_binned"], 1), st[d + "_visited"] / px,
        st[d + "_active"] / px, st[d + "_wave_execs"] / (tiles * 4), st[d + "_active"] / max(st[d + "_wave_execs"], 1)))
print("fwd staging: %.0f%% of staged entries cover a pixel of their tile; %.1f of 16 tile rows non-empty per staged entry (%.1f per covering entry)" % (
    100.0 * st["fwd_staged_hit"] / max(st["fwd_staged"], 1), st["fwd_rows_hit"] / max(st["fwd_staged"], 1), st["fwd_rows_hit"] / max(st["fwd_staged_hit"], 1)))
h = np.array(st["bwd_lane_hist"], dtype=np.float64)
tot = h.sum()
cum = np.cumsum(h) / tot
work = np.cumsum(h * np.arange(65)) / (h * np.arange(65)).sum()
print("bwd execs by active lanes: " + "  ".join("<=%d: %.0f%% of execs, %.0f%% of lane-work" % (k, 100 * cum[k], 100 * work[k]) for k in (4, 8, 16, 24, 32, 48, 63)))
q = st["bwd_quadrant_execs"] / max(st["bwd_wave_execs"], 1)
print("bwd: %.2f of 4 quadrants (4x4 pixels) live per executed (wave, entry): %.1f live lanes per live quadrant; "
      "quadrant-granular execution would need %.0f%% of the lane-slots" % (q, st["bwd_active"] / max(st["bwd_quadrant_execs"], 1), 100.0 * q / 4))
