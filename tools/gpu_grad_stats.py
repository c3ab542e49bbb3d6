#!/usr/bin/env python3
"""GPU-box tool: distribution of gradient differences (GPU vs oracle fp32 vs exactly-summed terms)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
S2D = importlib.import_module("2dgaussiansplatting_amd")

def run(name, tgt, n, steps, opacity=False, splats=None):
    o = O.OracleTrainer(tgt, n, optimize_opacity=opacity)
    if splats is not None:
        o.splats[:] = splats
    for _ in range(steps):
        o.step()
    o.forward()
    w32, dsum, dabs = o.backward_stats()
    w = w32.view(np.float32).reshape(-1, 9).astype(np.float64)
    with S2D.Trainer(o.W, o.H, n) as t:
        t.set_target(tgt); t.set_splats(o.splats.view(S2D.SPLAT_DTYPE))
        t.forward(); t.backward()
        g = t.get_grads().view(np.float32).reshape(-1, 9).astype(np.float64)
    nz = dabs > 0
    eg = np.abs(g - dsum)[nz] / dabs[nz]
    er = np.abs(w - dsum)[nz] / dabs[nz]
    d = np.abs(g - w)
    rel = d[nz] / np.maximum(np.abs(w[nz]), 1e-300)
    cond = dabs[nz] / np.maximum(np.abs(dsum[nz]), 1e-300)
    print("%s: n=%d steps=%d" % (name, n, steps))
    print("   |gpu-exact|/dabs   max %.2e  p99 %.2e  med %.2e" % (eg.max(), np.percentile(eg, 99), np.median(eg)))
    print("   |ref32-exact|/dabs max %.2e  p99 %.2e  med %.2e" % (er.max(), np.percentile(er, 99), np.median(er)))
    print("   |gpu-ref32|/|ref32| max %.2e p99.9 %.2e p99 %.2e med %.2e ; frac>1e-4: %.2e" % (rel.max(), np.percentile(rel, 99.9), np.percentile(rel, 99), np.median(rel), (rel > 1e-4).mean()))
    for kappa in (1.0, 0.1, 0.05, 0.02, 0.01):
        e = d[nz] / np.maximum(np.abs(w[nz]), kappa * dabs[nz])
        print("   kappa=%.2f: max |gpu-ref32|/max(|ref32|,kappa*dabs) = %.2e" % (kappa, e.max()))
    print("   condition dabs/|sum|: med %.1f p99 %.1f max %.1e" % (np.median(cond), np.percentile(cond, 99), cond.max()))
    assert np.all(g[~nz] == 0)

mini = O.target_rgba32f(O.load_s2di(os.path.join(O.GOLDEN, "squirrel_cls_mini_268x213.s2di")))
run("mini", mini, 1024, 0)
run("mini", mini, 2000, 5)
run("mini-opacity", mini, 1024, 30, True)
run("mini", mini, 2000, 60)
full = O.target_rgba32f(O.load_s2di(os.path.join(O.GOLDEN, "squirrel_cls_535x426.s2di")))
run("native", full, 50000, 2)
