"""GPU-box tool: a short single-context run for a kernel trace: python3 tools/gpu_mode_trace.py W H N iters [deterministic|fp16|plain]"""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
S2D = importlib.import_module("2dgaussiansplatting_amd")
W, H, n, iters = (int(a) for a in sys.argv[1:5])
mode = sys.argv[5] if len(sys.argv) > 5 else "plain"
with S2D.Trainer(W, H, n, deterministic=mode == "deterministic", fp16_images=mode == "fp16") as t:
    t.set_target_synthetic()
    t.init()
    t.step(16, want_mse=False)
    t.step(iters, want_mse=False)
    t.synchronize()
