import os
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer CPU-only oracle runs")


def pytest_sessionstart(session):
    """On a GPU box, let torch initialise its HIP runtime BEFORE this repository's library touches the device: torch
    wheels carry their own libamdhip64, and in a process where the system's copy (what libsplat2d_hip.so links) came up
    first torch later finds "no HIP GPUs" -- so tests that use torch tensors beside the library (the rank-thread and
    row-level ABI tests) would depend on which test ran first.  bench.py imports torch first for the same reason."""
    if os.path.exists("/dev/kfd"):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:  # noqa: BLE001 - CPU-only runs and torch-less environments need none of this
            pass
