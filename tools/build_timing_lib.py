#!/usr/bin/env python3
"""Container tool: build the DIAGNOSTIC library lib/libsplat2d_hip_timing.so (-DS2D_PHASE_TIMING): the shipped
kernels plus per-wave shader-clock accounting of the raster kernels' phases.  Never loaded by the product, tests or
bench.py; tools/gpu_phase_timing.py selects it through S2D_LIBRARY."""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = importlib.import_module("2dgaussiansplatting_amd._build")
out = os.path.join(B.LIB_DIR, "libsplat2d_hip_timing.so")
os.makedirs(B.LIB_DIR, exist_ok=True)
cmd = [B.hipcc()] + B.HIPCC_FLAGS + ["-DS2D_PHASE_TIMING", "-o", out] + [os.path.join(B.CSRC, s) for s in B.HIP_SOURCES]
print(" ".join(cmd))
subprocess.check_call(cmd)
print("built", out)
