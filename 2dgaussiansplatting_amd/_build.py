"""Build helpers: compile the HIP library (gfx950) and the C++ host program in-tree.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the
resulting .so travels to the GPU box with the repo snapshot (it is git-ignored, not
gpurun-ignored).
"""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libsplat2d_hip.so")
HOST_DIR = os.path.join(PKG_DIR, "host")
TRAIN_BIN = os.path.join(LIB_DIR, "splat2d_train")

HIP_SOURCES = ["s2d_api.hip", "s2d_scan_sort.hip", "s2d_binning.hip", "s2d_tilelists.hip", "s2d_raster.hip", "s2d_optim.hip",
               "s2d_halo.hip", "s2d_multi.hip"]
HIP_HEADERS = ["s2d_device.h", "s2d_math.h"]

# -ffp-contract=off: the kernels keep the reference's evaluation order (no FMA contraction) wherever a
# discrete decision or the framebuffer depends on it; fp32 divide/sqrt stay correctly rounded (hipcc default).
# -amdgpu-atomic-optimizer-strategy=None: LLVM rewrites an atomic whose address is wave-uniform into a reduction over the active
# lanes (a scalar s_ff1 / v_readlane loop, v_mbcnt, a second exec switch) followed by one lane's atomic.  The backward blend adds
# the ninth gradient from ONE lane per executed (wave, entry): the rewrite put ~10 vector and ~12 scalar instructions around a
# single ds_add_f32, two thirds of what that gradient cost (profiles/r04/ab_ninth_gradient_atomic.txt).  The counters that do
# add from all lanes to one address are diagnostics (S2D_CFG_COUNT_PAIRS).
HIPCC_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
               "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
               "-Wall", "-Wno-unused-function", "-pthread", "-ldl"]


def kernel_source_digest():
    """sha256[:16] over the sources of the raster kernels (s2d_raster.hip and the two headers it is made of) with
    comments and whitespace removed: what ties a recorded profile of the dominant kernel (profiles/traffic.json) to the
    code it measured (bench.py drops the figure when the digests differ)."""
    import hashlib
    import re
    h = hashlib.sha256()
    for name in ("s2d_device.h", "s2d_math.h", "s2d_raster.hip"):
        if os.path.exists(os.path.join(CSRC, name)):
            text = open(os.path.join(CSRC, name)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            text = re.sub(r"//[^\n]*", "", text)
            h.update(name.encode())
            h.update(re.sub(r"\s+", "", text).encode())
    return h.hexdigest()[:16]


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X library cannot be built (there is no CPU fallback)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip_library(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HIP_HEADERS] + [os.path.join(ROOT, "include", "splat2d.h"),
                                                                   os.path.join(ROOT, "include", "splat2d_test.h")]
    if force or _stale(LIB_PATH, deps):
        os.makedirs(LIB_DIR, exist_ok=True)
        cmd = [hipcc()] + HIPCC_FLAGS + ["-o", LIB_PATH] + srcs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB_PATH


def build_host_program(force=False, verbose=False):
    """The headless C++ host loop (mirrors main.cpp's frame loop) on top of the C ABI."""
    src = os.path.join(HOST_DIR, "splat2d_train.cpp")
    if not os.path.exists(src):
        return None
    deps = [src, os.path.join(HOST_DIR, "image_io.h"), os.path.join(HOST_DIR, "overlay.h"), os.path.join(HOST_DIR, "jpeg_decode.h"),
            os.path.join(ROOT, "include", "splat2d.h"), LIB_PATH]
    if force or _stale(TRAIN_BIN, deps):
        rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
        # -ffp-contract=off: overlay.h restates main.cpp:441-477 operation by operation (tests compare its vertices bitwise)
        cmd = ["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-o", TRAIN_BIN, src,
               "-L", LIB_DIR, "-lsplat2d_hip", "-lz", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(rocm, "lib")]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return TRAIN_BIN


def build_experiment(name, defines=(), csrc=None, verbose=False):
    """An A/B or diagnostic build of the same ABI for tools/gpu_ab*.py (`S2D_LIBRARY=<path>` selects it): written to
    <repo>/build/, NEVER into the package's lib/ -- lib/ holds the product (libsplat2d_hip.so, splat2d_train) and nothing else.
    csrc: a directory with a patched copy of the sources (default: the tree's)."""
    out_dir = os.path.join(ROOT, "build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libsplat2d_hip_%s.so" % name)
    src_dir = csrc or CSRC
    cmd = [hipcc()] + HIPCC_FLAGS + ["-I", os.path.join(ROOT, "include")] + ["-D" + d for d in defines] + ["-o", out] + \
          [os.path.join(src_dir, s) for s in HIP_SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_all(force=False, verbose=False):
    lib = build_hip_library(force=force, verbose=verbose)
    build_host_program(force=force, verbose=verbose)
    return lib
