"""CPU tests of the boundary: the C-ABI library builds for gfx950, loads, exports every symbol the header
declares, agrees with the ctypes binding on struct layouts, and fails loudly (no fallback) without a GPU."""
import ctypes as C
import importlib
import os
import re
import subprocess
import sys
import tempfile

import pytest

import oracle_lib as O

S2D = importlib.import_module("2dgaussiansplatting_amd")
HEADER = os.path.join(O.ROOT, "include", "splat2d.h")
TEST_HEADER = os.path.join(O.ROOT, "include", "splat2d_test.h")  # test / inspection hooks, outside the drop-in boundary


@pytest.fixture(scope="module")
def lib():
    S2D._build.build_hip_library()
    return S2D.load_library()


def declared_symbols(headers=(HEADER, TEST_HEADER)):
    names = set()
    for h in headers:
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(s2d_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_every_declared_symbol_is_exported(lib):
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(S2D.ABI_SYMBOLS) == names  # the binding's own list is the headers' list
    # the product header holds no test hook, the test header nothing else
    assert not [n for n in declared_symbols((HEADER,)) if n.startswith(("s2d_test_", "s2d_debug_"))]
    assert all(n.startswith(("s2d_test_", "s2d_debug_")) for n in declared_symbols((TEST_HEADER,)))
    # one version number: the header's macro, the library's answer, the binding's constant
    macro = int(re.search(r"#define\s+S2D_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert lib.s2d_abi_version() == macro == S2D.ABI_VERSION


def test_struct_layouts_match_header():
    code = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "splat2d.h"
    int main(void) {
        printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(s2d_splat), sizeof(s2d_splat_adam), sizeof(s2d_config),
               sizeof(s2d_stats), offsetof(s2d_config, stream), offsetof(s2d_config, training_rate),
               offsetof(s2d_stats, iterations), offsetof(s2d_stats, fwd_wave_execs));
        
        return 0;
    }'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write(code)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(O.ROOT, "include"), src, "-o", exe])
        got = [int(v) for v in subprocess.check_output([exe]).split()]
    assert got[0] == 36 == S2D.SPLAT_DTYPE.itemsize      # struct Splat, main.cpp:85-93
    assert got[1] == 72 == S2D.ADAM_DTYPE.itemsize       # struct SplatAdam, main.cpp:158-166
    assert got[2] == C.sizeof(S2D._Config)
    assert got[3] == C.sizeof(S2D._Stats)
    assert got[4] == S2D._Config.stream.offset and got[5] == S2D._Config.training_rate.offset
    assert got[6] == S2D._Stats.iterations.offset and got[7] == S2D._Stats.fwd_wave_execs.offset
    assert O.SPLAT_DTYPE == S2D.SPLAT_DTYPE and O.ADAM_DTYPE.itemsize == S2D.ADAM_DTYPE.itemsize


def test_create_rejects_bad_configs(lib):
    h = C.c_void_p()
    cfg = S2D._Config()
    assert lib.s2d_create(C.byref(cfg), C.byref(h)) == 1      # struct_size missing
    cfg.struct_size = C.sizeof(S2D._Config)
    cfg.width, cfg.height, cfg.n_splats = 0, 10, 1
    assert lib.s2d_create(C.byref(cfg), C.byref(h)) == 1
    cfg.width, cfg.height = 64, 64
    cfg.row_begin, cfg.row_end = 8, 64                          # slab must start on a tile row
    assert lib.s2d_create(C.byref(cfg), C.byref(h)) == 1
    cfg.row_begin, cfg.row_end = 32, 16
    assert lib.s2d_create(C.byref(cfg), C.byref(h)) == 1
    assert lib.s2d_create(None, C.byref(h)) == 1
    assert lib.s2d_forward(None) == 1 and lib.s2d_step(None, 1, 0, None) == 1


def test_slab_ownership_calls_reject_bad_arguments(lib):
    """s2d_halo_* / s2d_rows_* / s2d_grads_combine check their arguments before any device work."""
    buf = (C.c_uint32 * 4)()
    rows = (C.c_int32 * 3)(0, 16, 32)
    assert lib.s2d_halo_masks(None, 2, rows, C.c_float(8.0), buf) == 1
    assert lib.s2d_halo_commit(None, buf, 0, 1) == 1
    assert lib.s2d_rows_gather(None, S2D.ROWS_SPLATS, buf, 1, buf) == 1
    assert lib.s2d_rows_scatter(None, S2D.ROWS_ADAM, buf, 1, buf) == 1
    assert lib.s2d_grads_combine(None, buf, 1, buf, 2, buf) == 1


def _gpu_present():
    return os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK)


@pytest.mark.skipif(_gpu_present(), reason="a GPU is present: the no-device error cannot be provoked")
def test_no_device_is_a_loud_error_not_a_fallback(lib):
    with pytest.raises(S2D.S2DError) as ei:
        S2D.Trainer(64, 64, 10)
    assert ei.value.code == 2 and "no CPU fallback" in str(ei.value)
    train = S2D._build.build_host_program()
    r = subprocess.run([train, "--synthetic", "32x32", "--splats", "4", "--iters", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr
    # the multi-device handle: the same loud error, in both schemes, and bad arguments are rejected before any device call
    for replicated in (False, True):
        with pytest.raises(S2D.S2DError) as ei:
            S2D.MultiTrainer(64, 64, 10, [0, 1], replicated=replicated)
        assert ei.value.code == 2 and "no CPU fallback" in str(ei.value)
    r = subprocess.run([train, "--synthetic", "32x32", "--splats", "4", "--iters", "1", "--gpus", "2"], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


def test_multi_handle_rejects_bad_arguments(lib):
    import ctypes as C
    cfg = S2D._Config()
    cfg.struct_size = C.sizeof(S2D._Config)
    cfg.width, cfg.height, cfg.n_splats = 64, 64, 10
    h = C.c_void_p()
    devs = (C.c_int32 * 2)(0, 1)
    assert lib.s2d_multi_create(C.byref(cfg), devs, 0, 0, C.byref(h)) == 1        # no devices
    assert lib.s2d_multi_create(C.byref(cfg), devs, 33, 0, C.byref(h)) == 1       # more ranks than hold-set bits
    assert lib.s2d_multi_create(C.byref(cfg), None, 2, 0, C.byref(h)) == 1
    cfg.row_begin, cfg.row_end = 0, 16                                             # the handle cuts the slabs itself
    assert lib.s2d_multi_create(C.byref(cfg), devs, 2, 0, C.byref(h)) == 1
    assert lib.s2d_multi_step(None, 1, 0, None) == 1 and lib.s2d_multi_device_count(None) == 0
    assert lib.s2d_multi_last_error(None) == b"null handle"
    lib.s2d_multi_destroy(None)


def test_product_sources_do_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under the package or include/ may include, link or load it."""
    pkg = os.path.join(O.ROOT, "2dgaussiansplatting_amd")
    for base in (pkg, os.path.join(O.ROOT, "include")):
        for dp, _, files in os.walk(base):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    assert "s2d_oracle" not in txt and "s2do_" not in txt and "oracle_lib" not in txt, os.path.join(dp, f)
    out = subprocess.check_output(["ldd", S2D._build.LIB_PATH], text=True)
    assert "oracle" not in out and "hostcheck" not in out


def test_hip_sources_have_no_cuda_or_dual_path():
    csrc = os.path.join(O.ROOT, "2dgaussiansplatting_amd", "csrc")
    for f in os.listdir(csrc):
        txt = open(os.path.join(csrc, f)).read()
        for bad in ("__HIP_PLATFORM_AMD__", "cuda_runtime", "__CUDACC__", "hipify", "triton"):
            assert bad not in txt, (f, bad)


@pytest.mark.parametrize("order", ["package_first", "torch_first"])
def test_one_hip_runtime_whatever_the_import_order(order):
    """torch's wheel bundles its own libamdhip64 (same SONAME, requested under another name), and
    a process in which libsplat2d_hip.so came up first used to end with TWO runtimes mapped.  load_library() maps the
    process's runtime first, so in a fresh process either import order leaves exactly one libamdhip64 image -- and
    librccl, once torch has brought it, is the only one of its kind too."""
    pytest.importorskip("torch")
    load = ("import importlib; S = importlib.import_module('2dgaussiansplatting_amd'); L = S.load_library(); "
            "assert L.s2d_abi_version() >= 1")
    body = (load + "; import torch") if order == "package_first" else ("import torch; " + load)
    code = ("import sys; sys.path.insert(0, %r); %s; "
            "maps = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}); "
            "rccl = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'librccl' in l}); "
            "print(len(maps), len(rccl), maps, S.hip_runtimes_mapped() == maps)" % (O.ROOT, body))
    env = {k: v for k, v in os.environ.items() if k != "S2D_HIP_RUNTIME"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    n_hip, n_rccl = r.stdout.split()[:2]
    assert n_hip == "1" and n_rccl in ("0", "1"), r.stdout
    assert r.stdout.strip().endswith("True")
