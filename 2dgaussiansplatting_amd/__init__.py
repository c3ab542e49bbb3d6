"""MI355X-native 2D Gaussian splatting trainer: Python host-side binding of the C ABI.

The product is `lib/libsplat2d_hip.so` (hand-written HIP for gfx950 behind include/splat2d.h).
This module only binds it with ctypes for tests and bench.py; `Trainer` holds the same state the
reference keeps in main()'s locals (/root/reference/main.cpp:272-278, :310-317) and exposes the
three passes under the reference's names.  There is no CPU path here: importing works anywhere,
but every compute call needs the built library and a gfx950 device and raises otherwise.

The package name starts with a digit, so import it with
    importlib.import_module("2dgaussiansplatting_amd")
"""
import ctypes as C
import os
import warnings

import numpy as np

from . import _build

__all__ = ["Trainer", "MultiTrainer", "S2DError", "load_library", "hip_runtimes_mapped", "SPLAT_DTYPE", "ADAM_DTYPE",
           "STATUS_NAMES"]

# == struct Splat (main.cpp:85-93), 36 bytes; == struct SplatAdam (main.cpp:158-166), 72 bytes
SPLAT_DTYPE = np.dtype([("pos", "<f4", 2), ("sx", "<f4"), ("sy", "<f4"), ("rot", "<f4"),
                        ("color", "<f4", 3), ("opacity", "<f4")])
ADAM_DTYPE = np.dtype([("mv", "<f4", (9, 2))])

S2D_STEP_OPTIMIZE_OPACITY = 0x1
S2D_CFG_COUNT_PAIRS = 0x1
S2D_CFG_FP16_IMAGES = 0x2
S2D_CFG_DETERMINISTIC = 0x4
S2D_CFG_EXACT_EXP = 0x8
S2D_CFG_ADAM_FP32 = 0x10
S2D_CFG_GENERIC_BINNING = 0x20
S2D_BWD_SKIP_OPACITY_GRAD = 0x1
S2D_FB_SKIP_IMAGE = 0x2
STATUS_NAMES = {0: "S2D_OK", 1: "S2D_E_INVALID", 2: "S2D_E_HIP", 3: "S2D_E_NONFINITE", 4: "S2D_E_NOMEM",
                5: "S2D_E_STATE"}


class S2DError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (STATUS_NAMES.get(code, code), msg))
        self.code = code


class _Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("width", C.c_int32), ("height", C.c_int32),
                ("n_splats", C.c_int32), ("device", C.c_int32), ("row_begin", C.c_int32),
                ("row_end", C.c_int32), ("training_rate", C.c_float), ("flags", C.c_uint32),
                ("rebin_interval", C.c_int32), ("rebin_margin", C.c_float), ("stream", C.c_void_p)]


ROWS_GRADS, ROWS_SPLATS, ROWS_ADAM = 0, 1, 2  # S2D_ROWS_* (row arrays of the slab-ownership calls)


class _Stats(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved", C.c_uint32),
                ("pairs_binned", C.c_uint64), ("pairs_capacity", C.c_uint64), ("rebins", C.c_uint64),
                ("fwd_visited", C.c_uint64), ("fwd_active", C.c_uint64), ("bwd_visited", C.c_uint64),
                ("bwd_active", C.c_uint64), ("fwd_staged", C.c_uint64), ("bwd_staged", C.c_uint64),
                ("fwd_wave_execs", C.c_uint64), ("bwd_wave_execs", C.c_uint64), ("bwd_lane_hist", C.c_uint64 * 65),
                ("iterations", C.c_int32), ("first_nonfinite_iteration", C.c_int32),
                ("fwd_staged_hit", C.c_uint64), ("fwd_rows_hit", C.c_uint64), ("bwd_quadrant_execs", C.c_uint64)]


ABI_VERSION = 2  # S2D_ABI_VERSION of include/splat2d.h as this binding was written (checked against the library at load)

# every symbol include/splat2d.h and include/splat2d_test.h declare
ABI_SYMBOLS = [
    "s2d_abi_version", "s2d_create", "s2d_destroy", "s2d_set_target", "s2d_set_target_synthetic",
    "s2d_init_splats", "s2d_set_splats", "s2d_get_splats", "s2d_set_adam", "s2d_get_adam", "s2d_forward",
    "s2d_get_image", "s2d_get_image_rows", "s2d_backward", "s2d_forward_backward", "s2d_get_grads", "s2d_adam_step", "s2d_step", "s2d_get_mse",
    "s2d_bind_grads_device", "s2d_grads_device_ptr", "s2d_stream", "s2d_get_sqerr_trace", "s2d_synchronize", "s2d_get_stats",
    "s2d_get_rebuild_count",
    "s2d_last_error", "s2d_test_sincos", "s2d_test_sort_pairs", "s2d_test_exclusive_scan",
    "s2d_debug_get_tile_lists",
    "s2d_halo_masks", "s2d_halo_commit", "s2d_rows_gather", "s2d_rows_scatter", "s2d_grads_combine",
    "s2d_multi_create", "s2d_multi_destroy", "s2d_multi_last_error", "s2d_multi_device_count", "s2d_multi_set_target",
    "s2d_multi_set_target_synthetic", "s2d_multi_init_splats", "s2d_multi_set_splats", "s2d_multi_get_splats",
    "s2d_multi_set_adam", "s2d_multi_get_adam", "s2d_multi_step", "s2d_multi_get_image", "s2d_multi_exchange_info", "s2d_multi_forward",
    "s2d_multi_device_info", "s2d_multi_set_stall_timeout", "s2d_test_multi_stall",
]

_lib = None


def hip_runtimes_mapped():
    """Paths of the HIP runtime images (libamdhip64) mapped into this process."""
    try:
        with open("/proc/self/maps") as f:
            return sorted({line.split()[-1] for line in f if "libamdhip64" in line})
    except OSError:
        return []


def _elf_dynamic_strings(path, tag):
    """Strings of the given dynamic-section tag (1 = DT_NEEDED, 14 = DT_SONAME) of a 64-bit little-endian ELF file; [] when
    the file cannot be read that way."""
    import struct
    try:
        with open(path, "rb") as f:
            head = f.read(64)
            if head[:6] != b"\x7fELF\x02\x01":
                return []
            shoff, = struct.unpack_from("<Q", head, 0x28)
            shentsize, shnum = struct.unpack_from("<HH", head, 0x3A)
            f.seek(shoff)
            sh = f.read(shentsize * shnum)
            out = []
            for k in range(shnum):
                _, typ, _, _, off, size, link, _, _, entsize = struct.unpack_from("<IIQQQQIIQQ", sh, k * shentsize)
                if typ != 6:  # SHT_DYNAMIC
                    continue
                _, _, _, _, stroff, strsize, _, _, _, _ = struct.unpack_from("<IIQQQQIIQQ", sh, link * shentsize)
                f.seek(off)
                dyn = f.read(size)
                f.seek(stroff)
                strtab = f.read(strsize)
                for j in range(0, size, 16):
                    t, v = struct.unpack_from("<qQ", dyn, j)
                    if t == tag:
                        out.append(strtab[v:strtab.index(b"\0", v)].decode())
            return out
    except (OSError, struct.error, ValueError):
        return []


def _bind_process_hip_runtime():
    """One HIP runtime per process.

    libsplat2d_hip.so needs `libamdhip64.so.7` (its DT_NEEDED entry, the runtime's SONAME).  PyTorch's ROCm wheel bundles
    its own copy of that runtime as torch/lib/libamdhip64.so -- same SONAME -- and its libraries ask for it as
    `libamdhip64.so`:
      * torch first: the bundled copy is loaded; our DT_NEEDED matches it by SONAME, nothing else is mapped;
      * this library first: the dynamic linker resolves `libamdhip64.so.7` to /opt/rocm/lib, and a later `import torch`
        asks for `libamdhip64.so`, which matches neither the name nor the SONAME of the loaded image, finds the bundled
        file and maps a SECOND runtime: two HSA clients in one process, streams and device pointers of one unknown to
        the other ("torch finds no HIP GPUs").
    So, when no runtime is mapped yet and a PyTorch with a bundled runtime is installed, map THAT file first (by path,
    without importing torch): ours then binds to it by SONAME and a later `import torch` finds the same file.
    S2D_HIP_RUNTIME=system keeps the dynamic linker's choice (a process that never imports torch);
    S2D_HIP_RUNTIME=<path> maps that file.
    """
    if hip_runtimes_mapped():
        return  # a runtime is already in the process: the DT_NEEDED entry binds to it by SONAME
    choice = os.environ.get("S2D_HIP_RUNTIME", "auto")
    if choice == "system":
        return
    path = None
    if choice not in ("auto", "torch"):
        path = choice
    else:
        import importlib.util
        try:
            spec = importlib.util.find_spec("torch")
        except (ImportError, ValueError):
            spec = None
        if spec is not None and spec.origin:
            cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(cand):
                path = cand
    if path:
        needed = _elf_dynamic_strings(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libsplat2d_hip.so"), 1)  # DT_NEEDED
        soname = _elf_dynamic_strings(path, 14)  # DT_SONAME
        want = [x for x in needed if x.startswith("libamdhip64")]
        if choice in ("auto", "torch") and want and soname and soname[0] not in want:
            # the bundled runtime would not satisfy our DT_NEEDED entry: mapping it would put TWO runtimes into a process
            # that never imports torch and works with the system's alone
            warnings.warn("2dgaussiansplatting_amd: %s has SONAME %s, libsplat2d_hip.so needs %s: leaving the choice of the HIP "
                          "runtime to the dynamic linker (import torch first if this process uses both)" % (path, soname[0], want[0]))
            return
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError as e:
            warnings.warn("2dgaussiansplatting_amd: could not pre-map the HIP runtime %s (%s): leaving the choice to the dynamic "
                          "linker" % (path, e))


def load_library(path=None):
    """dlopen the HIP library.  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("S2D_LIBRARY") or _build.LIB_PATH  # S2D_LIBRARY: A/B builds of the same ABI
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the trainer has no CPU fallback)" % path)
    _bind_process_hip_runtime()
    L = C.CDLL(path)
    mapped = hip_runtimes_mapped()
    if len(mapped) > 1:
        raise RuntimeError("two HIP runtimes are mapped into this process (%s): device pointers and streams of one mean "
                           "nothing to the other.  Import this package before anything else loads a differently named "
                           "libamdhip64, or point S2D_HIP_RUNTIME at the runtime the process uses" % ", ".join(mapped))
    vp, i32, u32, i64 = C.c_void_p, C.c_int32, C.c_uint32, C.c_int64
    ab = path != _build.LIB_PATH  # an A/B build of an older ABI may lack newer entry points (tools/gpu_ab.py)

    def sig(name, argtypes=None, restype=None):
        if ab and not hasattr(L, name):
            return
        f = getattr(L, name)
        if argtypes is not None:
            f.argtypes = argtypes
        if restype is not None:
            f.restype = restype

    sig("s2d_abi_version", restype=C.c_int)
    if not ab and L.s2d_abi_version() != ABI_VERSION:
        raise RuntimeError("%s was built for ABI version %d, this binding for %d: rebuild it (python -c 'import "
                           "__graft_entry__ as g; g.build()')" % (path, L.s2d_abi_version(), ABI_VERSION))
    sig("s2d_create", [C.POINTER(_Config), C.POINTER(vp)])
    sig("s2d_destroy", [vp])
    if hasattr(L, "s2d_destroy"):
        L.s2d_destroy.restype = None
    sig("s2d_set_target", [vp, vp])
    sig("s2d_set_target_synthetic", [vp])
    sig("s2d_init_splats", [vp])
    sig("s2d_set_splats", [vp, vp])
    sig("s2d_get_splats", [vp, vp])
    sig("s2d_set_adam", [vp, vp, C.c_float, C.c_float, i32])
    sig("s2d_get_adam", [vp, vp, vp, vp, vp])
    sig("s2d_forward", [vp])
    sig("s2d_get_image", [vp, vp])
    sig("s2d_get_image_rows", [vp, vp])
    sig("s2d_backward", [vp, u32])
    sig("s2d_forward_backward", [vp, u32])
    sig("s2d_get_grads", [vp, vp])
    sig("s2d_adam_step", [vp, u32])
    sig("s2d_step", [vp, i32, u32, vp])
    sig("s2d_get_mse", [vp, vp])
    sig("s2d_bind_grads_device", [vp, vp])
    sig("s2d_grads_device_ptr", [vp], restype=vp)
    sig("s2d_stream", [vp], restype=vp)
    sig("s2d_get_sqerr_trace", [vp, i32, i32, vp])
    sig("s2d_synchronize", [vp])
    sig("s2d_get_stats", [vp, C.POINTER(_Stats)])
    sig("s2d_get_rebuild_count", [vp, vp])
    sig("s2d_last_error", [vp], restype=C.c_char_p)
    sig("s2d_test_sincos", [i32, vp, i32, vp, vp])
    sig("s2d_test_sort_pairs", [i32, vp, vp, i64, i32])
    sig("s2d_test_exclusive_scan", [i32, vp, i64, vp])
    sig("s2d_debug_get_tile_lists", [vp, vp, vp, vp, i64, vp, i64])
    sig("s2d_halo_masks", [vp, i32, vp, C.c_float, vp])
    sig("s2d_halo_commit", [vp, vp, i32, i32])
    sig("s2d_rows_gather", [vp, i32, vp, i32, vp])
    sig("s2d_rows_scatter", [vp, i32, vp, i32, vp])
    sig("s2d_grads_combine", [vp, vp, i32, vp, i32, vp])
    sig("s2d_multi_create", [C.POINTER(_Config), vp, i32, u32, C.POINTER(vp)])
    sig("s2d_multi_destroy", [vp])
    if hasattr(L, "s2d_multi_destroy"):
        L.s2d_multi_destroy.restype = None
    sig("s2d_multi_last_error", [vp], restype=C.c_char_p)
    sig("s2d_multi_device_count", [vp])
    sig("s2d_multi_set_target", [vp, vp])
    sig("s2d_multi_set_target_synthetic", [vp])
    sig("s2d_multi_init_splats", [vp])
    sig("s2d_multi_set_splats", [vp, vp])
    sig("s2d_multi_get_splats", [vp, vp])
    sig("s2d_multi_set_adam", [vp, vp, C.c_float, C.c_float, i32])
    sig("s2d_multi_get_adam", [vp, vp, vp, vp, vp])
    sig("s2d_multi_step", [vp, i32, u32, vp])
    sig("s2d_multi_get_image", [vp, vp])
    sig("s2d_multi_exchange_info", [vp, vp])
    sig("s2d_multi_forward", [vp])
    sig("s2d_multi_device_info", [vp, i32, vp, vp, vp, C.c_char_p, i32, C.c_char_p, i32])
    sig("s2d_multi_set_stall_timeout", [vp, i32])
    sig("s2d_test_multi_stall", [vp, i32, i32, i32])
    if path == _build.LIB_PATH:
        _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Trainer:
    """The reference's training state and its three passes, on one MI355X (or one row slab of the image).

    Mirrors main()'s locals: splats / splatAdams / beta1t / beta2t / iterations (main.cpp:272-278),
    imageRef (:254), image0 (:310), optimizeOpacity (:317).
    """

    def __init__(self, width, height, n_splats, device=0, row_begin=0, row_end=0, training_rate=0.0,
                 rebin_interval=0, rebin_margin=0.0, count_pairs=False, fp16_images=False, deterministic=False, exact_exp=False,
                 adam_fp32=False, generic_binning=False, stream=None, chunk_pairs=None):
        self.L = load_library()
        self.W, self.H, self.n = int(width), int(height), int(n_splats)
        cfg = _Config()
        cfg.struct_size = C.sizeof(_Config)
        cfg.width, cfg.height, cfg.n_splats, cfg.device = self.W, self.H, self.n, int(device)
        cfg.row_begin, cfg.row_end = int(row_begin), int(row_end)
        self.row_begin, self.row_end = (0, self.H) if (int(row_begin), int(row_end)) == (0, 0) else (int(row_begin), int(row_end))
        cfg.training_rate = float(training_rate)
        cfg.flags = ((S2D_CFG_COUNT_PAIRS if count_pairs else 0) | (S2D_CFG_FP16_IMAGES if fp16_images else 0) |
                     (S2D_CFG_DETERMINISTIC if deterministic else 0) | (S2D_CFG_EXACT_EXP if exact_exp else 0) |
                     (S2D_CFG_ADAM_FP32 if adam_fp32 else 0) | (S2D_CFG_GENERIC_BINNING if generic_binning else 0))
        cfg.rebin_interval = int(rebin_interval)
        cfg.rebin_margin = float(rebin_margin)
        cfg.stream = stream
        self.stream = stream  # HIP stream handle the context works on; None: a stream the library owns
        h = C.c_void_p()
        # chunk_pairs (tests): the (tile, splat) pair budget beyond which a scene is rendered by index ranges of the splats;
        # the library reads S2D_CHUNK_PAIRS when the context is created (default 2^30)
        old_env = os.environ.get("S2D_CHUNK_PAIRS")
        if chunk_pairs is not None:
            os.environ["S2D_CHUNK_PAIRS"] = str(int(chunk_pairs))
        try:
            rc = self.L.s2d_create(C.byref(cfg), C.byref(h))
        finally:
            if chunk_pairs is not None:
                if old_env is None:
                    del os.environ["S2D_CHUNK_PAIRS"]
                else:
                    os.environ["S2D_CHUNK_PAIRS"] = old_env
        self._h = h
        if rc != 0:
            msg = self.L.s2d_last_error(h).decode() if h else "s2d_create rejected the configuration"
            if h:
                self.L.s2d_destroy(h)
            self._h = None
            raise S2DError(rc, msg)
        self.optimize_opacity = False  # main.cpp:317
        self.lean_backward = False     # backward() may skip the opacity gradient while optimize_opacity is off

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self.L.s2d_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, rc):
        if rc != 0:
            raise S2DError(rc, self.L.s2d_last_error(self._h).decode())

    def _flags(self):
        return S2D_STEP_OPTIMIZE_OPACITY if self.optimize_opacity else 0

    # -- state
    def set_target(self, rgba32f):
        a = np.ascontiguousarray(rgba32f, dtype=np.float32)
        assert a.shape == (self.H, self.W, 4), a.shape
        self._ck(self.L.s2d_set_target(self._h, _p(a)))

    def set_target_synthetic(self):
        self._ck(self.L.s2d_set_target_synthetic(self._h))

    def init(self):
        """init(), main.cpp:280-305 (also what the Restart button calls, :828-831)."""
        self._ck(self.L.s2d_init_splats(self._h))

    def set_splats(self, splats):
        a = np.ascontiguousarray(splats, dtype=SPLAT_DTYPE)
        assert a.shape == (self.n,)
        self._ck(self.L.s2d_set_splats(self._h, _p(a)))

    def get_splats(self):
        a = np.zeros(self.n, dtype=SPLAT_DTYPE)
        self._ck(self.L.s2d_get_splats(self._h, _p(a)))
        return a

    def set_adam(self, adams, beta1t, beta2t, iterations):
        a = np.ascontiguousarray(adams, dtype=ADAM_DTYPE)
        assert a.shape == (self.n,)
        self._ck(self.L.s2d_set_adam(self._h, _p(a), float(beta1t), float(beta2t), int(iterations)))

    def get_adam(self):
        a = np.zeros(self.n, dtype=ADAM_DTYPE)
        b1, b2, it = C.c_float(), C.c_float(), C.c_int32()
        self._ck(self.L.s2d_get_adam(self._h, _p(a), C.byref(b1), C.byref(b2), C.byref(it)))
        return a, np.float32(b1.value), np.float32(b2.value), it.value

    # -- the three passes
    def forward(self):
        self._ck(self.L.s2d_forward(self._h))

    def get_image(self):
        a = np.zeros((self.H, self.W, 4), dtype=np.float32)
        self._ck(self.L.s2d_get_image(self._h, _p(a)))
        return a

    def get_image_rows(self):
        """The slab's rows of image0 only (row_begin .. row_end)."""
        a = np.zeros((self.row_end - self.row_begin, self.W, 4), dtype=np.float32)
        self._ck(self.L.s2d_get_image_rows(self._h, _p(a)))
        return a

    def backward(self, skip_opacity_grad=None):
        """Backward pass.  skip_opacity_grad=None: skip dSplats.opacity exactly when optimize_opacity is off and
        `lean_backward` was requested (bench / training loops); False: always compute it, as the reference does."""
        if skip_opacity_grad is None:
            skip_opacity_grad = self.lean_backward and not self.optimize_opacity
        self._ck(self.L.s2d_backward(self._h, S2D_BWD_SKIP_OPACITY_GRAD if skip_opacity_grad else 0))

    def forward_backward(self, skip_opacity_grad=None, skip_image=False):
        """forward() + backward() in one launch per tile (same results); skip_opacity_grad as backward();
        skip_image: do not store image0 (S2D_FB_SKIP_IMAGE)."""
        if skip_opacity_grad is None:
            skip_opacity_grad = self.lean_backward and not self.optimize_opacity
        self._ck(self.L.s2d_forward_backward(self._h, (S2D_BWD_SKIP_OPACITY_GRAD if skip_opacity_grad else 0) |
                                             (S2D_FB_SKIP_IMAGE if skip_image else 0)))

    def get_grads(self):
        a = np.zeros(self.n, dtype=SPLAT_DTYPE)
        self._ck(self.L.s2d_get_grads(self._h, _p(a)))
        return a

    def adam_step(self):
        self._ck(self.L.s2d_adam_step(self._h, self._flags()))

    def step(self, iters=1, want_mse=True):
        """iters whole iterations; returns the MSE values the reference would print (main.cpp:807)."""
        out = np.zeros(iters, dtype=np.float64) if want_mse else None
        self._ck(self.L.s2d_step(self._h, int(iters), self._flags(), _p(out) if want_mse else None))
        return out

    def mse(self):
        v = C.c_double()
        self._ck(self.L.s2d_get_mse(self._h, C.byref(v)))
        return v.value

    def sqerr_trace(self, first_iteration, count):
        out = np.zeros(count, dtype=np.float64)
        self._ck(self.L.s2d_get_sqerr_trace(self._h, int(first_iteration), int(count), _p(out)))
        return out

    def synchronize(self):
        self._ck(self.L.s2d_synchronize(self._h))

    # -- multi-GPU plumbing
    def bind_grads(self, device_ptr):
        self._ck(self.L.s2d_bind_grads_device(self._h, C.c_void_p(device_ptr) if device_ptr else None))

    def grads_device_ptr(self):
        return self.L.s2d_grads_device_ptr(self._h)

    # slab ownership (include/splat2d.h, "Slab ownership"): raw DEVICE pointers in, work queued on the context's
    # stream; distributed.HipHaloOps wraps these around torch tensors
    def halo_masks(self, row_bounds, margin_rows, masks_ptr):
        b = (C.c_int32 * len(row_bounds))(*row_bounds)
        self._ck(self.L.s2d_halo_masks(self._h, len(row_bounds) - 1, b, C.c_float(margin_rows), C.c_void_p(masks_ptr)))

    def halo_commit(self, masks_ptr, rank, added=1):
        self._ck(self.L.s2d_halo_commit(self._h, C.c_void_p(masks_ptr), rank, added))

    def rows_gather(self, what, ids_ptr, count, out_ptr):
        self._ck(self.L.s2d_rows_gather(self._h, what, C.c_void_p(ids_ptr), count, C.c_void_p(out_ptr)))

    def rows_scatter(self, what, ids_ptr, count, in_ptr):
        self._ck(self.L.s2d_rows_scatter(self._h, what, C.c_void_p(ids_ptr), count, C.c_void_p(in_ptr)))

    def grads_combine(self, rows_ptr, n_rows, src_ptr, world, recv_ptr):
        self._ck(self.L.s2d_grads_combine(self._h, C.c_void_p(rows_ptr), n_rows, C.c_void_p(src_ptr), world,
                                          C.c_void_p(recv_ptr)))

    # -- diagnostics
    def stats(self):
        s = _Stats()
        s.struct_size = C.sizeof(_Stats)
        self._ck(self.L.s2d_get_stats(self._h, C.byref(s)))
        out = {k: getattr(s, k) for k, _ in _Stats._fields_ if k not in ("struct_size", "reserved")}
        out["bwd_lane_hist"] = list(s.bwd_lane_hist)
        return out

    def rebuild_count(self):
        """Times the tile lists were rebuilt so far (host-side counter: no device synchronisation)."""
        v = C.c_uint64()
        self._ck(self.L.s2d_get_rebuild_count(self._h, C.byref(v)))
        return v.value

    def tile_lists(self):
        st = self.stats()
        tx, ty = C.c_int32(), C.c_int32()
        self._ck(self.L.s2d_debug_get_tile_lists(self._h, C.byref(tx), C.byref(ty), None, 0, None, 0))
        off = np.zeros(tx.value * ty.value + 1, dtype=np.uint32)
        lst = np.zeros(max(int(st["pairs_binned"]), 1), dtype=np.uint32)
        self._ck(self.L.s2d_debug_get_tile_lists(self._h, C.byref(tx), C.byref(ty), _p(off), len(off),
                                                 _p(lst), len(lst)))
        return tx.value, ty.value, off, lst[:int(st["pairs_binned"])]


class MultiTrainer:
    """Several GPUs behind one handle (s2d_multi_*, csrc/s2d_multi.hip): the Trainer's state and step() on a list of
    devices -- row slabs and, inside the library, either slab ownership with peer-to-peer copies of the shared
    gradient rows (default) or replicated state with an RCCL all-reduce of all gradients (replicated=True).
    share_gpu: every rank on devices[0] (a rehearsal where there are fewer GPUs than ranks)."""

    SCHEMES = {0: "none", 1: "ownership", 2: "replicated"}

    def __init__(self, width, height, n_splats, devices, share_gpu=False, training_rate=0.0, rebin_interval=0,
                 fp16_images=False, deterministic=False, replicated=False):
        self.L = load_library()
        self.W, self.H, self.n = int(width), int(height), int(n_splats)
        cfg = _Config()
        cfg.struct_size = C.sizeof(_Config)
        cfg.width, cfg.height, cfg.n_splats = self.W, self.H, self.n
        cfg.training_rate = float(training_rate)
        cfg.rebin_interval = int(rebin_interval)
        cfg.flags = (S2D_CFG_FP16_IMAGES if fp16_images else 0) | (S2D_CFG_DETERMINISTIC if deterministic else 0)
        devs = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        rc = self.L.s2d_multi_create(C.byref(cfg), devs, len(devices), (1 if share_gpu else 0) | (2 if replicated else 0), C.byref(h))
        self._h = h
        if rc != 0:
            msg = self.L.s2d_multi_last_error(h).decode() if h else "s2d_multi_create rejected the configuration"
            if h:
                self.L.s2d_multi_destroy(h)
            self._h = None
            raise S2DError(rc, msg)
        self.optimize_opacity = False

    def close(self):
        if getattr(self, "_h", None):
            self.L.s2d_multi_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise S2DError(rc, self.L.s2d_multi_last_error(self._h).decode())

    def set_target(self, rgba32f):
        a = np.ascontiguousarray(rgba32f, dtype=np.float32)
        assert a.shape == (self.H, self.W, 4), a.shape
        self._ck(self.L.s2d_multi_set_target(self._h, _p(a)))

    def set_target_synthetic(self):
        self._ck(self.L.s2d_multi_set_target_synthetic(self._h))

    def init(self):
        self._ck(self.L.s2d_multi_init_splats(self._h))

    def set_splats(self, splats):
        a = np.ascontiguousarray(splats, dtype=SPLAT_DTYPE)
        assert a.shape == (self.n,)
        self._ck(self.L.s2d_multi_set_splats(self._h, _p(a)))

    def get_splats(self):
        a = np.zeros(self.n, dtype=SPLAT_DTYPE)
        self._ck(self.L.s2d_multi_get_splats(self._h, _p(a)))
        return a

    def get_adam(self):
        a = np.zeros(self.n, dtype=ADAM_DTYPE)
        b1, b2, it = C.c_float(), C.c_float(), C.c_int32()
        self._ck(self.L.s2d_multi_get_adam(self._h, _p(a), C.byref(b1), C.byref(b2), C.byref(it)))
        return a, np.float32(b1.value), np.float32(b2.value), it.value

    def set_adam(self, adams, beta1t, beta2t, iterations):
        a = np.ascontiguousarray(adams, dtype=ADAM_DTYPE)
        self._ck(self.L.s2d_multi_set_adam(self._h, _p(a), float(beta1t), float(beta2t), int(iterations)))

    def step(self, iters=1, want_mse=True):
        out = np.zeros(iters, dtype=np.float64) if want_mse else None
        flags = S2D_STEP_OPTIMIZE_OPACITY if self.optimize_opacity else 0
        self._ck(self.L.s2d_multi_step(self._h, int(iters), flags, _p(out) if want_mse else None))
        return out

    def forward(self):
        self._ck(self.L.s2d_multi_forward(self._h))

    def get_image(self):
        a = np.zeros((self.H, self.W, 4), dtype=np.float32)
        self._ck(self.L.s2d_multi_get_image(self._h, _p(a)))
        return a

    def device_info(self, rank):
        """Which GPU runs which rows: {rank, device, row_begin, row_end, pci_bus_id, name} of one rank."""
        dev, r0, r1 = C.c_int32(), C.c_int32(), C.c_int32()
        pci, name = C.create_string_buffer(64), C.create_string_buffer(256)
        self._ck(self.L.s2d_multi_device_info(self._h, int(rank), C.byref(dev), C.byref(r0), C.byref(r1), pci, 64, name, 256))
        return {"rank": int(rank), "device": dev.value, "row_begin": r0.value, "row_end": r1.value,
                "pci_bus_id": pci.value.decode(), "name": name.value.decode()}

    def set_stall_timeout(self, milliseconds):
        """Every wait of one rank for another (and for its own stream) gives up after this long: step() then raises
        S2D_E_STATE naming the rank instead of hanging (0 = wait without bound)."""
        self._ck(self.L.s2d_multi_set_stall_timeout(self._h, int(milliseconds)))

    def test_stall(self, rank, iteration, milliseconds):
        """Failure injection (include/splat2d_test.h): rank `rank` stops answering at `iteration`."""
        self._ck(self.L.s2d_test_multi_stall(self._h, int(rank), int(iteration), int(milliseconds)))

    def exchange_info(self):
        """scheme, gradient rows swapped per iteration (all ranks), state rows handed over, splats held (all ranks)."""
        a = np.zeros(4, dtype=np.int64)
        self._ck(self.L.s2d_multi_exchange_info(self._h, _p(a)))
        return {"scheme": self.SCHEMES[int(a[0])], "rows_per_iteration": int(a[1]), "state_handovers": int(a[2]), "held": int(a[3])}
