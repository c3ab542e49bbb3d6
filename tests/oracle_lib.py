"""ctypes loader for the CPU oracle (oracle/libs2d_oracle.so).  Test infrastructure only."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")

SPLAT_DTYPE = np.dtype([("pos", "<f4", 2), ("sx", "<f4"), ("sy", "<f4"), ("rot", "<f4"),
                        ("color", "<f4", 3), ("opacity", "<f4")])
ADAM_DTYPE = np.dtype([("mv", "<f4", (9, 2))])  # pos[2], sx, sy, rot, color[3], opacity; each {m, v}
assert SPLAT_DTYPE.itemsize == 36 and ADAM_DTYPE.itemsize == 72


class Counters(C.Structure):
    _fields_ = [("visited", C.c_uint64), ("active", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    so = os.path.join(ORACLE_DIR, "libs2d_oracle.so")
    src = os.path.join(ORACLE_DIR, "s2d_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    vp, i, f, d = C.c_void_p, C.c_int, C.c_float, C.c_double
    L.s2do_init.argtypes = [vp, vp, i, i, i]
    L.s2do_forward_rows.argtypes = [vp, i, i, i, i, i, vp, vp]
    L.s2do_backward_rows.argtypes = [vp, i, i, i, i, i, vp, vp, vp, vp, vp]
    L.s2do_backward_rows_stats.argtypes = [vp, i, i, i, i, i, vp, vp, vp, vp, vp, vp]
    L.s2do_adam_step.argtypes = [vp, vp, vp, i, i, i, vp, vp, i, f]
    L.s2do_adam_step.restype = i
    L.s2do_sqerr_rows.argtypes = [vp, vp, i, i, i, i]
    L.s2do_sqerr_rows.restype = d
    L.s2do_mse.argtypes = [vp, vp, i, i]
    L.s2do_mse.restype = d
    L.s2do_step.argtypes = [vp, vp, i, i, i, vp, vp, vp, vp, vp, vp, i, vp]
    L.s2do_step.restype = i
    L.s2do_step_mt.argtypes = [vp, vp, i, i, i, vp, vp, vp, vp, vp, vp, i, vp, i]
    L.s2do_step_mt.restype = i
    L.s2do_cosf.argtypes = [f]; L.s2do_cosf.restype = f
    L.s2do_sinf.argtypes = [f]; L.s2do_sinf.restype = f
    L.s2do_exp_approx.argtypes = [f]; L.s2do_exp_approx.restype = f
    L.s2do_pcg3d.argtypes = [vp]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def load_s2di(path):
    """Read a .s2di fixture -> (H, W, 3) uint8."""
    with open(path, "rb") as fh:
        magic = fh.read(4)
        assert magic == b"S2DI", magic
        w, h, c = struct.unpack("<III", fh.read(12))
        data = np.frombuffer(fh.read(w * h * c), dtype=np.uint8).reshape(h, w, c)
    return data


def target_rgba32f(rgb8):
    """Image2DRGBA8_to_Image2DRGBA32 (main.cpp:258): byte / 255.0f, alpha = 1."""
    h, w, _ = rgb8.shape
    out = np.empty((h, w, 4), dtype=np.float32)
    out[..., :3] = rgb8.astype(np.float32) / np.float32(255.0)
    out[..., 3] = 1.0
    return out


def synthetic_target(W, H):
    """ref(x,y) = (x/W, 1 - x/W, y/H, 1)  (main.cpp:261-267 generator + blue), fp32 ops."""
    x = np.arange(W, dtype=np.float32) / np.float32(W)
    y = np.arange(H, dtype=np.float32) / np.float32(H)
    out = np.empty((H, W, 4), dtype=np.float32)
    out[..., 0] = x[None, :]
    out[..., 1] = np.float32(1.0) - x[None, :]
    out[..., 2] = y[:, None]
    out[..., 3] = 1.0
    return out


class OracleTrainer:
    """Holds the locals of the reference's main() that the three passes own (main.cpp:272-278, 310-314)."""

    def __init__(self, target, n, optimize_opacity=False):
        self.L = lib()
        self.H, self.W = target.shape[:2]
        self.n = n
        self.ref = np.ascontiguousarray(target, dtype=np.float32)
        self.splats = np.zeros(n, dtype=SPLAT_DTYPE)
        self.adams = np.zeros(n, dtype=ADAM_DTYPE)
        self.dsplats = np.zeros(n, dtype=SPLAT_DTYPE)
        self.image0 = np.zeros((self.H, self.W, 4), dtype=np.float32)
        self.image1 = np.zeros((self.H, self.W, 4), dtype=np.float32)
        self.beta1t = np.ones(1, dtype=np.float32)
        self.beta2t = np.ones(1, dtype=np.float32)
        self.iterations = 0
        self.optimize_opacity = bool(optimize_opacity)
        self.init()

    def init(self):
        self.L.s2do_init(_p(self.splats), _p(self.adams), self.n, self.W, self.H)
        self.beta1t[0] = 1.0
        self.beta2t[0] = 1.0
        self.iterations = 0

    def forward(self, y0=0, y1=None, counters=None):
        y1 = self.H if y1 is None else y1
        self.L.s2do_forward_rows(_p(self.splats), self.n, self.W, self.H, y0, y1, _p(self.image0),
                                 C.byref(counters) if counters is not None else None)
        return self.image0

    def backward(self, y0=0, y1=None, counters=None, zero=True):
        y1 = self.H if y1 is None else y1
        if zero:
            self.dsplats[:] = 0
        self.L.s2do_backward_rows(_p(self.splats), self.n, self.W, self.H, y0, y1, _p(self.image0),
                                  _p(self.ref), _p(self.image1), _p(self.dsplats),
                                  C.byref(counters) if counters is not None else None)
        return self.dsplats

    def backward_stats(self):
        """Backward pass + (dsum, dabs): the fp32 contributions summed in double, and the sum of their magnitudes."""
        self.dsplats[:] = 0
        dsum = np.zeros((self.n, 9), dtype=np.float64)
        dabs = np.zeros((self.n, 9), dtype=np.float64)
        self.L.s2do_backward_rows_stats(_p(self.splats), self.n, self.W, self.H, 0, self.H, _p(self.image0),
                                        _p(self.ref), _p(self.image1), _p(self.dsplats), _p(dsum), _p(dabs))
        return self.dsplats, dsum, dabs

    def adam(self, lr=0.05):
        return self.L.s2do_adam_step(_p(self.splats), _p(self.adams), _p(self.dsplats), self.n, self.W, self.H,
                                     _p(self.beta1t), _p(self.beta2t), int(self.optimize_opacity), lr)

    def mse(self):
        return self.L.s2do_mse(_p(self.image0), _p(self.ref), self.W, self.H)

    def step(self, threads=1):
        mse = C.c_double(0.0)
        args = [_p(self.splats), _p(self.adams), self.n, self.W, self.H, _p(self.ref), _p(self.image0),
                _p(self.image1), _p(self.dsplats), _p(self.beta1t), _p(self.beta2t),
                int(self.optimize_opacity), C.byref(mse)]
        if threads > 1:
            st = self.L.s2do_step_mt(*args, threads)
        else:
            st = self.L.s2do_step(*args)
        self.iterations += 1
        return st, mse.value
