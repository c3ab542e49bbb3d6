"""CPU tests of the host program's image I/O: .s2di / PPM / PNG conversions through
`splat2d_train --convert`, which touches no GPU.  PNG decoding is checked against PIL on the filter types,
colour types and palettes PIL can produce."""
import importlib
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

import oracle_lib as O

S2D = importlib.import_module("2dgaussiansplatting_amd")
MINI = os.path.join(O.GOLDEN, "squirrel_cls_mini_268x213.s2di")


@pytest.fixture(scope="module")
def exe():
    S2D._build.build_hip_library()
    return S2D._build.build_host_program()


def conv(exe, a, b, ok=True):
    r = subprocess.run([exe, "--convert", str(a), str(b)], capture_output=True, text=True)
    assert (r.returncode == 0) == ok, r.stderr
    return r


def test_roundtrip_s2di_png_ppm(exe, tmp_path):
    conv(exe, MINI, tmp_path / "a.png")
    conv(exe, tmp_path / "a.png", tmp_path / "a.ppm")
    conv(exe, tmp_path / "a.ppm", tmp_path / "a.s2di")
    assert open(tmp_path / "a.s2di", "rb").read() == open(MINI, "rb").read()
    ref = O.load_s2di(MINI)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "a.png").convert("RGB")), ref)   # our PNG is a valid PNG
    assert np.array_equal(np.asarray(Image.open(tmp_path / "a.ppm")), ref)


@pytest.mark.parametrize("mode,kw", [("RGB", dict(optimize=True)), ("RGB", dict(compress_level=0)), ("RGBA", {}), ("L", {}),
                                      ("LA", {}), ("P", {})])
def test_png_reader_matches_pil(exe, tmp_path, mode, kw):
    rgb = O.load_s2di(MINI)
    im = Image.fromarray(rgb)
    im = im.convert("P", palette=Image.ADAPTIVE) if mode == "P" else im.convert(mode)
    src = tmp_path / ("in_%s.png" % mode)
    im.save(src, **kw)
    conv(exe, src, tmp_path / "out.s2di")
    assert np.array_equal(O.load_s2di(str(tmp_path / "out.s2di")), np.asarray(Image.open(src).convert("RGB")))


def test_bad_files_are_rejected(exe, tmp_path):
    bad = tmp_path / "bad.png"
    data = bytearray(open(MINI, "rb").read())
    open(bad, "wb").write(b"\x89PNG\r\n\x1a\n" + bytes(data[:100]))
    conv(exe, bad, tmp_path / "x.ppm", ok=False)
    conv(exe, MINI, tmp_path / "ok.png")
    corrupt = bytearray(open(tmp_path / "ok.png", "rb").read())
    corrupt[200] ^= 0xFF                                  # CRC must catch a flipped byte
    open(tmp_path / "corrupt.png", "wb").write(corrupt)
    conv(exe, tmp_path / "corrupt.png", tmp_path / "y.ppm", ok=False)
    conv(exe, tmp_path / "missing.ppm", tmp_path / "z.ppm", ok=False)
