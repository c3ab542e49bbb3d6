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


# ---------------------------------------------------------------------------------------------
# JPEG (2dgaussiansplatting_amd/host/jpeg_decode.h): restated IJG islow IDCT / fancy upsampling / YCC->RGB
# ---------------------------------------------------------------------------------------------
def _jpeg_cases():
    for sub in (0, 1, 2):                       # 4:4:4, 4:2:2, 4:2:0
        for prog in (False, True):
            yield dict(quality=85, subsampling=sub, progressive=prog)
    yield dict(quality=30, subsampling=2, progressive=True, optimize=True)
    yield dict(quality=98, subsampling=0)
    yield dict(quality=75, subsampling=2, restart_marker_blocks=5)
    yield dict(quality=75, subsampling=1, progressive=True, restart_marker_rows=1)


@pytest.mark.parametrize("kw", list(_jpeg_cases()), ids=lambda k: "-".join("%s%s" % (a[:4], b) for a, b in k.items()))
@pytest.mark.parametrize("crop", [(268, 213), (267, 211), (17, 9), (8, 8), (1, 1)])
def test_jpeg_decoder_matches_pil_bit_for_bit(exe, tmp_path, kw, crop):
    rgb = O.load_s2di(MINI)[:crop[1], :crop[0]]
    src = tmp_path / "in.jpg"
    try:
        Image.fromarray(rgb).save(src, format="JPEG", **kw)
    except TypeError:
        pytest.skip("this Pillow cannot write restart markers")
    conv(exe, src, tmp_path / "out.s2di")
    got = O.load_s2di(str(tmp_path / "out.s2di"))
    want = np.asarray(Image.open(src).convert("RGB"))
    assert got.shape == want.shape
    assert np.array_equal(got, want), "max abs diff %d" % np.abs(got.astype(int) - want.astype(int)).max()


def test_jpeg_greyscale_and_corrupt(exe, tmp_path):
    rgb = O.load_s2di(MINI)
    Image.fromarray(rgb).convert("L").save(tmp_path / "g.jpg", quality=80)
    conv(exe, tmp_path / "g.jpg", tmp_path / "g.s2di")
    assert np.array_equal(O.load_s2di(str(tmp_path / "g.s2di")), np.asarray(Image.open(tmp_path / "g.jpg").convert("RGB")))
    data = open(tmp_path / "g.jpg", "rb").read()
    open(tmp_path / "trunc.jpg", "wb").write(data[:200])
    conv(exe, tmp_path / "trunc.jpg", tmp_path / "t.ppm", ok=False)


@pytest.mark.skipif(not os.path.exists("/root/reference/bin/squirrel_cls_mini.jpg"), reason="reference tree not present (GPU box)")
def test_decoding_the_reference_jpegs_reproduces_the_committed_fixtures(exe, tmp_path):
    """The host tool reads the reference's own inputs (main.cpp:257) and lands on exactly the pixels of
    tests/golden/*.s2di (decoded once with PIL), i.e. on the sha256 prefixes recorded in SURVEY.md section 8c."""
    for jpg, fixture in (("squirrel_cls_mini.jpg", "squirrel_cls_mini_268x213.s2di"), ("squirrel_cls.jpg", "squirrel_cls_535x426.s2di")):
        conv(exe, "/root/reference/bin/" + jpg, tmp_path / "o.s2di")
        assert open(tmp_path / "o.s2di", "rb").read() == open(os.path.join(O.GOLDEN, fixture), "rb").read()


def test_decoders_survive_mutated_files_under_sanitizers(tmp_path):
    """Fuzz of the host program's readers (host/image_io.h: .s2di, PPM, PNG; host/jpeg_decode.h: baseline and progressive JPEG):
    tests/hostfuzz/decode_fuzz.cpp, built with -fsanitize=address,undefined, decodes thousands of mutated files (truncations, bit
    flips, marker soup, holes, damaged headers) per format in one process.  Every file must be rejected or decoded into a
    consistent image; no out-of-bounds access, no undefined behaviour.  (Found this way: signed overflow in the IDCT on corrupt
    coefficients -- 64-bit intermediates now.)"""
    import numpy as np
    from PIL import Image
    cxx = "/opt/rocm/lib/llvm/bin/clang++" if os.path.exists("/opt/rocm/lib/llvm/bin/clang++") else "g++"
    exe = str(tmp_path / "decode_fuzz")
    subprocess.check_call([cxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-I", os.path.join(O.ROOT, "include"), "-o", exe, os.path.join(O.ROOT, "tests", "hostfuzz", "decode_fuzz.cpp"), "-lz"])
    rng = np.random.default_rng(0)
    img = Image.fromarray((rng.random((37, 53, 3)) * 255).astype(np.uint8))
    files = {"a.jpg": ("jpg", dict(quality=85)), "p.jpg": ("jpg", dict(quality=70, progressive=True, subsampling=2)),
             "g.jpg": ("jpg", dict(quality=60)), "a.png": ("png", {}), "a.ppm": ("ppm", {})}
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    for name, (kind, kw) in files.items():
        path = str(tmp_path / name)
        (img.convert("L") if name == "g.jpg" else img).save(path, **kw)
        p = subprocess.run([exe, path, kind, "2500", "7"], capture_output=True, text=True, env=env, timeout=600)
        assert p.returncode == 0, (name, p.stderr[-1500:])
        n_dec, n_rej = [int(v) for v in __import__("re").findall(r"(\d+) decoded, (\d+) rejected", p.stdout)[0]]
        assert n_dec + n_rej == 2500 and n_rej > 100, p.stdout      # the mutations really were malformed often enough
