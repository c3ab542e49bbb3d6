#!/usr/bin/env python3
"""Decode the reference's two input JPEGs once (PIL / libjpeg-turbo) into raw fixtures.

The reference loads `squirrel_cls_mini.jpg` through prlib/stb_image
(/root/reference/main.cpp:253-259).  prlib is not vendored, so the decode is done
here, once, in the build container; oracle, CPU baseline and GPU path all read
the SAME committed raw pixels, so decoder differences cannot enter parity.

Output format ".s2di" (little endian):  b"S2DI", u32 width, u32 height, u32 channels(=3),
then height*width*3 bytes RGB8 row-major.  sha256(RGB bytes)[:16] must equal the
values recorded in SURVEY.md §8(c): mini 84fc7f3b4eba07ae, full ba7ed1b221888ba2.

BASELINE.json configs[1] names the big image as "512x512", but the file is 535x426: the 512x512 fixture is a documented
derivative -- centre crop to 426x426 (columns 54..479), then Lanczos resampling to 512x512 (PIL Image.resize,
Image.LANCZOS), done once here and committed raw.  No reference output exists for it; it is a
parity-test case against the oracle like every other size.

Run only in the build container (needs /root/reference); never on the GPU box.
"""
import hashlib, struct, sys
import numpy as np
from PIL import Image

SRC = "/root/reference/bin/"
OUT = "tests/golden/"
EXPECT = {"squirrel_cls_mini.jpg": "84fc7f3b4eba07ae", "squirrel_cls.jpg": "ba7ed1b221888ba2"}

def main():
    for name, want in EXPECT.items():
        rgb = np.asarray(Image.open(SRC + name).convert("RGB"), dtype=np.uint8)
        got = hashlib.sha256(rgb.tobytes()).hexdigest()[:16]
        if got != want:
            sys.exit(f"{name}: sha {got} != survey {want}")
        h, w, _ = rgb.shape
        out = OUT + name.replace(".jpg", f"_{w}x{h}.s2di")
        with open(out, "wb") as f:
            f.write(b"S2DI" + struct.pack("<III", w, h, 3) + rgb.tobytes())
        print(out, w, h, got)
        if name == "squirrel_cls.jpg":
            x0 = (w - h) // 2
            sq = Image.fromarray(rgb[:, x0:x0 + h]).resize((512, 512), Image.LANCZOS)
            rgb512 = np.asarray(sq, dtype=np.uint8)
            out = OUT + "squirrel_cls_512x512.s2di"
            with open(out, "wb") as f:
                f.write(b"S2DI" + struct.pack("<III", 512, 512, 3) + rgb512.tobytes())
            print(out, 512, 512, hashlib.sha256(rgb512.tobytes()).hexdigest()[:16], "(centre crop %d..%d, Lanczos)" % (x0, x0 + h - 1))

if __name__ == "__main__":
    main()
