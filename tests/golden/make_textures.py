#!/usr/bin/env python3
"""Fixtures of the reference's one fully textured asset (web/plane.obj + models/plane_material.mtl + its two base-colour
images), made once in the build container where /root/reference exists.  DATA only: the three MTL files byte for byte, and
the two map_Kd images of plane_material.mtl decoded with PIL and box-filtered to 512 x 512 (the originals are 2048^2 /
4096^2: 12 / 48 MB raw), stored as gzip'd binary PPM (P6).  Tests read these files, never /root/reference.

  python tests/golden/make_textures.py            # rewrites tests/golden/{*.mtl, *_512.ppm.gz}
"""
import gzip
import os
import shutil

import numpy as np
from PIL import Image

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
SIZE = 512

for name in ("plane_material.mtl", "dragon_material.mtl", "2Dplane_material.mtl"):
    shutil.copyfile(os.path.join(REF, "models", name), os.path.join(HERE, name))

for name in ("Material_85_baseColor.jpg", "Material_86_baseColor.png"):
    with Image.open(os.path.join(REF, "models", name)) as im:
        src = im.size
        rgb = np.asarray(im.convert("RGB").resize((SIZE, SIZE), Image.BOX), dtype=np.uint8)
    out = os.path.join(HERE, os.path.splitext(name)[0] + f"_{SIZE}.ppm.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(b"P6\n%d %d\n255\n" % (SIZE, SIZE) + rgb.tobytes())
    print(f"{name}: {src[0]}x{src[1]} -> {SIZE}x{SIZE}, {os.path.getsize(out)} bytes, mean RGB {rgb.reshape(-1, 3).mean(0).round(2)}")
