"""Generates the filter fixtures under tests/golden/ (run in the build container only; needs /root/reference and PIL).
Fixtures are DATA: the reference's own pictures test_picture/cat.bmp and dog.bmp (README.md:43 names them as the inputs
of the filter programs), copied byte for byte, and the outputs of the CPU filter oracle (oracle/filter_oracle.c) on them
with the programs' own parameters:

  gaus_filter.cpp   (size 35, sigma 3, :10-11)            -> gaus_cat_low.png, gaus_cat_high.png
  hybrid_image.cpp  (size 31, sigma 10, += 12, alpha 0.6)  -> hybrid_dog_low.png, hybrid_cat_high.png, hybrid_dog_cat.png
  filters_golden.json   sha256 of the decoded BGR pixels and of every output (BGR order, as cv::Mat holds them)

The pictures are decoded with PIL here, independently of the product's own BMP reader (imageio.hpp), which the tests
check against these hashes.  The expected outputs are oracle-derived, not OpenCV-derived: OpenCV is absent, parity with
its filter2D stays unpinned (see oracle/filter_oracle.c)."""
import hashlib
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import filters as F  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    from PIL import Image
    ref = "/root/reference/test_picture"
    meta = {}
    bgr = {}
    for n in ("cat", "dog"):
        shutil.copyfile(os.path.join(ref, n + ".bmp"), os.path.join(HERE, n + ".bmp"))
        rgb = np.asarray(Image.open(os.path.join(ref, n + ".bmp")).convert("RGB"))
        bgr[n] = np.ascontiguousarray(rgb[:, :, ::-1])          # what cv::imread returns
        meta[n + "_bgr_sha256"] = sha(bgr[n])
        meta[n + "_shape"] = list(bgr[n].shape)

    def save(name, img):
        Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(os.path.join(HERE, name + ".png"), optimize=True)
        meta[name + "_sha256"] = sha(img)

    k = F.gaussian_kernel(35, 3)                                # gaus_filter.cpp:57-62 on cat.bmp
    low = F.filter2d(bgr["cat"], k)
    save("gaus_cat_low", low)
    save("gaus_cat_high", F.highpass(bgr["cat"], low, 0))
    k = F.gaussian_kernel(31, 10)                               # hybrid_image.cpp:120-138: dog low + cat high
    low = F.filter2d(bgr["dog"], k)
    tmp = F.filter2d(bgr["cat"], k)
    high = F.highpass(bgr["cat"], tmp, 12)
    save("hybrid_dog_low", low)
    save("hybrid_cat_high", high)
    save("hybrid_dog_cat", F.blend(low, high, 0.6))
    with open(os.path.join(HERE, "filters_golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
