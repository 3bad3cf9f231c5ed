"""Generates the golden fixtures under tests/golden/ (run in the build container only; needs
/root/reference and PIL).  Fixtures are DATA: decoded pixels of the reference's own test images
(test_picture/left.png, right.png -- README.md:43-44 names them as the SAD.cpp inputs) and the
expected outputs of the CPU oracle on them, pinned by the SHA-256 values recorded in SURVEY.md
section 4 (T2).  The reference ships no expected output of its own ("parity unpinned").

  tsukuba_left.ppm / tsukuba_right.ppm   P6, RGB order as stored in the PNG (imread would give BGR)
  tsukuba_disp_w7_d30.pgm                expected getDisparity output, winSize=7 DSR=30 (SAD.cpp:9,11)
  tsukuba_loc_w7_d30.pgm                 raw loc before the *16 encode
  rds_seed1_320x240_w9_d64_{u8,bin}_{disp,loc}.pgm   synthetic stereogram vectors
  golden.json                            sha256 of every vector + sample values
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def write_pnm(path, arr):
    with open(path, "wb") as f:
        if arr.ndim == 2:
            f.write(b"P5\n%d %d\n255\n" % (arr.shape[1], arr.shape[0]))
        else:
            f.write(b"P6\n%d %d\n255\n" % (arr.shape[1], arr.shape[0]))
        f.write(np.ascontiguousarray(arr).tobytes())


def main():
    from PIL import Image
    ref = "/root/reference/test_picture"
    meta = {}
    rgbL = np.array(Image.open(os.path.join(ref, "left.png")).convert("RGB"))
    rgbR = np.array(Image.open(os.path.join(ref, "right.png")).convert("RGB"))
    write_pnm(os.path.join(HERE, "tsukuba_left.ppm"), rgbL)
    write_pnm(os.path.join(HERE, "tsukuba_right.ppm"), rgbR)
    bgrL = np.ascontiguousarray(rgbL[:, :, ::-1])     # imread() default IMREAD_COLOR -> BGR (SAD.cpp:65-66)
    bgrR = np.ascontiguousarray(rgbR[:, :, ::-1])
    meta["tsukuba_blue_left_sha256"] = sha(bgrL[:, :, 0])
    meta["tsukuba_blue_right_sha256"] = sha(bgrR[:, :, 0])
    disp = oracle.literal(bgrL, bgrR, 7, 30, 0)
    loc = oracle.literal(bgrL, bgrR, 7, 30, oracle.RAW_LOC)
    assert np.array_equal(disp, oracle.box(bgrL, bgrR, 7, 30, 0))
    assert np.array_equal(disp, oracle.numpy_restatement(bgrL, bgrR, 7, 30))
    write_pnm(os.path.join(HERE, "tsukuba_disp_w7_d30.pgm"), disp)
    write_pnm(os.path.join(HERE, "tsukuba_loc_w7_d30.pgm"), loc)
    meta["tsukuba_disp_sha256"] = sha(disp)
    meta["tsukuba_loc_sha256"] = sha(loc)
    meta["tsukuba_loc_hist"] = np.bincount(loc[:281, :377].ravel(), minlength=30).tolist()
    meta["tsukuba_loc_row0"] = loc[0, :40].tolist()
    for binary, tag in ((0, "u8"), (1, "bin")):
        Lr, Rr = oracle.rds(1, 320, 240, 4, 20, binary)
        d = oracle.literal_mt(Lr, Rr, 9, 64, 0)
        l = oracle.literal_mt(Lr, Rr, 9, 64, oracle.RAW_LOC)
        assert np.array_equal(d, oracle.box(Lr, Rr, 9, 64, 0))
        write_pnm(os.path.join(HERE, "rds_seed1_320x240_w9_d64_%s_disp.pgm" % tag), d)
        write_pnm(os.path.join(HERE, "rds_seed1_320x240_w9_d64_%s_loc.pgm" % tag), l)
        meta["rds1_%s" % tag] = {"L": sha(Lr), "R": sha(Rr), "disp": sha(d), "loc": sha(l),
                                 "L0": Lr[0, :8].tolist(), "R0": Rr[0, :8].tolist(), "loc0": l[0, :8].tolist()}
    meta["splitmix64_seed1"] = ["%#x" % oracle.splitmix64(1, i) for i in (1, 2, 3)]
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(meta, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
