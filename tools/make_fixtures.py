#!/usr/bin/env python3
"""Generate tests/golden/* from the reference's own test data (run in the
authoring container only; /root/reference does not exist on the GPU box).

Every output is DATA held by the reference's tests, re-encoded compactly:
  case{0,1,2}_templates.npz  <- test/case*/{circle,test}_templ.yaml  (template fixtures)
  case1_train_bgr.npz        <- test/case1/train.png  (BGR pixels as cv::imread would return)
  case1_test_bgr.npz         <- test/case1/test.png
  case2_train_bgr.npz, case2_test_bgr.npz
  case0_circle_bgr.npz       <- test/case0/templ/circle.png (the training image of test.cpp:scale_test)
  case0_info_scales.npy      <- the `scale:` values of test/case0/circle_info.yaml (float32)
  similarity_lut.sha256      <- digest of the 256 SIMILARITY_LUT entries (line2Dup.cpp:635)
No reference source text is copied.
"""
import hashlib
import os
import re
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from shape_based_matching_amd.templates import read_class_yaml  # noqa: E402

REF = os.environ.get("SBM_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")


def imread_bgr(path):
    """cv::imread(path) default flags: 8-bit, 3-channel BGR, alpha dropped."""
    return np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1].copy()


def parse_similarity_lut(src_path):
    txt = open(src_path).read()
    m = re.search(r"SIMILARITY_LUT\[256\]\s*=\s*\{([^}]*)\}", txt)
    lut3 = int(re.search(r"LUT3\s*=\s*(\d+)\s*;", txt).group(1))
    vals = [lut3 if t.strip() == "LUT3" else int(t) for t in m.group(1).split(",")]
    assert len(vals) == 256
    return np.asarray(vals, np.uint8)


def main():
    os.makedirs(OUT, exist_ok=True)
    for case, name in ((0, "circle"), (1, "test"), (2, "test")):
        ts = read_class_yaml(f"{REF}/test/case{case}/{name}_templ.yaml")
        ts.save_npz(f"{OUT}/case{case}_templates.npz")
        print(f"case{case}: {ts.n_templates} templates, {len(ts.features)} features")
    for case, which in ((1, "train"), (1, "test"), (2, "train"), (2, "test")):
        a = imread_bgr(f"{REF}/test/case{case}/{which}.png")
        np.savez_compressed(f"{OUT}/case{case}_{which}_bgr.npz", bgr=a)
        print(f"case{case}_{which}: {a.shape}")
    a = imread_bgr(f"{REF}/test/case0/templ/circle.png")
    np.savez_compressed(f"{OUT}/case0_circle_bgr.npz", bgr=a)
    scales = [float(x) for x in re.findall(r"scale: ([0-9.e+-]+)", open(f"{REF}/test/case0/circle_info.yaml").read())]
    np.save(f"{OUT}/case0_info_scales.npy", np.asarray(scales, np.float32))
    print(f"case0 circle: {a.shape}, {len(scales)} infos")
    lut = parse_similarity_lut(f"{REF}/line2Dup.cpp")
    with open(f"{OUT}/similarity_lut.sha256", "w") as fh:
        fh.write(hashlib.sha256(lut.tobytes()).hexdigest() + "\n")


if __name__ == "__main__":
    main()
