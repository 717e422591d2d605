#!/usr/bin/env python3
"""Generate the README known-answer fixture (runs in the BUILD container only: it reads /root/reference).

The reference publishes exactly one known answer for the forward path (README.md:92, screenshot
imgs/image-20230208213627060.png): inference.exe (cpu/src/inference.cpp:35-70) loads
cpu/checkpoints/AlexNet_aug_1e-3/iter_395000_train_0.918_valid_0.913.model and prints
    dog.jpg   -> dog   0.850634
    panda.jpg -> panda 0.999978
    bird.jpg  -> bird  0.999998

This script reproduces the *inputs* of that run as data:
  * readme_kat_images_u8.npy  -- the three images after cv::imread (BGR, 8-bit) and
    cv::resize(224x224, INTER_LINEAR) (inference.cpp:55,61), stored HWC uint8.  JPEG decode is PIL's
    libjpeg-turbo (ISLOW, fancy upsampling: the same defaults cv::imread uses); the resize is a
    restatement of OpenCV's 8-bit INTER_LINEAR (11-bit fixed-point coefficients, separable, the
    ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2)>>2 vertical pass).
  * readme_kat_checkpoint.model -- byte copy of the reference's checkpoint DATA file (445 068 B).
  * readme_kat_expected.json -- the three probabilities printed in the README screenshot.
tests/test_oracle_golden.py then feeds these through oracle/ (and, on the GPU, through the HIP path).
"""
import json
import os
import shutil
import sys

import numpy as np
from PIL import Image

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
COEF_BITS = 11
ONE = 1 << COEF_BITS


def _sat_short(v):
    return np.clip(np.rint(v), -32768, 32767).astype(np.int32)


def _axis_tables(src, dst):
    """OpenCV resize() linear tables: source index, 2 fixed-point coefficients, and the first dst index
    (xmax) from which the right neighbour would fall outside the source."""
    scale = src / dst
    ofs = np.zeros(dst, np.int64)
    coef = np.zeros((dst, 2), np.int32)
    xmax = dst
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - s)
        if s < 0:
            f, s = np.float32(0), 0
        if s + 2 >= src + 1:  # sx + ksize2 >= ssize.width with ksize2 = 1
            xmax = min(xmax, d)
            if s >= src - 1:
                f, s = np.float32(0), src - 1
        ofs[d] = s
        coef[d, 0] = _sat_short((np.float32(1) - f) * np.float32(ONE))
        coef[d, 1] = _sat_short(f * np.float32(ONE))
    return ofs, coef, xmax


def cv_resize_linear_u8(img, dst_w, dst_h):
    """img: HxWxC uint8 -> dst_h x dst_w x C uint8, OpenCV INTER_LINEAR semantics for CV_8U."""
    h, w, _ = img.shape
    xo, xa, xmax = _axis_tables(w, dst_w)
    yo, yb, _ = _axis_tables(h, dst_h)
    src = img.astype(np.int32)
    # horizontal pass on every source row -> int32 (scaled by ONE)
    right = np.minimum(xo + 1, w - 1)
    rows = src[:, xo, :] * xa[None, :, 0, None] + src[:, right, :] * xa[None, :, 1, None]
    if xmax < dst_w:
        rows[:, xmax:, :] = src[:, xo[xmax:], :] * ONE
    # vertical pass
    y1 = np.minimum(yo + 1, h - 1)
    s0 = rows[yo]
    s1 = rows[y1]
    b0 = yb[:, 0][:, None, None]
    b1 = yb[:, 1][:, None, None]
    out = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def main():
    names = ["dog", "panda", "bird"]
    imgs = []
    for n in names:
        rgb = np.asarray(Image.open(f"{REF}/datasets/images/{n}.jpg").convert("RGB"))
        bgr = rgb[:, :, ::-1]  # cv::imread gives BGR
        imgs.append(cv_resize_linear_u8(bgr, 224, 224))
    np.save(os.path.join(HERE, "readme_kat_images_u8.npy"), np.stack(imgs))
    shutil.copyfile(
        f"{REF}/cpu/checkpoints/AlexNet_aug_1e-3/iter_395000_train_0.918_valid_0.913.model",
        os.path.join(HERE, "readme_kat_checkpoint.model"),
    )
    os.chmod(os.path.join(HERE, "readme_kat_checkpoint.model"), 0o644)
    with open(os.path.join(HERE, "readme_kat_expected.json"), "w") as f:
        json.dump(
            {
                "source": "README.md:92 (imgs/image-20230208213627060.png), cpu/src/inference.cpp:28-70",
                "categories": ["dog", "panda", "bird"],
                "images": names,
                "argmax": [0, 1, 2],
                "prob": [0.850634, 0.999978, 0.999998],
            },
            f,
            indent=1,
        )
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    sys.exit(main())
