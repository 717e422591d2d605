#!/usr/bin/env python3
"""Generate the Grad-CAM known-answer fixture (runs in the BUILD container only: it reads /root/reference).

The reference ships the six pictures its own grad_cam.exe wrote (cpu/output/0.png ... 5.png, 224x224 RGB) for six named
images of datasets/images/ with the shipped checkpoint (cpu/src/grad_cam.cpp:31-44).  They are the only reference-held
vectors that pin AlexNet::grad_cam (alexnet.cpp:95-142) and conv_layer_3's 64x13x13 activations.

This script stores, as DATA:
  * gradcam_kat_images_u8.npz  -- the six inputs after cv::imread + cv::resize(224x224) (grad_cam.cpp:62-68), HWC BGR uint8
    (same JPEG decode / INTER_LINEAR restatement as make_readme_kat.py), key "images", plus "names";
  * gradcam_kat_expected_0..5.png -- byte copies of the reference's output pictures.
The picture pipeline behind `network.grad_cam("conv_layer_3")` (grad_cam.cpp:73-91) is restated in tests/gradcam_picture.py;
tests/test_oracle_golden.py runs oracle -> picture and compares with the PNGs, tests/test_host_mirror.py does the same through
the HIP path.  What this script measured when it was written is kept in gradcam_kat_report.json.
"""
import json
import os
import shutil
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
from make_readme_kat import cv_resize_linear_u8  # noqa: E402

REF = "/root/reference"
NAMES = ["dog", "bird_2", "panda", "dog_3", "panda_2", "bird"]  # grad_cam.cpp:37-44, in output order 0..5
CKPT = f"{REF}/cpu/checkpoints/AlexNet_aug_1e-3/iter_395000_train_0.918_valid_0.913.model"


def main():
    import gradcam_picture as G
    from oracle import pyoracle as O

    imgs = []
    for n in NAMES:
        rgb = np.asarray(Image.open(f"{REF}/datasets/images/{n}.jpg").convert("RGB"))
        imgs.append(cv_resize_linear_u8(rgb[:, :, ::-1], 224, 224))
    imgs = np.stack(imgs)
    np.savez_compressed(os.path.join(HERE, "gradcam_kat_images_u8.npz"), images=imgs, names=np.array(NAMES))
    report = {"source": "cpu/output/0..5.png written by cpu/src/grad_cam.cpp:62-91 (checkpoint AlexNet_aug_1e-3/iter_395000)",
              "names": NAMES, "per_image": []}
    for k, n in enumerate(NAMES):
        dst = os.path.join(HERE, f"gradcam_kat_expected_{k}.png")
        shutil.copyfile(f"{REF}/cpu/output/{k}.png", dst)
        os.chmod(dst, 0o644)
        exp = G.load_expected_bgr(dst)
        x = G.to_input(imgs[k : k + 1])
        net = O.Net(1, 3)
        net.load_checkpoint(CKPT)
        probs = O.softmax(net.forward(x))
        _, cam8 = O.grad_cam(net.conv_out(2))
        row = {"image": n, "argmax": int(probs.argmax()), "prob": float(probs.max())}
        for tag, first_channel_max in (("max_over_first_channel", True), ("max_over_all_channels", False)):
            d = np.abs(G.picture(cam8, imgs[k], first_channel_max).astype(np.int32) - exp.astype(np.int32))
            row[tag] = {"exact": float((d == 0).mean()), "within_1": float((d <= 1).mean()), "within_2": float((d <= 2).mean()),
                        "max": int(d.max())}
        report["per_image"].append(row)
        print(row)
    with open(os.path.join(HERE, "gradcam_kat_report.json"), "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__":
    sys.exit(main())
