"""Test-side restatement of the picture pipeline around AlexNet::grad_cam in the reference's driver
(cpu/src/grad_cam.cpp:73-91): everything between the 13x13 8-bit map `network.grad_cam("conv_layer_3")` returns and the PNG that
driver writes.  OpenCV is not in this image, so the three OpenCV calls are restated from their documented 8-bit semantics:

  cv::Mat cam = 255 - network.grad_cam("conv_layer_3");        :75   8-bit, saturating
  cv::resize(cam, cam, {224, 224});                            :77   INTER_LINEAR on CV_8UC1 (11-bit fixed point: the restatement
                                                                     tests/golden/make_readme_kat.py already uses for the inputs)
  cv::applyColorMap(cam, heat_map, cv::COLORMAP_JET);          :80   256-entry BGR table, see jet_lut()
  heat_map = heat_map / 255 + origin / 255;  (CV_32FC3)        :81-83
  maxValue = *std::max_element(heat_map.begin<float>(), heat_map.end<float>());   :84
  heat_map = heat_map / maxValue * 255 -> CV_8UC3              :85-87 saturate_cast = round-half-even, clamped

One reference quirk matters and is reproduced (tests/golden/gradcam_kat_report.json holds the evidence): `begin<float>()` on a
3-channel float Mat walks PIXELS (the iterator steps by the Mat's 12-byte element size) and reads the first float of each, so
maxValue is the maximum of channel 0 (blue) only; the other two channels may exceed it and saturate at 255.  With the maximum taken
over all three channels the six shipped pictures are off by up to 26 grey levels; with the first-channel maximum 99.8-99.99 % of
their pixels are within 2 (78-89 % identical; the rest is JPEG-decoder and table rounding noise, at most 4 levels).
"""
import os
import sys

import numpy as np

_G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
if _G not in sys.path:
    sys.path.insert(0, _G)
from make_readme_kat import cv_resize_linear_u8  # noqa: E402  (a function; its main() is what reads /root/reference)


def jet_lut():
    """cv::COLORMAP_JET as a [256][3] BGR uint8 table: 64-entry ramps of 4 levels per index through x.5 values --
    blue 127.5 + 4 i rising on 0..31, flat to 95, falling to 159; green rising on 32..95, flat to 159, falling to 223; red = blue
    mirrored -- rounded half-to-even like the float -> 8-bit conversion of the table."""
    i = np.arange(256, dtype=np.float64)
    r = np.minimum(1.5 + 4 * (i - 96), 127.5 + 4 * (255 - i))
    g = np.minimum(1.5 + 4 * (i - 32), 1.5 + 4 * (223 - i))
    b = np.minimum(127.5 + 4 * i, 1.5 + 4 * (159 - i))
    return np.clip(np.rint(np.clip(np.stack([b, g, r], 1), 0, 255)), 0, 255).astype(np.uint8)


def to_input(images_u8):
    """Tensor3D::read_from_opencv_mat (data_format.cpp:13-23) on [N][H][W][3] BGR bytes: plane c <- img[3 i + c] * 1.f / 255"""
    x = images_u8.astype(np.float32) * np.float32(1.0) / np.float32(255)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))


def picture(cam_u8, origin_u8, first_channel_max=True):
    """cam_u8: the [13][13] uint8 map AlexNet::grad_cam returns (opecv_mat(1)); origin_u8: the resized [224][224][3] BGR input.
    Returns the [224][224][3] BGR uint8 picture grad_cam.cpp:87 hands to cv::imwrite."""
    cam = (255 - cam_u8.astype(np.int32)).astype(np.uint8)
    H, W = origin_u8.shape[:2]
    big = cv_resize_linear_u8(cam[:, :, None], W, H)[:, :, 0]
    heat = jet_lut()[big].astype(np.float32)
    blend = heat * np.float32(1.0 / 255) + origin_u8.astype(np.float32) * np.float32(1.0 / 255)
    mx = blend[:, :, 0].max() if first_channel_max else blend.max()
    out = blend / np.float32(mx) * np.float32(255)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def load_expected_bgr(path):
    from PIL import Image

    return np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1]


def load_kat(golden_dir):
    z = np.load(os.path.join(golden_dir, "gradcam_kat_images_u8.npz"))
    names = [str(n) for n in z["names"]]
    exp = [load_expected_bgr(os.path.join(golden_dir, f"gradcam_kat_expected_{k}.png")) for k in range(len(names))]
    return z["images"], names, exp


def compare(pic, exp):
    """(fraction of bytes within 2 grey levels, fraction identical, worst difference)"""
    d = np.abs(pic.astype(np.int32) - exp.astype(np.int32))
    return float((d <= 2).mean()), float((d == 0).mean()), int(d.max())
