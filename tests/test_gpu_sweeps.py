"""Seeded random-shape sweeps against the CPU oracle (round 6): the scripts under tests/sweeps/ (test infrastructure: they import the oracle) run as part of the -m gpu suite with fixed
seeds and small counts.  They exist because hand-picked shapes left a hole: the first run of tests/sweeps/fuzz_conv.py found wgrad_sp_any wrong on
outputs whose last 21-column block does not end on a 16-byte unit (profiles/NOTEBOOK.md, round 6)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(*args):
    r = subprocess.run([sys.executable, *args], cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = "\n".join((r.stdout + r.stderr).strip().splitlines()[-25:])
    assert r.returncode == 0, tail
    return r.stdout


@pytest.mark.parametrize("what", ["conv", "pool", "bn", "lin", "misc"])
def test_random_shapes_of_every_layer_type_vs_oracle(what):
    """tests/sweeps/fuzz_layers.py: convolutions with wide planes / 1x1, 5x5, 7x7 filters / many units per workgroup (conv2d.cpp:69-199), MaxPool2D
    forward + mask + backward bit-exact (pool2d.cpp:40-107), BatchNorm2D training passes and the fused BatchNorm -> ReLU -> MaxPool forms
    (batchnorm2d.cpp:24-158), LinearLayer and the fused loss head (linear.cpp:33-90, func.cpp:16-73); misc: ReLU, SGD, softmax + cross
    entropy, Dropout, Grad-CAM, the uint8 batch stager at random sizes"""
    out = _run("tests/sweeps/fuzz_layers.py", what, "24", "11")
    assert "FUZZ OK" in out


@pytest.mark.parametrize("mode", ["small", "big"])
def test_random_networks_fused_equals_unfused_and_match_oracle(mode):
    """tests/sweeps/fuzz_nets.py: random layer lists through architectures::Sequential::train_step with every fusion on / the pool block off /
    all fusions off -- losses, parameters, gradients, the input delta and every get_output() bit for bit over three full steps, a step with a
    SMALLER batch and a full one behind it; first-step logits / loss and the small step's loss / gradients against oracle.pyoracle.SeqNet
    (alexnet.cpp:35-65 for an arbitrary list); the reference's own loop (cnn.cpp:79-90) from a device batch and from host tensors; inference
    under no_grad; small: also through a one-rank RCCL communicator with the exchange forced on"""
    args = ["tests/sweeps/fuzz_nets.py", "10", "21"] + (["big"] if mode == "big" else ["dp"])
    out = _run(*args)
    assert "FUZZ NETS OK" in out


def test_random_shapes_on_the_per_width_kernel_instances_vs_oracle():
    """tests/sweeps/fuzz_conv.py widths: the plane widths of the per-width instances (7 / 14 / 28 / 56 / 112 / 110 / 55 / 27 / 13, 224-wide
    3-channel first layers) with random heights, batches and channel counts, all three passes against the oracle (conv2d.cpp:69-199)"""
    out = _run("tests/sweeps/fuzz_conv.py", "30", "31", "widths")
    assert "FUZZ OK" in out


def test_anything_the_descriptor_admits_vs_oracle_or_refused():
    """tests/sweeps/fuzz_conv.py wild: odd filters 1..7, strides 1..4, padding 0..4 (also beyond k/2), 1..40 channels, planes of 1..40 pixels: all
    three passes against the oracle (conv2d.cpp:69-199 on the zero-padded input), geometries without an output pixel refused with a status"""
    out = _run("tests/sweeps/fuzz_conv.py", "40", "41", "wild")
    assert "FUZZ OK" in out
