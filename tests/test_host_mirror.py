"""The C++17 host mirror (cnn_amd/host): library/ABI checks on CPU, behaviour against the oracle on the GPU."""
import json
import os

import numpy as np
import pytest

from tests.util import REL_TOL, assert_close, normal_scaled, uniform01


def test_host_library_loads_and_exports():
    from cnn_amd import hostapi

    lib = hostapi.load()
    for name in hostapi.SIGNATURES:
        assert hasattr(lib, name)


def test_reference_headers_surface():
    """the host headers keep the reference's public names (architectures.h:34-46,69,96,109,131; data_format.h:11-53)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    arch = open(os.path.join(root, "cnn_amd/host/include/architectures.h")).read()
    for needle in ("class Layer", "virtual std::vector<tensor> forward(const std::vector<tensor>& input) = 0;",
                   "virtual std::vector<tensor> backward(std::vector<tensor>& delta) = 0;",
                   "virtual void update_gradients(const data_type learning_rate = 1e-4)",
                   "Conv2D(std::string _name, const int _in_channels = 3, const int _out_channels = 16, const int _kernel_size = 3,",
                   "MaxPool2D(std::string _name, const int _kernel_size = 2, const int _step = 2)", "ReLU(std::string _name)",
                   "LinearLayer(std::string _name, const int _in_channels, const int _out_channels);", "class WithoutGrad",
                   "extern bool no_grad;", "extern data_type random_times;", "int get_params_num() const;"):
        assert needle in arch, needle
    fmt = open(os.path.join(root, "cnn_amd/host/include/data_format.h")).read()
    for needle in ("const int C, H, W;", "data_type* data;", "std::string name;", "using tensor = std::shared_ptr<Tensor3D>;",
                   "std::shared_ptr<Tensor3D> pad(const int padding = 1) const;", "int argmax() const;"):
        assert needle in fmt, needle


@pytest.mark.gpu
def test_host_net_readme_known_answer_and_checkpoint_roundtrip(golden_dir, tmp_path):
    from cnn_amd import hostapi
    from oracle import pyoracle as O

    imgs = np.load(os.path.join(golden_dir, "readme_kat_images_u8.npy"))
    exp = json.load(open(os.path.join(golden_dir, "readme_kat_expected.json")))
    ckpt = os.path.join(golden_dir, "readme_kat_checkpoint.model")
    x = np.ascontiguousarray((imgs.astype(np.float32) * np.float32(1.0) / np.float32(255)).transpose(0, 3, 1, 2))
    net = hostapi.HostAlexNet(3)
    assert net.n_params == 111267
    net.load_checkpoint(ckpt)  # AlexNet::load_weights, byte layout of alexnet.cpp:69-90
    hostapi.load().cnnh_set_no_grad(1)  # inference.cpp:50 runs under WithoutGrad
    try:
        for i in range(3):  # inference.cpp feeds one image at a time into buffers sized by the first call
            single = hostapi.HostAlexNet(3)
            single.load_checkpoint(ckpt)
            probs = O.softmax(single.forward_host(x[i : i + 1]))
            assert int(probs.argmax()) == exp["argmax"][i] and abs(float(probs.max()) - exp["prob"][i]) < 3e-6
            single.close()
    finally:
        hostapi.load().cnnh_set_no_grad(0)
    out = tmp_path / "roundtrip.model"
    net.save_checkpoint(out)
    assert open(out, "rb").read() == open(ckpt, "rb").read()  # bit-identical file through device memory


@pytest.mark.gpu
def test_host_net_train_steps_vs_oracle():
    """cnn.cpp:79-90 driven through the C++ classes (host tensors in, host softmax/CE) against the oracle"""
    from cnn_amd import hostapi
    from oracle import pyoracle as O

    B = 4
    x = uniform01(30, (B, 3, 224, 224))
    labels = (np.arange(B) % 3).astype(np.int32)
    onet = O.Net(B, 3)
    p0 = normal_scaled(31, (onet.n_params,))
    onet.params[:] = p0
    net = hostapi.HostAlexNet(3)
    net.set_params(p0)
    for step in range(2):
        loss, probs = net.train_step_host(x, labels, 1e-3)
        oloss, oprobs = onet.train_step(x, labels, 1e-3)
        assert np.isclose(loss, oloss, rtol=1e-4), (step, loss, oloss)
        assert_close(probs, oprobs, REL_TOL, f"step{step} probs")
        assert_close(net.get_grads(), onet.grads, 2e-4, f"step{step} grads")
        assert_close(net.get_params(), onet.params, REL_TOL, f"step{step} params")
    # Layer::get_output() materialises any layer's activation on the host (alexnet.cpp:97,105 contract)
    onet.forward(x)
    hostapi.load().cnnh_set_no_grad(1)
    net.forward_host(x)
    hostapi.load().cnnh_set_no_grad(0)
    assert_close(net.layer_output("conv_layer_3", (B, 64, 13, 13)), onet.conv_out(2), REL_TOL, "conv_layer_3 output")
    assert_close(net.layer_output("max_pool_1", (B, 16, 55, 55)), onet.pool_out(), REL_TOL, "max_pool_1 output")


@pytest.mark.gpu
def test_host_net_device_batch_and_external_arena():
    """zero-copy device batch + caller-owned (torch) arenas: what bench.py uses for the data-parallel step"""
    import torch

    from cnn_amd import hostapi
    from oracle import pyoracle as O

    B = 3
    x = uniform01(40, (B, 3, 224, 224))
    labels = np.array([2, 0, 1], np.int32)
    onet = O.Net(B, 3)
    p0 = normal_scaled(41, (onet.n_params,))
    onet.params[:] = p0
    params = torch.from_numpy(p0.copy()).cuda()
    grads = torch.zeros_like(params)
    net = hostapi.HostAlexNet(3, params, grads)
    net.set_params(p0)  # (binding copies each layer's own init into the arena; overwrite with the test weights)
    xd = torch.from_numpy(x).cuda()
    loss = net.train_step_device(xd, labels, 1e-3, do_update=False)
    oloss, _ = onet.train_step(x, labels, 1e-3)
    assert np.isclose(loss, oloss, rtol=1e-4)
    assert_close(grads.cpu().numpy(), onet.grads, 2e-4, "grads in the caller's arena")
    net.update(1e-3, 1.0)
    torch.cuda.synchronize()
    assert_close(params.cpu().numpy(), onet.params, REL_TOL, "params in the caller's arena")
