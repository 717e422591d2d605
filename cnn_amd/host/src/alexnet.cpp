// alexnet.cpp -- class AlexNet (cpu/src/alexnet.cpp:10-90) on top of Sequential: the reference's fixed layer list, one flat
// parameter arena, fused / prepared kernels.  This is the ONE translation unit of cnn_amd/host/src that the reference's own
// cpu/src/alexnet.cpp can stand in for: that file DEFINES the same members (constructor, forward, backward,
// update_gradients(lr), save_weights, load_weights) as plain walks over layers_sequence, and compiles against
// cnn_amd/host/include up to its OpenCV-typed grad_cam (:95; tests/test_boundary_compile.py compiles it where the reference tree
// exists).  tests/caller/plain_list_caller.cpp drives the same public Layer API WITHOUT this container and ends byte-identical.
#include <cassert>

#include "architectures.h"

using namespace architectures;

// ---------------------------------------------------------------------------------------------------------------
// AlexNet: the reference's list (alexnet.cpp:12-31)
namespace {
void build_alexnet(Sequential& net, int num_classes, bool batch_norm) {
    // every convolution is 3x3 with the constructor's default stride 2; one MaxPool(2,2); batch_norm inserts a BatchNorm2D
    // between each convolution and its ReLU (alexnet.cpp:13,17,20,23)
    const int chans[5] = {3, 16, 32, 64, 128};
    for (int l = 1; l <= 4; ++l) {
        const std::string id = std::to_string(l);
        net.add(new Conv2D("conv_layer_" + id, chans[l - 1], chans[l], 3));
        if (batch_norm) net.add(new BatchNorm2D("bn_layer_" + id, chans[l]));
        net.add(new ReLU("relu_layer_" + id));
        if (l == 1) net.add(new MaxPool2D("max_pool_1", 2, 2));
    }
    net.add(new LinearLayer("linear_1", 6 * 6 * 128, num_classes));
}
}  // namespace

AlexNet::AlexNet(const int num_classes, const bool batch_norm) {
    build_alexnet(*this, num_classes, batch_norm);
    finalize();
}

AlexNet::AlexNet(const int num_classes, data_type* params_dev, data_type* grads_dev, const bool batch_norm) {
    build_alexnet(*this, num_classes, batch_norm);
    finalize(params_dev, grads_dev);
}

std::vector<tensor> AlexNet::forward(const std::vector<tensor>& input) { return Sequential::forward(input); }
void AlexNet::backward(std::vector<tensor>& delta_start) { Sequential::backward(delta_start); }
void AlexNet::update_gradients(const data_type learning_rate) { Sequential::update_gradients(learning_rate); }
void AlexNet::update_gradients(const data_type learning_rate, const data_type grad_scale) {
    Sequential::update_gradients(learning_rate, grad_scale);
}
void AlexNet::save_weights(const std::filesystem::path& save_path) const { Sequential::save_weights(save_path); }
void AlexNet::load_weights(const std::filesystem::path& checkpoint_path) { Sequential::load_weights(checkpoint_path); }

