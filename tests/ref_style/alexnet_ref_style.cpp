// alexnet_ref_style.cpp -- TEST FIXTURE: what a hermosayhl/CNN user's own model + training-loop code looks like, written against
// cnn_amd/host/include instead of cpu/include.  It restates, call for call,
//   * cpu/src/alexnet.cpp:10-90 -- the AlexNet members exactly as the reference DEFINES them (the constructor fills
//     layers_sequence with `new Conv2D(...)` etc.; forward / backward / update_gradients / save_weights / load_weights are plain
//     walks over the list).  Linked INSTEAD of cnn_amd/host/src/alexnet.cpp, it proves that the reference's own container drops
//     in on top of the device layers: no arena, no fusion wiring, no prepared filters -- just Layer::forward / backward.
//   * cpu/src/cnn.cpp:77-93 -- one training iteration: forward, softmax, one_hot, cross_entroy_backward, backward,
//     update_gradients, argmax (run_reference_style_steps below; the DataLoader is replaced by caller-provided host images).
// tests/test_boundary_compile.py syntax-checks this file (and the reference's own cpu/src/func.cpp when /root/reference is
// present) on the CPU box, and builds + runs it on the GPU box against the arena-based AlexNet.
// The reference's source text is not copied: only its calls are reproduced.
#include <cstdio>
#include <cstring>
#include <iostream>

#include "architectures.h"
#include "func.h"

using namespace architectures;

// ---- cpu/src/alexnet.cpp:10-33
AlexNet::AlexNet(const int num_classes, const bool batch_norm) {
    this->layers_sequence.emplace_back(new Conv2D("conv_layer_1", 3, 16, 3));
    if (batch_norm) this->layers_sequence.emplace_back(new BatchNorm2D("bn_layer_1", 16));
    this->layers_sequence.emplace_back(new ReLU("relu_layer_1"));
    this->layers_sequence.emplace_back(new MaxPool2D("max_pool_1", 2, 2));
    this->layers_sequence.emplace_back(new Conv2D("conv_layer_2", 16, 32, 3));
    if (batch_norm) this->layers_sequence.emplace_back(new BatchNorm2D("bn_layer_2", 32));
    this->layers_sequence.emplace_back(new ReLU("relu_layer_2"));
    this->layers_sequence.emplace_back(new Conv2D("conv_layer_3", 32, 64, 3));
    if (batch_norm) this->layers_sequence.emplace_back(new BatchNorm2D("bn_layer_3", 64));
    this->layers_sequence.emplace_back(new ReLU("relu_layer_3"));
    this->layers_sequence.emplace_back(new Conv2D("conv_layer_4", 64, 128, 3));
    if (batch_norm) this->layers_sequence.emplace_back(new BatchNorm2D("bn_layer_4", 128));
    this->layers_sequence.emplace_back(new ReLU("relu_layer_4"));
    this->layers_sequence.emplace_back(new LinearLayer("linear_1", 6 * 6 * 128, num_classes));
}

// (the arena-adopting constructor is an addition of this build; a reference-style caller never uses it)
AlexNet::AlexNet(const int num_classes, data_type*, data_type*, const bool batch_norm) : AlexNet(num_classes, batch_norm) {}

// ---- cpu/src/alexnet.cpp:35-46
std::vector<tensor> AlexNet::forward(const std::vector<tensor>& input) {
    assert(input.size() > 0);
    if (this->print_info) input[0]->print_shape();
    std::vector<tensor> output(input);
    for (const auto& layer : this->layers_sequence) {
        output = layer->forward(output);
        if (this->print_info) output[0]->print_shape();
    }
    return output;
}

// ---- cpu/src/alexnet.cpp:49-59
void AlexNet::backward(std::vector<tensor>& delta_start) {
    if (this->print_info) delta_start[0]->print_shape();
    for (auto layer = layers_sequence.rbegin(); layer != layers_sequence.rend(); ++layer) {
        delta_start = (*layer)->backward(delta_start);
        if (this->print_info) delta_start[0]->print_shape();
    }
}

// ---- cpu/src/alexnet.cpp:62-65
void AlexNet::update_gradients(const data_type learning_rate) {
    for (auto& layer : this->layers_sequence) layer->update_gradients(learning_rate);
}
void AlexNet::update_gradients(const data_type learning_rate, const data_type) { this->update_gradients(learning_rate); }

// ---- cpu/src/alexnet.cpp:69-77
void AlexNet::save_weights(const std::filesystem::path& save_path) const {
    std::ofstream writer(save_path.c_str(), std::ios::binary);
    for (const auto& layer : this->layers_sequence) layer->save_weights(writer);
    std::cout << "weights have been saved to " << save_path.string() << std::endl;
    writer.close();
}

// ---- cpu/src/alexnet.cpp:80-90
void AlexNet::load_weights(const std::filesystem::path& checkpoint_path) {
    if (!std::filesystem::exists(checkpoint_path)) {
        std::cout << "checkpoint  " << checkpoint_path << " does not exist !\n";
        return;
    }
    std::ifstream reader(checkpoint_path.c_str(), std::ios::binary);
    for (auto& layer : this->layers_sequence) layer->load_weights(reader);
    std::cout << "load weights from" << checkpoint_path.string() << std::endl;
    reader.close();
}

// ---- cpu/src/cnn.cpp:77-93, `steps` iterations on one caller-provided batch (images: [B][3][H][W] host floats).  Loads the
// checkpoint first (cnn.cpp:57 style), returns the mean loss, writes the last step's predictions.
extern "C" float run_reference_style_steps(const char* checkpoint, const float* images, const int* labels_in, int B, int H, int W,
                                           int steps, float learning_rate, int* predict_out, const char* save_to) {
    const int num_classes = 3;
    AlexNet network(num_classes, false);
    network.load_weights(checkpoint);
    std::vector<tensor> sample_images;
    for (int b = 0; b < B; ++b) {
        tensor t(new Tensor3D(3, H, W, "batch_" + std::to_string(b)));
        std::memcpy(t->data, images + (size_t)b * 3 * H * W, sizeof(float) * 3 * H * W);
        sample_images.emplace_back(t);
    }
    const std::vector<int> sample_labels(labels_in, labels_in + B);
    std::vector<int> predict(B, -1);
    float mean_loss = 0.f;
    for (int iter = 1; iter <= steps; ++iter) {
        const auto output = network.forward(sample_images);
        const auto probs = softmax(output);
        auto loss_delta = cross_entroy_backward(probs, one_hot(sample_labels, num_classes));
        mean_loss += loss_delta.first;
        network.backward(loss_delta.second);
        network.update_gradients(learning_rate);
        for (int b = 0; b < B; ++b) predict[b] = probs[b]->argmax();
    }
    for (int b = 0; b < B; ++b) predict_out[b] = predict[b];
    if (save_to && save_to[0]) network.save_weights(save_to);
    std::printf("reference-style loop: %d steps, mean loss %s\n", steps, float_to_string(mean_loss / steps, 6).c_str());
    return mean_loss / steps;
}
