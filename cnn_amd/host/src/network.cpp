// network.cpp -- the reference's AlexNet container (cpu/src/alexnet.cpp:10-90) over the device layers, plus the
// flat parameter / gradient arena that makes the SGD step one kernel and the data-parallel exchange one all-reduce.
#include <cassert>
#include <iostream>
#include <iterator>

#include "architectures.h"
#include "host_util.h"

using namespace architectures;
using cnn_amd_host::dev_alloc;
using cnn_amd_host::must;

void AlexNet::build(int num_classes, bool batch_norm) {
    // alexnet.cpp:12-31: every convolution is 3x3 with the constructor's default stride 2; one MaxPool(2,2);
    // batch_norm inserts a BatchNorm2D between each convolution and its ReLU (:13,17,20,23)
    layers_sequence.emplace_back(new Conv2D("conv_layer_1", 3, 16, 3));
    if (batch_norm) layers_sequence.emplace_back(new BatchNorm2D("bn_layer_1", 16));
    layers_sequence.emplace_back(new ReLU("relu_layer_1"));
    layers_sequence.emplace_back(new MaxPool2D("max_pool_1", 2, 2));
    layers_sequence.emplace_back(new Conv2D("conv_layer_2", 16, 32, 3));
    if (batch_norm) layers_sequence.emplace_back(new BatchNorm2D("bn_layer_2", 32));
    layers_sequence.emplace_back(new ReLU("relu_layer_2"));
    layers_sequence.emplace_back(new Conv2D("conv_layer_3", 32, 64, 3));
    if (batch_norm) layers_sequence.emplace_back(new BatchNorm2D("bn_layer_3", 64));
    layers_sequence.emplace_back(new ReLU("relu_layer_3"));
    layers_sequence.emplace_back(new Conv2D("conv_layer_4", 64, 128, 3));
    if (batch_norm) layers_sequence.emplace_back(new BatchNorm2D("bn_layer_4", 128));
    layers_sequence.emplace_back(new ReLU("relu_layer_4"));
    layers_sequence.emplace_back(new LinearLayer("linear_1", 6 * 6 * 128, num_classes));
    // fusion wiring (architectures::fuse_layers): Conv2D -> ReLU forward, ReLU -> MaxPool2D backward
    for (auto it = layers_sequence.begin(); it != layers_sequence.end(); ++it) {
        auto next = std::next(it);
        if (next == layers_sequence.end()) break;
        if (auto* relu = dynamic_cast<ReLU*>(next->get())) {
            if (auto* conv = dynamic_cast<Conv2D*>(it->get())) conv->set_fused_relu(relu);
        }
        if (auto* pool = dynamic_cast<MaxPool2D*>(next->get())) {
            if (auto* relu = dynamic_cast<ReLU*>(it->get())) pool->set_fused_relu_below(relu);
        }
        // ReLU -> Conv2D / LinearLayer: the ReLU's backward pass runs inside the consumer's data-gradient kernel
        if (auto* relu = dynamic_cast<ReLU*>(it->get())) {
            if (auto* conv = dynamic_cast<Conv2D*>(next->get())) conv->set_relu_below(relu);
            if (auto* lin = dynamic_cast<LinearLayer*>(next->get())) lin->set_relu_below(relu);
        }
        // Conv2D -> ReLU -> MaxPool2D: one forward kernel, backward from the pooled domain
        auto next2 = std::next(next);
        if (next2 != layers_sequence.end()) {
            auto* conv = dynamic_cast<Conv2D*>(it->get());
            auto* relu = dynamic_cast<ReLU*>(next->get());
            auto* pool = dynamic_cast<MaxPool2D*>(next2->get());
            if (conv && relu && pool) conv->set_fused_pool(pool);
        }
    }
    n_params = 0;
    for (const auto& layer : layers_sequence) n_params += layer->param_count();
}

void AlexNet::bind(data_type* p, data_type* g) {
    param_arena = p;
    grad_arena = g;
    size_t off = 0;  // checkpoint order == layer order (alexnet.cpp:73-74)
    for (auto& layer : layers_sequence) {
        const size_t n = layer->param_count();
        if (n) layer->bind_arena(p + off, g + off);
        off += n;
    }
}

AlexNet::AlexNet(const int num_classes, const bool batch_norm) {
    build(num_classes, batch_norm);
    owns_arena = true;
    data_type* p = (data_type*)dev_alloc(sizeof(data_type) * n_params);
    data_type* g = (data_type*)dev_alloc(sizeof(data_type) * n_params);
    must(cnn_memset_zero(g, sizeof(data_type) * n_params, stream), "cnn_memset_zero");
    bind(p, g);
}

AlexNet::AlexNet(const int num_classes, data_type* params_dev, data_type* grads_dev, const bool batch_norm) {
    build(num_classes, batch_norm);
    owns_arena = false;
    bind(params_dev, grads_dev);
}

AlexNet::~AlexNet() {
    layers_sequence.clear();
    if (owns_arena) {
        cnn_device_free(param_arena);
        cnn_device_free(grad_arena);
    }
}

// One cnn_conv2d_prepare_filters call for all convolutions (instead of one small re-layout launch inside every forward
// and backward call); possible once every layer has seen its input shape, i.e. from the second forward pass on.
void AlexNet::prepare_filters() {
    std::vector<Conv2D*> convs;
    for (auto& layer : layers_sequence)
        if (auto* c = dynamic_cast<Conv2D*>(layer.get())) convs.push_back(c);
    for (auto* c : convs)
        if (!c->shape_known()) return;
    if (convs.empty() || convs.size() > 6) return;
    std::vector<cnn_conv2d_desc> descs;
    std::vector<const float*> w, b;
    std::vector<void*> f(convs.size()), g(convs.size());
    for (size_t i = 0; i < convs.size(); ++i) {
        descs.push_back(convs[i]->current_desc());
        w.push_back(convs[i]->filters_dev());
        b.push_back(convs[i]->bias_dev());
        convs[i]->prepared_buffers(&f[i], &g[i]);
    }
    must(cnn_conv2d_prepare_filters((int)convs.size(), descs.data(), w.data(), b.data(), f.data(), g.data(), stream),
         "cnn_conv2d_prepare_filters");
    for (auto* c : convs) c->set_prepared(true);
    filters_prepared = true;
}

std::vector<tensor> AlexNet::forward(const std::vector<tensor>& input) {
    assert(input.size() > 0);
    if (print_info) input[0]->print_shape();
    if (fuse_layers && !filters_prepared) prepare_filters();
    std::vector<tensor> output(input);
    for (const auto& layer : layers_sequence) {
        output = layer->forward(output);
        if (print_info) output[0]->print_shape();
    }
    return output;
}

void AlexNet::backward(std::vector<tensor>& delta_start) {
    if (print_info) delta_start[0]->print_shape();
    for (auto layer = layers_sequence.rbegin(); layer != layers_sequence.rend(); ++layer) {
        delta_start = (*layer)->backward(delta_start);
        if (print_info) delta_start[0]->print_shape();
    }
    // the layers' weight gradients were computed on the library's side stream: order them before whatever follows
    must(cnn_amd_side_stream_join(stream), "cnn_amd_side_stream_join");
}

void AlexNet::parameters_changed() {
    filters_prepared = false;  // re-prepared at the start of the next forward pass
    for (auto& layer : layers_sequence)
        if (auto* c = dynamic_cast<Conv2D*>(layer.get())) c->set_prepared(false);
}

void AlexNet::update_gradients(const data_type learning_rate, const data_type grad_scale) {
    must(cnn_sgd_update(param_arena, grad_arena, n_params, learning_rate, grad_scale, stream), "cnn_sgd_update");
    parameters_changed();
}

void AlexNet::save_weights(const std::filesystem::path& save_path) const {
    std::ofstream writer(save_path.c_str(), std::ios::binary);
    for (const auto& layer : layers_sequence) layer->save_weights(writer);
    std::cout << "weights have been saved to " << save_path.string() << std::endl;
    writer.close();
}

void AlexNet::load_weights(const std::filesystem::path& checkpoint_path) {
    if (!std::filesystem::exists(checkpoint_path)) {  // alexnet.cpp:81-84: report and carry on
        std::cout << "checkpoint file  " << checkpoint_path << " does not exist !\n";
        return;
    }
    std::ifstream reader(checkpoint_path.c_str(), std::ios::binary);
    parameters_changed();
    for (auto& layer : layers_sequence) layer->load_weights(reader);
    std::cout << "load weights from" << checkpoint_path.string() << std::endl;
    reader.close();
}
