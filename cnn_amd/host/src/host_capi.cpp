// host_capi.cpp -- a flat C handle API over the C++ host classes, used by tests/ and bench.py to drive the SAME
// code a reference user would call (AlexNet::forward / backward / update_gradients, func.cpp's softmax and
// cross_entroy_backward) from Python.  Not part of the drop-in boundary (that is include/cnn_amd.h).
#include <cstring>
#include <stdexcept>
#include <filesystem>
#include <map>
#include <vector>

#include "architectures.h"
#include "func.h"
#include "host_util.h"

using namespace architectures;
using cnn_amd_host::must;

namespace {
struct Handle {
    Sequential* net;
    int classes;
    int in_C = 3;
    std::vector<tensor> host_input;     // B host tensors the caller's images are copied into (cnn.cpp's DataLoader role)
    std::vector<tensor> device_input;   // zero-copy views of a caller-owned contiguous device batch
    std::map<const float*, std::vector<tensor> > device_inputs;  // ... one set per batch buffer (a staged input alternates between a few)
    std::vector<tensor> last_output;
};
std::vector<tensor>& host_batch(Handle* h, int B, int C, int H, int W) {
    if ((int)h->host_input.size() != B) {
        h->host_input.clear();
        for (int b = 0; b < B; ++b) h->host_input.emplace_back(new Tensor3D(C, H, W, "input_" + std::to_string(b)));
    }
    return h->host_input;
}
}  // namespace

extern "C" {

void* cnnh_net_create(int classes, float* params_dev, float* grads_dev) {
    Handle* h = new Handle();
    h->classes = classes;
    h->net = (params_dev && grads_dev) ? new AlexNet(classes, params_dev, grads_dev) : new AlexNet(classes, false);
    return h;
}
// batch_norm != 0: AlexNet(classes, true) (alexnet.cpp:13,17,20,23)
void* cnnh_net_create_ex(int classes, float* params_dev, float* grads_dev, int batch_norm) {
    Handle* h = new Handle();
    h->classes = classes;
    h->net = (params_dev && grads_dev) ? new AlexNet(classes, params_dev, grads_dev, batch_norm != 0)
                                       : new AlexNet(classes, batch_norm != 0);
    return h;
}
// ---- generic layer lists: cnnh_seq_create, cnnh_seq_add_* in layer order, cnnh_seq_finalize; afterwards every cnnh_net_*
// entry point works on the handle.  (The layer kinds / argument orders are the reference's constructors, architectures.h:69,
// 96, 109, 131, 167.)
void* cnnh_seq_create(int in_channels, int classes) {
    Handle* h = new Handle();
    h->classes = classes;
    h->in_C = in_channels;
    h->net = new Sequential();
    return h;
}
void cnnh_seq_add_conv(void* hv, const char* name, int ci, int co, int k, int stride, int pad) {
    ((Handle*)hv)->net->add(new Conv2D(name, ci, co, k, stride, pad));
}
void cnnh_seq_add_bn(void* hv, const char* name, int channels) { ((Handle*)hv)->net->add(new BatchNorm2D(name, channels)); }
void cnnh_seq_add_dropout(void* hv, const char* name, float p) { ((Handle*)hv)->net->add(new Dropout(name, p)); }
void cnnh_seq_add_relu(void* hv, const char* name) { ((Handle*)hv)->net->add(new ReLU(name)); }
void cnnh_seq_add_pool(void* hv, const char* name, int k, int step) { ((Handle*)hv)->net->add(new MaxPool2D(name, k, step)); }
void cnnh_seq_add_linear(void* hv, const char* name, int n_in, int n_out) { ((Handle*)hv)->net->add(new LinearLayer(name, n_in, n_out)); }
void cnnh_seq_finalize(void* hv, float* params_dev, float* grads_dev) {
    Handle* h = (Handle*)hv;
    if (params_dev && grads_dev)
        h->net->finalize(params_dev, grads_dev);
    else
        h->net->finalize();
}
// the two BASELINE stacks from the C++ builders (sequential.cpp); the Python side checks them against cnn_amd/stacks.py
void* cnnh_stack_create(const char* which, int classes, int batch_norm) {
    Handle* h = new Handle();
    h->classes = classes;
    h->net = new Sequential();
    const std::string w(which);
    if (w == "vgg11")
        build_vgg11(*h->net, classes, batch_norm != 0);
    else if (w == "resnet18")
        build_resnet18(*h->net, classes, batch_norm != 0);
    else {
        delete h->net;
        delete h;
        return nullptr;
    }
    return h;
}
// "<name>:<param_count>\n" per layer, in order; returns bytes needed
size_t cnnh_net_describe(void* hv, char* buf, size_t cap) {
    std::string out;
    for (const auto& layer : ((Handle*)hv)->net->layers()) out += layer->name + ":" + std::to_string(layer->param_count()) + "\n";
    if (buf && out.size() + 1 <= cap) std::memcpy(buf, out.c_str(), out.size() + 1);
    return out.size() + 1;
}
// data parallelism: RCCL communicator from cnn_comm_init_rank / cnn_comm_init_all (include/cnn_amd.h)
void cnnh_net_set_comm(void* hv, void* comm, int world) { ((Handle*)hv)->net->set_comm(comm, world); }
void cnnh_net_allreduce_gradients(void* hv) { ((Handle*)hv)->net->allreduce_gradients(); }

void cnnh_net_destroy(void* hv) {
    Handle* h = (Handle*)hv;
    delete h->net;
    delete h;
}
size_t cnnh_net_num_params(void* hv) { return ((Handle*)hv)->net->num_params(); }
float* cnnh_net_params_device(void* hv) { return ((Handle*)hv)->net->params_device(); }
float* cnnh_net_grads_device(void* hv) { return ((Handle*)hv)->net->grads_device(); }
void cnnh_set_stream(void* hip_stream) { architectures::stream = hip_stream; }
void cnnh_set_no_grad(int on) { architectures::no_grad = on != 0; }
void cnnh_set_fuse_layers(int on) { architectures::fuse_layers = on != 0; }
void cnnh_set_fuse_pool_block(int on) { architectures::fuse_pool_block = on != 0; }
void cnnh_set_input_gradient(int on) { architectures::input_gradient = on != 0; }

void cnnh_net_set_params(void* hv, const float* host) {
    Handle* h = (Handle*)hv;
    h->net->flush_deferred();
    must(cnn_memcpy_h2d(h->net->params_device(), host, sizeof(float) * h->net->num_params(), stream), "cnn_memcpy_h2d");
    must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
    h->net->parameters_changed();
}
void cnnh_net_get_params(void* hv, float* host) {
    Handle* h = (Handle*)hv;
    h->net->flush_deferred();
    must(cnn_memcpy_d2h(host, h->net->params_device(), sizeof(float) * h->net->num_params(), stream), "cnn_memcpy_d2h");
    must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
}
void cnnh_net_get_grads(void* hv, float* host) {
    Handle* h = (Handle*)hv;
    h->net->flush_deferred();
    must(cnn_memcpy_d2h(host, h->net->grads_device(), sizeof(float) * h->net->num_params(), stream), "cnn_memcpy_d2h");
    must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
}
int cnnh_net_load_checkpoint(void* hv, const char* path) {
    if (!std::filesystem::exists(path)) return 1;
    ((Handle*)hv)->net->load_weights(path);
    return 0;
}
void cnnh_net_save_checkpoint(void* hv, const char* path) { ((Handle*)hv)->net->save_weights(path); }

// forward on HOST images [B][3][H][W] (each sample becomes one host Tensor3D, like the reference's DataLoader buffers);
// logits_out [B][classes]
void cnnh_net_forward_host(void* hv, const float* x, int B, int H, int W, float* logits_out) {
    Handle* h = (Handle*)hv;
    auto& in = host_batch(h, B, h->in_C, H, W);
    const size_t len = (size_t)h->in_C * H * W;
    for (int b = 0; b < B; ++b) std::memcpy(in[b]->data, x + len * b, sizeof(float) * len);
    h->last_output = h->net->forward(in);
    for (int b = 0; b < B; ++b) std::memcpy(logits_out + (size_t)b * h->classes, h->last_output[b]->data, sizeof(float) * h->classes);
}

// one iteration of cnn.cpp:79-90 on HOST images: forward, softmax, one_hot, cross_entroy_backward, backward, SGD
float cnnh_net_train_step_host(void* hv, const float* x, const int* labels, int B, int H, int W, float lr, float* probs_out) {
    Handle* h = (Handle*)hv;
    auto& in = host_batch(h, B, h->in_C, H, W);
    const size_t len = (size_t)h->in_C * H * W;
    for (int b = 0; b < B; ++b) std::memcpy(in[b]->data, x + len * b, sizeof(float) * len);
    const auto output = h->net->forward(in);
    const auto probs = softmax(output);
    auto loss_delta = cross_entroy_backward(probs, one_hot(std::vector<int>(labels, labels + B), h->classes));
    h->net->backward(loss_delta.second);
    h->net->update_gradients(lr);
    if (probs_out)
        for (int b = 0; b < B; ++b) std::memcpy(probs_out + (size_t)b * h->classes, probs[b]->data, sizeof(float) * h->classes);
    return loss_delta.first;
}

// the same step on a caller-owned contiguous DEVICE batch (zero-copy views); returns the loss.  do_update = 0 leaves
// the gradients in the arena so the caller can all-reduce them before cnnh_net_update().
float cnnh_net_train_step_device(void* hv, float* x_dev, const int* labels, int B, int H, int W, float lr, int do_update) {
    Handle* h = (Handle*)hv;
    if ((int)h->device_input.size() != B || h->device_input[0]->dev != x_dev) {
        h->device_input.clear();
        const size_t len = (size_t)h->in_C * H * W;
        for (int b = 0; b < B; ++b)
            h->device_input.emplace_back(Tensor3D::device_view(h->in_C, H, W, x_dev + len * b, "input_" + std::to_string(b)));
    }
    const auto output = h->net->forward(h->device_input);
    const auto probs = softmax(output);
    auto loss_delta = cross_entroy_backward(probs, one_hot(std::vector<int>(labels, labels + B), h->classes));
    h->net->backward(loss_delta.second);
    if (do_update) h->net->update_gradients(lr);
    return loss_delta.first;
}
// Sequential::train_step: the whole iteration on the device (device-side loss, no read-back, the host never blocks);
// labels_dev int32 [B] on the device.  cnnh_net_last_loss fetches the loss of the latest step (synchronises).
void cnnh_net_train_step_device_loss(void* hv, float* x_dev, const int* labels_dev, int B, int H, int W, float lr) {
    Handle* h = (Handle*)hv;
    auto& views = h->device_inputs[x_dev];
    if ((int)views.size() != B) {
        views.clear();
        const size_t len = (size_t)h->in_C * H * W;
        for (int b = 0; b < B; ++b) views.emplace_back(Tensor3D::device_view(h->in_C, H, W, x_dev + len * b, "input_" + std::to_string(b)));
    }
    h->net->train_step(views, labels_dev, lr);
}
float cnnh_net_last_loss(void* hv) { return ((Handle*)hv)->net->last_loss(); }
// the delta with respect to the network INPUT (the first layer's data gradient, conv2d.cpp:168-199; alexnet.cpp:55 discards it) of
// the last backward pass -- after flush_deferred(), because train_step may have deferred that kernel.  0 ok, 1 no such tensor, 2 cap
int cnnh_net_input_delta(void* hv, float* out, size_t cap_floats) {
    Handle* h = (Handle*)hv;
    auto* conv = dynamic_cast<Conv2D*>(h->net->layers().front().get());
    if (conv == nullptr || conv->delta_dev() == nullptr) return 1;
    if (conv->delta_floats() > cap_floats) return 2;
    if (!conv->delta_computed()) return 3;  // (the last train_step ran with architectures::input_gradient = false)
    h->net->flush_deferred();
    must(cnn_memcpy_d2h(out, conv->delta_dev(), sizeof(float) * conv->delta_floats(), stream), "cnn_memcpy_d2h");
    must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
    return 0;
}
// Sequential::flush_deferred: a data gradient train_step deferred into the next pass is launched / ordered now
void cnnh_net_flush(void* hv) { ((Handle*)hv)->net->flush_deferred(); }
void cnnh_net_update(void* hv, float lr, float grad_scale) { ((Handle*)hv)->net->update_gradients(lr, grad_scale); }
// Sequential::update_gradients(lr): with a communicator set, all-reduce + lr/world; otherwise the plain step
void cnnh_net_update_auto(void* hv, float lr) { ((Handle*)hv)->net->update_gradients(lr); }

// host copy of a layer's last output through Layer::get_output() (the Grad-CAM contract, alexnet.cpp:97,105)
int cnnh_net_layer_output(void* hv, const char* layer_name, float* out, size_t cap_floats) {
    Handle* h = (Handle*)hv;
    h->net->flush_deferred();  // (a re-materialising get_output() uses the layer's workspace: nothing deferred may still read it)
    for (const auto& layer : h->net->layers()) {
        if (layer->name != layer_name) continue;
        std::vector<tensor> ts;
        try {
            ts = layer->get_output();
        } catch (const std::exception& e) {  // (a fused-away tensor whose parameters are gone: layers.cpp materialize())
            std::fprintf(stderr, "%s\n", e.what());
            return 3;
        }
        size_t off = 0;
        for (const auto& t : ts) {
            const size_t n = (size_t)t->get_length();
            if (off + n > cap_floats) return 2;
            std::memcpy(out + off, t->data, sizeof(float) * n);
            off += n;
        }
        return 0;
    }
    return 1;
}

// Sequential::grad_cam (alexnet.cpp:95-142): image_out H*W bytes, cam_out (nullable) B*H*W floats; returns 0, or 1 for an unknown layer
int cnnh_net_grad_cam(void* hv, const char* layer_name, unsigned char* image_out, size_t image_cap, float* cam_out, size_t cam_cap) {
    Handle* h = (Handle*)hv;
    bool found = false;
    for (const auto& layer : h->net->layers()) found = found || layer->name == layer_name;
    if (!found) return 1;
    std::vector<float> cam;
    std::vector<uchar> img;
    try {  // (grad_cam reads the layer through get_output(): same "tensor cannot be re-computed" case as cnnh_net_layer_output -> 3)
        img = h->net->grad_cam(layer_name, cam_out ? &cam : nullptr);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 3;
    }
    if (img.size() > image_cap || (cam_out && cam.size() > cam_cap)) return 2;
    std::memcpy(image_out, img.data(), img.size());
    if (cam_out) std::memcpy(cam_out, cam.data(), sizeof(float) * cam.size());
    return 0;
}

}  // extern "C"
