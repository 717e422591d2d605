// plain_list_caller.cpp -- a caller of the drop-in boundary that is NOT this build's container: stand-alone layers held in a
// std::vector built from a run-time spec string, walked with index loops, every layer owning its own parameters (no flat arena,
// no fusion wiring, no prepared filters).  It drives the public Layer API only -- forward / backward / update_gradients /
// save_weights / load_weights -- plus func.h's loss glue, the way a user of the reference's headers would
// (cpu/include/architectures.h:34-46, cpu/include/func.h).  tests/test_boundary_compile.py links it BESIDE libcnn_amd_host.so
// and checks that two training steps end in a checkpoint that is byte-identical to architectures::AlexNet's (arena, fused and
// prepared kernels): the fusions of the container change nothing a caller can observe.
//
// spec grammar (one layer per ';'):  conv <name> <ci> <co> <k> <stride> | relu <name> | pool <name> <k> <step> | linear <name> <in> <out>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "architectures.h"
#include "func.h"

namespace {

struct PlainNet {
    std::vector<std::unique_ptr<architectures::Layer>> stack;

    explicit PlainNet(const std::string& spec) {
        std::istringstream all(spec);
        std::string item;
        while (std::getline(all, item, ';')) {
            std::istringstream in(item);
            std::string kind, name;
            if (!(in >> kind >> name)) continue;
            int a = 0, b = 0, c = 0, d = 0;
            if (kind == "conv") {
                in >> a >> b >> c >> d;
                stack.emplace_back(new architectures::Conv2D(name, a, b, c, d));
            } else if (kind == "relu") {
                stack.emplace_back(new architectures::ReLU(name));
            } else if (kind == "pool") {
                in >> a >> b;
                stack.emplace_back(new architectures::MaxPool2D(name, a, b));
            } else if (kind == "linear") {
                in >> a >> b;
                stack.emplace_back(new architectures::LinearLayer(name, a, b));
            } else {
                std::fprintf(stderr, "plain_list_caller: unknown layer kind '%s'\n", kind.c_str());
                std::abort();
            }
        }
    }

    std::vector<tensor> run(const std::vector<tensor>& images) {
        std::vector<tensor> cur = images;
        for (size_t i = 0; i < stack.size(); ++i) cur = stack[i]->forward(cur);
        return cur;
    }
    void run_back(std::vector<tensor>& delta) {
        for (size_t i = stack.size(); i-- > 0;) delta = stack[i]->backward(delta);
    }
    void step(float lr) {
        for (size_t i = 0; i < stack.size(); ++i) stack[i]->update_gradients(lr);
    }
    bool restore(const char* path) {
        std::ifstream in(path, std::ios::binary);
        if (!in) return false;
        for (auto& layer : stack) layer->load_weights(in);
        return true;
    }
    void store(const char* path) const {
        std::ofstream out(path, std::ios::binary);
        for (const auto& layer : stack) layer->save_weights(out);
    }
};

}  // namespace

// `steps` iterations of forward -> softmax -> cross_entroy_backward -> backward -> update on one fixed batch of HOST images
// [B][3][H][W]; returns the mean loss, writes the last step's arg-max predictions, saves the parameters to `save_to`.
extern "C" float plain_list_train(const char* spec, const char* checkpoint, const float* images, const int* labels, int B, int H, int W,
                                  int classes, int steps, float lr, int* predict_out, const char* save_to) {
    PlainNet net(spec);
    if (checkpoint && checkpoint[0] && !net.restore(checkpoint)) return -1.f;
    std::vector<tensor> batch;
    const size_t len = (size_t)3 * H * W;
    for (int b = 0; b < B; ++b) {
        batch.emplace_back(new Tensor3D(3, H, W, "img" + std::to_string(b)));
        std::memcpy(batch.back()->data, images + len * b, sizeof(float) * len);
    }
    const std::vector<int> y(labels, labels + B);
    double total = 0;
    for (int it = 0; it < steps; ++it) {
        const std::vector<tensor> logits = net.run(batch);
        const std::vector<tensor> probs = softmax(logits);
        auto loss_and_delta = cross_entroy_backward(probs, one_hot(y, classes));
        total += loss_and_delta.first;
        net.run_back(loss_and_delta.second);
        net.step(lr);
        for (int b = 0; b < B; ++b) predict_out[b] = probs[b]->argmax();
    }
    if (save_to && save_to[0]) net.store(save_to);
    return (float)(total / steps);
}
