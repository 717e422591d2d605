// func.cpp -- softmax / one_hot / cross-entropy with the reference's semantics (cpu/src/func.cpp:6-81), host side.
#include <cassert>
#include <cfloat>
#include <cmath>
#include <algorithm>
#include <cstdio>

#include "func.h"

namespace {
inline data_type clamped_exp(const data_type x) {  // func.cpp:7-11
    if (x >= 88) return FLT_MAX;
    if (x <= -50) return 0.f;
    return std::exp(x);
}
}  // namespace

std::vector<tensor> softmax(const std::vector<tensor>& input) {
    std::vector<tensor> output;
    output.reserve(input.size());
    for (const auto& in : input) {
        const int n = in->get_length();
        tensor probs(new Tensor3D(n));
        const data_type mx = in->max();
        data_type sum = 0;
        for (int i = 0; i < n; ++i) {
            probs->data[i] = clamped_exp(in->data[i] - mx);
            sum += probs->data[i];
        }
        for (int i = 0; i < n; ++i) probs->data[i] /= sum;
        for (int i = 0; i < n; ++i)
            if (std::isnan(probs->data[i])) probs->data[i] = 0.f;  // func.cpp:33
        output.emplace_back(std::move(probs));
    }
    return output;
}

std::vector<tensor> one_hot(const std::vector<int>& labels, const int num_classes) {
    std::vector<tensor> codes;
    codes.reserve(labels.size());
    for (const int label : labels) {
        assert(label >= 0 && label < num_classes);
        tensor sample(new Tensor3D(num_classes));
        for (int i = 0; i < num_classes; ++i) sample->data[i] = (i == label) ? 1.f : 0.f;
        codes.emplace_back(std::move(sample));
    }
    return codes;
}

// delta = p - y with NO 1/B (the layers average, conv2d.cpp:148, linear.cpp:62); loss = -(sum log(p)*y)/B
std::pair<data_type, std::vector<tensor> > cross_entroy_backward(const std::vector<tensor>& probs,
                                                                  const std::vector<tensor>& labels) {
    const int batch_size = (int)labels.size();
    const int n = probs[0]->get_length();
    std::vector<tensor> delta;
    delta.reserve(batch_size);
    data_type loss = 0;
    for (int b = 0; b < batch_size; ++b) {
        tensor piece(new Tensor3D(n));
        for (int i = 0; i < n; ++i) {
            piece->data[i] = probs[b]->data[i] - labels[b]->data[i];
            loss += std::log(probs[b]->data[i]) * labels[b]->data[i];
        }
        delta.emplace_back(std::move(piece));
    }
    loss = loss * (-1.0) / batch_size;
    return std::make_pair(loss, delta);
}

// func.cpp:75-81: fixed-point rendering with `precision` decimals (the drivers build checkpoint file names with it)
std::string float_to_string(const float value, const int precision) {
    char text[64];
    const int n = std::snprintf(text, sizeof(text), "%.*f", precision < 0 ? 0 : precision, static_cast<double>(value));
    return std::string(text, n < 0 ? 0 : std::min<size_t>((size_t)n, sizeof(text) - 1));
}
