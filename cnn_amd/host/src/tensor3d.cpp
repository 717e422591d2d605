// tensor3d.cpp -- Tensor3D helpers (semantics of cpu/src/data_format.cpp:13-158) + device-view plumbing.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>

#include "architectures.h"
#include "cnn_amd.h"

namespace {
void must(int rc, const char* what) {
    if (rc != 0) {
        std::fprintf(stderr, "cnn_amd host: %s failed (%d): %s\n", what, rc, cnn_amd_last_error());
        std::abort();
    }
}
}  // namespace

Tensor3D::Tensor3D(const int _C, const int _H, const int _W, const std::string _name)
    : C(_C), H(_H), W(_W), data(new data_type[(size_t)_C * _H * _W]), name(std::move(_name)) {}

Tensor3D::Tensor3D(const std::tuple<int, int, int>& shape, const std::string _name)
    : C(std::get<0>(shape)), H(std::get<1>(shape)), W(std::get<2>(shape)),
      data(new data_type[(size_t)std::get<0>(shape) * std::get<1>(shape) * std::get<2>(shape)]), name(std::move(_name)) {}

Tensor3D::Tensor3D(const int length, const std::string _name)
    : C(length), H(1), W(1), data(new data_type[length]), name(std::move(_name)) {}

Tensor3D::Tensor3D(ViewTag, int _C, int _H, int _W, data_type* dev_ptr, const std::string& _name)
    : C(_C), H(_H), W(_W), data(nullptr), name(_name), dev(dev_ptr) {}

std::shared_ptr<Tensor3D> Tensor3D::device_view(int C, int H, int W, data_type* dev_ptr, const std::string& name) {
    return std::shared_ptr<Tensor3D>(new Tensor3D(ViewTag{}, C, H, W, dev_ptr, name));
}

Tensor3D::~Tensor3D() noexcept {
    delete[] data;  // (the reference uses scalar delete on new[] memory, data_format.cpp:154 -- not reproduced)
    data = nullptr;
}

namespace {
// (the host classes are single-threaded by construction, like the reference: SURVEY 8b -- the counter itself is atomic so that two
// threads driving two containers do not race on it; a device view's HOST copy is a cache of the device tensor: host-side edits of
// `data` that were never uploaded are overwritten by the next refresh, see data_format.h)
std::atomic<unsigned long long> g_device_epoch{1};
}
void Tensor3D::device_work_enqueued() { g_device_epoch.fetch_add(1, std::memory_order_relaxed); }

void Tensor3D::sync_to_host() {
    if (!dev) return;
    if (!data) data = new data_type[(size_t)get_length()];
    host_epoch = g_device_epoch;
    must(cnn_memcpy_d2h(data, dev, sizeof(data_type) * (size_t)get_length(), architectures::stream), "cnn_memcpy_d2h");
    must(cnn_stream_synchronize(architectures::stream), "cnn_stream_synchronize");
}

void Tensor3D::sync_to_device() const {
    if (!dev || !data) return;
    must(cnn_memcpy_h2d(dev, data, sizeof(data_type) * (size_t)get_length(), architectures::stream), "cnn_memcpy_h2d");
}

// data_format.cpp:13-23: OpenCV stores BGR interleaved; plane c <- byte c of every pixel, scaled by 1/255
void Tensor3D::read_from_opencv_mat(const uchar* const img_ptr) {
    const int length = H * W;
    for (int i = 0; i < length; ++i) {
        const int p = 3 * i;
        data[i] = img_ptr[p] * 1.f / 255;
        data[length + i] = img_ptr[p + 1] * 1.f / 255;
        data[2 * length + i] = img_ptr[p + 2] * 1.f / 255;
    }
}

// data_format.cpp:85-105 without cv::Mat: saturate_cast<uchar>(255 * v) = round-to-nearest, clamped
std::vector<uchar> Tensor3D::opecv_mat(const int CH) const {
    ensure_host();
    const int length = H * W;
    std::vector<uchar> out((size_t)length * CH);
    auto sat = [](data_type v) {
        const long r = std::lrint(255 * v);
        return (uchar)(r < 0 ? 0 : (r > 255 ? 255 : r));
    };
    for (int i = 0; i < length; ++i)
        for (int c = 0; c < CH; ++c) out[(size_t)CH * i + c] = sat(data[i + c * length]);
    return out;
}

void Tensor3D::ensure_host() const {
    if (dev != nullptr && (data == nullptr || host_epoch != g_device_epoch)) const_cast<Tensor3D*>(this)->sync_to_host();
}
void Tensor3D::refresh_host() const { ensure_host(); }
void Tensor3D::mark_host_fresh() const { host_epoch = g_device_epoch; }

void Tensor3D::set_zero() {
    if (data) std::memset(data, 0, sizeof(data_type) * (size_t)get_length());
    if (dev) must(cnn_memset_zero(dev, sizeof(data_type) * (size_t)get_length(), architectures::stream), "cnn_memset_zero");
}

data_type Tensor3D::max() const {
    ensure_host();
    return data[argmax()];
}

// first maximum, strict '>' (data_format.cpp:37-48)
int Tensor3D::argmax() const {
    ensure_host();
    if (data == nullptr) return 0;
    const int length = get_length();
    int best = 0;
    for (int i = 1; i < length; ++i)
        if (data[i] > data[best]) best = i;
    return best;
}

data_type Tensor3D::min() const {
    ensure_host();
    return data[argmin()];
}

int Tensor3D::argmin() const {
    ensure_host();
    if (data == nullptr) return 0;
    const int length = get_length();
    int best = 0;
    for (int i = 1; i < length; ++i)
        if (data[i] < data[best]) best = i;
    return best;
}

void Tensor3D::div(const data_type times) {
    ensure_host();
    const int length = get_length();
    for (int i = 0; i < length; ++i) data[i] /= times;
}

void Tensor3D::normalize(const std::vector<data_type> mean, const std::vector<data_type> std_div) {
    if (C != 3) return;
    ensure_host();
    const int plane = H * W;
    for (int ch = 0; ch < C; ++ch)
        for (int i = 0; i < plane; ++i) data[ch * plane + i] = (data[ch * plane + i] - mean[ch]) / std_div[ch];
}

int Tensor3D::get_length() const { return C * H * W; }
std::tuple<int, int, int> Tensor3D::get_shape() const { return std::make_tuple(C, H, W); }

void Tensor3D::print_shape() const { std::cout << name << "  ==>  " << C << " x " << H << " x " << W << "\n"; }

void Tensor3D::print(const int _C) const {
    ensure_host();
    std::cout << name << "  content is : ";
    const int start = _C * H * W;
    for (int i = 0; i < H; ++i) {
        for (int j = 0; j < W; ++j)
            std::cout << std::setiosflags(std::ios::fixed) << std::setprecision(3) << data[start + i * W + j] << "   ";
        std::cout << "\n";
    }
}

std::shared_ptr<Tensor3D> Tensor3D::rot180() const {
    ensure_host();
    std::shared_ptr<Tensor3D> rot(new Tensor3D(C, H, W, name + "_rot180"));
    const int plane = H * W;
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < plane; ++i) rot->data[c * plane + i] = data[c * plane + plane - 1 - i];
    return rot;
}

std::shared_ptr<Tensor3D> Tensor3D::pad(const int padding) const {
    ensure_host();
    const int nW = W + 2 * padding, nH = H + 2 * padding;
    std::shared_ptr<Tensor3D> padded(new Tensor3D(C, nH, nW, name + "_pad"));
    std::memset(padded->data, 0, sizeof(data_type) * (size_t)C * nH * nW);
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < H; ++i)
            std::memcpy(padded->data + (size_t)c * nH * nW + (size_t)(padding + i) * nW + padding,
                        data + (size_t)c * H * W + (size_t)i * W, sizeof(data_type) * W);
    return padded;
}
