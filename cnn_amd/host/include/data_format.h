// data_format.h -- Tensor3D of the MI355X build.  Same public surface as the reference's
// cpu/include/data_format.h:10-53 (fields C,H,W,data,name; the three constructors; set_zero/max/argmax/min/argmin/
// div/normalize/get_length/get_shape/print_shape/print/rot180/pad), so model and driver code written against the
// reference compiles against this header.  Differences, all additive:
//   * a tensor may be a NON-OWNING VIEW of device memory (`dev`), which is how layers hand each other a batch: the B
//     tensors of a std::vector<tensor> are views at stride C*H*W into ONE contiguous NCHW arena owned by the layer;
//   * `data` (host) of a device view is materialised on demand by sync_to_host(); Layer::get_output() does that for
//     you (alexnet.cpp:97,105 is the only place the reference reads intermediate activations);
//   * the two OpenCV entry points (read_from_opencv_mat / opecv_mat) take / return raw 8-bit BGR buffers instead of
//     cv::Mat, because OpenCV is not part of this build (image decode is out of scope).
#ifndef CNN_AMD_DATA_FORMAT_H
#define CNN_AMD_DATA_FORMAT_H

// The reference's translation units get the C / C++ standard headers below transitively through <opencv2/core.hpp>
// (cpu/include/data_format.h:7) and rely on it: cpu/src/alexnet.cpp:37 uses assert, cpu/src/func.cpp:8,10 FLT_MAX and
// std::exp without including anything themselves.  OpenCV is not part of this build, so this header provides them.
#include <algorithm>
#include <cassert>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <memory>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

using data_type = float;
using uchar = unsigned char;

class Tensor3D {
public:
    const int C, H, W;
    data_type* data;   // host buffer (may be null for a device view until sync_to_host())
    std::string name;
    data_type* dev = nullptr;  // device view (not owned); null for a plain host tensor

    Tensor3D(const int _C, const int _H, const int _W, const std::string _name = "pipeline");
    Tensor3D(const std::tuple<int, int, int>& shape, const std::string _name = "pipeline");
    Tensor3D(const int length, const std::string _name = "pipeline");  // length x 1 x 1
    // device view factory (no host allocation)
    static std::shared_ptr<Tensor3D> device_view(int C, int H, int W, data_type* dev_ptr, const std::string& name);

    void read_from_opencv_mat(const uchar* const img_ptr);  // interleaved 8-bit BGR, HxWx3 -> planar /255
    std::vector<uchar> opecv_mat(const int CH = 3) const;   // planar -> interleaved 8-bit, saturating
    void set_zero();
    data_type max() const;
    int argmax() const;
    data_type min() const;
    int argmin() const;
    void div(const data_type times);
    void normalize(const std::vector<data_type> mean = {0.406, 0.456, 0.485},
                   const std::vector<data_type> std_div = {0.225, 0.224, 0.229});
    int get_length() const;
    std::tuple<int, int, int> get_shape() const;
    void print_shape() const;
    void print(const int _C = 0) const;
    std::shared_ptr<Tensor3D> rot180() const;
    std::shared_ptr<Tensor3D> pad(const int padding = 1) const;
    ~Tensor3D() noexcept;

    // ---- device residency (additions) ----
    bool on_device() const { return dev != nullptr; }
    void sync_to_host();          // D2H of the view into `data` (allocated on first use); blocks until done
    void sync_to_device() const;  // H2D of `data` into the view; enqueued on architectures::stream
    // A device view's host copy is only as fresh as the last D2H.  Every Layer::forward / backward call (anything that enqueues
    // device work through the host classes) bumps a global epoch; the host-side readers below (max / argmax / print / div / ...,
    // and func.h's softmax) copy back again when the view's copy is older than that -- so reading a layer's output after the
    // NEXT pass returns that pass' values, like the reference's plain host tensors do.
    static void device_work_enqueued();
    void mark_host_fresh() const; // the caller just filled `data` with the view's current contents itself
    void refresh_host() const;    // = sync_to_host() when the host copy is stale (or missing); no-op for a plain host tensor

private:
    struct ViewTag {};
    Tensor3D(ViewTag, int _C, int _H, int _W, data_type* dev_ptr, const std::string& _name);
    // the host-side helpers below read `data`: a device view that has not been copied back yet is synced first (a Layer's
    // output tensors are such views: softmax() on them, max(), print() ... just work)
    void ensure_host() const;
    mutable unsigned long long host_epoch = 0;  // epoch of the last D2H (0 = never)
};
using tensor = std::shared_ptr<Tensor3D>;

#endif  // CNN_AMD_DATA_FORMAT_H
