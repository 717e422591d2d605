// architectures.h -- the reference's layer API (cpu/include/architectures.h:12-138, 196-215) on MI355X.
// Constructors, virtuals and globals keep the reference's names, argument order and defaults; every forward /
// backward / update body is a call into libcnn_amd.so (include/cnn_amd.h).  Errors from the C ABI abort, like the
// reference's asserts (it has no error channel).
#ifndef CNN_AMD_ARCHITECTURES_H
#define CNN_AMD_ARCHITECTURES_H

#include <filesystem>
#include <fstream>
#include <list>
#include <random>

#include "data_format.h"

extern "C" {
#include "cnn_amd.h"
}

namespace architectures {

extern data_type random_times;  // init scale divisor (architectures.cpp:6)
extern bool no_grad;            // skip everything only backward needs (architectures.cpp:8)
extern void* stream;            // hipStream_t all layers enqueue on (addition; default stream when null)
// addition: let the AlexNet container run Conv2D+ReLU forward and MaxPool2D+ReLU backward as one kernel each
// (bit-identical results, every layer's output / delta tensor is still produced); false = one kernel per layer call
extern bool fuse_layers;
// addition (default ON since round 3): inside a container, Conv2D -> ReLU -> MaxPool2D(2,2) blocks run as ONE kernel and their
// backward passes work from the pooled domain; convolutions followed by a ReLU write only the ReLU output where the kernel
// supports it.  The tensors that are not written in such a pass (the block's Conv2D / ReLU outputs, pre-activations) stay
// OBSERVABLE: Layer::get_output() re-computes them on demand -- one forward launch from the recorded input with the parameters
// the last forward pass used (the container keeps a snapshot across its SGD step) -- bit-identical to the unfused pass, so
// the reference's contract (alexnet.cpp:97,105: every layer's output readable after forward) holds.  false = every tensor is
// written by the pass itself.
extern bool fuse_pool_block;
// addition: LinearLayer::forward normally copies its (tiny) output to the host right away, because the reference's callers
// read `output[b]->data` directly (softmax, func.cpp:24-28; argmax, cnn.cpp:92) -- one blocking D2H per forward pass.  While
// this flag is set the copy is skipped (Layer::get_output() / Tensor3D::sync_to_host() still materialise it on demand):
// Sequential::train_step sets it around its forward pass, so a train step never waits for the device.
extern bool lazy_host_sync;
// addition (round 4, default TRUE = the reference's behaviour): Conv2D::backward of the network's FIRST layer computes the delta with
// respect to the input image (conv2d.cpp:168-199) although nothing consumes it (alexnet.cpp:53-55 drops the last backward's return value).
// false: Sequential::train_step does not compute it for a pool-fused first block -- what every training framework does for an input that
// needs no gradient; the layer's delta tensors are then NOT valid after the step (the accessors of the C handle API report that).  Grad-CAM
// and the plain forward / backward / update_gradients sequence are unaffected.  bench.py reports this mode as an extra, labelled leg only.
extern bool input_gradient;

class WithoutGrad final {
public:
    explicit WithoutGrad() { architectures::no_grad = true; }
    ~WithoutGrad() noexcept { architectures::no_grad = false; }
};

// One contiguous NCHW device buffer + the B Tensor3D views over it that the std::vector<tensor> API needs.
class BatchBuffer {
public:
    BatchBuffer() = default;
    ~BatchBuffer();
    BatchBuffer(const BatchBuffer&) = delete;
    BatchBuffer& operator=(const BatchBuffer&) = delete;
    void allocate(int B, int C, int H, int W, const std::string& name);
    bool empty() const { return views.empty(); }
    data_type* base = nullptr;
    size_t sample_len = 0;
    std::vector<tensor> views;
};

// returns the device base pointer of a batch: zero-copy when the tensors are consecutive views of one arena,
// otherwise uploads / gathers them into `staging` (host tensors, or views from different arenas)
const data_type* batch_device_pointer(const std::vector<tensor>& batch, BatchBuffer& staging, const std::string& who);
data_type* batch_device_pointer_mut(std::vector<tensor>& batch, BatchBuffer& staging, const std::string& who);

class Layer {
public:
    const std::string name;
    std::vector<tensor> output;

public:
    Layer(std::string& _name) : name(std::move(_name)) {}
    virtual ~Layer() = default;
    virtual std::vector<tensor> forward(const std::vector<tensor>& input) = 0;
    virtual std::vector<tensor> backward(std::vector<tensor>& delta) = 0;
    virtual void update_gradients(const data_type learning_rate = 1e-4) {}
    virtual void save_weights(std::ofstream& writer) const {}
    virtual void load_weights(std::ifstream& reader) {}
    // host-readable copy of the last forward's output (alexnet.cpp:97,105 contract): syncs device -> host
    virtual std::vector<tensor> get_output() const;
    // ---- additions: flat parameter arena support (one SGD kernel, one all-reduce per step) ----
    virtual size_t param_count() const { return 0; }
    virtual void bind_arena(data_type* params_dev, data_type* grads_dev) {}
};

class ReLU;
class MaxPool2D;
class BatchNorm2D;

class Conv2D : public Layer {
private:
    ReLU* fused_relu = nullptr;  // the ReLU layer right behind this convolution (set by the container), or null
    MaxPool2D* fused_pool = nullptr;  // ... and the MaxPool2D(2,2) behind that ReLU: Conv -> ReLU -> MaxPool in one kernel
    ReLU* relu_below = nullptr;       // the ReLU layer whose output is this layer's input: its backward pass is fused in
    bool publish_backward = false;
    MaxPool2D* pool_below = nullptr;  // the MaxPool2D whose output is this layer's input, when that pool closes a fusable
                                      // Conv2D -> ReLU -> MaxPool2D(2,2) block (fuse_pool_block): see backward()
    bool pool_fused_pass = false;     // this pass' forward went through the pooled kernel: backward receives d(pool output)
    void* prep_fwd = nullptr;    // prepared filters (forward / data gradient layouts)
    void* prep_dgrad = nullptr;
    void* prep_dgrad_alt = nullptr;   // second data-gradient image: a DEFERRED data gradient keeps reading the filters of its own step
    // ---- re-materialisation of an output tensor the pass did not write (fuse_pool_block) ----
    mutable bool out_valid = true;    // out_buf holds the last forward's output
    bool recompute_lost = false;      // see params_of_last_forward_lost()
    bool delta_valid = true;          // see delta_computed()
    const data_type* last_x = nullptr;  // device pointer of the last forward's input (recorded even under no_grad)
    int last_B = 0;
    const data_type* snapshot = nullptr;   // the container's copy of this layer's parameters BEFORE its latest SGD step ...
    const bool* snapshot_active = nullptr;  // ... and whether that step came after the last forward pass
    bool prepared_active = false;
    void* prep_event = nullptr;  // see wait_before_forward
    const int in_channels, out_channels, kernel_size, stride;
    const int params_for_one_kernel;
    const int padding;  // extension: the reference has no padding (conv2d.cpp:41-42 keeps it at 0)
    std::default_random_engine seed;
    // device state: weights [Co][Ci][k][k] then bias [Co], same order as the checkpoint (conv2d.cpp:220-226)
    data_type* params = nullptr;
    data_type* grads = nullptr;
    bool owns_params = true;
    bool grads_ready = false;
    BatchBuffer out_buf, delta_buf, in_stage, delta_stage;
    const data_type* saved_input = nullptr;  // device pointer of the last forward's input (the reference keeps __input)
    std::vector<tensor> saved_input_tensors;  // keeps host-uploaded inputs alive like conv2d.cpp:62
    int in_H = 0, in_W = 0, batch = 0;
    void* workspace = nullptr;
    size_t workspace_bytes = 0;
    void ensure_workspace(int B, int H, int W);
    // data parallelism (Sequential::set_comm): rank 0 measures the implicit-GEMM tile of this geometry, every replica pins ITS choice
    void* comm = nullptr;
    int comm_world = 1, comm_rank = 0;
    void tune_geometry(const cnn_conv2d_desc& d0);
    data_type* w_dev() const { return params; }
    data_type* b_dev() const { return params + (size_t)out_channels * params_for_one_kernel; }

public:
    // (the trailing _padding argument is an extension for VGG / ResNet-shaped stacks; default = the reference's 0)
    Conv2D(std::string _name, const int _in_channels = 3, const int _out_channels = 16, const int _kernel_size = 3,
           const int _stride = 2, const int _padding = 0);
    ~Conv2D() override;
    std::vector<tensor> forward(const std::vector<tensor>& input) override;
    std::vector<tensor> backward(std::vector<tensor>& delta) override;
    void update_gradients(const data_type learning_rate = 1e-4) override;
    void save_weights(std::ofstream& writer) const override;
    void load_weights(std::ifstream& reader) override;
    int get_params_num() const;
    void set_fused_relu(ReLU* relu) { fused_relu = relu; }  // addition (see architectures::fuse_layers)
    void set_comm(void* rccl_comm, int world, int rank) { comm = rccl_comm; comm_world = world; comm_rank = rank; }
    void set_fused_pool(MaxPool2D* pool) { fused_pool = pool; }
    void set_relu_below(ReLU* relu) { relu_below = relu; }
    void set_pool_below(MaxPool2D* pool) { pool_below = pool; }
    // addition: the layer that consumes this layer's data gradient next is a convolution (only fused-away layers in between):
    // the data-gradient kernel then publishes its completion (cnn_amd_publish_next_kernel) for that layer's weight-gradient fork
    void set_publish_backward(bool on) { publish_backward = on; }
    // additions: filter re-layout hoisted out of forward / backward (cnn_conv2d_prepare_filters); the container prepares
    // all layers with one call after every parameter change and switches the layers to the *_prepared entry points
    bool shape_known() const { return batch > 0; }
    cnn_conv2d_desc current_desc() const { return cnn_conv2d_desc{batch, in_channels, in_H, in_W, out_channels, kernel_size, stride, padding, 0}; }
    const data_type* filters_dev() const { return params; }
    const data_type* bias_dev() const { return b_dev(); }
    void prepared_buffers(void** fwd, void** dgrad);  // allocated on first use
    void set_prepared(bool on) { prepared_active = on; }
    // the prepared images of this layer (and of every later one) are written on another stream: the next forward call waits
    void wait_before_forward(void* event) { prep_event = event; }
    size_t param_count() const override { return (size_t)get_params_num(); }
    void bind_arena(data_type* params_dev, data_type* grads_dev) override;
    // ---- additions for fuse_pool_block ----
    // alexnet.cpp:97,105: the output of the last forward, also when that pass fused it away (see architectures::fuse_pool_block)
    std::vector<tensor> get_output() const override;
    void materialize() const;  // writes the missing output tensor(s) of the last forward pass now (no-op when present)
    void set_param_snapshot(const data_type* snap, const bool* active) { snapshot = snap; snapshot_active = active; }
    // the parameters the last forward pass used are gone (written from outside, or stepped twice without a forward pass in between):
    // an output that pass did not write can no longer be re-computed -- get_output() of it then fails loudly instead of returning a
    // tensor the pass never produced
    void params_of_last_forward_lost() { recompute_lost = true; }
    // The pieces of Conv2D::backward a container schedules itself for the pool-fused FIRST block of a network (nothing consumes
    // that layer's data gradient, conv2d.cpp:168-199 / alexnet.cpp:55):
    struct DeferredDgrad {   // everything the data gradient of one pass needs, valid until that pass' buffers are rewritten
        const void* prepared = nullptr;
        const data_type* dpool = nullptr;
        const int* mask = nullptr;
        const data_type* pooled = nullptr;
        int B = 0;
        int flags = 0;  // cnn_conv2d_desc.flags of the forward call that wrote the mask
        bool valid = false;
    };
    bool pool_fused_pending() const { return pool_fused_pass; }
    // will forward() of a batch of B samples run the Conv2D -> ReLU -> MaxPool2D(2,2) kernel (the pool then writes the set of
    // buffers a deferred data gradient of the previous pass does NOT read)?  Otherwise the unfused MaxPool2D::forward rewrites set 0.
    bool next_pass_pool_fused(int B) const;
    // cnn_conv2d_desc.flags of the last pool-fused forward pass (CNN_CONV2D_POOL_MASK_PACKED where the library supports it: the
    // block's three kernels then move one byte per pooling window instead of an int32 index -- 111 MB less per step of the reference
    // net); the pass' backward calls must hand the mask back with the same flags
    int pool_mask_flags = 0;
    // weight / bias gradient from the pooled domain on `stream`; with fused_sgd also this layer's SGD step (the old values go to
    // the snapshot) and its filter images for the next pass in the same launch (cnn_conv2d_backward_weight_pooled2_sgd_keep).
    // Returns what the data gradient of this pass will need (launch_deferred_dgrad).
    DeferredDgrad backward_weight_pooled(std::vector<tensor>& delta, bool fused_sgd, data_type learning_rate, data_type grad_scale);
    void prepare_own_filters();  // fwd image + the NEXT data-gradient image from the current parameters, on architectures::stream
    void launch_deferred_dgrad(const DeferredDgrad& job, void* on_stream);
    const data_type* delta_dev() const { return delta_buf.base; }  // the data gradient of the last backward pass (device)
    bool delta_computed() const { return delta_valid; }   // false: the last train_step skipped it (architectures::input_gradient = false)
    void set_delta_computed(bool on) { delta_valid = on; }
    size_t delta_floats() const { return delta_buf.sample_len * delta_buf.views.size(); }
};

class MaxPool2D : public Layer {
private:
    const int kernel_size, step, padding;
    BatchBuffer out_buf, delta_buf, in_stage, delta_stage;
    int* mask = nullptr;  // device int32 [B][C*Ho*Wo], flat indices into the sample's C*H*W (pool2d.cpp:81)
    // a container that DEFERS the data gradient of the block in front (it reads pooled + mask of ITS pass while the next forward
    // pass is already writing new ones) lets the fused forward alternate between two sets
    BatchBuffer out_buf_alt;
    int* mask_alt = nullptr;
    bool alternate = false;
    int cur_set = 0;
    int in_C = 0, in_H = 0, in_W = 0, batch = 0;
    ReLU* fused_relu_below = nullptr;  // the ReLU layer whose output is this pool's input (set by the container), or null
    BatchNorm2D* fused_bn_below = nullptr;  // ... and the BatchNorm2D in front of THAT (BatchNorm2D -> ReLU -> this pool), or null
    bool forward_done = false;       // this pass' output + mask were written by the producing Conv2D kernel
    bool backward_passthrough = false;  // ... and the delta stays in the pooled domain for that Conv2D's backward

public:
    MaxPool2D(std::string _name, const int _kernel_size = 2, const int _step = 2)
        : Layer(_name), kernel_size(_kernel_size), step(_step), padding(0) {}
    ~MaxPool2D() override;
    void set_fused_relu_below(ReLU* relu) { fused_relu_below = relu; }  // addition (see architectures::fuse_layers)
    void set_fused_bn_below(BatchNorm2D* bn) { fused_bn_below = bn; }   // addition: the three layers' backward passes as two kernels
    void own_backward() { backward_passthrough = false; }  // (a fused forward whose producer is NOT a Conv2D: backward() runs here)
    bool plain_2x2() const { return kernel_size == 2 && step == 2 && padding == 0; }
    // additions used by Conv2D when the container fused Conv2D -> ReLU -> this pool into one kernel
    bool fusable_2x2() const { return kernel_size == 2 && step == 2; }
    void fused_forward_target(int B, int C, int H, int W, bool record, data_type** pooled, int** mask_out);  // arms forward_done
    const data_type* pooled_dev() const { return cur_set ? out_buf_alt.base : out_buf.base; }
    const int* mask_dev() const { return cur_set ? mask_alt : mask; }
    void enable_alternate_sets() { alternate = true; }
    // pool-fused pass: the delta of this pool's output passes through untouched (the block's Conv2D consumes it).  The block's
    // ReLU::backward -- in the pooled domain: d(pool_out) masked by (pool_out <= 0), because at an argmax position the ReLU
    // output equals the pooled value and everywhere else the delta is 0 either way -- is carried by the mask itself: the fused
    // forward kernel sets bit 31 of a window's mask entry when its pooled value is <= 0, and the pooled-domain gradient kernels
    // then never match that window.
    bool passthrough_armed() const { return backward_passthrough; }
    std::vector<tensor> forward(const std::vector<tensor>& input) override;
    std::vector<tensor> backward(std::vector<tensor>& delta) override;
};

class ReLU : public Layer {
private:
    BatchBuffer out_buf, in_stage, delta_stage;
    bool forward_done = false;   // this pass' output was already written by the producing Conv2D kernel
    bool backward_done = false;  // this pass' delta was already masked by the consuming MaxPool2D kernel
    bool out_valid = true;       // false: the pool-fused pass did not write this layer's output (get_output re-computes it)
    const Conv2D* producer = nullptr;  // the convolution whose kernel writes this layer's output when fused
    const BatchNorm2D* bn_producer = nullptr;  // ... or the normalisation (BatchNorm2D -> ReLU -> MaxPool2D in one apply pass)

public:
    ReLU(std::string _name) : Layer(_name) {}
    std::vector<tensor> get_output() const override;
    bool output_valid() const { return out_valid; }
    void set_producer(const Conv2D* conv) { producer = conv; }
    void set_bn_producer(const BatchNorm2D* bn) { bn_producer = bn; }
    data_type* rematerialize_target() { out_valid = true; return out_buf.base; }
    data_type* rematerialize_target_const() const { const_cast<ReLU*>(this)->out_valid = true; return out_buf.base; }
    // additions used by Conv2D / MaxPool2D when the container fused this layer into their kernels
    data_type* fused_forward_target(int B, int C, int H, int W);  // output arena (allocated on first use); arms forward_done
    void fused_backward_done() { backward_done = true; }
    void fused_forward_skipped(int B, int C, int H, int W);  // pool-fused pass: the output is never materialised
    std::vector<tensor> forward(const std::vector<tensor>& input) override;
    std::vector<tensor> backward(std::vector<tensor>& delta) override;
};

class LinearLayer : public Layer {
private:
    const int in_channels, out_channels;
    data_type* params = nullptr;  // W [in][out] then bias [out] (linear.cpp:105-108)
    data_type* grads = nullptr;
    bool owns_params = true;
    bool grads_ready = false;
    std::tuple<int, int, int> delta_shape;
    BatchBuffer out_buf, delta_buf, in_stage, delta_stage;
    const data_type* saved_input = nullptr;
    std::vector<tensor> saved_input_tensors;
    int batch = 0;
    ReLU* relu_below = nullptr;  // the ReLU layer whose output is this layer's input: its backward pass is fused in
    bool publish_backward = false;
    bool head_dx_done = false;   // this pass' data gradient was written by the loss-head kernel (forward_loss_head(with_dx))
    void* ev_head = nullptr;
    void* own_stream = nullptr;  // the weight / bias gradient of such a pass runs here, beside everything else ...
    void* ev_wb = nullptr;
    bool wb_pending = false;     // ... until join_pending() orders it into a stream

public:
    void set_relu_below(ReLU* relu) { relu_below = relu; }  // addition (see architectures::fuse_layers)
    void set_publish_backward(bool on) { publish_backward = on; }  // (see Conv2D::set_publish_backward)
    // the weight / bias gradient of a loss-head pass may still be running on the layer's own stream: `on_stream` waits for it
    // (containers call this before they read the gradient arena; update_gradients() of a stand-alone layer does it itself)
    void join_pending(void* on_stream);
    // addition: forward + softmax + cross-entropy delta in ONE kernel (cnn_linear_forward_softmax_xent, out_channels <= 8):
    // labels_dev int32 [B]; delta_dev [B][out] receives p - onehot (func.cpp:56-73), loss_terms_dev [B] log p[label]
    bool loss_head_supported() const { return out_channels <= 8; }
    // with_dx: the same kernel also writes this layer's data gradient (cnn_linear_forward_softmax_xent_dx: a sample's row needs only
    // that sample's delta); backward() then only launches the weight / bias gradient, on the library's side stream
    std::vector<tensor> forward_loss_head(const std::vector<tensor>& input, const int* labels_dev, data_type* probs_dev,
                                          data_type* delta_dev, data_type* loss_terms_dev, bool with_dx = false);
    int out_features() const { return out_channels; }
    int in_features() const { return in_channels; }
    LinearLayer(std::string _name, const int _in_channels, const int _out_channels);
    ~LinearLayer() override;
    std::vector<tensor> forward(const std::vector<tensor>& input) override;
    std::vector<tensor> backward(std::vector<tensor>& delta) override;
    void update_gradients(const data_type learning_rate = 1e-4) override;
    void save_weights(std::ofstream& writer) const override;
    void load_weights(std::ifstream& reader) override;
    size_t param_count() const override { return (size_t)in_channels * out_channels + out_channels; }
    void bind_arena(data_type* params_dev, data_type* grads_dev) override;
};

// architectures.h:143-173 / batchnorm2d.cpp.  Device state: gamma [C], beta [C], moving_mean [C], moving_var [C] in ONE
// block in checkpoint order (batchnorm2d.cpp:168-173); the gradient block has the same shape, its moving_* half stays 0
// so that the flat-arena SGD kernel leaves the statistics untouched.  The reference's normed_input buffer is not kept
// (backward recomputes it from the recorded input and the saved batch statistics).
class BatchNorm2D : public Layer {
private:
    const int out_channels;
    const data_type eps;
    const data_type momentum;
    data_type* params = nullptr;
    data_type* grads = nullptr;
    bool owns_params = true;
    bool grads_ready = false;
    bool backward_done_by_pool = false;  // this pass' backward() already ran inside the pool's (see backward_pooled)
    data_type* saved_stats = nullptr;  // batch mean [C] then batch variance [C] (buffer_mean / buffer_var)
    BatchBuffer out_buf, in_stage, delta_stage;
    const data_type* saved_input = nullptr;
    std::vector<tensor> saved_input_tensors;
    int in_H = 0, in_W = 0, batch = 0;
    void* workspace = nullptr;
    size_t workspace_bytes = 0;
    void* comm = nullptr;        // RCCL communicator: statistics over the GLOBAL batch (sync-BN), see Sequential::set_comm
    int comm_world = 1;
    data_type* sync_sums = nullptr;  // [C] + [C] + [C][4] all-reduce operands
    ReLU* fused_relu = nullptr;      // the ReLU layer right behind this one (set by the container): its output comes from the apply pass
    MaxPool2D* fused_pool = nullptr; // ... and the MaxPool2D behind THAT: pooled in the same pass
    // ---- fuse_pool_block: a training pass with a fused ReLU behind this layer writes ONLY the ReLU output (round 4); get_output() of
    // this layer re-computes the normalised tensor from the recorded input, the saved batch statistics and the gamma / beta of that pass
    mutable bool out_valid = true;
    bool recompute_lost = false;
    int last_B = 0;
    const data_type* snapshot = nullptr;    // the container's copy of this layer's parameters BEFORE its latest SGD step ...
    const bool* snapshot_active = nullptr;  // ... and whether that step came after the last forward pass

public:
    void set_comm(void* rccl_comm, int world) { comm = rccl_comm; comm_world = world; }
    void set_fused_relu(ReLU* relu) { fused_relu = relu; }
    void set_fused_pool(MaxPool2D* pool) { fused_pool = pool; }  // addition: BatchNorm2D -> ReLU -> MaxPool2D(2, 2) in one apply pass
    void materialize_relu() const;  // the ReLU output of a pass that wrote only the pooled tensor (ReLU::get_output)
    // additions used by the MaxPool2D behind this layer's ReLU (BatchNorm2D -> ReLU -> MaxPool2D(2, 2)): its backward pass runs this
    // layer's from the pooled domain (cnn_batchnorm2d_backward_pooled: bit-identical) and arms the pass-through of backward()
    bool backward_pooled_possible(int B) const;
    void backward_pooled(const data_type* dpool, const int* mask, const data_type* pooled, data_type* dx, int B);
    std::vector<tensor> get_output() const override;
    void materialize() const;
    void set_param_snapshot(const data_type* snap, const bool* active) { snapshot = snap; snapshot_active = active; }
    void params_of_last_forward_lost() { recompute_lost = true; }
    BatchNorm2D(std::string _name, const int _out_channels, const data_type _eps = 1e-5, const data_type _momentum = 0.1);
    ~BatchNorm2D() override;
    std::vector<tensor> forward(const std::vector<tensor>& input) override;
    std::vector<tensor> backward(std::vector<tensor>& delta) override;
    void update_gradients(const data_type learning_rate = 1e-4) override;
    void save_weights(std::ofstream& writer) const override;
    void load_weights(std::ifstream& reader) override;
    size_t param_count() const override { return (size_t)4 * out_channels; }
    void bind_arena(data_type* params_dev, data_type* grads_dev) override;
};

// architectures.h:177-191 / dropout.cpp: CHANNEL dropout.  The reference keeps it out of its network (alexnet.cpp:28, "poor at
// test time", README.md:16) but ships the layer; same constructor, generator (seed 1314) and bookkeeping.  Training zeroes the
// first int(p * C) channels of every sample (dropout.cpp:34-41 tests the channel index, the shuffled sequence only fills the
// mask), no_grad scales by 1 - p; backward zeroes the same channels of the delta in place and hands it back.
class Dropout : public Layer {
private:
    data_type p;
    int selected_num = 0;
    std::vector<int> sequence;
    std::default_random_engine drop;
    std::vector<int> mask;
    BatchBuffer out_buf, in_stage, delta_stage;
    int in_C = 0, in_H = 0, in_W = 0;

public:
    Dropout(std::string _name, const data_type _p = 0.5) : Layer(_name), p(_p), drop(1314) {}
    std::vector<tensor> forward(const std::vector<tensor>& input) override;
    std::vector<tensor> backward(std::vector<tensor>& delta) override;
};

// The reference's model container is a strictly sequential std::list<std::shared_ptr<Layer>> (architectures.h:200) that
// forward walks front to back (alexnet.cpp:41-42), backward back to front (:53-55), update / save / load front to back
// (:63-64, :73-74, :86-87).  Sequential is that container for ANY layer list (the reference hard-wires one list in
// AlexNet's constructor); AlexNet below is the reference's list on top of it.  Additions over the reference's container:
//   * one flat parameter / gradient arena in checkpoint order (= layer order), so the SGD step is one kernel, a .model
//     file one copy and the data-parallel exchange one (or a few bucketed) all-reduce(s);
//   * fusion wiring between neighbouring layers (architectures::fuse_layers / fuse_pool_block);
//   * filter preparation hoisted out of the per-layer calls (cnn_conv2d_prepare_filters), redone after every parameter change;
//   * an optional RCCL communicator (set_comm): the batch is then sharded over `world` replicas, BatchNorm2D layers
//     normalise over the GLOBAL batch (sync-BN) and update_gradients() sums the gradient arena over the replicas first.
class Sequential {
public:
    bool print_info = false;

protected:
    std::list<std::shared_ptr<Layer> > layers_sequence;
    data_type* param_arena = nullptr;  // every layer's parameters in checkpoint order
    data_type* grad_arena = nullptr;
    size_t n_params = 0;
    bool owns_arena = false;
    bool finalized = false;
    bool filters_prepared = false;  // the layers' prepared filters match the current parameters
    void* comm = nullptr;           // RCCL communicator (cnn_comm_*), not owned
    int comm_world = 1;
    void* bn_comm = nullptr;        // a second communicator over the same ranks for BatchNorm2D's sync-BN reductions (owned; cnn_comm_split)
    void* comm_stream = nullptr;    // the exchange runs on its own stream, gated by events
    void* ev_grads = nullptr;
    void* ev_prep = nullptr;   // filter images of layers 2.. prepared on the library's side stream (prepare_filters)
    void* ev_prep_fork = nullptr;
    void* ev_comm = nullptr;
    bool grads_reduced = false;     // this step's gradient arena has been summed over the replicas
    std::vector<size_t> layer_offsets;  // arena offset of every layer's parameter block (finalize)
    // ---- fuse_pool_block support (round 3) ----
    data_type* param_prev = nullptr;  // the parameters as they were before the latest SGD step (Conv2D::materialize)
    bool params_stepped = false;      // ... which happened after the last forward pass
    // the pool-fused FIRST block of the network (Conv2D -> ReLU -> MaxPool2D(2,2) at the front), if there is one: train_step
    // defers its data gradient into the next forward pass and runs the rest of the step's tail underneath its weight gradient
    Conv2D* block_conv = nullptr;
    MaxPool2D* block_pool = nullptr;
    Layer* behind_block = nullptr;     // the layer that consumes the pool's output (its backward rewrites d(pool output))
    Layer* release_after = nullptr;    // the deferred data gradient starts behind this layer's forward kernel
    void* defer_stream = nullptr;
    void* ev_defer_done = nullptr;
    void* ev_tail = nullptr;
    void* ev_side_tail = nullptr;
    bool side_tail_pending = false;  // the side stream still runs the previous step's tail (fused_tail, TAIL_BEHIND_BLOCK)
    void prepare_later_filters(void* on_stream);
    Conv2D::DeferredDgrad pending_dgrad;
    bool defer_in_flight = false;
    bool fused_tail(std::vector<tensor>& delta, const data_type learning_rate);  // false: not applicable to this pass
    // big arenas (the VGG / ResNet-shaped stacks: 37 - 45 MB) are exchanged in BUCKETS while the backward pass is still running:
    // layers are walked back to front, so finished gradients form a growing suffix of the arena; every >= bucket_floats of it
    // go out on the communication stream behind an event.  Small arenas (the reference net: 445 KB, latency-bound) stay one call.
    size_t bucket_floats = (size_t)2 << 20;
    void flush_bucket(size_t lo, size_t hi);
    bool exchange_active() const;
    void wire();
    void bind(data_type* p, data_type* g);
    void prepare_filters();

public:
    Sequential() = default;
    virtual ~Sequential();
    Sequential(const Sequential&) = delete;
    Sequential& operator=(const Sequential&) = delete;
    // append a layer (takes ownership); all add() calls come before finalize()
    Sequential& add(Layer* layer);
    // wires the fusions and moves every layer's parameters into ONE arena (owned, or caller-provided device buffers of
    // num_params() floats each, e.g. so that a framework can own / all-reduce them)
    void finalize();
    void finalize(data_type* params_dev, data_type* grads_dev);
    std::vector<tensor> forward(const std::vector<tensor>& input);
    void backward(std::vector<tensor>& delta_start);
    void update_gradients(const data_type learning_rate = 1e-4);
    // one SGD kernel over the flat arena; grad_scale folds the 1/G of a data-parallel all-reduce(sum) done by the caller
    void update_gradients(const data_type learning_rate, const data_type grad_scale);
    void save_weights(const std::filesystem::path& save_path) const;
    void load_weights(const std::filesystem::path& checkpoint_path);
    // alexnet.cpp:95-142 (cv::Mat -> the H*W 8-bit pixels of that matrix, row-major; see cnn_grad_cam).  Like the reference it
    // assumes a forward pass with gradients enabled has just run, walks backward() down to the named layer and reads that
    // layer's get_output(); `cam_out` (optional) receives the normalised [B][H][W] map the picture is taken from.
    std::vector<uchar> grad_cam(const std::string& layer_name, std::vector<data_type>* cam_out = nullptr) const;
    // additions
    size_t num_params() const;
    data_type* params_device() const { return param_arena; }
    data_type* grads_device() const { return grad_arena; }
    // MUST be called after writing the parameter arena from outside (memcpy, collective, ...): the convolutions keep
    // re-arranged copies of their filters between update_gradients() calls
    void parameters_changed();
    const std::list<std::shared_ptr<Layer> >& layers() const { return layers_sequence; }

protected:
    void invalidate_filter_images();  // (the part of parameters_changed() the container's own SGD step needs too)

public:
    // data parallelism over `world` replicas of this container (one per GPU), comm from cnn_comm_init_rank / _init_all:
    // backward() then leaves LOCAL gradients in the arena, update_gradients(lr) all-reduces them (RCCL, fp32 sum, on a
    // communication stream) and applies lr * (1/world) * sum.  Pass comm = nullptr to switch it off again.
    void set_comm(void* rccl_comm, int world);
    // sums the gradient arena over the replicas now (idempotent per backward pass); update_gradients(lr) calls it
    void allreduce_gradients();
    // One iteration of the reference's training loop (cnn.cpp:79-90: forward, softmax, cross_entroy_backward, backward,
    // update_gradients) that never leaves the device: the loss glue of func.cpp:16-73 runs as a kernel (fused into the last
    // LinearLayer's forward kernel when it has <= 8 outputs), nothing is copied back, the host never blocks.  labels_dev:
    // int32 [B] on the device.  The last layer must be a LinearLayer (the logits).  last_loss() fetches the loss of the most
    // recent step (-(1/B) * sum log p[label], func.cpp:67,71) -- the only call here that synchronises.
    void train_step(const std::vector<tensor>& input, const int* labels_dev, const data_type learning_rate);
    data_type last_loss();
    const data_type* last_probs_device() const { return loss_probs.base; }
    // train_step defers the data gradient of a pool-fused first block (no consumer: alexnet.cpp:55 discards it) into the next
    // forward pass.  flush_deferred() launches a still-pending one and orders it before later work on architectures::stream; every
    // other entry point of the container does that itself, a caller only needs it before timing ends or before it reads that
    // layer's delta tensors through raw device pointers.
    void flush_deferred();
private:
    BatchBuffer loss_probs, loss_delta, logits_stage;   // [B][classes] each
    data_type* loss_terms = nullptr;      // [B] log p[label] (fused head) ...
    data_type* loss_sum = nullptr;        // ... or the ordered sum (unfused head); [1]
    bool loss_in_terms = false;
    int loss_batch = 0, loss_last_B = 0;
};

// The reference's fixed network (alexnet.cpp:10-33).  forward / backward / update_gradients / save_weights / load_weights are
// declared on AlexNet itself like in the reference (architectures.h:203-213), so cpu/src/alexnet.cpp -- which DEFINES them --
// compiles against this header unchanged up to its OpenCV-typed grad_cam (:95, out of scope: SURVEY.md section 8f).
class AlexNet : public Sequential {
public:
    AlexNet(const int num_classes = 3, const bool batch_norm = false);
    // addition: adopt caller-provided device arenas (e.g. torch tensors) so the gradient arena can be all-reduced
    AlexNet(const int num_classes, data_type* params_dev, data_type* grads_dev, const bool batch_norm = false);
    std::vector<tensor> forward(const std::vector<tensor>& input);
    void backward(std::vector<tensor>& delta_start);
    void update_gradients(const data_type learning_rate = 1e-4);
    void update_gradients(const data_type learning_rate, const data_type grad_scale);
    void save_weights(const std::filesystem::path& save_path) const;
    void load_weights(const std::filesystem::path& checkpoint_path);
};

// BASELINE.json configs[3] / [4] as layer lists of the reference's own layer types (mirrored by cnn_amd/stacks.py):
// VGG-11-shaped: eight 3x3 stride-1 pad-1 convolutions 3->64->128->256->256->512->512->512->512, ReLU after each,
// MaxPool2D(2,2) after convolutions 1, 2, 4, 6, 8, LinearLayer(512*7*7 -> classes) for 224x224 inputs.
void build_vgg11(Sequential& net, const int num_classes = 3, const bool batch_norm = false);
// ResNet-18-shaped: the convolution shapes of ResNet-18 as a strictly sequential list (no residual adds: alexnet.cpp:41):
// 7x7 s2 p3 stem, MaxPool2D(2,2), four stages of four convolutions (3x3 s1 p1 inside a stage; stage entries 3x3 s2 p1,
// 1x1 s2, 3x3 s2 p1), BatchNorm2D + ReLU after every convolution, LinearLayer(512*7*7 -> classes).
void build_resnet18(Sequential& net, const int num_classes = 3, const bool batch_norm = true);

}  // namespace architectures

#endif  // CNN_AMD_ARCHITECTURES_H
