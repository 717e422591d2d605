// layers.cpp -- Conv2D / MaxPool2D / ReLU / LinearLayer with the reference's interface
// (cpu/include/architectures.h:49-138); every body is a call into the C ABI of libcnn_amd.so.
#include <algorithm>
#include <cassert>
#include <cstdio>
#include <stdexcept>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "architectures.h"
#include "host_util.h"

// A batch SMALLER than the first call's (the layers' buffers are sized by that one: conv2d.cpp:47-52) is processed as what it is: forward()
// and backward() hand on as many tensors as they were given, not the whole buffer's views -- with the full vector a layer below would
// also walk the stale samples behind the batch (round 6: found by tests/sweeps/fuzz_nets.py; the reference itself returns its whole
// `output` / `delta_output` members, conv2d.cpp:93 / :201, and cannot run a smaller batch at all: its loss glue indexes the labels by them)
static std::vector<tensor> first_n(const std::vector<tensor>& v, int B) {
    if ((size_t)B >= v.size()) return v;
    return std::vector<tensor>(v.begin(), v.begin() + B);
}


using namespace architectures;
using cnn_amd_host::dev_alloc;
using cnn_amd_host::must;

data_type architectures::random_times = 10.f;  // architectures.cpp:6
bool architectures::no_grad = false;           // architectures.cpp:8
void* architectures::stream = nullptr;
bool architectures::fuse_layers = true;
bool architectures::fuse_pool_block = true;
bool architectures::lazy_host_sync = false;
bool architectures::input_gradient = true;

// ---------------------------------------------------------------------------------------------------------------
BatchBuffer::~BatchBuffer() {
    views.clear();
    if (base) cnn_device_free(base);
}

void BatchBuffer::allocate(int B, int C, int H, int W, const std::string& name) {
    sample_len = (size_t)C * H * W;
    base = (data_type*)dev_alloc(sizeof(data_type) * sample_len * B);
    views.reserve(B);
    for (int b = 0; b < B; ++b)
        views.emplace_back(Tensor3D::device_view(C, H, W, base + sample_len * b, name + "_" + std::to_string(b)));
}

namespace {
bool is_contiguous_device_batch(const std::vector<tensor>& batch) {
    const size_t len = (size_t)batch[0]->get_length();
    for (size_t b = 0; b < batch.size(); ++b)
        if (!batch[b]->on_device() || batch[b]->dev != batch[0]->dev + len * b) return false;
    return true;
}
// host tensors / scattered views -> one contiguous device buffer (one H2D for an all-host batch)
data_type* stage_batch(const std::vector<tensor>& batch, BatchBuffer& staging, const std::string& who) {
    const int B = (int)batch.size();
    const size_t len = (size_t)batch[0]->get_length();
    if (staging.empty() || staging.views.size() < (size_t)B || staging.sample_len != len) {
        staging.views.clear();
        if (staging.base) cnn_device_free(staging.base);
        staging.base = nullptr;
        staging.allocate(B, batch[0]->C, batch[0]->H, batch[0]->W, who + "_staging");
    }
    bool all_host = true;
    for (const auto& t : batch) all_host = all_host && !t->on_device();
    if (all_host) {
        std::vector<data_type> packed(len * B);
        for (int b = 0; b < B; ++b) std::memcpy(packed.data() + len * b, batch[b]->data, sizeof(data_type) * len);
        must(cnn_memcpy_h2d(staging.base, packed.data(), sizeof(data_type) * len * B, architectures::stream), "cnn_memcpy_h2d");
        must(cnn_stream_synchronize(architectures::stream), "cnn_stream_synchronize");  // `packed` dies here
    } else {
        for (int b = 0; b < B; ++b) {
            data_type* dst = staging.base + len * b;
            if (batch[b]->on_device())
                must(cnn_memcpy_d2d(dst, batch[b]->dev, sizeof(data_type) * len, architectures::stream), "cnn_memcpy_d2d");
            else {
                must(cnn_memcpy_h2d(dst, batch[b]->data, sizeof(data_type) * len, architectures::stream), "cnn_memcpy_h2d");
                must(cnn_stream_synchronize(architectures::stream), "cnn_stream_synchronize");
            }
        }
    }
    return staging.base;
}
}  // namespace

const data_type* architectures::batch_device_pointer(const std::vector<tensor>& batch, BatchBuffer& staging,
                                                     const std::string& who) {
    assert(!batch.empty());
    if (is_contiguous_device_batch(batch)) return batch[0]->dev;
    return stage_batch(batch, staging, who);
}

data_type* architectures::batch_device_pointer_mut(std::vector<tensor>& batch, BatchBuffer& staging,
                                                   const std::string& who) {
    assert(!batch.empty());
    if (is_contiguous_device_batch(batch)) return batch[0]->dev;
    return stage_batch(batch, staging, who);
}

std::vector<tensor> Layer::get_output() const {
    for (const auto& t : output) t->sync_to_host();
    return output;
}

// ---------------------------------------------------------------------------------------------------------------
// Conv2D
Conv2D::Conv2D(std::string _name, const int _in_channels, const int _out_channels, const int _kernel_size,
               const int _stride, const int _padding)
    : Layer(_name), in_channels(_in_channels), out_channels(_out_channels), kernel_size(_kernel_size), stride(_stride),
      params_for_one_kernel(_in_channels * _kernel_size * _kernel_size), padding(_padding) {
    assert(_kernel_size >= 1 && _in_channels > 0 && _out_channels > 0 && _stride > 0 && _padding >= 0);
    const size_t n = param_count();
    params = (data_type*)dev_alloc(sizeof(data_type) * n);
    grads = (data_type*)dev_alloc(sizeof(data_type) * n);
    // same generator, seed and draw order as conv2d.cpp:23-30 (bias first, then the filters), stored weights-then-bias
    std::vector<data_type> host(n);
    this->seed.seed(212);
    std::normal_distribution<float> engine(0.0, 1.0);
    data_type* bias = host.data() + (size_t)out_channels * params_for_one_kernel;
    for (int o = 0; o < out_channels; ++o) bias[o] = engine(this->seed) / random_times;
    for (size_t i = 0; i < (size_t)out_channels * params_for_one_kernel; ++i) host[i] = engine(this->seed) / random_times;
    must(cnn_memcpy_h2d(params, host.data(), sizeof(data_type) * n, stream), "cnn_memcpy_h2d");
    must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
}

Conv2D::~Conv2D() {
    if (owns_params) {
        cnn_device_free(params);
        cnn_device_free(grads);
    }
    if (workspace) cnn_device_free(workspace);
    if (prep_fwd) cnn_device_free(prep_fwd);
    if (prep_dgrad) cnn_device_free(prep_dgrad);
    if (prep_dgrad_alt) cnn_device_free(prep_dgrad_alt);
}

void Conv2D::prepared_buffers(void** fwd, void** dgrad) {
    if (prep_fwd == nullptr) {
        const cnn_conv2d_desc d = current_desc();
        const size_t n = cnn_conv2d_prepared_bytes(&d);
        prep_fwd = dev_alloc(n);
        prep_dgrad = dev_alloc(n);
    }
    *fwd = prep_fwd;
    *dgrad = prep_dgrad;
}

void Conv2D::bind_arena(data_type* params_dev, data_type* grads_dev) {
    must(cnn_memcpy_d2d(params_dev, params, sizeof(data_type) * param_count(), stream), "cnn_memcpy_d2d");
    must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
    if (owns_params) {
        cnn_device_free(params);
        cnn_device_free(grads);
    }
    params = params_dev;
    grads = grads_dev;
    owns_params = false;
    prepared_active = false;
}

void Conv2D::ensure_workspace(int B, int H, int W) {
    cnn_conv2d_desc d{B, in_channels, H, W, out_channels, kernel_size, stride, padding, 0};
    const size_t need = cnn_conv2d_workspace_bytes(&d);
    if (need == 0) must(1, "cnn_conv2d_workspace_bytes");
    if (need > workspace_bytes) {
        if (workspace) cnn_device_free(workspace);
        workspace = dev_alloc(need);
        workspace_bytes = need;
    }
}

// cnn_conv2d_autotune for a geometry this layer sees for the first time -- with the CALLER's scratch (include/cnn_amd.h: "no allocation
// inside"), freed again right away; inside a data-parallel container only rank 0 measures and every replica pins rank 0's choice, so that
// all of them run the same kernels (the measurement is not reproducible from box to box)
void Conv2D::tune_geometry(const cnn_conv2d_desc& d0) {
    // Inside a data-parallel container every replica measures FOR ITSELF by default: no collective on the first forward pass of a layer.
    // Replicas that pin different tiles differ in the last bits of their LOCAL gradients only -- the all-reduced sums, and with them the
    // parameters, stay identical on every rank (bench.py asserts the digests).  CNN_AMD_DP_SHARE_TUNE=1 restores rank 0's measurement
    // broadcast to all replicas (same kernels everywhere); that makes the first forward() of every Conv2D a BLOCKING COLLECTIVE --
    // one host thread (or process) per rank, identical shapes and batch sizes on every rank -- and has not run on more than one rank
    // yet (VERDICT / ADVICE r4), hence opt-in.
    char text[8] = {0};
    const bool share = cnn_amd_get_option("DP_SHARE_TUNE", text, sizeof(text)) == 0 && std::atoi(text) != 0;
    const bool shared = share && comm != nullptr && comm_world > 1;
    if (!shared || comm_rank == 0) {
        const size_t need = cnn_conv2d_autotune_workspace_bytes(&d0);
        void* scratch = nullptr;
        if (need > 0 && cnn_device_alloc(&scratch, need) == CNN_AMD_OK) {  // (no room to measure: the rule-based tiles stay)
            must(cnn_conv2d_autotune_ws(&d0, scratch, need, stream), "cnn_conv2d_autotune_ws");
            must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
            cnn_device_free(scratch);
        }
    }
    if (!shared) return;
    int32_t choice[4] = {CNN_TUNE_NONE, CNN_TUNE_NONE, CNN_TUNE_NONE, CNN_TUNE_NONE};
    if (comm_rank == 0) must(cnn_conv2d_tune_export(&d0, choice), "cnn_conv2d_tune_export");
    void* dev = dev_alloc(sizeof(choice));
    must(cnn_memcpy_h2d(dev, choice, sizeof(choice), stream), "cnn_memcpy_h2d");
    must(cnn_comm_broadcast(comm, dev, sizeof(choice), 0, stream), "cnn_comm_broadcast");
    must(cnn_memcpy_d2h(choice, dev, sizeof(choice), stream), "cnn_memcpy_d2h");
    must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
    cnn_device_free(dev);
    if (comm_rank != 0) must(cnn_conv2d_tune_import(&d0, choice), "cnn_conv2d_tune_import");
}

std::vector<tensor> Conv2D::forward(const std::vector<tensor>& input) {
    Tensor3D::device_work_enqueued();  // host copies of device views made before this call are stale from here on
    const int B = (int)input.size();
    const int H = input[0]->H, W = input[0]->W;
    const int out_H = cnn_conv2d_out_dim(H, kernel_size, stride, padding);
    const int out_W = cnn_conv2d_out_dim(W, kernel_size, stride, padding);
    if (out_buf.empty()) {  // shape-static like conv2d.cpp:47-52: buffers are sized by the first batch
        out_buf.allocate(B, out_channels, out_H, out_W, name + "_output");
        output = out_buf.views;
        batch = B;
        // the shape is known now: let the library measure which tile its implicit-GEMM kernels should use for it (once per
        // geometry and process, before any filter preparation)
        cnn_conv2d_desc d0{B, in_channels, H, W, out_channels, kernel_size, stride, padding, 0};
        tune_geometry(d0);
    }
    assert(B <= batch && "batch larger than the first forward's (conv2d.cpp:47 has the same restriction)");
    // shape-static like the reference (conv2d.cpp:47-60 caches its offsets on the first call): the buffers and the prepared
    // filters were sized for the first input shape -- a different one would overrun DEVICE memory, so it is fatal here
    if ((in_H && in_H != H) || (in_W && in_W != W) || input[0]->C != in_channels) {
        std::fprintf(stderr, "cnn_amd host: %s: input %dx%dx%d after a first forward with %dx%dx%d (layers are shape-static)\n", name.c_str(),
                     input[0]->C, H, W, in_channels, in_H, in_W);
        std::abort();
    }
    in_H = H;
    in_W = W;
    ensure_workspace(B, H, W);
    if (prep_event != nullptr) {  // this layer's filter images come from the side stream (Sequential::prepare_filters)
        must(cnn_stream_wait_event(stream, prep_event), "cnn_stream_wait_event");
        prep_event = nullptr;
    }
    const data_type* x = batch_device_pointer(input, in_stage, name);
    if (!no_grad) {  // conv2d.cpp:62: keep the input for the weight gradient
        saved_input = x;
        saved_input_tensors = input;
    }
    last_x = x;  // (a pointer, not a copy: get_output() of a fused-away tensor re-computes it from here)
    last_B = B;
    out_valid = true;
    recompute_lost = false;
    cnn_conv2d_desc d{B, in_channels, H, W, out_channels, kernel_size, stride, padding, 0};
    const bool prepared = prepared_active && fuse_layers && B == batch;
    pool_fused_pass = false;
    if (prepared && fuse_pool_block && fused_relu != nullptr && fused_pool != nullptr && fused_pool->fusable_2x2() &&
        cnn_conv2d_relu_maxpool2_supported(&d)) {
        // Conv2D -> ReLU -> MaxPool2D(2,2) in one kernel: only the pool's output and mask are written; this layer's and the
        // ReLU's output tensors are NOT materialised in such a pass (their backward passes run from the pooled domain)
        data_type* pooled = nullptr;
        int* pmask = nullptr;
        d.flags = cnn_conv2d_pool_mask_packed_supported(&d) ? CNN_CONV2D_POOL_MASK_PACKED : 0;  // (the mask buffer fits either form)
        pool_mask_flags = d.flags;
        fused_pool->fused_forward_target(B, out_channels, out_H, out_W, !no_grad, &pooled, &pmask);
        fused_relu->fused_forward_skipped(B, out_channels, out_H, out_W);
        must(cnn_conv2d_relu_maxpool2_forward_prepared(&d, x, prep_fwd, pooled, pmask, stream), "cnn_conv2d_relu_maxpool2_forward_prepared");
        pool_fused_pass = !no_grad;
        out_valid = false;  // (re-computed by get_output() if anybody asks)
        return output;
    }
    if (fused_relu != nullptr && fuse_layers) {  // the ReLU behind this layer gets its output from the same kernel
        data_type* y_relu = fused_relu->fused_forward_target(B, out_channels, out_H, out_W);
        // fuse_pool_block: the pre-activation tensor is not written either where the kernel supports it -- ReLU::backward masks
        // by the ReLU's own output (relu.cpp:35-40), nothing in a train step reads it; get_output() re-computes it on demand
        const bool relu_only = prepared && fuse_pool_block && !no_grad && cnn_conv2d_relu_only_supported(&d) != 0;
        out_valid = !relu_only;
        if (prepared)
            must(cnn_conv2d_forward_prepared(&d, x, prep_fwd, b_dev(), relu_only ? nullptr : out_buf.base, y_relu, stream),
                 "cnn_conv2d_forward_prepared");
        else
            must(cnn_conv2d_forward_relu(&d, x, w_dev(), b_dev(), out_buf.base, y_relu, workspace, workspace_bytes, stream),
                 "cnn_conv2d_forward_relu");
    } else if (prepared) {
        must(cnn_conv2d_forward_prepared(&d, x, prep_fwd, b_dev(), out_buf.base, nullptr, stream), "cnn_conv2d_forward_prepared");
    } else {
        must(cnn_conv2d_forward(&d, x, w_dev(), b_dev(), out_buf.base, workspace, workspace_bytes, stream), "cnn_conv2d_forward");
    }
    return first_n(output, B);
}

bool Conv2D::next_pass_pool_fused(int B) const {
    if (!(prepared_active && fuse_layers && B == batch && fuse_pool_block && fused_relu != nullptr && fused_pool != nullptr &&
          fused_pool->fusable_2x2()))
        return false;
    cnn_conv2d_desc d{B, in_channels, in_H, in_W, out_channels, kernel_size, stride, padding, 0};
    return cnn_conv2d_relu_maxpool2_supported(&d) != 0;
}

std::vector<tensor> Conv2D::backward(std::vector<tensor>& delta) {
    Tensor3D::device_work_enqueued();  // host copies of device views made before this call are stale from here on
    const int B = (int)delta.size();
    assert(saved_input != nullptr && "backward without a recorded forward (no_grad?)");
    const data_type* dy = batch_device_pointer(delta, delta_stage, name + "_dy");
    cnn_conv2d_desc d{B, in_channels, in_H, in_W, out_channels, kernel_size, stride, padding, 0};
    // conv2d.cpp:117-199: weight/bias gradients (recomputed, not accumulated, averaged over the batch) and the data
    // gradient in one call; the library overlaps the two on an internal side stream
    if (delta_buf.empty()) delta_buf.allocate(batch, in_channels, in_H, in_W, name + "_delta");
    BatchBuffer& dbuf = delta_buf;
    const size_t need = cnn_conv2d_backward_workspace_bytes(&d);
    if (need > workspace_bytes) {
        if (workspace) cnn_device_free(workspace);
        workspace = dev_alloc(need);
        workspace_bytes = need;
    }
    if (pool_fused_pass) {
        // dy is the delta of the POOL output (the pool and the ReLU passed it through untouched)
        assert(prepared_active && fused_pool != nullptr && B == batch);
        // (pooled = NULL: the fused forward kernel marked the windows whose pooled value is <= 0 in the mask itself -- bit 31 --, so
        // the block's ReLU::backward needs no tensor of its own here)
        const data_type* pooled = nullptr;
        d.flags = pool_mask_flags;
        must(cnn_conv2d_backward_pooled2_prepared(&d, saved_input, dy, fused_pool->mask_dev(), pooled, prep_dgrad, grads,
                                                  grads + (size_t)out_channels * params_for_one_kernel, dbuf.base, (float)B,
                                                  workspace, workspace_bytes, stream, /*defer_join=*/1),
             "cnn_conv2d_backward_pooled2_prepared");
        pool_fused_pass = false;
    } else if (prepared_active && fuse_layers && B == batch) {
        // relu_below: this layer's input IS that ReLU's output, so its backward mask is applied in the data-gradient epilogue.
        // pool_below in a pool-fused pass: this layer's input is the POOL output and its delta stays in the pooled domain,
        // UNMASKED -- the block's ReLU::backward is carried by the marked pool mask (bit 31, cnn_conv2d_relu_maxpool2_forward):
        // no second read of the pool output here (49.6 MB per step on the reference net)
        const bool pooled_pass = relu_below == nullptr && pool_below != nullptr && pool_below->passthrough_armed();
        const data_type* rb = relu_below != nullptr ? saved_input : nullptr;
        if (publish_backward && (rb != nullptr || pooled_pass)) must(cnn_amd_publish_next_kernel(stream), "cnn_amd_publish_next_kernel");
        must(cnn_conv2d_backward_prepared_relu(&d, saved_input, dy, prep_dgrad, rb, grads,
                                               grads + (size_t)out_channels * params_for_one_kernel, dbuf.base, (float)B,
                                               workspace, workspace_bytes, stream, /*defer_join=*/1),
             "cnn_conv2d_backward_prepared_relu");
        if (rb && relu_below) relu_below->fused_backward_done();
    } else
        must(cnn_conv2d_backward(&d, saved_input, dy, w_dev(), grads, grads + (size_t)out_channels * params_for_one_kernel,
                                 dbuf.base, (float)B, workspace, workspace_bytes, stream, /*defer_join=*/1),
             "cnn_conv2d_backward");  // joined in update_gradients / AlexNet::backward
    grads_ready = true;
    delta_valid = true;
    return first_n(dbuf.views, B);
}

// ---- fuse_pool_block: outputs the pass did not write, and the container-scheduled backward of a pool-fused first block ----
void Conv2D::materialize() const {
    const bool relu_missing = fused_relu != nullptr && !fused_relu->output_valid();
    if (out_valid && !relu_missing) return;
    assert(last_x != nullptr && "get_output() of a fused-away tensor before any forward pass");
    if (recompute_lost) {
        // (an exception, not abort(): inspecting a layer after load_weights() must not kill the process.  The C wrapper
        // cnnh_net_layer_output turns it into a return code.)
        throw std::runtime_error("cnn_amd host: " + name + ": get_output() of a tensor the last forward pass did not write, after the parameters that pass "
                                 "used were overwritten (set from outside, or stepped twice): call forward() again, or set "
                                 "architectures::fuse_pool_block = false");
    }
    // the parameters the last forward pass used: the container's snapshot when its SGD step has run since
    const data_type* w = (snapshot != nullptr && snapshot_active != nullptr && *snapshot_active) ? snapshot : params;
    const data_type* b = w + (size_t)out_channels * params_for_one_kernel;
    Conv2D* self = const_cast<Conv2D*>(this);
    self->ensure_workspace(last_B, in_H, in_W);
    cnn_conv2d_desc d{last_B, in_channels, in_H, in_W, out_channels, kernel_size, stride, padding, 0};
    // the unprepared entry points re-arrange the filters themselves: bit-identical to the prepared kernels the pass would have run
    if (fused_relu != nullptr && fuse_layers)
        must(cnn_conv2d_forward_relu(&d, last_x, w, b, out_buf.base, fused_relu->rematerialize_target(), workspace, workspace_bytes, stream),
             "cnn_conv2d_forward_relu");
    else
        must(cnn_conv2d_forward(&d, last_x, w, b, out_buf.base, workspace, workspace_bytes, stream), "cnn_conv2d_forward");
    out_valid = true;
}

std::vector<tensor> Conv2D::get_output() const {
    materialize();
    return Layer::get_output();
}

Conv2D::DeferredDgrad Conv2D::backward_weight_pooled(std::vector<tensor>& delta, bool fused_sgd, data_type learning_rate, data_type grad_scale) {
    Tensor3D::device_work_enqueued();
    const int B = (int)delta.size();
    assert(pool_fused_pass && prepared_active && fused_pool != nullptr && B == batch && saved_input != nullptr);
    const data_type* dy = batch_device_pointer(delta, delta_stage, name + "_dy");  // d(pool output)
    cnn_conv2d_desc d{B, in_channels, in_H, in_W, out_channels, kernel_size, stride, padding, 0};
    if (delta_buf.empty()) delta_buf.allocate(batch, in_channels, in_H, in_W, name + "_delta");
    const size_t need = cnn_conv2d_backward_workspace_bytes(&d);
    if (need > workspace_bytes) {
        if (workspace) cnn_device_free(workspace);
        workspace = dev_alloc(need);
        workspace_bytes = need;
    }
    if (prep_dgrad_alt == nullptr) prep_dgrad_alt = dev_alloc(cnn_conv2d_prepared_bytes(&d));
    const data_type* pooled = nullptr;  // (the ReLU mask rides in bit 31 of the pool mask, see Conv2D::backward)
    data_type* gw = grads;
    data_type* gb = grads + (size_t)out_channels * params_for_one_kernel;
    DeferredDgrad job;
    job.prepared = prep_dgrad;  // the filters of THIS step
    job.dpool = dy;
    job.mask = fused_pool->mask_dev();
    job.pooled = pooled;
    job.B = B;
    job.flags = pool_mask_flags;
    d.flags = pool_mask_flags;
    job.valid = true;
    if (fused_sgd) {
        // window kernel + ONE small launch: slab reduction, gw / gb, this layer's SGD step (old values -> snapshot) and the filter
        // images of the updated filters: forward in place, data gradient into the OTHER buffer (job.prepared stays intact)
        data_type* snap_w = const_cast<data_type*>(snapshot);
        must(cnn_conv2d_backward_weight_pooled2_sgd_keep(&d, saved_input, dy, job.mask, pooled, gw, gb, (float)B, w_dev(), b_dev(), learning_rate,
                                                         grad_scale, prep_fwd, prep_dgrad_alt, snap_w,
                                                         snap_w ? snap_w + (size_t)out_channels * params_for_one_kernel : nullptr, workspace,
                                                         workspace_bytes, stream),
             "cnn_conv2d_backward_weight_pooled2_sgd_keep");
        std::swap(prep_dgrad, prep_dgrad_alt);
    } else {
        must(cnn_conv2d_backward_weight_pooled2(&d, saved_input, dy, job.mask, pooled, gw, gb, (float)B, workspace, workspace_bytes, stream),
             "cnn_conv2d_backward_weight_pooled2");
    }
    pool_fused_pass = false;
    grads_ready = true;
    return job;
}

void Conv2D::prepare_own_filters() {
    cnn_conv2d_desc d = current_desc();
    if (prep_dgrad_alt == nullptr) prep_dgrad_alt = dev_alloc(cnn_conv2d_prepared_bytes(&d));
    const float* w = w_dev();
    const float* b = b_dev();
    void* f = prep_fwd;
    void* g = prep_dgrad_alt;
    must(cnn_conv2d_prepare_filters(1, &d, &w, &b, &f, &g, stream), "cnn_conv2d_prepare_filters");
    std::swap(prep_dgrad, prep_dgrad_alt);
    prepared_active = true;
}

void Conv2D::launch_deferred_dgrad(const DeferredDgrad& job, void* on_stream) {
    assert(job.valid);
    delta_valid = true;
    cnn_conv2d_desc d{job.B, in_channels, in_H, in_W, out_channels, kernel_size, stride, padding, job.flags};
    must(cnn_conv2d_backward_data_pooled2_prepared(&d, job.dpool, job.mask, job.pooled, job.prepared, delta_buf.base, on_stream),
         "cnn_conv2d_backward_data_pooled2_prepared");
}

void Conv2D::update_gradients(const data_type learning_rate) {
    assert(grads_ready);  // conv2d.cpp:206
    must(cnn_amd_side_stream_join(stream), "cnn_amd_side_stream_join");
    must(cnn_sgd_update(params, grads, param_count(), learning_rate, 1.f, stream), "cnn_sgd_update");
    prepared_active = false;
}

// checkpoint layout conv2d.cpp:220-226: Co filters then Co biases == the device layout, one contiguous block
void Conv2D::save_weights(std::ofstream& writer) const {
    std::vector<data_type> host(param_count());
    must(cnn_memcpy_d2h(host.data(), params, sizeof(data_type) * host.size(), stream), "cnn_memcpy_d2h");
    must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
    writer.write(reinterpret_cast<const char*>(host.data()), static_cast<std::streamsize>(sizeof(data_type) * host.size()));
}

void Conv2D::load_weights(std::ifstream& reader) {
    prepared_active = false;
    std::vector<data_type> host(param_count());
    reader.read(reinterpret_cast<char*>(host.data()), static_cast<std::streamsize>(sizeof(data_type) * host.size()));
    must(cnn_memcpy_h2d(params, host.data(), sizeof(data_type) * host.size(), stream), "cnn_memcpy_h2d");
    must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
}

int Conv2D::get_params_num() const { return (params_for_one_kernel + 1) * out_channels; }

// ---------------------------------------------------------------------------------------------------------------
// MaxPool2D
MaxPool2D::~MaxPool2D() {
    if (mask) cnn_device_free(mask);
    if (mask_alt) cnn_device_free(mask_alt);
}

void MaxPool2D::fused_forward_target(int B, int C, int H, int W, bool record, data_type** pooled, int** mask_out) {
    const int out_H = cnn_maxpool2d_out_dim(H, kernel_size, step), out_W = cnn_maxpool2d_out_dim(W, kernel_size, step);
    if (out_buf.empty()) {
        out_buf.allocate(B, C, out_H, out_W, name + "_output");
        output = out_buf.views;
        batch = B;
    }
    assert(B <= batch);
    in_C = C; in_H = H; in_W = W;
    // (sized for the int32 form; the packed one-byte form of the fused block needs at most (out_W + 3) bytes per row + 64)
    if (record && mask == nullptr) mask = (int*)dev_alloc(sizeof(int) * (size_t)batch * C * out_H * out_W + 64);
    if (alternate) {  // the deferred data gradient of the previous pass still reads the other set
        cur_set ^= 1;
        if (cur_set && out_buf_alt.empty()) out_buf_alt.allocate(batch, C, out_H, out_W, name + "_output");
        if (cur_set && record && mask_alt == nullptr) mask_alt = (int*)dev_alloc(sizeof(int) * (size_t)batch * C * out_H * out_W + 64);
        output = cur_set ? out_buf_alt.views : out_buf.views;
    }
    *pooled = cur_set ? out_buf_alt.base : out_buf.base;
    *mask_out = record ? (cur_set ? mask_alt : mask) : nullptr;
    forward_done = true;
    backward_passthrough = record;
}

std::vector<tensor> MaxPool2D::forward(const std::vector<tensor>& input) {
    Tensor3D::device_work_enqueued();  // host copies of device views made before this call are stale from here on
    if (forward_done) {  // written by the producing convolution's / BatchNorm2D's kernel in this pass
        forward_done = false;
        return first_n(output, (int)input.size());
    }
    backward_passthrough = false;
    const int B = (int)input.size();
    const int C = input[0]->C, H = input[0]->H, W = input[0]->W;
    const int out_H = cnn_maxpool2d_out_dim(H, kernel_size, step), out_W = cnn_maxpool2d_out_dim(W, kernel_size, step);
    if (out_buf.empty()) {
        out_buf.allocate(B, C, out_H, out_W, name + "_output");
        output = out_buf.views;
        batch = B;
    }
    if (cur_set != 0) {  // (an unfused pass always uses the first set)
        cur_set = 0;
        output = out_buf.views;
    }
    assert(B <= batch);
    if (in_C && (in_C != C || in_H != H || in_W != W)) {  // (the output / mask buffers were sized by the first shape)
        std::fprintf(stderr, "cnn_amd host: %s: input shape changed after the first forward (layers are shape-static)\n", name.c_str());
        std::abort();
    }
    in_C = C; in_H = H; in_W = W;
    if (!no_grad && mask == nullptr)  // pool2d.cpp:23-31 (allocated lazily so a no_grad first call is not fatal)
        mask = (int*)dev_alloc(sizeof(int) * (size_t)batch * C * out_H * out_W + 64);
    const data_type* x = batch_device_pointer(input, in_stage, name);
    must(cnn_maxpool2d_forward(x, out_buf.base, no_grad ? nullptr : mask, B, C, H, W, kernel_size, step, stream),
         "cnn_maxpool2d_forward");
    return first_n(output, B);
}

std::vector<tensor> MaxPool2D::backward(std::vector<tensor>& delta) {
    Tensor3D::device_work_enqueued();  // host copies of device views made before this call are stale from here on
    if (backward_passthrough) {  // the producing Conv2D's backward kernels consume the pooled-domain delta directly
        backward_passthrough = false;
        if (fused_relu_below != nullptr) fused_relu_below->fused_backward_done();
        return delta;
    }
    const int B = (int)delta.size();
    assert(mask != nullptr && "backward without a recorded forward (no_grad?)");
    if (delta_buf.empty()) delta_buf.allocate(batch, in_C, in_H, in_W, name + "_delta");
    const data_type* dy = batch_device_pointer(delta, delta_stage, name + "_dy");
    if (fused_bn_below != nullptr && fused_relu_below != nullptr && fuse_layers && fusable_2x2() && padding == 0 &&
        fused_bn_below->backward_pooled_possible(B)) {
        // BatchNorm2D -> ReLU -> this pool: the three backward passes as two kernels that rebuild the delta between them from the pooled
        // domain (bit-identical to the sequence; neither this layer's nor the ReLU's input gradient is materialised)
        fused_bn_below->backward_pooled(dy, mask_dev(), pooled_dev(), delta_buf.base, B);
        fused_relu_below->fused_backward_done();
        return first_n(delta_buf.views, B);
    }
    if (fused_relu_below != nullptr && fuse_layers) {  // also applies the ReLU::backward of the layer in front
        must(cnn_maxpool2d_backward_relu(dy, mask_dev(), pooled_dev(), delta_buf.base, B, in_C, in_H, in_W, kernel_size, step, stream),
             "cnn_maxpool2d_backward_relu");
        fused_relu_below->fused_backward_done();
    } else {
        must(cnn_maxpool2d_backward(dy, mask_dev(), delta_buf.base, B, in_C, in_H, in_W, kernel_size, step, stream),
             "cnn_maxpool2d_backward");
    }
    return first_n(delta_buf.views, B);
}

// ---------------------------------------------------------------------------------------------------------------
// ReLU
data_type* ReLU::fused_forward_target(int B, int C, int H, int W) {
    if (out_buf.empty()) {
        out_buf.allocate(B, C, H, W, name + "_output");
        output = out_buf.views;
    }
    assert((size_t)B <= out_buf.views.size() && out_buf.sample_len == (size_t)C * H * W);
    forward_done = true;
    out_valid = true;
    return out_buf.base;
}

void ReLU::fused_forward_skipped(int B, int C, int H, int W) {
    fused_forward_target(B, C, H, W);  // (keeps the layer's output tensors in place for the pass-through; not written)
    out_valid = false;
}

// alexnet.cpp:97,105 for a layer whose output the pool-fused pass did not write: the producing convolution re-computes it
std::vector<tensor> ReLU::get_output() const {
    if (!out_valid && producer != nullptr) producer->materialize();
    if (!out_valid && bn_producer != nullptr) bn_producer->materialize_relu();
    return Layer::get_output();
}

std::vector<tensor> ReLU::forward(const std::vector<tensor>& input) {
    Tensor3D::device_work_enqueued();  // host copies of device views made before this call are stale from here on
    const int B = (int)input.size();
    if (forward_done) {  // written by the producing convolution's kernel in this pass
        forward_done = false;
        return first_n(output, B);
    }
    if (out_buf.empty()) {
        out_buf.allocate(B, input[0]->C, input[0]->H, input[0]->W, name + "_output");
        output = out_buf.views;
    }
    assert((size_t)B <= out_buf.views.size());
    if (out_buf.sample_len != (size_t)input[0]->get_length()) {
        std::fprintf(stderr, "cnn_amd host: %s: input shape changed after the first forward (layers are shape-static)\n", name.c_str());
        std::abort();
    }
    const data_type* x = batch_device_pointer(input, in_stage, name);
    out_valid = true;
    must(cnn_relu_forward(x, out_buf.base, out_buf.sample_len * B, stream), "cnn_relu_forward");
    return first_n(output, B);
}

// relu.cpp:30-44: masks the caller's delta IN PLACE and hands the same tensors back
std::vector<tensor> ReLU::backward(std::vector<tensor>& delta) {
    Tensor3D::device_work_enqueued();  // host copies of device views made before this call are stale from here on
    const int B = (int)delta.size();
    if (backward_done) {  // masked by the pool's backward kernel in this pass
        backward_done = false;
        for (int b = 0; b < B; ++b) delta[b]->name = name + "_delta_" + std::to_string(b);
        return delta;
    }
    const bool in_place = delta[0]->on_device();
    data_type* d = batch_device_pointer_mut(delta, delta_stage, name + "_dy");
    must(cnn_relu_backward(out_buf.base, d, out_buf.sample_len * B, stream), "cnn_relu_backward");
    if (!in_place || d == delta_stage.base) {  // host (or scattered) deltas were staged: write the result back
        for (int b = 0; b < B; ++b) {
            if (delta[b]->on_device())
                must(cnn_memcpy_d2d(delta[b]->dev, d + out_buf.sample_len * b, sizeof(data_type) * out_buf.sample_len, stream),
                     "cnn_memcpy_d2d");
            else
                must(cnn_memcpy_d2h(delta[b]->data, d + out_buf.sample_len * b, sizeof(data_type) * out_buf.sample_len, stream),
                     "cnn_memcpy_d2h");
        }
        must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
    }
    for (int b = 0; b < B; ++b) delta[b]->name = name + "_delta_" + std::to_string(b);
    return delta;
}

// ---------------------------------------------------------------------------------------------------------------
// BatchNorm2D
BatchNorm2D::BatchNorm2D(std::string _name, const int _out_channels, const data_type _eps, const data_type _momentum)
    : Layer(_name), out_channels(_out_channels), eps(_eps), momentum(_momentum) {
    assert(_out_channels > 0);
    const size_t n = param_count();
    params = (data_type*)dev_alloc(sizeof(data_type) * n);
    grads = (data_type*)dev_alloc(sizeof(data_type) * n);
    saved_stats = (data_type*)dev_alloc(sizeof(data_type) * 2 * out_channels);
    std::vector<data_type> host(n, 0);  // batchnorm2d.cpp:18-20: gamma = 1, beta = 0, moving_mean = moving_var = 0
    for (int o = 0; o < out_channels; ++o) host[o] = 1;
    must(cnn_memcpy_h2d(params, host.data(), sizeof(data_type) * n, stream), "cnn_memcpy_h2d");
    must(cnn_memset_zero(grads, sizeof(data_type) * n, stream), "cnn_memset_zero");
    must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
}

BatchNorm2D::~BatchNorm2D() {
    if (owns_params) {
        cnn_device_free(params);
        cnn_device_free(grads);
    }
    cnn_device_free(saved_stats);
    if (sync_sums) cnn_device_free(sync_sums);
    if (workspace) cnn_device_free(workspace);
}

void BatchNorm2D::bind_arena(data_type* params_dev, data_type* grads_dev) {
    must(cnn_memcpy_d2d(params_dev, params, sizeof(data_type) * param_count(), stream), "cnn_memcpy_d2d");
    // the moving_* half of the gradient block must be (and stay) zero: the arena-wide SGD step runs over it
    must(cnn_memset_zero(grads_dev, sizeof(data_type) * param_count(), stream), "cnn_memset_zero");
    must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
    if (owns_params) {
        cnn_device_free(params);
        cnn_device_free(grads);
    }
    params = params_dev;
    grads = grads_dev;
    owns_params = false;
}

std::vector<tensor> BatchNorm2D::forward(const std::vector<tensor>& input) {
    Tensor3D::device_work_enqueued();  // host copies of device views made before this call are stale from here on
    const int B = (int)input.size();
    const int H = input[0]->H, W = input[0]->W;
    assert(input[0]->C == out_channels);
    if (out_buf.empty()) {  // batchnorm2d.cpp:30-35
        out_buf.allocate(B, out_channels, H, W, name + "_output");
        output = out_buf.views;
        batch = B;
    }
    assert(B <= batch);
    if ((in_H && in_H != H) || (in_W && in_W != W)) {
        std::fprintf(stderr, "cnn_amd host: %s: input shape changed after the first forward (layers are shape-static)\n", name.c_str());
        std::abort();
    }
    in_H = H;
    in_W = W;
    const size_t need = cnn_batchnorm2d_workspace_bytes(B, out_channels, H, W);
    if (need > workspace_bytes) {
        if (workspace) cnn_device_free(workspace);
        workspace = dev_alloc(need);
        workspace_bytes = need;
    }
    const data_type* x = batch_device_pointer(input, in_stage, name);
    if (!no_grad) {  // batchnorm2d.cpp:42
        saved_input = x;
        saved_input_tensors = input;
    }
    const int C = out_channels;
    // the ReLU behind this layer gets its output from the same apply pass (both outputs are written)
    data_type* y_relu = (fused_relu != nullptr && fuse_layers) ? fused_relu->fused_forward_target(B, C, H, W) : nullptr;
    // fuse_pool_block: a training pass does not write the normalised tensor when the ReLU output comes from the same kernel -- nothing in a
    // train step reads it (ReLU::backward masks by its own output, backward() below recomputes from x and the saved statistics: 589 MB of
    // stores per step of the ResNet-18-shaped stack); get_output() re-computes it on demand
    const bool relu_only = y_relu != nullptr && !no_grad && fuse_pool_block;
    data_type* y_out = relu_only ? nullptr : out_buf.base;
    out_valid = !relu_only;
    recompute_lost = false;
    last_B = B;
    if (!no_grad && comm != nullptr && comm_world > 1) {
        // the batch is sharded over comm_world replicas: the reference normalises over the WHOLE batch (batchnorm2d.cpp:46-63),
        // so the per-channel sums are exchanged (two [C] all-reduces: mean first, then the squared deviations around the
        // GLOBAL mean -- still the reference's two-pass variance); equal shards assumed (count = B * world * H * W)
        if (!sync_sums) sync_sums = (data_type*)dev_alloc(sizeof(data_type) * 6 * C);
        const float count = (float)((double)B * comm_world * H * W);
        data_type* s1 = sync_sums;
        data_type* s2 = sync_sums + C;
        must(cnn_batchnorm2d_partial_sums(x, nullptr, 0.f, s1, B, C, H, W, workspace, workspace_bytes, stream), "cnn_batchnorm2d_partial_sums");
        must(cnn_allreduce_grads(comm, s1, (size_t)C, stream), "cnn_allreduce_grads");
        must(cnn_batchnorm2d_partial_sums(x, s1, count, s2, B, C, H, W, workspace, workspace_bytes, stream), "cnn_batchnorm2d_partial_sums");
        must(cnn_allreduce_grads(comm, s2, (size_t)C, stream), "cnn_allreduce_grads");
        if (y_relu)
            must(cnn_batchnorm2d_forward_from_sums_relu(x, y_out, y_relu, params, params + C, params + 2 * C, params + 3 * C,
                                                        saved_stats, saved_stats + C, s1, s2, count, B, C, H, W, eps, momentum, stream),
                 "cnn_batchnorm2d_forward_from_sums_relu");
        else
            must(cnn_batchnorm2d_forward_from_sums(x, out_buf.base, params, params + C, params + 2 * C, params + 3 * C, saved_stats,
                                                   saved_stats + C, s1, s2, count, B, C, H, W, eps, momentum, stream),
                 "cnn_batchnorm2d_forward_from_sums");
        return first_n(output, B);
    }
    if (y_relu && fused_pool != nullptr && fused_pool->plain_2x2() && cnn_batchnorm2d_forward_relu_pool_supported(B, C, H, W) != 0) {
        // BatchNorm2D -> ReLU -> MaxPool2D(2, 2): the apply pass pools as well.  A training pass under fuse_pool_block writes neither the
        // normalised tensor nor the ReLU output (backward() re-computes from x, the pool's backward pass takes ReLU' from the pooled value);
        // both stay observable: get_output() re-computes them from the statistics this pass saved
        data_type* pooled = nullptr;
        int* pmask = nullptr;
        fused_pool->fused_forward_target(B, C, H, W, /*record=*/!no_grad, &pooled, &pmask);
        fused_pool->own_backward();
        if (relu_only) {
            fused_relu->fused_forward_skipped(B, C, H, W);
            fused_relu->set_bn_producer(this);
            y_relu = nullptr;
        }
        must(cnn_batchnorm2d_forward_relu_pool(x, y_out, y_relu, pooled, pmask, params, params + C, params + 2 * C, params + 3 * C, saved_stats,
                                               saved_stats + C, B, C, H, W, eps, momentum, no_grad ? 0 : 1, workspace, workspace_bytes, stream),
             "cnn_batchnorm2d_forward_relu_pool");
    } else if (y_relu)
        must(cnn_batchnorm2d_forward_relu(x, y_out, y_relu, params, params + C, params + 2 * C, params + 3 * C, saved_stats,
                                          saved_stats + C, B, C, H, W, eps, momentum, no_grad ? 0 : 1, workspace, workspace_bytes,
                                          stream),
             "cnn_batchnorm2d_forward_relu");
    else
        must(cnn_batchnorm2d_forward(x, out_buf.base, params, params + C, params + 2 * C, params + 3 * C, saved_stats,
                                     saved_stats + C, B, C, H, W, eps, momentum, no_grad ? 0 : 1, workspace, workspace_bytes,
                                     stream),
             "cnn_batchnorm2d_forward");
    return first_n(output, B);
}

// the normalised tensor of a pass that wrote only the ReLU output: gamma * ((x - mean) * inv_std) + beta with the batch statistics that pass
// saved and the gamma / beta it used (the container's snapshot once its SGD step has run) -- the evaluation entry with the saved statistics in
// the place of the moving ones runs the training pass' arithmetic (same expression, contraction off): bit-identical to a pass that writes it
void BatchNorm2D::materialize() const {
    if (out_valid) return;
    assert(saved_input != nullptr && "get_output() of a fused-away tensor before any forward pass");
    if (recompute_lost) {
        // (an exception, not abort(): inspecting a layer after load_weights() must not kill the process.  The C wrapper
        // cnnh_net_layer_output turns it into a return code.)
        throw std::runtime_error("cnn_amd host: " + name + ": get_output() of a tensor the last forward pass did not write, after the parameters that pass "
                                 "used were overwritten (set from outside, or stepped twice): call forward() again, or set "
                                 "architectures::fuse_pool_block = false");
    }
    const int C = out_channels;
    const data_type* gb = (snapshot != nullptr && snapshot_active != nullptr && *snapshot_active) ? snapshot : params;
    must(cnn_batchnorm2d_forward(saved_input, out_buf.base, gb, gb + C, saved_stats, saved_stats + C, nullptr, nullptr, last_B, C, in_H, in_W, eps,
                                 momentum, 0, workspace, workspace_bytes, stream),
         "cnn_batchnorm2d_forward");
    out_valid = true;
}

// the ReLU output of such a pass, on demand: relu(gamma * ((x - mean) * inv_std) + beta) with the saved statistics -- the evaluation entry's
// arithmetic is the training pass' (see materialize())
void BatchNorm2D::materialize_relu() const {
    if (fused_relu == nullptr || fused_relu->output_valid()) return;
    assert(saved_input != nullptr && "get_output() of a fused-away tensor before any forward pass");
    if (recompute_lost)
        throw std::runtime_error("cnn_amd host: " + name + ": get_output() of the ReLU output the last forward pass did not write, after the parameters that "
                                 "pass used were overwritten: call forward() again, or set architectures::fuse_pool_block = false");
    const int C = out_channels;
    const data_type* gb = (snapshot != nullptr && snapshot_active != nullptr && *snapshot_active) ? snapshot : params;
    must(cnn_batchnorm2d_forward_relu(saved_input, nullptr, fused_relu->rematerialize_target_const(), gb, gb + C, saved_stats, saved_stats + C, nullptr,
                                      nullptr, last_B, C, in_H, in_W, eps, momentum, 0, workspace, workspace_bytes, stream),
         "cnn_batchnorm2d_forward_relu");
}

std::vector<tensor> BatchNorm2D::get_output() const {
    materialize();
    return Layer::get_output();
}

// batchnorm2d.cpp:98-158: gamma / beta gradients (sums over the batch, not averaged) and the data gradient written
// IN PLACE into the caller's delta, which is handed back
bool BatchNorm2D::backward_pooled_possible(int B) const {
    return saved_input != nullptr && !(comm != nullptr && comm_world > 1) && B == last_B &&
           cnn_batchnorm2d_backward_pooled_supported(B, out_channels, in_H, in_W) != 0;
}

void BatchNorm2D::backward_pooled(const data_type* dpool, const int* mask, const data_type* pooled, data_type* dx, int B) {
    const int C = out_channels;
    must(cnn_batchnorm2d_backward_pooled(saved_input, dpool, mask, pooled, dx, params, saved_stats, saved_stats + C, grads, grads + C, B, C, in_H,
                                         in_W, eps, workspace, workspace_bytes, stream),
         "cnn_batchnorm2d_backward_pooled");
    backward_done_by_pool = true;
}

std::vector<tensor> BatchNorm2D::backward(std::vector<tensor>& delta) {
    Tensor3D::device_work_enqueued();  // host copies of device views made before this call are stale from here on
    if (backward_done_by_pool) {  // (the pool behind this layer's ReLU ran this pass from the pooled domain: delta already is dx)
        backward_done_by_pool = false;
        grads_ready = true;
        return delta;
    }
    const int B = (int)delta.size();
    assert(saved_input != nullptr && "backward without a recorded forward (no_grad?)");
    const int C = out_channels;
    const bool in_place = delta[0]->on_device();
    data_type* d = batch_device_pointer_mut(delta, delta_stage, name + "_dy");
    if (comm != nullptr && comm_world > 1) {
        // sync-BN: one [C][4] all-reduce; gamma / beta gradients come out as the FULL-batch sums on every replica (the arena
        // all-reduce + the 1/world of the SGD step leave them unchanged: world * sum / world)
        const float count = (float)((double)B * comm_world * in_H * in_W);
        data_type* s4 = sync_sums + 2 * C;
        must(cnn_batchnorm2d_backward_sums(saved_input, d, params, saved_stats, saved_stats + C, s4, B, C, in_H, in_W, eps, workspace,
                                           workspace_bytes, stream),
             "cnn_batchnorm2d_backward_sums");
        must(cnn_allreduce_grads(comm, s4, (size_t)4 * C, stream), "cnn_allreduce_grads");
        must(cnn_batchnorm2d_backward_from_sums(saved_input, d, params, saved_stats, saved_stats + C, s4, count, grads, grads + C, B, C,
                                                in_H, in_W, eps, stream),
             "cnn_batchnorm2d_backward_from_sums");
    } else
        must(cnn_batchnorm2d_backward(saved_input, d, params, saved_stats, saved_stats + C, grads, grads + C, B, C, in_H, in_W,
                                      eps, workspace, workspace_bytes, stream),
             "cnn_batchnorm2d_backward");
    if (!in_place || d == delta_stage.base) {  // host (or scattered) deltas were staged: write the result back
        const size_t len = out_buf.sample_len;
        for (int b = 0; b < B; ++b) {
            if (delta[b]->on_device())
                must(cnn_memcpy_d2d(delta[b]->dev, d + len * b, sizeof(data_type) * len, stream), "cnn_memcpy_d2d");
            else
                must(cnn_memcpy_d2h(delta[b]->data, d + len * b, sizeof(data_type) * len, stream), "cnn_memcpy_d2h");
        }
        must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
    }
    grads_ready = true;
    return delta;
}

void BatchNorm2D::update_gradients(const data_type learning_rate) {  // batchnorm2d.cpp:160-166: gamma and beta only
    assert(grads_ready);
    must(cnn_sgd_update(params, grads, (size_t)2 * out_channels, learning_rate, 1.f, stream), "cnn_sgd_update");
}

// batchnorm2d.cpp:168-182: gamma, beta, moving_mean, moving_var == the device block
void BatchNorm2D::save_weights(std::ofstream& writer) const {
    std::vector<data_type> host(param_count());
    must(cnn_memcpy_d2h(host.data(), params, sizeof(data_type) * host.size(), stream), "cnn_memcpy_d2h");
    must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
    writer.write(reinterpret_cast<const char*>(host.data()), static_cast<std::streamsize>(sizeof(data_type) * host.size()));
}

void BatchNorm2D::load_weights(std::ifstream& reader) {
    std::vector<data_type> host(param_count());
    reader.read(reinterpret_cast<char*>(host.data()), static_cast<std::streamsize>(sizeof(data_type) * host.size()));
    must(cnn_memcpy_h2d(params, host.data(), sizeof(data_type) * host.size(), stream), "cnn_memcpy_h2d");
    must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
}

// ---------------------------------------------------------------------------------------------------------------
// Dropout (dropout.cpp)
std::vector<tensor> Dropout::forward(const std::vector<tensor>& input) {
    Tensor3D::device_work_enqueued();  // host copies of device views made before this call are stale from here on
    const int B = (int)input.size();
    const int C = input[0]->C, H = input[0]->H, W = input[0]->W;
    if (sequence.empty()) {  // dropout.cpp:13-23
        sequence.assign(C, 0);
        for (int o = 0; o < C; ++o) sequence[o] = o;
        selected_num = int(p * C);
        assert(C > selected_num);
        mask.assign(C, 0);
        out_buf.allocate(B, C, H, W, name + "_output");
        output = out_buf.views;
        in_C = C; in_H = H; in_W = W;
    }
    assert((size_t)B <= out_buf.views.size() && C == in_C && H == in_H && W == in_W);
    std::shuffle(sequence.begin(), sequence.end(), drop);  // dropout.cpp:26 (the draw happens in both modes)
    if (!no_grad)
        for (int i = 0; i < C; ++i) mask[i] = i >= selected_num ? sequence[i] : -1;  // dropout.cpp:31-33
    const data_type* x = batch_device_pointer(input, in_stage, name);
    must(cnn_dropout_forward(x, out_buf.base, B, C, H, W, selected_num, no_grad ? 0 : 1, 1 - p, stream), "cnn_dropout_forward");
    return first_n(output, B);
}

// dropout.cpp:57-69: in place on the caller's delta
std::vector<tensor> Dropout::backward(std::vector<tensor>& delta) {
    Tensor3D::device_work_enqueued();  // host copies of device views made before this call are stale from here on
    const int B = (int)delta.size();
    const bool in_place = delta[0]->on_device();
    data_type* d = batch_device_pointer_mut(delta, delta_stage, name + "_dy");
    int dropped = 0;
    for (int o = 0; o < in_C; ++o) dropped += mask[o] == -1;  // (= selected_num after a training forward, 0 before any)
    must(cnn_dropout_backward(d, B, in_C, in_H, in_W, dropped, stream), "cnn_dropout_backward");
    if (!in_place || d == delta_stage.base) {  // host (or scattered) deltas were staged: write the result back
        const size_t len = out_buf.sample_len;
        for (int b = 0; b < B; ++b) {
            if (delta[b]->on_device())
                must(cnn_memcpy_d2d(delta[b]->dev, d + len * b, sizeof(data_type) * len, stream), "cnn_memcpy_d2d");
            else
                must(cnn_memcpy_d2h(delta[b]->data, d + len * b, sizeof(data_type) * len, stream), "cnn_memcpy_d2h");
        }
        must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
    }
    return delta;
}

// ---------------------------------------------------------------------------------------------------------------
// LinearLayer
LinearLayer::LinearLayer(std::string _name, const int _in_channels, const int _out_channels)
    : Layer(_name), in_channels(_in_channels), out_channels(_out_channels) {
    const size_t n = param_count();
    params = (data_type*)dev_alloc(sizeof(data_type) * n);
    grads = (data_type*)dev_alloc(sizeof(data_type) * n);
    std::vector<data_type> host(n);
    std::default_random_engine e(1998);  // linear.cpp:14-18: bias first, then the matrix
    std::normal_distribution<float> engine(0.0, 1.0);
    data_type* bias = host.data() + (size_t)in_channels * out_channels;
    for (int i = 0; i < out_channels; ++i) bias[i] = engine(e) / random_times;
    for (size_t i = 0; i < (size_t)in_channels * out_channels; ++i) host[i] = engine(e) / random_times;
    must(cnn_memcpy_h2d(params, host.data(), sizeof(data_type) * n, stream), "cnn_memcpy_h2d");
    must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
}

LinearLayer::~LinearLayer() {
    if (own_stream) {
        cnn_stream_synchronize(own_stream);
        cnn_stream_destroy(own_stream);
    }
    if (ev_wb) cnn_event_destroy(ev_wb);
    if (ev_head) cnn_event_destroy(ev_head);
    if (owns_params) {
        cnn_device_free(params);
        cnn_device_free(grads);
    }
}

void LinearLayer::bind_arena(data_type* params_dev, data_type* grads_dev) {
    must(cnn_memcpy_d2d(params_dev, params, sizeof(data_type) * param_count(), stream), "cnn_memcpy_d2d");
    must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
    if (owns_params) {
        cnn_device_free(params);
        cnn_device_free(grads);
    }
    params = params_dev;
    grads = grads_dev;
    owns_params = false;
}

std::vector<tensor> LinearLayer::forward(const std::vector<tensor>& input) {
    Tensor3D::device_work_enqueued();  // host copies of device views made before this call are stale from here on
    const int B = (int)input.size();
    delta_shape = input[0]->get_shape();  // linear.cpp:25
    assert(input[0]->get_length() == in_channels);
    if (out_buf.empty()) {
        out_buf.allocate(B, out_channels, 1, 1, name + "_output");
        output = out_buf.views;
        batch = B;
    }
    assert(B <= batch);
    const data_type* x = batch_device_pointer(input, in_stage, name);
    if (!no_grad) {
        saved_input = x;
        saved_input_tensors = input;
    }
    must(cnn_linear_forward(x, params, params + (size_t)in_channels * out_channels, out_buf.base, B, in_channels,
                            out_channels, stream),
         "cnn_linear_forward");
    if (lazy_host_sync) return first_n(output, B);  // materialised on demand (Layer::get_output / Tensor3D::sync_to_host)
    // the callers read the logits on the host right away (softmax, func.cpp:24-28; argmax, cnn.cpp:92): one D2H
    std::vector<data_type> host((size_t)B * out_channels);
    must(cnn_memcpy_d2h(host.data(), out_buf.base, sizeof(data_type) * host.size(), stream), "cnn_memcpy_d2h");
    must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
    for (int b = 0; b < B; ++b) {
        if (!output[b]->data) output[b]->data = new data_type[out_channels];
        std::memcpy(output[b]->data, host.data() + (size_t)b * out_channels, sizeof(data_type) * out_channels);
        output[b]->mark_host_fresh();
    }
    return first_n(output, B);
}

std::vector<tensor> LinearLayer::forward_loss_head(const std::vector<tensor>& input, const int* labels_dev, data_type* probs_dev,
                                                   data_type* delta_dev, data_type* loss_terms_dev, bool with_dx) {
    Tensor3D::device_work_enqueued();  // host copies of device views made before this call are stale from here on
    const int B = (int)input.size();
    delta_shape = input[0]->get_shape();
    assert(input[0]->get_length() == in_channels && loss_head_supported());
    if (out_buf.empty()) {
        out_buf.allocate(B, out_channels, 1, 1, name + "_output");
        output = out_buf.views;
        batch = B;
    }
    assert(B <= batch);
    const data_type* x = batch_device_pointer(input, in_stage, name);
    saved_input = x;
    saved_input_tensors = input;
    head_dx_done = false;
    if (with_dx && B == batch) {
        if (delta_buf.empty())
            delta_buf.allocate(batch, std::get<0>(delta_shape), std::get<1>(delta_shape), std::get<2>(delta_shape), "linear_delta");
        // (this kernel is now the last one in front of the next layer's weight-gradient fork: it carries the event)
        if (publish_backward) must(cnn_amd_publish_next_kernel(stream), "cnn_amd_publish_next_kernel");
        must(cnn_linear_forward_softmax_xent_dx(x, params, params + (size_t)in_channels * out_channels, labels_dev, out_buf.base, probs_dev,
                                                delta_dev, loss_terms_dev, delta_buf.base, (relu_below != nullptr && fuse_layers) ? 1 : 0, B,
                                                in_channels, out_channels, stream),
             "cnn_linear_forward_softmax_xent_dx");
        head_dx_done = true;
        return first_n(output, B);
    }
    must(cnn_linear_forward_softmax_xent(x, params, params + (size_t)in_channels * out_channels, labels_dev, out_buf.base, probs_dev,
                                         delta_dev, loss_terms_dev, B, in_channels, out_channels, stream),
         "cnn_linear_forward_softmax_xent");
    return first_n(output, B);
}

std::vector<tensor> LinearLayer::backward(std::vector<tensor>& delta) {
    Tensor3D::device_work_enqueued();  // host copies of device views made before this call are stale from here on
    const int B = (int)delta.size();
    assert(saved_input != nullptr && "backward without a recorded forward (no_grad?)");
    const data_type* dy = batch_device_pointer(delta, delta_stage, name + "_dy");
    if (delta_buf.empty())
        delta_buf.allocate(batch, std::get<0>(delta_shape), std::get<1>(delta_shape), std::get<2>(delta_shape),
                           "linear_delta");
    if (head_dx_done) {
        // the data gradient (and the ReLU::backward in front) came out of the loss-head kernel; what is left needs every sample's
        // delta and nobody on the critical path waits for it: weight / bias gradient on the library's side stream, like the
        // convolutions' (ordered before the join / the step's tail)
        head_dx_done = false;
        void* side = nullptr;
        if (cnn_amd_get_option("LINEAR_WB_OWN_STREAM", nullptr, 0) != 0) {
            // default: the library's side stream, in front of the convolutions' weight gradients (measured: 588-598 k images/s
            // against 569-577 k with a stream of its own -- a fourth concurrent kernel costs the critical ones more than the
            // earlier start of the weight-gradient chain gains)
            must(cnn_amd_side_stream_get(&side), "cnn_amd_side_stream_get");
        } else {
            if (own_stream == nullptr) {
                must(cnn_stream_create(&own_stream), "cnn_stream_create");
                must(cnn_event_create(&ev_wb), "cnn_event_create");
            }
            side = own_stream;
        }
        if (cnn_amd_published_is_last(stream)) {
            must(cnn_amd_wait_published(side), "cnn_amd_wait_published");
        } else {
            if (ev_head == nullptr) must(cnn_event_create(&ev_head), "cnn_event_create");
            must(cnn_event_record(ev_head, stream), "cnn_event_record");
            must(cnn_stream_wait_event(side, ev_head), "cnn_stream_wait_event");
        }
        must(cnn_linear_backward(saved_input, dy, params, grads, grads + (size_t)in_channels * out_channels, nullptr, B, in_channels,
                                 out_channels, (float)B, side),
             "cnn_linear_backward");
        if (side == own_stream) {
            must(cnn_event_record(ev_wb, own_stream), "cnn_event_record");
            wb_pending = true;
        }
        if (relu_below != nullptr && fuse_layers) relu_below->fused_backward_done();
        grads_ready = true;
        return first_n(delta_buf.views, B);
    }
    if (relu_below != nullptr && fuse_layers) {  // the input IS that ReLU's output: its backward mask in the same kernel
        if (publish_backward) must(cnn_amd_publish_next_kernel(stream), "cnn_amd_publish_next_kernel");
        must(cnn_linear_backward_relu(saved_input, dy, params, grads, grads + (size_t)in_channels * out_channels, delta_buf.base, B,
                                      in_channels, out_channels, (float)B, stream),
             "cnn_linear_backward_relu");
        relu_below->fused_backward_done();
    } else {
        must(cnn_linear_backward(saved_input, dy, params, grads, grads + (size_t)in_channels * out_channels, delta_buf.base, B,
                                 in_channels, out_channels, (float)B, stream),
             "cnn_linear_backward");
    }
    grads_ready = true;
    return first_n(delta_buf.views, B);
}

void LinearLayer::join_pending(void* on_stream) {
    if (!wb_pending) return;
    must(cnn_stream_wait_event(on_stream, ev_wb), "cnn_stream_wait_event");
    wb_pending = false;
}

void LinearLayer::update_gradients(const data_type learning_rate) {
    assert(grads_ready);  // linear.cpp:97
    join_pending(stream);
    must(cnn_sgd_update(params, grads, param_count(), learning_rate, 1.f, stream), "cnn_sgd_update");
}

void LinearLayer::save_weights(std::ofstream& writer) const {
    std::vector<data_type> host(param_count());
    must(cnn_memcpy_d2h(host.data(), params, sizeof(data_type) * host.size(), stream), "cnn_memcpy_d2h");
    must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
    writer.write(reinterpret_cast<const char*>(host.data()), static_cast<std::streamsize>(sizeof(data_type) * host.size()));
}

void LinearLayer::load_weights(std::ifstream& reader) {
    std::vector<data_type> host(param_count());
    reader.read(reinterpret_cast<char*>(host.data()), static_cast<std::streamsize>(sizeof(data_type) * host.size()));
    must(cnn_memcpy_h2d(params, host.data(), sizeof(data_type) * host.size(), stream), "cnn_memcpy_h2d");
    must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
}
