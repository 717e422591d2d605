// sequential.cpp -- the reference's model container (cpu/src/alexnet.cpp:35-90: a std::list of Layers walked in order) for any
// layer list, plus the flat parameter / gradient arena that makes the SGD step one kernel and the data-parallel exchange
// one all-reduce, the fusion wiring between neighbouring layers, and the BASELINE workloads as layer lists.
// (class AlexNet itself lives in alexnet.cpp -- the one file of this directory the reference's own cpu/src/alexnet.cpp can
// replace, see INTEGRATION.md.)
#include <algorithm>
#include <cassert>
#include <cstdlib>
#include <iostream>
#include <iterator>

#include "architectures.h"
#include "host_util.h"

using namespace architectures;
using cnn_amd_host::dev_alloc;
using cnn_amd_host::must;

// ---------------------------------------------------------------------------------------------------------------
// Sequential
Sequential& Sequential::add(Layer* layer) {
    assert(!finalized && "Sequential::add after finalize()");
    layers_sequence.emplace_back(layer);
    return *this;
}

size_t Sequential::num_params() const {
    if (finalized) return n_params;
    size_t n = 0;
    for (const auto& layer : layers_sequence) n += layer->param_count();
    return n;
}

// fusion wiring (architectures::fuse_layers): neighbouring layers that can share a kernel get to know each other; whether a
// fused kernel is actually used is decided per call by the layers (geometry support, fuse_layers / fuse_pool_block flags)
void Sequential::wire() {
    for (auto it = layers_sequence.begin(); it != layers_sequence.end(); ++it) {
        auto next = std::next(it);
        if (next == layers_sequence.end()) break;
        // Conv2D -> ReLU: the ReLU's output is written by the convolution kernel's epilogue
        if (auto* relu = dynamic_cast<ReLU*>(next->get())) {
            if (auto* conv = dynamic_cast<Conv2D*>(it->get())) {
                conv->set_fused_relu(relu);
                relu->set_producer(conv);
            }
            // BatchNorm2D -> ReLU: ... or by the normalisation's apply pass
            if (auto* bn = dynamic_cast<BatchNorm2D*>(it->get())) bn->set_fused_relu(relu);
        }
        // ReLU -> MaxPool2D: the ReLU's backward pass runs inside the pool's backward kernel
        if (auto* pool = dynamic_cast<MaxPool2D*>(next->get())) {
            if (auto* relu = dynamic_cast<ReLU*>(it->get())) {
                pool->set_fused_relu_below(relu);
                // BatchNorm2D -> ReLU -> MaxPool2D: ... and the normalisation's backward pass takes its delta from the pooled domain
                if (it != layers_sequence.begin())
                    if (auto* bn = dynamic_cast<BatchNorm2D*>(std::prev(it)->get())) {
                        pool->set_fused_bn_below(bn);
                        bn->set_fused_pool(pool);  // (forward: one apply pass writes the pooled tensor and the mask)
                    }
            }
        }
        // ReLU -> Conv2D / LinearLayer: the ReLU's backward pass runs inside the consumer's data-gradient kernel
        if (auto* relu = dynamic_cast<ReLU*>(it->get())) {
            if (auto* conv = dynamic_cast<Conv2D*>(next->get())) conv->set_relu_below(relu);
            if (auto* lin = dynamic_cast<LinearLayer*>(next->get())) lin->set_relu_below(relu);
        }
        // Conv2D -> ReLU -> {Conv2D, LinearLayer} (or Conv2D -> ReLU -> MaxPool2D -> Conv2D): the data gradient of the upper layer is
        // the last kernel in front of the lower convolution's weight-gradient fork whenever the ReLU (and pool) backward passes are
        // fused away -- it then carries the fork event in its own dispatch packet (cnn_amd_publish_next_kernel)
        if (dynamic_cast<Conv2D*>(it->get()) && dynamic_cast<ReLU*>(next->get())) {
            auto after = std::next(next);
            if (after != layers_sequence.end() && dynamic_cast<MaxPool2D*>(after->get())) after = std::next(after);
            if (after != layers_sequence.end()) {
                if (auto* conv = dynamic_cast<Conv2D*>(after->get())) conv->set_publish_backward(true);
                if (auto* lin = dynamic_cast<LinearLayer*>(after->get())) lin->set_publish_backward(true);
            }
        }
        // Conv2D -> ReLU -> MaxPool2D: one forward kernel, backward from the pooled domain (opt-in: fuse_pool_block)
        auto next2 = std::next(next);
        if (next2 != layers_sequence.end()) {
            auto* conv = dynamic_cast<Conv2D*>(it->get());
            auto* relu = dynamic_cast<ReLU*>(next->get());
            auto* pool = dynamic_cast<MaxPool2D*>(next2->get());
            if (conv && relu && pool) {
                conv->set_fused_pool(pool);
                // ... and the convolution BEHIND such a block may fold the block's ReLU::backward into its data gradient
                auto next3 = std::next(next2);
                if (next3 != layers_sequence.end())
                    if (auto* behind = dynamic_cast<Conv2D*>(next3->get())) behind->set_pool_below(pool);
                // the block at the very FRONT of the network: nothing consumes its data gradient (alexnet.cpp:55), train_step may
                // defer that kernel into the next forward pass and run the step's tail under the block's weight gradient
                if (it == layers_sequence.begin() && pool->fusable_2x2() && next3 != layers_sequence.end()) {
                    block_conv = conv;
                    block_pool = pool;
                    behind_block = next3->get();
                }
            }
        }
    }
}

// the deferred data gradient is released behind the THIRD convolution's forward kernel (or the last one of a shorter net): there
// it overlaps the latency-bound deep layers, the linear layer and the loss instead of the HBM-bound first ones (measured on the
// reference net, profiles/NOTEBOOK.md section 4.4)
static Layer* pick_release_layer(const std::list<std::shared_ptr<Layer> >& layers) {
    char text[16] = {0};
    int want = 3;  // (DX0_RELEASE=n: behind the n-th convolution instead; tuning)
    if (cnn_amd_get_option("DX0_RELEASE", text, sizeof(text)) == 0 && std::atoi(text) >= 2) want = std::atoi(text);
    Layer* pick = nullptr;
    int convs = 0;
    for (const auto& layer : layers)
        if (dynamic_cast<Conv2D*>(layer.get())) {
            pick = layer.get();
            if (++convs == want) break;
        }
    return convs >= 2 ? pick : nullptr;
}

void Sequential::bind(data_type* p, data_type* g) {
    param_arena = p;
    grad_arena = g;
    size_t off = 0;  // checkpoint order == layer order (alexnet.cpp:73-74)
    layer_offsets.clear();
    for (auto& layer : layers_sequence) {
        const size_t n = layer->param_count();
        layer_offsets.push_back(off);
        if (n) layer->bind_arena(p + off, g + off);
        // fuse_pool_block: Conv2D::get_output() of a fused-away tensor needs the parameters of the last forward pass -- the
        // container keeps them across its SGD step (cnn_sgd_update_keep / ..._sgd_keep write them here)
        if (auto* conv = dynamic_cast<Conv2D*>(layer.get())) conv->set_param_snapshot(param_prev + off, &params_stepped);
        if (auto* bn = dynamic_cast<BatchNorm2D*>(layer.get())) bn->set_param_snapshot(param_prev + off, &params_stepped);
        off += n;
    }
    if (block_conv != nullptr) {
        release_after = pick_release_layer(layers_sequence);
        if (release_after == block_conv) release_after = nullptr;
        if (release_after != nullptr) block_pool->enable_alternate_sets();
    }
}

void Sequential::finalize() {
    assert(!finalized);
    wire();
    n_params = 0;
    for (const auto& layer : layers_sequence) n_params += layer->param_count();
    owns_arena = true;
    data_type* p = (data_type*)dev_alloc(sizeof(data_type) * (n_params ? n_params : 1));
    data_type* g = (data_type*)dev_alloc(sizeof(data_type) * (n_params ? n_params : 1));
    param_prev = (data_type*)dev_alloc(sizeof(data_type) * (n_params ? n_params : 1));
    must(cnn_memset_zero(g, sizeof(data_type) * n_params, stream), "cnn_memset_zero");
    bind(p, g);
    finalized = true;
}

void Sequential::finalize(data_type* params_dev, data_type* grads_dev) {
    assert(!finalized && params_dev && grads_dev);
    wire();
    n_params = 0;
    for (const auto& layer : layers_sequence) n_params += layer->param_count();
    owns_arena = false;
    param_prev = (data_type*)dev_alloc(sizeof(data_type) * (n_params ? n_params : 1));
    bind(params_dev, grads_dev);
    finalized = true;
}

Sequential::~Sequential() {
    if (side_tail_pending) {  // (the side stream may still write the arenas freed below)
        cnn_stream_wait_event(stream, ev_side_tail);
        cnn_stream_synchronize(stream);
        side_tail_pending = false;
    }
    if (defer_in_flight || pending_dgrad.valid) {  // (a deferred kernel may still read the layers' buffers)
        pending_dgrad.valid = false;
        cnn_stream_synchronize(defer_stream);
        cnn_stream_synchronize(stream);
    }
    layers_sequence.clear();
    if (param_prev) cnn_device_free(param_prev);
    if (defer_stream) cnn_stream_destroy(defer_stream);
    if (ev_defer_done) cnn_event_destroy(ev_defer_done);
    if (ev_tail) cnn_event_destroy(ev_tail);
    if (ev_side_tail) cnn_event_destroy(ev_side_tail);
    if (owns_arena) {
        cnn_device_free(param_arena);
        cnn_device_free(grad_arena);
    }
    if (loss_terms) cnn_device_free(loss_terms);
    if (loss_sum) cnn_device_free(loss_sum);
    if (bn_comm) cnn_comm_destroy(bn_comm);
    if (comm_stream) cnn_stream_destroy(comm_stream);
    if (ev_grads) cnn_event_destroy(ev_grads);
    if (ev_prep) cnn_event_destroy(ev_prep);
    if (ev_prep_fork) cnn_event_destroy(ev_prep_fork);
    if (ev_comm) cnn_event_destroy(ev_comm);
}

// cnn_conv2d_prepare_filters for all convolutions (instead of one small re-layout launch inside every forward and backward
// call), at most 6 layers per call; possible once every layer has seen its input shape, i.e. from the second forward pass on.
void Sequential::prepare_filters() {
    std::vector<Conv2D*> convs;
    for (auto& layer : layers_sequence)
        if (auto* c = dynamic_cast<Conv2D*>(layer.get())) convs.push_back(c);
    for (auto* c : convs)
        if (!c->shape_known()) return;
    // Deep stacks (11 M parameters in the ResNet-shaped one: 130 us of re-packing per step): only the first layer's images are
    // prepared on the caller's stream; the rest is queued on the library's side stream and runs UNDER the first layers' forward
    // kernels, and the second convolution's forward call waits for it (Conv2D::wait_before_forward).
    size_t later_params = 0;
    for (size_t i = 1; i < convs.size(); ++i) later_params += convs[i]->param_count();
    const bool no_async = cnn_amd_get_option("SYNC_PREPARE", nullptr, 0) == 0;  // (A/B switch, read through the library's option table)
    const bool async = !no_async && convs.size() >= 2 && later_params >= (size_t)1 << 20;
    void* side = nullptr;
    if (async) {
        if (!ev_prep) {
            must(cnn_event_create(&ev_prep), "cnn_event_create");
            must(cnn_event_create(&ev_prep_fork), "cnn_event_create");
        }
        must(cnn_amd_side_stream_get(&side), "cnn_amd_side_stream_get");
        must(cnn_event_record(ev_prep_fork, stream), "cnn_event_record");  // behind the SGD step / the load that changed the parameters
        must(cnn_stream_wait_event(side, ev_prep_fork), "cnn_stream_wait_event");
    }
    for (size_t first = 0; first < convs.size(); first += (async && first == 0) ? 1 : 6) {
        const size_t n = (async && first == 0) ? 1 : std::min<size_t>(6, convs.size() - first);
        std::vector<cnn_conv2d_desc> descs;
        std::vector<const float*> w, b;
        std::vector<void*> f(n), g(n);
        for (size_t i = 0; i < n; ++i) {
            Conv2D* c = convs[first + i];
            descs.push_back(c->current_desc());
            w.push_back(c->filters_dev());
            b.push_back(c->bias_dev());
            c->prepared_buffers(&f[i], &g[i]);
        }
        must(cnn_conv2d_prepare_filters((int)n, descs.data(), w.data(), b.data(), f.data(), g.data(), (async && first > 0) ? side : stream),
             "cnn_conv2d_prepare_filters");
    }
    if (async) {
        must(cnn_event_record(ev_prep, side), "cnn_event_record");
        convs[1]->wait_before_forward(ev_prep);
    }
    for (auto* c : convs) c->set_prepared(true);
    filters_prepared = true;
}

std::vector<tensor> Sequential::forward(const std::vector<tensor>& input) {
    assert(input.size() > 0);
    if (print_info) input[0]->print_shape();
    flush_deferred();
    if (finalized && fuse_layers && !filters_prepared) prepare_filters();
    params_stepped = false;  // (the layers' outputs now belong to the current parameters)
    std::vector<tensor> output(input);
    for (const auto& layer : layers_sequence) {
        output = layer->forward(output);
        if (print_info) output[0]->print_shape();
    }
    return output;
}

void Sequential::flush_deferred() {
    if (side_tail_pending) {  // (the previous step's reductions / SGD / filter images of the later layers, see fused_tail)
        must(cnn_stream_wait_event(stream, ev_side_tail), "cnn_stream_wait_event");
        side_tail_pending = false;
    }
    if (pending_dgrad.valid) {  // not released yet: run it in order on the compute stream
        block_conv->launch_deferred_dgrad(pending_dgrad, stream);
        pending_dgrad.valid = false;
    }
    if (defer_in_flight) {
        must(cnn_stream_wait_event(stream, ev_defer_done), "cnn_stream_wait_event");
        defer_in_flight = false;
    }
}

// gradients [lo, hi) of the arena are final in `stream` order: send them off on the communication stream
void Sequential::flush_bucket(size_t lo, size_t hi) {
    if (hi <= lo) return;
    for (auto& layer : layers_sequence)
        if (auto* lin = dynamic_cast<LinearLayer*>(layer.get())) lin->join_pending(stream);
    must(cnn_amd_side_stream_join(stream), "cnn_amd_side_stream_join");  // (weight gradients of light layers run on the side stream)
    must(cnn_event_record(ev_grads, stream), "cnn_event_record");
    must(cnn_stream_wait_event(comm_stream, ev_grads), "cnn_stream_wait_event");
    must(cnn_allreduce_grads(comm, grad_arena + lo, hi - lo, comm_stream), "cnn_allreduce_grads");
}

void Sequential::backward(std::vector<tensor>& delta_start) {
    if (print_info) delta_start[0]->print_shape();
    flush_deferred();
    const bool force_buckets = cnn_amd_get_option("DP_FORCE_BUCKETS", nullptr, 0) == 0;  // (tests: exercise the path with one rank)
    const bool bucketed = finalized && comm != nullptr && (comm_world > 1 || force_buckets) && n_params >= 2 * bucket_floats;
    size_t pending_hi = n_params, idx = layers_sequence.size();
    for (auto layer = layers_sequence.rbegin(); layer != layers_sequence.rend(); ++layer) {
        delta_start = (*layer)->backward(delta_start);
        if (print_info) delta_start[0]->print_shape();
        --idx;
        if (bucketed && (*layer)->param_count() > 0 && pending_hi - layer_offsets[idx] >= bucket_floats) {
            flush_bucket(layer_offsets[idx], pending_hi);
            pending_hi = layer_offsets[idx];
        }
    }
    grads_reduced = false;
    for (auto& layer : layers_sequence)
        if (auto* lin = dynamic_cast<LinearLayer*>(layer.get())) lin->join_pending(stream);
    if (bucketed) {
        flush_bucket(0, pending_hi);
        must(cnn_event_record(ev_comm, comm_stream), "cnn_event_record");
        // (measurement: how long the compute stream waits here is the EXPOSED part of the exchange -- bench.py reads it at N > 1)
        cnn_amd_timing_span_begin(stream, "span:exchange_wait");
        must(cnn_stream_wait_event(stream, ev_comm), "cnn_stream_wait_event");
        cnn_amd_timing_span_end(stream);
        grads_reduced = true;
        return;
    }
    // the layers' weight gradients were computed on the library's side stream: order them before whatever follows
    must(cnn_amd_side_stream_join(stream), "cnn_amd_side_stream_join");
}

std::vector<uchar> Sequential::grad_cam(const std::string& layer_name, std::vector<data_type>* cam_out) const {
    // alexnet.cpp:97-102: from the logits down to (not including) the named layer.  The reference computes this delta and never
    // uses it (its channel weights are means of the FEATURE MAP, :111-119); the walk is kept for its side effects on the layers.
    const_cast<Sequential*>(this)->flush_deferred();  // (the walk below rewrites the delta a deferred data gradient reads)
    std::vector<tensor> delta = layers_sequence.back()->get_output();
    auto layer = layers_sequence.rbegin();
    for (; layer != layers_sequence.rend(); ++layer) {
        if ((*layer)->name == layer_name) break;
        delta = (*layer)->backward(delta);
    }
    assert(layer != layers_sequence.rend() && "grad_cam: no layer of that name");
    must(cnn_amd_side_stream_join(stream), "cnn_amd_side_stream_join");
    const std::vector<tensor> feature_map = (*layer)->get_output();  // alexnet.cpp:105
    const int B = (int)feature_map.size(), C = feature_map[0]->C, H = feature_map[0]->H, W = feature_map[0]->W;
    BatchBuffer staging;
    const data_type* fea = batch_device_pointer(feature_map, staging, "grad_cam");
    void* cam_dev = nullptr;
    void* img_dev = nullptr;
    must(cnn_device_alloc(&cam_dev, sizeof(data_type) * (size_t)B * H * W), "cnn_device_alloc");
    must(cnn_device_alloc(&img_dev, (size_t)H * W), "cnn_device_alloc");
    must(cnn_grad_cam(fea, B, C, H, W, (data_type*)cam_dev, (unsigned char*)img_dev, stream), "cnn_grad_cam");
    std::vector<uchar> image((size_t)H * W);
    must(cnn_memcpy_d2h(image.data(), img_dev, image.size(), stream), "cnn_memcpy_d2h");
    if (cam_out) {
        cam_out->resize((size_t)B * H * W);
        must(cnn_memcpy_d2h(cam_out->data(), cam_dev, sizeof(data_type) * cam_out->size(), stream), "cnn_memcpy_d2h");
    }
    must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
    must(cnn_device_free(cam_dev), "cnn_device_free");
    must(cnn_device_free(img_dev), "cnn_device_free");
    return image;
}

void Sequential::invalidate_filter_images() {
    filters_prepared = false;  // re-prepared at the start of the next forward pass
    for (auto& layer : layers_sequence)
        if (auto* c = dynamic_cast<Conv2D*>(layer.get())) c->set_prepared(false);
}

void Sequential::parameters_changed() {
    params_stepped = false;    // (an outside write: there is no snapshot of what the last forward pass used)
    invalidate_filter_images();
    for (auto& layer : layers_sequence) {
        if (auto* c = dynamic_cast<Conv2D*>(layer.get())) c->params_of_last_forward_lost();
        if (auto* bn = dynamic_cast<BatchNorm2D*>(layer.get())) bn->params_of_last_forward_lost();
    }
}

void Sequential::set_comm(void* rccl_comm, int world) {
    assert(world >= 1);
    comm = rccl_comm;
    comm_world = rccl_comm ? world : 1;
    int rank = 0;
    if (comm != nullptr) {
        int w = 0;
        must(cnn_comm_info(comm, &w, &rank), "cnn_comm_info");
        assert(w == world && "Sequential::set_comm: world does not match the communicator");
    }
    // BatchNorm2D's sync-BN reductions are issued on the compute stream in the middle of the backward walk while buckets of the gradient
    // exchange are in flight on the communication stream.  DEFAULT (round 5, VERDICT / ADVICE r4): they share `comm` -- one communicator,
    // whose collectives RCCL serialises in issue order, identical on every rank.  CNN_AMD_BN_OWN_COMM=1 gives them a communicator of
    // their own over the same ranks (cnn_comm_split): two queues instead of one, but two communicators whose kernels are in flight
    // together are only safe when both can be co-resident on every rank (host issue order is the same everywhere, device execution order
    // is not guaranteed -- the pattern RCCL documents as deadlock-prone), and the split itself is a blocking collective inside set_comm
    // (one host thread or process per rank).  It has never run on more than one GPU: opt-in until it has.
    if (bn_comm != nullptr) {
        cnn_comm_destroy(bn_comm);
        bn_comm = nullptr;
    }
    bool has_bn = false;
    for (auto& layer : layers_sequence) has_bn = has_bn || dynamic_cast<BatchNorm2D*>(layer.get()) != nullptr;
    char text[8] = {0};
    const bool own = cnn_amd_get_option("BN_OWN_COMM", text, sizeof(text)) == 0 && std::atoi(text) != 0;
    if (comm != nullptr && has_bn && own && cnn_comm_split(comm, 0, rank, &bn_comm) != CNN_AMD_OK) bn_comm = nullptr;
    for (auto& layer : layers_sequence) {
        if (auto* bn = dynamic_cast<BatchNorm2D*>(layer.get())) bn->set_comm(bn_comm ? bn_comm : comm, comm_world);
        if (auto* conv = dynamic_cast<Conv2D*>(layer.get())) conv->set_comm(comm, comm_world, rank);
    }
    if (comm && !comm_stream) {
        must(cnn_stream_create(&comm_stream), "cnn_stream_create");
        must(cnn_event_create(&ev_grads), "cnn_event_create");
        must(cnn_event_create(&ev_comm), "cnn_event_create");
    }
}

// C1 of SURVEY.md section 2.1: ONE in-place fp32 sum over the flat gradient arena (RCCL over xGMI), on the communication
// stream, fenced against the compute stream by two events
// a communicator with more than one rank -- or, for tests on one GPU, any communicator with the DP_FORCE_EXCHANGE switch set (the
// sums are then identities and the step must equal the plain one bit for bit)
bool Sequential::exchange_active() const {
    if (comm == nullptr) return false;
    return comm_world > 1 || cnn_amd_get_option("DP_FORCE_EXCHANGE", nullptr, 0) == 0;
}

void Sequential::allreduce_gradients() {
    if (!exchange_active() || grads_reduced) return;
    assert(finalized && "data parallelism needs the flat gradient arena: call finalize()");
    must(cnn_event_record(ev_grads, stream), "cnn_event_record");
    must(cnn_stream_wait_event(comm_stream, ev_grads), "cnn_stream_wait_event");
    must(cnn_allreduce_grads(comm, grad_arena, n_params, comm_stream), "cnn_allreduce_grads");
    must(cnn_event_record(ev_comm, comm_stream), "cnn_event_record");
    cnn_amd_timing_span_begin(stream, "span:exchange_wait");
    must(cnn_stream_wait_event(stream, ev_comm), "cnn_stream_wait_event");
    cnn_amd_timing_span_end(stream);
    grads_reduced = true;
}

void Sequential::update_gradients(const data_type learning_rate) {
    flush_deferred();
    if (!finalized) {  // plain list of stand-alone layers: the reference's loop (alexnet.cpp:62-65)
        for (auto& layer : layers_sequence) layer->update_gradients(learning_rate);
        return;
    }
    if (exchange_active()) {
        allreduce_gradients();
        update_gradients(learning_rate, 1.f / (data_type)comm_world);
    } else {
        update_gradients(learning_rate, 1.f);
    }
}

void Sequential::update_gradients(const data_type learning_rate, const data_type grad_scale) {
    assert(finalized && "the grad_scale form works on the flat arena: call finalize()");
    // (the old values go to the snapshot: Conv2D::get_output() of a fused-away tensor re-computes it with them)
    if (params_stepped)  // a second step without a forward pass in between: the snapshot would now receive already-stepped values
        for (auto& layer : layers_sequence) {
            if (auto* c = dynamic_cast<Conv2D*>(layer.get())) c->params_of_last_forward_lost();
            if (auto* bn = dynamic_cast<BatchNorm2D*>(layer.get())) bn->params_of_last_forward_lost();
        }
    must(cnn_sgd_update_keep(param_arena, grad_arena, n_params, learning_rate, grad_scale, param_prev, stream), "cnn_sgd_update_keep");
    invalidate_filter_images();
    params_stepped = true;
}

// filter images of every convolution behind the pool-fused front block, from the current parameters, on `on_stream`
void Sequential::prepare_later_filters(void* on_stream) {
    std::vector<Conv2D*> later;
    for (auto& layer : layers_sequence)
        if (auto* c = dynamic_cast<Conv2D*>(layer.get()))
            if (c != block_conv) later.push_back(c);
    for (size_t first = 0; first < later.size(); first += 6) {
        const size_t n = std::min<size_t>(6, later.size() - first);
        std::vector<cnn_conv2d_desc> descs;
        std::vector<const float*> w, b;
        std::vector<void*> f(n), g(n);
        for (size_t i = 0; i < n; ++i) {
            Conv2D* c = later[first + i];
            descs.push_back(c->current_desc());
            w.push_back(c->filters_dev());
            b.push_back(c->bias_dev());
            c->prepared_buffers(&f[i], &g[i]);
        }
        must(cnn_conv2d_prepare_filters((int)n, descs.data(), w.data(), b.data(), f.data(), g.data(), on_stream), "cnn_conv2d_prepare_filters");
    }
}

// The end of a train step whose first block ran pool-fused (profiles/NOTEBOOK.md section 4.13), called where the backward walk reaches that
// block's convolution; `delta` = d(pool output).  Instead of [block wgrad || block dgrad] -> join -> (all-reduce) -> SGD over the
// arena -> filter images:
//   * side stream, behind the data gradient of the layer behind the block (the last reader of the later layers' filter images):
//     the recorded slab reductions of those layers, [their share of the gradient exchange,] their SGD step and filter images --
//     all of it UNDER the block's weight-gradient kernel on the compute stream;
//   * compute stream: the block's weight gradient; single rank: its slab reduction, SGD step and filter images are ONE more small
//     launch; with a communicator: reduce, all-reduce of this layer's few floats, SGD, filter images;
//   * the block's data gradient (no consumer: alexnet.cpp:55) is only RECORDED here: the next forward pass releases it on a third
//     stream behind its release layer (flush_deferred() otherwise), with the filters and pooled-domain tensors of ITS step.
// Arithmetic and summation orders are those of the plain sequence: parameters come out bit-identical.
bool Sequential::fused_tail(std::vector<tensor>& delta, const data_type learning_rate) {
    if (!finalized || !fuse_layers || block_conv == nullptr || release_after == nullptr || !block_conv->pool_fused_pending() ||
        !filters_prepared || layer_offsets.empty() || layer_offsets[0] != 0)
        return false;
    if (cnn_amd_get_option("NO_FUSED_TAIL", nullptr, 0) == 0) return false;  // (A/B switch)
    const size_t lo = block_conv->param_count();  // the block's convolution owns arena[0, lo)
    const bool dp = exchange_active();
    const data_type scale = dp ? 1.f / (data_type)comm_world : 1.f;
    if (ev_tail == nullptr) must(cnn_event_create(&ev_tail), "cnn_event_create");
    void* side = nullptr;
    must(cnn_amd_side_stream_get(&side), "cnn_amd_side_stream_get");
    // TAIL_BEHIND_BLOCK (single rank, own arena): the later layers' reductions / SGD / filter images do not run under the block's
    // weight gradient but BEHIND it, and the compute stream does not wait for them here: they overlap the next step's first
    // forward kernel, whose successor waits (train_step / flush_deferred)
    const bool tail_behind = !dp && owns_arena && cnn_amd_get_option("TAIL_BEHIND_BLOCK", nullptr, 0) == 0;
    if (tail_behind) {
        if (ev_side_tail == nullptr) must(cnn_event_create(&ev_side_tail), "cnn_event_create");
        pending_dgrad = block_conv->backward_weight_pooled(delta, /*fused_sgd=*/true, learning_rate, scale);
        block_conv->set_delta_computed(input_gradient);
        if (!input_gradient) pending_dgrad.valid = false;
        must(cnn_event_record(ev_tail, stream), "cnn_event_record");
        must(cnn_stream_wait_event(side, ev_tail), "cnn_stream_wait_event");
        for (auto& layer : layers_sequence)
            if (auto* lin = dynamic_cast<LinearLayer*>(layer.get())) lin->join_pending(side);
        must(cnn_amd_flush_reduces(side), "cnn_amd_flush_reduces");
        if (n_params > lo)
            must(cnn_sgd_update_keep(param_arena + lo, grad_arena + lo, n_params - lo, learning_rate, scale, param_prev + lo, side),
                 "cnn_sgd_update_keep");
        prepare_later_filters(side);
        must(cnn_event_record(ev_side_tail, side), "cnn_event_record");
        side_tail_pending = true;
        grads_reduced = false;
        params_stepped = true;
        return true;
    }
    if (cnn_amd_published_is_last(stream)) {  // the data gradient just launched carries the event in its dispatch packet
        must(cnn_amd_wait_published(side), "cnn_amd_wait_published");
    } else {
        must(cnn_event_record(ev_tail, stream), "cnn_event_record");
        must(cnn_stream_wait_event(side, ev_tail), "cnn_stream_wait_event");
    }
    // DX0_AT_TAIL (A/B switch): the block's data gradient starts right here, beside its weight gradient (they read the same dpool /
    // mask lines), on the third stream -- instead of in the next forward pass
    const bool dgrad_now = cnn_amd_get_option("DX0_AT_TAIL", nullptr, 0) == 0;
    if (dgrad_now) {
        if (defer_stream == nullptr) {
            must(cnn_stream_create(&defer_stream), "cnn_stream_create");
            must(cnn_event_create(&ev_defer_done), "cnn_event_create");
        }
        if (cnn_amd_published_is_last(stream)) must(cnn_amd_wait_published(defer_stream), "cnn_amd_wait_published");
        else must(cnn_stream_wait_event(defer_stream, ev_tail), "cnn_stream_wait_event");
    }
    for (auto& layer : layers_sequence)  // (a loss-head pass: the linear layer's weight / bias gradient runs on its own stream)
        if (auto* lin = dynamic_cast<LinearLayer*>(layer.get())) lin->join_pending(side);
    must(cnn_amd_flush_reduces(side), "cnn_amd_flush_reduces");
    if (n_params > lo) {
        // bucket 1 of the exchange: everything behind the block is final ~one weight-gradient kernel before the step ends
        if (dp) must(cnn_allreduce_grads(comm, grad_arena + lo, n_params - lo, side), "cnn_allreduce_grads");
        must(cnn_sgd_update_keep(param_arena + lo, grad_arena + lo, n_params - lo, learning_rate, scale, param_prev + lo, side),
             "cnn_sgd_update_keep");
    }
    prepare_later_filters(side);
    // compute stream: the block's weight gradient (+ its share of the tail)
    pending_dgrad = block_conv->backward_weight_pooled(delta, /*fused_sgd=*/!dp, learning_rate, scale);
    block_conv->set_delta_computed(input_gradient);
    if (!input_gradient) pending_dgrad.valid = false;  // (architectures::input_gradient: nobody wants d(loss) / d(input image))
    if (dgrad_now && pending_dgrad.valid) {
        block_conv->launch_deferred_dgrad(pending_dgrad, defer_stream);
        must(cnn_event_record(ev_defer_done, defer_stream), "cnn_event_record");
        pending_dgrad.valid = false;
        defer_in_flight = true;
    }
    if (dp) {  // bucket 2: this layer's few floats
        must(cnn_allreduce_grads(comm, grad_arena, lo, stream), "cnn_allreduce_grads");
        must(cnn_sgd_update_keep(param_arena, grad_arena, lo, learning_rate, scale, param_prev, stream), "cnn_sgd_update_keep");
        block_conv->prepare_own_filters();
    }
    must(cnn_amd_side_stream_join(stream), "cnn_amd_side_stream_join");
    grads_reduced = dp;
    params_stepped = true;  // (filters_prepared stays true: every image above was made from the updated parameters)
    return true;
}

// cnn.cpp:79-90 without leaving the device
void Sequential::train_step(const std::vector<tensor>& input, const int* labels_dev, const data_type learning_rate) {
    assert(!input.empty() && labels_dev != nullptr && !layers_sequence.empty());
    auto* head = dynamic_cast<LinearLayer*>(layers_sequence.back().get());
    assert(head != nullptr && "train_step: the last layer must be a LinearLayer (the logits)");
    const int B = (int)input.size();
    const int classes = head->out_features();
    if (loss_delta.empty() || loss_batch < B) {
        assert(loss_delta.empty() && "train_step: batch larger than the first call's");
        loss_probs.allocate(B, classes, 1, 1, "probs");
        loss_delta.allocate(B, classes, 1, 1, "loss_delta");
        loss_terms = (data_type*)dev_alloc(sizeof(data_type) * B);
        loss_sum = (data_type*)dev_alloc(sizeof(data_type));
        loss_batch = B;
    }
    if (print_info) input[0]->print_shape();
    if (finalized && fuse_layers && !filters_prepared) {
        flush_deferred();  // (a full re-preparation rewrites filter images a pending data gradient may read)
        prepare_filters();
    }
    // a data gradient deferred by the previous step reads the pool's output / mask set of THAT step.  A pass that does
    // not run the block pool-fused (smaller batch, fuse_pool_block off) goes through MaxPool2D::forward, which always rewrites
    // set 0 on the compute stream: nothing would order the two -- run the pending kernel first, in order.
    if (pending_dgrad.valid && (block_conv == nullptr || !block_conv->next_pass_pool_fused((int)input.size()))) flush_deferred();
    params_stepped = false;
    const bool was_lazy = lazy_host_sync;
    lazy_host_sync = true;
    std::vector<tensor> output(input);
    const bool fused_head = fuse_layers && head->loss_head_supported();
    const bool head_dx = fused_head && cnn_amd_get_option("NO_HEAD_DX", nullptr, 0) != 0;  // (A/B switch)
    Layer* release_layer = release_after;
    for (const auto& layer : layers_sequence) {
        // the previous step's deferred data gradient starts behind this layer's forward kernel, on its own stream
        const bool release_here = pending_dgrad.valid && layer.get() == release_layer;
        if (side_tail_pending && layer.get() == behind_block) {  // the first layer whose parameters / filter images the side tail writes
            must(cnn_stream_wait_event(stream, ev_side_tail), "cnn_stream_wait_event");
            side_tail_pending = false;
        }
        if (release_here) must(cnn_amd_publish_next_kernel(stream), "cnn_amd_publish_next_kernel");
        if (layer.get() == head && fused_head)
            output = head->forward_loss_head(output, labels_dev, loss_probs.base, loss_delta.base, loss_terms, head_dx);
        else
            output = layer->forward(output);
        if (release_here) {
            if (defer_stream == nullptr) {
                char prio[16] = {0};
                const int level = cnn_amd_get_option("DEFER_PRIO", prio, sizeof(prio)) == 0 ? std::atoi(prio) : 0;  // (measurement switch)
                must(cnn_stream_create_priority(&defer_stream, level), "cnn_stream_create_priority");
                must(cnn_event_create(&ev_defer_done), "cnn_event_create");
            }
            if (cnn_amd_published_is_last(stream)) {
                must(cnn_amd_wait_published(defer_stream), "cnn_amd_wait_published");
            } else {
                if (ev_tail == nullptr) must(cnn_event_create(&ev_tail), "cnn_event_create");
                must(cnn_event_record(ev_tail, stream), "cnn_event_record");
                must(cnn_stream_wait_event(defer_stream, ev_tail), "cnn_stream_wait_event");
            }
            block_conv->launch_deferred_dgrad(pending_dgrad, defer_stream);
            must(cnn_event_record(ev_defer_done, defer_stream), "cnn_event_record");
            pending_dgrad.valid = false;
            defer_in_flight = true;
        }
        if (print_info) output[0]->print_shape();
    }
    lazy_host_sync = was_lazy;
    if (!fused_head) {
        const data_type* logits = batch_device_pointer(output, logits_stage, "logits");
        must(cnn_softmax_xent(logits, labels_dev, loss_probs.base, loss_delta.base, loss_sum, B, classes, stream), "cnn_softmax_xent");
    }
    loss_in_terms = fused_head;
    loss_last_B = B;
    std::vector<tensor> delta(loss_delta.views.begin(), loss_delta.views.begin() + B);
    if (pending_dgrad.valid) flush_deferred();  // (never released: no release layer in this pass)
    if (block_conv != nullptr && block_conv->pool_fused_pending() && release_after != nullptr) {
        // backward walk with the fused tail at the front block (see fused_tail)
        for (auto layer = layers_sequence.rbegin(); layer != layers_sequence.rend(); ++layer) {
            if (layer->get() == behind_block && defer_in_flight) {
                // the deferred data gradient of the PREVIOUS step reads d(pool output), which this layer's backward rewrites (only
                // this stream's next kernel depends on it, not the weight gradient forked off beside it)
                must(cnn_stream_wait_event_local(stream, ev_defer_done), "cnn_stream_wait_event_local");
                defer_in_flight = false;
            }
            if (layer->get() == block_conv && fused_tail(delta, learning_rate)) return;
            delta = (*layer)->backward(delta);
            if (print_info) delta[0]->print_shape();
        }
        grads_reduced = false;
        for (auto& layer : layers_sequence)
            if (auto* lin = dynamic_cast<LinearLayer*>(layer.get())) lin->join_pending(stream);
        must(cnn_amd_side_stream_join(stream), "cnn_amd_side_stream_join");
        update_gradients(learning_rate);  // (flush_deferred() inside: a deferred kernel still in flight is ordered here)
        return;
    }
    backward(delta);
    update_gradients(learning_rate);
}

data_type Sequential::last_loss() {
    assert(loss_batch > 0 && "last_loss before the first train_step");
    if (loss_in_terms) {  // ordered sum over the per-sample terms (the reference's order, func.cpp:60-71)
        must(cnn_loss_from_terms(loss_terms, loss_sum, loss_last_B, stream), "cnn_loss_from_terms");
        loss_in_terms = false;
    }
    data_type host = 0;
    must(cnn_memcpy_d2h(&host, loss_sum, sizeof(data_type), stream), "cnn_memcpy_d2h");
    must(cnn_stream_synchronize(stream), "cnn_stream_synchronize");
    return host / (data_type)loss_last_B;
}

void Sequential::save_weights(const std::filesystem::path& save_path) const {
    const_cast<Sequential*>(this)->flush_deferred();
    std::ofstream writer(save_path.c_str(), std::ios::binary);
    for (const auto& layer : layers_sequence) layer->save_weights(writer);
    std::cout << "weights have been saved to " << save_path.string() << std::endl;
    writer.close();
}

void Sequential::load_weights(const std::filesystem::path& checkpoint_path) {
    if (!std::filesystem::exists(checkpoint_path)) {  // alexnet.cpp:81-84: report and carry on
        std::cout << "checkpoint file  " << checkpoint_path << " does not exist !\n";
        return;
    }
    std::ifstream reader(checkpoint_path.c_str(), std::ios::binary);
    flush_deferred();
    parameters_changed();
    for (auto& layer : layers_sequence) layer->load_weights(reader);
    std::cout << "load weights from" << checkpoint_path.string() << std::endl;
    reader.close();
}

// ---------------------------------------------------------------------------------------------------------------
// BASELINE.json configs[3] / [4] (mirrored by cnn_amd/stacks.py, which the parity tests build the oracle from)
namespace {
struct StackBuilder {
    Sequential& net;
    bool batch_norm;
    int C = 3, H = 224, W = 224, n_conv = 0, n_pool = 0;
    void conv(int co, int k, int s, int pad) {
        const std::string id = std::to_string(++n_conv);
        net.add(new Conv2D("conv_layer_" + id, C, co, k, s, pad));
        if (batch_norm) net.add(new BatchNorm2D("bn_layer_" + id, co));
        net.add(new ReLU("relu_layer_" + id));
        C = co;
        H = cnn_conv2d_out_dim(H, k, s, pad);
        W = cnn_conv2d_out_dim(W, k, s, pad);
    }
    void pool(int k, int step) {
        net.add(new MaxPool2D("max_pool_" + std::to_string(++n_pool), k, step));
        H = cnn_maxpool2d_out_dim(H, k, step);
        W = cnn_maxpool2d_out_dim(W, k, step);
    }
    void linear(int out) { net.add(new LinearLayer("linear_1", C * H * W, out)); }
};
}  // namespace

void architectures::build_vgg11(Sequential& net, const int num_classes, const bool batch_norm) {
    StackBuilder b{net, batch_norm};
    const int chans[8] = {64, 128, 256, 256, 512, 512, 512, 512};
    for (int i = 1; i <= 8; ++i) {
        b.conv(chans[i - 1], 3, 1, 1);
        if (i == 1 || i == 2 || i == 4 || i == 6 || i == 8) b.pool(2, 2);
    }
    b.linear(num_classes);
}

void architectures::build_resnet18(Sequential& net, const int num_classes, const bool batch_norm) {
    StackBuilder b{net, batch_norm};
    b.conv(64, 7, 2, 3);
    b.pool(2, 2);
    for (int i = 0; i < 4; ++i) b.conv(64, 3, 1, 1);
    b.conv(128, 3, 2, 1);
    for (int i = 0; i < 3; ++i) b.conv(128, 3, 1, 1);
    b.conv(256, 1, 2, 0);
    for (int i = 0; i < 3; ++i) b.conv(256, 3, 1, 1);
    b.conv(512, 3, 2, 1);
    for (int i = 0; i < 3; ++i) b.conv(512, 3, 1, 1);
    b.linear(num_classes);
}
