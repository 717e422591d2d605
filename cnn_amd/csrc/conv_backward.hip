// conv_backward.hip -- Conv2D::backward (cpu/src/conv2d.cpp:97-202) as ONE call: the weight/bias-gradient kernels
// (conv2d.cpp:117-159) and the data-gradient kernels (conv2d.cpp:168-199) only share their inputs, and for the
// reference net's layers each of them alone is latency-bound, so they run CONCURRENTLY: fork an internal side stream
// off the caller's stream with an event, join it back with a second event.  The pattern is capturable in a hipGraph.
#include "common.h"

using namespace cnn_amd;

namespace cnn_amd {
size_t igemm_workspace_floats(const cnn_conv2d_desc* d);  // conv_igemm.hip
void wgrad_defer_reduce(bool on);                          // conv_wgrad.hip: with defer_join the final slab reductions of the
int wgrad_flush_reduces(hipStream_t s);                    // weight gradients run in one launch just before the join
bool thin_dgrad_supported(const cnn_conv2d_desc* d);       // conv_dgrad_thin.hip
}

namespace {
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr, lead = nullptr;
    int device = -1;
};
int get_side(SideStream** out) {
    // one side stream per (host thread, device): a thread that alternates between devices keeps both (re-creating on every
    // switch leaked a stream + two events each time)
    static thread_local SideStream sides[16];
    int dev = 0;
    CNN_HIP_CHECK(hipGetDevice(&dev));
    SideStream& side = sides[(dev >= 0 ? dev : 0) % 16];
    if (side.stream == nullptr || side.device != dev) {
        void* made = nullptr;
        if (int rc = cnn_stream_create_priority(&made, CNN_OPT_INT("SIDE_PRIO", 0))) return rc;  // (SIDE_PRIO: measurement switch)
        side.stream = as_stream(made);
        CNN_HIP_CHECK(hipEventCreateWithFlags(&side.fork, hipEventDisableTiming));
        CNN_HIP_CHECK(hipEventCreateWithFlags(&side.join, hipEventDisableTiming));
        CNN_HIP_CHECK(hipEventCreateWithFlags(&side.lead, hipEventDisableTiming));
        side.device = dev;
    }
    *out = &side;
    return CNN_AMD_OK;
}
// Rounds 3-4: layers whose two gradient kernels are each MFMA-bound and fill the chip for a long time (>= 20 GFLOP: the VGG-shaped stack)
// gained nothing from running side by side -- the implicit GEMM and the register-direct weight gradient halved each other's rate and
// thrashed each other's L2 -- so their weight gradient stayed on the caller's stream, behind the data gradient.  Round 5: with both
// gradients LDS-staged (conv_rows / wgrad_sp: one workgroup per CU each, operands from L2 once) the pair fills each other's tails instead:
// VGG-shaped step 2 579 -> 2 631 images/s with every layer forked (measured at limits 20 / 130 / 250 / 1000 GFLOP: 2 579 / 2 579 / 2 611 /
// 2 631).  The switch stays (SERIAL_BWD_GFLOP=<limit>; 0 = every layer serial); the default no longer serialises any layer.
bool heavy_layer(const cnn_conv2d_desc* d) {
    const double Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, d->pad), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, d->pad);
    const double flops = 2.0 * d->B * d->Co * Ho * Wo * d->Ci * d->k * d->k;
    const double limit = CNN_OPT("SERIAL_BWD_GFLOP").as_double(1e9) * 1e9;
    return flops >= limit;
}
// the side stream waits for everything queued on `main` so far.  When the last thing the library launched on `main` is a published
// kernel (common.h) its dispatch event is that point already: no marker packet on the critical stream.
int fork_side(SideStream* side, hipStream_t main) {
    const PublishState& p = publish_state();
    if (p.valid && !p.stale && p.stream == main) {
        CNN_HIP_CHECK(hipStreamWaitEvent(side->stream, p.ev[p.cur], 0));
        return CNN_AMD_OK;
    }
    CNN_HIP_CHECK(hipEventRecord(side->fork, main));
    CNN_HIP_CHECK(hipStreamWaitEvent(side->stream, side->fork, 0));
    return CNN_AMD_OK;
}
size_t dgrad_region_bytes(const cnn_conv2d_desc* d) { return ((igemm_workspace_floats(d) + 63) / 64) * 64 * sizeof(float); }
}  // namespace

extern "C" {

size_t cnn_conv2d_backward_workspace_bytes(const cnn_conv2d_desc* d) {
    const size_t w = cnn_conv2d_workspace_bytes(d);  // upper bound for the weight-gradient part
    if (w == 0) return 0;
    return dgrad_region_bytes(d) + w;
}

int cnn_amd_side_stream_get(void** side_stream) {
    CNN_REQUIRE(side_stream != nullptr, "cnn_amd_side_stream_get: null pointer");
    SideStream* side = nullptr;
    if (int rc = get_side(&side)) return rc;
    *side_stream = (void*)side->stream;
    return CNN_AMD_OK;
}

int cnn_amd_side_stream_join(void* stream) {
    SideStream* side = nullptr;
    if (int rc = get_side(&side)) return rc;
    // the recorded reductions: on the side stream in front of the join (concurrent with the caller's last kernels: starved,
    // ~100 us instead of 8, but off the critical path: 491k vs 480k images/s) or, with CNN_AMD_REDUCE_ON_MAIN=1, on the
    // caller's stream behind it
    const bool on_side = !((CNN_OPT_SET("REDUCE_ON_MAIN") && CNN_OPT_INT("REDUCE_ON_MAIN", 0) != 0));
    if (on_side)
        if (int rc = wgrad_flush_reduces(side->stream)) return rc;
    CNN_HIP_CHECK(hipEventRecord(side->join, side->stream));
    CNN_HIP_CHECK(hipStreamWaitEvent(as_stream(stream), side->join, 0));
    publish_mark_stale(as_stream(stream));  // (a fork taken from an earlier published kernel would miss the side stream's work)
    if (!on_side)
        if (int rc = wgrad_flush_reduces(as_stream(stream))) return rc;
    return CNN_AMD_OK;
}

int cnn_conv2d_backward(const cnn_conv2d_desc* d, const float* x, const float* dy, const float* w, float* gw, float* gb,
                        float* dx, float divisor, void* ws, size_t ws_bytes, void* stream, int defer_join) {
    CNN_REQUIRE(d && x && dy && w && gw && dx && ws, "cnn_conv2d_backward: null pointer");
    const size_t dbytes = dgrad_region_bytes(d);
    if (ws_bytes < dbytes + 256)
        return fail(CNN_AMD_E_WORKSPACE, "cnn_conv2d_backward: workspace %zu B too small", ws_bytes);
    SideStream* side = nullptr;
    if (int rc = get_side(&side)) return rc;
    hipStream_t main = as_stream(stream);
    char* base = (char*)ws;
    if (heavy_layer(d)) {
        if (int rc = cnn_conv2d_backward_data(d, dy, w, dx, base, dbytes, main)) return rc;
        return cnn_conv2d_backward_weight(d, x, dy, gw, gb, divisor, base + dbytes, ws_bytes - dbytes, main);
    }
    if (int rc = fork_side(side, main)) return rc;
    wgrad_defer_reduce(defer_join != 0);
    const int rcw = cnn_conv2d_backward_weight(d, x, dy, gw, gb, divisor, base + dbytes, ws_bytes - dbytes, side->stream);
    wgrad_defer_reduce(false);
    if (rcw) return rcw;
    if (int rc = cnn_conv2d_backward_data(d, dy, w, dx, base, dbytes, main)) return rc;
    if (!defer_join) {
        CNN_HIP_CHECK(hipEventRecord(side->join, side->stream));
        CNN_HIP_CHECK(hipStreamWaitEvent(main, side->join, 0));
    }
    return CNN_AMD_OK;
}

int cnn_conv2d_backward_prepared(const cnn_conv2d_desc* d, const float* x, const float* dy, const void* prepared_dgrad,
                                 float* gw, float* gb, float* dx, float divisor, void* ws, size_t ws_bytes, void* stream,
                                 int defer_join) {
    return cnn_conv2d_backward_prepared_relu(d, x, dy, prepared_dgrad, nullptr, gw, gb, dx, divisor, ws, ws_bytes, stream, defer_join);
}

int cnn_conv2d_backward_prepared_relu(const cnn_conv2d_desc* d, const float* x, const float* dy, const void* prepared_dgrad,
                                      const float* relu_below, float* gw, float* gb, float* dx, float divisor, void* ws,
                                      size_t ws_bytes, void* stream, int defer_join) {
    CNN_REQUIRE(d && x && dy && prepared_dgrad && gw && dx && ws, "cnn_conv2d_backward_prepared: null pointer");
    SideStream* side = nullptr;
    if (int rc = get_side(&side)) return rc;
    hipStream_t main = as_stream(stream);
    if (heavy_layer(d)) {
        if (int rc = relu_below ? cnn_conv2d_backward_data_relu_prepared(d, dy, prepared_dgrad, relu_below, dx, main)
                                : cnn_conv2d_backward_data_prepared(d, dy, prepared_dgrad, dx, main))
            return rc;
        return cnn_conv2d_backward_weight(d, x, dy, gw, gb, divisor, ws, ws_bytes, main);
    }
    if (int rc = fork_side(side, main)) return rc;
    // A THIN layer's data gradient (Ci = 3: a VALU kernel of thousands of small waves) takes every wave slot of the chip if it is
    // dispatched first, and the MFMA weight gradient beside it then trickles in as those waves retire (the 7x7 stem of the ResNet-shaped
    // stack: 490 us for a 200 us kernel).  The data gradient therefore waits for the side stream to have reached the weight gradient's
    // launch: its workgroups take their CUs first, the VALU waves fill what is left (ResNet-shaped step 7 830 -> 7 910 images/s; the 3x3
    // stride-1 first layer of the VGG-shaped stack, whose weight gradient is register-direct, lost 0.3 % and keeps the old order).
    // THIN_DGRAD_LEAD=0: both at once as before.
    const bool lead = thin_dgrad_supported(d) && d->s == 2 && CNN_OPT_INT("THIN_DGRAD_LEAD", 1) != 0;
    if (lead) CNN_HIP_CHECK(hipEventRecord(side->lead, side->stream));
    wgrad_defer_reduce(defer_join != 0);
    const int rcw = cnn_conv2d_backward_weight(d, x, dy, gw, gb, divisor, ws, ws_bytes, side->stream);
    wgrad_defer_reduce(false);
    if (rcw) return rcw;
    if (lead) CNN_HIP_CHECK(hipStreamWaitEvent(main, side->lead, 0));
    if (int rc = relu_below ? cnn_conv2d_backward_data_relu_prepared(d, dy, prepared_dgrad, relu_below, dx, main)
                            : cnn_conv2d_backward_data_prepared(d, dy, prepared_dgrad, dx, main))
        return rc;
    if (!defer_join) {
        CNN_HIP_CHECK(hipEventRecord(side->join, side->stream));
        CNN_HIP_CHECK(hipStreamWaitEvent(main, side->join, 0));
    }
    return CNN_AMD_OK;
}

/* the weight / bias gradient half of cnn_conv2d_backward*(defer_join = 1) alone: forked off `stream` onto the side stream, final slab
 * reduction recorded for the join -- for callers that obtain the data gradients of several layers another way */
int cnn_conv2d_backward_weight_side(const cnn_conv2d_desc* d, const float* x, const float* dy, float* gw, float* gb, float divisor, void* ws,
                                    size_t ws_bytes, void* stream) {
    CNN_REQUIRE(d && x && dy && gw && ws, "cnn_conv2d_backward_weight_side: null pointer");
    SideStream* side = nullptr;
    if (int rc = get_side(&side)) return rc;
    hipStream_t main = as_stream(stream);
    if (heavy_layer(d)) return cnn_conv2d_backward_weight(d, x, dy, gw, gb, divisor, ws, ws_bytes, main);
    if (int rc = fork_side(side, main)) return rc;
    wgrad_defer_reduce(true);
    const int rcw = cnn_conv2d_backward_weight(d, x, dy, gw, gb, divisor, ws, ws_bytes, side->stream);
    wgrad_defer_reduce(false);
    return rcw;
}

/* Conv2D::backward of the pool-fused first block: weight / bias gradient on the side stream, data gradient on `stream`,
 * both rebuilt from the pooled domain (cnn_conv2d_backward_weight_pooled2 / _data_pooled2_prepared) */
int cnn_conv2d_backward_pooled2_prepared(const cnn_conv2d_desc* d, const float* x, const float* dpool, const int32_t* mask,
                                         const float* pooled, const void* prepared_dgrad, float* gw, float* gb, float* dx,
                                         float divisor, void* ws, size_t ws_bytes, void* stream, int defer_join) {
    CNN_REQUIRE(d && x && dpool && mask && prepared_dgrad && gw && dx && ws, "cnn_conv2d_backward_pooled2_prepared: null pointer");
    SideStream* side = nullptr;
    if (int rc = get_side(&side)) return rc;
    hipStream_t main = as_stream(stream);
    if (int rc = fork_side(side, main)) return rc;
    wgrad_defer_reduce(defer_join != 0);
    const int rcw = cnn_conv2d_backward_weight_pooled2(d, x, dpool, mask, pooled, gw, gb, divisor, ws, ws_bytes, side->stream);
    wgrad_defer_reduce(false);
    if (rcw) return rcw;
    if (int rc = cnn_conv2d_backward_data_pooled2_prepared(d, dpool, mask, pooled, prepared_dgrad, dx, main)) return rc;
    if (!defer_join) {
        CNN_HIP_CHECK(hipEventRecord(side->join, side->stream));
        CNN_HIP_CHECK(hipStreamWaitEvent(main, side->join, 0));
    }
    return CNN_AMD_OK;
}

}  // extern "C"
