// elementwise.hip -- HBM-bound streaming kernels: ReLU fwd/bwd (relu.cpp:21-26,35-40), the SGD step
// (conv2d.cpp:205-217, linear.cpp:95-102) and the softmax / cross-entropy glue (func.cpp:16-73).
// All are 16 B/lane vectorised grid-stride loops; roofline = HBM (8 / 12 / 12 B per element).
#include <cfloat>
#include <cstdint>

#include "common.h"

using namespace cnn_amd;

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float relu_f(float v) { return v >= 0.f ? v : 0.f; }        // relu.cpp:25
__device__ __forceinline__ float relu_b(float y, float d) { return y <= 0.f ? 0.f : d; }  // relu.cpp:38

__global__ __launch_bounds__(kBlock) void relu_fwd_vec(const float4* __restrict__ x, float4* __restrict__ y,
                                                       size_t n4) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (size_t)gridDim.x * kBlock) {
        float4 v = x[i];
        v.x = relu_f(v.x); v.y = relu_f(v.y); v.z = relu_f(v.z); v.w = relu_f(v.w);
        y[i] = v;
    }
}
__global__ __launch_bounds__(kBlock) void relu_fwd_scalar(const float* __restrict__ x, float* __restrict__ y,
                                                          size_t begin, size_t n) {
    for (size_t i = begin + (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
        y[i] = relu_f(x[i]);
}

__global__ __launch_bounds__(kBlock) void relu_bwd_vec(const float4* __restrict__ y, float4* __restrict__ d,
                                                       size_t n4) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (size_t)gridDim.x * kBlock) {
        const float4 yv = y[i];
        float4 dv = d[i];
        dv.x = relu_b(yv.x, dv.x); dv.y = relu_b(yv.y, dv.y); dv.z = relu_b(yv.z, dv.z); dv.w = relu_b(yv.w, dv.w);
        d[i] = dv;
    }
}
__global__ __launch_bounds__(kBlock) void relu_bwd_scalar(const float* __restrict__ y, float* __restrict__ d,
                                                          size_t begin, size_t n) {
    for (size_t i = begin + (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
        d[i] = relu_b(y[i], d[i]);
}

// Dropout (cpu/src/dropout.cpp): CHANNEL dropout -- the first `dropped` channels of every sample are zeroed in training
// (:34-41: the loop tests the channel INDEX o against selected_num; the shuffled `sequence` only feeds the mask bookkeeping,
// so the dropped set is always channels 0 .. selected_num-1), the whole tensor is scaled by 1 - p under no_grad (:44-53).
// backward (:57-69) zeroes the same channels of the delta in place.  16 B/lane where the channel planes allow it.
__global__ __launch_bounds__(kBlock) void dropout_fwd(const float* __restrict__ x, float* __restrict__ y, size_t n, int C, int area,
                                                      int dropped, int training, float keep) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const int o = (int)((i / (size_t)area) % (size_t)C);
        const float v = x[i];
        y[i] = training ? (o >= dropped ? v : 0.f) : v * keep;
    }
}
__global__ __launch_bounds__(kBlock) void dropout_bwd(float* __restrict__ d, size_t n, int C, int area, int dropped) {
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const int o = (int)((i / (size_t)area) % (size_t)C);
        if (o < dropped) d[i] = 0.f;
    }
}

// sgd_one(): common.h
// (the <= 3 trailing elements of an arena whose length is not a multiple of 4 ride along in workgroup 0)
// `keep` (nullable, wave-uniform): receives the value every parameter had BEFORE the step (cnn_sgd_update_keep)
__global__ __launch_bounds__(kBlock) void sgd_vec(float4* __restrict__ p, const float4* __restrict__ g, size_t n4,
                                                  size_t n, float lr, float scale, bool scaled, float4* __restrict__ keep) {
    if (blockIdx.x == 0 && n4 * 4 + threadIdx.x < n) {
        float* ps = (float*)p;
        const float* gs = (const float*)g;
        const size_t i = n4 * 4 + threadIdx.x;
        if (keep) ((float*)keep)[i] = ps[i];
        ps[i] = sgd_one(ps[i], gs[i], lr, scale, scaled);
    }
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (size_t)gridDim.x * kBlock) {
        float4 pv = p[i];
        const float4 gv = g[i];
        if (keep) keep[i] = pv;
        pv.x = sgd_one(pv.x, gv.x, lr, scale, scaled); pv.y = sgd_one(pv.y, gv.y, lr, scale, scaled);
        pv.z = sgd_one(pv.z, gv.z, lr, scale, scaled); pv.w = sgd_one(pv.w, gv.w, lr, scale, scaled);
        p[i] = pv;
    }
}
__global__ __launch_bounds__(kBlock) void sgd_scalar(float* __restrict__ p, const float* __restrict__ g,
                                                     size_t begin, size_t n, float lr, float scale, bool scaled, float* __restrict__ keep) {
    for (size_t i = begin + (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        if (keep) keep[i] = p[i];
        p[i] = sgd_one(p[i], g[i], lr, scale, scaled);
    }
}

// func.cpp:6-12
__device__ __forceinline__ float clamped_exp(float v) {
    if (v >= 88.f) return FLT_MAX;
    if (v <= -50.f) return 0.f;
    return expf(v);
}

// One workgroup; thread t owns samples t, t+256, ...  Per-sample arithmetic is the reference's sequential
// loop (func.cpp:22-33, 63-68); the loss terms are then added in ascending sample order by one lane so the
// fp32 loss sum has the reference's order (func.cpp:60-71).
__global__ __launch_bounds__(kBlock) void softmax_xent_kernel(const float* __restrict__ logits,
                                                              const int32_t* __restrict__ labels,
                                                              float* __restrict__ probs, float* __restrict__ delta,
                                                              float* __restrict__ loss_sum, int B, int classes) {
    __shared__ float terms[kBlock];
    float running = 0.f;
    for (int base = 0; base < B; base += kBlock) {
        const int b = base + threadIdx.x;
        float term = 0.f;
        if (b < B) {
            const float* in = logits + (size_t)b * classes;
            float mx = in[0];  // Tensor3D::max = first maximum, strict '>' (data_format.cpp:37-48)
            for (int i = 1; i < classes; ++i)
                if (in[i] > mx) mx = in[i];
            float sum = 0.f;
            for (int i = 0; i < classes; ++i) sum += clamped_exp(in[i] - mx);
            const int label = labels[b];
            for (int i = 0; i < classes; ++i) {
                float p = clamped_exp(in[i] - mx) / sum;
                if (isnan(p)) p = 0.f;
                const float yv = (i == label) ? 1.f : 0.f;
                if (probs) probs[(size_t)b * classes + i] = p;
                delta[(size_t)b * classes + i] = p - yv;  // no 1/B here (func.cpp:64)
                term += logf(p) * yv;                     // func.cpp:65, incl. its log(0)*0 = NaN behaviour
            }
        }
        terms[threadIdx.x] = term;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int cnt = min(kBlock, B - base);
            for (int i = 0; i < cnt; ++i) running += terms[i];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss_sum) loss_sum[0] = -running;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }


// ---- AlexNet::grad_cam (alexnet.cpp:107-140) on a [B][C][H][W] feature map -------------------------------------------------
// weights[b][o] = (sum_i fea[b][o][i]) / area, sequential in i (:111-119; the reference takes the channel mean of the FEATURE MAP,
// not of a gradient); cam[b][i] = sum_o weights[b][o] * fea[b][o][i], sequential in o, multiply-then-add (:124-131); ReLU as
// `if (v < 0) v = 0` (:134).  One workgroup per sample: thread t owns channels t, t+256, ... in the first phase, pixels in the second.
__global__ __launch_bounds__(kBlock) void grad_cam_maps(const float* __restrict__ fea, float* __restrict__ cam, int C, int area) {
    extern __shared__ float wts[];
    const float* fb = fea + (size_t)blockIdx.x * C * area;
    for (int o = threadIdx.x; o < C; o += kBlock) {
        const float* f = fb + (size_t)o * area;
        float m = 0.f;
        for (int i = 0; i < area; ++i) m += f[i];
        wts[o] = m / area;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < area; i += kBlock) {
#pragma clang fp contract(off)
        float acc = 0.f;
        for (int o = 0; o < C; ++o) {
            const float prod = wts[o] * fb[(size_t)o * area + i];
            acc = acc + prod;
        }
        if (acc < 0.f) acc = 0.f;
        cam[(size_t)blockIdx.x * area + i] = acc;
    }
}
// min-max normalisation over the WHOLE [B][H][W] tensor (:136-139; Tensor3D::min/max = first extremum with strict comparisons,
// data_format.cpp:37-62: a NaN is only ever returned from element 0) and the 8-bit image of the first plane (opecv_mat(1),
// data_format.cpp:98-103: saturate_cast<uchar>(255 * v) = round to nearest even, clamped)
__global__ __launch_bounds__(kBlock) void grad_cam_normalise(float* __restrict__ cam, unsigned char* __restrict__ image, size_t n, int area) {
    __shared__ float smin[kBlock], smax[kBlock];
    const float first = cam[0];
    float lo = INFINITY, hi = -INFINITY;
    for (size_t i = threadIdx.x; i < n; i += kBlock) {
        const float v = cam[i];
        if (v < lo) lo = v;
        if (v > hi) hi = v;
    }
    smin[threadIdx.x] = lo;
    smax[threadIdx.x] = hi;
    __syncthreads();
    for (int off = kBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            if (smin[threadIdx.x + off] < smin[threadIdx.x]) smin[threadIdx.x] = smin[threadIdx.x + off];
            if (smax[threadIdx.x + off] > smax[threadIdx.x]) smax[threadIdx.x] = smax[threadIdx.x + off];
        }
        __syncthreads();
    }
    // element 0 is the start value of both scans: a NaN there survives every comparison; all-NaN tails leave +-inf, which the
    // reference cannot produce (it starts from element 0) -- fall back to element 0 then
    float mn = smin[0], mx = smax[0];
    if (first != first) mn = mx = first;
    else {
        if (!(mn <= first)) mn = first;
        if (!(mx >= first)) mx = first;
    }
    const float res = mx - mn;
    __syncthreads();
    for (size_t i = threadIdx.x; i < n; i += kBlock) {
        const float v = (cam[i] - mn) / res;
        cam[i] = v;
        if (image && i < (size_t)area) {
            const float sv = 255.f * v;
            int r = (sv != sv) ? 0 : (sv >= 255.5f ? 255 : (sv <= -0.5f ? 0 : __float2int_rn(sv)));
            image[i] = (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
    }
}

}  // namespace

extern "C" {

int cnn_relu_forward(const float* x, float* y, size_t n, void* stream) {
    if (n == 0) return CNN_AMD_OK;
    CNN_REQUIRE(x && y, "cnn_relu_forward: null pointer");
    hipStream_t s = as_stream(stream);
    size_t done = 0;
    if (aligned16(x) && aligned16(y) && n >= 4) {
        const size_t n4 = n / 4;
        CNN_KLAUNCH(s, "relu_fwd_vec", (relu_fwd_vec<<<stream_grid(n4, kBlock), kBlock, 0, s>>>((const float4*)x, (float4*)y, n4)),
                    "n=%zu", n);
        done = n4 * 4;
    }
    if (done < n) {
        CNN_KLAUNCH(s, "relu_fwd_scalar", (relu_fwd_scalar<<<stream_grid(n - done, kBlock), kBlock, 0, s>>>(x, y, done, n)),
                    "tail n=%zu", n - done);
    }
    return CNN_AMD_OK;
}

int cnn_relu_backward(const float* y, float* dy, size_t n, void* stream) {
    if (n == 0) return CNN_AMD_OK;
    CNN_REQUIRE(y && dy, "cnn_relu_backward: null pointer");
    hipStream_t s = as_stream(stream);
    size_t done = 0;
    if (aligned16(y) && aligned16(dy) && n >= 4) {
        const size_t n4 = n / 4;
        CNN_KLAUNCH(s, "relu_bwd_vec", (relu_bwd_vec<<<stream_grid(n4, kBlock), kBlock, 0, s>>>((const float4*)y, (float4*)dy, n4)),
                    "n=%zu", n);
        done = n4 * 4;
    }
    if (done < n) {
        CNN_KLAUNCH(s, "relu_bwd_scalar", (relu_bwd_scalar<<<stream_grid(n - done, kBlock), kBlock, 0, s>>>(y, dy, done, n)),
                    "tail n=%zu", n - done);
    }
    return CNN_AMD_OK;
}

int cnn_sgd_update(float* params, const float* grads, size_t n, float lr, float grad_scale, void* stream) {
    return cnn_sgd_update_keep(params, grads, n, lr, grad_scale, nullptr, stream);
}

int cnn_sgd_update_keep(float* params, const float* grads, size_t n, float lr, float grad_scale, float* previous, void* stream) {
    if (n == 0) return CNN_AMD_OK;
    CNN_REQUIRE(params && grads, "cnn_sgd_update: null pointer");
    hipStream_t s = as_stream(stream);
    const bool scaled = grad_scale != 1.0f;
    size_t done = 0;
    if (aligned16(params) && aligned16(grads) && (previous == nullptr || aligned16(previous)) && n >= 4) {
        const size_t n4 = n / 4;
        CNN_KLAUNCH(s, "sgd_vec",
                    (sgd_vec<<<stream_grid(n4, kBlock), kBlock, 0, s>>>((float4*)params, (const float4*)grads, n4, n, lr,
                                                                       grad_scale, scaled, (float4*)previous)),
                    "n=%zu", n);
        done = n;
    }
    if (done < n) {
        CNN_KLAUNCH(s, "sgd_scalar",
                    (sgd_scalar<<<stream_grid(n - done, kBlock), kBlock, 0, s>>>(params, grads, done, n, lr, grad_scale, scaled, previous)),
                    "tail n=%zu", n - done);
    }
    return CNN_AMD_OK;
}

int cnn_softmax_xent(const float* logits, const int32_t* labels, float* probs, float* delta, float* loss_sum, int B,
                     int classes, void* stream) {
    CNN_REQUIRE(logits && labels && delta, "cnn_softmax_xent: null pointer");
    CNN_REQUIRE(B > 0 && classes > 0, "cnn_softmax_xent: B=%d classes=%d", B, classes);
    hipStream_t s = as_stream(stream);
    CNN_KLAUNCH(s, "softmax_xent_kernel",
                (softmax_xent_kernel<<<1, kBlock, 0, s>>>(logits, labels, probs, delta, loss_sum, B, classes)), "B=%d classes=%d",
                B, classes);
    return CNN_AMD_OK;
}

int cnn_dropout_forward(const float* x, float* y, int B, int C, int H, int W, int dropped_channels, int training, float keep,
                        void* stream) {
    CNN_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0 && dropped_channels >= 0 && dropped_channels <= C, "cnn_dropout_forward: bad arguments");
    const size_t n = (size_t)B * C * H * W;
    hipStream_t s = as_stream(stream);
    CNN_KLAUNCH(s, "dropout_fwd", (dropout_fwd<<<stream_grid(n, kBlock), kBlock, 0, s>>>(x, y, n, C, H * W, dropped_channels, training, keep)),
                "n=%zu", n);
    return CNN_AMD_OK;
}

int cnn_dropout_backward(float* dy_inout, int B, int C, int H, int W, int dropped_channels, void* stream) {
    CNN_REQUIRE(dy_inout && B > 0 && C > 0 && H > 0 && W > 0 && dropped_channels >= 0 && dropped_channels <= C, "cnn_dropout_backward: bad arguments");
    const size_t n = (size_t)B * C * H * W;
    hipStream_t s = as_stream(stream);
    CNN_KLAUNCH(s, "dropout_bwd", (dropout_bwd<<<stream_grid(n, kBlock), kBlock, 0, s>>>(dy_inout, n, C, H * W, dropped_channels)), "n=%zu", n);
    return CNN_AMD_OK;
}


int cnn_grad_cam(const float* feature, int B, int C, int H, int W, float* cam, unsigned char* image, void* stream) {
    CNN_REQUIRE(feature && cam, "cnn_grad_cam: null pointer");
    CNN_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && (long long)H * W < (1ll << 31), "cnn_grad_cam: B=%d C=%d H=%d W=%d", B, C, H, W);
    CNN_REQUIRE((size_t)C * sizeof(float) <= 64 * 1024, "cnn_grad_cam: C=%d channels exceed the 64 KB weight table", C);
    hipStream_t s = as_stream(stream);
    const int area = H * W;
    CNN_KLAUNCH(s, "grad_cam_maps", (grad_cam_maps<<<B, kBlock, (size_t)C * sizeof(float), s>>>(feature, cam, C, area)), "B%d C%d %dx%d", B, C, H, W);
    CNN_KLAUNCH(s, "grad_cam_normalise", (grad_cam_normalise<<<1, kBlock, 0, s>>>(cam, image, (size_t)B * area, area)), "n=%zu", (size_t)B * area);
    return CNN_AMD_OK;
}

}  // extern "C"
