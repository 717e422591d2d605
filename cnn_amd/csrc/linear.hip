// linear.hip -- LinearLayer forward / backward (cpu/src/linear.cpp:33-43, 56-90).  W is [in][out] row-major.
// The reference net's layer is 4608 -> 3: a skinny contraction that is HBM-bound on x (4*in bytes per sample),
// so these are wave-reduction / streaming kernels, not MFMA tiles.
#include <cfloat>

#include "common.h"

using namespace cnn_amd;

namespace {

constexpr int kBlock = 256;
constexpr int kOutTile = 8;  // outputs accumulated per pass in registers

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// y[b][j] = sum_i x[b][i]*W[i][j] + bias[j].  One workgroup per sample; lanes stride over i (coalesced x),
// W rows are `out` contiguous floats (L2-resident: 4608*3*4 = 55 KB).
__global__ __launch_bounds__(kBlock) void linear_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y, int in,
                                                     int out) {
    __shared__ float part[kBlock / kWave][kOutTile];
    const int b = blockIdx.x;
    const float* xb = x + (size_t)b * in;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j0 = 0; j0 < out; j0 += kOutTile) {
        const int nj = min(kOutTile, out - j0);
        float acc[kOutTile];
#pragma unroll
        for (int j = 0; j < kOutTile; ++j) acc[j] = 0.f;
#pragma unroll 6  // (six independent x / W loads in flight per thread: the loop is latency-, not bandwidth-bound)
        for (int i = threadIdx.x; i < in; i += kBlock) {
            const float xv = xb[i];
            const float* wr = w + (size_t)i * out + j0;
#pragma unroll
            for (int j = 0; j < kOutTile; ++j)
                if (j < nj) acc[j] = __builtin_fmaf(xv, wr[j], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < kOutTile; ++j) {
            const float s = wave_sum(acc[j]);
            if (lane == 0) part[wave][j] = s;
        }
        __syncthreads();
        if (threadIdx.x < nj) {
            float s = 0.f;
            for (int wv = 0; wv < kBlock / kWave; ++wv) s += part[wv][threadIdx.x];
            y[(size_t)b * out + j0 + threadIdx.x] = s + bias[j0 + threadIdx.x];
        }
        __syncthreads();
    }
}

struct __attribute__((packed, aligned(4))) w3 { float a, b, c; };  // one W row of a 3-output layer: a single 12-byte load

// func.cpp:6-12
__device__ __forceinline__ float clamped_exp_l(float v) {
    if (v >= 88.f) return FLT_MAX;
    if (v <= -50.f) return 0.f;
    return expf(v);
}

// linear_fwd (out <= kOutTile) + the sample's softmax / cross-entropy (func.cpp:16-33, 60-71) in the same workgroup:
// probs (nullable), delta = probs - onehot, and the sample's loss term log(p[label]) into loss_terms[b].  The ordered sum
// over samples (the reference's order) is a separate, on-demand reduction: cnn_loss_from_terms.
// DX != 0 (round 3): the workgroup also writes its sample's row of LinearLayer::backward's data gradient (linear.cpp:73-90),
// dx[b][i] = sum_j delta[b][j] * W[i][j] -- it needs nothing but this sample's delta and the W rows the threads hold anyway --
// with the arithmetic of linear_bwd_fused (same products, same order: bit-identical); DX == 2 also applies the ReLU::backward of
// the layer in front (x <= 0 ? 0 : dx).  The critical path of a train step then has ONE head kernel between the last forward
// convolution and the first data gradient; the weight / bias gradient of the layer (which needs every sample's delta) runs
// beside the convolutions' weight gradients (linear_bwd_fused<.., false>).
template <bool BATCH, int DX>
__global__ __launch_bounds__(kBlock) void linear_fwd_softmax_xent(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, const int32_t* __restrict__ labels,
                                                                  float* __restrict__ y, float* __restrict__ probs,
                                                                  float* __restrict__ delta, float* __restrict__ loss_terms, int in,
                                                                  int out, float* __restrict__ dx) {
    __shared__ float part[kBlock / kWave][kOutTile];
    __shared__ float logit[kOutTile];
    __shared__ float dl[kOutTile];
    constexpr int UK = 18;
    float xk[UK];  // (BATCH && out == 3 && in == UK * kBlock: the thread's x values and W rows stay in registers for the dx pass)
    w3 wk[UK];
    bool kept = false;
    const int b = blockIdx.x;
    const float* xb = x + (size_t)b * in;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[kOutTile];
#pragma unroll
    for (int j = 0; j < kOutTile; ++j) acc[j] = 0.f;
    int i = threadIdx.x;
    if (BATCH && out == 3) {
        // the reference net's head: every load of 18 iterations (one 4-byte x value + one 12-byte W row each) is issued before
        // the first FMA -- in the train step this kernel shares the chip with an HBM-bound kernel, and each dependent round
        // trip costs microseconds there.  Same products, same order as the generic loop below.
        constexpr int U = UK;
        for (; i + (U - 1) * kBlock < in; i += U * kBlock) {
            float xv[U];
            w3 wv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xv[u] = xb[i + u * kBlock];
                wv[u] = *reinterpret_cast<const w3*>(w + (size_t)(i + u * kBlock) * 3);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc[0] = __builtin_fmaf(xv[u], wv[u].a, acc[0]);  // explicit: contraction must not depend on the loop shape
                acc[1] = __builtin_fmaf(xv[u], wv[u].b, acc[1]);
                acc[2] = __builtin_fmaf(xv[u], wv[u].c, acc[2]);
            }
            if (DX != 0 && in == U * kBlock) {
                kept = true;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    xk[u] = xv[u];
                    wk[u] = wv[u];
                }
            }
        }
    }
    if (BATCH && out == 3) {  // what the 18-wide loop left (the stacks' heads: in = 98 * 256): four iterations' loads at once, same order
        constexpr int U = 4;
        for (; i + (U - 1) * kBlock < in; i += U * kBlock) {
            float xv[U];
            w3 wv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xv[u] = xb[i + u * kBlock];
                wv[u] = *reinterpret_cast<const w3*>(w + (size_t)(i + u * kBlock) * 3);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc[0] = __builtin_fmaf(xv[u], wv[u].a, acc[0]);
                acc[1] = __builtin_fmaf(xv[u], wv[u].b, acc[1]);
                acc[2] = __builtin_fmaf(xv[u], wv[u].c, acc[2]);
            }
        }
    }
#pragma unroll 6
    for (; i < in; i += kBlock) {
        const float xv = xb[i];
        const float* wr = w + (size_t)i * out;
#pragma unroll
        for (int j = 0; j < kOutTile; ++j)
            if (j < out) acc[j] = __builtin_fmaf(xv, wr[j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < kOutTile; ++j) {
        const float s = wave_sum(acc[j]);
        if (lane == 0) part[wave][j] = s;
    }
    __syncthreads();
    if (threadIdx.x < out) {
        float s = 0.f;
        for (int wv = 0; wv < kBlock / kWave; ++wv) s += part[wv][threadIdx.x];
        s += bias[threadIdx.x];
        y[(size_t)b * out + threadIdx.x] = s;
        logit[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // the reference's sequential per-sample arithmetic (same as softmax_xent_kernel)
        float mx = logit[0];
        for (int i = 1; i < out; ++i)
            if (logit[i] > mx) mx = logit[i];
        float sum = 0.f;
        for (int i = 0; i < out; ++i) sum += clamped_exp_l(logit[i] - mx);
        const int label = labels[b];
        float term = 0.f;
        for (int i = 0; i < out; ++i) {
            float pr = clamped_exp_l(logit[i] - mx) / sum;
            if (isnan(pr)) pr = 0.f;
            const float yv = (i == label) ? 1.f : 0.f;
            if (probs) probs[(size_t)b * out + i] = pr;
            delta[(size_t)b * out + i] = pr - yv;
            if (DX != 0) dl[i] = pr - yv;
            term += logf(pr) * yv;
        }
        loss_terms[b] = term;
    }
    if constexpr (DX != 0) {
        __syncthreads();
        float* dxb = dx + (size_t)b * in;
        if (kept) {  // out == 3, in == 18 * 256: no loads at all
            const float d0 = dl[0], d1 = dl[1], d2 = dl[2];
#pragma unroll
            for (int u = 0; u < UK; ++u) {
                float sj = 0.f;  // (linear_bwd_fused's expression, term by term)
                sj = __builtin_fmaf(d0, wk[u].a, sj);  // (explicit: the contraction must not depend on the loop shape)
                sj = __builtin_fmaf(d1, wk[u].b, sj);
                sj = __builtin_fmaf(d2, wk[u].c, sj);
                dxb[threadIdx.x + u * kBlock] = (DX == 2 && xk[u] <= 0.f) ? 0.f : sj;
            }
        } else {
            int i2 = threadIdx.x;
            if (BATCH && out == 3) {  // (round 6) seven iterations' loads in flight (the stacks' heads: 98 iterations per thread, one round trip each before)
                constexpr int U = 7;
                const float d0 = dl[0], d1 = dl[1], d2 = dl[2];
                for (; i2 + (U - 1) * kBlock < in; i2 += U * kBlock) {
                    w3 wv[U];
                    float xv[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        wv[u] = *reinterpret_cast<const w3*>(w + (size_t)(i2 + u * kBlock) * 3);
                        xv[u] = DX == 2 ? xb[i2 + u * kBlock] : 1.f;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        float sj = 0.f;
                        sj = __builtin_fmaf(d0, wv[u].a, sj);
                        sj = __builtin_fmaf(d1, wv[u].b, sj);
                        sj = __builtin_fmaf(d2, wv[u].c, sj);
                        dxb[i2 + u * kBlock] = (DX == 2 && xv[u] <= 0.f) ? 0.f : sj;
                    }
                }
            }
            for (; i2 < in; i2 += kBlock) {
                const float* wr = w + (size_t)i2 * out;
                float sj = 0.f;
#pragma unroll
                for (int j = 0; j < kOutTile; ++j)
                    if (j < out) sj = __builtin_fmaf(dl[j], wr[j], sj);
                dxb[i2] = (DX == 2 && xb[i2] <= 0.f) ? 0.f : sj;
            }
        }
    }
}

// loss_sum[0] = -(terms[0] + terms[1] + ...) in ascending sample order (func.cpp:60-71)
__global__ void loss_from_terms(const float* __restrict__ terms, float* __restrict__ loss_sum, int B) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float running = 0.f;
        for (int i = 0; i < B; ++i) running += terms[i];
        loss_sum[0] = -running;
    }
}

// gW[i][j] = (sum_b x[b][i]*dy[b][j]) / divisor.  A workgroup owns 64 input neurons; its 4 waves each take a
// quarter of the samples (coalesced x rows, dy broadcast through the scalar cache) and are combined in fixed order.
constexpr int kNeur = 64, kBGroups = kBlock / kNeur;
__global__ __launch_bounds__(kBlock) void linear_bwd_w(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ gw, int B, int in, int out,
                                                       float divisor) {
    __shared__ float red[kBGroups][kOutTile][kNeur];
    const int il = threadIdx.x & (kNeur - 1), grp = threadIdx.x / kNeur;
    const int i = blockIdx.x * kNeur + il;
    const int per = (B + kBGroups - 1) / kBGroups;
    const int bb = grp * per, be = min(B, bb + per);
    for (int j0 = 0; j0 < out; j0 += kOutTile) {
        const int nj = min(kOutTile, out - j0);
        float acc[kOutTile];
#pragma unroll
        for (int j = 0; j < kOutTile; ++j) acc[j] = 0.f;
        if (i < in) {
#pragma unroll 4
            for (int b = bb; b < be; ++b) {
                const float xv = x[(size_t)b * in + i];
                const float* d = dy + (size_t)b * out + j0;
#pragma unroll
                for (int j = 0; j < kOutTile; ++j)
                    if (j < nj) acc[j] += xv * d[j];
            }
        }
#pragma unroll
        for (int j = 0; j < kOutTile; ++j) red[grp][j][il] = acc[j];
        __syncthreads();
        if (grp == 0 && i < in)
            for (int j = 0; j < nj; ++j) {
                float t = 0.f;
#pragma unroll
                for (int g2 = 0; g2 < kBGroups; ++g2) t += red[g2][j][il];
                gw[(size_t)i * out + j0 + j] = t / divisor;
            }
        __syncthreads();
    }
}

// gb[j] = (sum_b dy[b][j]) / divisor: one wave per output, lanes over samples, fixed shuffle tree
__global__ __launch_bounds__(64) void linear_bwd_b(const float* __restrict__ dy, float* __restrict__ gb, int B, int out,
                                                   float divisor) {
    const int j = blockIdx.x;
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += 64) s += dy[(size_t)b * out + j];
    s = wave_sum(s);
    if (threadIdx.x == 0) gb[j] = s / divisor;
}

// dx[b][i] = sum_j dy[b][j]*W[i][j]
__global__ __launch_bounds__(kBlock) void linear_bwd_x(const float* __restrict__ dy, const float* __restrict__ w,
                                                       float* __restrict__ dx, int in, int out) {
    const int b = blockIdx.y;
    const float* d = dy + (size_t)b * out;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < in; i += gridDim.x * kBlock) {
        const float* wr = w + (size_t)i * out;
        float s = 0.f;
        for (int j = 0; j < out; ++j) s += d[j] * wr[j];
        dx[(size_t)b * in + i] = s;
    }
}

// All three gradients of a skinny layer (out <= kOutTile) in ONE launch: a workgroup owns 64 input neurons, its kFG
// sample groups each walk B / kFG samples -- per (sample, neuron): one coalesced x load, the sample's dy row (broadcast),
// gW partial sums in registers and the finished dx element written straight away -- and are combined through LDS in a
// fixed order.  Workgroup 0 also reduces the bias gradient.  x is read once, dx written once (linear.cpp:56-90).
// RELU: x is the output of the ReLU layer feeding this layer; dx is stored as (x <= 0 ? 0 : dx) = that layer's backward pass
// (relu.cpp:38) -- the mask value is the x element this thread has loaded anyway.
constexpr int kFG = 16;
// OUT is a compile-time constant and the x loads of a 16-sample batch are issued -- unconditionally, on clamped indices -- before the
// first FMA (round 6: with a runtime `out` the compiler had put every load of the loop behind its own branch and s_waitcnt vmcnt(0):
// sixteen dependent round trips, 30 us for 4.7 MB); the sample group is wave-uniform, so a sample's dy row comes through the scalar cache.
template <int OUT, bool RELU, bool WRITE_DX>  // WRITE_DX = false: weight / bias gradient only (dx came from the head kernel)
__global__ __launch_bounds__(kNeur * kFG) void linear_bwd_fused(const float* __restrict__ x, const float* __restrict__ dy,
                                                               const float* __restrict__ w, float* __restrict__ gw,
                                                               float* __restrict__ gb, float* __restrict__ dx, int B,
                                                               int in, float divisor) {
    static_assert(kNeur == kWave, "a sample group is one wave");
    __shared__ float red[kFG][OUT + 1][kNeur];
    const int il = threadIdx.x & (kNeur - 1), grp = __builtin_amdgcn_readfirstlane(threadIdx.x / kNeur);
    const int i = blockIdx.x * kNeur + il;
    const bool live = i < in;
    const int ic = live ? i : in - 1;  // (lanes behind the last neuron load a valid address and store nothing)
    const int per = (B + kFG - 1) / kFG;
    const int bb = grp * per, be = min(B, bb + per);
    float wr[OUT], acc[OUT], bsum = 0.f;
#pragma unroll
    for (int j = 0; j < OUT; ++j) {
        acc[j] = 0.f;
        wr[j] = WRITE_DX ? w[(size_t)ic * OUT + j] : 0.f;
    }
    constexpr int U = 16;
    for (int b0 = bb; b0 < be; b0 += U) {
        float xv[U], dv[U][OUT];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = x[(size_t)min(b0 + u, be - 1) * in + ic];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < OUT; ++j) dv[u][j] = dy[(size_t)min(b0 + u, be - 1) * OUT + j];  // (scalar loads: the row is wave-uniform)
        const int nu = min(U, be - b0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u < nu) {  // (wave-uniform)
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < OUT; ++j) {
                    acc[j] = __builtin_fmaf(xv[u], dv[u][j], acc[j]);
                    s = __builtin_fmaf(dv[u][j], wr[j], s);
                }
                if (WRITE_DX && live) dx[(size_t)(b0 + u) * in + i] = (RELU && xv[u] <= 0.f) ? 0.f : s;
            }
        }
    }
    if (blockIdx.x == 0)  // lane j < OUT of every group: bias partial of output j (all loads of a batch before the first sum)
        for (int b0 = bb; b0 < be; b0 += U) {
            float bv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) bv[u] = dy[(size_t)min(b0 + u, be - 1) * OUT + min(il, OUT - 1)];
#pragma unroll
            for (int u = 0; u < U; ++u) bsum += b0 + u < be ? bv[u] : 0.f;
        }
#pragma unroll
    for (int j = 0; j < OUT; ++j) red[grp][j][il] = acc[j];
    red[grp][OUT][il] = bsum;
    __syncthreads();
    if (grp == 0) {
        if (live)
#pragma unroll
            for (int j = 0; j < OUT; ++j) {
                float t = 0.f;
#pragma unroll
                for (int g2 = 0; g2 < kFG; ++g2) t += red[g2][j][il];
                gw[(size_t)i * OUT + j] = t / divisor;
            }
        if (blockIdx.x == 0 && il < OUT) {
            float t = 0.f;
#pragma unroll
            for (int g2 = 0; g2 < kFG; ++g2) t += red[g2][OUT][il];
            gb[il] = t / divisor;
        }
    }
}

template <int OUT>
static void launch_bwd_fused(hipStream_t s, int variant, const float* x, const float* dy, const float* w, float* gw, float* gb, float* dx,
                             int B, int in, float divisor) {
    const dim3 grid(ceil_div(in, kNeur)), block(kNeur * kFG);
    if (variant == 0) linear_bwd_fused<OUT, false, false><<<grid, block, 0, s>>>(x, dy, w, gw, gb, nullptr, B, in, divisor);
    else if (variant == 1) launch_pub(linear_bwd_fused<OUT, true, true>, grid, block, 0, s, x, dy, w, gw, gb, dx, B, in, divisor);
    else launch_pub(linear_bwd_fused<OUT, false, true>, grid, block, 0, s, x, dy, w, gw, gb, dx, B, in, divisor);
}
// variant 0: weight / bias gradient alone; 1: all three + the ReLU::backward of the layer in front; 2: all three
static void launch_bwd_fused(hipStream_t s, int variant, const float* x, const float* dy, const float* w, float* gw, float* gb, float* dx,
                             int B, int in, int out, float divisor) {
    switch (out) {
        case 1: return launch_bwd_fused<1>(s, variant, x, dy, w, gw, gb, dx, B, in, divisor);
        case 2: return launch_bwd_fused<2>(s, variant, x, dy, w, gw, gb, dx, B, in, divisor);
        case 3: return launch_bwd_fused<3>(s, variant, x, dy, w, gw, gb, dx, B, in, divisor);
        case 4: return launch_bwd_fused<4>(s, variant, x, dy, w, gw, gb, dx, B, in, divisor);
        case 5: return launch_bwd_fused<5>(s, variant, x, dy, w, gw, gb, dx, B, in, divisor);
        case 6: return launch_bwd_fused<6>(s, variant, x, dy, w, gw, gb, dx, B, in, divisor);
        case 7: return launch_bwd_fused<7>(s, variant, x, dy, w, gw, gb, dx, B, in, divisor);
        default: return launch_bwd_fused<8>(s, variant, x, dy, w, gw, gb, dx, B, in, divisor);
    }
}

}  // namespace

extern "C" {

int cnn_linear_forward(const float* x, const float* w, const float* bias, float* y, int B, int in, int out,
                       void* stream) {
    CNN_REQUIRE(x && w && bias && y, "cnn_linear_forward: null pointer");
    CNN_REQUIRE(B > 0 && in > 0 && out > 0, "cnn_linear_forward: B=%d in=%d out=%d", B, in, out);
    hipStream_t s = as_stream(stream);
    CNN_KLAUNCH(s, "linear_fwd", (linear_fwd<<<B, kBlock, 0, s>>>(x, w, bias, y, in, out)), "B%d in%d out%d", B, in, out);
    return CNN_AMD_OK;
}

static int linear_backward_impl(const float* x, const float* dy, const float* w, float* gw, float* gb, float* dx, int B,
                                int in, int out, float divisor, void* stream, bool relu_below) {
    CNN_REQUIRE(dy != nullptr, "cnn_linear_backward: dy is null");
    CNN_REQUIRE(B > 0 && in > 0 && out > 0, "cnn_linear_backward: B=%d in=%d out=%d", B, in, out);
    CNN_REQUIRE(B <= 65535, "cnn_linear_backward: B=%d exceeds the grid.y limit", B);
    hipStream_t s = as_stream(stream);
    if (gw && gb && !dx && x && w && out <= kOutTile) {  // the gradients of the parameters alone, in linear_bwd_fused's summation order
        CNN_KLAUNCH(s, "linear_bwd_fused/wb", launch_bwd_fused(s, 0, x, dy, w, gw, gb, nullptr, B, in, out, divisor), "B%d in%d out%d", B, in, out);
        return CNN_AMD_OK;
    }
    if (gw && gb && dx && x && w && out <= kOutTile) {
        if (relu_below)
            CNN_KLAUNCH(s, "linear_bwd_fused+relu", launch_bwd_fused(s, 1, x, dy, w, gw, gb, dx, B, in, out, divisor), "B%d in%d out%d", B, in, out);
        else
            CNN_KLAUNCH(s, "linear_bwd_fused", launch_bwd_fused(s, 2, x, dy, w, gw, gb, dx, B, in, out, divisor), "B%d in%d out%d", B, in, out);
        return CNN_AMD_OK;
    }
    if (gw) {
        CNN_REQUIRE(x != nullptr, "cnn_linear_backward: x is null");
        CNN_KLAUNCH(s, "linear_bwd_w", (linear_bwd_w<<<ceil_div(in, kNeur), kBlock, 0, s>>>(x, dy, gw, B, in, out, divisor)),
                    "B%d in%d out%d", B, in, out);
    }
    if (gb) {
        CNN_KLAUNCH(s, "linear_bwd_b", (linear_bwd_b<<<out, 64, 0, s>>>(dy, gb, B, out, divisor)), "B%d out%d", B,
                    out);
    }
    if (dx) {
        CNN_REQUIRE(w != nullptr, "cnn_linear_backward: w is null");
        dim3 grid(ceil_div(in, kBlock) > 64 ? 64 : ceil_div(in, kBlock), B);
        CNN_KLAUNCH(s, "linear_bwd_x", (linear_bwd_x<<<grid, kBlock, 0, s>>>(dy, w, dx, in, out)), "B%d in%d out%d", B, in, out);
    }
    if (relu_below && dx) return cnn_relu_backward(x, dx, (size_t)B * in, stream);  // (same result from the separate kernel)
    return CNN_AMD_OK;
}

int cnn_linear_forward_softmax_xent(const float* x, const float* w, const float* bias, const int32_t* labels, float* logits,
                                    float* probs, float* delta, float* loss_terms, int B, int in, int out, void* stream) {
    CNN_REQUIRE(x && w && bias && labels && logits && delta && loss_terms, "cnn_linear_forward_softmax_xent: null pointer");
    CNN_REQUIRE(B > 0 && in > 0 && out > 0 && out <= kOutTile, "cnn_linear_forward_softmax_xent: B=%d in=%d out=%d (out <= %d)", B, in, out,
                kOutTile);
    hipStream_t s = as_stream(stream);
    const OptVal e = CNN_OPT_VAL("NO_HEAD_BATCH");  // A/B switch: the plain 6-deep loop for every layer width
    if (e && atoi(e) != 0)
        CNN_KLAUNCH(s, "linear_fwd+softmax_xent",
                    (linear_fwd_softmax_xent<false, 0><<<B, kBlock, 0, s>>>(x, w, bias, labels, logits, probs, delta, loss_terms, in, out, nullptr)),
                    "B%d in%d out%d", B, in, out);
    else
        CNN_KLAUNCH(s, "linear_fwd+softmax_xent",
                    (linear_fwd_softmax_xent<true, 0><<<B, kBlock, 0, s>>>(x, w, bias, labels, logits, probs, delta, loss_terms, in, out, nullptr)),
                    "B%d in%d out%d", B, in, out);
    return CNN_AMD_OK;
}

int cnn_linear_forward_softmax_xent_dx(const float* x, const float* w, const float* bias, const int32_t* labels, float* logits,
                                       float* probs, float* delta, float* loss_terms, float* dx, int relu_below, int B, int in, int out,
                                       void* stream) {
    CNN_REQUIRE(x && w && bias && labels && logits && delta && loss_terms && dx, "cnn_linear_forward_softmax_xent_dx: null pointer");
    CNN_REQUIRE(B > 0 && in > 0 && out > 0 && out <= kOutTile, "cnn_linear_forward_softmax_xent_dx: B=%d in=%d out=%d (out <= %d)", B, in, out,
                kOutTile);
    hipStream_t s = as_stream(stream);
    if (relu_below)
        CNN_KLAUNCH(s, "linear_fwd+softmax_xent+dx+relu",
                    (launch_pub(linear_fwd_softmax_xent<true, 2>, dim3(B), dim3(kBlock), 0, s, x, w, bias, labels, logits, probs, delta, loss_terms, in, out, dx)),
                    "B%d in%d out%d", B, in, out);
    else
        CNN_KLAUNCH(s, "linear_fwd+softmax_xent+dx",
                    (launch_pub(linear_fwd_softmax_xent<true, 1>, dim3(B), dim3(kBlock), 0, s, x, w, bias, labels, logits, probs, delta, loss_terms, in, out, dx)),
                    "B%d in%d out%d", B, in, out);
    return CNN_AMD_OK;
}

int cnn_loss_from_terms(const float* loss_terms, float* loss_sum, int B, void* stream) {
    CNN_REQUIRE(loss_terms && loss_sum && B > 0, "cnn_loss_from_terms: bad arguments");
    hipStream_t s = as_stream(stream);
    CNN_KLAUNCH(s, "loss_from_terms", (loss_from_terms<<<1, 64, 0, s>>>(loss_terms, loss_sum, B)), "B=%d", B);
    return CNN_AMD_OK;
}

int cnn_linear_backward(const float* x, const float* dy, const float* w, float* gw, float* gb, float* dx, int B,
                        int in, int out, float divisor, void* stream) {
    return linear_backward_impl(x, dy, w, gw, gb, dx, B, in, out, divisor, stream, false);
}

int cnn_linear_backward_relu(const float* x, const float* dy, const float* w, float* gw, float* gb, float* dx, int B,
                             int in, int out, float divisor, void* stream) {
    CNN_REQUIRE(x && dx, "cnn_linear_backward_relu: null pointer");
    return linear_backward_impl(x, dy, w, gw, gb, dx, B, in, out, divisor, stream, true);
}

}  // extern "C"
