// linear.hip -- LinearLayer forward / backward (cpu/src/linear.cpp:33-43, 56-90).  W is [in][out] row-major.
// The reference net's layer is 4608 -> 3: a skinny contraction that is HBM-bound on x (4*in bytes per sample),
// so these are wave-reduction / streaming kernels, not MFMA tiles.
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
        for (int i = threadIdx.x; i < in; i += kBlock) {
            const float xv = xb[i];
            const float* wr = w + (size_t)i * out + j0;
#pragma unroll
            for (int j = 0; j < kOutTile; ++j)
                if (j < nj) acc[j] += xv * wr[j];
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

// gW[i][j] = (sum_b x[b][i]*dy[b][j]) / divisor.  One thread per input neuron i (coalesced over i for every b),
// dy rows broadcast from LDS in tiles of samples.
constexpr int kBTile = 64;
__global__ __launch_bounds__(kBlock) void linear_bwd_w(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ gw, int B, int in, int out,
                                                       float divisor) {
    __shared__ float dtile[kBTile * kOutTile];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    for (int j0 = 0; j0 < out; j0 += kOutTile) {
        const int nj = min(kOutTile, out - j0);
        float acc[kOutTile];
#pragma unroll
        for (int j = 0; j < kOutTile; ++j) acc[j] = 0.f;
        for (int b0 = 0; b0 < B; b0 += kBTile) {
            const int nb = min(kBTile, B - b0);
            __syncthreads();
            for (int t = threadIdx.x; t < nb * kOutTile; t += kBlock) {
                const int bb = t / kOutTile, j = t % kOutTile;
                dtile[t] = (j < nj) ? dy[(size_t)(b0 + bb) * out + j0 + j] : 0.f;
            }
            __syncthreads();
            if (i < in) {
                for (int bb = 0; bb < nb; ++bb) {
                    const float xv = x[(size_t)(b0 + bb) * in + i];
#pragma unroll
                    for (int j = 0; j < kOutTile; ++j) acc[j] += xv * dtile[bb * kOutTile + j];
                }
            }
        }
        if (i < in)
            for (int j = 0; j < nj; ++j) gw[(size_t)i * out + j0 + j] = acc[j] / divisor;
    }
}

// gb[j] = (sum_b dy[b][j]) / divisor: sequential over b like linear.cpp:66-71 (B*out is tiny)
__global__ void linear_bwd_b(const float* __restrict__ dy, float* __restrict__ gb, int B, int out, float divisor) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= out) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dy[(size_t)b * out + j];
    gb[j] = s / divisor;
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

int cnn_linear_backward(const float* x, const float* dy, const float* w, float* gw, float* gb, float* dx, int B,
                        int in, int out, float divisor, void* stream) {
    CNN_REQUIRE(dy != nullptr, "cnn_linear_backward: dy is null");
    CNN_REQUIRE(B > 0 && in > 0 && out > 0, "cnn_linear_backward: B=%d in=%d out=%d", B, in, out);
    CNN_REQUIRE(B <= 65535, "cnn_linear_backward: B=%d exceeds the grid.y limit", B);
    hipStream_t s = as_stream(stream);
    if (gw) {
        CNN_REQUIRE(x != nullptr, "cnn_linear_backward: x is null");
        CNN_KLAUNCH(s, "linear_bwd_w", (linear_bwd_w<<<ceil_div(in, kBlock), kBlock, 0, s>>>(x, dy, gw, B, in, out, divisor)),
                    "B%d in%d out%d", B, in, out);
    }
    if (gb) {
        CNN_KLAUNCH(s, "linear_bwd_b", (linear_bwd_b<<<ceil_div(out, 64), 64, 0, s>>>(dy, gb, B, out, divisor)), "B%d out%d", B,
                    out);
    }
    if (dx) {
        CNN_REQUIRE(w != nullptr, "cnn_linear_backward: w is null");
        dim3 grid(ceil_div(in, kBlock) > 64 ? 64 : ceil_div(in, kBlock), B);
        CNN_KLAUNCH(s, "linear_bwd_x", (linear_bwd_x<<<grid, kBlock, 0, s>>>(dy, w, dx, in, out)), "B%d in%d out%d", B, in, out);
    }
    return CNN_AMD_OK;
}

}  // extern "C"
