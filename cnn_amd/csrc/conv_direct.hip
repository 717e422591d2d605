// conv_direct.hip -- direct (non-GEMM) Conv2D forward / data-gradient kernels for THIN first layers (Ci <= 4):
// the reference net's conv_layer_1 (3 -> 16, 3x3, stride 2; alexnet.cpp:12) has K = Ci*k*k = 27 and 7.7 FLOP/B, i.e. it
// is HBM-bound and a poor fit for 16x16x4 / 32x32x2 MFMA tiles.  Here every lane owns output pixels,
// the filters are wave-uniform (scalar loads -> SGPR operands of v_fmac), there is no LDS, no barrier and no weight
// re-layout kernel; loads and stores are contiguous runs of the NCHW rows.
//   forward (conv2d.cpp:69-92): lane = one output pixel (p,q), CO accumulators, sum order ci -> kx -> ky then + bias
//                               (the reference's order, conv2d.cpp:78-87);
//   dgrad   (conv2d.cpp:168-199 as a gather): lane = one pixel (hh,ww) of the ceil(H/s) x ceil(W/s) grid and produces the
//                               s*s*CI outputs dx[ci][s*hh+ph][s*ww+pw] from the ceil(k/s)^2 window of dy; the S outputs
//                               of a row are adjacent in memory and stored together.
// Roofline: HBM.  forward 4*(B*Ci*H*W + B*Co*Ho*Wo) bytes, dgrad 4*(B*Co*Ho*Wo + B*Ci*H*W) bytes per launch.
#include <cstdlib>

#include "common.h"

using namespace cnn_amd;

namespace {

constexpr int kWaves = 4;
constexpr int kBlock = 64 * kWaves;

// one wave per output row (b, p); lanes stride over q
template <int CI, int CO, int K, int S>
__global__ __launch_bounds__(kBlock) void conv_direct_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          float* __restrict__ y_relu, int B, int H, int W, int Ho,
                                                          int Wo) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long rows = (long long)B * Ho;
    for (long long r = (long long)blockIdx.x * kWaves + wave; r < rows; r += (long long)gridDim.x * kWaves) {
        const int b = (int)(r / Ho), p = (int)(r - (long long)b * Ho);
        const float* xb = x + (size_t)b * CI * H * W + (size_t)(p * S) * W;
        float* yb = y + (size_t)b * CO * Ho * Wo + (size_t)p * Wo;
        float* rb = y_relu ? y_relu + (size_t)b * CO * Ho * Wo + (size_t)p * Wo : nullptr;  // fused ReLU::forward (relu.cpp:25)
        for (int q = lane; q < Wo; q += 64) {
            // the pixel's CI x K x K patch lives in registers; output channels are the outer loop so that each channel's
            // CI*K*K filter taps are one contiguous (scalar) load
            float patch[CI * K * K];
#pragma unroll
            for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const float* xr = xb + (size_t)ci * H * W + (size_t)kx * W + q * S;
#pragma unroll
                    for (int ky = 0; ky < K; ++ky) patch[(ci * K + kx) * K + ky] = xr[ky];
                }
#pragma unroll 4
            for (int co = 0; co < CO; ++co) {
                const float* wc = w + (size_t)co * CI * K * K;  // wave-uniform
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < CI * K * K; ++t) acc = fmaf(patch[t], wc[t], acc);  // ci -> kx -> ky, like conv2d.cpp:80-84
                const float v = acc + bias[co];
                yb[(size_t)co * Ho * Wo + q] = v;
                if (rb) rb[(size_t)co * Ho * Wo + q] = v >= 0.f ? v : 0.f;
            }
        }
    }
}

// one wave per grid row (b, hh); lanes stride over ww.  pad == 0.
template <int CI, int CO, int K, int S>
__global__ __launch_bounds__(kBlock) void conv_direct_dgrad(const float* __restrict__ dy, const float* __restrict__ w,
                                                            float* __restrict__ dx, int B, int H, int W, int Ho, int Wo) {
    constexpr int J = (K + S - 1) / S;  // dy rows / cols a grid pixel reads: hh - j, ww - j for j < J
    const int U = (H + S - 1) / S, V = (W + S - 1) / S;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long rows = (long long)B * U;
    for (long long r = (long long)blockIdx.x * kWaves + wave; r < rows; r += (long long)gridDim.x * kWaves) {
        const int b = (int)(r / U), hh = (int)(r - (long long)b * U);
        const float* dyb = dy + (size_t)b * CO * Ho * Wo;
        float* dxb = dx + (size_t)b * CI * H * W;
        for (int ww = lane; ww < V; ww += 64) {
            float acc[S][S][CI];
#pragma unroll
            for (int ph = 0; ph < S; ++ph)
#pragma unroll
                for (int pw = 0; pw < S; ++pw)
#pragma unroll
                    for (int ci = 0; ci < CI; ++ci) acc[ph][pw][ci] = 0.f;
#pragma unroll 4
            for (int co = 0; co < CO; ++co) {
                float d[J][J];
#pragma unroll
                for (int jr = 0; jr < J; ++jr)
#pragma unroll
                    for (int jc = 0; jc < J; ++jc) {
                        const int pr = hh - jr, pc = ww - jc;
                        d[jr][jc] = (pr >= 0 && pr < Ho && pc >= 0 && pc < Wo) ? dyb[((size_t)co * Ho + pr) * Wo + pc] : 0.f;
                    }
                const float* wc = w + (size_t)co * CI * K * K;  // wave-uniform
#pragma unroll
                for (int ph = 0; ph < S; ++ph)
#pragma unroll
                    for (int pw = 0; pw < S; ++pw)
#pragma unroll
                        for (int jr = 0; jr < J; ++jr)
#pragma unroll
                            for (int jc = 0; jc < J; ++jc) {
                                constexpr int dummy = 0;
                                (void)dummy;
                                const int kx = ph + S * jr, ky = pw + S * jc;  // compile-time after unrolling
                                if (kx < K && ky < K) {
#pragma unroll
                                    for (int ci = 0; ci < CI; ++ci)
                                        acc[ph][pw][ci] = fmaf(wc[(ci * K + kx) * K + ky], d[jr][jc], acc[ph][pw][ci]);
                                }
                            }
            }
#pragma unroll
            for (int ci = 0; ci < CI; ++ci)
#pragma unroll
                for (int ph = 0; ph < S; ++ph) {
                    const int h = hh * S + ph;
                    if (h >= H) continue;
                    float* row = dxb + ((size_t)ci * H + h) * W + (size_t)ww * S;
                    if (S == 2 && (W & 1) == 0) {  // both outputs of the row exist and the pair is 8-byte aligned
                        *(float2*)row = make_float2(acc[ph][0][ci], acc[ph][S - 1][ci]);
                    } else {
#pragma unroll
                        for (int pw = 0; pw < S; ++pw)
                            if (ww * S + pw < W) row[pw] = acc[ph][pw][ci];
                    }
                }
        }
    }
}

inline unsigned wave_grid(long long rows) {
    long long need = (rows + kWaves - 1) / kWaves;
    const long long cap = (long long)kNumCU * 32;
    return (unsigned)(need < 1 ? 1 : (need > cap ? cap : need));
}

}  // namespace

namespace cnn_amd {

#define CONV_TAG(d) "B%d Ci%d %dx%d Co%d k%d s%d p%d", (d)->B, (d)->Ci, (d)->H, (d)->W, (d)->Co, (d)->k, (d)->s, (d)->pad

// returns 1 when the geometry has a direct kernel (and it was launched), 0 when the caller must use the implicit GEMM,
// < 0 on error
bool direct_conv_supported(const cnn_conv2d_desc* d) {
    return d->Ci == 3 && d->Co == 16 && d->k == 3 && d->s == 2 && d->pad == 0 && !getenv("CNN_AMD_NO_DIRECT");
}

int direct_conv_forward(const cnn_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y,
                        float* y_relu, hipStream_t s) {
    const int Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, 0), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, 0);
    const long long rows = (long long)d->B * Ho;
    CNN_KLAUNCH(s, y_relu ? "conv_direct_fwd<3,16,3,2>+relu" : "conv_direct_fwd<3,16,3,2>",
                (conv_direct_fwd<3, 16, 3, 2><<<wave_grid(rows), kBlock, 0, s>>>(x, w, bias, y, y_relu, d->B, d->H, d->W, Ho, Wo)),
                CONV_TAG(d));
    return CNN_AMD_OK;
}

int direct_conv_dgrad(const cnn_conv2d_desc* d, const float* dy, const float* w, float* dx, hipStream_t s) {
    const int Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, 0), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, 0);
    const long long rows = (long long)d->B * ((d->H + d->s - 1) / d->s);
    CNN_KLAUNCH(s, "conv_direct_dgrad<3,16,3,2>",
                (conv_direct_dgrad<3, 16, 3, 2><<<wave_grid(rows), kBlock, 0, s>>>(dy, w, dx, d->B, d->H, d->W, Ho, Wo)),
                CONV_TAG(d));
    return CNN_AMD_OK;
}

}  // namespace cnn_amd
