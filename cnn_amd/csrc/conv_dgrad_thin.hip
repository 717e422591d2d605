// conv_dgrad_thin.hip -- Conv2D data gradient (cpu/src/conv2d.cpp:168-199, as a gather) for a THIN INPUT: Ci = 3, 3x3, stride 1,
// padding 0 or 1 -- the first layer of the VGG-shaped stack (3 -> 64 at 224x224; BASELINE configs[3]).  The reference computes
// this gradient although nothing consumes it (alexnet.cpp:55 discards it), so it has to be produced, and the implicit GEMM is
// the wrong tool: M = Ci = 3 of 16 MFMA rows, 7 TFLOP/s, 3.1 ms at batch 128.  HBM-bound: dy (B*Co*Ho*Wo floats) is read once,
// dx (B*3*H*W) written once; 2*27*Co FLOP per input pixel.
//
//   dx[b][ci][h][w] = sum_co sum_{kx,ky} w[co][ci][kx][ky] * dy[b][co][h + pad - kx][w + pad - ky]      (zero outside dy)
//
// VALU kernel: a lane owns RH = 4 vertically adjacent pixels of one column (12 running sums), a wave a 4 x 64 pixel patch of
// one image.  Per dy channel it loads the (RH + 2) x 3 values its pixels' windows cover (row segments of 256 B, neighbouring
// lanes / waves overlap in L1) and spends 4*27 FMAs on them with the 27 filter taps of that channel as SCALAR operands (the
// filters are wave-uniform: s_load, no LDS, no re-layout: the "prepared" image of such a layer is a verbatim copy of w).
// Rows outside dy are skipped wave-uniformly, columns outside it selected to 0 per lane.  An optional fused ReLU::backward
// (relu.cpp:35-40) masks dx by the output of the ReLU layer in front.
#include <cstdlib>

#include "common.h"

using namespace cnn_amd;

namespace {

constexpr int kThinWaves = 4;

template <int CI, int RH>
__global__ __launch_bounds__(kThinWaves * 64) void conv_dgrad_thin_s1k3(const float* __restrict__ dy, const float* __restrict__ w,
                                                                        const float* __restrict__ relu_below, float* __restrict__ dx, int B,
                                                                        int Co, int H, int W, int Ho, int Wo, int pad, int bands, int segs) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * kThinWaves + (threadIdx.x >> 6));
    const int per_img = bands * segs;
    const int b = wid / per_img;
    if (b >= B) return;
    const int r = wid - b * per_img;
    const int band = r / segs, seg = r - band * segs;
    const int h0 = band * RH, wc = seg * 64 + lane;
    // the (RH + 2) x 3 patch of dy this lane's windows cover: rows r0 .. r0 + RH + 1, columns wc + pad - ky.  Offsets inside a
    // channel plane are computed ONCE (32-bit, added to a scalar channel base: no address arithmetic in the loop); an element
    // outside dy re-reads element 0 and is selected to 0.
    const int r0 = h0 + pad - 2;
    unsigned voff[RH + 2][3];
    bool ok[RH + 2][3];
#pragma unroll
    for (int j = 0; j < RH + 2; ++j)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int rr = r0 + j, c = wc + pad - ky;
            ok[j][ky] = wc < W && (unsigned)rr < (unsigned)Ho && (unsigned)c < (unsigned)Wo;
            voff[j][ky] = ok[j][ky] ? (unsigned)(rr * Wo + c) : 0u;
        }
    float acc[RH][CI];
#pragma unroll
    for (int i = 0; i < RH; ++i)
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) acc[i][ci] = 0.f;
    const size_t oplane = (size_t)Ho * Wo;
    const float* dyc = dy + (size_t)b * Co * oplane;  // wave-uniform, advanced per channel
    const float* wc0 = w;                              // wave-uniform: scalar loads
    auto load_patch = [&](float (&v)[RH + 2][3], const float* base) {
#pragma unroll
        for (int j = 0; j < RH + 2; ++j)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) v[j][ky] = base[voff[j][ky]];
    };
    auto accumulate = [&](const float (&raw)[RH + 2][3], const float* wt) {
        float v[RH + 2][3];
#pragma unroll
        for (int j = 0; j < RH + 2; ++j)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) v[j][ky] = ok[j][ky] ? raw[j][ky] : 0.f;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float wv = wt[(ci * 3 + kx) * 3 + ky];
#pragma unroll
                    for (int i = 0; i < RH; ++i) acc[i][ci] = __builtin_fmaf(wv, v[i + 2 - kx][ky], acc[i][ci]);  // dy row h0+i+pad-kx
                }
    };
    // two channels per turn: the second channel's 18 loads are in flight while the first one's FMAs run
    int co = 0;
    for (; co + 1 < Co; co += 2) {
        float va[RH + 2][3], vb[RH + 2][3];
        load_patch(va, dyc);
        load_patch(vb, dyc + oplane);
        accumulate(va, wc0);
        accumulate(vb, wc0 + CI * 9);
        dyc += 2 * oplane;
        wc0 += 2 * CI * 9;
    }
    if (co < Co) {
        float va[RH + 2][3];
        load_patch(va, dyc);
        accumulate(va, wc0);
    }
    if (wc < W) {
#pragma unroll
        for (int i = 0; i < RH; ++i) {
            const int h = h0 + i;
            if (h < H) {
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) {
                    const size_t o = (((size_t)b * CI + ci) * H + h) * W + wc;
                    float val = acc[i][ci];
                    if (relu_below) val = (relu_below[o] <= 0.f) ? 0.f : val;
                    dx[o] = val;
                }
            }
        }
    }
}

}  // namespace

namespace cnn_amd {

#define CONV_TAG(d) "B%d Ci%d %dx%d Co%d k%d s%d p%d", (d)->B, (d)->Ci, (d)->H, (d)->W, (d)->Co, (d)->k, (d)->s, (d)->pad

bool thin_dgrad_supported(const cnn_conv2d_desc* d) {
    const char* e = getenv("CNN_AMD_DGRAD_THIN");
    if (e && atoi(e) == 0) return false;
    return d->Ci == 3 && d->k == 3 && d->s == 1 && d->pad >= 0 && d->pad <= 1 && (long long)d->B * ((d->H + 3) / 4) * ((d->W + 63) / 64) < (1ll << 31) - 8;
}

// w: the filters in the reference layout [Co][3][3][3] (for the *_prepared entry points: the verbatim copy cnn_conv2d_prepare_filters made)
int thin_dgrad(const cnn_conv2d_desc* d, const float* dy, const float* w, const float* relu_below, float* dx, hipStream_t s) {
    const int Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, d->pad), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, d->pad);
    const int bands = (d->H + 3) / 4, segs = (d->W + 63) / 64;
    const long long waves = (long long)d->B * bands * segs;
    const unsigned grid = (unsigned)((waves + kThinWaves - 1) / kThinWaves);
    CNN_KLAUNCH(s, relu_below ? "conv_dgrad_thin<3,s1>+relu" : "conv_dgrad_thin<3,s1>",
                (conv_dgrad_thin_s1k3<3, 4><<<grid, kThinWaves * 64, 0, s>>>(dy, w, relu_below, dx, d->B, d->Co, d->H, d->W, Ho, Wo, d->pad, bands, segs)),
                CONV_TAG(d));
    return CNN_AMD_OK;
}

}  // namespace cnn_amd
