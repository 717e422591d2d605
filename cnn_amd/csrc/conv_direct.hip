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
template <int CI, int CO, int K, int S, int UN>
__global__ __launch_bounds__(kBlock) void conv_direct_dgrad(const float* __restrict__ dy, const float* __restrict__ w,
                                                            float* __restrict__ dx, int B, int H, int W, int Ho, int Wo, int dbg) {
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
#pragma unroll(UN)
            for (int co = 0; co < CO; ++co) {
                float d[J][J];
#pragma unroll
                for (int jr = 0; jr < J; ++jr)
#pragma unroll
                    for (int jc = 0; jc < J; ++jc) {
                        const int pr = hh - jr, pc = ww - jc;
                        if ((dbg & 1) && (jr | jc)) { d[jr][jc] = d[0][0]; continue; }
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
            if (dbg & 8) { if (acc[0][0][0] == 123.456f) dxb[0] = 1.f; continue; }
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

// ---- packed-math data gradient for the 3 -> 16, 3x3, stride-2 layer ------------------------------------------------
// PMC counters of conv_direct_dgrad (profiles/r01) show the kernel is bound by instruction issue, not memory: 810 VALU
// and 745 SALU instructions per 64 pixels against 272 useful packed FMAs -- hipcc shuffles the wave-uniform weights into
// SGPR pairs with s_mov and spills SGPRs through v_writelane.  Here the weights are re-packed ONCE per call into the
// exact operand order of the inner loop (pack_dgrad_weights: 16 float2 per output channel), so each v_pk_fma_f32 takes
// an s_load'ed SGPR pair directly: 15 packed FMAs per channel, no scalar shuffling.
// Accumulator pairs of one grid pixel (class = output parity (ph,pw), alexnet.cpp:12 geometry):
//   P[cls] = (ci0, ci1) of class cls;  Q0 = ci2 of classes (0,0),(0,1);  Q1 = ci2 of classes (1,0),(1,1).
typedef float v2f __attribute__((ext_vector_type(2)));
// two adjacent floats stored with ONE dwordx2 even when only 4-byte aligned (odd row pitch): global memory accesses may
// be unaligned on gfx950, and hipcc emits global_store_dwordx2 for this type
struct __attribute__((packed, aligned(4))) f2u { float x, y; };
struct __attribute__((packed, aligned(4))) f3u { float x, y, z; };

__device__ void pack_dgrad_3_16_3_2_body(const float* __restrict__ w, float* __restrict__ wp, int tid) {
    const int co = tid;
    if (co >= 16) return;
    const float* wc = w + (size_t)co * 27;
    auto W = [&](int ci, int kx, int ky) { return wc[(ci * 3 + kx) * 3 + ky]; };
    float* o = wp + (size_t)co * 32;
    int n = 0;
    auto put = [&](float a, float b) { o[n++] = a; o[n++] = b; };
    // d00 = dy[hh][ww]: tap (kx,ky) = (ph,pw)
    for (int cls = 0; cls < 4; ++cls) put(W(0, cls >> 1, cls & 1), W(1, cls >> 1, cls & 1));
    put(W(2, 0, 0), W(2, 0, 1));
    put(W(2, 1, 0), W(2, 1, 1));
    // d01 = dy[hh][ww-1]: classes with pw = 0, ky = 2
    put(W(0, 0, 2), W(1, 0, 2));
    put(W(0, 1, 2), W(1, 1, 2));
    put(W(2, 0, 2), 0.f);
    put(W(2, 1, 2), 0.f);
    // d10 = dy[hh-1][ww]: classes with ph = 0, kx = 2
    put(W(0, 2, 0), W(1, 2, 0));
    put(W(0, 2, 1), W(1, 2, 1));
    put(W(2, 2, 0), W(2, 2, 1));
    // d11 = dy[hh-1][ww-1]: class (0,0), tap (2,2)
    put(W(0, 2, 2), W(1, 2, 2));
    put(W(2, 2, 2), 0.f);
    put(0.f, 0.f);
}

__global__ void pack_dgrad_weights_3_16_3_2(const float* __restrict__ w, float* __restrict__ wp) {
    pack_dgrad_3_16_3_2_body(w, wp, blockIdx.x * blockDim.x + threadIdx.x);
}

// dy is read through a raw buffer descriptor: address = SGPR base + SGPR offset (image, channel, row) + per-lane byte
// offset, i.e. NO vector address arithmetic per load, and a lane whose tap lies outside the image uses an offset beyond
// num_records, for which the hardware returns 0 -- no masks, no branches (requires the dy tensor to be < 2 GiB).
constexpr unsigned kBufOOB = 0x7ffffffcu;

// n / d for the small non-negative operands of the item -> (image, pixel) -> (row, column) decodes below, without the ~40
// instruction integer-division sequence: magic = 2^32 / d + 1 (host side), one correction step covers the rounding
__device__ __forceinline__ int fast_div(int n, unsigned magic, int d) {
    if (d == 1) return n;  // (2^32 / 1 does not fit the 32-bit magic)
    int q = (int)__umulhi((unsigned)n, magic);
    if (q * d > n) --q;
    return q;
}
inline unsigned div_magic(int d) { return (unsigned)((1ull << 32) / (unsigned)d) + 1u; }

// One wave-item = 64 consecutive grid pixels of one image (rows are crossed, so every lane is busy); CB channels per
// batch: all 4*CB loads of a batch are in flight before its first FMA.
template <int CB, int DBG = 0>
__global__ __launch_bounds__(kBlock) void conv_dgrad_pk_3_16_3_2(const float* __restrict__ dy, const v2f* __restrict__ wp,
                                                                 float* __restrict__ dx, int B, int H, int W, int Ho,
                                                                 int Wo, int items_per_img, unsigned m_ipi, unsigned m_row) {
    constexpr int CO = 16, CI = 3;
    const int U = (H + 1) / 2, V = (W + 1) / 2;
    const int UV = U * V;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long items = (long long)B * items_per_img;
    const int plane = Ho * Wo;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((unsigned)B * CO * plane * 4u), 0x00020000);
    for (int it = xcd_swizzle(blockIdx.x, gridDim.x) * kWaves + wave; it < (int)items; it += gridDim.x * kWaves) {
        const int b = fast_div(it, m_ipi, items_per_img);
        const int n = (it - b * items_per_img) * 64 + lane;
        const bool live = n < UV;
        const int hh = fast_div(live ? n : 0, m_row, V), ww = (live ? n : 0) - hh * V;
        // taps: dy rows hh (jr = 0) / hh-1 (jr = 1), columns ww (jc = 0) / ww-1 (jc = 1); a missing tap gets an out-of-range
        // offset and reads as 0
        const bool r0ok = live && hh < Ho, r1ok = live && hh >= 1, c0ok = ww < Wo, c1ok = ww >= 1;
        const unsigned o = (unsigned)(hh * Wo + ww) * 4u;
        const unsigned v00 = (r0ok && c0ok) ? o : kBufOOB;
        const unsigned v01 = (r0ok && c1ok) ? o - 4u : kBufOOB;
        const unsigned v10 = (r1ok && c0ok) ? o - (unsigned)Wo * 4u : kBufOOB;
        const unsigned v11 = (r1ok && c1ok) ? o - (unsigned)Wo * 4u - 4u : kBufOOB;
        const int soff = b * CO * plane * 4;  // byte offset of dy[b][0][0][0]; wave-uniform
        v2f P0 = {0.f, 0.f}, P1 = P0, P2 = P0, P3 = P0, Q0 = P0, Q1 = P0;
#pragma unroll 1
        for (int cg = 0; cg < CO; cg += CB) {
            float dv[CB][4];
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int so = soff + (cg + u) * plane * 4;
                dv[u][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)v00, so, 0));
                dv[u][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)v01, so, 0));
                dv[u][2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)v10, so, 0));
                dv[u][3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)v11, so, 0));
            }
            if constexpr (DBG & 4) {
#pragma unroll
                for (int u = 0; u < CB; ++u) P0.x += dv[u][0] + dv[u][1] + dv[u][2] + dv[u][3];
                continue;
            }
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                // (the fence keeps hipcc from hoisting the weight s_loads of all channels to the top, which would need
                //  hundreds of SGPRs and spill them through v_writelane)
                asm volatile("" ::: "memory");
                const v2f* q = wp + (cg + u) * 16;  // wave-uniform -> scalar loads straight into SGPR pairs
                const v2f a = {dv[u][0], dv[u][0]}, bq = {dv[u][1], dv[u][1]}, c = {dv[u][2], dv[u][2]},
                          e = {dv[u][3], dv[u][3]};
                P0 = __builtin_elementwise_fma(q[0], a, P0);
                P1 = __builtin_elementwise_fma(q[1], a, P1);
                P2 = __builtin_elementwise_fma(q[2], a, P2);
                P3 = __builtin_elementwise_fma(q[3], a, P3);
                Q0 = __builtin_elementwise_fma(q[4], a, Q0);
                Q1 = __builtin_elementwise_fma(q[5], a, Q1);
                P0 = __builtin_elementwise_fma(q[6], bq, P0);
                P2 = __builtin_elementwise_fma(q[7], bq, P2);
                Q0 = __builtin_elementwise_fma(q[8], bq, Q0);
                Q1 = __builtin_elementwise_fma(q[9], bq, Q1);
                P0 = __builtin_elementwise_fma(q[10], c, P0);
                P1 = __builtin_elementwise_fma(q[11], c, P1);
                Q0 = __builtin_elementwise_fma(q[12], c, Q0);
                P0 = __builtin_elementwise_fma(q[13], e, P0);
                Q0 = __builtin_elementwise_fma(q[14], e, Q0);
            }
        }
        if (!live) continue;
        if ((DBG & 8) && P0.x != 123.25f) continue;
        // class (ph,pw) -> dx[ci][2hh+ph][2ww+pw]; the two pw of a row are adjacent in memory
        float* dxb = dx + (size_t)b * CI * H * W;
        const int h0 = 2 * hh;
        const size_t col = (size_t)ww * 2;
        const bool pair = ((W & 1) == 0);  // then both outputs of a row exist and the pair is 8-byte aligned
        auto st = [&](int ci, int ph, float v0, float v1) {
            const int h = h0 + ph;
            if (h >= H) return;
            float* row = dxb + ((size_t)ci * H + h) * W + col;
            if (pair) {
                *(float2*)row = make_float2(v0, v1);
            } else {
                row[0] = v0;
                if (col + 1 < (size_t)W) row[1] = v1;
            }
        };
        st(0, 0, P0.x, P1.x); st(0, 1, P2.x, P3.x);
        st(1, 0, P0.y, P1.y); st(1, 1, P2.y, P3.y);
        st(2, 0, Q0.x, Q0.y); st(2, 1, Q1.x, Q1.y);
    }
}

// ---- conv_layer_1's data gradient straight from the pooled domain ------------------------------------------------------
// The delta of the convolution output, dy = ReLU::backward(MaxPool2D::backward(dpool)), has at most one non-zero per 2x2
// pooling window and is fully described by (dpool, mask, pooled).  Each lane owns a 4x4 block of dx (= the 2x2 blocks
// hh = 2bh + {0,1}, ww = 2bw + {0,1} of conv_dgrad_pk_3_16_3_2) for all three input channels; the 3x3 dy neighbourhood it
// needs (rows 2bh-1 .. 2bh+1, columns 2bw-1 .. 2bw+1) lies in the four windows (bh-1 | bh) x (bw-1 | bw): three loads per
// window and channel (12 for four 2x2 blocks instead of the 16 dy loads of the unfused kernel), the same packed FMAs in
// the same order (bit-identical dx), one weight s_load per FOUR blocks, 16-byte stores.
template <int CB, bool RM>  // RM: apply the ReLU mask from `pooled` (false: dpool is pre-masked, `pooled` is not read)
__global__ __launch_bounds__(kBlock) void conv_dgrad_pool_pk_3_16_3_2(const float* __restrict__ dpool, const int32_t* __restrict__ pmask,
                                                                      const float* __restrict__ pooled, const v2f* __restrict__ wp,
                                                                      float* __restrict__ dx, int B, int H, int W, int Ho, int Wo,
                                                                      int items_per_img, unsigned m_ipi, unsigned m_row) {
    constexpr int CO = 16, CI = 3;
    const int U = (H + 1) / 2, V = (W + 1) / 2, U2 = (U + 1) / 2, V2 = (V + 1) / 2;
    const int PHo = Ho / 2, PWo = Wo / 2, pplane = PHo * PWo, plane = Ho * Wo;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long items = (long long)B * items_per_img;
    const int pbytes = (int)((unsigned)B * CO * pplane * 4u);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dpool, 0, pbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)pmask, 0, pbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(RM ? pooled : dpool), 0, pbytes, 0x00020000);
    for (int it = xcd_swizzle(blockIdx.x, gridDim.x) * kWaves + wave; it < (int)items; it += gridDim.x * kWaves) {
        const int b = fast_div(it, m_ipi, items_per_img);
        const int n = (it - b * items_per_img) * 64 + lane;
        const bool live = n < U2 * V2;
        const int bh = fast_div(live ? n : 0, m_row, V2), bw = (live ? n : 0) - bh * V2;
        // windows (bh-1 | bh) x (bw-1 | bw); a window that does not exist gets an out-of-range offset and reads as 0
        const bool r0 = live && bh >= 1 && bh - 1 < PHo, r1 = live && bh < PHo, c0 = bw >= 1 && bw - 1 < PWo, c1 = bw < PWo;
        const unsigned o = (unsigned)(bh * PWo + bw) * 4u;
        unsigned vw[4];
        vw[0] = (r0 && c0) ? o - (unsigned)PWo * 4u - 4u : kBufOOB;
        vw[1] = (r0 && c1) ? o - (unsigned)PWo * 4u : kBufOOB;
        vw[2] = (r1 && c0) ? o - 4u : kBufOOB;
        vw[3] = (r1 && c1) ? o : kBufOOB;
        const int e00 = (2 * bh - 1) * Wo + (2 * bw - 1);  // flat index (within a channel) of D[0][0]
        const int soff = b * CO * pplane * 4;
        v2f acc[4][6];
#pragma unroll
        for (int sb = 0; sb < 4; ++sb)
#pragma unroll
            for (int k = 0; k < 6; ++k) acc[sb][k] = v2f{0.f, 0.f};
#pragma unroll 1
        for (int cg = 0; cg < CO; cg += CB) {
            float g[CB][4], pl[CB][4];
            int mk[CB][4];
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int so = soff + (cg + u) * pplane * 4;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    g[u][w] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, (int)vw[w], so, 0));
                    mk[u][w] = __builtin_amdgcn_raw_buffer_load_b32(rm, (int)vw[w], so, 0);
                    if constexpr (RM) pl[u][w] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, (int)vw[w], so, 0));
                    else pl[u][w] = 1.f;
                }
            }
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                // D[i][j] = dy[co][2bh-1+i][2bw-1+j]: window row (i+1)/2, window column (j+1)/2 of the four above
                float D[3][3];
                const int cbase = (cg + u) * plane + e00;
#pragma unroll
                for (int w = 0; w < 4; ++w) g[u][w] = (pl[u][w] <= 0.f) ? 0.f : g[u][w];  // relu.cpp:38
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int w = ((i + 1) >> 1) * 2 + ((j + 1) >> 1);
                        D[i][j] = (mk[u][w] == cbase + i * Wo + j) ? g[u][w] : 0.f;  // pool2d.cpp:96-107
                    }
                asm volatile("" ::: "memory");  // (keeps the weight s_loads of all channels from being hoisted, see above)
                const v2f* q = wp + (cg + u) * 16;
#pragma unroll
                for (int sb = 0; sb < 4; ++sb) {
                    const int sr = sb >> 1, sc = sb & 1;
                    const v2f a = {D[sr + 1][sc + 1], D[sr + 1][sc + 1]}, bq = {D[sr + 1][sc], D[sr + 1][sc]},
                              c = {D[sr][sc + 1], D[sr][sc + 1]}, e = {D[sr][sc], D[sr][sc]};
                    v2f& P0 = acc[sb][0]; v2f& P1 = acc[sb][1]; v2f& P2 = acc[sb][2]; v2f& P3 = acc[sb][3];
                    v2f& Q0 = acc[sb][4]; v2f& Q1 = acc[sb][5];
                    P0 = __builtin_elementwise_fma(q[0], a, P0);
                    P1 = __builtin_elementwise_fma(q[1], a, P1);
                    P2 = __builtin_elementwise_fma(q[2], a, P2);
                    P3 = __builtin_elementwise_fma(q[3], a, P3);
                    Q0 = __builtin_elementwise_fma(q[4], a, Q0);
                    Q1 = __builtin_elementwise_fma(q[5], a, Q1);
                    P0 = __builtin_elementwise_fma(q[6], bq, P0);
                    P2 = __builtin_elementwise_fma(q[7], bq, P2);
                    Q0 = __builtin_elementwise_fma(q[8], bq, Q0);
                    Q1 = __builtin_elementwise_fma(q[9], bq, Q1);
                    P0 = __builtin_elementwise_fma(q[10], c, P0);
                    P1 = __builtin_elementwise_fma(q[11], c, P1);
                    Q0 = __builtin_elementwise_fma(q[12], c, Q0);
                    P0 = __builtin_elementwise_fma(q[13], e, P0);
                    Q0 = __builtin_elementwise_fma(q[14], e, Q0);
                }
            }
        }
        if (!live) continue;
        // dx[ci][4bh + 2sr + ph][4bw + 2sc + pw]: one row of the 4x4 block = the (pw0, pw1) pairs of sub-blocks sc = 0, 1
        float* dxb = dx + (size_t)b * CI * H * W;
        const bool quad = (W & 3) == 0;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int sr = rr >> 1, ph = rr & 1, h = 4 * bh + rr;
                float v[4];
#pragma unroll
                for (int sc = 0; sc < 2; ++sc) {
                    const v2f* A = acc[sr * 2 + sc];
                    if (ci == 2) {
                        v[2 * sc] = ph ? A[5].x : A[4].x;
                        v[2 * sc + 1] = ph ? A[5].y : A[4].y;
                    } else {
                        const v2f lo = ph ? A[2] : A[0], hi = ph ? A[3] : A[1];
                        v[2 * sc] = ci ? lo.y : lo.x;
                        v[2 * sc + 1] = ci ? hi.y : hi.x;
                    }
                }
                if (h >= H) continue;
                float* row = dxb + ((size_t)ci * H + h) * W + 4 * (size_t)bw;
                if (quad) {
                    *(float4*)row = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (4 * bw + k < W) row[k] = v[k];
                }
            }
    }
}

// ---- the same recipe for wider stride-2 3x3 layers (even CI): data gradient ---------------------------------------------
// On gfx950 the packed fp32 VALU rate equals the fp32 MFMA rate (157 TFLOP/s), and for the reference net's small layers
// the MFMA kernels spend most of their time staging operands through LDS.  Here the weights stream through SGPR pairs
// and dy comes straight from L1/L2: per dy channel 9 (tap, class) groups x CI/2 packed FMAs on the 4 x CI running sums
// of one grid pixel.  Group order (pack_dgrad_weights_s2): tap(0,0) classes 0..3 | tap(0,1) classes 0,2 | tap(1,0)
// classes 0,1 | tap(1,1) class 0; class = ph*2 + pw reads filter tap (kx,ky) = (ph + 2*jr, pw + 2*jc).
__device__ void pack_dgrad_s2_body(const float* __restrict__ w, float* __restrict__ wp, int CI, int CO, int tid, int nt) {
    const int gcls[9] = {0, 1, 2, 3, 0, 2, 0, 1, 0}, gjr[9] = {0, 0, 0, 0, 0, 0, 1, 1, 1}, gjc[9] = {0, 0, 0, 0, 1, 1, 0, 0, 1};
    const int total = CO * 9 * CI;
    for (int i = tid; i < total; i += nt) {
        const int ci = i % CI, g = (i / CI) % 9, co = i / (9 * CI);
        const int cls = gcls[g], kx = (cls >> 1) + 2 * gjr[g], ky = (cls & 1) + 2 * gjc[g];
        wp[i] = w[((size_t)(co * CI + ci) * 3 + kx) * 3 + ky];
    }
}

__global__ void pack_dgrad_weights_s2(const float* __restrict__ w, float* __restrict__ wp, int CI, int CO) {
    pack_dgrad_s2_body(w, wp, CI, CO, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

template <int CI, int CB, int PX>  // PX grid pixels per lane (1 in production; 2 shares the weight reads but gains nothing)
__global__ __launch_bounds__(kBlock) void conv_dgrad_pk_s2(const float* __restrict__ dy, const v2f* __restrict__ wp,
                                                           float* __restrict__ dx, const float* __restrict__ relu_below, int B,
                                                           int CO, int H, int W, int Ho, int Wo, int items_per_img,
                                                           unsigned m_ipi, unsigned m_row) {
    constexpr int HP = CI / 2;
    const int U = (H + 1) / 2, V = (W + 1) / 2;
    const int UV = U * V;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int items = B * items_per_img;
    const int plane = Ho * Wo;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((unsigned)B * CO * plane * 4u), 0x00020000);
    // the packed filters of ALL dy channels live in LDS (CO * 9 * CI floats) and are read with wave-uniform addresses
    // (broadcast ds_reads): 9*CI floats per channel are too many to stream through SGPRs without a scalar-load round
    // trip per group
    extern __shared__ __attribute__((aligned(16))) float wlds[];
    for (int i = threadIdx.x; i < CO * 9 * HP; i += kBlock) ((v2f*)wlds)[i] = wp[i];
    __syncthreads();
    for (int it = xcd_swizzle(blockIdx.x, gridDim.x) * kWaves + wave; it < items; it += gridDim.x * kWaves) {
        const int b = fast_div(it, m_ipi, items_per_img);
        const int soff = b * CO * plane * 4;
        int hh[PX], ww[PX];
        bool live[PX];
        unsigned v00[PX], v01[PX], v10[PX], v11[PX];
#pragma unroll
        for (int x = 0; x < PX; ++x) {
            const int n = (it - b * items_per_img) * (64 * PX) + x * 64 + lane;
            live[x] = n < UV;
            hh[x] = fast_div(live[x] ? n : 0, m_row, V);
            ww[x] = (live[x] ? n : 0) - hh[x] * V;
            const bool r0ok = live[x] && hh[x] < Ho, r1ok = live[x] && hh[x] >= 1, c0ok = ww[x] < Wo, c1ok = ww[x] >= 1;
            const unsigned o = (unsigned)(hh[x] * Wo + ww[x]) * 4u;
            v00[x] = (r0ok && c0ok) ? o : kBufOOB;
            v01[x] = (r0ok && c1ok) ? o - 4u : kBufOOB;
            v10[x] = (r1ok && c0ok) ? o - (unsigned)Wo * 4u : kBufOOB;
            v11[x] = (r1ok && c1ok) ? o - (unsigned)Wo * 4u - 4u : kBufOOB;
        }
        v2f A[PX][4][HP];
#pragma unroll
        for (int x = 0; x < PX; ++x)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < HP; ++j) A[x][c][j] = v2f{0.f, 0.f};
#pragma unroll 1
        for (int cg = 0; cg < CO; cg += CB) {
            float dv[CB][PX][4];
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int so = soff + (cg + u) * plane * 4;
#pragma unroll
                for (int x = 0; x < PX; ++x) {
                    dv[u][x][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)v00[x], so, 0));
                    dv[u][x][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)v01[x], so, 0));
                    dv[u][x][2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)v10[x], so, 0));
                    dv[u][x][3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)v11[x], so, 0));
                }
            }
            // the CB*9 (channel, group) steps of this batch as one software pipeline: the broadcast LDS reads of step s+1
            // are issued before the FMAs of step s (hipcc alone keeps only two reads in flight and waits on each)
            {
                constexpr int gcls[9] = {0, 1, 2, 3, 0, 2, 0, 1, 0}, gtap[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
                const v2f* q = (const v2f*)wlds + cg * 9 * HP;  // wave-uniform LDS address; [channel][group][HP] is contiguous
                // pipeline granule = QP weight pairs (a quarter / half group): 2*QP staging registers keep the kernel at
                // 128 VGPRs = 4 waves per SIMD, which lets all B*U*V/64 items of the reference layer run in one round
                constexpr int QP = HP >= 8 ? 4 : HP, NQ = HP / QP;
                v2f wcur[QP], wnxt[QP];
#pragma unroll
                for (int j = 0; j < QP; ++j) wcur[j] = q[j];
#pragma unroll
                for (int st = 0; st < CB * 9 * NQ; ++st) {
                    const int u = st / (9 * NQ), g = (st / NQ) % 9, jb = (st % NQ) * QP;
                    if (st + 1 < CB * 9 * NQ) {
#pragma unroll
                        for (int j = 0; j < QP; ++j) wnxt[j] = q[(st + 1) * QP + j];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < QP; ++j)
#pragma unroll
                        for (int x = 0; x < PX; ++x) {
                            const float d = dv[u][x][gtap[g]];
                            A[x][gcls[g]][jb + j] = __builtin_elementwise_fma(wcur[j], v2f{d, d}, A[x][gcls[g]][jb + j]);
                        }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < QP; ++j) wcur[j] = wnxt[j];
                }
            }
        }
        const bool pair = ((W & 1) == 0);
#pragma unroll
        for (int x = 0; x < PX; ++x) {
            if (!live[x]) continue;
            float* dxb = dx + (size_t)b * CI * H * W;
            const bool two = pair || ww[x] * 2 + 1 < W;
            // fused ReLU::backward of the layer in front (relu.cpp:38): the mask values of one channel pair (4 row
            // segments) are fetched together, then applied and stored -- a load per store would serialise 2*CI round trips
#pragma unroll
            for (int j = 0; j < HP; ++j) {
                float m0[4], m1[4];
                if (relu_below) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int ci = 2 * j + (k >> 1), h = 2 * hh[x] + (k & 1);
                        const float* mrow = relu_below + ((size_t)b * CI * H * W + ((size_t)ci * H + (h < H ? h : 0)) * W + (size_t)ww[x] * 2);
                        m0[k] = mrow[0];
                        m1[k] = two ? mrow[1] : 1.f;
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int half = k >> 1, ph = k & 1;
                    const int ci = 2 * j + half, h = 2 * hh[x] + ph;
                    if (h >= H) continue;
                    float v0 = half ? A[x][ph * 2 + 0][j].y : A[x][ph * 2 + 0][j].x;
                    float v1 = half ? A[x][ph * 2 + 1][j].y : A[x][ph * 2 + 1][j].x;
                    if (relu_below) {
                        v0 = (m0[k] <= 0.f) ? 0.f : v0;
                        v1 = (m1[k] <= 0.f) ? 0.f : v1;
                    }
                    float* row = dxb + ((size_t)ci * H + h) * W + (size_t)ww[x] * 2;
                    if (two) {
                        *(f2u*)row = f2u{v0, v1};
                    } else {
                        row[0] = v0;
                    }
                }
            }
        }
    }
}

// ---- packed-math forward for the same layer (+ optional fused ReLU output) ------------------------------------------
// Same recipe as conv_dgrad_pk: weights re-packed once per call into operand order ([tap][co pair], then the bias pairs),
// x read through a raw buffer descriptor (uniform base + SGPR (image, channel, row) offset + lane offset; tail lanes read
// out of range = 0), 64 consecutive output pixels per wave-item so that every store is one contiguous 256-byte run.
// Sum order per output: ci -> kx -> ky, then + bias (conv2d.cpp:78-87).
__device__ void pack_fwd_3_16_3_2_body(const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ wp,
                                       int tid, int nt) {
    for (int i = tid; i < 27 * 16; i += nt) {
        const int t = i >> 4, co = i & 15;
        wp[i] = w[co * 27 + t];
    }
    if (tid < 16) wp[27 * 16 + tid] = bias[tid];
}

__global__ void pack_fwd_weights_3_16_3_2(const float* __restrict__ w, const float* __restrict__ bias,
                                          float* __restrict__ wp) {
    pack_fwd_3_16_3_2_body(w, bias, wp, threadIdx.x, blockDim.x);
}

// all packing jobs of a network in ONE launch (cnn_conv2d_prepare_filters): blockIdx.y = job
struct PackJob {
    int kind;  // 0 forward 3->16, 1 dgrad 3->16, 2 dgrad stride-2 generic
    const float* w;
    const float* bias;
    float* out;
    int CI, CO;
};
struct PackBatch {
    int n;
    PackJob j[8];
};
__global__ void pack_batch(const PackBatch pb) {
    const PackJob& jb = pb.j[blockIdx.y];
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    if (jb.kind == 0) pack_fwd_3_16_3_2_body(jb.w, jb.bias, jb.out, tid, nt);
    else if (jb.kind == 1) pack_dgrad_3_16_3_2_body(jb.w, jb.out, tid);
    else pack_dgrad_s2_body(jb.w, jb.out, jb.CI, jb.CO, tid, nt);
}

// ---- end of a single-rank train step for THIS layer in one launch: slab reduction -> gradient -> SGD -> re-packed filters -------
// The three things that otherwise follow the first layer's weight-gradient kernel one launch each (slab_reduce, sgd_vec over the
// arena, pack_batch before the next forward pass: ~6 us each, the chip idle in between) for 16 x 28 numbers.  One workgroup, so
// that nothing has to be synchronised across workgroups:
//   1. g[i] = (sum over slots of slabs[slot][i]) / divisor in slab_reduce_batch's order (conv_wgrad.hip: eight slot-lanes per
//      element, each adding its slots in ascending order, then the eight partial sums in order) -- bit-identical to the unfused path;
//   2. gw / gb receive the gradient (they stay readable), w / bias the SGD step (sgd_one: cnn_sgd_update's arithmetic);
//   3. the forward / data-gradient filter images of the updated filters (pack_fwd_3_16_3_2_body / pack_dgrad_3_16_3_2_body), read
//      from the LDS copy of the new values.
constexpr int kFinThreads = 1024, kFinLanes = 8, kFinN = 16 * 28;
__global__ __launch_bounds__(kFinThreads) void first_layer_finish(const float* __restrict__ slabs, int nslots, float divisor,
                                                                  float* __restrict__ gw, float* __restrict__ gb, float* __restrict__ w,
                                                                  float* __restrict__ bias, float lr, float scale, int scaled,
                                                                  float* __restrict__ fwd_img, float* __restrict__ dgrad_img,
                                                                  float* __restrict__ w_keep, float* __restrict__ bias_keep) {
    __shared__ float red[kFinLanes][kFinN];
    __shared__ float wl[16 * 27], bl[16];
    constexpr int PAIRS = kFinLanes * kFinN, PER = (PAIRS + kFinThreads - 1) / kFinThreads;  // (element, slot-lane) pairs per thread
    float acc[PER];
    int idx[PER], sl[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int pr = threadIdx.x + q * kFinThreads;
        acc[q] = 0.f;
        sl[q] = pr / kFinN;
        idx[q] = pr - sl[q] * kFinN;
    }
    const bool last_live = threadIdx.x + (PER - 1) * kFinThreads < PAIRS;
    if (!last_live) idx[PER - 1] = 0, sl[PER - 1] = 0;  // (loads something valid, stores nothing)
    // branch-free batches: every load of kBatch slot rounds is issued before the first add (the adds keep their order)
    constexpr int kBatch = 8;
    for (int s0 = 0; s0 < nslots; s0 += kBatch * kFinLanes) {
        float v[kBatch][PER];
#pragma unroll
        for (int u = 0; u < kBatch; ++u)
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int s = s0 + u * kFinLanes + sl[q];
                v[u][q] = slabs[(size_t)(s < nslots ? s : 0) * kFinN + idx[q]];
            }
#pragma unroll
        for (int u = 0; u < kBatch; ++u)
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const float sum = acc[q] + v[u][q];
                acc[q] = (s0 + u * kFinLanes + sl[q] < nslots) ? sum : acc[q];
            }
    }
#pragma unroll
    for (int q = 0; q < PER; ++q)
        if (q + 1 < PER || last_live) red[sl[q]][idx[q]] = acc[q];
    __syncthreads();
    if (threadIdx.x < kFinN) {
        const int i = threadIdx.x;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < kFinLanes; ++k) t += red[k][i];
        const float g = t / divisor;
        const int row = i / 28, col = i - row * 28;
        if (col < 27) {
            const int j = row * 27 + col;
            gw[j] = g;
            const float old = w[j];
            if (w_keep) w_keep[j] = old;  // (the filters the last forward pass used: Conv2D::get_output() re-materialisation)
            const float v = sgd_one(old, g, lr, scale, scaled != 0);
            w[j] = v;
            wl[j] = v;
        } else {
            if (gb) gb[row] = g;
            const float oldb = bias ? bias[row] : 0.f;
            if (bias && bias_keep) bias_keep[row] = oldb;
            const float v = bias ? sgd_one(oldb, g, lr, scale, scaled != 0) : 0.f;
            if (bias) bias[row] = v;
            bl[row] = v;
        }
    }
    __syncthreads();
    if (fwd_img) pack_fwd_3_16_3_2_body(wl, bl, fwd_img, threadIdx.x, kFinThreads);
    if (dgrad_img) pack_dgrad_3_16_3_2_body(wl, dgrad_img, threadIdx.x);
}

// ---- the same data gradient with the pooled-domain operands staged through LDS (round 3) ----------------------------------------
// conv_dgrad_pool_pk above is bound by its LOAD INSTRUCTIONS, not by bytes: every lane fetches the four windows around its 4x4 block
// of dx itself -- 4 x (dpool, mask) x 16 channels = 128 four-byte loads per lane, and every window is fetched by four different
// lanes (253 MB algorithmic, 85 us = 0.37 of the HBM peak, fetch at 1.0x because the duplicates hit L1/L2 -- but they all pass
// through the texture-address path).  Here a workgroup item = kDR block rows x <= kDSeg block columns of one image: its
// (kDR + 1) x 64 window tile of every channel is fetched ONCE per workgroup with coalesced row loads (raw buffer loads: a window
// outside the pooled plane, or a slot this item does not need, is an out-of-range offset = 0 -- no pad handling), parked in LDS as
// [array][channel][window row][slot], and the lanes pick their 2 x 2 windows with two ds_read2_b32 per array and channel
// (consecutive lanes -> consecutive banks).  40 row loads per wave and item instead of 128 per lane; each window row is fetched by
// at most two items.  Wave w owns block row bh0 + w, lane l block column bw0 + l; arithmetic, FMA order and stores are those of
// conv_dgrad_pool_pk: dx is bit-identical.
constexpr int kDR = 4;     // block rows per item = waves per workgroup
constexpr int kDSeg = 62;  // block columns per item (slots 0 .. 62 of a 64-slot row = window columns bw0 - 1 .. bw0 + 61)
// MODE: 0 = (dpool, int32 mask) | 1 = + pooled (the ReLU mask from the tensor) | 2 = (dpool, packed one-byte mask: include/cnn_amd.h)
template <int MODE>
__global__ __launch_bounds__(kBlock) void conv_dgrad_pool_lds_3_16_3_2(const float* __restrict__ dpool, const int32_t* __restrict__ pmask,
                                                                       const float* __restrict__ pooled, const v2f* __restrict__ wp,
                                                                       float* __restrict__ dx, int B, int H, int W, int Ho, int Wo,
                                                                       int ngroups, int nsegs, unsigned m_gs, unsigned m_seg) {
    constexpr bool RM = MODE == 1, PK8 = MODE == 2;
    constexpr int CO = 16, CI = 3, NA = RM ? 3 : 2, ROWS = kDR + 1;
    static_assert(kDR == kWaves, "one block row per wave");
    __shared__ float tile[NA][CO][ROWS][64];
    const int U = (H + 1) / 2, V = (W + 1) / 2, U2 = (U + 1) / 2, V2 = (V + 1) / 2;
    const int PHo = Ho / 2, PWo = Wo / 2, pplane = PHo * PWo, plane = Ho * Wo;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pbytes = (int)((unsigned)B * CO * pplane * 4u);
    const int pitch8 = pool_mask_pitch(PWo), plane8 = PHo * pitch8;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dpool, 0, pbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)pmask, 0, PK8 ? (int)((unsigned)B * CO * plane8) : pbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(RM ? pooled : dpool), 0, pbytes, 0x00020000);
    const int item = (int)xcd_swizzle(blockIdx.x, gridDim.x);
    const int gs = ngroups * nsegs;
    const int b = fast_div(item, m_gs, gs);
    const int rem = item - b * gs;
    const int grp = fast_div(rem, m_seg, nsegs), sg = rem - grp * nsegs;
    const int bh0 = grp * kDR, bw0 = sg * kDSeg;
    // ---- stage: (channel, tile row) pairs dealt to the waves; slot l of a row = window column bw0 - 1 + l ----
    {
        const int wc = bw0 - 1 + lane;
        const bool cok = wc >= 0 && wc < PWo && lane <= kDSeg;
        constexpr int PER = CO * ROWS / kWaves;  // 20
        float vd[PER], vm[PER], vp[RM ? PER : 1];
#pragma unroll
        for (int jj = 0; jj < PER; ++jj) {
            const int j = jj * kWaves + wave;
            const int ch = j / ROWS, r = j - ch * ROWS;
            const int wr = bh0 - 1 + r;
            const unsigned off = (cok && wr >= 0 && wr < PHo) ? (unsigned)(wr * PWo + wc) * 4u : kBufOOB;
            const int so = (b * CO + ch) * pplane * 4;
            vd[jj] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, (int)off, so, 0));
            if constexpr (PK8) {  // (a window outside the plane reads 0 = "maximum at (0,0)" with a delta of 0: contributes nothing)
                const unsigned off8 = (off != kBufOOB) ? (unsigned)(wr * pitch8 + wc) : kBufOOB;
                vm[jj] = __builtin_bit_cast(float, (int)__builtin_amdgcn_raw_buffer_load_b8(rm, (int)off8, (b * CO + ch) * plane8, 0));
            } else
                vm[jj] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, (int)off, so, 0));
            if constexpr (RM) vp[jj] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, (int)off, so, 0));
        }
#pragma unroll
        for (int jj = 0; jj < PER; ++jj) {
            const int j = jj * kWaves + wave;
            const int ch = j / ROWS, r = j - ch * ROWS;
            tile[0][ch][r][lane] = vd[jj];
            tile[1][ch][r][lane] = vm[jj];
            if constexpr (RM) tile[2][ch][r][lane] = vp[jj];
        }
    }
    __syncthreads();
    const int bh = bh0 + wave, bw = bw0 + lane;
    if (bh >= U2) return;  // (no barrier below)
    const bool live = lane < kDSeg && bw < V2;
    const int e00 = (2 * bh - 1) * Wo + (2 * bw - 1);  // flat index (within a channel) of D[0][0]
    v2f acc[4][6];
#pragma unroll
    for (int sb = 0; sb < 4; ++sb)
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[sb][k] = v2f{0.f, 0.f};
    const int l1 = lane < 63 ? lane + 1 : 63;  // (lane 63 is never live)
#pragma unroll 2
    for (int ch = 0; ch < CO; ++ch) {
        // windows (bh-1 | bh) x (bw-1 | bw) = tile rows wave, wave + 1 x slots lane, lane + 1
        float g[4], pl[4];
        int mk[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int r = wave + (w >> 1), sl = (w & 1) ? l1 : lane;
            g[w] = tile[0][ch][r][sl];
            mk[w] = __builtin_bit_cast(int, tile[1][ch][r][sl]);
            if constexpr (RM) pl[w] = tile[2][ch][r][sl];
            else pl[w] = 1.f;
        }
        float D[3][3];
        const int cbase = ch * plane + e00;
#pragma unroll
        for (int w = 0; w < 4; ++w) g[w] = (pl[w] <= 0.f) ? 0.f : g[w];  // relu.cpp:38
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int w = ((i + 1) >> 1) * 2 + ((j + 1) >> 1);
                if constexpr (PK8) D[i][j] = (mk[w] == ((i + 1) & 1) * 2 + ((j + 1) & 1)) ? g[w] : 0.f;  // (the pixel's place inside ITS window)
                else D[i][j] = (mk[w] == cbase + i * Wo + j) ? g[w] : 0.f;  // pool2d.cpp:96-107
            }
        asm volatile("" ::: "memory");  // (keeps the weight s_loads of all channels from being hoisted, see above)
        const v2f* q = wp + ch * 16;
#pragma unroll
        for (int sb = 0; sb < 4; ++sb) {
            const int sr = sb >> 1, sc = sb & 1;
            const v2f a = {D[sr + 1][sc + 1], D[sr + 1][sc + 1]}, bq = {D[sr + 1][sc], D[sr + 1][sc]},
                      c = {D[sr][sc + 1], D[sr][sc + 1]}, e = {D[sr][sc], D[sr][sc]};
            v2f& P0 = acc[sb][0]; v2f& P1 = acc[sb][1]; v2f& P2 = acc[sb][2]; v2f& P3 = acc[sb][3];
            v2f& Q0 = acc[sb][4]; v2f& Q1 = acc[sb][5];
            P0 = __builtin_elementwise_fma(q[0], a, P0);
            P1 = __builtin_elementwise_fma(q[1], a, P1);
            P2 = __builtin_elementwise_fma(q[2], a, P2);
            P3 = __builtin_elementwise_fma(q[3], a, P3);
            Q0 = __builtin_elementwise_fma(q[4], a, Q0);
            Q1 = __builtin_elementwise_fma(q[5], a, Q1);
            P0 = __builtin_elementwise_fma(q[6], bq, P0);
            P2 = __builtin_elementwise_fma(q[7], bq, P2);
            Q0 = __builtin_elementwise_fma(q[8], bq, Q0);
            Q1 = __builtin_elementwise_fma(q[9], bq, Q1);
            P0 = __builtin_elementwise_fma(q[10], c, P0);
            P1 = __builtin_elementwise_fma(q[11], c, P1);
            Q0 = __builtin_elementwise_fma(q[12], c, Q0);
            P0 = __builtin_elementwise_fma(q[13], e, P0);
            Q0 = __builtin_elementwise_fma(q[14], e, Q0);
        }
    }
    if (!live) return;
    // dx[ci][4bh + 2sr + ph][4bw + 2sc + pw]: one row of the 4x4 block = the (pw0, pw1) pairs of sub-blocks sc = 0, 1
    float* dxb = dx + (size_t)b * CI * H * W;
    const bool quad = (W & 3) == 0;
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int sr = rr >> 1, ph = rr & 1, h = 4 * bh + rr;
            float v[4];
#pragma unroll
            for (int sc = 0; sc < 2; ++sc) {
                const v2f* A = acc[sr * 2 + sc];
                if (ci == 2) {
                    v[2 * sc] = ph ? A[5].x : A[4].x;
                    v[2 * sc + 1] = ph ? A[5].y : A[4].y;
                } else {
                    const v2f lo = ph ? A[2] : A[0], hi = ph ? A[3] : A[1];
                    v[2 * sc] = ci ? lo.y : lo.x;
                    v[2 * sc + 1] = ci ? hi.y : hi.x;
                }
            }
            if (h >= H) continue;
            float* row = dxb + ((size_t)ci * H + h) * W + 4 * (size_t)bw;
            if (quad) {
                {  // (dx has no consumer in a train step: streaming stores keep it out of the caches the other kernels live in)
                    typedef float nt4 __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(nt4{v[0], v[1], v[2], v[3]}, (nt4*)row);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (4 * bw + k < W) row[k] = v[k];
            }
        }
}

typedef int v2i __attribute__((ext_vector_type(2)));

template <bool RELU>
__global__ __launch_bounds__(kBlock) void conv_fwd_pk_3_16_3_2(const float* __restrict__ x, const v2f* __restrict__ wp,
                                                               float* __restrict__ y, float* __restrict__ y_relu, int B,
                                                               int H, int W, int Ho, int Wo, int items_per_img,
                                                               unsigned m_ipi, unsigned m_row) {
    constexpr int CI = 3, CO = 16, K = 3;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long items = (long long)B * items_per_img;
    const int plane = Ho * Wo;
    for (int it = xcd_swizzle(blockIdx.x, gridDim.x) * kWaves + wave; it < (int)items; it += gridDim.x * kWaves) {
        const int b = fast_div(it, m_ipi, items_per_img);
        const int n = (it - b * items_per_img) * 64 + lane;
        const bool live = n < plane;
        const int p = fast_div(live ? n : 0, m_row, Wo), q = (live ? n : 0) - p * Wo;
        // x[.][2p + kx][2q .. 2q+2]: ONE 12-byte load per input row (plain global loads, scalar row base + per-lane offset;
        // a live lane's addresses are inside the image, the others re-read element 0 and store nothing)
        const unsigned vo = live ? (unsigned)(2 * p * W + 2 * q) : 0u;  // x[.][2p][2q], in floats
        float patch[CI * K * K];
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const float* rowp = x + (size_t)((b * CI + ci) * H + kx) * W;  // wave-uniform
                const f3u pr = *(const f3u*)(rowp + vo);
                patch[(ci * K + kx) * K + 0] = pr.x;
                patch[(ci * K + kx) * K + 1] = pr.y;
                patch[(ci * K + kx) * K + 2] = pr.z;
            }
        v2f acc[CO / 2];
#pragma unroll
        for (int j = 0; j < CO / 2; ++j) acc[j] = v2f{0.f, 0.f};
#pragma unroll
        for (int t = 0; t < CI * K * K; ++t) {
            asm volatile("" ::: "memory");  // stream the weight s_loads tap by tap (see conv_dgrad_pk)
            const v2f* qw = wp + t * (CO / 2);
            const v2f pv = {patch[t], patch[t]};
#pragma unroll
            for (int j = 0; j < CO / 2; ++j) acc[j] = __builtin_elementwise_fma(qw[j], pv, acc[j]);
        }
        if (!live) continue;
        const v2f* qb = wp + CI * K * K * (CO / 2);
        float* yb = y + (size_t)b * CO * plane + n;
        float* rb = RELU ? y_relu + (size_t)b * CO * plane + n : nullptr;
#pragma unroll
        for (int j = 0; j < CO / 2; ++j) {
            const v2f v = acc[j] + qb[j];
            yb[(size_t)(2 * j) * plane] = v.x;
            yb[(size_t)(2 * j + 1) * plane] = v.y;
            if (RELU) {
                rb[(size_t)(2 * j) * plane] = v.x >= 0.f ? v.x : 0.f;
                rb[(size_t)(2 * j + 1) * plane] = v.y >= 0.f ? v.y : 0.f;
            }
        }
    }
}

// ---- Conv2D -> ReLU -> MaxPool2D(2, 2) of the reference's first block (alexnet.cpp:12-15) in one kernel --------------
// The separate kernels move 558 MB (conv reads x, writes y and relu(y)) + 250 MB (pool reads relu(y), writes pooled and
// mask); nothing downstream reads y or relu(y) again (the backward pass needs pooled + mask only, see
// cnn_maxpool2d_backward_relu), so this kernel writes ONLY pooled and mask: 154 MB in, 100 MB out.
// Lane pairs (2i, 2i+1) own one pooling window: lane j of the pair computes the conv outputs of window column j for both
// window rows (5 input rows x 3 floats x 3 channels = 45 loads for 2 x 27 taps), in exactly the tap order of
// conv_fwd_pk_3_16_3_2 (bit-identical y), applies bias and ReLU, swaps values with its neighbour through DPP and the
// even lane scans the window in the reference's order (pool2d.cpp:67-75: first maximum wins, strict '<').
// PK8: the mask is the packed one-byte form (include/cnn_amd.h, CNN_CONV2D_POOL_MASK_PACKED): rows of `pitch8` bytes.
template <int DBG, bool PK8>  // DBG: tuning ablations: 1 no FMAs, 2 one weight fetch for all taps
__global__ __launch_bounds__(kBlock) void conv_fwd_pool_pk_3_16_3_2(const float* __restrict__ x, const v2f* __restrict__ wp,
                                                                    float* __restrict__ pooled, int32_t* __restrict__ mask,
                                                                    int B, int H, int W, int Ho, int Wo, int PHo, int PWo,
                                                                    int items_per_img, unsigned m_ipi, unsigned m_prow, int pitch8) {
    constexpr int CI = 3, CO = 16, K = 3;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long items = (long long)B * items_per_img;
    const int half = 2 * PHo * PWo;  // window columns per image
    const int out_bytes = (int)((unsigned)B * CO * PHo * PWo * 4u);
    const int mask_bytes = PK8 ? (int)((unsigned)B * CO * PHo * pitch8) : out_bytes;
    const __amdgpu_buffer_rsrc_t rpool = __builtin_amdgcn_make_buffer_rsrc((void*)pooled, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc((void*)(mask ? mask : (int32_t*)pooled), 0, mask ? mask_bytes : out_bytes, 0x00020000);
    for (int it = xcd_swizzle(blockIdx.x, gridDim.x) * kWaves + wave; it < (int)items; it += gridDim.x * kWaves) {
        const int b = fast_div(it, m_ipi, items_per_img);
        const int n2 = (it - b * items_per_img) * 64 + lane;
        const bool live = n2 < half;
        const int pp = (live ? n2 : 0) >> 1, j = n2 & 1;
        const int ph = fast_div(pp, m_prow, PWo), pw = pp - ph * PWo;
        const int q = 2 * pw + j;  // conv column; conv rows 2ph and 2ph + 1
        // x[.][4ph + r][2q .. 2q+2]: ONE 12-byte load per input row (the wave reads one contiguous run, neighbouring lanes
        // overlap by a float) -- 15 instead of 45 load instructions per item (68 -> 59 us with 8 + 4 bytes, -> this).  Plain global loads (scalar row
        // base + per-lane offset): every address of a live lane is inside the image (4ph + 4 <= H - 1, 2q + 2 <= W - 1), a
        // lane that owns no window re-reads element 0 and stores nothing.  (Raw buffer loads of mixed width at one offset
        // are mis-lowered by hipcc, tools/probes/bufvec_probe.cpp.)
        const unsigned vo = live ? (unsigned)(4 * ph * W + 2 * q) : 0u;  // x[.][4ph][2q], in floats
        float patch[CI][5][K];
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const float* rowp = x + (size_t)((b * CI + ci) * H + r) * W;  // wave-uniform
                const f3u pr = *(const f3u*)(rowp + vo);
                patch[ci][r][0] = pr.x;
                patch[ci][r][1] = pr.y;
                patch[ci][r][2] = pr.z;
            }
        v2f acc0[CO / 2], acc1[CO / 2];
#pragma unroll
        for (int c = 0; c < CO / 2; ++c) acc0[c] = acc1[c] = v2f{0.f, 0.f};
#pragma unroll
        for (int t = 0; t < CI * K * K; ++t) {
            if (DBG != 2 || t == 0) asm volatile("" ::: "memory");  // stream the weight s_loads tap by tap (see conv_dgrad_pk)
            const int ci = t / 9, kx = (t - ci * 9) / 3, ky = t - ci * 9 - kx * 3;
            const v2f* qw = wp + (DBG == 2 ? 0 : t) * (CO / 2);
            const v2f p0 = {patch[ci][kx][ky], patch[ci][kx][ky]}, p1 = {patch[ci][kx + 2][ky], patch[ci][kx + 2][ky]};
            if (DBG == 1) {
                acc0[t & 7] += p0;
                acc1[t & 7] += p1;
                continue;
            }
#pragma unroll
            for (int c = 0; c < CO / 2; ++c) {
                acc0[c] = __builtin_elementwise_fma(qw[c], p0, acc0[c]);
                acc1[c] = __builtin_elementwise_fma(qw[c], p1, acc1[c]);
            }
        }
        asm volatile("" ::: "memory");
        const v2f* qb = wp + CI * K * K * (CO / 2);
        // (all selects first, ONE branch around the stores: a branch per channel made hipcc keep 16 exec masks in SGPRs
        //  and spill the weight registers through v_writelane -- 3098 lane moves in the first version of this kernel)
        float bestv[CO];
        int offv[CO];
#pragma unroll
        for (int c = 0; c < CO; ++c) {
            const float bsv = (c & 1) ? qb[c / 2].y : qb[c / 2].x;
            float a = ((c & 1) ? acc0[c / 2].y : acc0[c / 2].x) + bsv;  // conv row 2ph
            float d = ((c & 1) ? acc1[c / 2].y : acc1[c / 2].x) + bsv;  // conv row 2ph + 1
            a = a >= 0.f ? a : 0.f;  // relu.cpp:21-26
            d = d >= 0.f ? d : 0.f;
            const float pa = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0xB1, 0xf, 0xf, true));
            const float pd = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, d), 0xB1, 0xf, 0xf, true));
            float best = a;
            int off = 0;
            const bool c1 = best < pa;
            best = c1 ? pa : best; off = c1 ? 1 : off;
            const bool c2 = best < d;
            best = c2 ? d : best; off = c2 ? Wo : off;
            const bool c3 = best < pd;
            best = c3 ? pd : best; off = c3 ? Wo + 1 : off;
            bestv[c] = best;
            offv[c] = off;
        }
        // raw buffer stores: one per-lane byte offset + a scalar channel offset (64-bit per-channel pointers in SGPR pairs
        // were the other half of the spills); lanes that own no window store out of range = nowhere
        const unsigned so_lane = (live && j == 0) ? (unsigned)(ph * PWo + pw) * 4u : kBufOOB;
        const int mbase = 2 * ph * Wo + 2 * pw;
        const int PP4 = PHo * PWo * 4;
        const int soff = b * CO * PP4;
#pragma unroll
        for (int c = 0; c < CO; ++c) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, bestv[c]), rpool, (int)so_lane, soff + c * PP4, 0);
        if (mask && !PK8) {
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                // bit 31 = "this window's pooled value is <= 0": the block's ReLU::backward (relu.cpp:37) blocks its delta.  The
                // pooled-domain gradient kernels compare the mask with the flat index they expect, so a marked window contributes
                // nothing there -- the ReLU mask costs no extra tensor read anywhere in the backward pass.
                const int marked = (c * Ho * Wo + mbase + offv[c]) | ((bestv[c] <= 0.f) ? (int)0x80000000 : 0);
                __builtin_amdgcn_raw_buffer_store_b32(marked, rmask, (int)so_lane, soff + c * PP4, 0);
            }
        }
        if (mask && PK8) {
            // one byte per window: 2 * (row of the maximum) + column, bit 7 = the mark above (offv is 0 | 1 | Wo | Wo + 1)
            const unsigned so8 = (live && j == 0) ? (unsigned)(ph * pitch8 + pw) : kBufOOB;
            const int PP1 = PHo * pitch8;
            const int soff8 = b * CO * PP1;
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                const int code = (offv[c] >= Wo ? 2 + (offv[c] - Wo) : offv[c]) | ((bestv[c] <= 0.f) ? 0x80 : 0);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)code, rmask, (int)so8, soff8 + c * PP1, 0);
            }
        }
    }
}

// packed one-byte pool mask -> the int32 flat index of cnn_maxpool2d_forward (bit 7 -> bit 31), for readers outside the fused block
__global__ __launch_bounds__(kBlock) void pool_mask_unpack_kernel(const unsigned char* __restrict__ packed, int32_t* __restrict__ mask, long long n,
                                                                  int Co, int Ho, int Wo, int pitch8) {
    const int PHo = Ho / 2, PWo = Wo / 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int pw = (int)(i % PWo);
        const long long r = i / PWo;  // (b * Co + co) * PHo + ph
        const int ph = (int)(r % PHo), co = (int)((r / PHo) % Co);
        const int code = packed[r * pitch8 + pw];
        mask[i] = (co * Ho * Wo + (2 * ph + ((code >> 1) & 1)) * Wo + 2 * pw + (code & 1)) | ((code & 0x80) ? (int)0x80000000 : 0);
    }
}

// ---- packed-math weight + bias gradient for the same layer ----------------------------------------------------------
// gw[co][t] = sum_pixels dy[co][pixel] * patch[t][pixel], t = (ci,kx,ky); gb[co] = sum_pixels dy[co][pixel]
// (conv2d.cpp:117-159 without the 1/B, which the slab reduction applies).  K = 27 is far too thin for MFMA tiles, so
// this is a VALU kernel: a workgroup walks 64-pixel items; per item the 27 patch rows and 16 dy rows (64 floats each) are
// fetched ONCE by the four waves together (raw buffer loads, tail pixels read 0) into a double-buffered LDS stage while
// the previous item is being consumed; wave w owns output channels 4w..4w+3 and keeps its 4 x (27+1) running sums in
// registers as 14 float pairs per channel (v_pk_fma_f32: patch pair x broadcast dy) + a bias sum.  At the end every wave
// reduces its sums across lanes and writes one slab row set; reduce_slabs() adds the slabs in a fixed order.
constexpr int kWgX = 7;       // patch rows each wave stages per item (4 x 7 = 28 >= 27 taps; the 28th is the zero pad)
// POOLED: the delta of the convolution output is not in memory; it is rebuilt on the fly from the pooled domain as
//     dy[co][n] = (mask[window] == co*Ho*Wo + n  &&  !(pooled[window] <= 0)) ? dpool[window] : 0,   window = (p/2, q/2)
// = MaxPool2D::backward (pool2d.cpp:96-107) followed by ReLU::backward (relu.cpp:35-40): three 4-byte loads per lane and
// channel instead of one, but the 202 MB delta tensor is neither written nor read.  `dy` then holds dpool.
template <int kWgDepth, int POOLED>  // items in flight per workgroup; POOLED: 0 dy | 1 pooled domain | 2 ... with dpool pre-masked
__global__ __launch_bounds__(kBlock) void conv_wgrad_pk_3_16_3_2(const float* __restrict__ x, const float* __restrict__ dy,
                                                                 const int32_t* __restrict__ pmask, const float* __restrict__ pooled,
                                                                 float* __restrict__ slabs, int B, int H, int W, int Ho,
                                                                 int Wo, int items_per_img, unsigned m_ipi, unsigned m_row) {
    constexpr int CI = 3, CO = 16, NP = 14;  // NP float pairs cover the 27 taps (+1 zero pad)
    __shared__ float stage[2][4 * kWgX][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long items = (long long)B * items_per_img;
    const int plane = Ho * Wo;
    const int PHo = Ho / 2, PWo = Wo / 2, pplane = PHo * PWo;
    const int dy_bytes = (int)((unsigned)B * CO * (POOLED ? pplane : plane) * 4u);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)(POOLED ? (const void*)pmask : (const void*)dy), 0, dy_bytes, 0x00020000);
    constexpr bool relu_mask = POOLED == 1;  // (2: dpool already carries the ReLU mask, `pooled` is not read)
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(relu_mask ? pooled : dy), 0, dy_bytes, 0x00020000);
    // wave w stages taps 7w .. 7w+6 of x (tap 27 does not exist: it reads out of range = the zero pad) and loads the dy
    // rows of ITS output channels 4w .. 4w+3, which never go through LDS.  Everything below is branch-free on purpose
    // (an item past the end reads out of range = 0 and adds nothing): with straight-line code hipcc counts the outstanding
    // loads (s_waitcnt vmcnt(N > 0)) instead of draining the queue at every control-flow merge.
    // x: wave w fetches the input rows (ci,kx) = 3w .. 3w+2 with ONE 12-byte load each (taps ky = 0..2; plain global loads,
    // scalar row base + per-lane offset: a live lane's addresses are inside the image, the others re-read element 0 and
    // meet dy = 0).  9 rows for 4 waves: wave 3 repeats row 8 (same values into the same LDS slots) so that the code stays
    // free of wave-dependent branches.  12 instead of 27 input load instructions per item.
    int row_id[3], row_off[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int r = wave * 3 + i < 9 ? wave * 3 + i : 8;
        row_id[i] = r;
        row_off[i] = ((r / 3) * H + (r % 3)) * W;
    }
    if (threadIdx.x < 64) stage[0][27][threadIdx.x] = stage[1][27][threadIdx.x] = 0.f;  // tap 27 = the zero pad of the 14th pair
    // each workgroup walks a CONTIGUOUS range of items: vertically neighbouring items share two of their three input
    // rows, which then come from this CU's L1 / this XCD's L2 instead of HBM (a strided assignment fetched x 2.5x)
    const long long per = (items + gridDim.x - 1) / gridDim.x;
    const long long it_end = (blockIdx.x + 1) * per < items ? (blockIdx.x + 1) * per : items;
    constexpr int ND = POOLED ? 13 : 4;  // POOLED: dpool[4] | mask[4] | pooled[4] | expected flat index of this lane's pixel
    f3u lx[kWgDepth][3];
    float ldy[kWgDepth][ND];
    auto issue = [&](long long it, f3u(&ax)[3], float(&ad)[ND]) {
        const bool inr = it < it_end;
        const int iti = inr ? (int)it : 0;
        const int b = fast_div(iti, m_ipi, items_per_img);
        const int n = (iti - b * items_per_img) * 64 + lane;
        const bool live = inr && n < plane;
        const int p = fast_div(live ? n : 0, m_row, Wo), q = (live ? n : 0) - p * Wo;
        const unsigned vx = live ? (unsigned)(2 * p * W + 2 * q) : 0u;  // x[.][2p][2q], in floats
        const float* xb = x + (size_t)b * CI * H * W;
#pragma unroll
        for (int i = 0; i < 3; ++i) ax[i] = *(const f3u*)(xb + row_off[i] + vx);
        if constexpr (POOLED) {
            const bool win = live && (p >> 1) < PHo && (q >> 1) < PWo;  // pixels outside every window have no delta
            const unsigned vd = win ? (unsigned)((p >> 1) * PWo + (q >> 1)) * 4u : kBufOOB;
            const int sd = (b * CO + 4 * wave) * pplane * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                ad[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, (int)vd, sd + c * pplane * 4, 0));
                ad[4 + c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, (int)vd, sd + c * pplane * 4, 0));
                if constexpr (relu_mask) ad[8 + c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, (int)vd, sd + c * pplane * 4, 0));
                else ad[8 + c] = 1.f;
            }
            ad[12] = __builtin_bit_cast(float, 4 * wave * plane + n);
        } else {
            const unsigned vd = live ? (unsigned)n * 4u : kBufOOB;
            const int sd = (b * CO + 4 * wave) * plane * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                ad[c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, (int)vd, sd + c * plane * 4, 0));
        }
    };

    v2f A[4][NP];
    float bs[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bs[c] = 0.f;
#pragma unroll
        for (int j = 0; j < NP; ++j) A[c][j] = v2f{0.f, 0.f};
    }
    const long long stride = 1;
    long long it = blockIdx.x * per;
#pragma unroll
    for (int dd = 0; dd < kWgDepth; ++dd) issue(it + dd * stride, lx[dd], ldy[dd]);
    int k = 0;
    auto consume = [&](f3u(&ax)[3], float(&ad)[ND]) {  // item `it`: registers -> LDS, refill the register set, accumulate
        float(*st)[64] = stage[k & 1];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            st[row_id[i] * 3 + 0][lane] = ax[i].x;
            st[row_id[i] * 3 + 1][lane] = ax[i].y;
            st[row_id[i] * 3 + 2][lane] = ax[i].z;
        }
        float d[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if constexpr (POOLED) {
                const int want = __builtin_bit_cast(int, ad[12]) + c * plane;
                const bool hit = __builtin_bit_cast(int, ad[4 + c]) == want;  // (a lane without a window loaded zeros: d = 0 anyway)
                d[c] = (hit && !(ad[8 + c] <= 0.f)) ? ad[c] : 0.f;
            } else {
                d[c] = ad[c];
            }
        }
        // LDS-only barrier: __syncthreads() would also drain the global loads that are deliberately still in flight (its
        // fence covers every address space -> s_waitcnt vmcnt(0)).  One barrier per item: the buffer written now was last
        // read two items ago.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        issue(it + kWgDepth * stride, ax, ad);
        v2f pv[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) pv[j] = v2f{st[2 * j][lane], st[2 * j + 1][lane]};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const v2f dd = {d[c], d[c]};
            bs[c] += d[c];
#pragma unroll
            for (int j = 0; j < NP; ++j) A[c][j] = __builtin_elementwise_fma(pv[j], dd, A[c][j]);
        }
        ++k;
        it += stride;
    };
    while (it < it_end) {  // (up to kWgDepth-1 trailing consume() calls see all-zero items)
#pragma unroll
        for (int dd = 0; dd < kWgDepth; ++dd) consume(lx[dd], ldy[dd]);
    }
    // ---- cross-lane reduction (fixed xor tree) and the slab row set of this workgroup ----
    float* out = slabs + (size_t)blockIdx.x * CO * 28 + (size_t)(4 * wave) * 28;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            float a = A[c][j].x, b2 = A[c][j].y;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                a += __shfl_xor(a, o, 64);
                b2 += __shfl_xor(b2, o, 64);
            }
            if (lane == 0) {
                out[c * 28 + 2 * j] = a;
                if (2 * j + 1 < 27) out[c * 28 + 2 * j + 1] = b2;
            }
        }
        float t = bs[c];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) t += __shfl_xor(t, o, 64);
        if (lane == 0) out[c * 28 + 27] = t;
    }
}

inline unsigned wave_grid(long long rows) {
    long long need = (rows + kWaves - 1) / kWaves;
    const long long cap = (long long)num_cus() * 32;
    return (unsigned)(need < 1 ? 1 : (need > cap ? cap : need));
}

}  // namespace

namespace cnn_amd {
bool dgrad_rd_supported(const cnn_conv2d_desc* d);  // conv_dgrad_rd.hip (takes precedence over the packed stride-2 kernel)
// conv_wgrad_win.hip: the window-major MFMA weight gradient of this layer (default; CNN_AMD_WG_WIN=0: the packed VALU kernel below)
int win_wgrad_slots(const cnn_conv2d_desc* d);
int win_wgrad_launch(const cnn_conv2d_desc* d, const float* x, const float* dy, const int32_t* mask, const float* pooled, float* slabs,
                     hipStream_t s);

#define CONV_TAG(d) "B%d Ci%d %dx%d Co%d k%d s%d p%d", (d)->B, (d)->Ci, (d)->H, (d)->W, (d)->Co, (d)->k, (d)->s, (d)->pad

// returns 1 when the geometry has a direct kernel (and it was launched), 0 when the caller must use the implicit GEMM,
// < 0 on error
bool direct_conv_supported(const cnn_conv2d_desc* d) {
    return d->Ci == 3 && d->Co == 16 && d->k == 3 && d->s == 2 && d->pad == 0 && !CNN_OPT_SET("NO_DIRECT");
}

// prepared: `ws` already holds the packed filters (cnn_conv2d_prepare_filters); w / bias are then unused
bool direct_fwd_pk_ok(const cnn_conv2d_desc* d) {
    return (long long)d->B * 3 * d->H * d->W * 4 < (1ll << 31) - 16 && !CNN_OPT_SET("FWD_NOPK");
}
int direct_conv_forward(const cnn_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y,
                        float* y_relu, void* ws, size_t ws_bytes, hipStream_t s, bool prepared) {
    const int Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, 0), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, 0);
    // measured: the packed kernel wins when the ReLU output is fused in (121 vs 142 us), the plain row kernel otherwise
    if ((y_relu != nullptr || prepared) && ws != nullptr && ws_bytes >= 28 * 16 * sizeof(float) && direct_fwd_pk_ok(d)) {
        if (!prepared)
            CNN_KLAUNCH(s, "pack_fwd_weights", (pack_fwd_weights_3_16_3_2<<<1, 256, 0, s>>>(w, bias, (float*)ws)), CONV_TAG(d));
        const int ipi = (Ho * Wo + 63) / 64;
        const long long witems = (long long)d->B * ipi;
        if (y_relu)
            CNN_KLAUNCH(s, "conv_fwd_pk<3,16,3,2>+relu",
                        (conv_fwd_pk_3_16_3_2<true><<<wave_grid(witems), kBlock, 0, s>>>(x, (const v2f*)ws, y, y_relu, d->B, d->H,
                                                                                        d->W, Ho, Wo, ipi, div_magic(ipi), div_magic(Wo))), CONV_TAG(d));
        else
            CNN_KLAUNCH(s, "conv_fwd_pk<3,16,3,2>",
                        (conv_fwd_pk_3_16_3_2<false><<<wave_grid(witems), kBlock, 0, s>>>(x, (const v2f*)ws, y, nullptr, d->B, d->H,
                                                                                         d->W, Ho, Wo, ipi, div_magic(ipi), div_magic(Wo))), CONV_TAG(d));
        return CNN_AMD_OK;
    }
    CNN_REQUIRE(!prepared, "internal: prepared forward without a packed kernel");
    const long long rows = (long long)d->B * Ho;
    CNN_KLAUNCH(s, y_relu ? "conv_direct_fwd<3,16,3,2>+relu" : "conv_direct_fwd<3,16,3,2>",
                (conv_direct_fwd<3, 16, 3, 2><<<wave_grid(rows), kBlock, 0, s>>>(x, w, bias, y, y_relu, d->B, d->H, d->W, Ho, Wo)),
                CONV_TAG(d));
    return CNN_AMD_OK;
}

// Conv2D -> ReLU -> MaxPool2D(2,2) fused (see conv_fwd_pool_pk_3_16_3_2)
bool direct_conv_pool_supported(const cnn_conv2d_desc* d) {
    const int Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, 0), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, 0);
    return direct_conv_supported(d) && direct_fwd_pk_ok(d) && Ho >= 2 && Wo >= 2 && (long long)16 * Ho * Wo < (1ll << 31) &&
           (long long)d->B * 16 * (Ho / 2) * (Wo / 2) * 4 < (1ll << 31) - 16 && !CNN_OPT_SET("NO_POOL_FUSION");
}
// the packed one-byte pool mask (include/cnn_amd.h): written by conv_fwd_pool_pk<.., true>, read by the LDS-staged data gradient and
// the window kernel only -- switches that select the older kernels of the block turn it off
bool direct_dgrad_pk_ok(const cnn_conv2d_desc* d);
bool direct_pool_mask_packed_ok(const cnn_conv2d_desc* d) {
    if (!direct_conv_pool_supported(d) || win_wgrad_slots(d) <= 0 || CNN_OPT_INT("DGRAD_POOL_LDS", 1) == 0) return false;
    if (CNN_OPT_INT("WG_POOL_RD", 0) != 0 || CNN_OPT_INT("POOL_MASK_PACKED", 1) == 0 || !direct_dgrad_pk_ok(d)) return false;
    const int U = (d->H + 1) / 2, V = (d->W + 1) / 2, U2 = (U + 1) / 2, V2 = (V + 1) / 2;
    return (long long)d->B * ((U2 + kDR - 1) / kDR) * ((V2 + kDSeg - 1) / kDSeg) < (1ll << 31);
}
size_t direct_pool_mask_bytes(const cnn_conv2d_desc* d) {
    const int Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, d->pad), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, d->pad);
    if (Ho < 2 || Wo < 2) return 0;
    if (d->flags & CNN_CONV2D_POOL_MASK_PACKED) return (size_t)d->B * d->Co * (Ho / 2) * pool_mask_pitch(Wo / 2) + 64;
    return (size_t)d->B * d->Co * (Ho / 2) * (Wo / 2) * sizeof(int32_t);
}
int direct_pool_mask_unpack(const cnn_conv2d_desc* d, const void* packed, int32_t* mask, hipStream_t s) {
    const int Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, d->pad), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, d->pad);
    CNN_REQUIRE(Ho >= 2 && Wo >= 2 && (long long)d->Co * Ho * Wo < (1ll << 31), "cnn_conv2d_pool_mask_unpack: geometry not covered");
    const long long n = (long long)d->B * d->Co * (Ho / 2) * (Wo / 2);
    CNN_KLAUNCH(s, "pool_mask_unpack", (pool_mask_unpack_kernel<<<wave_grid((n + 63) / 64), kBlock, 0, s>>>((const unsigned char*)packed, mask, n, d->Co,
                                                                                                        Ho, Wo, pool_mask_pitch(Wo / 2))), CONV_TAG(d));
    return CNN_AMD_OK;
}
int direct_conv_pool_forward(const cnn_conv2d_desc* d, const float* x, const float* w, const float* bias, float* pooled,
                             int32_t* mask, void* ws, size_t ws_bytes, hipStream_t s, bool prepared) {
    CNN_REQUIRE(direct_conv_pool_supported(d), "cnn_conv2d_relu_maxpool2_forward: geometry not covered (3->16 channels, 3x3 stride 2)");
    CNN_REQUIRE(ws != nullptr && ws_bytes >= 28 * 16 * sizeof(float), "cnn_conv2d_relu_maxpool2_forward: workspace too small");
    const int Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, 0), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, 0);
    const int PHo = Ho / 2, PWo = Wo / 2;
    if (!prepared)
        CNN_KLAUNCH(s, "pack_fwd_weights", (pack_fwd_weights_3_16_3_2<<<1, 256, 0, s>>>(w, bias, (float*)ws)), CONV_TAG(d));
    const int ipi = (2 * PHo * PWo + 63) / 64;
    const long long witems = (long long)d->B * ipi;
    const int dbg = CNN_MEASURE_INT("DBG", 0);
    const bool pk8 = (d->flags & CNN_CONV2D_POOL_MASK_PACKED) != 0;
    CNN_REQUIRE(!pk8 || direct_pool_mask_packed_ok(d), "cnn_conv2d_relu_maxpool2_forward: packed pool mask not available (cnn_conv2d_pool_mask_packed_supported)");
    const int pitch8 = pool_mask_pitch(PWo);
#define FP_LAUNCH(DBG_, PK8_)                                                                                                  \
    CNN_KLAUNCH(s, PK8_ ? "conv_fwd_pool_pk<3,16,3,2>+m8" : "conv_fwd_pool_pk<3,16,3,2>",                                      \
                (conv_fwd_pool_pk_3_16_3_2<DBG_, PK8_><<<wave_grid(witems), kBlock, 0, s>>>(x, (const v2f*)ws, pooled, mask, d->B, d->H, d->W, Ho, \
                                                                                     Wo, PHo, PWo, ipi, div_magic(ipi), div_magic(PWo), pitch8)),  \
                CONV_TAG(d))
    if (pk8) FP_LAUNCH(0, true);
    else if (dbg == 1) FP_LAUNCH(1, false);
    else if (dbg == 2) FP_LAUNCH(2, false);
    else FP_LAUNCH(0, false);
#undef FP_LAUNCH
    return CNN_AMD_OK;
}

bool direct_dgrad_pk_ok(const cnn_conv2d_desc* d) {
    const int Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, 0), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, 0);
    return (long long)d->B * 16 * Ho * Wo * 4 < (1ll << 31) - 16 && !CNN_OPT_SET("DG_NOPK");
}
int direct_conv_dgrad(const cnn_conv2d_desc* d, const float* dy, const float* w, float* dx, void* ws, size_t ws_bytes,
                      hipStream_t s, bool prepared) {
    const int Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, 0), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, 0);
    if (ws != nullptr && ws_bytes >= 16 * 32 * sizeof(float) && direct_dgrad_pk_ok(d)) {
        if (!prepared)
            CNN_KLAUNCH(s, "pack_dgrad_weights", (pack_dgrad_weights_3_16_3_2<<<1, 64, 0, s>>>(w, (float*)ws)), CONV_TAG(d));
        const int V = (d->W + 1) / 2, UVg = ((d->H + 1) / 2) * V;
        const int ipi = (UVg + 63) / 64;
        const long long witems = (long long)d->B * ipi;
        // CNN_AMD_DBG = 4 / 8 / 12 (tuning only): compile-time ablations without the FMAs / the stores / both, the source
        // of the breakdown in profiles/NOTEBOOK.md section 6
        const int dbg = CNN_MEASURE_INT("DBG", 0);
#define PK_LAUNCH(DBG_)                                                                                              \
    CNN_KLAUNCH(s, "conv_dgrad_pk<3,16,3,2>",                                                                       \
                (conv_dgrad_pk_3_16_3_2<4, DBG_><<<wave_grid(witems), kBlock, 0, s>>>(dy, (const v2f*)ws, dx, d->B, d->H, d->W, Ho, \
                                                                                     Wo, ipi, div_magic(ipi), div_magic(V))), \
                CONV_TAG(d))
        if (dbg == 4) PK_LAUNCH(4);
        else if (dbg == 8) PK_LAUNCH(8);
        else if (dbg == 12) PK_LAUNCH(12);
        else PK_LAUNCH(0);  // 4 dy channels per load batch (2: 91.5 us, 4: 91.0, 8: 92.5, 16: SGPR spills)
#undef PK_LAUNCH
        return CNN_AMD_OK;
    }
    CNN_REQUIRE(!prepared, "internal: prepared dgrad without a packed kernel");
    // fallback (dy tensor >= 2 GiB or no workspace): the plain row kernel
    const long long rows = (long long)d->B * ((d->H + d->s - 1) / d->s);
    CNN_KLAUNCH(s, "conv_direct_dgrad<3,16,3,2>",
                (conv_direct_dgrad<3, 16, 3, 2, 4><<<wave_grid(rows), kBlock, 0, s>>>(dy, w, dx, d->B, d->H, d->W, Ho, Wo, 0)),
                CONV_TAG(d));
    return CNN_AMD_OK;
}


// conv_layer_1's data gradient from the pooled domain (see conv_dgrad_pool_pk_3_16_3_2)
int direct_conv_dgrad_pooled(const cnn_conv2d_desc* d, const float* dpool, const int32_t* mask, const float* pooled, const float* w,
                             float* dx, void* ws, size_t ws_bytes, hipStream_t s, bool prepared) {
    CNN_REQUIRE(direct_conv_pool_supported(d) && direct_dgrad_pk_ok(d), "cnn_conv2d_backward_data_pooled2: geometry not covered");
    CNN_REQUIRE(ws != nullptr && ws_bytes >= 16 * 32 * sizeof(float), "cnn_conv2d_backward_data_pooled2: workspace too small");
    const int Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, 0), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, 0);
    if (!prepared)
        CNN_KLAUNCH(s, "pack_dgrad_weights", (pack_dgrad_weights_3_16_3_2<<<1, 64, 0, s>>>(w, (float*)ws)), CONV_TAG(d));
    const int U = (d->H + 1) / 2, V = (d->W + 1) / 2, U2 = (U + 1) / 2, V2 = (V + 1) / 2;
    const int ipi = (U2 * V2 + 63) / 64;
    const long long witems = (long long)d->B * ipi;
    // CNN_AMD_DX0_GRID=n: at most n workgroups (grid-stride over the rest) -- this HBM-bound kernel usually runs BESIDE the
    // latency-bound head of the net (deferred launch, cnn_amd/pynet.py); a smaller grid leaves wave slots and memory queues
    // to those kernels
    unsigned grid = wave_grid(witems);
    if (const OptVal e = CNN_OPT_VAL("DX0_GRID"))
        if (atoi(e) > 0 && (unsigned)atoi(e) < grid) grid = (unsigned)atoi(e);
    // LDS-staged variant (default; DGRAD_POOL_LDS=0: the per-lane-load kernel): one item per workgroup
    const int ngroups = (U2 + kDR - 1) / kDR, nsegs = (V2 + kDSeg - 1) / kDSeg;
    const long long nitems = (long long)d->B * ngroups * nsegs;
    const bool pk8 = (d->flags & CNN_CONV2D_POOL_MASK_PACKED) != 0;
    CNN_REQUIRE(!pk8 || (direct_pool_mask_packed_ok(d) && pooled == nullptr),
                "cnn_conv2d_backward_data_pooled2: packed pool mask not available here (cnn_conv2d_pool_mask_packed_supported; pooled must be NULL)");
    if (CNN_OPT_INT("DGRAD_POOL_LDS", 1) != 0 && nitems < (1ll << 31)) {
#define DL_LAUNCH(MODE_, NAME_)                                                                                                          \
    CNN_KLAUNCH(s, NAME_,                                                                                                                \
                (conv_dgrad_pool_lds_3_16_3_2<MODE_><<<(unsigned)nitems, kBlock, 0, s>>>(dpool, mask, pooled, (const v2f*)ws, dx, d->B, d->H, d->W, Ho, \
                                                                                         Wo, ngroups, nsegs, div_magic(ngroups * nsegs), div_magic(nsegs))), \
                CONV_TAG(d))
        if (pk8) DL_LAUNCH(2, "conv_dgrad_pk<3,16,3,2>+poolm8");
        else if (pooled) DL_LAUNCH(1, "conv_dgrad_pk<3,16,3,2>+pool");
        else DL_LAUNCH(0, "conv_dgrad_pk<3,16,3,2>+poolm");
#undef DL_LAUNCH
        return CNN_AMD_OK;
    }
    if (pooled)
        CNN_KLAUNCH(s, "conv_dgrad_pk<3,16,3,2>+pool",
                    (conv_dgrad_pool_pk_3_16_3_2<2, true><<<grid, kBlock, 0, s>>>(dpool, mask, pooled, (const v2f*)ws, dx, d->B, d->H,
                                                                                              d->W, Ho, Wo, ipi, div_magic(ipi), div_magic(V2))),
                    CONV_TAG(d));
    else
        CNN_KLAUNCH(s, "conv_dgrad_pk<3,16,3,2>+poolm",
                    (conv_dgrad_pool_pk_3_16_3_2<2, false><<<grid, kBlock, 0, s>>>(dpool, mask, nullptr, (const v2f*)ws, dx, d->B, d->H,
                                                                                               d->W, Ho, Wo, ipi, div_magic(ipi), div_magic(V2))),
                    CONV_TAG(d));
    return CNN_AMD_OK;
}

// Packed VALU data gradient for k = 3, stride 2, pad 0 layers with 16 input channels (conv_layer_2 of the reference net).
bool pk_dgrad_s2_supported(const cnn_conv2d_desc* d) {
    const int Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, 0), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, 0);
    const OptVal e = CNN_OPT_VAL("PK_DGRAD");
    if (e && atoi(e) == 0) return false;
    return d->k == 3 && d->s == 2 && d->pad == 0 && (d->Ci == 16 || (e && d->Ci == 32)) && d->Co % 4 == 0 && d->Co * 9 * d->Ci * 4 <= 144 * 1024 &&
           (long long)d->B * d->Co * Ho * Wo * 4 < (1ll << 31) - 16 && (long long)d->B * (((d->H + 1) / 2) * ((d->W + 1) / 2) + 63) / 64 < (1ll << 30);
}
size_t pk_dgrad_s2_workspace_floats(const cnn_conv2d_desc* d) { return (size_t)d->Co * 9 * d->Ci; }

int pk_dgrad_s2(const cnn_conv2d_desc* d, const float* dy, const float* w, float* dx, void* ws, hipStream_t s, bool prepared,
                const float* relu_below) {
    const int Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, 0), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, 0);
    const int total = d->Co * 9 * d->Ci;
    if (!prepared)
        CNN_KLAUNCH(s, "pack_dgrad_weights", (pack_dgrad_weights_s2<<<(total + 255) / 256, 256, 0, s>>>(w, (float*)ws, d->Ci, d->Co)),
                    CONV_TAG(d));
    const int V = (d->W + 1) / 2, UVg = ((d->H + 1) / 2) * V;
    const int ipi = (UVg + 63) / 64;
    const long long witems = (long long)d->B * ipi;
    const size_t wl = (size_t)total * sizeof(float);
    // (two pixels per lane, other batch sizes: no gain -- the loop is bound by the LDS-read / FMA interleave, profiles/NOTEBOOK.md 9)
    if (d->Ci == 16) {
        CNN_KLAUNCH(s, relu_below ? "conv_dgrad_pk_s2<16>+relu" : "conv_dgrad_pk_s2<16>",
                    (conv_dgrad_pk_s2<16, 4, 1><<<wave_grid(witems), kBlock, wl, s>>>(dy, (const v2f*)ws, dx, relu_below, d->B, d->Co, d->H, d->W, Ho,
                                                                                     Wo, ipi, div_magic(ipi), div_magic(V))),
                    CONV_TAG(d));
    } else {  // opt-in (CNN_AMD_PK_DGRAD=1): 32 input channels, slower than the implicit GEMM (71 vs 56 us)
        static DeviceOnce attr_once;
        if (attr_once.needed()) {
            CNN_HIP_CHECK(hipFuncSetAttribute((const void*)conv_dgrad_pk_s2<32, 4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              160 * 1024));
            attr_once.mark();
        }
        CNN_KLAUNCH(s, relu_below ? "conv_dgrad_pk_s2<32>+relu" : "conv_dgrad_pk_s2<32>",
                    (conv_dgrad_pk_s2<32, 4, 1><<<wave_grid(witems), kBlock, wl, s>>>(dy, (const v2f*)ws, dx, relu_below, d->B, d->Co, d->H, d->W, Ho,
                                                                                     Wo, ipi, div_magic(ipi), div_magic(V))),
                    CONV_TAG(d));
    }
    return CNN_AMD_OK;
}

// Packs, in ONE launch, the filters of every listed layer that runs on a kernel of this file; sets bit i of *fwd_done /
// *dgrad_done for the layers it handled (the implicit-GEMM layers are prepared by conv_igemm.hip).
int direct_prepare_batch(int n, const cnn_conv2d_desc* descs, const float* const* w, const float* const* bias,
                         void* const* fwd, void* const* dgrad, hipStream_t s, unsigned* fwd_done, unsigned* dgrad_done) {
    PackBatch pb;
    pb.n = 0;
    *fwd_done = *dgrad_done = 0;
    for (int i = 0; i < n; ++i) {
        const cnn_conv2d_desc* d = &descs[i];
        if (direct_conv_supported(d) && direct_fwd_pk_ok(d) && fwd && fwd[i] && pb.n < 8) {
            pb.j[pb.n++] = PackJob{0, w[i], bias[i], (float*)fwd[i], 3, 16};
            *fwd_done |= 1u << i;
        }
        if (direct_conv_supported(d) && direct_dgrad_pk_ok(d) && dgrad && dgrad[i] && pb.n < 8) {
            pb.j[pb.n++] = PackJob{1, w[i], nullptr, (float*)dgrad[i], 3, 16};
            *dgrad_done |= 1u << i;
        } else if (!direct_conv_supported(d) && pk_dgrad_s2_supported(d) && !dgrad_rd_supported(d) && dgrad && dgrad[i] && pb.n < 8) {
            pb.j[pb.n++] = PackJob{2, w[i], nullptr, (float*)dgrad[i], d->Ci, d->Co};
            *dgrad_done |= 1u << i;
        }
    }
    if (pb.n > 0) CNN_KLAUNCH(s, "pack_batch", (pack_batch<<<dim3(8, pb.n), 256, 0, s>>>(pb)), "jobs=%d", pb.n);
    return CNN_AMD_OK;
}
// can this layer's forward / data gradient run from prepared filters?
bool direct_prepared_fwd_ok(const cnn_conv2d_desc* d) { return direct_conv_supported(d) && direct_fwd_pk_ok(d); }
bool direct_prepared_dgrad_ok(const cnn_conv2d_desc* d) {
    return (direct_conv_supported(d) && direct_dgrad_pk_ok(d)) || (!direct_conv_supported(d) && pk_dgrad_s2_supported(d) && !dgrad_rd_supported(d));
}

// number of slabs (workgroups) the packed weight-gradient kernel writes; 0 when the geometry / sizes rule it out
int direct_wgrad_slots(const cnn_conv2d_desc* d) {
    if (!direct_conv_supported(d) || CNN_OPT_SET("WG_NOPK")) return 0;
    if (const int ws = win_wgrad_slots(d)) return ws;
    const int Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, 0), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, 0);
    if ((long long)d->B * 16 * Ho * Wo * 4 >= (1ll << 31) - 16 || (long long)d->B * 3 * d->H * d->W * 4 >= (1ll << 31) - 16) return 0;
    const long long items = (long long)d->B * ((Ho * Wo + 63) / 64);
    const long long cap = (long long)num_cus() * 2;  // two resident workgroups per CU (~200 VGPRs each); 1 or 3 measured slower
    return (int)(items < cap ? items : cap);
}

// writes direct_wgrad_slots(d) slabs of 16 x 28 floats ([27 weight sums | 1 bias sum] per output channel)
int direct_conv_wgrad(const cnn_conv2d_desc* d, const float* x, const float* dy, float* slabs, hipStream_t s) {
    if (win_wgrad_slots(d) > 0) return win_wgrad_launch(d, x, dy, nullptr, nullptr, slabs, s);
    const int Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, 0), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, 0);
    const int ipi = (Ho * Wo + 63) / 64;
    // three items of loads in flight per workgroup (2: 117 us, 3: 115 us, 4+: the extra registers cost more than they hide)
    CNN_KLAUNCH(s, "conv_wgrad_pk<3,16,3,2>",
                (conv_wgrad_pk_3_16_3_2<3, 0><<<direct_wgrad_slots(d), kBlock, 0, s>>>(x, dy, nullptr, nullptr, slabs, d->B, d->H, d->W,
                                                                                          Ho, Wo, ipi, div_magic(ipi), div_magic(Wo))),
                CONV_TAG(d));
    return CNN_AMD_OK;
}

// the same slabs from the pooled domain (dpool, mask, pooled of the 2x2 / stride-2 pool behind this layer's ReLU)
int direct_conv_wgrad_pooled(const cnn_conv2d_desc* d, const float* x, const float* dpool, const int32_t* mask, const float* pooled,
                             float* slabs, hipStream_t s) {
    CNN_REQUIRE(!(d->flags & CNN_CONV2D_POOL_MASK_PACKED) || (direct_pool_mask_packed_ok(d) && pooled == nullptr),
                "cnn_conv2d_backward_weight_pooled2: packed pool mask not available here (cnn_conv2d_pool_mask_packed_supported; pooled must be NULL)");
    if (win_wgrad_slots(d) > 0) return win_wgrad_launch(d, x, dpool, mask, pooled, slabs, s);
    const int Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, 0), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, 0);
    const int ipi = (Ho * Wo + 63) / 64;
    if (pooled)
        CNN_KLAUNCH(s, "conv_wgrad_pk<3,16,3,2>+pool",
                    (conv_wgrad_pk_3_16_3_2<3, 1><<<direct_wgrad_slots(d), kBlock, 0, s>>>(x, dpool, mask, pooled, slabs, d->B, d->H, d->W,
                                                                                          Ho, Wo, ipi, div_magic(ipi), div_magic(Wo))),
                    CONV_TAG(d));
    else
        CNN_KLAUNCH(s, "conv_wgrad_pk<3,16,3,2>+poolm",
                    (conv_wgrad_pk_3_16_3_2<3, 2><<<direct_wgrad_slots(d), kBlock, 0, s>>>(x, dpool, mask, nullptr, slabs, d->B, d->H, d->W,
                                                                                          Ho, Wo, ipi, div_magic(ipi), div_magic(Wo))),
                    CONV_TAG(d));
    return CNN_AMD_OK;
}

// see first_layer_finish; slabs: the [nslots][16][28] output of direct_conv_wgrad*(), images as cnn_conv2d_prepare_filters writes them
int direct_first_layer_finish(const cnn_conv2d_desc* d, const float* slabs, int nslots, float divisor, float* gw, float* gb, float* w,
                              float* bias, float lr, float grad_scale, void* fwd_img, void* dgrad_img, float* w_keep, float* bias_keep,
                              hipStream_t s) {
    CNN_REQUIRE(direct_conv_supported(d), "first_layer_finish: geometry not covered");
    CNN_REQUIRE(!fwd_img || direct_fwd_pk_ok(d), "first_layer_finish: this layer has no packed forward kernel");
    CNN_REQUIRE(!dgrad_img || direct_dgrad_pk_ok(d), "first_layer_finish: this layer has no packed data-gradient kernel");
    CNN_KLAUNCH(s, "first_layer_finish",
                (first_layer_finish<<<1, kFinThreads, 0, s>>>(slabs, nslots, divisor, gw, gb, w, bias, lr, grad_scale, grad_scale != 1.0f ? 1 : 0,
                                                              (float*)fwd_img, (float*)dgrad_img, w_keep, bias_keep)),
                CONV_TAG(d));
    return CNN_AMD_OK;
}

}  // namespace cnn_amd
