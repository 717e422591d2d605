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
#include <type_traits>

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


// ---- Ci = 3 under a K x K filter with stride 2 and pad (K-1)/2: the data gradient of the ResNet-shaped stack's 7x7 stem -------------
// (implicit GEMM: M = Ci = 3 of 32 MFMA rows, 17 TFLOP/s, 0.9 ms at batch 64 -- 1.2 ms beside the weight gradient.)
// A lane owns the 2 x 2 block of input pixels (2hh + ph, 2ww + pw) of one grid position, a wave 64 consecutive grid positions of
// one image (rows are crossed: every lane is busy).  With stride 2 the four pixels of a block use DISJOINT tap sets -- pixel parity
// ph takes the row taps r with r + ph odd, likewise the columns -- so every one of the 3*K*K filter values of a dy channel is used
// exactly once per block: 147 FMAs (K = 7) on the (K+1)/2 x (K+1)/2 = 4 x 4 dy neighbourhood (rows hh-1 .. hh+2, columns
// ww-1 .. ww+2), the filter value as a SCALAR operand straight from the reference layout (wave-uniform s_loads; the "prepared"
// image of such a layer is a verbatim copy of w).  Two channels per turn: 32 loads in flight while the previous 294 FMAs run.
//   dx[ci][y][x] = sum_co sum_{r,c} w[co][ci][r][c] * dy[co][(y + P - r)/2][(x + P - c)/2]   over the taps where both are integers
template <int K>
__global__ __launch_bounds__(kThinWaves * 64) void conv_dgrad_thin_s2(const float* __restrict__ dy, const float* __restrict__ w,
                                                                      const float* __restrict__ relu_below, float* __restrict__ dx, int B,
                                                                      int Co, int H, int W, int Ho, int Wo, int U, int V, int items_per_img) {
    constexpr int CI = 3, P = (K - 1) / 2, NB = (K + 1) / 2;  // NB x NB dy neighbourhood
    static_assert(K % 4 == 3, "tap <-> neighbour mapping below is written for K = 3, 7, 11");
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * kThinWaves + (threadIdx.x >> 6));
    const int b = wid / items_per_img;
    if (b >= B) return;
    const int n = (wid - b * items_per_img) * 64 + lane;
    const bool live = n < U * V;
    const int hh = (live ? n : 0) / V, ww = (live ? n : 0) - hh * V;
    // neighbour (j, i) = dy[hh - NB/2 + 1 ... ]: row hh + 1 - NB/2 + j ... for K = 7: rows hh-1 .. hh+2.  Row tap r of parity class ph
    // (r + ph + P even  <=>  (2hh + ph + P - r) / 2 integer) reads neighbour row j = (P + ph - r) / 2 + NB/2 - 1
    constexpr int J0 = NB / 2 - 1;  // index of row hh - ... : oy = hh + (P + ph - r)/2 - ... ; j = oy - (hh - J0)   (K = 7: J0 = 1)
    // byte offsets inside a channel plane; a neighbour outside dy gets an offset beyond the buffer: the raw buffer load returns 0
    // (no per-element selects, no exec masks -- 16 loop-invariant lane masks cost 32 SGPRs and spilled)
    constexpr unsigned kOOB = 0x7ffffffcu;
    unsigned voff[NB][NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int oy = hh - J0 + j, ox = ww - J0 + i;
            const bool ok = live && (unsigned)oy < (unsigned)Ho && (unsigned)ox < (unsigned)Wo;
            voff[j][i] = ok ? (unsigned)(oy * Wo + ox) * 4u : kOOB;
        }
    float acc[2][2][CI];
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int pw = 0; pw < 2; ++pw)
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) acc[ph][pw][ci] = 0.f;
    const int oplane = Ho * Wo;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((unsigned)B * Co * oplane * 4u), 0x00020000);
    int soff = b * Co * oplane * 4;  // byte offset of dy[b][co]: wave-uniform, advanced per channel
    const float* wc0 = w;            // wave-uniform: scalar loads
    auto load_patch = [&](float (&v)[NB][NB], int so) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < NB; ++i) v[j][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff[j][i], so, 0));
    };
    // The filter values are wave-uniform: s_load into SGPRs, a scalar operand of every FMA.  Left to hipcc, the s_loads of ALL 147
    // values (x 2 channels) are pulled to the top of the unrolled block -- 300-500 SGPR spills through v_writelane; compiler fences
    // (the loads are invariant) and scheduling barriers do not stop it, and a real loop over row pairs with a switch body costs a
    // branch tree + an exposed s_load latency per 14 FMAs (688 us).  So the loads are issued by hand: inline-asm s_load of one
    // (ci, r) row pair (14 values: x8 + x4 + x2) into one of two register sets, one pair ahead of the FMAs; SMEM returns out of
    // order, so the only safe wait is lgkmcnt(0), placed where the NEXT pair has not been issued yet.
    typedef float s8f __attribute__((ext_vector_type(8)));
    typedef float s4f __attribute__((ext_vector_type(4)));
    typedef float s2f __attribute__((ext_vector_type(2)));
    struct RowPair {
        s8f a;
        s4f b;
        s2f c;
    };
    constexpr int NPAIR = (CI * K + 1) / 2;
    static_assert(K == 7, "row-pair loads below are x8 + x4 + x2 (+ x4 + x2 + x1 for the odd last row)");
    auto issue = [&](auto Q, RowPair& rp, const float* wq) {
        constexpr int q = decltype(Q)::value;
        constexpr int off = q * 2 * K * 4;  // byte offset of row 2q
        // (the running sums as inputs: the FMAs of the pair that last used this register set come first -- otherwise hipcc keeps
        // every pair's values live and runs all FMAs at the end)
#define THIN_ACCS "v"(acc[0][0][0]), "v"(acc[0][1][0]), "v"(acc[1][0][0]), "v"(acc[1][1][0]), "v"(acc[0][0][1]), "v"(acc[0][1][1]), \
                  "v"(acc[1][0][1]), "v"(acc[1][1][1]), "v"(acc[0][0][2]), "v"(acc[0][1][2]), "v"(acc[1][0][2]), "v"(acc[1][1][2])
        if constexpr (2 * q + 1 < CI * K) {
            asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(rp.a) : "s"(wq), "n"(off), THIN_ACCS);
            asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(rp.b) : "s"(wq), "n"(off + 32));
            asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(rp.c) : "s"(wq), "n"(off + 48));
        } else {  // the last row of a channel stands alone: exactly its 7 values (nothing behind the last channel is touched)
            asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(rp.b) : "s"(wq), "n"(off), THIN_ACCS);
            asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(rp.c) : "s"(wq), "n"(off + 16));
            float last;
            asm volatile("s_load_dword %0, %1, %2" : "=s"(last) : "s"(wq), "n"(off + 24));
            rp.a[0] = last;
        }
#undef THIN_ACCS
    };
    auto landed = [&](RowPair& rp) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(rp.a), "+s"(rp.b), "+s"(rp.c)); };
    auto fmas = [&](auto Q, const RowPair& rp, const float (&v)[NB][NB]) {
        constexpr int q = decltype(Q)::value;
#pragma unroll
        for (int e = 0; e < 2 * K; ++e) {
            const int cr = 2 * q + e / K, c = e % K;
            if (cr < CI * K) {
                const int ci = cr / K, r = cr - ci * K;
                float wv;
                if constexpr (2 * q + 1 < CI * K) wv = e < 8 ? rp.a[e] : (e < 12 ? rp.b[e - 8] : rp.c[e - 12]);
                else wv = e < 4 ? rp.b[e] : (e < 6 ? rp.c[e - 4] : rp.a[0]);
                // pixel parities this tap belongs to, and the neighbour it multiplies:  y + P - r = 2 oy  with  y = 2hh + ph
                const int ph = (r + P) & 1, pw = (c + P) & 1;
                const int j = (P + ph - r) / 2 + J0, i = (P + pw - c) / 2 + J0;  // (numerators are even by construction)
                acc[ph][pw][ci] = __builtin_fmaf(wv, v[j][i], acc[ph][pw][ci]);
            }
        }
    };
    auto accumulate = [&](const float (&v)[NB][NB], const float* wq) {
        RowPair ra, rb;
        ra.a = rb.a = s8f{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        issue(std::integral_constant<int, 0>(), ra, wq);
#define THIN_PAIR(Q, CUR, NXT)                                                                 \
    landed(CUR);                                                                               \
    if constexpr (Q + 1 < NPAIR) issue(std::integral_constant<int, (Q + 1 < NPAIR ? Q + 1 : Q)>(), NXT, wq); \
    fmas(std::integral_constant<int, Q>(), CUR, v);
        THIN_PAIR(0, ra, rb) THIN_PAIR(1, rb, ra) THIN_PAIR(2, ra, rb) THIN_PAIR(3, rb, ra) THIN_PAIR(4, ra, rb) THIN_PAIR(5, rb, ra)
        THIN_PAIR(6, ra, rb) THIN_PAIR(7, rb, ra) THIN_PAIR(8, ra, rb) THIN_PAIR(9, rb, ra) THIN_PAIR(10, ra, rb)
#undef THIN_PAIR
        static_assert(NPAIR == 11, "THIN_PAIR sequence above");
    };
    int co = 0;
    for (; co + 1 < Co; co += 2) {
        float va[NB][NB], vb[NB][NB];
        load_patch(va, soff);
        load_patch(vb, soff + oplane * 4);
        accumulate(va, wc0);
        accumulate(vb, wc0 + CI * K * K);
        soff += 2 * oplane * 4;
        wc0 += 2 * CI * K * K;
    }
    if (co < Co) {
        float va[NB][NB];
        load_patch(va, soff);
        accumulate(va, wc0);
    }
    if (live) {
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const int y = 2 * hh + ph;
                if (y < H) {
#pragma unroll
                    for (int pw = 0; pw < 2; ++pw) {
                        const int x = 2 * ww + pw;
                        if (x < W) {
                            const size_t o = (((size_t)b * CI + ci) * H + y) * W + x;
                            float val = acc[ph][pw][ci];
                            if (relu_below) val = (relu_below[o] <= 0.f) ? 0.f : val;
                            dx[o] = val;
                        }
                    }
                }
            }
    }
}


// ---- the same data gradient on PACKED fp32 FMAs (round 6) ----------------------------------------------------------------------------
// The scalar-operand kernel above spends 147 v_fma_f32 per dy channel and grid position; on gfx950 v_pk_fma_f32 does two FMAs per lane
// in the same issue slot.  Taps c and c + 1 (c odd) of a filter row multiply the SAME dy neighbour and feed the two column parities of
// one pixel row: with the running sums kept as pairs (pw = 0, pw = 1) of (row parity ph, ci) they are ONE packed FMA -- filter pair as
// an SGPR pair, the neighbour broadcast to both halves.  Tap 0 (pw = 1 only) stays a scalar FMA on the pair's second half: 4 instead of 7
// instructions per filter row, 84 instead of 147 per channel, the SAME products added in the SAME order (bit-identical to the kernel
// above).  The filters come re-packed, one 32-byte row per (channel, ci, tap row): [w0, 0, w1, w2, w3, w4, w5, w6] -- a row is one
// s_load_dwordx8 whose register pairs are the packed operands; three rows per register set, one set ahead of the FMAs.
constexpr int kThinPkRow = 8;
__global__ void thin_pack_k7(const float* __restrict__ w, float* __restrict__ wp, int rows) {  // rows = Co * 3 * 7
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* src = w + (size_t)r * 7;
    float* dst = wp + (size_t)r * kThinPkRow;
    dst[0] = src[0];
    dst[1] = 0.f;
#pragma unroll
    for (int c = 1; c < 7; ++c) dst[1 + c] = src[c];
}

__global__ __launch_bounds__(kThinWaves * 64) void conv_dgrad_thin_s2_pk7(const float* __restrict__ dy, const float* __restrict__ wp,
                                                                         const float* __restrict__ relu_below, float* __restrict__ dx, int B,
                                                                         int Co, int H, int W, int Ho, int Wo, int U, int V, int items_per_img) {
    constexpr int K = 7, CI = 3, P = 3, NB = 4, J0 = 1, ROWS = CI * K;  // (geometry as in conv_dgrad_thin_s2<7>)
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef float s8f __attribute__((ext_vector_type(8)));
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * kThinWaves + (threadIdx.x >> 6));
    const int b = wid / items_per_img;
    if (b >= B) return;
    const int n = (wid - b * items_per_img) * 64 + lane;
    const bool live = n < U * V;
    const int hh = (live ? n : 0) / V, ww = (live ? n : 0) - hh * V;
    constexpr unsigned kOOB = 0x7ffffffcu;
    unsigned voff[NB][NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int oy = hh - J0 + j, ox = ww - J0 + i;
            const bool ok = live && (unsigned)oy < (unsigned)Ho && (unsigned)ox < (unsigned)Wo;
            voff[j][i] = ok ? (unsigned)(oy * Wo + ox) * 4u : kOOB;
        }
    v2f acc[2][CI];  // [row parity][ci] = (column parity 0, column parity 1)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) acc[ph][ci] = v2f{0.f, 0.f};
    const int oplane = Ho * Wo;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((unsigned)B * Co * oplane * 4u), 0x00020000);
    int soff = b * Co * oplane * 4;
    const float* wq0 = wp;  // wave-uniform
    // (one 16-byte load per patch row instead of four 4-byte loads was measured SLOWER: 343 against 262 us -- NB round 6)
    auto load_patch = [&](float (&v)[NB][NB], int so) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int i = 0; i < NB; ++i) v[j][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff[j][i], so, 0));
    };
    struct Rows3 {
        s8f r0, r1, r2;
    };
    constexpr int NG = ROWS / 3;  // 7 groups of three rows per channel
    static_assert(NG * 3 == ROWS, "three rows per register set");
#define THIN_ACCS "v"(a00), "v"(a01), "v"(a02), "v"(a10), "v"(a11), "v"(a12)
    auto issue = [&](auto G, Rows3& rs, const float* wq) {  // (the running sums as inputs: see the scalar kernel)
        constexpr int off = decltype(G)::value * 3 * kThinPkRow * 4;
        v2f &a00 = acc[0][0], &a01 = acc[0][1], &a02 = acc[0][2], &a10 = acc[1][0], &a11 = acc[1][1], &a12 = acc[1][2];  // (asm operands inside a generic lambda do not capture: named first)
        asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(rs.r0) : "s"(wq), "n"(off), THIN_ACCS);
        asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(rs.r1) : "s"(wq), "n"(off + 32));
        asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(rs.r2) : "s"(wq), "n"(off + 64));
    };
#undef THIN_ACCS
    auto landed = [&](Rows3& rs) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(rs.r0), "+s"(rs.r1), "+s"(rs.r2)); };
    auto row = [&](auto CR, const s8f& q, const float (&v)[NB][NB]) {
        constexpr int cr = decltype(CR)::value, ci = cr / K, r = cr % K;
        constexpr int ph = (r + P) & 1, j = (P + ph - r) / 2 + J0;
        v2f a = acc[ph][ci];
        a.y = __builtin_fmaf(q[0], v[j][3], a.y);                                        // tap 0: column parity 1, neighbour ww + 2
        a = __builtin_elementwise_fma(v2f{q[2], q[3]}, v2f{v[j][2], v[j][2]}, a);        // taps 1, 2: neighbour ww + 1
        a = __builtin_elementwise_fma(v2f{q[4], q[5]}, v2f{v[j][1], v[j][1]}, a);        // taps 3, 4: neighbour ww
        a = __builtin_elementwise_fma(v2f{q[6], q[7]}, v2f{v[j][0], v[j][0]}, a);        // taps 5, 6: neighbour ww - 1
        acc[ph][ci] = a;
    };
    auto fmas = [&](auto G, const Rows3& rs, const float (&v)[NB][NB]) {
        constexpr int g = decltype(G)::value;
        row(std::integral_constant<int, 3 * g>(), rs.r0, v);
        row(std::integral_constant<int, 3 * g + 1>(), rs.r1, v);
        row(std::integral_constant<int, 3 * g + 2>(), rs.r2, v);
    };
    auto accumulate = [&](const float (&v)[NB][NB], const float* wq) {
        Rows3 ra, rb;
        issue(std::integral_constant<int, 0>(), ra, wq);
#define THIN_GROUP(G, CUR, NXT)                                                                  \
    landed(CUR);                                                                                 \
    if constexpr (G + 1 < NG) issue(std::integral_constant<int, (G + 1 < NG ? G + 1 : G)>(), NXT, wq); \
    fmas(std::integral_constant<int, G>(), CUR, v);
        THIN_GROUP(0, ra, rb) THIN_GROUP(1, rb, ra) THIN_GROUP(2, ra, rb) THIN_GROUP(3, rb, ra) THIN_GROUP(4, ra, rb) THIN_GROUP(5, rb, ra)
        THIN_GROUP(6, ra, rb)
#undef THIN_GROUP
        static_assert(NG == 7, "THIN_GROUP sequence above");
    };
    int co = 0;
    for (; co + 1 < Co; co += 2) {
        float va[NB][NB], vb[NB][NB];
        load_patch(va, soff);
        load_patch(vb, soff + oplane * 4);
        accumulate(va, wq0);
        accumulate(vb, wq0 + ROWS * kThinPkRow);
        soff += 2 * oplane * 4;
        wq0 += 2 * ROWS * kThinPkRow;
    }
    if (co < Co) {
        float va[NB][NB];
        load_patch(va, soff);
        accumulate(va, wq0);
    }
    if (live) {
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const int y = 2 * hh + ph;
                if (y < H) {
#pragma unroll
                    for (int pw = 0; pw < 2; ++pw) {
                        const int x = 2 * ww + pw;
                        if (x < W) {
                            const size_t o = (((size_t)b * CI + ci) * H + y) * W + x;
                            float val = pw == 0 ? acc[ph][ci].x : acc[ph][ci].y;
                            if (relu_below) val = (relu_below[o] <= 0.f) ? 0.f : val;
                            dx[o] = val;
                        }
                    }
                }
            }
    }
}

}  // namespace

namespace cnn_amd {

#define CONV_TAG(d) "B%d Ci%d %dx%d Co%d k%d s%d p%d", (d)->B, (d)->Ci, (d)->H, (d)->W, (d)->Co, (d)->k, (d)->s, (d)->pad

bool thin_dgrad_supported(const cnn_conv2d_desc* d) {
    const OptVal e = CNN_OPT_VAL("DGRAD_THIN");
    if (e && atoi(e) == 0) return false;
    if (d->Ci == 3 && d->k == 7 && d->s == 2 && d->pad == 3)  // the 7x7 stem: conv_dgrad_thin_s2
        return (long long)d->B * d->Co * cnn_conv2d_out_dim(d->H, 7, 2, 3) * cnn_conv2d_out_dim(d->W, 7, 2, 3) < (1ll << 29) &&  // (32-bit byte offsets)
               (long long)d->B * (((d->H + 1) / 2) * ((d->W + 1) / 2) + 63) / 64 < (1ll << 31) - 8;
    return d->Ci == 3 && d->k == 3 && d->s == 1 && d->pad >= 0 && d->pad <= 1 && (long long)d->B * ((d->H + 3) / 4) * ((d->W + 63) / 64) < (1ll << 31) - 8;
}

// the packed image of the 7x7 stem's data gradient (conv_dgrad_thin_s2_pk7): floats, and the kernel that makes it; 0 = this layer has none
size_t thin_dgrad_packed_floats(const cnn_conv2d_desc* d) {
    if (!(d->Ci == 3 && d->k == 7 && d->s == 2 && d->pad == 3) || !thin_dgrad_supported(d)) return 0;
    if (CNN_OPT_SET("DGRAD_THIN_PK") && CNN_OPT_INT("DGRAD_THIN_PK", 1) == 0) return 0;  // (A/B: the scalar-operand kernel)
    return (size_t)d->Co * 3 * 7 * kThinPkRow;
}
int thin_dgrad_pack(const cnn_conv2d_desc* d, const float* w, float* image, hipStream_t s) {
    const int rows = d->Co * 3 * 7;
    CNN_KLAUNCH(s, "thin_pack_k7", (thin_pack_k7<<<(rows + 255) / 256, 256, 0, s>>>(w, image, rows)), "Co%d", d->Co);
    return CNN_AMD_OK;
}

// w: the filters in the reference layout [Co][3][k][k] (for the *_prepared entry points: the verbatim copy cnn_conv2d_prepare_filters made);
// packed (nullable): the image thin_dgrad_pack made of them, for layers with thin_dgrad_packed_floats() > 0
int thin_dgrad(const cnn_conv2d_desc* d, const float* dy, const float* w, const float* packed, const float* relu_below, float* dx, hipStream_t s) {
    const int Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, d->pad), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, d->pad);
    if (d->s == 2) {
        const int U = (d->H + 1) / 2, V = (d->W + 1) / 2, ipi = (U * V + 63) / 64;
        const long long nw = (long long)d->B * ipi;
        const unsigned g2 = (unsigned)((nw + kThinWaves - 1) / kThinWaves);
        if (packed) {
            CNN_KLAUNCH(s, relu_below ? "conv_dgrad_thin_pk<3,k7s2>+relu" : "conv_dgrad_thin_pk<3,k7s2>",
                        (conv_dgrad_thin_s2_pk7<<<g2, kThinWaves * 64, 0, s>>>(dy, packed, relu_below, dx, d->B, d->Co, d->H, d->W, Ho, Wo, U, V, ipi)),
                        CONV_TAG(d));
            return CNN_AMD_OK;
        }
        CNN_REQUIRE(w != nullptr, "conv_dgrad_thin: the filters are null");
        CNN_KLAUNCH(s, relu_below ? "conv_dgrad_thin<3,k7s2>+relu" : "conv_dgrad_thin<3,k7s2>",
                    (conv_dgrad_thin_s2<7><<<g2, kThinWaves * 64, 0, s>>>(dy, w, relu_below, dx, d->B, d->Co, d->H, d->W, Ho, Wo, U, V, ipi)),
                    CONV_TAG(d));
        return CNN_AMD_OK;
    }
    const int bands = (d->H + 3) / 4, segs = (d->W + 63) / 64;
    const long long waves = (long long)d->B * bands * segs;
    const unsigned grid = (unsigned)((waves + kThinWaves - 1) / kThinWaves);
    CNN_KLAUNCH(s, relu_below ? "conv_dgrad_thin<3,s1>+relu" : "conv_dgrad_thin<3,s1>",
                (conv_dgrad_thin_s1k3<3, 4><<<grid, kThinWaves * 64, 0, s>>>(dy, w, relu_below, dx, d->B, d->Co, d->H, d->W, Ho, Wo, d->pad, bands, segs)),
                CONV_TAG(d));
    return CNN_AMD_OK;
}

}  // namespace cnn_amd
