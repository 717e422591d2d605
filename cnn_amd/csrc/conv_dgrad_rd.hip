// conv_dgrad_rd.hip -- "register-direct" Conv2D data gradient (cpu/src/conv2d.cpp:168-199) for the 3x3 / stride-2 /
// pad-0 layers with Co in {64, 128} and Ci a multiple of 32 (conv_layer_3 / _4 of the reference net):
//     dx[ci][h][w] = sum_{co,kx,ky : (h-kx), (w-ky) even} dy[co][(h-kx)/2][(w-ky)/2] * w[co][ci][kx][ky].
// A grid pixel (u,v) owns the 2x2 block dx[.][2u+ph][2v+pw] (its four parity classes).  Class (ph,pw) only sees the taps
// kx = ph + 2jr, ky = pw + 2jc, i.e. the dy values D[jr][jc] = dy[co][u-jr][v-jc], jr,jc in {0,1}:
//     class (0,0): 4 taps | (0,1): 2 | (1,0): 2 | (1,1): 1      = the 9 filter taps, no structurally-zero products.
// GEMM per class on v_mfma_f32_32x32x2_f32: M = 32 input channels ci, N = 32 consecutive grid pixels, K = (co, tap);
// k-slot kg of the MFMA owns the dy channels [kg*Co/2, (kg+1)*Co/2).
//   * B operand: lane (pixel, kg) loads, per dy channel, TWO 8-byte pairs (rows u and u-1, columns v-1 | v) straight into
//     registers: 2 loads feed the 9 MFMA steps of all four classes (the implicit-GEMM kernel re-stages dy through LDS per
//     class and multiplies the 7/16 empty (class, tap) slots it cannot skip inside a tile);
//   * A operand: lane (ci, kg) needs w[co][ci][0..8] -- NINE CONSECUTIVE floats of the reference's own filter layout per dy
//     channel: two 16-byte loads + one 4-byte load straight from L1/L2 (the filters are L2-resident) feed the 9 steps.
//     Neither operand touches LDS: no upload phase, no barrier, no re-laid-out filter copy;
//   * borders (u-1 < 0, v-1 < 0, u >= Ho, v >= Wo) are handled by clamped addresses + per-lane select codes, the optional
//     ReLU::backward of the layer in front (relu.cpp:35-40) in the store epilogue; stores are 8-byte (pw = 0,1) pairs,
//     contiguous across the wave.
// Software pipeline: a ring of 4 channel groups (2 dy channels each) of both operands in flight, pinned by RD_PIPE_FENCE
// (see conv_wgrad_rd.hip).
#include <cstdlib>
#include <type_traits>

#include "common.h"

using namespace cnn_amd;

namespace cnn_amd {
bool igemm_preferred(const cnn_conv2d_desc* d, int mode);  // conv_igemm.hip (mode 0 forward, 1 data gradient)
}

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
struct __attribute__((packed, aligned(4))) f2u {
    float x, y;
};
struct __attribute__((packed, aligned(4))) f4u {
    float x, y, z, w;
};
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
struct Taps {  // the 9 filter taps of one (co, ci)
    v4f a, b;
    float c;
};

#define RD_PIPE_FENCE(reg) asm volatile("" : "+v"(reg) : : "memory")

struct DgRdParams {
    const float* dy;
    const float* w;     // [Co][Ci][3][3], or with tr != 0 the prepared transposed copy [Co][9][Ci]
    int tr;
    const float* relu_below;  // nullable: dx = (relu_below <= 0) ? 0 : dx
    float* dx;
    int B, Ci, H, W, Ho, Wo, U, V, UV;
    int pixels, tiles;       // B*U*V, ceil(pixels / 32)
    unsigned m_uv, m_v;      // magic multipliers
    int dbg;                 // CNN_AMD_DGRAD_RD_DBG=1: workgroup 0 prints its phase cycle counts
};

__device__ __forceinline__ int fdiv(int n, unsigned magic, int d) {
    if (d == 1) return n;
    int q = (int)__umulhi((unsigned)n, magic);
    if (q * d > n) --q;
    return q;
}

// CO dy channels (even), NW waves per workgroup, UC dy channels per pipeline group
// HALF (Ci = 16): the 32 MFMA rows hold 16 channels x TWO parity classes -- rows 0-15 class (ph,0), rows 16-31 class (ph,1) --
// so a dy channel takes 4 + 2 = 6 MFMA steps on two accumulators instead of 9 steps on four half-empty ones
template <int CO, int NW, int UC, int NB, bool HALF = false>
__global__ __launch_bounds__(NW * 64) void conv_dgrad_rd_s2_kernel(const DgRdParams p) {
    constexpr int CH = CO / 2;   // dy channels per k-slot
    constexpr int G = CH / UC;   // pipeline groups per tile
    static_assert(CH % UC == 0 && G % NB == 0, "ring index must line up from tile to tile");
    const int lane = threadIdx.x & 63, n = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ci0 = blockIdx.y * 32;
    // this lane's filter column: w[kg*CH + .][ci0 + n][.]  (a lane beyond Ci reads channel ci0: its rows are not stored)
    const int nci = HALF ? (n & 15) : n;       // channel row of this lane inside the tile
    const bool upper = HALF && n >= 16;        // HALF: this lane's MFMA row belongs to the pw = 1 class
    const unsigned wlane = (unsigned)((kg * CH * p.Ci + ci0 + (ci0 + nci < p.Ci ? nci : 0)) * 9);
    const unsigned wlane_tr = (unsigned)(kg * CH * p.Ci * 9 + ci0 + (ci0 + nci < p.Ci ? nci : 0));
    const unsigned plane = (unsigned)(p.Ho * p.Wo);
    const int tstep = gridDim.x * NW;

    // per tile and lane: offsets of the row-u and row-(u-1) pairs, select codes for columns v and v-1 (0 none, 1 first,
    // 2 second element of the pair), row validity, and the dx offset of the 2x2 block
    struct Loc {
        unsigned o0, o1;   // element offsets into dy (channel kg*CH), rows u and u-1, pair start column
        bool s0, s1;       // D[.][0] (column v) / D[.][1] (column v-1) is the SECOND element of the loaded pair
        bool v00, v01, v10, v11;  // D[jr][jc] exists (row and column inside dy)
        unsigned xo;       // element offset of dx[b][ci0][2u][2v]
        int u, v;
        bool live;
    };
    auto locate = [&](int tile, Loc& L) {
        const int pi = tile * 32 + n;
        L.live = pi < p.pixels;
        const int pic = L.live ? pi : p.pixels - 1;
        const int b = fdiv(pic, p.m_uv, p.UV), rem = pic - b * p.UV;
        const int u = fdiv(rem, p.m_v, p.V), v = rem - u * p.V;
        L.u = u; L.v = v;
        int cs = v - 1;  // pair start column, clamped so that both elements lie inside the row
        cs = cs < 0 ? 0 : (cs > p.Wo - 2 ? p.Wo - 2 : cs);
        L.s0 = v != cs;
        L.s1 = v - 1 != cs;
        const bool cv0 = v < p.Wo, cv1 = v >= 1 && v - 1 < p.Wo;
        const bool r0 = L.live && u < p.Ho, r1 = L.live && u >= 1 && u - 1 < p.Ho;
        L.v00 = r0 && cv0; L.v01 = r0 && cv1; L.v10 = r1 && cv0; L.v11 = r1 && cv1;
        const int ur0 = u < p.Ho ? u : p.Ho - 1, ur1 = u >= 1 ? (u - 1 < p.Ho ? u - 1 : p.Ho - 1) : 0;
        const unsigned cb = (unsigned)((b * CO + kg * CH) * (int)plane);
        L.o0 = cb + (unsigned)(ur0 * p.Wo + cs);
        L.o1 = cb + (unsigned)(ur1 * p.Wo + cs);
        L.xo = (unsigned)(((b * p.Ci + ci0) * p.H + 2 * u) * p.W + 2 * v);
    };
    // one channel group = UC dy channels: 2 pairs of dy + 9 filter taps (16 + 16 + 4 bytes) each.  (Issuing these loads
    // through inline asm with hand-counted s_waitcnt -- hipcc drains the queue, vmcnt(0), at the top of every ring turn --
    // changed nothing measurable and is unsafe: the compiler may move a "defined" destination register before the data
    // has arrived.)
    auto load_group = [&](v2f (&buf)[UC * 2], int g, unsigned o0, unsigned o1) {
#pragma unroll
        for (int uu = 0; uu < UC; ++uu) {
            const float* base = p.dy + (size_t)(g * UC + uu) * plane;  // wave-uniform
            const f2u q0 = *(const f2u*)(base + o0), q1 = *(const f2u*)(base + o1);
            buf[uu * 2 + 0] = v2f{q0.x, q0.y};
            buf[uu * 2 + 1] = v2f{q1.x, q1.y};
        }
    };
    auto load_filters = [&](Taps (&a)[UC], int g) {
#pragma unroll
        for (int uu = 0; uu < UC; ++uu) {
            const float* base = p.w + (size_t)(g * UC + uu) * p.Ci * 9;  // wave-uniform
            if (p.tr) {  // prepared [co][tap][ci]: the 32 lanes of a k-slot read 32 consecutive floats per tap
                float t[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) t[k] = base[wlane_tr + k * p.Ci];
                a[uu].a = v4f{t[0], t[1], t[2], t[3]};
                a[uu].b = v4f{t[4], t[5], t[6], t[7]};
                a[uu].c = t[8];
            } else {
                const f4u qa = *(const f4u*)(base + wlane), qb = *(const f4u*)(base + wlane + 4);
                a[uu].a = v4f{qa.x, qa.y, qa.z, qa.w};
                a[uu].b = v4f{qb.x, qb.y, qb.z, qb.w};
                a[uu].c = base[wlane + 8];
            }
        }
    };
    // (two selects per value; a 3-way code here made hipcc emit a divergent switch -- ~40 branches per channel)
    auto pick = [](const v2f& v, bool second, bool valid) {
        const float t = second ? v.y : v.x;
        return valid ? t : 0.f;
    };

    int tile = blockIdx.x * NW + wave;
    if (tile >= p.tiles) return;
    const long long dbg_t0 = p.dbg ? clock64() : 0;
    Loc cur, nxt;
    locate(tile, cur);
    nxt = cur;
    v2f xb[NB][UC * 2];
    Taps ab[NB][UC];
    {
        const bool more0 = tile + tstep < p.tiles;
        if (more0) locate(tile + tstep, nxt);
#pragma unroll
        for (int g = 0; g < NB - 1; ++g) {  // (G >= NB: these are groups of the first tile)
            load_group(xb[g], g, cur.o0, cur.o1);
            load_filters(ab[g], g);
        }
    }
    for (; tile < p.tiles; tile += tstep) {
        const bool more = tile + tstep < p.tiles;
        if (more) locate(tile + tstep, nxt);
        constexpr int NACC = HALF ? 2 : 4;
        f32x16 acc[NACC];  // classes (ph,pw) = 00, 01, 10, 11; HALF: ph = 0, 1 (pw in the row halves)
#pragma unroll
        for (int c = 0; c < NACC; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        // (a run-time loop over ring turns: fully unrolled, hipcc kept hundreds of addresses live -- 432 to 512 VGPRs.  The
        //  request for group g+3 is branch-free: behind the tile's last group it simply addresses the next tile.)
#pragma unroll 1
        for (int g0 = 0; g0 < G; g0 += NB) {
#pragma unroll
            for (int ri = 0; ri < NB; ++ri) {
                const int g = g0 + ri, gp = g + NB - 1, rp = (ri + NB - 1) % NB;
                const bool wrap = gp >= G;
                const int gq = wrap ? gp - G : gp;
                load_group(xb[rp], gq, wrap ? nxt.o0 : cur.o0, wrap ? nxt.o1 : cur.o1);
                load_filters(ab[rp], gq);
                RD_PIPE_FENCE(ab[ri][0].c);
#pragma unroll
                for (int uu = 0; uu < UC; ++uu) {
                    const v2f p0 = xb[ri][uu * 2 + 0], p1 = xb[ri][uu * 2 + 1];
                    const float d00 = pick(p0, cur.s0, cur.v00), d01 = pick(p0, cur.s1, cur.v01);  // D[0][jc]
                    const float d10 = pick(p1, cur.s0, cur.v10), d11 = pick(p1, cur.s1, cur.v11);  // D[1][jc]
                    const Taps& t9 = ab[ri][uu];
                    const float a[9] = {t9.a.x, t9.a.y, t9.a.z, t9.a.w, t9.b.x, t9.b.y, t9.b.z, t9.b.w, t9.c};  // taps kx*3 + ky
                    if constexpr (HALF) {
                        // rows 0-15 | 16-31:  D00: taps (0,0) | (0,1);  D10: (2,0) | (2,1);  D01: (0,2) | -;  D11: (2,2) | -
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(upper ? a[1] : a[0], d00, acc[0], 0, 0, 0);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(upper ? a[7] : a[6], d10, acc[0], 0, 0, 0);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(upper ? 0.f : a[2], d01, acc[0], 0, 0, 0);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(upper ? 0.f : a[8], d11, acc[0], 0, 0, 0);
                        //                     D00: taps (1,0) | (1,1);  D01: (1,2) | -
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(upper ? a[4] : a[3], d00, acc[1], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(upper ? 0.f : a[5], d01, acc[1], 0, 0, 0);
                        continue;
                    }
                    // class (0,0): taps (0,0) D00, (0,2) D01, (2,0) D10, (2,2) D11
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], d00, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], d01, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[6], d10, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[8], d11, acc[0], 0, 0, 0);
                    // class (0,1): taps (0,1) D00, (2,1) D10
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], d00, acc[1], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[7], d10, acc[1], 0, 0, 0);
                    // class (1,0): taps (1,0) D00, (1,2) D01
                    acc[NACC - 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], d00, acc[NACC - 2], 0, 0, 0);
                    acc[NACC - 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[5], d01, acc[NACC - 2], 0, 0, 0);
                    // class (1,1): tap (1,1) D00
                    acc[NACC - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4], d00, acc[NACC - 1], 0, 0, 0);
                }
            }
        }
        const long long dbg_t1 = p.dbg ? clock64() : 0;
        // ---- epilogue: dx[ci][2u + ph][2v .. 2v+1]
        if constexpr (HALF) {
            if (cur.live) {  // accumulator register r (< 8) = channel row, r + 8 = the same channel's pw = 1 value
                const bool w1 = 2 * cur.v + 1 < p.W, h1 = 2 * cur.u + 1 < p.H;
                float m0[8][2], m1[8][2];
                if (p.relu_below) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int rowl = (i & 3) + 8 * (i >> 2) + 4 * kg;
                        const size_t o = (size_t)cur.xo + (size_t)(ci0 + rowl < p.Ci ? rowl : 0) * p.H * p.W;
#pragma unroll
                        for (int ph = 0; ph < 2; ++ph) {
                            const size_t oo = o + (size_t)((ph == 1 && h1) ? p.W : 0);
                            m0[i][ph] = p.relu_below[oo];
                            m1[i][ph] = w1 ? p.relu_below[oo + 1] : 1.f;
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int rowl = (i & 3) + 8 * (i >> 2) + 4 * kg;
                    if (ci0 + rowl >= p.Ci) continue;
                    const size_t o = (size_t)cur.xo + (size_t)rowl * p.H * p.W;
#pragma unroll
                    for (int ph = 0; ph < 2; ++ph) {
                        if (ph == 1 && !h1) continue;
                        const size_t oo = o + (size_t)ph * p.W;
                        float v0 = acc[ph][i], v1 = acc[ph][i + 8];
                        if (p.relu_below) {
                            v0 = (m0[i][ph] <= 0.f) ? 0.f : v0;
                            v1 = (m1[i][ph] <= 0.f) ? 0.f : v1;
                        }
                        if (w1) *(f2u*)(p.dx + oo) = f2u{v0, v1};
                        else p.dx[oo] = v0;
                    }
                }
            }
        } else if (cur.live) {
            const bool w1 = 2 * cur.v + 1 < p.W, h1 = 2 * cur.u + 1 < p.H;
            // 8 accumulator rows at a time: all mask values of the batch are requested before the first one is used (a
            // load per store would serialise 32 round trips)
#pragma unroll
            for (int rb = 0; rb < 16; rb += 8) {
                float m0[8][2], m1[8][2];
                if (p.relu_below) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int r = rb + i, rowl = (r & 3) + 8 * (r >> 2) + 4 * kg;
                        const size_t o = (size_t)cur.xo + (size_t)(ci0 + rowl < p.Ci ? rowl : 0) * p.H * p.W;
#pragma unroll
                        for (int ph = 0; ph < 2; ++ph) {
                            const size_t oo = o + (size_t)((ph == 1 && h1) ? p.W : 0);
                            m0[i][ph] = p.relu_below[oo];
                            m1[i][ph] = w1 ? p.relu_below[oo + 1] : 1.f;
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = rb + i, rowl = (r & 3) + 8 * (r >> 2) + 4 * kg;
                    if (ci0 + rowl >= p.Ci) continue;
                    const size_t o = (size_t)cur.xo + (size_t)rowl * p.H * p.W;
#pragma unroll
                    for (int ph = 0; ph < 2; ++ph) {
                        if (ph == 1 && !h1) continue;
                        const size_t oo = o + (size_t)ph * p.W;
                        float v0 = acc[ph * 2][r], v1 = acc[ph * 2 + 1][r];
                        if (p.relu_below) {
                            v0 = (m0[i][ph] <= 0.f) ? 0.f : v0;
                            v1 = (m1[i][ph] <= 0.f) ? 0.f : v1;
                        }
                        if (w1) *(f2u*)(p.dx + oo) = f2u{v0, v1};
                        else p.dx[oo] = v0;
                    }
                }
            }
        }
        if (p.dbg && threadIdx.x == 0 && (blockIdx.x | blockIdx.y) == 0)
            printf("conv_dgrad_rd wg 0: MFMA loop done %lld, stores drained %lld cycles after start\n", dbg_t1 - dbg_t0, clock64() - dbg_t0);
        cur = nxt;
    }
}

// ---- stride 1 (the north-star shape): dx[ci][h][w] = sum_{co,kx,ky} dy[co][h-kx][w-ky] * w[co][ci][kx][ky] --------------------
// One class, nine taps per dy channel: lane (pixel, kg) loads three 12-byte windows (rows h, h-1, h-2; columns w-2 .. w,
// clamped into the row at the borders) that feed the 9 MFMA steps of MT output tiles; lane (ci, kg) reads the nine taps of
// each tile from the prepared [co][tap][ci] copy (32 consecutive floats per k-slot) or, unprepared, from w itself.
struct __attribute__((packed, aligned(4))) f3u {
    float x, y, z;
};
template <int CO, int MT, int NB, int UC>
__global__ __launch_bounds__(256) void conv_dgrad_rd_s1_kernel(const DgRdParams p) {
    constexpr int CH = CO / 2, G = CH / UC;  // UC dy channels per pipeline group
    static_assert(G % NB == 0, "ring index must line up from tile to tile");
    const int lane = threadIdx.x & 63, n = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ci0 = blockIdx.y * 32 * MT;
    unsigned wl[MT], wl_tr[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int c = ci0 + mt * 32 + n < p.Ci ? ci0 + mt * 32 + n : ci0;
        wl[mt] = (unsigned)((kg * CH * p.Ci + c) * 9);
        wl_tr[mt] = (unsigned)(kg * CH * p.Ci * 9 + c);
    }
    const unsigned plane = (unsigned)(p.Ho * p.Wo);
    const int tstep = gridDim.x * 4;
    struct Loc {
        unsigned o[3];     // element offsets into dy (channel kg*CH): rows h, h-1, h-2 (clamped), window start column
        bool isy[3], isz[3], ok[3][3];  // column w-ky is element y / z of the window; D[kx][ky] exists
        unsigned xo;       // element offset of dx[b][ci0][h][w]
        bool live;
    };
    auto locate = [&](int tile, Loc& L) {
        const int pi = tile * 32 + n;
        L.live = pi < p.pixels;
        const int pic = L.live ? pi : p.pixels - 1;
        const int b = fdiv(pic, p.m_uv, p.UV), rem = pic - b * p.UV;  // UV = H*W, V = W here
        const int h = fdiv(rem, p.m_v, p.V), w = rem - h * p.V;
        int cs = w - 2;
        cs = cs < 0 ? 0 : (cs > p.Wo - 3 ? p.Wo - 3 : cs);
        bool cok[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int e = w - ky - cs;  // element of the window that holds column w - ky
            cok[ky] = w - ky >= 0 && w - ky < p.Wo;
            L.isy[ky] = e == 1;
            L.isz[ky] = e == 2;
        }
        const unsigned cb = (unsigned)((b * CO + kg * CH) * (int)plane);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int r = h - kx;
            const bool rok = L.live && r >= 0 && r < p.Ho;
            const int rc = r < 0 ? 0 : (r >= p.Ho ? p.Ho - 1 : r);
            L.o[kx] = cb + (unsigned)(rc * p.Wo + cs);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) L.ok[kx][ky] = rok && cok[ky];
        }
        L.xo = (unsigned)((b * p.Ci + ci0) * p.UV + rem);
    };
    struct Grp {
        f3u win[UC][3];
        float a[UC][MT][9];
    };
    auto load_group = [&](Grp& g, int gi, const unsigned (&o)[3]) {
#pragma unroll
        for (int u = 0; u < UC; ++u) {
            const int co_i = gi * UC + u;
            const float* base = p.dy + (size_t)co_i * plane;  // wave-uniform
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) g.win[u][kx] = *(const f3u*)(base + o[kx]);
            const float* wb = p.w + (size_t)co_i * p.Ci * 9;  // wave-uniform
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if (p.tr) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) g.a[u][mt][k] = wb[wl_tr[mt] + k * p.Ci];
                } else {
                    const f4u qa = *(const f4u*)(wb + wl[mt]), qb = *(const f4u*)(wb + wl[mt] + 4);
                    g.a[u][mt][0] = qa.x; g.a[u][mt][1] = qa.y; g.a[u][mt][2] = qa.z; g.a[u][mt][3] = qa.w;
                    g.a[u][mt][4] = qb.x; g.a[u][mt][5] = qb.y; g.a[u][mt][6] = qb.z; g.a[u][mt][7] = qb.w;
                    g.a[u][mt][8] = wb[wl[mt] + 8];
                }
            }
        }
    };

    int tile = blockIdx.x * 4 + wave;
    if (tile >= p.tiles) return;
    Loc cur, nxt;
    locate(tile, cur);
    nxt = cur;
    Grp ring[NB];
    {
        if (tile + tstep < p.tiles) locate(tile + tstep, nxt);
#pragma unroll
        for (int g = 0; g < NB - 1; ++g) load_group(ring[g], g, cur.o);
    }
    for (; tile < p.tiles; tile += tstep) {
        const bool more = tile + tstep < p.tiles;
        if (more) locate(tile + tstep, nxt);
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
#pragma unroll 1
        for (int g0 = 0; g0 < G; g0 += NB) {
#pragma unroll
            for (int ri = 0; ri < NB; ++ri) {
                const int g = g0 + ri, gp = g + NB - 1, rp = (ri + NB - 1) % NB;
                const bool wrap = gp >= G;
                const unsigned oo[3] = {wrap ? nxt.o[0] : cur.o[0], wrap ? nxt.o[1] : cur.o[1], wrap ? nxt.o[2] : cur.o[2]};
                load_group(ring[rp], wrap ? gp - G : gp, oo);
                RD_PIPE_FENCE(ring[ri].a[0][0][0]);
                (void)g;
#pragma unroll
                for (int u = 0; u < UC; ++u)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const f3u v = ring[ri].win[u][kx];
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky) {
                            float d = cur.isz[ky] ? v.z : (cur.isy[ky] ? v.y : v.x);
                            d = cur.ok[kx][ky] ? d : 0.f;
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
                                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[ri].a[u][mt][kx * 3 + ky], d, acc[mt], 0, 0, 0);
                        }
                    }
            }
        }
        if (cur.live) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float m[16];
                if (p.relu_below) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rowl = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                        m[r] = p.relu_below[(size_t)cur.xo + (size_t)(ci0 + rowl < p.Ci ? rowl : 0) * p.UV];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rowl = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    if (ci0 + rowl >= p.Ci) continue;
                    float v = acc[mt][r];
                    if (p.relu_below) v = (m[r] <= 0.f) ? 0.f : v;
                    p.dx[(size_t)cur.xo + (size_t)rowl * p.UV] = v;
                }
            }
        }
        cur = nxt;
    }
}

// ---- Ci = 16 (conv_layer_2 of the reference net): the WHOLE filter bank as MFMA A operands in registers -----------------
// v_mfma_f32_16x16x4_f32: M = 16 input channels (one slice of Ci per wave), N = 16 consecutive grid pixels, K = 4 dy channels
// of ONE filter tap: step (c4, tap) multiplies A = w[4 c4 + k][ci][tap] with B = D_tap of channel 4 c4 + k, where the tap picks
// the class it belongs to ((0,0): taps 0 2 6 8 <- D00 D01 D10 D11 | (0,1): 1 7 <- D00 D10 | (1,0): 3 5 <- D00 D01 | (1,1): 4 <- D00)
// = 2.25 CO steps per 16 pixels with no zero padding in M, N or K.  The 9*CO*16 filter values of a slice are exactly 2.25*CO
// lane registers (72 for CO = 32, 144 for 64) that a wave loads ONCE -- no LDS and no filter traffic in the loop.  The only
// streamed operand is dy: per 4 channels (9 steps) a lane loads TWO 8-byte pairs (rows u and u-1, columns v-1 | v) of its
// channel; rows outside the image read 0 through an out-of-range buffer offset, border columns shift the pair and pick.
#ifndef CNN_M16_DGRAD_NB
#define CNN_M16_DGRAD_NB 4
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kM16OOB = 0x7ffffffcu;

// KS > 1 (Co = KS*CO): the dy channels are split over KS waves of a workgroup (NW = slices*KS waves: one pixel partition per
// workgroup); wave `half` 1.. hands its partial sums to wave 0 of the same slice through LDS (one barrier per pixel group).
// PD (pad 1, the stage entries of the ResNet-shaped stack): the same arithmetic on the zero-padded image -- grid position (u, v) owns
// padded pixels (2u | 2u+1, 2v | 2v+1) = real pixels (2u-1 | 2u, 2v-1 | 2v); dy and the tap classes are untouched, only where dx
// (and the fused ReLU mask) lives moves by one row and one column: pairs lose their 8-byte alignment, so four 4-byte stores per
// channel instead of two 8-byte ones, each with its own in-image test.
template <int CO, int NW, bool PREP, int KS, bool PD = false>
__global__ __launch_bounds__(NW * 64) void conv_dgrad_m16_s2_kernel(const DgRdParams p) {
    constexpr int C4 = CO / 4, NA = CO * 9 / 4;
    constexpr int NB = CNN_M16_DGRAD_NB;  // ring of granules (4 dy channels: two 8-byte loads, 9 MFMA steps); NB - 1 granules in flight
    static_assert(C4 % NB == 0, "static ring indices");
    const int lane = threadIdx.x & 63;
    const int n = lane & 15, k = lane >> 4;
    // Ci = 16*slices: a wave owns one 16-channel slice; the slices of a pixel group sit side by side in a workgroup (their dy
    // reads hit L1)
    const int slices = p.Ci >> 4;
    const int wid = xcd_swizzle(blockIdx.x, gridDim.x) * NW + (threadIdx.x >> 6);
    const int slice = wid % slices, half = KS > 1 ? (wid / slices) % KS : 0, wave_id = wid / (slices * KS);
    const int nwaves = p.tiles;  // (tiles: pixel partitions, set by the host)
    const int groups = (p.pixels + 15) >> 4;
    if (wave_id >= nwaves || wave_id >= groups) return;
    float wa[NA];  // [c4*9 + tap]
    if constexpr (PREP) {
#pragma unroll
        for (int j = 0; j < NA; ++j) wa[j] = p.w[((half * slices + slice) * NA + j) * 64 + lane];
    } else {
#pragma unroll
        for (int j = 0; j < NA; ++j) wa[j] = p.w[half * CO * p.Ci * 9 + m16_filter_index(j, lane, p.Ci, slice)];
    }
    const int plane = p.Ho * p.Wo;
    const size_t hw = (size_t)p.H * p.W;
    const unsigned chs = (unsigned)hw * 4u;
    const bool odd = (p.W & 1) != 0;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)((unsigned)p.B * CO * KS * plane * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.dx, 0, (int)((unsigned)p.B * p.Ci * (unsigned)hw * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t mrs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.relu_below ? p.relu_below : p.dx), 0, (int)((unsigned)p.B * p.Ci * (unsigned)hw * 4u), 0x00020000);
    struct Loc {
        unsigned o0, o1;  // dy byte offsets of this lane's pair in rows u, u-1 (channel k of a granule), or out of range: reads 0
        bool ca, cb;      // column case: a: pair = (v-1, v) | b: v = 0, pair = (0, 1) | neither: v = Wo, pair = (Wo-2, Wo-1)
        unsigned x0, x1;  // dx byte offsets of rows 2u, 2u+1 at column 2v, channel 4k (or out of range: no store)
        bool w1;          // column 2v+1 exists (false only in the last column of an odd W)
        unsigned xq[2][2];  // PD: byte offset of dx[row 2u-1+ph][column 2v-1+pw] or out of range
    };
    auto locate = [&](int g, Loc& L) {
        const int pix = g * 16 + n;
        const bool live = g < groups && pix < p.pixels;
        const int pp = live ? pix : 0;
        const int b = fdiv(pp, p.m_uv, p.UV);
        const int rem = pp - b * p.UV;
        const int u = fdiv(rem, p.m_v, p.V);
        const int v = rem - u * p.V;
        L.cb = v == 0;
        L.ca = !L.cb && v < p.Wo;
        const int cs = L.ca ? v - 1 : (L.cb ? 0 : p.Wo - 2);  // (v <= Wo always: V = Wo + 1)
        const unsigned base = (unsigned)((b * CO * KS + half * CO + k) * plane + u * p.Wo + cs) * 4u;
        L.o0 = (live && u < p.Ho) ? base : kM16OOB;
        L.o1 = (live && u >= 1) ? base - (unsigned)p.Wo * 4u : kM16OOB;
        if constexpr (PD) {
            const int y0 = 2 * u - 1, c0 = 2 * v - 1;
            const int xb = (int)((((size_t)b * p.Ci + 16 * slice + 4 * k) * hw) * 4u) + (y0 * p.W + c0) * 4;  // (may be "negative": only used when valid)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                for (int pw = 0; pw < 2; ++pw) {
                    const bool ok = live && (unsigned)(y0 + ph) < (unsigned)p.H && (unsigned)(c0 + pw) < (unsigned)p.W;
                    L.xq[ph][pw] = ok ? (unsigned)(xb + (ph * p.W + pw) * 4) : kM16OOB;
                }
            L.x0 = L.x1 = kM16OOB;
            L.w1 = true;
        } else {
            const unsigned xb = (unsigned)(((size_t)b * p.Ci + 16 * slice + 4 * k) * hw + (size_t)(2 * u) * p.W + 2 * v) * 4u;
            L.x0 = live ? xb : kM16OOB;
            L.x1 = (live && 2 * u + 1 < p.H) ? xb + (unsigned)p.W * 4u : kM16OOB;
            L.w1 = 2 * v + 1 < p.W;
        }
    };
    auto load_g = [&](v2f (&buf)[2], int c4, const Loc& L) {
        const int so = c4 * 4 * plane * 4;
        buf[0] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)L.o0, so, 0));
        buf[1] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)L.o1, so, 0));
    };
    Loc cur, nxt;
    locate(wave_id, cur);
    v2f ring[NB][2];
#pragma unroll
    for (int i = 0; i < NB - 1; ++i) load_g(ring[i], i, cur);
    for (int g = wave_id; g < groups; g += nwaves) {
        locate(g + nwaves, nxt);  // (behind the last group: every offset out of range)
        // the fused ReLU::backward mask (relu.cpp:38) of this group's 8 output pairs is requested now and used in the epilogue
        // (a lane in the last column of an odd W reads the pair one element to the left and uses its second half)
        v2f mk[4][2];
        if constexpr (PD) {
            if (p.relu_below && half == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int ph = 0; ph < 2; ++ph) {
                        mk[r][ph].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mrs, (int)cur.xq[ph][0], (int)(r * chs), 0));
                        mk[r][ph].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mrs, (int)cur.xq[ph][1], (int)(r * chs), 0));
                    }
            }
        } else
        if (p.relu_below && half == 0) {
            const unsigned m0 = (cur.w1 || cur.x0 == kM16OOB) ? cur.x0 : cur.x0 - 4u, m1 = (cur.w1 || cur.x1 == kM16OOB) ? cur.x1 : cur.x1 - 4u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mk[r][0] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(mrs, (int)m0, (int)(r * chs), 0));
                mk[r][1] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(mrs, (int)m1, (int)(r * chs), 0));
            }
        }
        f32x4 acc[4];  // classes (ph,pw) = 00, 01, 10, 11; register r of lane (n, k) = channel 4k + r of pixel n
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c4 = 0; c4 < C4; ++c4) {
            // request granule c4+3 (from c4 = C4-3 on: the first three of the next group), then consume granule c4
            const int gn = c4 + NB - 1;
            load_g(ring[gn % NB], gn % C4, gn < C4 ? cur : nxt);
            RD_PIPE_FENCE(ring[c4 % NB][0]);
            const v2f r0 = ring[c4 % NB][0], r1 = ring[c4 % NB][1];
            const float d00 = cur.ca ? r0.y : (cur.cb ? r0.x : 0.f), d01 = cur.ca ? r0.x : (cur.cb ? 0.f : r0.y);  // D[0][jc]
            const float d10 = cur.ca ? r1.y : (cur.cb ? r1.x : 0.f), d11 = cur.ca ? r1.x : (cur.cb ? 0.f : r1.y);  // D[1][jc]
            const float* a = &wa[c4 * 9];
            // issue order pinned: an accumulator is reused every other step at the earliest (a dependent MFMA waits 40 cycles,
            // an independent one issues after 32; hipcc's own order put the four class-(0,0) steps back to back)
#define M16_STEP(ACC, A_, B_)                                              \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(A_, B_, ACC, 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0)
            M16_STEP(acc[1], a[1], d00);
            M16_STEP(acc[0], a[0], d00);
            M16_STEP(acc[2], a[3], d00);
            M16_STEP(acc[0], a[2], d01);
            M16_STEP(acc[3], a[4], d00);
            M16_STEP(acc[0], a[6], d10);
            M16_STEP(acc[1], a[7], d10);
            M16_STEP(acc[0], a[8], d11);
            M16_STEP(acc[2], a[5], d01);
#undef M16_STEP
        }
        if constexpr (KS > 1) {
            __shared__ float red[2][KS - 1][NW / KS][16][64];  // two buffers, alternating per pixel group: one barrier per group
            const int rb = ((g - wave_id) / nwaves) & 1;
            if (half > 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[rb][half - 1][slice][c * 4 + r][lane] = acc[c][r];
            }
            __syncthreads();
            if (half > 0) {
                cur = nxt;
                continue;
            }
#pragma unroll
            for (int h = 0; h < KS - 1; ++h)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[c][r] += red[rb][h][slice][c * 4 + r][lane];
        }
        if constexpr (PD) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    float v0 = acc[ph * 2][r], v1 = acc[ph * 2 + 1][r];
                    if (p.relu_below) {
                        v0 = (mk[r][ph].x <= 0.f) ? 0.f : v0;
                        v1 = (mk[r][ph].y <= 0.f) ? 0.f : v1;
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), xrs, (int)cur.xq[ph][0], (int)(r * chs), 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), xrs, (int)cur.xq[ph][1], (int)(r * chs), 0);
                }
            cur = nxt;
            continue;
        }
        // epilogue, branch-free per lane: buffer stores whose offset is out of range for dead lanes / the row behind the tensor
        const unsigned pp0 = cur.w1 ? cur.x0 : kM16OOB, pp1 = cur.w1 ? cur.x1 : kM16OOB;  // (pw = 0,1) pairs
        const unsigned ss0 = cur.w1 ? kM16OOB : cur.x0, ss1 = cur.w1 ? kM16OOB : cur.x1;  // single element, last column of an odd W
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                float v0 = acc[ph * 2][r], v1 = acc[ph * 2 + 1][r];
                if (p.relu_below) {
                    const float k0 = cur.w1 ? mk[r][ph].x : mk[r][ph].y;
                    v0 = (k0 <= 0.f) ? 0.f : v0;
                    v1 = (mk[r][ph].y <= 0.f) ? 0.f : v1;
                }
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v2f{v0, v1}), xrs, (int)(ph ? pp1 : pp0), (int)(r * chs), 0);
                if (odd) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), xrs, (int)(ph ? ss1 : ss0), (int)(r * chs), 0);
            }
        cur = nxt;
    }
}

inline unsigned magic_of(int d) { return (unsigned)((1ull << 32) / (unsigned)d) + 1u; }

struct DgRdPlan {
    DgRdParams p;
    int co, nw, cgroups, blocks_x, mt;
    bool m16;  // Ci = 16: conv_dgrad_m16_s2_kernel (filters from the reference layout: the prepared copy is verbatim)
    size_t lds, img_floats;
};

inline bool m16_wanted(const cnn_conv2d_desc* d) {
    // (Ci, Co) = (16, 32) | (32, 64) | (64, 128: dy channels split over two waves): 72 | 144 | 144 filter registers per wave
    if (d->s != 2 || !((d->Ci == 16 && d->Co == 32) || (d->Ci == 32 && d->Co == 64) || (d->Ci == 64 && d->Co == 128))) return false;
    if (d->pad != 0 && !(d->pad == 1 && d->Ci == 64 && d->Co == 128)) return false;  // (pad 1: only the instance the ResNet-shaped stack needs)
    if ((long long)d->B * d->Co * cnn_conv2d_out_dim(d->H, 3, 2, d->pad) * cnn_conv2d_out_dim(d->W, 3, 2, d->pad) >= (1ll << 29) ||
        (long long)d->B * d->Ci * d->H * d->W >= (1ll << 29))
        return false;  // (32-bit buffer offsets)
    const OptVal e = CNN_OPT_VAL("DGRAD_M16");
    if (e && atoi(e) == 0) return false;
    if (d->Ci >= 32 && e && atoi(e) == 1) return false;  // (=1: only the Ci = 16 shape, =2: not the split Co = 128 shape; for A/B runs)
    if (d->Ci == 64 && e && atoi(e) == 2) return false;
    return true;
}

bool make_plan(const cnn_conv2d_desc* d, DgRdPlan* pl) {
    if (d->k != 3 || (d->s != 2 && d->s != 1)) return false;
    if (d->pad != 0 && !(d->pad == 1 && d->s == 2 && m16_wanted(d))) return false;
    if (d->s == 1 && ((d->Co != 64 && d->Co != 128) || d->Ci % 32 != 0)) return false;
    if ((d->Co != 32 && d->Co != 64 && d->Co != 128) || d->Ci % 16 != 0) return false;  // (Ci = 16: half of the 32 MFMA rows idle)
    // Co = 32 / Ci = 16 (conv_layer_2): measured 117 us against 95 us for the packed VALU kernel -> opt-in only
    pl->m16 = m16_wanted(d);
    if (d->Co == 32 && !pl->m16 && !((CNN_OPT_SET("DGRAD_RD32") && CNN_OPT_INT("DGRAD_RD32", 0) != 0))) return false;
    if (const OptVal e = CNN_OPT_VAL("DGRAD_RD"))
        if (atoi(e) == 0) return false;
    DgRdParams& p = pl->p;
    p.B = d->B; p.Ci = d->Ci; p.H = d->H; p.W = d->W;
    p.Ho = cnn_conv2d_out_dim(d->H, 3, d->s, d->pad);
    p.Wo = cnn_conv2d_out_dim(d->W, 3, d->s, d->pad);
    if (p.Ho < 1 || p.Wo < (d->s == 1 ? 3 : 2)) return false;
    if (d->s == 1) { p.U = d->H; p.V = d->W; }          // every dx pixel is a grid pixel
    else { p.U = (d->H + 2 * d->pad + 1) / 2; p.V = (d->W + 2 * d->pad + 1) / 2; }  // (pad 1: the grid of the zero-padded image)
    p.UV = p.U * p.V;
    const long long pixels = (long long)d->B * p.UV;
    if (pixels >= (1ll << 30) || (long long)d->B * d->Ci * d->H * d->W >= (1ll << 31) || (long long)d->B * d->Co * p.Ho * p.Wo >= (1ll << 31))
        return false;
    p.pixels = (int)pixels;
    p.tiles = (int)((pixels + 31) / 32);
    p.m_uv = magic_of(p.UV);
    p.m_v = magic_of(p.V);
    p.dbg = CNN_MEASURE_INT("DGRAD_RD_DBG", 0);
    pl->co = d->Co;
    pl->mt = (d->s == 1 && d->Ci % 64 == 0) ? 2 : 1;  // stride 1: two 32-channel tiles per wave share the dy windows
    if (const OptVal e = CNN_OPT_VAL("DGRAD_RD_MT")) pl->mt = (atoi(e) == 2 && d->s == 1 && d->Ci % 64 == 0) ? 2 : 1;
    pl->cgroups = (d->Ci + 32 * pl->mt - 1) / (32 * pl->mt);
    pl->img_floats = (size_t)d->Co * d->Ci * 9;  // (prepared buffer = a verbatim copy of w)
    pl->lds = 0;
    pl->nw = 4;
    const long long bx = 2 * num_cus() / pl->cgroups;
    const long long need = (p.tiles + pl->nw - 1) / pl->nw;
    pl->blocks_x = (int)(bx < 1 ? 1 : (bx > need ? need : bx));
    if (pl->m16) {  // persistent waves over 16-pixel groups
        int per_cu = CNN_OPT_INT("DGRAD_M16_WG", 2);
        if (per_cu < 1 || per_cu > 8) per_cu = 2;
        const int slices = d->Ci / 16;
        const long long g = (pixels + 15) / 16;
        long long parts = (long long)per_cu * num_cus() * pl->nw / slices;  // pixel partitions: waves / slices
        if (d->Co == 128) parts = (CNN_OPT_SET("DGRAD_M16_WG") ? per_cu : 1) * num_cus();  // one 8-wave workgroup (4 slices x 2 halves) per partition
        if (parts > g) parts = g;
        p.tiles = (int)parts;
        pl->blocks_x = d->Co == 128 ? (int)parts : (int)((parts * slices + pl->nw - 1) / pl->nw);
    }
    // the stride-1 kernel only: cnn_conv2d_autotune may have measured the implicit GEMM faster for this geometry
    if (!pl->m16 && d->s == 1 && cnn_amd::igemm_preferred(d, 1)) return false;
    return true;
}

template <int CO, int NW>
int launch(const DgRdPlan& pl, hipStream_t s, const char* name, const cnn_conv2d_desc* d) {
    const int var = CNN_OPT_INT("DGRAD_RD_VAR", 0);
    auto kern = var == 3 ? conv_dgrad_rd_s2_kernel<CO, NW, 2, 2> : conv_dgrad_rd_s2_kernel<CO, NW, 2, 4>;
    const dim3 grid(pl.blocks_x, pl.cgroups);
    CNN_KLAUNCH(s, name, (kern<<<grid, NW * 64, pl.lds, s>>>(pl.p)), "B%d Ci%d %dx%d Co%d k%d s%d p%d", d->B, d->Ci, d->H, d->W,
                d->Co, d->k, d->s, d->pad);
    return CNN_AMD_OK;
}

inline int prepared_transposed() { return CNN_OPT_SET("DGRAD_RD_NOTR") ? 0 : 1; }

}  // namespace

namespace cnn_amd {

bool dgrad_rd_supported(const cnn_conv2d_desc* d) {
    DgRdPlan pl;
    return make_plan(d, &pl);
}

size_t dgrad_rd_prepared_floats(const cnn_conv2d_desc* d) {
    DgRdPlan pl;
    return make_plan(d, &pl) ? pl.img_floats : 0;
}

// prepared buffer of layer d: a re-ordered copy of the filters, tr = 1: img[(co*9 + tap)*Ci + ci] = w[(co*Ci + ci)*9 + tap],
// tr = 0: verbatim.  The copy itself runs inside conv_fwd_rd.hip's prepare kernel (one launch for both kernel families).
// returns 0 when the layer is not covered
int dgrad_rd_prepare_layout(const cnn_conv2d_desc* d, int* transposed) {
    DgRdPlan pl;
    if (!make_plan(d, &pl)) return 0;
    *transposed = pl.m16 ? 2 : prepared_transposed();
    return 1;
}

// [co][ci][tap] -> [co][tap][ci] into the caller's workspace (the unprepared entry points)
__global__ __launch_bounds__(256) void dgrad_rd_transpose_kernel(const float* __restrict__ w, float* __restrict__ img, int Co, int Ci) {
    const int total = Co * Ci * 9;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int ci = i % Ci, r = i / Ci, tap = r % 9, co = r / 9;
        img[i] = w[((size_t)co * Ci + ci) * 9 + tap];
    }
}

// w == nullptr: `img` holds the prepared filters; otherwise `ws` (>= Co*Ci*9 floats, may be null) receives the transposed copy
int dgrad_rd_backward_data(const cnn_conv2d_desc* d, const float* dy, const float* w, const float* img, const float* relu_below,
                           float* dx, void* ws, size_t ws_bytes, hipStream_t s) {
    DgRdPlan pl;
    if (!make_plan(d, &pl)) return fail(CNN_AMD_E_BADARG, "conv_dgrad_rd: geometry not covered");
    pl.p.dy = dy; pl.p.w = w ? w : img; pl.p.tr = w ? 0 : prepared_transposed(); pl.p.relu_below = relu_below; pl.p.dx = dx;
    if (pl.m16) {
        pl.p.tr = w ? 0 : 2;  // prepared: lane-major operand order (m16_filter_index); otherwise gathered from the reference layout
        char nm[64];
        snprintf(nm, sizeof(nm), "conv_dgrad_rd<2,%d,m16>/dgrad%s", d->Co, relu_below ? "+relu" : "");
#define M16(CO_, PREP_, NW_, KS_)                                                                                                        \
    CNN_KLAUNCH(s, nm, (launch_pub(conv_dgrad_m16_s2_kernel<CO_, NW_, PREP_, KS_>, dim3(pl.blocks_x), dim3(NW_ * 64), 0, s, pl.p)), "B%d Ci%d %dx%d Co%d k%d s%d p%d", \
                d->B, d->Ci, d->H, d->W, d->Co, d->k, d->s, d->pad)
        if (d->Co == 32) { if (pl.p.tr == 2) M16(32, true, 4, 1); else M16(32, false, 4, 1); }
        else if (d->Co == 64) { if (pl.p.tr == 2) M16(64, true, 4, 1); else M16(64, false, 4, 1); }
        else if (d->pad == 1) {
#define M16P(PREP_)                                                                                                                        \
    CNN_KLAUNCH(s, nm, (launch_pub(conv_dgrad_m16_s2_kernel<64, 8, PREP_, 2, true>, dim3(pl.blocks_x), dim3(8 * 64), 0, s, pl.p)), "B%d Ci%d %dx%d Co%d k%d s%d p%d", \
                d->B, d->Ci, d->H, d->W, d->Co, d->k, d->s, d->pad)
            if (pl.p.tr == 2) M16P(true); else M16P(false);
#undef M16P
        }
        else { if (pl.p.tr == 2) M16(64, true, 8, 2); else M16(64, false, 8, 2); }
#undef M16
        return CNN_AMD_OK;
    }
    if (w && ws && ws_bytes >= pl.img_floats * sizeof(float) && prepared_transposed()) {
        const unsigned gx = (unsigned)((pl.img_floats + 255) / 256);
        CNN_KLAUNCH(s, "dgrad_rd_transpose", (dgrad_rd_transpose_kernel<<<gx > 512 ? 512 : gx, 256, 0, s>>>(w, (float*)ws, d->Co, d->Ci)),
                    "B%d Ci%d %dx%d Co%d k%d s%d p%d", d->B, d->Ci, d->H, d->W, d->Co, d->k, d->s, d->pad);
        pl.p.w = (const float*)ws;
        pl.p.tr = 1;
    }
    char name[64];
    snprintf(name, sizeof(name), "conv_dgrad_rd<2,%d>/dgrad%s", d->Co, relu_below ? "+relu" : "");
    if (d->s == 1) {
        const dim3 grid(pl.blocks_x, pl.cgroups);
        snprintf(name, sizeof(name), "conv_dgrad_rd<1,%d,%d>/dgrad%s", d->Co, pl.mt, relu_below ? "+relu" : "");
#define S1(CO_, MT_, UC_) CNN_KLAUNCH(s, name, (conv_dgrad_rd_s1_kernel<CO_, MT_, 4, UC_><<<grid, 256, 0, s>>>(pl.p)), "B%d Ci%d %dx%d Co%d k%d s%d p%d", \
                                      d->B, d->Ci, d->H, d->W, d->Co, d->k, d->s, d->pad)
        const int uc = CNN_OPT_INT("DGRAD_RD_UC", 1);
        if (d->Co == 64 && pl.mt == 2) S1(64, 2, 1);
        else if (d->Co == 64) { if (uc == 2) S1(64, 1, 2); else S1(64, 1, 1); }
        else if (pl.mt == 2) S1(128, 2, 1);
        else { if (uc == 2) S1(128, 1, 2); else S1(128, 1, 1); }
#undef S1
        return CNN_AMD_OK;
    }
    if (d->Ci == 16) {  // two classes per MFMA tile
        const dim3 grid(pl.blocks_x, pl.cgroups);
#define SH(CO_) CNN_KLAUNCH(s, name, (conv_dgrad_rd_s2_kernel<CO_, 4, 2, 4, true><<<grid, 256, 0, s>>>(pl.p)), "B%d Ci%d %dx%d Co%d k%d s%d p%d", \
                            d->B, d->Ci, d->H, d->W, d->Co, d->k, d->s, d->pad)
        if (d->Co == 32) SH(32); else if (d->Co == 64) SH(64); else SH(128);
#undef SH
        return CNN_AMD_OK;
    }
    if (d->Co == 32) return launch<32, 4>(pl, s, name, d);
    if (d->Co == 64) return launch<64, 4>(pl, s, name, d);
    return launch<128, 4>(pl, s, name, d);
}

}  // namespace cnn_amd
