// conv_rows_s2.hip -- Conv2D forward (cpu/src/conv2d.cpp:69-92) and data gradient (conv2d.cpp:168-199) of 3x3 / STRIDE-2 layers -- the
// reference's default stride (architectures.h:69; the centre walk x += stride at conv2d.cpp:76-77, 183-184) -- as the LDS-staged sibling
// of conv_rows.hip (round 6): both operands staged by buffer-addressed LDS DMA with out-of-range zero fill, v_mfma_f32_16x16x4_f32 with
// the PIXELS as the M operand, in-place AGPR accumulators, two buffers, one barrier per stage, a workgroup walking a range of units.
//
// What stride 2 changes:
//   * FLAT PIXEL PACKING.  An output row is 27 / 13 / 6 (reference net) or 28 / 14 / 7 (ResNet-shaped stage entries) pixels: per-row 16-pixel
//     blocks would idle 16 - 60 % of the lanes.  A unit's pixels -- RPU output rows x all columns (x PK whole samples for the smallest
//     planes) -- are numbered flat, f = (sample * RPU + row) * DW + col, and cut into 16-pixel blocks; lane n of block nb keeps the
//     staged-plane offset of ITS pixel in a register (boff[nb], computed once per kernel: every unit of a launch has the same shape), so
//     an operand read is still one per-lane base + a compile-time immediate (tap and channel group).
//   * ROWS ARE STAGED ONE BY ONE: a staged plane is [rows][RP] with RP = the row length rounded up to 16 bytes; the 16-byte DMA units of a
//     row start at the row (the global source needs 4-byte alignment only: tools/probes/buflds16_probe.cpp).  Any staged row that lies
//     outside the image -- the halo row above a pad-1 plane, the dy row behind the last one in the data gradient -- is moved as ZEROS
//     (lanes out of the descriptor's range): no row masks, no alignment rules between row length and unit size, any width.
//   * FORWARD reads its B operand at stride 2 (x[ci][2r + kx - p][2c + ky - p]): sixteen lanes of one k-group touch every second bank.
//     The planes of ODD channels are staged one float to the right (their source starts one float early), so the two k-groups of a
//     32-lane LDS access sit on banks of different parity: conflict-free like the stride-1 reads of conv_rows.hip.
//   * DATA GRADIENT = the four parity classes of the transposed convolution, no zero insertion: with y = 2 yy + py,
//         dx[ci][y][x] = sum_{co} sum_{kx: (py + p - kx) even} sum_{ky: (px + p - ky) even} w[co][ci][kx][ky] * dy[co][yy + (py + p - kx)/2][xx + (px + p - ky)/2]
//     -- every tap belongs to exactly one class (4 + 2 + 2 + 1 taps).  A workgroup owns a block of (yy, xx) positions and ALL FOUR classes
//     (four accumulator sets): the B operand is dy at (yy + da, xx + db), da, db in {0, -1} (pad 0) or {0, +1} (pad 1) -- FOUR stride-1
//     LDS reads feed the NINE MFMAs of a channel group and pixel block -- and the epilogue interleaves the classes px = 0 / 1 of a lane's
//     pixels into consecutive dx floats.  Rows / columns no window covers (conv2d.cpp:168: zero-filled) come out as sums over staged zeros.
// GEMM view: M = pixels (MFMA src A), N = output channels of the pass (16 per wave; MT = 16 x WM per workgroup), K = (channel, tap).
// Waves: WM over channels x WP = 4 / WM over the unit's pixel blocks.
#include <cstdlib>

#include "common.h"
#include "rows_common.h"

using namespace cnn_amd;

namespace {

struct S2Params {
    const float* x;     // staged tensor: forward x [B][C][HI][WI]; data gradient dy [B][C][HO][WO]
    const float* wt;    // prepared filters [channel tile][chunk][8][9][QW] (conv_rows.hip rows_prep, mode 0 / 2)
    const float* bias;  // nullable (data gradient)
    float* y;           // forward: y [B][M][HO][WO] (nullable when y_relu is given); data gradient: dx [B][M][HI][WI]
    float* y_relu;      // forward, nullable: the output of the ReLU layer behind this one (relu.cpp:25)
    const float* relu_below;  // data gradient, nullable: output of the ReLU layer in front -- its backward pass (relu.cpp:37) on the way out
    int B, C, M;        // C = reduction channels, M = output channels of the pass
    int nchunk;         // C / 8
    int units_total, units_per_block;
    int dbg;
};

template <int MODE, int HI, int WI, int PAD, int WM, int RPU, int PK>
struct S2Geom {
    static_assert(MODE == 0 || MODE == 1, "0 forward, 1 data gradient");
    static_assert(PAD == 0 || PAD == 1, "padding");
    static_assert(WM == 1 || WM == 2 || WM == 4, "waves over channels");
    static constexpr int CK = 8, KSTEPS = 2;
    static constexpr int HO = (HI + 2 * PAD - 3) / 2 + 1, WO = (WI + 2 * PAD - 3) / 2 + 1;
    static constexpr int WP = 4 / WM, MT = 16 * WM;
    static constexpr int SH = MODE == 0 ? HI : HO, SW = MODE == 0 ? WI : WO;                 // staged planes
    static constexpr int DH = MODE == 0 ? HO : (HI + 1) / 2, DW = MODE == 0 ? WO : (WI + 1) / 2;  // pixel domain
    static constexpr int OH = MODE == 0 ? HO : HI, OW = MODE == 0 ? WO : WI;                 // output planes
    static_assert(PK == 1 || RPU == DH, "packed samples: whole planes");
    static constexpr int NRB = (DH + RPU - 1) / RPU;          // units per sample (PK == 1)
    static constexpr int PXS = RPU * DW, PX = PK * PXS;       // pixels of a unit: per sample, in all
    static constexpr int NB = (PX + 15) / 16, NBW = (NB + WP - 1) / WP;
    static constexpr int XR = MODE == 0 ? 2 * RPU + 1 : RPU + 1;  // staged rows per unit and sample
    static constexpr int NEG = MODE == 0 ? PAD : 1 - PAD;     // staged rows that may lie above the image: first staged row = ROWMUL * r0 - NEG
    static constexpr int ROWMUL = MODE == 0 ? 2 : 1;
    static constexpr int SKEW = MODE == 0 ? 1 : 0;            // forward: planes of odd channels one float to the right
    static constexpr int RP = (SW + SKEW + 3) / 4 * 4;        // LDS row pitch (floats)
    static constexpr int UPR = RP / 4;                        // 16-byte units per row
    static constexpr int SUBP = XR * RP;                      // a sample's rows
    static constexpr int QXP = stride16(PK * SUBP);           // channel plane stride: 16 (mod 32)
    static constexpr int QW = MT % 32 == 16 ? MT : MT + 16;   // filter row stride: 9 * QW = 16 (mod 32)
    static_assert(QXP % 32 == 16 && (9 * QW) % 32 == 16, "bank halves");
    static constexpr int XIMG = CK * QXP, WIMG = CK * 9 * QW;
    static constexpr int NIX = (XIMG / 4 + 63) / 64, NIWT = (WIMG / 4 + 63) / 64;  // DMA instructions per stage
    static constexpr int NIWX = (NIX + 3) / 4, NIWW = (NIWT + 3) / 4;              // per wave
    static constexpr int NSLOT = NIWX + NIWW;
    static constexpr int XS = NIX * 256, WS = NIWT * 256;
    static constexpr int BUF = XS + WS;
    static constexpr int DUMP = 2 * BUF;
    static constexpr size_t lds_bytes = (size_t)(2 * BUF + 4 * 256) * sizeof(float);
    static_assert(lds_bytes <= 160 * 1024, "LDS plan");
    static constexpr int BACK = NEG * SW + 4;                 // floats the descriptor starts in front of the tensor
    static constexpr int NCLS = MODE == 0 ? 1 : 4;            // accumulator sets
    // sub-steps of a stage: forward (tap, 4-channel group): 1 A value; data gradient (group, B(da, db)): the taps that read that B
    static constexpr int NKS = MODE == 0 ? 9 * KSTEPS : 4 * KSTEPS;
    // data gradient: the tap rows (columns) whose dy row (column) offset is 0 (j = 0: two of them) or the other one (j = 1: one)
    static constexpr int n_k(int j) { return j == 0 ? 2 : 1; }
    static constexpr int k_of(int j, int i) { return PAD == 0 ? (j == 0 ? i : 2) : (j == 0 ? 1 + i : 0); }
    static constexpr int d_of(int j) { return j == 0 ? 0 : (PAD == 0 ? -1 : 1); }
    static constexpr int cls_of(int k) { return (k + PAD) & 1; }  // parity class of the dx row (column) tap row (column) k feeds
    // pixel f of a unit (f < PX): sample, domain row, column
    static constexpr int f_sp(int f) { return f / PXS; }
    static constexpr int f_r(int f) { return (f % PXS) / DW; }
    static constexpr int f_c(int f) { return f % DW; }
    // lanes n of block gnb whose tap column leaves the staged row: forward column 2c + ky - PAD, data gradient c + db, outside [0, SW)
    static constexpr unsigned colmask(int gnb, int kd) {
        unsigned m = 0;
        for (int n = 0; n < 16; ++n) {
            const int f = 16 * gnb + n;
            if (f >= PX) continue;
            const int c = MODE == 0 ? 2 * f_c(f) + kd - PAD : f_c(f) + kd;
            if (c < 0 || c >= SW) m |= 1u << n;
        }
        return m;
    }
};

template <int MODE, int HI, int WI, int PAD, int WM, int RPU, int PK>
__global__ __launch_bounds__(256) void conv_s2_kernel(const S2Params p) {
    using G = S2Geom<MODE, HI, WI, PAD, WM, RPU, PK>;
    constexpr int NBW = G::NBW, CK = G::CK, WP = G::WP, MT = G::MT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wp = wave / WM;
    const int co0 = blockIdx.y * MT;
    constexpr int SHW = G::SH * G::SW, OHW = G::OH * G::OW;

    const int u_lo = blockIdx.x * p.units_per_block;
    const int u_hi = u_lo + p.units_per_block < p.units_total ? u_lo + p.units_per_block : p.units_total;
    if (u_lo >= u_hi) return;

    // ---- this wave's share of a stage's x DMA, decoded once: 16-byte unit q of the image = (channel plane, sample, staged row, unit of the row)
    unsigned xd_off[G::NIWX];
    int xd_rr[G::NIWX];
#pragma unroll
    for (int i = 0; i < G::NIWX; ++i) {
        const int j = i * 4 + wave, q = j * 64 + lane;
        const int plane = q / (G::QXP / 4), e = q - plane * (G::QXP / 4);
        const int sp = e / (G::XR * G::UPR), rr = (e / G::UPR) % G::XR, u = e % G::UPR;
        const bool have = j < G::NIX && plane < CK && sp < PK;
        // float index from the descriptor's base (BACK floats in front of the tensor), without the stage's (sample, chunk, first row) part
        xd_off[i] = have ? (unsigned)((sp * p.C + plane) * SHW + rr * G::SW + 4 * u - (G::SKEW ? (plane & 1) : 0) + 4) * 4u : kOob;
        xd_rr[i] = rr;
    }
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.x - G::BACK), 0, (int)(((unsigned)p.B * p.C * SHW + G::BACK) * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.wt + (size_t)blockIdx.y * p.nchunk * G::WIMG), 0, (int)((unsigned)p.nchunk * G::WIMG * 4u), 0x00020000);

    float* const dump = smem + G::DUMP + wave * 256;
    // slot k of the DMA of stage (first sample b, first domain row r0, chunk cc) into `buf` = [x image][filter image]
    auto dma_slot = [&](int k, int b, int r0, int cc, float* buf) {
        if (k < G::NIWX) {
            const int j = k * 4 + wave;
            float* d = j < G::NIX ? buf + j * 256 : dump;
            const int row0 = G::ROWMUL * r0 - G::NEG;  // tensor row of staged row 0
            // rows outside the image are moved as zeros (a lane offset outside the descriptor)
            const unsigned voff = (unsigned)(row0 + xd_rr[k]) < (unsigned)G::SH ? xd_off[k] : kOob;
            blds16(xrs, voff, (unsigned)((b * p.C + cc * CK) * SHW + (row0 + G::NEG) * G::SW) * 4u, d);
        } else {
            const int i = k - G::NIWX, j = i * 4 + wave;
            float* d = j < G::NIWT ? buf + G::XS + j * 256 : dump;
            const unsigned q = (unsigned)(j * 64 + lane);
            blds16(wrs, (j < G::NIWT && q * 4 < (unsigned)G::WIMG) ? q * 16u : kOob, (unsigned)cc * (unsigned)(G::WIMG * 4), d);
        }
    };

    // ---- per-lane operand bases (floats inside a buffer): the staged-plane offset of this lane's pixel in each of the wave's blocks
    int boff[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        int f = 16 * (wp * NBW + nb) + n;
        if (f >= G::PX) f = 0;  // (a lane behind the unit's last pixel computes pixel 0 again; never stored)
        const int sp = f / G::PXS, rem = f - sp * G::PXS, r = rem / G::DW, c = rem - r * G::DW;
        if constexpr (MODE == 0) boff[nb] = kq * G::QXP + sp * G::SUBP + 2 * r * G::RP + 2 * c + (kq & 1) - PAD;
        else boff[nb] = kq * G::QXP + sp * G::SUBP + (r + G::NEG) * G::RP + c;
    }
    const int a_base = G::XS + kq * 9 * G::QW + wm * 16 + n;

    f32x4 acc[G::NCLS][NBW];
    auto zero_acc = [&]() {
#pragma unroll
        for (int cl = 0; cl < G::NCLS; ++cl)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) acc[cl][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    // units: (sample group, row block); two nested loops so that the stage loop carries the accumulators and nothing else (conv_rows.hip)
    int b = (u_lo / G::NRB) * PK, r0 = (u_lo % G::NRB) * RPU;
    {
#pragma unroll
        for (int k = 0; k < G::NSLOT; ++k) dma_slot(k, b, r0, 0, smem);
    }
    constexpr int NKS = G::NKS;
    constexpr int PER_KS = (G::NSLOT + NKS - 1) / NKS;
    int t = 0;  // stages so far: buffer parity
    const long long dbg_c0 = p.dbg == 9 ? clock64() : 0, dbg_w0 = p.dbg == 9 ? wall_clock64() : 0;
    for (int u = u_lo; u < u_hi; ++u) {
        int bu = b, r0u = r0;  // the unit behind this one (behind the last one: this one again)
        if (u + 1 < u_hi) {
            if (r0 + RPU < G::DH) r0u = r0 + RPU;
            else { r0u = 0; bu = b + PK; }
        }
        for (int cc = 0; cc < p.nchunk; ++cc, ++t) {
            // (stage 0 of a unit behind the first was waited for in front of the previous unit's stores: conv_rows.hip)
            if (cc != 0 || u == u_lo) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const float* cur = smem + (t & 1) * G::BUF;
            float* nxt = smem + ((t + 1) & 1) * G::BUF;
            const bool last_cc = cc + 1 == p.nchunk;
            const int bn = last_cc ? bu : b, r0n = last_cc ? r0u : r0, ccn = last_cc ? (u + 1 < u_hi ? 0 : cc) : cc + 1;
            struct Ops {
                float a[MODE == 0 ? 1 : 4];
                float b[NBW];
            };
            auto read_ops = [&](Ops& o, int ks) {
                if constexpr (MODE == 0) {
                    const int tap = ks / G::KSTEPS, s = ks % G::KSTEPS, kx = tap / 3, ky = tap % 3;
                    o.a[0] = cur[a_base + (s * 36 + tap) * G::QW];
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) {
                        float bv = cur[boff[nb] + s * 4 * G::QXP + kx * G::RP + ky];
                        unsigned cm = 0;
#pragma unroll
                        for (int w = 0; w < WP; ++w) cm |= G::colmask(w * NBW + nb, ky);  // (is any wave's block touched at all: compile time)
                        if (cm != 0) {
                            unsigned mine = G::colmask(nb, ky);
#pragma unroll
                            for (int w = 1; w < WP; ++w) mine = wp == w ? G::colmask(w * NBW + nb, ky) : mine;
                            bv = ((mine >> n) & 1u) ? 0.f : bv;
                        }
                        o.b[nb] = bv;
                    }
                } else {
                    const int s = ks / 4, j = ks % 4, ja = j >> 1, jb = j & 1;
                    int cnt = 0;
#pragma unroll
                    for (int ia = 0; ia < G::n_k(ja); ++ia)
#pragma unroll
                        for (int ib = 0; ib < G::n_k(jb); ++ib) o.a[cnt++] = cur[a_base + (s * 36 + G::k_of(ja, ia) * 3 + G::k_of(jb, ib)) * G::QW];
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) {
                        float bv = cur[boff[nb] + s * 4 * G::QXP + G::d_of(ja) * G::RP + G::d_of(jb)];
                        unsigned cm = 0;
#pragma unroll
                        for (int w = 0; w < WP; ++w) cm |= G::colmask(w * NBW + nb, G::d_of(jb));
                        if (cm != 0) {
                            unsigned mine = G::colmask(nb, G::d_of(jb));
#pragma unroll
                            for (int w = 1; w < WP; ++w) mine = wp == w ? G::colmask(w * NBW + nb, G::d_of(jb)) : mine;
                            bv = ((mine >> n) & 1u) ? 0.f : bv;
                        }
                        o.b[nb] = bv;
                    }
                }
            };
            auto run_ops = [&](Ops& o, int ks) {
                if constexpr (MODE == 0) {
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) mfma16(acc[0][nb], o.b[nb], o.a[0]);
                } else {
                    const int j = ks % 4, ja = j >> 1, jb = j & 1;
                    int cnt = 0;
#pragma unroll
                    for (int ia = 0; ia < G::n_k(ja); ++ia)
#pragma unroll
                        for (int ib = 0; ib < G::n_k(jb); ++ib) {
                            const int cl = G::cls_of(G::k_of(ja, ia)) * 2 + G::cls_of(G::k_of(jb, ib));
#pragma unroll
                            for (int nb = 0; nb < NBW; ++nb) mfma16(acc[cl][nb], o.b[nb], o.a[cnt]);
                            ++cnt;
                        }
                }
            };
            // read-ahead distance in sub-steps: a forward sub-step is NBW MFMAs (3 .. 13): two ahead; a data-gradient one 1 .. 4 x NBW: one
            constexpr int D = MODE == 0 ? 2 : 1;
            Ops ops[D + 1];
#pragma unroll
            for (int d = 0; d < D; ++d) read_ops(ops[d], d);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                if (ks + D < NKS) read_ops(ops[(ks + D) % (D + 1)], ks + D);
#pragma unroll
                for (int k = ks * PER_KS; k < (ks + 1) * PER_KS && k < G::NSLOT; ++k) dma_slot(k, bn, r0n, ccn, nxt);
                Ops& o = ops[ks % (D + 1)];
                asm volatile("" : "+v"(o.a[0]) : : "memory");  // (reads above stay above, MFMAs below stay below: conv_rows.hip)
                run_ops(o, ks);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the next unit's stage 0, issued early in the last stage)
        acc_settle_n<G::NCLS * NBW>(&acc[0][0]);
        // ---- this unit is complete.  D[i][j]: lane (j = n, kq) holds rows i = 4 kq + e: four CONSECUTIVE pixels of channel n per tile.
        const int co = co0 + wm * 16 + n;
        if constexpr (MODE == 0) {
            const float bs = (p.bias != nullptr && co < p.M) ? p.bias[co] : 0.f;
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                const int f = 16 * (wp * NBW + nb) + 4 * kq;
                f32x4 v, vr;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[0][nb][e] + bs;
                    vr[e] = v[e] >= 0.f ? v[e] : 0.f;
                }
                if constexpr (PK == 1) {
                    // a unit's pixels are RPU * WO consecutive floats of the output plane
                    const int lim0 = (G::OH - r0) * G::OW, lim = lim0 < G::PX ? lim0 : G::PX;
                    const int nval = co < p.M ? lim - f : 0;
                    const size_t at = ((size_t)b * p.M + co) * OHW + (size_t)r0 * G::OW + f;
                    if (nval >= 4) {
                        if (p.y != nullptr) *(f32x4u*)(p.y + at) = v;
                        if (p.y_relu != nullptr) *(f32x4u*)(p.y_relu + at) = vr;
                    } else {
#pragma unroll
                        for (int e = 0; e < 3; ++e)
                            if (e < nval) {
                                if (p.y != nullptr) p.y[at + e] = v[e];
                                if (p.y_relu != nullptr) p.y_relu[at + e] = vr[e];
                            }
                    }
                } else {
                    // packed samples: pixel f = pixel f % PXS of sample b + f / PXS
                    const int sp = f / G::PXS, pl = f - sp * G::PXS;
                    const bool whole = co < p.M && sp < PK && b + sp < p.B && pl + 3 < G::PXS;
                    if (whole) {
                        const size_t at = ((size_t)(b + sp) * p.M + co) * OHW + pl;
                        if (p.y != nullptr) *(f32x4u*)(p.y + at) = v;
                        if (p.y_relu != nullptr) *(f32x4u*)(p.y_relu + at) = vr;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int fe = f + e, se = fe / G::PXS, pe = fe - se * G::PXS;
                            if (co < p.M && fe < G::PX && b + se < p.B) {
                                const size_t at = ((size_t)(b + se) * p.M + co) * OHW + pe;
                                if (p.y != nullptr) p.y[at] = v[e];
                                if (p.y_relu != nullptr) p.y_relu[at] = vr[e];
                            }
                        }
                    }
                }
            }
        } else {
            // data gradient: pixel (yy, xx) of class (py, px) is dx[2 (r0 + r) + py][2 c + px]; the two column classes of a pixel are two
            // consecutive floats
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                const int f0 = 16 * (wp * NBW + nb) + 4 * kq;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int f = f0 + e;
                    const int sp = f / G::PXS, rem = f - sp * G::PXS, r = rem / G::DW, c = rem - r * G::DW;
                    const bool ok = co < p.M && f < G::PX && b + sp < p.B;
#pragma unroll
                    for (int py = 0; py < 2; ++py) {
                        const int y = 2 * (r0 + r) + py, x = 2 * c;
                        if (!ok || y >= G::OH) continue;
                        const size_t at = ((size_t)(b + sp) * p.M + co) * OHW + (size_t)y * G::OW + x;
                        float v0 = acc[py * 2 + 0][nb][e], v1 = acc[py * 2 + 1][nb][e];
                        if (p.relu_below != nullptr) {
                            v0 = p.relu_below[at] <= 0.f ? 0.f : v0;
                            if (x + 1 < G::OW) v1 = p.relu_below[at + 1] <= 0.f ? 0.f : v1;
                        }
                        p.y[at] = v0;
                        if (x + 1 < G::OW) p.y[at + 1] = v1;
                    }
                }
            }
        }
        zero_acc();
        b = bu; r0 = r0u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.dbg == 9 && threadIdx.x == 0 && (blockIdx.x | blockIdx.y) == 0) {
        const long long c = clock64() - dbg_c0, w = wall_clock64() - dbg_w0;
        printf("conv_s2 block 0: %lld shader cycles in %lld ticks of 10 ns -> %.0f MHz, %d stages\n", c, w, (double)c / ((double)w / 100.0), t);
    }
}

// ---- the instances: one per (pass, plane size, padding) of the BASELINE workloads ---------------------------------------------------
struct S2Plan {
    S2Params p;
    int inst, mt, qw, ntiles, blocks;
    size_t wt_floats;
    size_t lds;
};

struct S2Inst {
    int mode, hi, wi, pad, wm, rpu, pk;
};
// (geometry: HI x WI planes of the layer's input, padding; waves over channels; domain rows per unit; samples per unit)
constexpr S2Inst kInst[] = {
    // reference net behind its first block (alexnet.cpp:17-29), batch 256: 16 -> 32 @ 55, 32 -> 64 @ 27, 64 -> 128 @ 13, pad 0
    {0, 55, 55, 0, 2, 7, 1}, {0, 27, 27, 0, 4, 7, 1}, {0, 13, 13, 0, 4, 6, 2},
    {1, 55, 55, 0, 1, 14, 1}, {1, 27, 27, 0, 2, 14, 1}, {1, 13, 13, 0, 4, 7, 2},
    // stage entries of the ResNet-shaped stack, batch 64: 64 -> 128 @ 56, 128 -> 256 @ 28, 256 -> 512 @ 14, pad 1
    {0, 56, 56, 1, 4, 4, 1}, {0, 28, 28, 1, 4, 14, 1}, {0, 14, 14, 1, 4, 7, 2},
    {1, 56, 56, 1, 4, 4, 1}, {1, 28, 28, 1, 2, 14, 1}, {1, 14, 14, 1, 4, 7, 2},
};
constexpr int kNumInst = (int)(sizeof(kInst) / sizeof(kInst[0]));

template <int I>
struct InstGeom {
    using type = S2Geom<kInst[I].mode, kInst[I].hi, kInst[I].wi, kInst[I].pad, kInst[I].wm, kInst[I].rpu, kInst[I].pk>;
};

template <int I>
void fill_plan(S2Plan* pl, int B) {
    using G = typename InstGeom<I>::type;
    pl->mt = G::MT;
    pl->qw = G::QW;
    pl->lds = G::lds_bytes;
    pl->p.units_total = kInst[I].pk > 1 ? (B + kInst[I].pk - 1) / kInst[I].pk : B * G::NRB;
}

template <int I>
int launch_inst(const S2Plan& pl, const char* tag, const cnn_conv2d_desc* d, hipStream_t s) {
    using G = typename InstGeom<I>::type;
    constexpr S2Inst c = kInst[I];
    auto kern = conv_s2_kernel<c.mode, c.hi, c.wi, c.pad, c.wm, c.rpu, c.pk>;
    static DeviceOnce attr_once;
    if (attr_once.needed()) {
        CNN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_once.mark();
    }
    char name[48];
    snprintf(name, sizeof(name), "conv_s2<%d,%d,%d>/%s", c.wi, c.pad, G::MT, tag);
    CNN_KLAUNCH(s, name, (kern<<<dim3(pl.blocks, pl.ntiles), 256, G::lds_bytes, s>>>(pl.p)), "B%d Ci%d %dx%d Co%d k%d s%d p%d", d->B, d->Ci, d->H, d->W, d->Co,
                d->k, d->s, d->pad);
    return CNN_AMD_OK;
}

template <int... I>
void fill_any(int inst, S2Plan* pl, int B, std::integer_sequence<int, I...>) {
    ((inst == I ? fill_plan<I>(pl, B) : (void)0), ...);
}
template <int... I>
int launch_any(int inst, const S2Plan& pl, const char* tag, const cnn_conv2d_desc* d, hipStream_t s, std::integer_sequence<int, I...>) {
    int rc = CNN_AMD_E_BADARG;
    ((inst == I ? (void)(rc = launch_inst<I>(pl, tag, d, s)) : (void)0), ...);
    return rc;
}

// mode 0: forward of d; mode 1: data gradient of d
bool make_s2_plan(const cnn_conv2d_desc* d, int mode, S2Plan* pl) {
    const OptVal e = CNN_OPT_VAL("CONV_S2");
    if (e && atoi(e) == 0) return false;
    if (d->k != 3 || d->s != 2 || d->B < 1) return false;
    int inst = -1;
    for (int i = 0; i < kNumInst; ++i)
        if (kInst[i].mode == mode && kInst[i].hi == d->H && kInst[i].wi == d->W && kInst[i].pad == d->pad) inst = i;
    if (inst < 0) return false;
    const int C = mode == 0 ? d->Ci : d->Co, M = mode == 0 ? d->Co : d->Ci;
    if (C < 8 || C % 8 != 0 || M < 8 || (long long)C * M * 9 >= (1ll << 28)) return false;
    const int Ho = (d->H + 2 * d->pad - 3) / 2 + 1, Wo = (d->W + 2 * d->pad - 3) / 2 + 1;
    if ((long long)d->B * d->Ci * d->H * d->W >= (1ll << 29) || (long long)d->B * d->Co * Ho * Wo >= (1ll << 29)) return false;
    S2Params& p = pl->p;
    p.B = d->B; p.C = C; p.M = M;
    p.nchunk = C / 8;
    pl->inst = inst;
    fill_any(inst, pl, d->B, std::make_integer_sequence<int, kNumInst>());
    pl->ntiles = (M + pl->mt - 1) / pl->mt;
    const int env = CNN_OPT_INT("S2_BLOCKS", 0);
    // workgroups: the chip's CUs x how many of these fit one CU's LDS, over the channel tiles
    int per_cu = (int)((160 * 1024) / pl->lds);
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 2) per_cu = 2;
    long long want = (env > 0 ? env : num_cus() * per_cu) / pl->ntiles;
    if (want < 1) want = 1;
    if (want > p.units_total) want = p.units_total;
    p.units_per_block = (int)((p.units_total + want - 1) / want);
    pl->blocks = (p.units_total + p.units_per_block - 1) / p.units_per_block;
    pl->wt_floats = (size_t)pl->ntiles * p.nchunk * 8 * 9 * pl->qw;
    p.dbg = CNN_MEASURE_INT("S2_DBG", 0);
    return true;
}

}  // namespace

namespace cnn_amd {

// what conv_rows.hip's public entry points need to know to serve a stride-2 layer through the same interface
bool s2_info(const cnn_conv2d_desc* d, int mode, int* mt, int* qw, int* nchunk, int* ntiles, size_t* wt_floats) {
    S2Plan pl;
    if (!make_s2_plan(d, mode, &pl)) return false;
    *mt = pl.mt; *qw = pl.qw; *nchunk = pl.p.nchunk; *ntiles = pl.ntiles; *wt_floats = pl.wt_floats;
    return true;
}

int s2_run(const cnn_conv2d_desc* d, int mode, const float* in, const float* image, const float* bias, float* out, float* out_relu,
           const float* relu_below, hipStream_t s) {
    S2Plan pl;
    if (!make_s2_plan(d, mode, &pl)) return fail(CNN_AMD_E_BADARG, "conv_s2: geometry not covered");
    CNN_REQUIRE(mode == 0 || out != nullptr, "conv_s2: the data gradient needs its output");
    pl.p.x = in; pl.p.wt = image; pl.p.bias = mode == 0 ? bias : nullptr; pl.p.y = out; pl.p.y_relu = mode == 0 ? out_relu : nullptr;
    pl.p.relu_below = mode == 1 ? relu_below : nullptr;
    const char* tag = mode == 0 ? (out_relu ? (out ? "fwd+relu" : "fwd,relu") : "fwd") : (relu_below ? "dgrad+relu" : "dgrad");
    return launch_any(pl.inst, pl, tag, d, s, std::make_integer_sequence<int, kNumInst>());
}

}  // namespace cnn_amd
