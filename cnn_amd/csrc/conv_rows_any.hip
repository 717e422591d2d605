// conv_rows_any.hip -- Conv2D forward (cpu/src/conv2d.cpp:69-92) and data gradient (conv2d.cpp:168-199) of 3x3 / stride-1 layers with ANY
// plane size up to 224 columns (conv2d.cpp:41-42 accepts every H, W): the runtime-width member of the row-kernel family (round 6).
// conv_rows.hip is instantiated for the plane widths of the BASELINE workloads (112 / 110 / 56 / 28 / 14 / 7) with every width-dependent
// quantity a compile-time constant; every other 3x3 / stride-1 geometry used to run on the implicit GEMM (57 - 68 % of the MFMA peak on
// the reference's own pad-0 VGG shapes 222 / 109 / 52 / 50 / 23 / 21).  Here the only compile-time geometry is a CLASS: the LDS row pitch
// RP (>= W, a multiple of 4) and the number of staged rows XRMAX -- so that every LDS address is still a per-lane base + a compile-time
// immediate (tap row kx * RP + tap column ky, channel group s * 4 * plane stride) -- and everything that depends on W, H, the padding's
// edge columns and the unit height is computed ONCE per lane at kernel start (conv_rows_s2.hip's scheme):
//   * FLAT PIXELS: a unit = RPU output rows x all WO columns of one sample (RPU * WO <= 448, or 288 in the small-unit instances), numbered
//     flat and cut into blocks of 16 pixels; lane n of block nb keeps the staged-plane offset of ITS pixel in a register (boff[nb]);
//   * rows are staged ONE BY ONE by buffer-addressed LDS DMA ([rows][RP]; the 16-byte units of a row start at the row: their sources need
//     4-byte alignment only); a row outside the image (padding above / below, the ragged end of a plane) is moved as zeros;
//   * the PADDING COLUMNS ARE REAL ZEROS IN LDS: every staged row is [16-byte unit of zeros][W floats][zeros up to the pitch] -- the left
//     unit and the units behind the row's end are DMA lanes outside the descriptor; the one unit that straddles the row's end (W % 4 != 0)
//     brings the next row's first floats, which the lane that issued it overwrites with zeros once its DMA has landed, in front of the
//     stage barrier.  No lane masks, no selects and no padding-dependent code in the MFMA loop (the first version selected edge taps away
//     per B value: 14 x 3 bit tests per k-step cost the padding-2 instances 15 - 25 %).
// GEMM view, MFMA (v_mfma_f32_16x16x4_f32, pixels as the M operand, in-place AGPR accumulators), filter image, two buffers, one barrier per
// stage, the hand-pipelined operand reads and the fused bias / ReLU / ReLU' epilogues are conv_rows.hip's.  Workgroup: 64 output channels
// (waves 2 over channels x 2 over the unit's pixel blocks), a wave 32 channels x 14 (or 9) blocks = 28 (18) accumulator tiles.  The data gradient of a
// pad-p layer is the same kernel on dy with padding 2 - p and the transposed + flipped filter image (rows_prep mode 1).
#include <cstdlib>

#include "common.h"
#include "rows_common.h"

using namespace cnn_amd;

namespace {

struct AnyParams {
    const float* x;     // staged tensor [B][C][H][W]
    const float* wt;    // prepared filters [co tile][chunk][8][9][QW] (conv_rows.hip rows_prep)
    const float* bias;  // nullable (data gradient)
    float* y;           // output [B][M][HO][WO] (nullable when y_relu is given)
    float* y_relu;      // nullable: the output of the ReLU layer behind this one (relu.cpp:25)
    const float* relu_below;  // nullable (data gradient): output of the ReLU layer in front -- its backward pass (relu.cpp:37) on the way out
    int B, C, M;
    int H, W, HO, WO;
    int pad;            // of THIS pass (0 .. 2)
    int rpu;            // output rows per unit
    int nrb;            // units per sample
    int nchunk;         // ceil(C / 8)
    int units_total, units_per_block;
};

constexpr int kCK = 8, kKSTEPS = 2, kMT = 64, kMA = 2, kWM = 2, kWP = 2;
// pixel blocks per wave: 14 (448 pixels per unit) or 9 (288: planes of 300 - 580 pixels would leave half of a second 448-pixel unit empty)
constexpr int kNbwBig = 14, kNbwSmall = 9;
constexpr int kLead = 4;  // floats in front of every staged plane: a tap column left of the image stays a non-negative LDS offset

constexpr int kLeft = 4;  // zero floats in front of every staged row (one DMA unit): tap columns -1, -2
template <int RP, int XRMAX>
struct AnyGeom {
    static_assert(RP % 4 == 0, "row pitch in 16-byte units");
    static constexpr int UPR = RP / 4;                       // 16-byte units per staged row
    static constexpr int PLANE = kLead + XRMAX * RP + 4;     // (+ 4: the last tap of the last row may touch the float behind its row)
    static constexpr int QXP = stride16(PLANE);              // channel plane stride: 16 (mod 32)
    static constexpr int QW = kMT + 16;                      // filter row stride: 9 * QW = 16 (mod 32)
    static_assert(QXP % 32 == 16 && (9 * QW) % 32 == 16, "bank halves");
    static constexpr int XIMG = kCK * QXP, WIMG = kCK * 9 * QW;
    static constexpr int NIX = (XIMG / 4 + 63) / 64, NIWT = (WIMG / 4 + 63) / 64;
    static constexpr int NIWX = (NIX + 3) / 4, NIWW = (NIWT + 3) / 4;
    static constexpr int NSLOT = NIWX + NIWW;
    static constexpr int XS = NIX * 256, WS = NIWT * 256;
    static constexpr int BUF = XS + WS;
    static constexpr int DUMP = 2 * BUF;
    static constexpr size_t lds_bytes = (size_t)(2 * BUF + 4 * 256) * sizeof(float);
    static_assert(lds_bytes <= 160 * 1024, "LDS plan");
    static constexpr int NKS = 9 * kKSTEPS;
    static_assert((kKSTEPS * 4 * QXP + 2 * RP + 2) * 4 < 65536 && (kKSTEPS * 36 + 9) * QW * 4 < 65536, "LDS immediates are 16 bits");
};

template <int RP, int XRMAX, int NBW>
__global__ __launch_bounds__(256) void conv_any_kernel(const AnyParams p) {
    using G = AnyGeom<RP, XRMAX>;
    const int PAD = p.pad;
    constexpr int NB = NBW, MA = kMA;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % kWM, wp = wave / kWM;
    const int co0 = blockIdx.y * kMT;
    const int H = p.H, W = p.W, HW = H * W, WO = p.WO, HWO = p.HO * WO;
    const int PX = p.rpu * WO;  // pixels of a unit

    const int u_lo = blockIdx.x * p.units_per_block;
    const int u_hi = u_lo + p.units_per_block < p.units_total ? u_lo + p.units_per_block : p.units_total;
    if (u_lo >= u_hi) return;

    // ---- this wave's share of a stage's x DMA, decoded once: 16-byte unit q of the image = (channel plane, staged row, unit of the row)
    unsigned xd_off[G::NIWX];
    int xd_rr[G::NIWX];  // staged row | channel plane << 8 | floats to zero at the end of the unit << 16 (row 255: nothing to move)
    static_assert(XRMAX < 255, "packed row index");
#pragma unroll
    for (int i = 0; i < G::NIWX; ++i) {
        const int j = i * 4 + wave, q = j * 64 + lane;
        const int plane = q / (G::QXP / 4), e = q - plane * (G::QXP / 4) - kLead / 4;
        const int rr = e >= 0 ? e / G::UPR : 0, u = e - rr * G::UPR - kLeft / 4;  // (unit 0 of a row: the zero floats left of it)
        // (units wholly behind the row's end are not fetched: zeros; the unit that straddles it brings 4u + 4 - W floats of the next row)
        const bool have = j < G::NIX && plane < kCK && e >= 0 && rr < p.rpu + 2 && u >= 0 && 4 * u < W;
        xd_off[i] = have ? (unsigned)(plane * HW + rr * W + 4 * u + 4) * 4u : kOob;
        const int fix = have && 4 * u + 4 > W ? 4 * u + 4 - W : 0;
        xd_rr[i] = have ? (rr | (plane << 8) | (fix << 16)) : 255;
    }
    const int BACK = PAD * W + 4;  // floats the descriptor starts in front of the tensor (the scalar offset of a stage is never negative)
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.x - BACK), 0, (int)(((unsigned)p.B * p.C * HW + BACK) * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.wt + (size_t)blockIdx.y * p.nchunk * G::WIMG), 0, (int)((unsigned)p.nchunk * G::WIMG * 4u), 0x00020000);

    float* const dump = smem + G::DUMP + wave * 256;
    // slot k of the DMA of stage (sample b, first output row r0, chunk cc) into `buf` = [x image][filter image]
    auto dma_slot = [&](int k, int b, int r0, int cc, float* buf) {
        if (k < G::NIWX) {
            const int j = k * 4 + wave;
            float* d = j < G::NIX ? buf + j * 256 : dump;
            const int row0 = r0 - PAD;  // tensor row of staged row 0
            // rows outside the image are moved as zeros (a lane offset outside the descriptor), and so are the channels behind the tensor's last
            const unsigned voff = ((unsigned)(row0 + (xd_rr[k] & 255)) < (unsigned)H && cc * kCK + ((xd_rr[k] >> 8) & 255) < p.C) ? xd_off[k] : kOob;
            blds16(xrs, voff, (unsigned)((b * p.C + cc * kCK) * HW + r0 * W) * 4u, d);
        } else {
            const int i = k - G::NIWX, j = i * 4 + wave;
            float* d = j < G::NIWT ? buf + G::XS + j * 256 : dump;
            const unsigned q = (unsigned)(j * 64 + lane);
            blds16(wrs, (j < G::NIWT && q * 4 < (unsigned)G::WIMG) ? q * 16u : kOob, (unsigned)cc * (unsigned)(G::WIMG * 4), d);
        }
    };

    // ---- per-lane operand bases (floats inside a buffer) and edge bits of this lane's pixel in each of the wave's blocks
    int boff[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        int f = 16 * (wp * NB + nb) + n;
        if (f >= PX) f = 0;  // (a lane behind the unit's last pixel computes pixel 0 again; never stored)
        const int r = f / WO, c = f - r * WO;
        boff[nb] = kq * G::QXP + kLead + r * RP + kLeft + c - PAD;
    }
    // the floats this lane has to zero behind the end of a row once its DMA of a stage has landed: byte address inside a buffer (0: none; the
    // count 4 - W % 4 is the same for every such unit); fix_slots (wave-uniform): the DMA slots in which ANY lane of this wave has one -- a
    // stage of 4 rows x 8 channels has 32 such units among 1 900
    unsigned fixup[G::NIWX];
    unsigned fix_slots = 0;
#pragma unroll
    for (int i = 0; i < G::NIWX; ++i) {
        const int j = i * 4 + wave, fix = (xd_rr[i] >> 16) & 3;
        fixup[i] = (xd_rr[i] == 255 || fix == 0) ? 0u : (unsigned)(j * 256 + lane * 4 + 4 - fix) * 4u;
        if (__builtin_amdgcn_ballot_w64(fixup[i] != 0u) != 0ull) fix_slots |= 1u << i;
    }
    fix_slots = (unsigned)__builtin_amdgcn_readfirstlane((int)fix_slots);
    const int nfix = (4 - (W & 3)) & 3;  // (uniform)
    const int a_base = G::XS + kq * 9 * G::QW + wm * (16 * MA) + n;

    f32x4 acc[MA][NB];
    auto zero_acc = [&]() {
#pragma unroll
        for (int ma = 0; ma < MA; ++ma)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[ma][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    int b = u_lo / p.nrb, r0 = (u_lo - b * p.nrb) * p.rpu;
    {
#pragma unroll
        for (int k = 0; k < G::NSLOT; ++k) dma_slot(k, b, r0, 0, smem);
    }
    constexpr int NKS = G::NKS;
    constexpr int PER_KS = (G::NSLOT + NKS - 1) / NKS;
    int t = 0;  // stages so far: buffer parity
    for (int u = u_lo; u < u_hi; ++u) {
        int bu = b, r0u = r0;  // the unit behind this one (behind the last one: this one again)
        if (u + 1 < u_hi) {
            if (r0 + p.rpu < p.HO) r0u = r0 + p.rpu;
            else { r0u = 0; bu = b + 1; }
        }
        for (int cc = 0; cc < p.nchunk; ++cc, ++t) {
            // (stage 0 of a unit behind the first was waited for in front of the previous unit's stores: conv_rows.hip)
            if (cc != 0 || u == u_lo) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (fix_slots != 0) {  // (wave-uniform) rows that do not end on a 16-byte unit: zero what the straddling unit brought of the next row
                // (a unit's stage 0 that was waited for in front of the previous unit's stores has landed as well: the wait above is skipped
                //  only when it already happened)
                char* cur = (char*)(smem + (t & 1) * G::BUF);
#pragma unroll
                for (int i = 0; i < G::NIWX; ++i) {
                    if (!((fix_slots >> i) & 1u)) continue;  // (scalar branch)
                    if (fixup[i] != 0u) {
                        float* z = (float*)(cur + fixup[i]);
                        z[0] = 0.f;
                        if (nfix >= 2) z[1] = 0.f;
                        if (nfix >= 3) z[2] = 0.f;
                    }
                }
            }
            __syncthreads();
            float* nxt = smem + ((t + 1) & 1) * G::BUF;
            const bool last_cc = cc + 1 == p.nchunk;
            const int bn = last_cc ? bu : b, r0n = last_cc ? r0u : r0, ccn = last_cc ? (u + 1 < u_hi ? 0 : cc) : cc + 1;
            const unsigned par = (unsigned)(t & 1) * (unsigned)(G::BUF * 4);
            unsigned bad[NB];  // byte addresses inside the current buffer
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bad[nb] = (unsigned)(boff[nb] * 4) + par;
            const unsigned aad = (unsigned)(a_base * 4) + par;
            struct Ops {
                float a[MA];
                float b[NB];
            };
            constexpr int NRD = MA + NB;  // reads of a batch
            Ops ops[3];
            auto issue_read = [&](auto KS, auto R) {
                constexpr int ks = decltype(KS)::value, r = decltype(R)::value;
                constexpr int tap = ks / kKSTEPS, s = ks % kKSTEPS, kx = tap / 3, ky = tap % 3;
                Ops& o = ops[ks % 3];
                if constexpr (r < MA) lds_rd<((s * 36 + tap) * G::QW + r * 16) * 4>(o.a[r], aad);
                else lds_rd<(s * 4 * G::QXP + kx * RP + ky) * 4>(o.b[r - MA], bad[r - MA]);
            };
            auto issue_batch = [&](auto KS) { static_for<NRD>([&](auto R) { issue_read(KS, R); }); };
            issue_batch(std::integral_constant<int, 0>());
            issue_batch(std::integral_constant<int, 1>());
            static_for<NKS>([&](auto KS) {
                constexpr int ks = decltype(KS)::value;
                constexpr int nm = MA * NB;                       // MFMAs of a k-step
                constexpr int nr2 = ks + 2 < NKS ? NRD : 0;       // reads of batch ks + 2, issued here
                constexpr int rpm = (nr2 + nm - 1) / nm;
#pragma unroll
                for (int k = ks * PER_KS; k < (ks + 1) * PER_KS && k < G::NSLOT; ++k) dma_slot(k, bn, r0n, ccn, nxt);
                lgkm_wait<(ks + 1 < NKS ? (NRD < 15 ? NRD : 15) : 0)>();  // (the reads of batch ks + 1 may be in flight: the counter has four bits)
                Ops& o = ops[ks % 3];
#pragma unroll
                for (int ma = 0; ma < MA; ++ma) asm volatile("" : "+v"(o.a[ma]));
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) asm volatile("" : "+v"(o.b[nb]));
                static_for<nm>([&](auto IM) {
                    constexpr int im = decltype(IM)::value, nb = im / MA, ma = im % MA;
                    mfma16(acc[ma][nb], o.b[nb], o.a[ma]);
                    if constexpr (nr2 > 0)
                        static_for<rpm>([&](auto J) {
                            constexpr int r = im * rpm + decltype(J)::value;
                            if constexpr (r < nr2) issue_read(std::integral_constant<int, (ks + 2 < NKS ? ks + 2 : 0)>(), std::integral_constant<int, r>());
                        });
                });
            });
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the next unit's stage 0, issued early in the last stage)
        acc_settle<MA * NB>(&acc[0][0]);
        // ---- this unit is complete: + bias, (ReLU' mask), store.  D[i][j]: lane (j = n, kq) holds rows i = 4 kq + e: four CONSECUTIVE pixels of
        //      channel n per tile -- and a unit's pixels are RPU * WO consecutive floats of the output plane: one 16-byte access per tile
        //      (at any float alignment).  The masks of one 16-channel block are fetched as one batch of loads before any of them is used.
        const int lim0 = (p.HO - r0) * WO, lim = lim0 < PX ? lim0 : PX;  // pixels of the unit inside the image
#pragma unroll
        for (int ma = 0; ma < MA; ++ma) {
            const int co = co0 + wm * (16 * MA) + ma * 16 + n;
            const float bs = (p.bias != nullptr && co < p.M) ? p.bias[co] : 0.f;
            const size_t cbase = ((size_t)b * p.M + co) * HWO + (size_t)r0 * WO;
            f32x4 mk[NB];
            int nval[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int f = 16 * (wp * NB + nb) + 4 * kq;
                nval[nb] = co < p.M ? lim - f : 0;
                mk[nb] = f32x4{1.f, 1.f, 1.f, 1.f};
                if (p.relu_below != nullptr) {
                    const float* m = p.relu_below + cbase + f;
                    if (nval[nb] >= 4) mk[nb] = *(const f32x4u*)m;
                    else {
#pragma unroll
                        for (int e = 0; e < 3; ++e)
                            if (e < nval[nb]) mk[nb][e] = m[e];
                    }
                }
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int f = 16 * (wp * NB + nb) + 4 * kq;
                const size_t at = cbase + f;
                f32x4 v, vr;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[ma][nb][e] + bs;
                    if (p.relu_below != nullptr) v[e] = mk[nb][e] <= 0.f ? 0.f : v[e];
                    vr[e] = v[e] >= 0.f ? v[e] : 0.f;
                }
                if (nval[nb] >= 4) {
                    if (p.y != nullptr) *(f32x4u*)(p.y + at) = v;
                    if (p.y_relu != nullptr) *(f32x4u*)(p.y_relu + at) = vr;
                } else {
#pragma unroll
                    for (int e = 0; e < 3; ++e)
                        if (e < nval[nb]) {
                            if (p.y != nullptr) p.y[at + e] = v[e];
                            if (p.y_relu != nullptr) p.y_relu[at + e] = vr[e];
                        }
                }
            }
        }
        zero_acc();
        b = bu; r0 = r0u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- the classes: (row pitch, staged rows).  A plane of width W takes the class with the smallest pitch >= W; its unit is
//      as many output rows as its staged rows and pixel blocks hold (make_any_plan) ----------------------------------------------------------------------------------
struct AnyClass {
    int rp, xrmax;
};
constexpr AnyClass kClasses[] = {{36, 30}, {64, 12}, {120, 6}, {232, 4}};  // planes up to 30 / 58 / 114 / 226 wide
constexpr int kNumClasses = (int)(sizeof(kClasses) / sizeof(kClasses[0]));

struct AnyPlan {
    AnyParams p;
    int cls, pad, nbw, ntiles, blocks;
    size_t wt_floats;
};

// mode 0: forward of d; mode 1: data gradient of d (= this kernel on dy with padding 2 - pad)
bool make_any_plan(const cnn_conv2d_desc* d, int mode, AnyPlan* pl) {
    const OptVal e = CNN_OPT_VAL("ROWS_ANY"), all = CNN_OPT_VAL("CONV_ROWS");  // (CONV_ROWS=0: the whole row-kernel family off)
    if ((e && atoi(e) == 0) || (all && atoi(all) == 0)) return false;
    if (d->k != 3 || d->s != 1 || d->pad < 0 || d->pad > 1 || d->B < 1) return false;
    const int Ho = d->H + 2 * d->pad - 2, Wo = d->W + 2 * d->pad - 2;
    if (Ho < 1 || Wo < 1) return false;
    const int wi = mode == 0 ? d->W : Wo, hi = mode == 0 ? d->H : Ho, pad = mode == 0 ? d->pad : 2 - d->pad;
    const int wo = mode == 0 ? Wo : d->W, ho = mode == 0 ? Ho : d->H;
    const int C = mode == 0 ? d->Ci : d->Co, M = mode == 0 ? d->Co : d->Ci;
    // (any channel count from 8 / 16 up: the channels behind the tensor's last one in the last 8-channel stage are DMA lanes outside the
    //  descriptor -- zeros -- and their filter rows are zeros in the image; output channels behind M are computed and not stored)
    if (C < 8 || M < 16 || (long long)C * M * 9 >= (1ll << 28)) return false;
    if ((long long)d->B * C * hi * wi >= (1ll << 29) || (long long)d->B * M * ho * wo >= (1ll << 31)) return false;
    int cls = -1;
    for (int i = kNumClasses - 1; i >= 0; --i)
        if (wi + kLeft + 2 <= kClasses[i].rp) cls = i;  // (the zero unit left of the row, two zero columns right of it)
    if (cls < 0 || wo < 4) return false;
    // unit shape: blocks per wave (14 or 9) x output rows, the pair that spends the fewest MFMA pixel slots on the plane -- units per plane x
    // unit capacity for ho * wo live pixels (23-wide outputs: two units of 288 pixels, not two of 448; 21-wide: ONE unit of 21 rows = 441
    // pixels, not 16 + 5 rows); ties go to the bigger / taller unit (fewer halo rows and filter stages per pixel)
    int rpu = 0, nbw = 0;
    long long best = -1;
    for (int cand : {kNbwBig, kNbwSmall}) {
        const int cap = 16 * kWP * cand;
        int rmax = kClasses[cls].xrmax - 2;
        if (rmax > cap / wo) rmax = cap / wo;
        if (rmax > ho) rmax = ho;
        for (int r = rmax; r >= 1; --r) {
            const long long cost = (long long)((ho + r - 1) / r) * cap;
            if (best < 0 || cost < best) { best = cost; rpu = r; nbw = cand; }
        }
    }
    if (rpu < 1) return false;
    pl->nbw = nbw;
    AnyParams& p = pl->p;
    p.B = d->B; p.C = C; p.M = M; p.H = hi; p.W = wi; p.HO = ho; p.WO = wo;
    p.pad = pad;
    p.rpu = rpu;
    p.nrb = (ho + rpu - 1) / rpu;
    p.nchunk = (C + kCK - 1) / kCK;
    p.units_total = d->B * p.nrb;
    pl->cls = cls; pl->pad = pad;
    pl->ntiles = (M + kMT - 1) / kMT;
    const int env = CNN_OPT_INT("ROWS_BLOCKS", 0);
    long long want = (env > 0 ? env : num_cus()) / pl->ntiles;
    if (want < 1) want = 1;
    if (want > p.units_total) want = p.units_total;
    p.units_per_block = (int)((p.units_total + want - 1) / want);
    pl->blocks = (p.units_total + p.units_per_block - 1) / p.units_per_block;
    pl->wt_floats = (size_t)pl->ntiles * p.nchunk * kCK * 9 * (kMT + 16);
    return true;
}

template <int CLS, int NBW>
int launch_any2(const AnyPlan& pl, const char* tag, const cnn_conv2d_desc* d, hipStream_t s) {
    constexpr AnyClass c = kClasses[CLS];
    using G = AnyGeom<c.rp, c.xrmax>;
    auto kern = conv_any_kernel<c.rp, c.xrmax, NBW>;
    static DeviceOnce attr_once;
    if (attr_once.needed()) {
        CNN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_once.mark();
    }
    char name[48];
    snprintf(name, sizeof(name), "conv_rows_any<%d,%dx%d>/%s", c.rp - kLeft - 2, pl.p.rpu, 16 * kWP * NBW, tag);
    CNN_KLAUNCH(s, name, (kern<<<dim3(pl.blocks, pl.ntiles), 256, G::lds_bytes, s>>>(pl.p)), "B%d Ci%d %dx%d Co%d k%d s%d p%d", d->B, d->Ci, d->H, d->W, d->Co,
                d->k, d->s, d->pad);
    return CNN_AMD_OK;
}

template <int CLS>
int launch_any1(const AnyPlan& pl, const char* tag, const cnn_conv2d_desc* d, hipStream_t s) {
    return pl.nbw == kNbwBig ? launch_any2<CLS, kNbwBig>(pl, tag, d, s) : launch_any2<CLS, kNbwSmall>(pl, tag, d, s);
}

}  // namespace

namespace cnn_amd {

// what conv_rows.hip's public entry points need to know to serve such a layer through the same interface
bool any_info(const cnn_conv2d_desc* d, int mode, int* mt, int* qw, int* ck, int* nchunk, int* ntiles, size_t* wt_floats) {
    AnyPlan pl;
    if (!make_any_plan(d, mode, &pl)) return false;
    *mt = kMT; *qw = kMT + 16; *ck = kCK; *nchunk = pl.p.nchunk; *ntiles = pl.ntiles; *wt_floats = pl.wt_floats;
    return true;
}

int any_run(const cnn_conv2d_desc* d, int mode, const float* in, const float* image, const float* bias, float* out, float* out_relu,
            const float* relu_below, hipStream_t s) {
    AnyPlan pl;
    if (!make_any_plan(d, mode, &pl)) return fail(CNN_AMD_E_BADARG, "conv_rows_any: geometry not covered");
    pl.p.x = in; pl.p.wt = image; pl.p.bias = mode == 0 ? bias : nullptr; pl.p.y = out; pl.p.y_relu = out_relu; pl.p.relu_below = relu_below;
    const char* tag = mode == 0 ? (out_relu ? (out ? "fwd+relu" : "fwd,relu") : "fwd") : (relu_below ? "dgrad+relu" : "dgrad");
    switch (pl.cls) {
        case 0: return launch_any1<0>(pl, tag, d, s);
        case 1: return launch_any1<1>(pl, tag, d, s);
        case 2: return launch_any1<2>(pl, tag, d, s);
        default: return launch_any1<3>(pl, tag, d, s);
    }
}

}  // namespace cnn_amd
