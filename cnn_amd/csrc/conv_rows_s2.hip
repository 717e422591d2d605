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
    int nchunk;         // ceil(C / CK)
    int units_total, units_per_block;
    int dbg;
};

template <int MODE, int HI, int WI, int PAD, int WM, int RPU, int PK, int CK_>
struct S2Geom {
    static_assert(MODE == 0 || MODE == 1, "0 forward, 1 data gradient");
    static_assert(PAD == 0 || PAD == 1, "padding");
    static_assert(WM == 1 || WM == 2 || WM == 4, "waves over channels");
    static_assert(CK_ == 8 || CK_ == 16, "channels per stage");
    static constexpr int CK = CK_, KSTEPS = CK / 4;
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
    static constexpr int NKS = MODE == 0 ? 9 * KSTEPS : 4 * KSTEPS;  // (sub-step ks: forward tap = ks / KSTEPS, group = ks % KSTEPS; data gradient group = ks / 4, B = ks % 4)
    // data gradient: the tap rows (columns) whose dy row (column) offset is 0 (j = 0: two of them) or the other one (j = 1: one)
    static constexpr int n_k(int j) { return j == 0 ? 2 : 1; }
    static constexpr int k_of(int j, int i) { return PAD == 0 ? (j == 0 ? i : 2) : (j == 0 ? 1 + i : 0); }
    static constexpr int d_of(int j) { return j == 0 ? 0 : (PAD == 0 ? -1 : 1); }
    static constexpr int cls_of(int k) { return (k + PAD) & 1; }  // parity class of the dx row (column) tap row (column) k feeds
    // ---- sub-step ks of a stage, as compile-time tables: how many A values, their filter-image offsets (floats), the parity class each
    //      feeds, the B operand's offset from a lane's pixel base, the column offset its lane mask is about
    static constexpr int ks_s(int ks) { return MODE == 0 ? ks % KSTEPS : ks / 4; }
    static constexpr int ks_ja(int ks) { return (ks % 4) >> 1; }
    static constexpr int ks_jb(int ks) { return (ks % 4) & 1; }
    static constexpr int ks_na(int ks) { return MODE == 0 ? 1 : n_k(ks_ja(ks)) * n_k(ks_jb(ks)); }
    static constexpr int ks_kx(int ks, int i) { return MODE == 0 ? (ks / KSTEPS) / 3 : k_of(ks_ja(ks), i / n_k(ks_jb(ks))); }
    static constexpr int ks_ky(int ks, int i) { return MODE == 0 ? (ks / KSTEPS) % 3 : k_of(ks_jb(ks), i % n_k(ks_jb(ks))); }
    static constexpr int ks_aoff(int ks, int i) { return (ks_s(ks) * 36 + ks_kx(ks, i) * 3 + ks_ky(ks, i)) * QW; }
    static constexpr int ks_cls(int ks, int i) { return MODE == 0 ? 0 : cls_of(ks_kx(ks, i)) * 2 + cls_of(ks_ky(ks, i)); }
    static constexpr int BIAS = MODE == 1 && PAD == 0 ? RP + 1 : 0;  // (keeps every read's immediate offset non-negative: folded into the lane bases)
    static constexpr int ks_boff(int ks) {
        return ks_s(ks) * 4 * QXP + (MODE == 0 ? ks_kx(ks, 0) * RP + ks_ky(ks, 0) : d_of(ks_ja(ks)) * RP + d_of(ks_jb(ks)) + BIAS);
    }
    static constexpr int ks_kd(int ks) { return MODE == 0 ? ks_ky(ks, 0) : d_of(ks_jb(ks)); }
    static constexpr int max_off() {
        int m = 0;
        for (int ks = 0; ks < NKS; ++ks) {
            m = ks_boff(ks) > m ? ks_boff(ks) : m;
            for (int i = 0; i < ks_na(ks); ++i) m = ks_aoff(ks, i) > m ? ks_aoff(ks, i) : m;
        }
        return m;
    }
    static_assert(max_off() * 4 < 65536, "LDS immediates are 16 bits");
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

template <int MODE, int HI, int WI, int PAD, int WM, int RPU, int PK, int CK_>
__global__ __launch_bounds__(256) void conv_s2_kernel(const S2Params p) {
    using G = S2Geom<MODE, HI, WI, PAD, WM, RPU, PK, CK_>;
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
    int xd_rr[G::NIWX];  // staged row | channel plane << 8
    static_assert(G::XR < 256, "packed row index");
#pragma unroll
    for (int i = 0; i < G::NIWX; ++i) {
        const int j = i * 4 + wave, q = j * 64 + lane;
        const int plane = q / (G::QXP / 4), e = q - plane * (G::QXP / 4);
        const int sp = e / (G::XR * G::UPR), rr = (e / G::UPR) % G::XR, u = e % G::UPR;
        const bool have = j < G::NIX && plane < CK && sp < PK;
        // float index from the descriptor's base (BACK floats in front of the tensor), without the stage's (sample, chunk, first row) part
        xd_off[i] = have ? (unsigned)((sp * p.C + plane) * SHW + rr * G::SW + 4 * u - (G::SKEW ? (plane & 1) : 0) + 4) * 4u : kOob;
        xd_rr[i] = rr | (plane << 8);
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
            // ... and so are the channels behind the tensor's last one (a ragged last chunk): what lies there is the next sample
            const unsigned voff = ((unsigned)(row0 + (xd_rr[k] & 255)) < (unsigned)G::SH && cc * CK + (xd_rr[k] >> 8) < p.C) ? xd_off[k] : kOob;
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
        else boff[nb] = kq * G::QXP + sp * G::SUBP + (r + G::NEG) * G::RP + c - G::BIAS;
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
    // the next stage's DMA is issued in the FIRST THIRD of this stage's sub-steps (a stage is 100 - 250 MFMAs per wave, 1.5 - 4 us: a slot
    // issued in its last sub-steps would have its whole memory latency exposed at the next stage's wait)
    constexpr int ISSUE_KS = NKS / 3 > 0 ? NKS / 3 : 1;
    constexpr int PER_KS = (G::NSLOT + ISSUE_KS - 1) / ISSUE_KS;
    int t = 0;  // stages so far: buffer parity
#ifdef CNN_AMD_MEASURE  // (S2_DBG=9: workgroup 0 prints its shader clock and where its cycles went)
    const long long dbg_c0 = p.dbg == 9 ? clock64() : 0, dbg_w0 = p.dbg == 9 ? wall_clock64() : 0;
    long long dbg_wait = 0, dbg_epi = 0;
#endif
    for (int u = u_lo; u < u_hi; ++u) {
        int bu = b, r0u = r0;  // the unit behind this one (behind the last one: this one again)
        if (u + 1 < u_hi) {
            if (r0 + RPU < G::DH) r0u = r0 + RPU;
            else { r0u = 0; bu = b + PK; }
        }
        for (int cc = 0; cc < p.nchunk; ++cc, ++t) {
            // (stage 0 of a unit behind the first was waited for in front of the previous unit's stores: conv_rows.hip)
#ifdef CNN_AMD_MEASURE
            const long long dbg_t0 = p.dbg == 9 ? clock64() : 0;
#endif
            if (cc != 0 || u == u_lo) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#ifdef CNN_AMD_MEASURE
            if (p.dbg == 9) dbg_wait += clock64() - dbg_t0;
#endif
            float* nxt = smem + ((t + 1) & 1) * G::BUF;
            const bool last_cc = cc + 1 == p.nchunk;
            const int bn = last_cc ? bu : b, r0n = last_cc ? r0u : r0, ccn = last_cc ? (u + 1 < u_hi ? 0 : cc) : cc + 1;
            // ---- the stage's sub-steps, software-pipelined by hand.  One wave per SIMD (the LDS plan leaves room for one workgroup per CU): the
            // MFMA pipe only stays busy while THIS wave issues an MFMA every 32 cycles, so the operand reads of sub-step ks + 2 are issued one or
            // two at a time BETWEEN the MFMAs of sub-step ks (an MFMA leaves ~6 free issue slots), as inline assembly in program order; the
            // compiler's own placement -- every ds_read of a sub-step in one clump in front of its MFMAs -- cost 60 - 100 idle cycles per
            // sub-step of 416 (measured: 63 % of the MFMA issue rate over the stage loop).  LDS returns in order: before the MFMAs of sub-step
            // ks only the reads of batch ks + 1 may still be in flight (s_waitcnt lgkmcnt(their count): at most 14).
            const unsigned par = (unsigned)(t & 1) * (unsigned)(G::BUF * 4);
            unsigned bad[NBW];  // byte addresses inside the current buffer
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) bad[nb] = (unsigned)(boff[nb] * 4) + par;
            const unsigned aad = (unsigned)(a_base * 4) + par;
            struct Ops {
                float a[MODE == 0 ? 1 : 4];
                float b[NBW];
            };
            Ops ops[3];
            // read r of batch ks: r < nA: A value r; else B value r - nA
            auto issue_read = [&](auto KS, auto R) {
                constexpr int ks = decltype(KS)::value, r = decltype(R)::value;
                Ops& o = ops[ks % 3];
                if constexpr (r < G::ks_na(ks)) lds_rd<G::ks_aoff(ks, r) * 4>(o.a[r], aad);
                else lds_rd<G::ks_boff(ks) * 4>(o.b[r - G::ks_na(ks)], bad[r - G::ks_na(ks)]);
            };
            auto issue_batch = [&](auto KS) {
                constexpr int ks = decltype(KS)::value;
                static_for<G::ks_na(ks) + NBW>([&](auto R) { issue_read(KS, R); });
            };
            issue_batch(std::integral_constant<int, 0>());
            if constexpr (NKS > 1) issue_batch(std::integral_constant<int, 1>());
            static_for<NKS>([&](auto KS) {
                constexpr int ks = decltype(KS)::value;
                constexpr int na = G::ks_na(ks), nm = na * NBW;  // MFMAs of this sub-step
                constexpr int nr2 = ks + 2 < NKS ? G::ks_na(ks + 2 < NKS ? ks + 2 : 0) + NBW : 0;  // reads of batch ks + 2, issued here
                constexpr int rpm = (nr2 + nm - 1) / nm;                                           // ... per MFMA
#pragma unroll
                for (int k = ks * PER_KS; k < (ks + 1) * PER_KS && k < G::NSLOT; ++k) dma_slot(k, bn, r0n, ccn, nxt);
                lgkm_wait<(ks + 1 < NKS ? G::ks_na(ks + 1 < NKS ? ks + 1 : 0) + NBW : 0)>();
                Ops& o = ops[ks % 3];
                // (the values of this batch are defined from here on: nothing that uses them may be scheduled above the wait)
#pragma unroll
                for (int i = 0; i < na; ++i) asm volatile("" : "+v"(o.a[i]));
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    asm volatile("" : "+v"(o.b[nb]));
                    // the tap columns that leave their staged row (a select: what lies there is the neighbouring row's data)
                    constexpr int kd = G::ks_kd(ks);
                    unsigned cm = 0;
#pragma unroll
                    for (int w = 0; w < WP; ++w) cm |= G::colmask(w * NBW + nb, kd);  // (is any wave's block touched at all: compile time)
                    if (cm != 0) {
                        unsigned mine = G::colmask(nb, kd);
#pragma unroll
                        for (int w = 1; w < WP; ++w) mine = wp == w ? G::colmask(w * NBW + nb, kd) : mine;
                        o.b[nb] = ((mine >> n) & 1u) ? 0.f : o.b[nb];
                    }
                }
                static_for<nm>([&](auto IM) {
                    constexpr int im = decltype(IM)::value, i = im / NBW, nb = im % NBW;
                    mfma16(acc[G::ks_cls(ks, i)][nb], o.b[nb], o.a[i]);
                    if constexpr (nr2 > 0)
                        static_for<rpm>([&](auto J) {
                            constexpr int r = im * rpm + decltype(J)::value;
                            if constexpr (r < nr2) issue_read(std::integral_constant<int, (ks + 2 < NKS ? ks + 2 : 0)>(), std::integral_constant<int, r>());
                        });
                });
            });
        }
#ifdef CNN_AMD_MEASURE
        const long long dbg_t1 = p.dbg == 9 ? clock64() : 0;
#endif
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
            // data gradient: pixel (yy, xx) of class (py, px) is dx[2 (r0 + r) + py][2 c + px].  A lane holds four consecutive pixels of a
            // block; two neighbours of one domain row are FOUR consecutive floats of a dx row (both column classes): one 16-byte store (and one
            // 16-byte load of the ReLU' mask) per pixel pair and row class.  (Scalar stores -- 64 different cache lines per instruction --
            // made the first version's epilogue three times as long as its MFMA loop.)
            // The masks of GN blocks (4 GN 16-byte loads per lane) are fetched as ONE batch before any of them is used: written as load -> select
            // -> store per tile the loads could not be moved above the previous tile's store (the compiler must assume y and relu_below alias),
            // and every tile paid a full memory round trip.
            constexpr int GN = 2;
            struct Pair {
                bool fast[2];
                size_t at[2], at1[2];
                bool ok, ok1;
                int y[2], y1[2], x, x1;
            };
            auto pair_of = [&](int nb, int q) {
                Pair g;
                const int f = 16 * (wp * NBW + nb) + 4 * kq + 2 * q;
                const int sp = f / G::PXS, rem = f - sp * G::PXS, r = rem / G::DW, c = rem - r * G::DW;
                g.ok = co < p.M && f < G::PX && b + sp < p.B;
                // the pair's second pixel: same row unless the first one is a row's last (odd domain widths only)
                const bool same = G::DW % 2 == 0 || c + 1 < G::DW;
                int sp1 = sp, r1 = r, c1 = c + 1;
                if (!same) { c1 = 0; r1 = r + 1; if (r1 == RPU) { r1 = 0; sp1 = sp + 1; } }
                g.ok1 = co < p.M && f + 1 < G::PX && b + sp1 < p.B;
                g.x = 2 * c; g.x1 = 2 * c1;
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    g.y[py] = 2 * (r0 + r) + py; g.y1[py] = 2 * (r0 + r1) + py;
                    g.at[py] = ((size_t)(b + sp) * p.M + co) * OHW + (size_t)g.y[py] * G::OW + g.x;
                    g.at1[py] = ((size_t)(b + sp1) * p.M + co) * OHW + (size_t)g.y1[py] * G::OW + g.x1;
                    g.fast[py] = g.ok && g.ok1 && same && g.y[py] < G::OH && g.x + 3 < G::OW;
                }
                return g;
            };
#pragma unroll
            for (int nb0 = 0; nb0 < NBW; nb0 += GN) {
                f32x4 mk[GN][2][2];
                if (p.relu_below != nullptr) {
#pragma unroll
                    for (int i = 0; i < GN; ++i)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            if (nb0 + i >= NBW) continue;
                            const Pair g = pair_of(nb0 + i, q);
#pragma unroll
                            for (int py = 0; py < 2; ++py) {
                                mk[i][q][py] = f32x4{1.f, 1.f, 1.f, 1.f};
                                if (g.fast[py]) mk[i][q][py] = *(const f32x4u*)(p.relu_below + g.at[py]);
                            }
                        }
                }
#pragma unroll
                for (int i = 0; i < GN; ++i)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int nb = nb0 + i;
                        if (nb >= NBW) continue;
                        const Pair g = pair_of(nb, q);
#pragma unroll
                        for (int py = 0; py < 2; ++py) {
                            f32x4 v{acc[py * 2 + 0][nb][2 * q], acc[py * 2 + 1][nb][2 * q], acc[py * 2 + 0][nb][2 * q + 1], acc[py * 2 + 1][nb][2 * q + 1]};
                            if (g.fast[py]) {
                                if (p.relu_below != nullptr) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = mk[i][q][py][e] <= 0.f ? 0.f : v[e];
                                }
                                *(f32x4u*)(p.y + g.at[py]) = v;
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const bool second = e >= 2;
                                    const bool good = second ? (g.ok1 && g.y1[py] < G::OH && g.x1 + (e & 1) < G::OW) : (g.ok && g.y[py] < G::OH && g.x + (e & 1) < G::OW);
                                    if (good) {
                                        const size_t a1 = (second ? g.at1[py] : g.at[py]) + (e & 1);
                                        float val = v[e];
                                        if (p.relu_below != nullptr) val = p.relu_below[a1] <= 0.f ? 0.f : val;
                                        p.y[a1] = val;
                                    }
                                }
                            }
                        }
                    }
            }
        }
        zero_acc();
        b = bu; r0 = r0u;
#ifdef CNN_AMD_MEASURE
        if (p.dbg == 9) dbg_epi += clock64() - dbg_t1;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef CNN_AMD_MEASURE
    if (p.dbg == 9 && threadIdx.x == 0 && (blockIdx.x | blockIdx.y) == 0) {
        const long long c = clock64() - dbg_c0, w = wall_clock64() - dbg_w0;
        printf("conv_s2 block 0: %lld shader cycles in %lld ticks of 10 ns -> %.0f MHz, %d stages; waits at stage starts %lld, epilogues (with their vmcnt wait) %lld cycles\n", c, w,
               (double)c / ((double)w / 100.0), t, dbg_wait, dbg_epi);
    }
#endif
}

// ---- the instances: one per (pass, plane size, padding) of the BASELINE workloads ---------------------------------------------------
struct S2Plan {
    S2Params p;
    int inst, mt, qw, ck, ntiles, blocks;
    size_t wt_floats;
    size_t lds;
};

struct S2Inst {
    int mode, hi, wi, pad, wm, rpu, pk, ck;
    int dflt;  // taken by the default dispatch (1) or only with CNN_AMD_CONV_S2=2 (0: measured slower in the step than the kernel it would replace)
};
// (geometry: HI x WI planes of the layer's input, padding; waves over channels; domain rows per unit; samples per unit)
// Measured on MI355X (tools/s2_layers.sh alone; bench.py --breakdown in the step), us, this kernel / the kernel it replaces:
//   reference net, batch 256   forward: 55-wide 31.6 / 31.8 alone, 32.5 / 30.8 in the step; 27-wide 24.8 / 28.9, 25.7 / 29.5; 13-wide 22.3 / 26.3, 23.9 / 27.0
//                              data gradient (+ ReLU'): 65 / 36, 49 / 32, 66 / 42 alone -- these layers are HBM-bound (0.5 MFLOP per output KB) and a
//                              kernel whose four waves per CU compute and store in lockstep leaves HBM idle while it computes: the register-direct
//                              kernels (many small workgroups, phases naturally interleaved) keep them
//                              IN THE STEP the forward instances lose too although they win alone (A/B on one box, 27- and 13-wide forward on:
//                              624 k against 643 k images/s): a workgroup that owns a CU's LDS shuts out the deferred first-layer data gradient
//                              that the step overlaps with these layers.  The reference net's layers stay where they were.
//   ResNet-shaped, batch 64    forward 83.7 / 107 (56-wide), 88.6 / 132 (14-wide); data gradient + ReLU' 102 / 104, 161 / 168 alone
constexpr S2Inst kInst[] = {
    // reference net behind its first block (alexnet.cpp:17-29), batch 256: 16 -> 32 @ 55, 32 -> 64 @ 27, 64 -> 128 @ 13, pad 0
    {0, 55, 55, 0, 2, 14, 1, 8, 0}, {0, 27, 27, 0, 4, 13, 1, 8, 0}, {0, 13, 13, 0, 4, 6, 2, 16, 0},
    {1, 55, 55, 0, 1, 14, 1, 16, 0}, {1, 27, 27, 0, 2, 14, 1, 16, 0}, {1, 13, 13, 0, 4, 7, 1, 16, 0},
    // stage entries of the ResNet-shaped stack, batch 64: 64 -> 128 @ 56, 128 -> 256 @ 28, 256 -> 512 @ 14, pad 1
    {0, 56, 56, 1, 4, 7, 1, 8, 1}, {0, 28, 28, 1, 4, 14, 1, 8, 1}, {0, 14, 14, 1, 4, 7, 2, 16, 1},
    {1, 56, 56, 1, 4, 4, 1, 16, 0}, {1, 28, 28, 1, 2, 14, 1, 16, 1}, {1, 14, 14, 1, 4, 7, 1, 16, 1},
};
constexpr int kNumInst = (int)(sizeof(kInst) / sizeof(kInst[0]));

template <int I>
struct InstGeom {
    using type = S2Geom<kInst[I].mode, kInst[I].hi, kInst[I].wi, kInst[I].pad, kInst[I].wm, kInst[I].rpu, kInst[I].pk, kInst[I].ck>;
};

template <int I>
void fill_plan(S2Plan* pl, int B) {
    using G = typename InstGeom<I>::type;
    pl->mt = G::MT;
    pl->qw = G::QW;
    pl->ck = G::CK;
    pl->lds = G::lds_bytes;
    pl->p.units_total = kInst[I].pk > 1 ? (B + kInst[I].pk - 1) / kInst[I].pk : B * G::NRB;
}

template <int I>
int launch_inst(const S2Plan& pl, const char* tag, const cnn_conv2d_desc* d, hipStream_t s) {
    using G = typename InstGeom<I>::type;
    constexpr S2Inst c = kInst[I];
    auto kern = conv_s2_kernel<c.mode, c.hi, c.wi, c.pad, c.wm, c.rpu, c.pk, c.ck>;
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
    const OptVal e = CNN_OPT_VAL("CONV_S2");  // 0: never; 2: every instance (tests, measurements); default: the instances marked dflt
    if (e && atoi(e) == 0) return false;
    const bool all = e && atoi(e) == 2;
    if (d->k != 3 || d->s != 2 || d->B < 1) return false;
    int inst = -1;
    for (int i = 0; i < kNumInst; ++i)
        if (kInst[i].mode == mode && kInst[i].hi == d->H && kInst[i].wi == d->W && kInst[i].pad == d->pad && (all || kInst[i].dflt)) inst = i;
    if (inst < 0) return false;
    const int C = mode == 0 ? d->Ci : d->Co, M = mode == 0 ? d->Co : d->Ci;
    if (C < 8 || M < 8 || (long long)C * M * 9 >= (1ll << 27)) return false;
    const int Ho = (d->H + 2 * d->pad - 3) / 2 + 1, Wo = (d->W + 2 * d->pad - 3) / 2 + 1;
    if ((long long)d->B * d->Ci * d->H * d->W >= (1ll << 29) || (long long)d->B * d->Co * Ho * Wo >= (1ll << 29)) return false;
    S2Params& p = pl->p;
    p.B = d->B; p.C = C; p.M = M;
    pl->inst = inst;
    fill_any(inst, pl, d->B, std::make_integer_sequence<int, kNumInst>());
    p.nchunk = (C + pl->ck - 1) / pl->ck;
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
    pl->wt_floats = (size_t)pl->ntiles * p.nchunk * pl->ck * 9 * pl->qw;
    p.dbg = CNN_MEASURE_INT("S2_DBG", 0);
    return true;
}

}  // namespace

namespace cnn_amd {

// what conv_rows.hip's public entry points need to know to serve a stride-2 layer through the same interface
bool s2_info(const cnn_conv2d_desc* d, int mode, int* mt, int* qw, int* ck, int* nchunk, int* ntiles, size_t* wt_floats) {
    S2Plan pl;
    if (!make_s2_plan(d, mode, &pl)) return false;
    *mt = pl.mt; *qw = pl.qw; *ck = pl.ck; *nchunk = pl.p.nchunk; *ntiles = pl.ntiles; *wt_floats = pl.wt_floats;
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
