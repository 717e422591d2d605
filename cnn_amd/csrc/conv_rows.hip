// conv_rows.hip -- Conv2D forward (cpu/src/conv2d.cpp:69-92) and data gradient (conv2d.cpp:168-199) of 3x3 / stride-1 layers with wide
// planes as ONE kernel family built like conv_wgrad_sp.hip (round 5): both operands staged through LDS by buffer-addressed DMA with
// out-of-range zero fill, every LDS address a per-lane base + a compile-time immediate, the stage's code one basic block.
//
//     y[b][co][r][c] = bias[co] + sum_{ci,kx,ky} w[co][ci][kx][ky] * x[b][ci][r + kx - pad][c + ky - pad]
// The data gradient of a pad-p layer is the same sum over dy with padding 2 - p, the filter transposed and flipped (prepared once per
// call by rows_prep into the layout the kernel stages: [co tile][channel chunk][channel][tap][co]).
//
// GEMM view: M = output channels (A = filters), N = pixels (B = input), K = (channel, tap), on v_mfma_f32_16x16x4_f32 -- 16-pixel
// column blocks waste 1.8 % of a 110- or 112-wide row where 32-pixel blocks waste 14 %.  A workgroup of four waves owns MT output
// channels x RG output rows of one sample: MT = 128: waves 4 (co) x 1, RG = 2;  MT = 64: waves 2 (co) x 2 (row pairs), RG = 4.  A wave:
// 32 co (two 16-row A blocks) x 2 rows x NB pixel blocks = 4 * NB accumulators of 4 registers.  A STAGE = 8 input channels: their RG + 2
// input rows ([channel][row][W], rows as they lie in HBM -- no halo columns: the two taps that leave a row on the left / right are
// selected away on the lanes concerned; rows above / below the image likewise, wave-uniform) and the 8 x 9 x MT filter block.  Two
// buffers; the DMA of stage t+1 is issued between the MFMAs of stage t; a workgroup walks a contiguous range of (sample, row block)
// units and the channel chunks of each, so prologue and stores overlap the next unit's staging.
// Plane strides = 16 (mod 32): the four k-groups of a 16x16x4 operand read (two per 32-lane LDS access) hit disjoint bank halves.
#include <cstdlib>

#include "common.h"
#include "rows_common.h"

using namespace cnn_amd;

namespace {

struct RowsParams {
    const float* x;     // input tensor [B][C][H][WI]
    const float* wt;    // prepared filters [co tile][chunk][8][9][QW]
    const float* bias;  // nullable (data gradient)
    float* y;           // output tensor [B][M][HO][WO] (nullable when y_relu is given: only the ReLU output is wanted)
    float* y_relu;      // nullable: the output of the ReLU layer behind this one (relu.cpp:25), written by the same pass
    const float* relu_below;  // nullable (data gradient): output of the ReLU layer in front -- its backward pass (relu.cpp:37) on the way out
    int B, C, H, M;     // C = reduction channels, M = output channels
    int HO;
    int nchunk;         // ceil(C / 8)
    int nrb;            // row blocks per sample
    int units_total, units_per_block;
    int dbg;
};

constexpr int kCK = 8;

// RSEL: rows of the staged block that lie below the image hold whatever follows the plane in memory and are selected away on the B operand
//       (wave-uniform selects, one per B value); without it they are staged as zeros like the rows above the image -- possible when the
//       first row below the image starts on a 16-byte unit of the LDS plane (the host checks: always for rows of 4k floats)
// SR:   narrow planes (56, 28 wide; pad 1, so that input and output rows have one pitch): SR consecutive rows form one SUPER-ROW of
//       SR * W = 112 pixels -- the pixel index of a 16-pixel block is flat across them (a tap is one constant offset in the staged plane
//       whatever the row), only the lanes that sit on a row's first / last column differ, by compile-time lane masks
// RW:   super-rows per wave
// WP:   2 = the two waves of a 32-channel co block split the super-row's pixel blocks (whole 14x14 planes as ONE super-row of 196 pixels =
//       13 blocks: a 64-channel tile of 2 (co) x 2 (pixel halves) waves keeps 256 workgroups busy at batch 64)
// MA:   16-channel co blocks per wave (2; 1 = a wave of 16 channels: four co waves on a 64-channel tile)
// PK:   > 1 = PACKED PLANES (7x7, pad 1): the whole planes of PK consecutive samples form one super-row of PK * 49 pixels (2 samples: 98
//       pixels = 7 blocks, 12.5 % idle lanes).  Staged per channel as [8 guard][plane 0: 49 + 3][plane 1: 49 + 3][guard] -- every plane
//       on a 16-byte unit -- with NO halo rows: a tap that leaves its plane at the top / bottom / left / right is selected away by a
//       compile-time lane mask per (block, tap) (what lies there is the neighbouring plane, the pad or the guard).
// CK:   input channels per stage (8; 16 where a stage of 8 would be short against its fixed costs: the 7x7 instance, 7 tiles per wave)
// PROD: a FIFTH wave does nothing but the stage DMA (round 6).  A wave issues one 1-KiB LDS-DMA instruction per ~100 cycles however idle the
//       memory system is (NB section 9: tools/micro/dma_issue.hip), and while it does, it issues no MFMA: with the small planes' large filter
//       share of a stage (7x7: 54 instructions per 252 MFMAs per wave) that was 10 - 17 % of the stage.  The MFMA waves of such an instance
//       never touch the vector-memory queue inside the stage loop.  Needs two waves on one SIMD: instances with <= 256 registers only.
template <int WI, int PAD, int MT, bool RSEL = false, int SR = 1, int RW_ = 2, int WP = 1, int MA_ = 2, int PK_ = 1, int CK_ = 8, bool PROD_ = false>
struct RowsGeom {
    static constexpr bool PROD = PROD_;
    static constexpr int NDW = PROD ? 1 : 4;  // waves that issue the stage DMA
    static constexpr int CK = CK_;
    static_assert(CK == 8 || CK == 16, "channels per stage");
    static_assert(MT == 128 || MT == 64, "output channels per workgroup");
    static constexpr int MA = MA_, PK = PK_;
    static constexpr int WO = WI + 2 * PAD - 2;
    static_assert(SR == 1 || (WO == WI && !RSEL), "super-rows: equal pitches, zero-staged halo rows");
    static_assert(PK == 1 || (PAD == 1 && SR == 1 && !RSEL && RW_ == 1 && WP == 1 && MT == 16 * MA * 4), "packed planes: one super-row per workgroup");
    static constexpr int PLANE = WI * WI, SP = (PLANE + 3) / 4 * 4, GUARD = (WI + 1 + 3) / 4 * 4;  // (packed planes)
    static constexpr int PX = PK > 1 ? PK * PLANE : SR * WO;  // pixels of a super-row
    static constexpr int NB = (PX + 15) / 16;             // 16-pixel blocks per super-row
    static constexpr int WM = MT / (16 * MA), WR = 4 / (WM * WP);  // waves over co x pixel halves x row groups
    static_assert(WM * WP * WR == 4, "four waves");
    static constexpr int NBW = (NB + WP - 1) / WP;        // pixel blocks per wave
    static constexpr int RW = RW_, RG = WR * RW;          // super-rows per wave / per workgroup
    static constexpr int ROWS = PK > 1 ? WI : RG * SR;    // output rows per workgroup unit
    static constexpr int XR = ROWS + 2;                   // staged input rows
    // packed planes: lanes n of block nb whose tap (kx, ky) leaves the plane of its pixel
    static constexpr unsigned tapmask(int nb, int kx, int ky) {
        unsigned m = 0;
        for (int n = 0; n < 16; ++n) {
            const int f = (16 * nb + n) % PLANE, r = f / WI + kx - 1, c = f % WI + ky - 1;
            if (r < 0 || r >= WI || c < 0 || c >= WI) m |= 1u << n;
        }
        return m;
    }
    // lanes n of block nb whose tap column ky leaves the row: column (16 nb + n) % WI + ky - PAD outside [0, WI)
    static constexpr unsigned colmask(int nb, int ky) {
        unsigned m = 0;
        for (int n = 0; n < 16; ++n) {
            const int f = 16 * nb + n, col = SR > 1 ? f % WI : f, c = col + ky - PAD;
            if (c < 0 || c >= WI) m |= 1u << n;
        }
        return m;
    }
    static constexpr int LEAD = (4 - (PAD * WI) % 4) % 4;  // row 0 of the image on a 16-byte unit boundary of the first row block
    // (the buffer descriptor starts PAD*WI + 4 floats in front of the tensor and every lane offset is a multiple of 16 bytes from it: image
    //  row 0 must then sit on a 16-byte unit of the staged plane.  Out-of-descriptor lanes moving ZEROS is gfx950 behaviour
    //  (tools/probes/buflds_probe.cpp); the library is built for gfx950 only and cnn_amd_device_arch() is checked by every caller.)
    static_assert((PAD * WI + LEAD) % 4 == 0, "row 0 of the image on a 16-byte unit of the staged plane");
    static constexpr int XSPAN = PK > 1 ? 2 * GUARD + PK * SP : LEAD + XR * WI;
    static constexpr int QXP = stride16(XSPAN + 4);       // x plane stride (floats)
    static constexpr int QW = MT + 16;                    // filter row stride: 9 * QW = 16 (mod 32)
    static_assert(QXP % 32 == 16 && (9 * QW) % 32 == 16, "bank halves");
    static constexpr int XIMG = CK * QXP, WIMG = CK * 9 * QW;
    static constexpr int NIX = (XIMG / 4 + 63) / 64, NIWT = (WIMG / 4 + 63) / 64;  // DMA instructions per stage
    static constexpr int NIWX = (NIX + NDW - 1) / NDW, NIWW = (NIWT + NDW - 1) / NDW;  // per issuing wave
    static constexpr int NSLOT = NIWX + NIWW;
    static constexpr int XS = NIX * 256, WS = NIWT * 256;  // image sizes in whole instructions
    static constexpr int BUF = XS + WS;
    static constexpr int DUMP = 2 * BUF;
    static constexpr size_t lds_bytes = (size_t)(2 * BUF + 4 * 256) * sizeof(float);
    static_assert(lds_bytes <= 160 * 1024, "LDS plan");
    static constexpr int KSTEPS = CK / 4;  // MFMA k-steps per tap
    static constexpr int THREADS = PROD ? 320 : 256;
};

template <int WI, int PAD, int MT, bool RSEL, int SR, int RW_, int WP, int MA, int PK, int CK, bool PROD>
__global__ __launch_bounds__(PROD ? 320 : 256) void conv_rows_kernel(const RowsParams p) {
    using G = RowsGeom<WI, PAD, MT, RSEL, SR, RW_, WP, MA, PK, CK, PROD>;
    constexpr int WO = G::WO, NB = G::NBW;  // (NB: the pixel blocks of THIS wave)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % G::WM, wp = (wave / G::WM) % WP, wr = wave / (G::WM * WP);
    const int co0 = blockIdx.y * MT;
    const int H = p.H, HWI = H * WI, HWO = p.HO * WO;

    const int u_lo = blockIdx.x * p.units_per_block;
    const int u_hi = u_lo + p.units_per_block < p.units_total ? u_lo + p.units_per_block : p.units_total;
    if (u_lo >= u_hi) return;

    // ---- this wave's share of a stage's x DMA, decoded once: byte offset from (first channel of the chunk, staged row 0) counted from
    //      PAD*WI + 4 floats in front of the tensor (xrs), and the last staged row the unit touches (units of rows above the image of the
    //      first row block are not fetched: for the first plane of the tensor they lie in front of the allocation)
    const int dw = PROD ? 0 : wave;  // this wave among the waves that issue the stage DMA (PROD: the fifth wave alone)
    unsigned xd_off[G::NIWX];
    int xd_row[G::NIWX];   // last staged row the unit touches | first one << 8
#pragma unroll
    for (int i = 0; i < G::NIWX; ++i) {
        const int j = i * G::NDW + dw, q = j * 64 + lane;
        const int plane = q / (G::QXP / 4), e = q - plane * (G::QXP / 4);
        const bool have = j < G::NIX && plane < CK && e * 4 < G::XSPAN;
        if constexpr (PK > 1) {
            // packed planes: unit e of the staged channel = four floats of plane (e*4 - GUARD) / SP, from float (e*4 - GUARD) % SP on (the
            // last unit of a plane carries its 49th float and three of whatever follows it in memory: the pad of the staged plane)
            const int l = e * 4 - G::GUARD, sp = l >= 0 ? l / G::SP : -1;
            const bool in = have && l >= 0 && sp < PK;
            xd_off[i] = in ? (unsigned)((sp * p.C + plane) * G::PLANE + (l - sp * G::SP)) * 4u : kOob;
            xd_row[i] = 0;
        } else {
        xd_off[i] = have ? (unsigned)(plane * HWI + e * 4 - G::LEAD + 4) * 4u : kOob;
        const int first = e * 4 - G::LEAD < 0 ? 0 : (e * 4 - G::LEAD) / WI;
        xd_row[i] = (((e + 1) * 4 - 1 - G::LEAD) / WI) | (first << 8);
        }
    }
    constexpr int BACK = PK > 1 ? 0 : PAD * WI + 4;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x - BACK), 0, (int)(((unsigned)p.B * p.C * HWI + BACK) * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.wt + (size_t)blockIdx.y * p.nchunk * G::WIMG), 0, (int)((unsigned)p.nchunk * G::WIMG * 4u), 0x00020000);

    float* const dump = smem + G::DUMP + dw * 256;
    // slot k of the DMA of stage (sample b, first output row r0, chunk cc) into `buf` = [x image][filter image]
    auto dma_slot = [&](int k, int b, int r0, int cc, float* buf) {
        if (k < G::NIWX) {
            const int j = k * G::NDW + dw;
            float* d = j < G::NIX ? buf + j * 256 : dump;
            const int nneg = PAD - r0;      // staged rows above the image
            const int nv = H + PAD - r0;    // first staged row below it
            unsigned voff = xd_off[k];
            if constexpr (PK == 1) {
                voff = ((xd_row[k] & 255) < nneg) ? kOob : voff;
                if (!RSEL && PAD > 0) voff = ((xd_row[k] >> 8) >= nv) ? kOob : voff;
            }
            blds16(xrs, voff, (unsigned)((b * p.C + cc * CK) * HWI + r0 * WI) * 4u, d);
        } else {
            const int i = k - G::NIWX, j = i * G::NDW + dw;
            float* d = j < G::NIWT ? buf + G::XS + j * 256 : dump;
            const unsigned q = (unsigned)(j * 64 + lane);
            blds16(wrs, (j < G::NIWT && q * 4 < (unsigned)G::WIMG) ? q * 16u : kOob, (unsigned)cc * (unsigned)(G::WIMG * 4), d);
        }
    };

    if constexpr (PROD) {
        if (wave == 4) {
            // ---- the producer: the MFMA waves' walk over (unit, chunk) stages, with nothing in it but the DMA of the stage behind, the wait
            //      for it and the stage barrier.  Stage t + 1 is issued right behind barrier t (every MFMA wave has left the buffer it goes to)
            //      and has landed (vmcnt(0), this wave's own instructions: all of them) before this wave arrives at barrier t + 1.
            int b = u_lo / p.nrb, r0 = (u_lo - b * p.nrb) * G::ROWS;
            if constexpr (PK > 1) b *= PK;
#pragma unroll
            for (int k = 0; k < G::NSLOT; ++k) dma_slot(k, b, r0, 0, smem);
            int t = 0;
            for (int u = u_lo; u < u_hi; ++u) {
                int bu = b, r0u = r0;
                if (u + 1 < u_hi) {
                    if (r0 + G::ROWS < p.HO) r0u = r0 + G::ROWS;
                    else { r0u = 0; bu = b + PK; }
                }
                for (int cc = 0; cc < p.nchunk; ++cc, ++t) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    const bool last_cc = cc + 1 == p.nchunk;
                    if (last_cc && u + 1 == u_hi) break;  // (nothing behind the last stage)
                    float* nxt = smem + ((t + 1) & 1) * G::BUF;
                    const int bn = last_cc ? bu : b, r0n = last_cc ? r0u : r0, ccn = last_cc ? 0 : cc + 1;
#pragma unroll
                    for (int k = 0; k < G::NSLOT; ++k) dma_slot(k, bn, r0n, ccn, nxt);
                }
                b = bu; r0 = r0u;
            }
            return;
        }
    }

    // ---- per-lane operand bases (floats inside a buffer)
    const int b_base = PK > 1 ? kq * G::QXP + G::GUARD - WI - 1 + n : kq * G::QXP + G::LEAD + n - PAD + wr * G::RW * SR * WI + wp * G::NBW * 16;
    const int a_base = G::XS + kq * 9 * G::QW + wm * (16 * MA) + n;

    f32x4 acc[MA][G::RW][NB];
    auto zero_acc = [&]() {
#pragma unroll
        for (int ma = 0; ma < MA; ++ma)
#pragma unroll
            for (int rw = 0; rw < G::RW; ++rw)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[ma][rw][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    // Two nested loops -- units, then the channel chunks of a unit -- so that the stage loop carries the accumulators and nothing else
    // happens to them inside it: written as ONE flat loop over (unit, chunk) stages with the store code behind an `if (last chunk)`, the
    // compiler kept the accumulators in VGPRs across the back edge and copied all of them into AGPRs and back around every stage's MFMAs
    // (224 v_accvgpr_read + 224 v_accvgpr_write per 504 MFMAs in the round-5 ISA).
    int b = u_lo / p.nrb, r0 = (u_lo - b * p.nrb) * G::ROWS;
    if constexpr (PK > 1) b *= PK;  // (packed planes: a unit = PK samples, nrb = 1)
    if constexpr (!PROD) {
#pragma unroll
        for (int k = 0; k < G::NSLOT; ++k) dma_slot(k, b, r0, 0, smem);
    }
    // the k-steps of a stage: (tap, 4-channel group) pairs; slots are spread over them
    constexpr int NKS = 9 * G::KSTEPS;
    constexpr int PER_KS = (G::NSLOT + NKS - 1) / NKS;
    int t = 0;  // stages so far: buffer parity
    const long long dbg_c0 = p.dbg == 9 ? clock64() : 0, dbg_w0 = p.dbg == 9 ? wall_clock64() : 0;  // (ROWS_DBG=9: block 0 prints its shader clock)
    for (int u = u_lo; u < u_hi; ++u) {
        // the unit behind this one (behind the last one: this one again)
        int bu = b, r0u = r0;
        if (u + 1 < u_hi) {
            if (r0 + G::ROWS < p.HO) r0u = r0 + G::ROWS;
            else { r0u = 0; bu = b + PK; }
        }
        for (int cc = 0; cc < p.nchunk; ++cc, ++t) {
            // this wave's share of the stage has landed.  Stage 0 of a unit behind the first: already waited for in front of the previous unit's
            // stores -- a wait here would also wait for THOSE to drain (vmcnt counts stores; measured on the north-star forward, where all 256
            // workgroups store 29 MB in one burst every 8 stages: 5.6 us per unit, 8 % of the kernel); they drain under this stage's MFMAs and
            // the next stage's wait finds them gone
            if (!PROD && (cc != 0 || u == u_lo)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (PROD: the producer waits; this wave only has stores out)
            if constexpr (PROD) __builtin_amdgcn_s_barrier();  // (the bare barrier: __syncthreads()'s fence would wait for the previous unit's stores)
            else __syncthreads();
            float* nxt = smem + ((t + 1) & 1) * G::BUF;
            // the stage behind this one (behind the last one of the range: that one again)
            const bool last_cc = cc + 1 == p.nchunk;
            const int bn = last_cc ? bu : b, r0n = last_cc ? r0u : r0, ccn = last_cc ? (u + 1 < u_hi ? 0 : cc) : cc + 1;
            // rows of the staged block that lie outside the image (wave-uniform; only with padding)
            unsigned rowbad = 0;
            if (PAD > 0 && RSEL) {  // (rows above the image are staged as zeros in either variant)
#pragma unroll
                for (int i = 0; i < G::XR; ++i) rowbad |= (r0 - PAD + i >= H ? 1u : 0u) << i;
            }
            // operands of one k-step (tap, 4-channel group): MA A values, RW * NB B values.  (round 6) Software-pipelined BY HAND: the reads of
            // k-step ks + 2 are issued as inline assembly one or two at a time BETWEEN the MFMAs of k-step ks -- with one wave per SIMD the MFMA
            // pipe only stays busy while this wave issues an MFMA every 32 cycles, and the compiler's placement (every ds_read of a k-step in one
            // clump in front of its MFMAs, pinned there by an empty asm) left it idle for the 60 - 100 cycles a clump takes to issue.  LDS
            // returns in order: before the MFMAs of k-step ks only the reads of batch ks + 1 may be in flight (s_waitcnt lgkmcnt: 4 bits).
            const unsigned par = (unsigned)(t & 1) * (unsigned)(G::BUF * 4);
            const unsigned bad0 = (unsigned)(b_base * 4) + par, aad = (unsigned)(a_base * 4) + par;
            struct Ops {
                float a[MA];
                float b[G::RW][NB];
            };
            constexpr int NRD = MA + G::RW * NB;  // reads of a batch
            // look-ahead of the operand ring: two k-steps -- or one where a batch is large (the tall units of the batch-64 layers: 50 reads = 49
            // MFMAs = 1500 cycles of cover for the LDS latency, and three batches of 50 registers would not fit beside 196 accumulators)
            constexpr int LA = NRD > 32 ? 1 : 2;
            Ops ops[LA + 1];
            auto issue_read = [&](auto KS, auto R) {
                constexpr int ks = decltype(KS)::value, r = decltype(R)::value;
                constexpr int tap = ks / G::KSTEPS, s = ks % G::KSTEPS, kx = tap / 3, ky = tap % 3;
                Ops& o = ops[ks % (LA + 1)];
                if constexpr (r < MA) {
                    lds_rd<((s * 36 + tap) * G::QW + r * 16) * 4>(o.a[r], aad);
                } else {
                    constexpr int rw = (r - MA) / NB, nb = (r - MA) % NB;
                    constexpr int imm = s * 4 * G::QXP + (rw * SR + kx) * WI + 16 * nb + ky;
                    if constexpr (PK > 1) {
                        // (packed planes: the pad between two staged planes moves the pixels of the later planes; per lane in a block that straddles)
                        constexpr int PAD3 = G::SP - G::PLANE;
                        constexpr int lo = (16 * nb) / G::PLANE, hi = (16 * nb + 15) / G::PLANE;
                        if constexpr (lo == hi) lds_rd<(imm + PAD3 * lo) * 4>(o.b[rw][nb], bad0);
                        else lds_rd<imm * 4>(o.b[rw][nb], bad0 + (unsigned)(PAD3 * ((16 * nb + n) / G::PLANE)) * 4u);
                    } else {
                        lds_rd<imm * 4>(o.b[rw][nb], bad0);
                    }
                }
            };
            auto issue_batch = [&](auto KS) { static_for<NRD>([&](auto R) { issue_read(KS, R); }); };
            issue_batch(std::integral_constant<int, 0>());
            if constexpr (LA == 2) issue_batch(std::integral_constant<int, 1>());
            static_for<NKS>([&](auto KS) {
                constexpr int ks = decltype(KS)::value;
                constexpr int tap = ks / G::KSTEPS, kx = tap / 3, ky = tap % 3;
                constexpr int nm = MA * G::RW * NB;               // MFMAs of a k-step
                constexpr int nr2 = ks + LA < NKS ? NRD : 0;      // reads of batch ks + LA, issued here
                constexpr int rpm = (nr2 + nm - 1) / nm;
                if constexpr (!PROD) {
#pragma unroll
                    for (int k = ks * PER_KS; k < (ks + 1) * PER_KS && k < G::NSLOT; ++k) dma_slot(k, bn, r0n, ccn, nxt);
                }
                lgkm_wait<(LA == 2 && ks + 1 < NKS ? (NRD < 15 ? NRD : 15) : 0)>();
                Ops& o = ops[ks % (LA + 1)];
                // (the values of this batch are defined from here on: nothing that uses them may be scheduled above the wait)
#pragma unroll
                for (int ma = 0; ma < MA; ++ma) asm volatile("" : "+v"(o.a[ma]));
#pragma unroll
                for (int rw = 0; rw < G::RW; ++rw) {
                    const bool bad = PAD > 0 && RSEL && ((rowbad >> (wr * G::RW + rw + kx)) & 1u);  // (RSEL: SR == 1)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        float bv = o.b[rw][nb];
                        asm volatile("" : "+v"(bv));
                        // the tap columns that leave their row (a select: what lies there is the neighbouring row's data)
                        constexpr unsigned kAll = 0xffffu;
                        if constexpr (WP == 1) {
                            const unsigned cm = PK > 1 ? G::tapmask(nb, kx, ky) : G::colmask(nb, ky);
                            if (cm == kAll) bv = 0.f;
                            else if (cm != 0) bv = ((cm >> n) & 1u) ? 0.f : bv;
                        } else {
                            const unsigned c0 = G::colmask(nb, ky), c1 = G::colmask(G::NBW + nb, ky);
                            if (c0 != 0 || c1 != 0) bv = (((wp ? c1 : c0) >> n) & 1u) ? 0.f : bv;
                            (void)kAll;
                        }
                        if (PAD > 0 && RSEL) bv = bad ? 0.f : bv;
                        o.b[rw][nb] = bv;
                    }
                }
                static_for<nm>([&](auto IM) {
                    constexpr int im = decltype(IM)::value, rw = im / (NB * MA), nb = (im / MA) % NB, ma = im % MA;
                    mfma16(acc[ma][rw][nb], o.b[rw][nb], o.a[ma]);
                    if constexpr (nr2 > 0)
                        static_for<rpm>([&](auto J) {
                            constexpr int r = im * rpm + decltype(J)::value;
                            if constexpr (r < nr2) issue_read(std::integral_constant<int, (ks + LA < NKS ? ks + LA : 0)>(), std::integral_constant<int, r>());
                        });
                });
            });
        }
        if constexpr (!PROD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the next unit's stage 0, issued early in the last stage: see above)
        acc_settle<MA * G::RW * NB>(&acc[0][0][0]);
        if (p.dbg != 1) {  // (ROWS_DBG=1, measurement only: no stores)
            // ---- this unit is complete: + bias, store.  The MFMAs ran with the PIXELS as the M operand (D[i][j]: lane (j = n, kq) holds rows
            //      i = 4 kq + r): a lane's four registers of a tile are four CONSECUTIVE pixels of output channel n -- one 16-byte store
            //      (and one 16-byte load of the ReLU' mask) per tile instead of four scattered dwords: 28 instead of 112 store instructions
            //      per wave and unit (measured on the north-star forward: the stores were 15 % of the kernel).  The mask values of one
            //      16-channel block are fetched as ONE batch of independent loads before any of them is used.
            if constexpr (PK > 1) {
                // packed planes: pixel f of the super-row = pixel f % 49 of sample b + f / 49; a lane's four pixels are one 16-byte access
                // unless they straddle two planes or leave the batch
#pragma unroll
                for (int ma = 0; ma < MA; ++ma) {
                    const int co = co0 + wm * (16 * MA) + ma * 16 + n;
                    const float bs = (p.bias != nullptr && co < p.M) ? p.bias[co] : 0.f;
                    auto elem = [&](int f, int e, size_t& idx) {
                        const int fe = f + e, se = fe / G::PLANE, pe = fe - se * G::PLANE;
                        idx = ((size_t)(b + se) * p.M + co) * G::PLANE + pe;
                        return co < p.M && fe < G::PX && b + se < p.B;
                    };
                    f32x4 mk[NB];
                    bool whole[NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int f = 16 * nb + 4 * kq, sp = f / G::PLANE, pl = f - sp * G::PLANE;
                        whole[nb] = co < p.M && sp < PK && b + sp < p.B && pl + 3 < G::PLANE;
                        mk[nb] = f32x4{1.f, 1.f, 1.f, 1.f};
                        if (p.relu_below != nullptr) {
                            size_t idx;
                            if (whole[nb]) { (void)elem(f, 0, idx); mk[nb] = *(const f32x4u*)(p.relu_below + idx); }
                            else {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (elem(f, e, idx)) mk[nb][e] = p.relu_below[idx];
                            }
                        }
                    }
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const int f = 16 * nb + 4 * kq;
                        f32x4 v, vr;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc[ma][0][nb][e] + bs;
                            if (p.relu_below != nullptr) v[e] = mk[nb][e] <= 0.f ? 0.f : v[e];
                            vr[e] = v[e] >= 0.f ? v[e] : 0.f;
                        }
                        size_t idx;
                        if (whole[nb]) {
                            (void)elem(f, 0, idx);
                            if (p.y != nullptr) *(f32x4u*)(p.y + idx) = v;
                            if (p.y_relu != nullptr) *(f32x4u*)(p.y_relu + idx) = vr;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (elem(f, e, idx)) {
                                    if (p.y != nullptr) p.y[idx] = v[e];
                                    if (p.y_relu != nullptr) p.y_relu[idx] = vr[e];
                                }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int ma = 0; ma < MA; ++ma) {
                    const int co = co0 + wm * (16 * MA) + ma * 16 + n;
                    const float bs = (p.bias != nullptr && co < p.M) ? p.bias[co] : 0.f;
                    const int bE = p.dbg == 3 ? (int)blockIdx.x : b, r0E = p.dbg == 3 ? 0 : r0;  // (ROWS_DBG=3, experiment: every unit of a workgroup stores to the same place)
                    const size_t cbase = ((size_t)bE * p.M + co) * HWO;
                    // FAST PATH (one 16-channel block of a wave whose super-rows lie inside the image -- every unit but the ragged last ones of a
                    // plane): no per-tile validity, one exec region per block (co < M), the tile offsets compile-time immediates off one per-lane
                    // base.  (The generic path below costs ~40 instructions per tile in compare / saveexec / branch sequences: measured on the
                    // north-star forward with the stores redirected to an L2-resident region, 2/3 of the epilogue's 6 us per unit was not HBM.)
                    bool all_inside = WP == 1;
#pragma unroll
                    for (int rw = 0; rw < G::RW; ++rw) all_inside = all_inside && (p.HO - (r0E + (wr * G::RW + rw) * SR)) * WO >= G::PX;
                    if (all_inside) {  // (wave-uniform)
                        if (co < p.M) {
#pragma unroll
                            for (int rw = 0; rw < G::RW; ++rw) {
                                const size_t rbase = cbase + (size_t)(r0E + wr * G::RW * SR) * WO + 4 * kq;  // this lane's first pixel of the wave's first super-row
                                constexpr int CNT_LAST = G::PX - 16 * (NB - 1);  // pixels of the last block (16: full)
                                const int cnt = CNT_LAST - 4 * kq;               // ... of this lane's four (>= 4: all)
                                auto at = [&](int nb) { return rbase + (size_t)(rw * SR * WO + 16 * nb); };
                                f32x4 mk[NB];
                                if (p.relu_below != nullptr) {
#pragma unroll
                                    for (int nb = 0; nb < NB; ++nb) {
                                        mk[nb] = f32x4{1.f, 1.f, 1.f, 1.f};
                                        if (nb + 1 < NB || CNT_LAST == 16 || cnt >= 4) mk[nb] = *(const f32x4u*)(p.relu_below + at(nb));
                                        else {
#pragma unroll
                                            for (int e = 0; e < 3; ++e)
                                                if (e < cnt) mk[nb][e] = p.relu_below[at(nb) + e];
                                        }
                                    }
                                }
                                f32x4 v[NB];
#pragma unroll
                                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        v[nb][e] = acc[ma][rw][nb][e] + bs;
                                        if (p.relu_below != nullptr) v[nb][e] = mk[nb][e] <= 0.f ? 0.f : v[nb][e];
                                    }
                                auto put = [&](float* dst, bool relu) {
#pragma unroll
                                    for (int nb = 0; nb < NB; ++nb) {
                                        f32x4 o = v[nb];
                                        if (relu) {
#pragma unroll
                                            for (int e = 0; e < 4; ++e) o[e] = o[e] >= 0.f ? o[e] : 0.f;
                                        }
                                        if (nb + 1 < NB || CNT_LAST == 16 || cnt >= 4) *(f32x4u*)(dst + at(nb)) = o;
                                        else {
#pragma unroll
                                            for (int e = 0; e < 3; ++e)
                                                if (e < cnt) dst[at(nb) + e] = o[e];
                                        }
                                    }
                                };
                                if (p.y != nullptr) put(p.y, false);
                                if (p.y_relu != nullptr) put(p.y_relu, true);
                            }
                        }
                        continue;
                    }
                    // (per super-row: the masks of its NB tiles as one batch of loads, then its stores -- RW x NB of them at once would be
                    //  245 registers on the tall units)
#pragma unroll
                    for (int rw = 0; rw < G::RW; ++rw) {
                        f32x4 mk[NB];
                        int nval[NB];
                        const int row0 = r0E + (wr * G::RW + rw) * SR;
                        const int lim0 = (p.HO - row0) * WO, lim = lim0 < G::PX ? lim0 : G::PX;  // pixels of the super-row inside the image
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const int f = 16 * (wp * G::NBW + nb) + 4 * kq;
                            nval[nb] = co < p.M ? lim - f : 0;
                            mk[nb] = f32x4{1.f, 1.f, 1.f, 1.f};
                            if (p.relu_below != nullptr) {
                                const float* m = p.relu_below + cbase + (size_t)row0 * WO + f;
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
                            const int f = 16 * (wp * G::NBW + nb) + 4 * kq;
                            const size_t at = cbase + (size_t)row0 * WO + f;  // (a super-row is PX consecutive floats of y)
                            f32x4 v, vr;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[e] = acc[ma][rw][nb][e] + bs;
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
                }
            }
        }
        zero_acc();
        b = bu; r0 = r0u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.dbg == 9 && threadIdx.x == 0 && (blockIdx.x | blockIdx.y) == 0) {
        const long long c = clock64() - dbg_c0, w = wall_clock64() - dbg_w0;
        printf("conv_rows block 0: %lld shader cycles in %lld ticks of 10 ns -> %.0f MHz, %d stages of %d MFMAs per wave\n", c, w, (double)c / ((double)w / 100.0), t,
               NKS * MA * G::RW * NB);
    }
}

// filters -> [co tile][chunk][channel 0..CK-1][tap][QW]; mode 0: forward (w[co][c][tap]); mode 1: data gradient (w[c][m][8 - tap]); mode 2: the
// stride-2 data gradient (w[c][m][tap]: transposed, not flipped)
struct RowsPrepJob {
    const float* w;
    float* wt;
    int Co, Ci, mode, MT, QW, nchunk, ntiles, CK;
};
__device__ inline void rows_prep_body(const RowsPrepJob& q) {
    // (32-bit index arithmetic: the host checks that the image has fewer than 2^31 elements; in 64 bits the division chain was most of the kernel)
    const unsigned total = (unsigned)q.ntiles * q.nchunk * q.CK * 9 * q.QW;
    const int C = q.mode == 0 ? q.Ci : q.Co, M = q.mode == 0 ? q.Co : q.Ci;
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int j = (int)(i % (unsigned)q.QW);
        unsigned r = i / (unsigned)q.QW;
        const int tap = (int)(r % 9); r /= 9;
        const int cl = (int)(r % q.CK); r /= q.CK;
        const int cc = (int)(r % q.nchunk);
        const int tile = (int)(r / q.nchunk);
        const int m = tile * q.MT + j, c = cc * q.CK + cl;
        float v = 0.f;
        if (j < q.MT && m < M && c < C)
            v = q.mode == 0 ? q.w[((size_t)m * q.Ci + c) * 9 + tap] : q.w[((size_t)c * q.Ci + m) * 9 + (q.mode == 1 ? 8 - tap : tap)];
        q.wt[i] = v;
    }
}
__global__ __launch_bounds__(256) void rows_prep(const RowsPrepJob q) { rows_prep_body(q); }
// the images of several layers / modes in ONE launch (cnn_conv2d_prepare_filters: a ResNet-shaped step re-packs 26 images): blockIdx.y = job
constexpr int kMaxRowsPrepJobs = 12;
struct RowsPrepBatch {
    RowsPrepJob job[kMaxRowsPrepJobs];
};
__global__ __launch_bounds__(256) void rows_prep_batch(const RowsPrepBatch b) { rows_prep_body(b.job[blockIdx.y]); }

struct RowsPlan {
    RowsParams p;
    int wi, pad, mt, qw, ntiles, blocks, rsel, sr, rows, ck, tall;
    size_t wt_floats;
};

// mode 0: forward of d; mode 1: data gradient of d
bool make_rows_plan(const cnn_conv2d_desc* d, int mode, RowsPlan* pl) {
    const OptVal e = CNN_OPT_VAL("CONV_ROWS");
    if (e && atoi(e) == 0) return false;
    if (d->k != 3 || d->s != 1 || d->pad < 0 || d->pad > 1 || d->B < 1) return false;
    const int Ho = d->H + 2 * d->pad - 2, Wo = d->W + 2 * d->pad - 2;
    if (Ho < 1 || Wo < 1) return false;
    const int wi = mode == 0 ? d->W : Wo, hi = mode == 0 ? d->H : Ho, pad = mode == 0 ? d->pad : 2 - d->pad;
    const int C = mode == 0 ? d->Ci : d->Co, M = mode == 0 ? d->Co : d->Ci;
    const int ho = mode == 0 ? Ho : d->H;
    // (the instances below: 112-wide planes with any padding, 56- and 28-wide ones with pad 1 as super-rows of 2 / 4 rows)
    // 14x14 planes with pad 1 whole: one super-row of 196 pixels per workgroup unit
    if (!((wi == 112 && pad <= 1) || (wi == 110 && pad == 2) || ((wi == 56 || wi == 28) && pad == 1) || ((wi == 14 || wi == 7) && hi == wi && pad == 1))) return false;
    if (wi == 28 && M <= 64) return false;  // (28-wide: one super-row of 4 rows per workgroup needs the 4 x 1 wave layout)
    const int ck = wi == 7 ? 16 : kCK;
    pl->ck = ck;
    if (C < 16 || C % ck != 0 || M < 32 || (long long)C * M * 9 >= (1ll << 28)) return false;  // (the filter image is indexed in 32 bits)  // (whole 8-channel chunks: no plane of a stage lies behind the sample's channels)
    if ((long long)d->B * C * hi * wi >= (1ll << 29) || (long long)d->B * M * ho * (wi + 2 * pad - 2) >= (1ll << 31)) return false;
    RowsParams& p = pl->p;
    p.B = d->B; p.C = C; p.H = hi; p.M = M; p.HO = ho;
    p.nchunk = (C + ck - 1) / ck;
    pl->wi = wi; pl->pad = pad;
    // (measured on the north-star forward: the 128-channel tile -- 4 x 1 waves, two rows each -- 117 TFLOP/s; with four rows per wave (224
    // accumulator registers) 95; as two 64-channel tiles of 2 x 2 waves 113)
    pl->mt = (M > 64 && wi != 14 && wi != 7) ? 128 : 64;
    pl->qw = pl->mt + 16;
    pl->ntiles = (M + pl->mt - 1) / pl->mt;
    pl->sr = wi == 56 ? 2 : (wi == 28 ? 4 : (wi == 14 ? 14 : 1));
    int rg = (wi == 14 || wi == 7) ? wi : (pl->mt == 128 ? (wi == 28 ? 1 : 2) : 4) * pl->sr;  // output rows per workgroup unit
    // (round 6) TALL units for the batch-64 layers of the ResNet-shaped stack: 64 samples x 7 units of 8 (4) rows = 448 units leave 32 of
    // 256 CUs without work (two units per workgroup: 224 workgroups).  A 56-wide plane as FOUR units of 14 rows (one wave = 16 channels x
    // 7 super-rows), a 28-wide plane as four units of 7 rows (ONE super-row of 196 pixels = 13 blocks, 6 % idle lanes) are 256 units.  Taken
    // when the estimated time -- units per workgroup x rows x (lanes per live pixel) -- is shorter (never at the VGG-shaped stack's batch 128).
    pl->tall = 0;
    if (const int tr = (wi == 56 && pl->mt == 64) ? 14 : ((wi == 28 && pl->mt == 128) ? 7 : 0); tr && ho % tr == 0 && CNN_OPT_INT("ROWS_TALL", 1) != 0) {
        const int envb = CNN_OPT_INT("ROWS_BLOCKS", 0);
        long long wantb = (envb > 0 ? envb : num_cus()) / pl->ntiles;
        if (wantb < 1) wantb = 1;
        auto cost = [&](int rows, double lane) {
            const long long units = (long long)d->B * ((ho + rows - 1) / rows);
            const long long w = wantb < units ? wantb : units;
            return (double)((units + w - 1) / w) * rows * lane;
        };
        const double lane_tall = wi == 28 ? 208.0 / 196.0 : 1.0;
        if (cost(tr, lane_tall) < cost(rg, 1.0) || CNN_OPT_INT("ROWS_TALL", 1) == 2) {
            pl->tall = 1;
            rg = tr;
            if (wi == 28) pl->sr = 7;
        }
    }
    pl->rows = rg;
    // zero staging of the rows below the image needs the first of them on a 16-byte unit of the plane: staged row (hi + pad - r0), r0 a
    // multiple of rg, lead pad (4 - pad*wi % 4) % 4 in front
    pl->rsel = 0;
    if (pad > 0 && wi != 7) {  // (7x7: no halo rows are staged at all)
        const int lead = (4 - (pad * wi) % 4) % 4;
        for (int r0 = 0; r0 < ho; r0 += rg)
            if (r0 + rg + 2 - pad > hi && ((hi + pad - r0) * wi + lead) % 4 != 0) pl->rsel = 1;
    }
    p.nrb = (ho + rg - 1) / rg;
    p.units_total = wi == 7 ? (d->B + 1) / 2 : d->B * p.nrb;  // (7x7: the planes of two samples per unit)
    // (round 5, before the accumulators stayed in AGPRs: batch-64 forward passes and 14x14 planes were faster on the implicit GEMM and were
    // kept there; measured since, tools/one_layer.py, forward / data gradient in TFLOP/s: 64 x 64 -> 64 @ 56x56 96 / 96 against 85 / 77,
    // 64 x 128 -> 128 @ 28x28 95 / 96 against 80 / 72, 64 x 256 -> 256 @ 14x14 94 / 94 against 73 / 67, 128 x 512 -> 512 @ 14x14 101 / 107
    // against 97 / 101)
    const int env = CNN_OPT_INT("ROWS_BLOCKS", 0);
    long long want = (env > 0 ? env : num_cus()) / pl->ntiles;
    if (want < 1) want = 1;
    if (want > p.units_total) want = p.units_total;
    p.units_per_block = (int)((p.units_total + want - 1) / want);
    pl->blocks = (p.units_total + p.units_per_block - 1) / p.units_per_block;
    pl->wt_floats = (size_t)pl->ntiles * p.nchunk * ck * 9 * pl->qw;
    p.dbg = CNN_MEASURE_INT("ROWS_DBG", 0);
    return true;
}

template <int WI, int PAD, int MT, bool RSEL, int SR = 1, int RW = 2, int WP = 1, int MA = 2, int PK = 1, int CK = 8, bool PROD = false>
int launch_rows2(const RowsPlan& pl, const char* tag, const cnn_conv2d_desc* d, hipStream_t s) {
    using G = RowsGeom<WI, PAD, MT, RSEL, SR, RW, WP, MA, PK, CK, PROD>;
    auto kern = conv_rows_kernel<WI, PAD, MT, RSEL, SR, RW, WP, MA, PK, CK, PROD>;
    if (G::ROWS != pl.rows || CK != pl.ck) return fail(CNN_AMD_E_BADARG, "conv_rows: plan / instance mismatch");
    static DeviceOnce attr_once;
    if (attr_once.needed()) {
        CNN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_once.mark();
    }
    // (measurement switch ROWS_LDS=<bytes>: a larger LDS request, e.g. 90000 = never two workgroups on one CU)
    size_t lds = G::lds_bytes;
    if (const int want = CNN_MEASURE_INT("ROWS_LDS", 0); want > (int)lds && want <= 160 * 1024) lds = (size_t)want;
    char name[48];
    if (pl.tall) snprintf(name, sizeof(name), "conv_rows<%d,%d,%d,r%d>/%s", WI, PAD, MT, G::ROWS, tag);
    else snprintf(name, sizeof(name), "conv_rows<%d,%d,%d>/%s", WI, PAD, MT, tag);
    CNN_KLAUNCH(s, name, (kern<<<dim3(pl.blocks, pl.ntiles), G::THREADS, lds, s>>>(pl.p)), "B%d Ci%d %dx%d Co%d k%d s%d p%d", d->B, d->Ci, d->H, d->W, d->Co,
                d->k, d->s, d->pad);
    return CNN_AMD_OK;
}

template <int WI, int PAD, int MT>
int launch_rows(const RowsPlan& pl, const char* tag, const cnn_conv2d_desc* d, hipStream_t s) {
    if constexpr (PAD > 0 && WI % 4 != 0) {
        if (pl.rsel) return launch_rows2<WI, PAD, MT, true>(pl, tag, d, s);
    }
    return launch_rows2<WI, PAD, MT, false>(pl, tag, d, s);
}

int launch_any(const RowsPlan& pl, const char* tag, const cnn_conv2d_desc* d, hipStream_t s) {
    if (pl.wi == 112 && pl.pad == 0) return pl.mt == 128 ? launch_rows<112, 0, 128>(pl, tag, d, s) : launch_rows<112, 0, 64>(pl, tag, d, s);
    if (pl.wi == 112 && pl.pad == 1) return pl.mt == 128 ? launch_rows<112, 1, 128>(pl, tag, d, s) : launch_rows<112, 1, 64>(pl, tag, d, s);
    if (pl.wi == 110) return pl.mt == 128 ? launch_rows<110, 2, 128>(pl, tag, d, s) : launch_rows<110, 2, 64>(pl, tag, d, s);
    if (pl.wi == 56 && pl.tall) return launch_rows2<56, 1, 64, false, 2, 7, 1, 1>(pl, tag, d, s);  // (4 co waves x 7 super-rows of two rows)
    if (pl.wi == 28 && pl.tall) return launch_rows2<28, 1, 128, false, 7, 1>(pl, tag, d, s);        // (one super-row of seven rows)
    if (pl.wi == 56) return pl.mt == 128 ? launch_rows2<56, 1, 128, false, 2, 2>(pl, tag, d, s) : launch_rows2<56, 1, 64, false, 2, 2>(pl, tag, d, s);
    // the 7x7 instance runs with a producer wave (ROWS_PROD=0: without; =2: the 14x14 instance too).  Measured, block 0's cycles for 258 k
    // cycles of MFMA issue (ROWS_DBG=9, batch 64): 7x7 341 k -> 316 k (168 -> 158 us); 14x14 300 k either way (its 31 instructions per 252
    // MFMAs were not what held it at 86 %), and 1 % slower at batch 128: it stays without.
    const int prod = CNN_OPT_INT("ROWS_PROD", 1);
    if (pl.wi == 14) return prod == 2 ? launch_rows2<14, 1, 64, false, 14, 1, 2, 2, 1, 8, true>(pl, tag, d, s) : launch_rows2<14, 1, 64, false, 14, 1, 2>(pl, tag, d, s);
    if (pl.wi == 7) return prod != 0 ? launch_rows2<7, 1, 64, false, 1, 1, 1, 1, 2, 16, true>(pl, tag, d, s) : launch_rows2<7, 1, 64, false, 1, 1, 1, 1, 2, 16>(pl, tag, d, s);
    return launch_rows2<28, 1, 128, false, 4, 1>(pl, tag, d, s);
}

}  // namespace

namespace cnn_amd {

// conv_rows_s2.hip: the stride-2 sibling is served through the same four entry points (its filter image has the same layout)
bool s2_info(const cnn_conv2d_desc* d, int mode, int* mt, int* qw, int* ck, int* nchunk, int* ntiles, size_t* wt_floats);
int s2_run(const cnn_conv2d_desc* d, int mode, const float* in, const float* image, const float* bias, float* out, float* out_relu,
           const float* relu_below, hipStream_t s);

// conv_rows_any.hip: the runtime-width member of the family takes every 3x3 / stride-1 geometry the instances above do not
bool any_info(const cnn_conv2d_desc* d, int mode, int* mt, int* qw, int* ck, int* nchunk, int* ntiles, size_t* wt_floats);
int any_run(const cnn_conv2d_desc* d, int mode, const float* in, const float* image, const float* bias, float* out, float* out_relu,
            const float* relu_below, hipStream_t s);

// the filter-image job of layer d in `mode`, whichever kernel family serves it
static bool prep_job(const cnn_conv2d_desc* d, int mode, const float* w, float* image, RowsPrepJob* q, size_t* floats) {
    int mt, qw, ck, nchunk, ntiles;
    if (s2_info(d, mode, &mt, &qw, &ck, &nchunk, &ntiles, floats)) {
        // (stride 2, data gradient: filters transposed but NOT flipped -- the kernel picks each parity class's taps by index: mode 2)
        *q = RowsPrepJob{w, image, d->Co, d->Ci, mode == 0 ? 0 : 2, mt, qw, nchunk, ntiles, ck};
        return true;
    }
    RowsPlan pl;
    if (!make_rows_plan(d, mode, &pl)) {
        if (!any_info(d, mode, &mt, &qw, &ck, &nchunk, &ntiles, floats)) return false;
        *q = RowsPrepJob{w, image, d->Co, d->Ci, mode, mt, qw, nchunk, ntiles, ck};
        return true;
    }
    *q = RowsPrepJob{w, image, d->Co, d->Ci, mode, pl.mt, pl.qw, pl.p.nchunk, pl.ntiles, pl.ck};
    *floats = pl.wt_floats;
    return true;
}

// floats of the row kernel's prepared filter image (0: geometry not covered in that mode); mode 0 = forward, 1 = data gradient
size_t rows_workspace_floats(const cnn_conv2d_desc* d, int mode) {
    static thread_local DescMemo memo[2];  // (called on every launch by the dispatch in conv_igemm.hip)
    size_t n = 0;
    if (memo[mode & 1].find(d, &n)) return n;
    RowsPrepJob q;
    if (!prep_job(d, mode, nullptr, nullptr, &q, &n)) n = 0;
    memo[mode & 1].put(d, n);
    return n;
}
// the filter image of layer d in `mode` into `image` (rows_workspace_floats floats, 16-byte aligned)
int rows_prepare(const cnn_conv2d_desc* d, int mode, const float* w, float* image, hipStream_t s) {
    RowsPrepJob q;
    size_t floats = 0;
    if (!prep_job(d, mode, w, image, &q, &floats)) return fail(CNN_AMD_E_BADARG, "conv_rows: geometry not covered");
    CNN_REQUIRE(w && image && (reinterpret_cast<uintptr_t>(image) & 15) == 0, "conv_rows: filter image must be 16-byte aligned");
    CNN_KLAUNCH(s, "rows_prep", (rows_prep<<<stream_grid(floats, 256), 256, 0, s>>>(q)),
                "B%d Ci%d %dx%d Co%d k%d s%d p%d", d->B, d->Ci, d->H, d->W, d->Co, d->k, d->s, d->pad);
    return CNN_AMD_OK;
}
// the same for n (layer, mode) pairs in one launch; entries the row kernel does not cover are an error
int rows_prepare_batch(int n, const cnn_conv2d_desc* const* d, const int* mode, const float* const* w, float* const* image, hipStream_t s) {
    for (int first = 0; first < n; first += kMaxRowsPrepJobs) {
        RowsPrepBatch b;
        const int cnt = n - first < kMaxRowsPrepJobs ? n - first : kMaxRowsPrepJobs;
        size_t most = 0;
        for (int i = 0; i < cnt; ++i) {
            size_t floats = 0;
            if (!prep_job(d[first + i], mode[first + i], w[first + i], image[first + i], &b.job[i], &floats)) return fail(CNN_AMD_E_BADARG, "conv_rows: geometry not covered");
            CNN_REQUIRE(w[first + i] && image[first + i] && (reinterpret_cast<uintptr_t>(image[first + i]) & 15) == 0, "conv_rows: filter image must be 16-byte aligned");
            most = floats > most ? floats : most;
        }
        unsigned gx = (unsigned)((most + 255) / 256);
        if (gx > 2048) gx = 2048;
        CNN_KLAUNCH(s, "rows_prep", (rows_prep_batch<<<dim3(gx, cnt), 256, 0, s>>>(b)), "jobs=%d", cnt);
    }
    return CNN_AMD_OK;
}
// forward (mode 0: in = x, out = y and / or y_relu) or data gradient (mode 1: in = dy, out = dx, relu_below nullable) from a prepared image
int rows_run(const cnn_conv2d_desc* d, int mode, const float* in, const float* image, const float* bias, float* out, float* out_relu,
             const float* relu_below, hipStream_t s) {
    if (d->s == 2) return s2_run(d, mode, in, image, bias, out, out_relu, relu_below, s);
    RowsPlan pl;
    if (!make_rows_plan(d, mode, &pl)) return any_run(d, mode, in, image, bias, out, out_relu, relu_below, s);
    pl.p.x = in; pl.p.wt = image; pl.p.bias = mode == 0 ? bias : nullptr; pl.p.y = out; pl.p.y_relu = out_relu; pl.p.relu_below = relu_below;
    const char* tag = mode == 0 ? (out_relu ? (out ? "fwd+relu" : "fwd,relu") : "fwd") : (relu_below ? "dgrad+relu" : "dgrad");
    return launch_any(pl, tag, d, s);
}

}  // namespace cnn_amd
