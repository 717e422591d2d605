// conv_wgrad_sp.hip -- Conv2D weight / bias gradient (cpu/src/conv2d.cpp:117-159) for 3x3 / stride-1 / pad-1 layers with SMALL
// PLANES (7x7 ... 56x56: the deep layers of the ResNet- / VGG-shaped stacks) as an output-stationary GEMM with both operands
// staged through LDS by DMA:
//     gw[co][ci][kx][ky] = sum_{b,r,c} dy[b][co][r][c] * x[b][ci][r + kx - 1][c + ky - 1]        (bias gradient: sum of dy)
//
// Why another kernel (round 5): the register-direct kernel (conv_wgrad_rd.hip) lets every lane stream its own 16-byte windows from
// L1 -- 64 different cache lines per load instruction, a 32-channel M tile, every x value fetched once per tap and per M tile.  With
// K = B*Ho*Wo in the hundreds of thousands (VGG shapes at batch 128) it reaches 100 TFLOP/s; on the batch-64 layers of the
// ResNet-shaped stack it is bound by the texture path's line rate: 40-45 TFLOP/s (profiles/r04/bench_resnet18_breakdown.txt).
//
// Here a workgroup of four waves (2 x 2) owns a 64 (co) x 64 (ci) x 9 (taps) tile of the gradient -- a wave 32 co x 32 ci x 9 taps
// = nine 32x32 accumulators of v_mfma_f32_32x32x2_f32 -- and walks STAGES of the reduction dimension (pixels).  MFMA column n of the
// tile of tap (kx, ky) is input channel ci0 + n: the B operand of lane (n, kg) is x_lds[ci n][pixel + tap offset], so the 32 lanes of
// a ds_read_b32 hit 32 different channel planes and the nine taps of a pixel are nine reads of ONE staged plane (x is fetched from
// HBM / L2 once per stage and tile, fully coalesced, not once per tap).  The two k-slots of an MFMA step (kg = lane / 32) are two
// pixels a CONSTANT LDS distance apart, so all LDS addresses are one per-lane base + compile-time immediates:
//   * PAIR (7x7 planes): a stage = two samples, kg = sample parity.  The 64-channel block of a sample is ONE contiguous run of
//     64*49 floats in HBM and is copied as it lies (plane stride 49: odd, conflict-free reads); taps that leave the plane are not
//     multiplied at all (361 of 441 MFMAs per sample pair remain -- 18 % of the nominal FLOPs are multiplications by padding).
//   * HALF (14 / 28 / 56 wide planes): a stage = RU output rows of one sample (with their two halo rows of x), kg = left / right half
//     of the row.  The halo row above the image is staged as zeros, the one below it is selected away; the one tap column that leaves
//     the image on the left (right) exists only for the kg = 0 (1) lanes and is masked there by a select.
// A 7-pixel SEGMENT of a row is the unit of the unrolled inner loop: 7 A values, three 9-float windows of x, 63 MFMAs; the next
// segment's operands are read while this one's MFMAs issue.  Two LDS buffers: the DMA of stage s+1 (buffer_load ... lds, 16 bytes per
// lane from 4-byte-aligned sources: rows of 7 or 14 floats are as good as rows of 28) is issued in slices between the MFMAs of stage s;
// one barrier per stage.
// Output: slabs[blockIdx.x][Co][Ci*9 + 1] like the register-direct kernel (reduce_slabs adds the pixel ranges in a fixed order);
// the bias gradient is accumulated on the VALU from the A registers.
#include <cstdlib>
#include <type_traits>

#include "common.h"

using namespace cnn_amd;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;

struct SpParams {
    const float* x;
    const float* dy;
    float* slabs;  // [gridDim.x][Co][pitch]
    int B, Ci, Co, H;
    int Ntot, pitch;  // Ci*9, Ntot + 1 (column Ntot = bias gradient)
    int nrb;          // HALF: row blocks per sample (H / RU)
    int stages_total, stages_per_block;
    int dbg;  // CNN_AMD_SP_DBG=9: workgroup 0 prints its shader-cycle count and the clock it ran at
};

// one LDS-DMA instruction through a buffer descriptor: lane l moves U floats from base + voff(l) + soff to lds + l*U.  A lane whose
// voff lies outside the buffer moves ZEROS (probed on MI355X: tools/probes/buflds_probe.cpp) -- that is how pad floats, channels
// behind the tensor, halo rows outside the image and the missing second sample of an odd batch are staged: every instruction runs
// with all 64 lanes and no branch around it.
constexpr unsigned kOob = 0x80000000u;
template <int U>
__device__ __forceinline__ void blds(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float* lds) {
    if constexpr (U == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_ptr)lds, 16, (int)voff, (int)soff, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_ptr)lds, 4, (int)voff, (int)soff, 0, 0);
}

constexpr int kTile = 64;  // output channels per workgroup tile

// CT = input channels per workgroup tile: 64 (waves 2 x 2 over co x ci), or 32 for planes too wide for two LDS buffers of 64 + 64 channel
//      rows (112-wide: waves 2 x 1 x 2 -- the two waves of a (co, ci) tile split the SEGMENTS of a stage and are added at the end).
// PAD = 1 (x and dy planes of equal size, taps -1 .. +1) or 0 (the reference's own geometry, conv2d.cpp:41-42: dy is W - 2 wide, taps 0 .. 2;
//      dy rows are staged contiguously and the W - WO columns behind a row's end are selected away on the A operand).
template <int W, int RU, bool PAIR, int U, int CT = 64, int PAD = 1>
struct SpGeom {
    static_assert(U == 1 || U == 4, "DMA unit: 4 or 16 bytes");
    static_assert(PAIR ? W % 7 == 0 : W % 14 == 0, "rows are whole 7-pixel segments (per half)");
    static_assert((CT == 64 || CT == 32) && (PAD == 0 || PAD == 1) && !(PAIR && (CT != 64 || PAD != 1)), "tile variants");
    static constexpr int WO = W - 2 * (1 - PAD);       // width of a dy row
    static constexpr int KW = kTile / CT;              // waves that share a (co, ci) tile and split the stage's segments
    static constexpr int HALFW = PAIR ? W : W / 2;     // pixels of a row per k-group
    static constexpr int NSEG = HALFW / 7;             // segments per row and k-group
    static_assert(NSEG % KW == 0, "segments per wave");
    static constexpr int NSEGW = NSEG / KW;            // ... per wave
    static constexpr int XROWS = PAIR ? RU : RU + 2;   // staged input rows per plane (HALF: with the halo rows)
    static constexpr int XLEN = XROWS * W, DLEN = RU * WO;
    // plane strides: odd for 4-byte units (32 lanes = 32 banks); a multiple of 4 with an odd piece count for 16-byte units (4-way)
    static constexpr int stride_for(int len) {
        return U == 1 ? (len | 1) : ((((len + 3) / 4) & 1) ? (len + 3) / 4 * 4 : (len + 3) / 4 * 4 + 4);
    }
    // HALF, 16-byte units: LEAD pad floats in front of every x plane put the first row of the image (staged row 1) on a unit boundary, so
    // that the units of the top halo row hold nothing else and can be staged as zeros for the first row block of a sample (the SOURCE
    // of a 16-byte unit needs no alignment: probed, tools/probes/buflds16_probe.cpp -- rows of 14 floats are as good as rows of 28)
    static constexpr int LEAD = (!PAIR && U == 4 && PAD == 1) ? (4 - W % 4) % 4 : 0;
    static constexpr int XSPAN = LEAD + XLEN;
    static constexpr int QX = PAIR ? XLEN : stride_for(XSPAN);
    static constexpr int QD = PAIR ? DLEN : stride_for(DLEN);
    static_assert(!PAIR || (XLEN & 1), "PAIR: odd plane stride (32 lanes = 32 banks)");
    static constexpr int NU = PAIR ? 2 : 1;            // samples per stage
    // DMA instructions (64 lanes x U floats) per stage
    static constexpr int PPX = QX / U, PPD = QD / U;                     // HALF: pieces per plane (pad pieces included)
    static constexpr int NIX = PAIR ? (kTile * XLEN / U + 63) / 64 : (CT * PPX + 63) / 64;  // per image (PAIR: per sample)
    static constexpr int NID = PAIR ? (kTile * DLEN / U + 63) / 64 : PPD;
    static constexpr int NIWX = (NIX + 3) / 4, NIWD = (NID + 3) / 4;     // per wave
    static constexpr int NIW = NU * (NIWX + NIWD);
    // a sample's image holds whole DMA instructions: the lanes behind the last float of the block write zeros (see blds)
    static constexpr int XS = NIX * 64 * U, DS = NID * 64 * U;           // floats per sample (HALF: == 64 planes)
    static_assert(PAIR || (XS >= CT * QX && DS == kTile * QD), "HALF: the images hold their planes");
    static constexpr int DIMG = NU * DS, XIMG = NU * XS;                 // floats; a buffer = [D image][X image]
    static constexpr int BUF = (DIMG + XIMG + 3) & ~3;
    static constexpr int DUMP = 2 * BUF;               // 4 waves x 64 lanes x U floats behind the buffers: where the DMA slots a wave has no
                                                       // instruction for put their (zero) data
    static constexpr int NSEGS = RU * NSEGW;                            // segments per stage and wave
    static constexpr int PER_SEG = (NIW + NSEGS - 1) / NSEGS;           // DMA slots issued per segment
    static constexpr int OP = CT * 9 + 1;                               // epilogue: LDS row pitch (odd)
    static constexpr size_t epi_bytes = (size_t)(32 * OP + 64) * sizeof(float);
    static constexpr size_t buf_bytes = (size_t)(2 * BUF + 4 * 64 * U) * sizeof(float);
    static constexpr size_t lds_bytes = buf_bytes > epi_bytes ? buf_bytes : epi_bytes;
    static_assert(lds_bytes <= 160 * 1024, "LDS plan");
};

template <int W, int RU, bool PAIR, int U, int CT = 64, int PAD = 1>
__global__ __launch_bounds__(256) void wgrad_sp_kernel(const SpParams p) {
    using G = SpGeom<W, RU, PAIR, U, CT, PAD>;
    constexpr int WO = G::WO;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, m = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = CT == 64 ? (wave & 1) : 0, kw = CT == 64 ? 0 : (wave & 1);
    const int ci0 = blockIdx.y * CT, co0 = blockIdx.z * kTile;
    const int nci = p.Ci - ci0 < CT ? p.Ci - ci0 : CT, nco = p.Co - co0 < kTile ? p.Co - co0 : kTile;
    const int H = p.H, HW = H * W;
    const int HO = H - 2 * (1 - PAD), HWO = HO * WO;  // the dy plane
    const long long dbg_e0 = p.dbg == 9 ? wall_clock64() : 0;

    // everything the DMA never writes (pad floats, planes of channels behind Ci / Co) reads as zero
    for (int i = tid * 4; i < 2 * G::BUF + 4 * 64 * U; i += 1024) *(float4*)(smem + i) = make_float4(0.f, 0.f, 0.f, 0.f);

    const int s_lo = blockIdx.x * p.stages_per_block;
    const int s_hi = s_lo + p.stages_per_block < p.stages_total ? s_lo + p.stages_per_block : p.stages_total;

    // ---- HALF: this wave's share of a stage's DMA, decoded once: the lane's byte offset from (channel ci0 / co0, first staged row) of
    //      the sample -- for x counted from W + 4 floats in front of it, see xrs --, bit 0 set: the unit holds nothing but the top halo
    //      row (+ lead pad).  kOob: nothing to move (pad unit / channel behind the tensor / no instruction for this wave).  A unit that
    //      runs over the end of its plane's rows fetches what follows them in memory (or zeros behind the tensor) into pad floats.
    unsigned dx_desc[PAIR ? 1 : G::NIWX], dd_desc[PAIR ? 1 : G::NIWD];
    if constexpr (!PAIR) {
#pragma unroll
        for (int i = 0; i < G::NIWX; ++i) {
            const int j = i * 4 + wave, q = j * 64 + lane;
            const int plane = q / G::PPX, e = q - plane * G::PPX;
            dx_desc[i] = (j < G::NIX && e * U < G::XSPAN && plane < nci)
                             ? ((unsigned)(plane * HW + e * U + 4 - G::LEAD) * 4u) | ((PAD == 1 && (e + 1) * U <= G::LEAD + W) ? 1u : 0u) : kOob;
        }
#pragma unroll
        for (int i = 0; i < G::NIWD; ++i) {
            const int j = i * 4 + wave, q = j * 64 + lane;
            const int plane = q / G::PPD, e = q - plane * G::PPD;
            dd_desc[i] = (j < G::NID && e * U < G::DLEN && plane < nco) ? (unsigned)(plane * HWO + e * U) * 4u : kOob;
        }
    }
    (void)dx_desc; (void)dd_desc;
    __syncthreads();
    // x is addressed from W + 4 floats in front of the tensor (never fetched: the units of the top halo row of row block 0 are staged as
    // zeros), so that the scalar offset of a stage is never negative
    constexpr int BACK = PAIR ? 0 : PAD * W + 4;  // (pad 0: staged row 0 is row r0 itself)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x - BACK), 0, (int)(((unsigned)p.B * p.Ci * HW + BACK) * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)((unsigned)p.B * p.Co * HWO * 4u), 0x00020000);

    // slot k of the DMA of a stage into buffer `buf` (k is a compile-time constant at every call site).  PAIR: `sb` = first sample of
    // the stage; HALF: `sb` = sample, `r0` = first output row
    float* const dump = smem + G::DUMP + wave * 64 * U;
    auto dma_slot = [&](int k, int sb, int r0, float* buf) {
        if constexpr (PAIR) {
            // [sample 0: X | D][sample 1: X | D]; a sample's 64-channel block is one contiguous run, copied as it lies
            const int u = k / (G::NIWX + G::NIWD), kk = k - u * (G::NIWX + G::NIWD);
            const bool isx = kk < G::NIWX;
            const int j = (isx ? kk : kk - G::NIWX) * 4 + wave;
            const int b = sb + u;
            const int q = j * 64 + lane;
            const bool have = j < (isx ? G::NIX : G::NID);
            float* d = have ? buf + (isx ? G::DIMG + u * G::XS : u * G::DS) + j * 64 * U : dump;
            const int len = (isx ? nci : nco) * G::XLEN;  // (XLEN == DLEN: whole planes)
            unsigned voff = (have && q * U < len) ? (unsigned)q * U * 4u : kOob;
            voff |= b < p.B ? 0u : kOob;  // (odd batch: the last stage's second sample)
            const unsigned soff = (unsigned)(b * (isx ? p.Ci : p.Co) + (isx ? ci0 : co0)) * (unsigned)(G::XLEN * 4);
            blds<U>(isx ? xrs : drs, voff, soff, d);
        } else {
            const bool isx = k < G::NIWX;
            const int i = isx ? k : k - G::NIWX;
            const int j = i * 4 + wave;
            float* d = j < (isx ? G::NIX : G::NID) ? buf + (isx ? G::DIMG : 0) + j * 64 * U : dump;
            if (isx) {
                // the halo row above the image (first row block of a sample) is staged as zeros -- and not fetched: for the first plane of
                // the tensor it lies in front of the allocation.  The halo row BELOW the image holds whatever follows the plane in memory:
                // its taps are selected away (seg_mfma)
                const unsigned desc = dx_desc[isx ? i : 0];
                const unsigned voff = (r0 == 0 && (desc & 1u)) ? kOob : desc & ~3u;
                blds<U>(xrs, voff, (unsigned)((sb * p.Ci + ci0) * HW + r0 * W) * 4u, d);
            } else {
                blds<U>(drs, dd_desc[isx ? 0 : i], (unsigned)((sb * p.Co + co0) * HWO + r0 * WO) * 4u, d);
            }
        }
    };

    // ---- per-lane operand bases (floats inside a buffer)
    // PAIR: kg = sample parity.  HALF: kg = half of the row; window element j of segment sg is column kg*HALFW + 7*sg + j - 1
    const int a_base = PAIR ? kg * G::DS + (wm * 32 + m) * G::QD : (wm * 32 + m) * G::QD + kg * G::HALFW + kw * G::NSEGW * 7;
    const int b_base = G::DIMG + (PAIR ? kg * G::XS + (wn * 32 + m) * G::QX
                                       : (wn * 32 + m) * G::QX + G::LEAD + kg * G::HALFW + kw * G::NSEGW * 7 - PAD);
    // HALF: the lanes whose first / last segment of a row touches the image's left / right edge
    const bool lmask = kg == 0 && kw == 0, rmask = kg == 1 && kw == G::KW - 1;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;

    // operands of one segment: A = 7 consecutive dy values, B = three windows of x (PAIR: elements 1..7 only -- columns -1 and W
    // do not exist and their taps are skipped)
    struct Ops {
        float a[7];
        float w[3][9];
    };
    auto read_ops = [&](Ops& o, const float* buf, int rr, int sg) {
        const float* ap = buf + a_base + rr * WO + sg * 7;
#pragma unroll
        for (int t = 0; t < 7; ++t) o.a[t] = ap[t];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int xr = PAIR ? rr + kx - 1 : rr + kx;  // staged row of this tap row
            if (PAIR && (xr < 0 || xr >= RU)) continue;
            const float* bp = buf + b_base + xr * W + sg * 7 - (PAIR ? 1 : 0);
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                if (PAIR && ((sg == 0 && j == 0) || (sg == G::NSEGW - 1 && j == 8))) continue;
                o.w[kx][j] = bp[j];
            }
        }
    };
    // (the DMA slots of the segment are issued between its MFMAs: slot group t in front of the MFMAs of pixel t)
    // bot (HALF, wave-uniform): this stage is the last row block of its sample -- the tap row below its last output row lies outside the
    // image and holds whatever follows the plane in memory: selected away (a select, not a product: it may be Inf / NaN).  (Skipping those
    // MFMAs instead -- one code variant per stage kind -- was measured SLOWER: 411-512 registers with spills for the 28 / 56-wide kernels.)
    auto seg_mfma = [&](const Ops& o, int rr, int sg, bool bot, auto&& slots) {
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            slots(t);
            float av = o.a[t];
            // pad 0: the columns behind the end of a dy row (the next row's first floats in LDS) belong to no output pixel
            if (PAD == 0 && sg == G::NSEGW - 1 && t >= 7 - (W - WO)) av = rmask ? 0.f : av;
            bsum += av;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                if (PAIR && (rr + kx - 1 < 0 || rr + kx - 1 >= RU)) continue;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int j = t + ky;
                    const bool left = sg == 0 && j == 0, right = sg == G::NSEGW - 1 && j == 8;
                    if (PAIR && (left || right)) continue;
                    float bv = o.w[kx][j];
                    if (!PAIR && PAD == 1 && left) bv = lmask ? 0.f : bv;
                    if (!PAIR && PAD == 1 && right) bv = rmask ? 0.f : bv;
                    if (!PAIR && PAD == 1 && kx == 2 && rr == RU - 1) bv = bot ? 0.f : bv;
                    acc[kx * 3 + ky] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[kx * 3 + ky], 0, 0, 0);
                }
            }
        }
    };

    const long long dbg_t0 = p.dbg == 9 ? clock64() : 0, dbg_w0 = p.dbg == 9 ? wall_clock64() : 0;
    // stage s -> (first sample, first output row)
    int sb = PAIR ? 2 * s_lo : s_lo / p.nrb, r0 = PAIR ? 0 : (s_lo - sb * p.nrb) * RU;
    if (s_lo < s_hi) {
#pragma unroll
        for (int k = 0; k < G::NIW; ++k) dma_slot(k, sb, r0, smem + (s_lo & 1) * G::BUF);
    }
    constexpr int PER_T = (G::PER_SEG + 6) / 7;  // DMA slots in front of one pixel's MFMAs
    for (int s = s_lo; s < s_hi; ++s) {
        // every wave waits for ITS OWN outstanding DMA (stage s), then the barrier publishes them and guarantees that every wave is
        // done reading the other buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const float* cur = smem + (s & 1) * G::BUF;
        float* nxt = smem + ((s + 1) & 1) * G::BUF;
        // the stage behind this one (behind the last one: that one again -- a harmless reload instead of a branch around every slot)
        int sbn = sb, r0n = r0;
        if (s + 1 < s_hi) {
            if (PAIR) sbn = sb + 2;
            else if (r0 + RU < HO) r0n = r0 + RU;
            else { r0n = 0; sbn = sb + 1; }
        }
        const bool bot = !PAIR && PAD == 1 && r0 + RU >= H;
        Ops ops[2];
        read_ops(ops[0], cur, 0, 0);
#pragma unroll
        for (int sgi = 0; sgi < G::NSEGS; ++sgi) {
            const int rr = sgi / G::NSEGW, sg = sgi % G::NSEGW;
            if (sgi + 1 < G::NSEGS) read_ops(ops[(sgi + 1) & 1], cur, (sgi + 1) / G::NSEGW, (sgi + 1) % G::NSEGW);
            seg_mfma(ops[sgi & 1], rr, sg, bot, [&](int t) {
#pragma unroll
                for (int k = sgi * G::PER_SEG + t * PER_T; k < sgi * G::PER_SEG + (t + 1) * PER_T && k < (sgi + 1) * G::PER_SEG && k < G::NIW; ++k)
                    dma_slot(k, sbn, r0n, nxt);
            });
        }
        sb = sbn;
        r0 = r0n;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the reload behind the last stage)
    const long long dbg_w1 = p.dbg == 9 ? wall_clock64() : 0;
    const long long dbg_c1 = p.dbg == 9 ? clock64() : 0;

    // ---- epilogue: the tile goes through LDS in two halves of 32 output channels, then to the slab in whole rows
    float* slab = p.slabs + (size_t)blockIdx.x * p.Co * p.pitch;
    float* outs = smem;                 // [32][OP]
    float* bias_s = smem + 32 * G::OP;  // [64]
    __syncthreads();
    {
        const float v = bsum + __shfl_xor(bsum, 32, 64);  // the two k-groups of channel co
        if (wn == 0 && kw == 0 && kg == 0) bias_s[wm * 32 + m] = v;
        if (G::KW == 2) {  // ... and the second wave of the tile, in a fixed order
            __syncthreads();
            if (kw == 1 && kg == 0) bias_s[wm * 32 + m] += v;
        }
    }
    for (int h = 0; h < 2; ++h) {
        if (wm == h && kw == 0) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
                    outs[row * G::OP + (wn * 32 + m) * 9 + t] = acc[t][r];
                }
        }
        __syncthreads();
        if (G::KW == 2) {
            if (wm == h && kw == 1) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
                        outs[row * G::OP + m * 9 + t] += acc[t][r];
                    }
            }
            __syncthreads();
        }
        const int ncol = nci * 9;
        for (int i = tid; i < 32 * CT * 9; i += 256) {
            const int row = i / (CT * 9), col = i - row * (CT * 9);
            if (h * 32 + row < nco && col < ncol) slab[(size_t)(co0 + h * 32 + row) * p.pitch + ci0 * 9 + col] = outs[row * G::OP + col];
        }
        if (blockIdx.y == 0 && tid < 32 && h * 32 + tid < nco) slab[(size_t)(co0 + h * 32 + tid) * p.pitch + p.Ntot] = bias_s[h * 32 + tid];
        __syncthreads();
    }
    if (p.dbg == 9 && threadIdx.x == 0 && (blockIdx.x | blockIdx.y | blockIdx.z) == 0)
        printf("wgrad_sp block 0: entry -> loop %lld, loop %lld (%lld shader cycles -> %.0f MHz), epilogue %lld ticks of 10 ns\n", dbg_w0 - dbg_e0,
               dbg_w1 - dbg_w0, dbg_c1 - dbg_t0, (double)(dbg_c1 - dbg_t0) / ((double)(dbg_w1 - dbg_w0) / 100.0), wall_clock64() - dbg_w1);
}

struct SpPlan {
    SpParams p;
    int mode;  // 0: not covered, 7 / 14 / 28 / 56: plane width
    int unit;  // DMA unit in floats
    int kblocks, gy, gz, ct;
};

constexpr int kRu14 = 7, kRu28 = 2, kRu56 = 1, kRu112 = 1;

bool make_sp_plan(const cnn_conv2d_desc* d, SpPlan* pl) {
    pl->mode = 0;
    const OptVal e = CNN_OPT_VAL("WGRAD_SP");
    if (e && atoi(e) == 0) return false;
    if (d->k != 3 || d->s != 1 || d->pad < 0 || d->pad > 1 || d->B < 1) return false;
    if (d->W != 7 && d->W != 14 && d->W != 28 && d->W != 56 && d->W != 112) return false;
    if (d->pad == 0 && d->W != 112) return false;  // (pad 0 is instantiated for the 112-wide planes only: the north-star shape)
    // (small channel counts: the 64 x 64 tile would be mostly padding)
    const int min_ch = e ? 1 : 32;
    if (d->Ci < min_ch || d->Co < min_ch) return false;
    int ru = 0;
    if (d->W == 7) { if (d->H != 7) return false; }
    else if (d->W == 14) ru = kRu14;
    else if (d->W == 28) ru = kRu28;
    else if (d->W == 56) ru = kRu56;
    else ru = kRu112;
    const int Ho = d->H - 2 * (1 - d->pad);
    if (Ho < 1 || (ru && Ho % ru != 0)) return false;
    // (buffer descriptors: byte sizes below 2^31)
    if ((long long)d->B * d->Ci * d->H * d->W >= (1ll << 29) || (long long)d->B * d->Co * d->H * d->W >= (1ll << 29)) return false;
    if ((long long)kTile * d->H * d->W >= (1 << 27)) return false;
    SpParams& p = pl->p;
    p.B = d->B; p.Ci = d->Ci; p.Co = d->Co; p.H = d->H;
    p.Ntot = d->Ci * 9; p.pitch = p.Ntot + 1;
    p.nrb = ru ? Ho / ru : 1;
    p.stages_total = ru ? d->B * p.nrb : (d->B + 1) / 2;
    pl->mode = d->W;
    // 16-byte DMA units everywhere (their sources need no alignment); SP_UNIT=1: the 4-byte instances (A/B, odd plane strides)
    pl->unit = 4;
    if (const OptVal u = CNN_OPT_VAL("SP_UNIT")) {
        if (atoi(u) == 1) pl->unit = 1;
    }
    pl->ct = d->W == 112 ? 32 : 64;
    pl->gy = (d->Ci + pl->ct - 1) / pl->ct;
    pl->gz = (d->Co + kTile - 1) / kTile;
    const int env = CNN_OPT_INT("SP_BLOCKS", 0);
    long long want = (env > 0 ? env : num_cus()) / ((long long)pl->gy * pl->gz);  // one workgroup per CU (LDS)
    if (want < 1) want = 1;
    if (want > p.stages_total) want = p.stages_total;
    p.stages_per_block = (int)((p.stages_total + want - 1) / want);
    pl->kblocks = (p.stages_total + p.stages_per_block - 1) / p.stages_per_block;
    p.dbg = CNN_MEASURE_INT("SP_DBG", 0);
    return true;
}

template <int W, int RU, bool PAIR, int U, int CT = 64, int PAD = 1>
int launch_sp(const SpPlan& pl, const cnn_conv2d_desc* d, hipStream_t s) {
    using G = SpGeom<W, RU, PAIR, U, CT, PAD>;
    auto kern = wgrad_sp_kernel<W, RU, PAIR, U, CT, PAD>;
    static DeviceOnce attr_once;
    if (attr_once.needed()) {
        CNN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds_bytes));
        attr_once.mark();
    }
    const dim3 grid(pl.kblocks, pl.gy, pl.gz);
    char name[48];
    snprintf(name, sizeof(name), CT == 64 ? "wgrad_sp<%d,%d,%d>" : "wgrad_sp<%d,%d,%d,ct32>", W, RU, U);
    CNN_KLAUNCH(s, name, (kern<<<grid, 256, G::lds_bytes, s>>>(pl.p)), "B%d Ci%d %dx%d Co%d k3 s1 p%d slabs%d", d->B, d->Ci, d->H, d->W, d->Co, PAD,
                pl.kblocks);
    return CNN_AMD_OK;
}

}  // namespace

namespace cnn_amd {

// conv_wgrad_sp2.hip: the stride-2 sibling is served through the same two entry points
int sp2_wgrad_slots(const cnn_conv2d_desc* d);
int sp2_wgrad_launch(const cnn_conv2d_desc* d, const float* x, const float* dy, float* slabs, hipStream_t s);

// conv_wgrad_sp_any.hip: the runtime-size member of the family takes every 3x3 / stride-1 geometry the instances here do not
int spa_wgrad_slots(const cnn_conv2d_desc* d);
int spa_wgrad_launch(const cnn_conv2d_desc* d, const float* x, const float* dy, float* slabs, hipStream_t s);

// number of partial slabs ([Co][Ci*9 + 1] floats each) the kernel writes, 0 when the geometry is not covered
int sp_wgrad_slots(const cnn_conv2d_desc* d) {
    if (d->s == 2) return sp2_wgrad_slots(d);
    SpPlan pl;
    return make_sp_plan(d, &pl) ? pl.kblocks : spa_wgrad_slots(d);
}

int sp_wgrad_launch(const cnn_conv2d_desc* d, const float* x, const float* dy, float* slabs, hipStream_t s) {
    if (d->s == 2) return sp2_wgrad_launch(d, x, dy, slabs, s);
    SpPlan pl;
    if (!make_sp_plan(d, &pl)) return spa_wgrad_launch(d, x, dy, slabs, s);
    pl.p.x = x; pl.p.dy = dy; pl.p.slabs = slabs;
    const int unit = pl.unit;
    switch (pl.mode) {
        case 7: return unit == 4 ? launch_sp<7, 7, true, 4>(pl, d, s) : launch_sp<7, 7, true, 1>(pl, d, s);
        case 14: return unit == 4 ? launch_sp<14, kRu14, false, 4>(pl, d, s) : launch_sp<14, kRu14, false, 1>(pl, d, s);
        case 28: return unit == 4 ? launch_sp<28, kRu28, false, 4>(pl, d, s) : launch_sp<28, kRu28, false, 1>(pl, d, s);
        case 112: return d->pad == 1 ? launch_sp<112, kRu112, false, 4, 32, 1>(pl, d, s) : launch_sp<112, kRu112, false, 4, 32, 0>(pl, d, s);
        default: return unit == 4 ? launch_sp<56, kRu56, false, 4>(pl, d, s) : launch_sp<56, kRu56, false, 1>(pl, d, s);
    }
}

}  // namespace cnn_amd
