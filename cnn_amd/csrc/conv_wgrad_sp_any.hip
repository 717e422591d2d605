// conv_wgrad_sp_any.hip -- Conv2D weight / bias gradient (cpu/src/conv2d.cpp:117-159) of 3x3 / stride-1 layers with ANY plane size
// (conv2d.cpp:41-42 accepts every H, W): the runtime-size member of the LDS-staged output-stationary family (round 6).
//     gw[co][ci][kx][ky] = sum_{b,r,c} dy[b][co][r][c] * x[b][ci][r + kx - p][c + ky - p]        (bias gradient: sum of dy)
// conv_wgrad_sp.hip is instantiated for the plane widths of the BASELINE workloads (7 / 14 / 28 / 56 / 112: left / right halves of a row or
// sample pairs on the two k-slots, rows of whole 7-pixel segments); every other 3x3 / stride-1 geometry ran on the register-direct kernel
// (67 - 108 TFLOP/s on the reference's own pad-0 VGG shapes at batch 128).  Same machine here -- a 64 (co) x 64 (ci) x 9 (taps) tile per
// workgroup, nine 32x32 accumulators of v_mfma_f32_32x32x2_f32 per wave, both operands staged through LDS by buffer-addressed DMA, two
// buffers, one barrier per stage, the DMA of stage s + 1 issued between the MFMAs of stage s, slabs out -- with NO plane size in any address
// computation of the loop:
//   * a STAGE is a fixed window of the output plane: one ROW PAIR (the two k-slots of an MFMA step are output rows r, r + 1: conv_wgrad_sp2.hip)
//     x CW columns (28 = four 7-pixel segments, or 21 = three), of one sample: dy [64 co][2 rows][CW], x [64 ci][4 rows][4 + CW + 2];
//   * every staged row is moved on its own, 16 bytes per lane from 4-byte-aligned sources, and everything outside the tensors IS ZERO IN
//     LDS: rows above / below the plane, the columns left of it (the staged x window starts four floats left of the block: one DMA unit,
//     outside the descriptor for the first block of a row), the units behind a row's end, and -- for widths that are no multiple of four --
//     the tail of the unit that straddles the end, which the lane that issued it overwrites once its DMA has landed (conv_rows_any.hip).
//     A delta of zero contributes nothing: no lane masks, no selects, no padding- or size-dependent code between the MFMAs.
// The planner picks CW by the fewest staged pixels per live pixel (21-wide outputs: one block of 21; 220-wide: eight of 28).
#include <cstdlib>
#include <type_traits>

#include "common.h"

using namespace cnn_amd;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;

struct SpaParams {
    const float* x;
    const float* dy;
    float* slabs;  // [gridDim.x][Co][pitch]
    int B, Ci, Co, H, W, HO, WO, pad;
    int Ntot, pitch;  // Ci*9, Ntot + 1 (column Ntot = bias gradient)
    int nrp, ncb;     // row pairs per plane, column blocks per row
    int stages_total, stages_per_block;
};

constexpr unsigned kOob = 0x80000000u;
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_ptr)lds, 16, (int)voff, (int)soff, 0, 0);
}

constexpr int kTile = 64;  // channels per workgroup tile, both ways (waves 2 x 2)
constexpr int kLeft = 4;   // floats the staged x window starts left of its column block (one DMA unit)

template <int CW>
struct SpaGeom {
    static_assert(CW % 7 == 0, "whole 7-pixel segments");
    static constexpr int NSEG = CW / 7;
    static constexpr int DP = (CW + 3) / 4 * 4;                 // staged dy row (floats)
    static constexpr int XW = (kLeft + CW + 2 + 3) / 4 * 4;     // staged x row: columns c0 - 4 ... c0 + CW + 1 (+ slack)
    static constexpr int DROWS = 2, XROWS = 4;
    static constexpr int DLEN = DROWS * DP, XLEN = XROWS * XW;
    // plane strides: a multiple of 4 floats with an odd number of 16-byte pieces
    static constexpr int stride_for(int len) { return (((len + 3) / 4) & 1) ? (len + 3) / 4 * 4 : (len + 3) / 4 * 4 + 4; }
    static constexpr int QD = stride_for(DLEN), QX = stride_for(XLEN);
    static constexpr int PPD = QD / 4, PPX = QX / 4;            // 16-byte pieces per plane
    static constexpr int UPD = DP / 4, UPX = XW / 4;            // ... per staged row
    static constexpr int NID = PPD, NIX = PPX;                  // DMA instructions per stage (64 planes: one plane per lane quarter ... see decode)
    static constexpr int NIWD = (NID + 3) / 4, NIWX = (NIX + 3) / 4;
    static constexpr int NIW = NIWD + NIWX;
    static constexpr int DS = NID * 256, XS = NIX * 256;
    static_assert(DS == kTile * QD && XS == kTile * QX, "the images hold their planes");
    static constexpr int BUF = DS + XS;                         // [D image][X image]
    static constexpr int DUMP = 2 * BUF;
    static constexpr int PER_SEG = (NIW + NSEG - 1) / NSEG;
    static constexpr int OP = kTile * 9 + 1;
    static constexpr size_t epi_bytes = (size_t)(32 * OP + 64) * sizeof(float);
    static constexpr size_t buf_bytes = (size_t)(2 * BUF + 4 * 256) * sizeof(float);
    static constexpr size_t lds_bytes = buf_bytes > epi_bytes ? buf_bytes : epi_bytes;
    static_assert(lds_bytes <= 160 * 1024, "LDS plan");
};

template <int CW>
__global__ __launch_bounds__(256) void wgrad_spa_kernel(const SpaParams p) {
    using G = SpaGeom<CW>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, m = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ci0 = blockIdx.y * kTile, co0 = blockIdx.z * kTile;
    const int nci = p.Ci - ci0 < kTile ? p.Ci - ci0 : kTile, nco = p.Co - co0 < kTile ? p.Co - co0 : kTile;
    const int H = p.H, W = p.W, HW = H * W, HO = p.HO, WO = p.WO, HWO = HO * WO, PAD = p.pad;

    for (int i = tid * 4; i < 2 * G::BUF + 4 * 256; i += 1024) *(float4*)(smem + i) = make_float4(0.f, 0.f, 0.f, 0.f);

    const int s_lo = blockIdx.x * p.stages_per_block;
    const int s_hi = s_lo + p.stages_per_block < p.stages_total ? s_lo + p.stages_per_block : p.stages_total;

    // ---- this wave's share of a stage's DMA, decoded once: 16-byte unit q of an image = (plane, staged row, unit of the row); the lane's
    //      byte offset from the stage's first element, its row and its first column inside the window (-1: nothing to move)
    unsigned dd_off[G::NIWD], dx_off[G::NIWX];
    int dd_rc[G::NIWD], dx_rc[G::NIWX];  // row | first column of the unit << 8, or -1
#pragma unroll
    for (int i = 0; i < G::NIWD; ++i) {
        const int j = i * 4 + wave, q = j * 64 + lane;
        const int plane = q / G::PPD, e = q - plane * G::PPD, rr = e / G::UPD, u = e - rr * G::UPD;
        const bool have = j < G::NID && rr < G::DROWS && plane < nco;
        dd_off[i] = (unsigned)(plane * HWO + rr * WO + 4 * u) * 4u;
        dd_rc[i] = have ? (rr | ((4 * u) << 8)) : -1;
    }
#pragma unroll
    for (int i = 0; i < G::NIWX; ++i) {
        const int j = i * 4 + wave, q = j * 64 + lane;
        const int plane = q / G::PPX, e = q - plane * G::PPX, rr = e / G::UPX, u = e - rr * G::UPX;
        const bool have = j < G::NIX && rr < G::XROWS && plane < nci;
        dx_off[i] = (unsigned)(plane * HW + rr * W + 4 * u) * 4u;
        dx_rc[i] = have ? (rr | ((4 * u) << 8)) : -1;
    }
    __syncthreads();
    // x is addressed from PAD * W + 4 floats in front of the tensor (never fetched: those lanes are outside the image), so that the scalar
    // offset of a stage -- (first row r0 - PAD, first column c0 - 4) -- is never negative
    const int BACK = PAD * W + kLeft;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x - BACK), 0, (int)(((unsigned)p.B * p.Ci * HW + BACK) * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)((unsigned)p.B * p.Co * HWO * 4u), 0x00020000);

    // slot k of the DMA of stage (sample sb, first output row r0, first output column c0) into `buf`
    float* const dump = smem + G::DUMP + wave * 256;
    auto dma_slot = [&](int k, int sb, int r0, int c0, float* buf) {
        if (k < G::NIWD) {
            const int j = k * 4 + wave;
            float* d = j < G::NID ? buf + j * 256 : dump;
            const int rc = dd_rc[k];
            // a row below the plane, a unit behind the row's end, a channel behind the tensor: zeros (a lane offset outside the descriptor)
            const bool ok = rc >= 0 && r0 + (rc & 255) < HO && c0 + (rc >> 8) < WO;
            blds16(drs, ok ? dd_off[k] : kOob, (unsigned)((sb * p.Co + co0) * HWO + r0 * WO + c0) * 4u, d);
        } else {
            const int i = k - G::NIWD, j = i * 4 + wave;
            float* d = j < G::NIX ? buf + G::DS + j * 256 : dump;
            const int rc = dx_rc[i];
            const bool ok = rc >= 0 && (unsigned)(r0 - PAD + (rc & 255)) < (unsigned)H && (unsigned)(c0 - kLeft + (rc >> 8)) < (unsigned)W;
            blds16(xrs, ok ? dx_off[i] : kOob, (unsigned)((sb * p.Ci + ci0) * HW + r0 * W + c0) * 4u, d);
        }
    };
    // the tail of the unit that straddles a row's end (widths that are no multiple of 4; last column block only), zeroed by the lane that
    // issued the unit once it has landed: what it brought there is the next row / plane.  In the last column block the straddling unit is
    // the same one in every stage (column WO - c0_last resp. W - c0_last + 4 of the window): which DMA slots hold one in ANY lane of this
    // wave is decided here, once (wave-uniform bit sets: most slots hold none)
    const int c0_last = (p.ncb - 1) * CW;
    // (units are cut from the BLOCK's first column, and a 21-column block starts anywhere modulo 4: what counts is the row's end relative to
    //  c0_last, not to column 0 -- round 6 fix: found by tests/sweeps/fuzz_conv.py on 30- / 39- / 40- / 59- / 60-wide outputs)
    const int nz_d = (4 - ((WO - c0_last) & 3)) & 3, nz_x = (4 - ((W - c0_last) & 3)) & 3;  // floats to zero (0: the rows end on a unit)
    unsigned fixd_slots = 0, fixx_slots = 0;
    bool fixd_mine[G::NIWD], fixx_mine[G::NIWX];
#pragma unroll
    for (int k = 0; k < G::NIWD; ++k) {
        const int rc = dd_rc[k], col = c0_last + (rc >> 8);
        fixd_mine[k] = nz_d != 0 && rc >= 0 && col < WO && col + 4 > WO;
        if (__builtin_amdgcn_ballot_w64(fixd_mine[k]) != 0ull) fixd_slots |= 1u << k;
    }
#pragma unroll
    for (int i = 0; i < G::NIWX; ++i) {
        const int rc = dx_rc[i], col = c0_last - kLeft + (rc >> 8);
        fixx_mine[i] = nz_x != 0 && rc >= 0 && col >= 0 && col < W && col + 4 > W;
        if (__builtin_amdgcn_ballot_w64(fixx_mine[i]) != 0ull) fixx_slots |= 1u << i;
    }
    fixd_slots = (unsigned)__builtin_amdgcn_readfirstlane((int)fixd_slots);
    fixx_slots = (unsigned)__builtin_amdgcn_readfirstlane((int)fixx_slots);
    auto fixups = [&](int c0, float* buf) {
        if (c0 != c0_last || (fixd_slots | fixx_slots) == 0) return;  // (wave-uniform)
#pragma unroll
        for (int k = 0; k < G::NIWD; ++k) {
            if (!((fixd_slots >> k) & 1u)) continue;
            if (fixd_mine[k]) {
                float* z = buf + (k * 4 + wave) * 256 + lane * 4 + 4 - nz_d;
                z[0] = 0.f;
                if (nz_d >= 2) z[1] = 0.f;
                if (nz_d >= 3) z[2] = 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < G::NIWX; ++i) {
            if (!((fixx_slots >> i) & 1u)) continue;
            if (fixx_mine[i]) {
                float* z = buf + G::DS + (i * 4 + wave) * 256 + lane * 4 + 4 - nz_x;
                z[0] = 0.f;
                if (nz_x >= 2) z[1] = 0.f;
                if (nz_x >= 3) z[2] = 0.f;
            }
        }
    };

    // ---- per-lane operand bases (floats inside a buffer): kg = parity of the output row inside its pair
    const int a_base = (wm * 32 + m) * G::QD + kg * G::DP;
    const int b_base = G::DS + (wn * 32 + m) * G::QX + kg * G::XW + kLeft - PAD;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;

    struct Ops {
        float a[7];
        float w[3][9];
    };
    auto read_ops = [&](Ops& o, const float* buf, int sg) {
        const float* ap = buf + a_base + sg * 7;
#pragma unroll
        for (int t = 0; t < 7; ++t) o.a[t] = ap[t];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float* bp = buf + b_base + kx * G::XW + sg * 7;
#pragma unroll
            for (int j = 0; j < 9; ++j) o.w[kx][j] = bp[j];
        }
    };
    auto seg_mfma = [&](const Ops& o, auto&& slots) {
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            slots(t);
            const float av = o.a[t];
            bsum += av;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) acc[kx * 3 + ky] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, o.w[kx][t + ky], acc[kx * 3 + ky], 0, 0, 0);
        }
    };

    // stage s -> (sample, row pair, column block): the column blocks of a row pair follow each other (their x windows share lines)
    auto decode = [&](int s, int& sb, int& r0, int& c0) {
        const int per = p.nrp * p.ncb;
        sb = s / per;
        const int rem = s - sb * per, rp = rem / p.ncb;
        r0 = 2 * rp;
        c0 = (rem - rp * p.ncb) * CW;
    };
    int sb, r0, c0;
    decode(s_lo < s_hi ? s_lo : 0, sb, r0, c0);
    if (s_lo < s_hi) {
#pragma unroll
        for (int k = 0; k < G::NIW; ++k) dma_slot(k, sb, r0, c0, smem + (s_lo & 1) * G::BUF);
    }
    constexpr int PER_T = (G::PER_SEG + 6) / 7;  // DMA slots in front of one pixel's MFMAs
    for (int s = s_lo; s < s_hi; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float* cur = smem + (s & 1) * G::BUF;
        fixups(c0, cur);
        __syncthreads();
        float* nxt = smem + ((s + 1) & 1) * G::BUF;
        // the stage behind this one (behind the last one: that one again -- a harmless reload instead of a branch around every slot)
        int sbn = sb, r0n = r0, c0n = c0;
        if (s + 1 < s_hi) {
            c0n = c0 + CW;
            if (c0n >= WO) {
                c0n = 0;
                r0n = r0 + 2;
                if (r0n >= HO) { r0n = 0; sbn = sb + 1; }
            }
        }
        Ops ops[2];
        read_ops(ops[0], cur, 0);
#pragma unroll
        for (int sg = 0; sg < G::NSEG; ++sg) {
            if (sg + 1 < G::NSEG) read_ops(ops[(sg + 1) & 1], cur, sg + 1);
            seg_mfma(ops[sg & 1], [&](int t) {
#pragma unroll
                for (int k = sg * G::PER_SEG + t * PER_T; k < sg * G::PER_SEG + (t + 1) * PER_T && k < (sg + 1) * G::PER_SEG && k < G::NIW; ++k)
                    dma_slot(k, sbn, r0n, c0n, nxt);
            });
        }
        sb = sbn; r0 = r0n; c0 = c0n;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the reload behind the last stage)

    // ---- epilogue: the tile goes through LDS in two halves of 32 output channels, then to the slab in whole rows (conv_wgrad_sp.hip)
    float* slab = p.slabs + (size_t)blockIdx.x * p.Co * p.pitch;
    float* outs = smem;                 // [32][OP]
    float* bias_s = smem + 32 * G::OP;  // [64]
    __syncthreads();
    {
        const float v = bsum + __shfl_xor(bsum, 32, 64);  // the two k-groups of channel co
        if (wn == 0 && kg == 0) bias_s[wm * 32 + m] = v;
    }
    for (int h = 0; h < 2; ++h) {
        if (wm == h) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
                    outs[row * G::OP + (wn * 32 + m) * 9 + t] = acc[t][r];
                }
        }
        __syncthreads();
        const int ncol = nci * 9;
        for (int i = tid; i < 32 * kTile * 9; i += 256) {
            const int row = i / (kTile * 9), col = i - row * (kTile * 9);
            if (h * 32 + row < nco && col < ncol) slab[(size_t)(co0 + h * 32 + row) * p.pitch + ci0 * 9 + col] = outs[row * G::OP + col];
        }
        if (blockIdx.y == 0 && tid < 32 && h * 32 + tid < nco) slab[(size_t)(co0 + h * 32 + tid) * p.pitch + p.Ntot] = bias_s[h * 32 + tid];
        __syncthreads();
    }
}

struct SpaPlan {
    SpaParams p;
    int cw, kblocks, gy, gz;
};

bool make_spa_plan(const cnn_conv2d_desc* d, SpaPlan* pl) {
    const OptVal e = CNN_OPT_VAL("WGRAD_SP_ANY"), all = CNN_OPT_VAL("WGRAD_SP");  // (WGRAD_SP=0: the whole family off)
    if ((e && atoi(e) == 0) || (all && atoi(all) == 0)) return false;
    if (d->k != 3 || d->s != 1 || d->pad < 0 || d->pad > 1 || d->B < 1) return false;
    const int Ho = d->H + 2 * d->pad - 2, Wo = d->W + 2 * d->pad - 2;
    if (Ho < 1 || Wo < 1) return false;
    const bool forced = e && atoi(e) == 2;
    const int min_ch = forced ? 1 : 32;  // (small channel counts: the 64 x 64 tile would be mostly padding)
    if (d->Ci < min_ch || d->Co < min_ch) return false;
    // (output rows of whole 16-pixel runs are the register-direct kernel's best case: measured at batch 128 / 64, 48-wide outputs: 107 / 87
    //  TFLOP/s there against 96 / 87 here -- everywhere else this kernel wins by 9 - 33 %)
    if (!forced && Wo % 16 == 0) return false;
    if ((long long)d->B * d->Ci * d->H * d->W >= (1ll << 29) || (long long)d->B * d->Co * Ho * Wo >= (1ll << 29)) return false;
    if ((long long)kTile * d->H * d->W >= (1 << 27)) return false;
    SpaParams& p = pl->p;
    p.B = d->B; p.Ci = d->Ci; p.Co = d->Co; p.H = d->H; p.W = d->W; p.HO = Ho; p.WO = Wo; p.pad = d->pad;
    p.Ntot = d->Ci * 9; p.pitch = p.Ntot + 1;
    // column block: the one that stages the fewest pixels per live pixel (ties: the wider block, fewer stages)
    const int n28 = (Wo + 27) / 28 * 28, n21 = (Wo + 20) / 21 * 21;
    pl->cw = n21 < n28 ? 21 : 28;
    p.nrp = (Ho + 1) / 2;
    p.ncb = (Wo + pl->cw - 1) / pl->cw;
    const long long stages = (long long)d->B * p.nrp * p.ncb;
    if (stages >= (1ll << 30)) return false;
    p.stages_total = (int)stages;
    pl->gy = (d->Ci + kTile - 1) / kTile;
    pl->gz = (d->Co + kTile - 1) / kTile;
    const int env = CNN_OPT_INT("SP_BLOCKS", 0);
    long long want = (env > 0 ? env : num_cus()) / ((long long)pl->gy * pl->gz);  // one workgroup per CU (LDS)
    if (want < 1) want = 1;
    if (want > p.stages_total) want = p.stages_total;
    p.stages_per_block = (int)((p.stages_total + want - 1) / want);
    pl->kblocks = (p.stages_total + p.stages_per_block - 1) / p.stages_per_block;
    return true;
}

template <int CW>
int launch_spa(const SpaPlan& pl, const cnn_conv2d_desc* d, hipStream_t s) {
    using G = SpaGeom<CW>;
    auto kern = wgrad_spa_kernel<CW>;
    static DeviceOnce attr_once;
    if (attr_once.needed()) {
        CNN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds_bytes));
        attr_once.mark();
    }
    const dim3 grid(pl.kblocks, pl.gy, pl.gz);
    char name[48];
    snprintf(name, sizeof(name), "wgrad_sp_any<2x%d>", CW);
    CNN_KLAUNCH(s, name, (kern<<<grid, 256, G::lds_bytes, s>>>(pl.p)), "B%d Ci%d %dx%d Co%d k3 s1 p%d slabs%d", d->B, d->Ci, d->H, d->W, d->Co, d->pad,
                pl.kblocks);
    return CNN_AMD_OK;
}

}  // namespace

namespace cnn_amd {

// number of partial slabs ([Co][Ci*9 + 1] floats each) the kernel writes, 0 when the geometry is not covered
int spa_wgrad_slots(const cnn_conv2d_desc* d) {
    SpaPlan pl;
    return make_spa_plan(d, &pl) ? pl.kblocks : 0;
}

int spa_wgrad_launch(const cnn_conv2d_desc* d, const float* x, const float* dy, float* slabs, hipStream_t s) {
    SpaPlan pl;
    if (!make_spa_plan(d, &pl)) return fail(CNN_AMD_E_BADARG, "wgrad_sp_any: geometry not covered");
    pl.p.x = x; pl.p.dy = dy; pl.p.slabs = slabs;
    return pl.cw == 28 ? launch_spa<28>(pl, d, s) : launch_spa<21>(pl, d, s);
}

}  // namespace cnn_amd
