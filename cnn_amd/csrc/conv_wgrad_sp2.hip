// conv_wgrad_sp2.hip -- Conv2D weight / bias gradient (cpu/src/conv2d.cpp:117-159) of 3x3 / STRIDE-2 layers -- the reference's default
// stride (architectures.h:69; the window walk x += stride at conv2d.cpp:76-77) -- as the stride-2 sibling of conv_wgrad_sp.hip (round 6):
//     gw[co][ci][kx][ky] = sum_{b,r,c} dy[b][co][r][c] * x[b][ci][2r + kx - p][2c + ky - p]        (bias gradient: sum of dy)
// Same machine as the stride-1 kernel: an output-stationary 64 (co) x CT (ci) x 9 (taps) tile per workgroup, nine 32x32 accumulators of
// v_mfma_f32_32x32x2_f32 per wave, both operands staged through LDS by buffer-addressed DMA with out-of-range zero fill, every LDS address
// one per-lane base + a compile-time immediate, two buffers, one barrier per stage, the DMA of stage s + 1 issued in slices between the
// MFMAs of stage s, slabs[workgroup][Co][Ci*9 + 1] out (reduce_slabs adds them in a fixed order).
//
// What stride 2 changes:
//   * the two k-slots of an MFMA step (kg = lane / 32) are two OUTPUT ROWS (r, r + 1) -- a constant WO floats apart in the staged dy, 2 W
//     floats apart in the staged x -- instead of the two halves of a row: rows of 28 / 14 / 7 (pad 1) or 27 / 13 / 6 (pad 0) pixels need no
//     divisibility, a 7-wide row is one segment.  A stage = RP row PAIRS of one sample; the slot behind the last row of a plane with an odd
//     number of rows is selected away on both operands (what is staged there is the next plane);
//   * the B operand of lane (ci n, kg) for pixel (r, c) and tap (kx, ky) is x_lds[n][(2 r + kx) * W + 2 c + ky - p]: a lane reads ITS channel
//     plane, so the stride-2 walk is only a different immediate and costs no LDS bank conflicts; a 7-pixel segment reads three windows of
//     15 floats (63 MFMAs per 52 LDS reads);
//   * pad 1 (even W): the tap row above the image exists only for the first row block and is staged as zeros (whole 16-byte units: LEAD pad
//     floats in front of every plane put image row 0 on a unit boundary); the tap column left of the image is the same column for all 64
//     lanes -- those MFMAs are not issued (CT = 64) or their B value is selected away on the wave that owns the row's first segment
//     (CT = 32).  Nothing is staged below or right of the image: with an even width 2 (WO - 1) + 1 = W - 1.  Pad 0: no halo at all.
#include <cstdlib>
#include <type_traits>

#include "common.h"

using namespace cnn_amd;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;

struct Sp2Params {
    const float* x;
    const float* dy;
    float* slabs;  // [gridDim.x][Co][pitch]
    int B, Ci, Co;
    int Ntot, pitch;  // Ci*9, Ntot + 1 (column Ntot = bias gradient)
    int nrb;          // row blocks per sample
    int stages_total, stages_per_block;
};

constexpr unsigned kOob = 0x80000000u;
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float* lds) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_ptr)lds, 16, (int)voff, (int)soff, 0, 0);
}

constexpr int kTile = 64;  // output channels per workgroup tile

// W x W planes, padding PAD, RP row pairs per stage, CT input channels per workgroup tile (64: waves 2 (co) x 2 (ci); 32: waves 2 (co) x 2
// (the two waves of a tile split the segments of a row and are added in the epilogue))
template <int W, int PAD, int RP, int CT>
struct Sp2Geom {
    static_assert(PAD == 0 || (PAD == 1 && W % 2 == 0), "pad 1: even planes (no tap leaves the image on the right / below)");
    static_assert(CT == 64 || CT == 32, "tile variants");
    static constexpr int H = W;
    static constexpr int WO = (W + 2 * PAD - 3) / 2 + 1, HO = WO;
    static constexpr int KW = kTile / CT;
    static constexpr int NSEG = (WO + 6) / 7;          // 7-pixel segments per row (the last one may be shorter)
    static_assert(NSEG % KW == 0 && (KW == 1 || WO % 7 == 0), "segments per wave");
    static constexpr int NSEGW = NSEG / KW;
    static constexpr int seglen(int sg) { return WO - 7 * sg < 7 ? WO - 7 * sg : 7; }
    static constexpr int ROWS = 2 * RP;                // output-row slots per stage
    static constexpr int NRB = (HO + ROWS - 1) / ROWS;
    static constexpr bool RAGGED = HO % ROWS != 0;     // the last stage of a sample has empty row slots
    static constexpr int XROWS = 2 * ROWS + 1;         // staged input rows: 2 r0 - PAD ... 2 r0 - PAD + 2 ROWS
    static constexpr int XLEN = XROWS * W, DLEN = ROWS * WO;
    // plane strides: a multiple of 4 floats with an odd number of 16-byte pieces
    static constexpr int stride_for(int len) { return (((len + 3) / 4) & 1) ? (len + 3) / 4 * 4 : (len + 3) / 4 * 4 + 4; }
    static constexpr int LEAD = PAD == 1 ? (4 - W % 4) % 4 : 0;  // image row 0 of the first row block on a 16-byte unit
    static constexpr int XSPAN = LEAD + XLEN;
    static constexpr int QX = stride_for(XSPAN), QD = stride_for(DLEN);
    static constexpr int PPX = QX / 4, PPD = QD / 4;   // 16-byte pieces per plane
    static constexpr int NIX = (CT * PPX + 63) / 64, NID = PPD;  // DMA instructions (64 lanes x 16 bytes) per stage
    static constexpr int NIWX = (NIX + 3) / 4, NIWD = (NID + 3) / 4;
    static constexpr int NIW = NIWX + NIWD;
    static constexpr int XS = NIX * 256, DS = NID * 256;
    static_assert(XS >= CT * QX && DS == kTile * QD, "the images hold their planes");
    static constexpr int BUF = DS + XS;                // [D image][X image]
    static constexpr int DUMP = 2 * BUF;
    static constexpr int NSEGS = RP * NSEGW;           // segments per stage and wave
    static constexpr int PER_SEG = (NIW + NSEGS - 1) / NSEGS;
    static constexpr int OP = CT * 9 + 1;
    static constexpr size_t epi_bytes = (size_t)(32 * OP + 64) * sizeof(float);
    static constexpr size_t buf_bytes = (size_t)(2 * BUF + 4 * 256) * sizeof(float);
    static constexpr size_t lds_bytes = buf_bytes > epi_bytes ? buf_bytes : epi_bytes;
    static_assert(lds_bytes <= 160 * 1024, "LDS plan");
    // the last tap column of a row's last pixel lies inside the image (so the highest B read, row 2 ROWS, stays inside the staged rows)
    static_assert(2 * WO - PAD <= W - 1, "operand reads inside the staged rows");
};

template <int W, int PAD, int RP, int CT>
__global__ __launch_bounds__(256) void wgrad_sp2_kernel(const Sp2Params p) {
    using G = Sp2Geom<W, PAD, RP, CT>;
    constexpr int WO = G::WO, HO = G::HO, HW = W * W, HWO = HO * WO;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, m = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = CT == 64 ? (wave & 1) : 0, kw = CT == 64 ? 0 : (wave & 1);
    const int ci0 = blockIdx.y * CT, co0 = blockIdx.z * kTile;
    const int nci = p.Ci - ci0 < CT ? p.Ci - ci0 : CT, nco = p.Co - co0 < kTile ? p.Co - co0 : kTile;

    // everything the DMA never writes reads as zero
    for (int i = tid * 4; i < 2 * G::BUF + 4 * 256; i += 1024) *(float4*)(smem + i) = make_float4(0.f, 0.f, 0.f, 0.f);

    const int s_lo = blockIdx.x * p.stages_per_block;
    const int s_hi = s_lo + p.stages_per_block < p.stages_total ? s_lo + p.stages_per_block : p.stages_total;

    // ---- this wave's share of a stage's DMA, decoded once: the lane's byte offset from (channel ci0 / co0, first staged row) of the sample
    //      -- for x counted from PAD*W + 4 floats in front of it, see xrs --, bit 0 set: the unit holds nothing but the halo row above the
    //      image (+ lead pad).  kOob: nothing to move (pad unit / channel behind the tensor / no instruction for this wave)
    unsigned dx_desc[G::NIWX], dd_desc[G::NIWD];
#pragma unroll
    for (int i = 0; i < G::NIWX; ++i) {
        const int j = i * 4 + wave, q = j * 64 + lane;
        const int plane = q / G::PPX, e = q - plane * G::PPX;
        dx_desc[i] = (j < G::NIX && e * 4 < G::XSPAN && plane < nci)
                         ? ((unsigned)(plane * HW + e * 4 + 4 - G::LEAD) * 4u) | ((PAD == 1 && (e + 1) * 4 <= G::LEAD + W) ? 1u : 0u) : kOob;
    }
#pragma unroll
    for (int i = 0; i < G::NIWD; ++i) {
        const int j = i * 4 + wave, q = j * 64 + lane;
        const int plane = q / G::PPD, e = q - plane * G::PPD;
        dd_desc[i] = (j < G::NID && e * 4 < G::DLEN && plane < nco) ? (unsigned)(plane * HWO + e * 4) * 4u : kOob;
    }
    __syncthreads();
    constexpr int BACK = PAD * W + 4;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x - BACK), 0, (int)(((unsigned)p.B * p.Ci * HW + BACK) * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)((unsigned)p.B * p.Co * HWO * 4u), 0x00020000);

    // slot k of the DMA of stage (sample sb, first output row r0) into `buf`
    float* const dump = smem + G::DUMP + wave * 256;
    auto dma_slot = [&](int k, int sb, int r0, float* buf) {
        if (k < G::NIWX) {
            const int j = k * 4 + wave;
            float* d = j < G::NIX ? buf + G::DS + j * 256 : dump;
            const unsigned desc = dx_desc[k];
            const unsigned voff = (r0 == 0 && (desc & 1u)) ? kOob : desc & ~3u;
            blds16(xrs, voff, (unsigned)((sb * p.Ci + ci0) * HW + 2 * r0 * W) * 4u, d);
        } else {
            const int i = k - G::NIWX, j = i * 4 + wave;
            float* d = j < G::NID ? buf + j * 256 : dump;
            blds16(drs, dd_desc[i], (unsigned)((sb * p.Co + co0) * HWO + r0 * WO) * 4u, d);
        }
    };

    // ---- per-lane operand bases (floats inside a buffer): kg = parity of the output row inside its pair
    const int a_base = (wm * 32 + m) * G::QD + kg * WO + kw * G::NSEGW * 7;
    const int b_base = G::DS + (wn * 32 + m) * G::QX + G::LEAD + kg * 2 * W + kw * G::NSEGW * 14 - PAD;
    const bool lmask = kw == 0;  // (CT = 32) this wave owns the first segment of a row: its first tap column lies left of the image

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;

    // operands of one segment of row pair i: A = up to 7 consecutive dy values, B = three windows of 2 * len + 1 x values
    struct Ops {
        float a[7];
        float w[3][15];
    };
    auto read_ops = [&](Ops& o, const float* buf, int i, int sg) {
        // (sg: the wave's segment; the row's segment kw * NSEGW + sg has the same length -- whole 7-pixel segments when two waves share a row)
        const int len = G::seglen(sg);
        const float* ap = buf + a_base + 2 * i * WO + sg * 7;
#pragma unroll
        for (int t = 0; t < 7; ++t)
            if (t < len) o.a[t] = ap[t];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float* bp = buf + b_base + (4 * i + kx) * W + sg * 14;
#pragma unroll
            for (int j = 0; j < 15; ++j)
                if (j < 2 * len + 1 && !(CT == 64 && PAD == 1 && sg == 0 && j == 0)) o.w[kx][j] = bp[j];
        }
    };
    // ok (per lane): this lane's row slot of pair i lies inside the plane (RAGGED instances, last row block of a sample)
    auto seg_mfma = [&](const Ops& o, int sg, bool ok, auto&& slots) {
        const int len = G::seglen(sg);
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            slots(t);
            if (t >= len) continue;
            float av = o.a[t];
            if (G::RAGGED) av = ok ? av : 0.f;
            bsum += av;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int j = 2 * t + ky;
                    const bool left = PAD == 1 && sg == 0 && j == 0;  // column -1
                    if (CT == 64 && left) continue;                   // (the same column for all lanes: nothing to add)
                    float bv = o.w[kx][j];
                    if (CT == 32 && left) bv = lmask ? 0.f : bv;
                    if (G::RAGGED) bv = ok ? bv : 0.f;                // (a select, not a product: what is staged there is another plane)
                    acc[kx * 3 + ky] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[kx * 3 + ky], 0, 0, 0);
                }
            }
        }
    };

    // stage s -> (sample, first output row)
    int sb = s_lo / p.nrb, r0 = (s_lo - sb * p.nrb) * G::ROWS;
    if (s_lo < s_hi) {
#pragma unroll
        for (int k = 0; k < G::NIW; ++k) dma_slot(k, sb, r0, smem + (s_lo & 1) * G::BUF);
    }
    constexpr int PER_T = (G::PER_SEG + 6) / 7;  // DMA slots in front of one pixel's MFMAs
    for (int s = s_lo; s < s_hi; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const float* cur = smem + (s & 1) * G::BUF;
        float* nxt = smem + ((s + 1) & 1) * G::BUF;
        // the stage behind this one (behind the last one: that one again -- a harmless reload instead of a branch around every slot)
        int sbn = sb, r0n = r0;
        if (s + 1 < s_hi) {
            if (r0 + G::ROWS < HO) r0n = r0 + G::ROWS;
            else { r0n = 0; sbn = sb + 1; }
        }
        const int nvr = HO - r0;  // row slots of this stage inside the plane
        Ops ops[2];
        read_ops(ops[0], cur, 0, 0);
#pragma unroll
        for (int sgi = 0; sgi < G::NSEGS; ++sgi) {
            const int i = sgi / G::NSEGW, sg = sgi % G::NSEGW;
            if (sgi + 1 < G::NSEGS) read_ops(ops[(sgi + 1) & 1], cur, (sgi + 1) / G::NSEGW, (sgi + 1) % G::NSEGW);
            seg_mfma(ops[sgi & 1], sg, 2 * i + kg < nvr, [&](int t) {
#pragma unroll
                for (int k = sgi * G::PER_SEG + t * PER_T; k < sgi * G::PER_SEG + (t + 1) * PER_T && k < (sgi + 1) * G::PER_SEG && k < G::NIW; ++k)
                    dma_slot(k, sbn, r0n, nxt);
            });
        }
        sb = sbn;
        r0 = r0n;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the reload behind the last stage)

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
}

// ---- the instances: (plane width, padding) of the BASELINE workloads' 3x3 / stride-2 layers -----------------------------------------
struct Sp2Inst {
    int w, pad, rp, ct;
    int dflt;  // taken by the default dispatch (1) or only with CNN_AMD_WGRAD_SP2=2 (0: not faster in the step than the kernel it would replace)
};
constexpr Sp2Inst kInst[] = {
    // stage entries of the ResNet-shaped stack, batch 64: 64 -> 128 @ 56, (128 -> 256 @ 28), 256 -> 512 @ 14, pad 1
    {56, 1, 1, 32, 1}, {28, 1, 1, 64, 1}, {14, 1, 4, 64, 1},
    // the reference net behind its first block (alexnet.cpp:20-29), batch 256: 32 -> 64 @ 27, 64 -> 128 @ 13, pad 0
    {27, 0, 1, 64, 0}, {13, 0, 3, 64, 0},
};
constexpr int kNumInst = (int)(sizeof(kInst) / sizeof(kInst[0]));

struct Sp2Plan {
    Sp2Params p;
    int inst, kblocks, gy, gz;
};

template <int I>
constexpr int inst_rows() { return 2 * kInst[I].rp; }

bool make_sp2_plan(const cnn_conv2d_desc* d, Sp2Plan* pl) {
    const OptVal e = CNN_OPT_VAL("WGRAD_SP2");  // 0: never; 2: every instance, any channel count (tests, measurements)
    if (e && atoi(e) == 0) return false;
    const bool all = e && atoi(e) == 2;
    if (d->k != 3 || d->s != 2 || d->B < 1 || d->H != d->W) return false;
    int inst = -1;
    for (int i = 0; i < kNumInst; ++i)
        if (kInst[i].w == d->W && kInst[i].pad == d->pad && (all || kInst[i].dflt)) inst = i;
    if (inst < 0) return false;
    const int min_ch = all ? 1 : 32;  // (small channel counts: the 64 x 64 tile would be mostly padding)
    if (d->Ci < min_ch || d->Co < min_ch) return false;
    const int Ho = (d->H + 2 * d->pad - 3) / 2 + 1;
    if ((long long)d->B * d->Ci * d->H * d->W >= (1ll << 29) || (long long)d->B * d->Co * Ho * Ho >= (1ll << 29)) return false;
    Sp2Params& p = pl->p;
    p.B = d->B; p.Ci = d->Ci; p.Co = d->Co;
    p.Ntot = d->Ci * 9; p.pitch = p.Ntot + 1;
    const int rows = 2 * kInst[inst].rp;
    p.nrb = (Ho + rows - 1) / rows;
    p.stages_total = d->B * p.nrb;
    pl->inst = inst;
    pl->gy = (d->Ci + kInst[inst].ct - 1) / kInst[inst].ct;
    pl->gz = (d->Co + kTile - 1) / kTile;
    const int env = CNN_OPT_INT("SP_BLOCKS", 0);
    long long want = (env > 0 ? env : num_cus()) / ((long long)pl->gy * pl->gz);  // one workgroup per CU (LDS)
    if (want < 1) want = 1;
    if (want > p.stages_total) want = p.stages_total;
    p.stages_per_block = (int)((p.stages_total + want - 1) / want);
    pl->kblocks = (p.stages_total + p.stages_per_block - 1) / p.stages_per_block;
    return true;
}

template <int I>
int launch_inst(const Sp2Plan& pl, const cnn_conv2d_desc* d, hipStream_t s) {
    constexpr Sp2Inst c = kInst[I];
    using G = Sp2Geom<c.w, c.pad, c.rp, c.ct>;
    auto kern = wgrad_sp2_kernel<c.w, c.pad, c.rp, c.ct>;
    static DeviceOnce attr_once;
    if (attr_once.needed()) {
        CNN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds_bytes));
        attr_once.mark();
    }
    const dim3 grid(pl.kblocks, pl.gy, pl.gz);
    char name[48];
    snprintf(name, sizeof(name), "wgrad_sp2<%d,%d,%d,ct%d>", c.w, c.pad, c.rp, c.ct);
    CNN_KLAUNCH(s, name, (kern<<<grid, 256, G::lds_bytes, s>>>(pl.p)), "B%d Ci%d %dx%d Co%d k3 s2 p%d slabs%d", d->B, d->Ci, d->H, d->W, d->Co, c.pad,
                pl.kblocks);
    return CNN_AMD_OK;
}

template <int... I>
int launch_any(int inst, const Sp2Plan& pl, const cnn_conv2d_desc* d, hipStream_t s, std::integer_sequence<int, I...>) {
    int rc = CNN_AMD_E_BADARG;
    ((inst == I ? (void)(rc = launch_inst<I>(pl, d, s)) : (void)0), ...);
    return rc;
}

}  // namespace

namespace cnn_amd {

// number of partial slabs ([Co][Ci*9 + 1] floats each) the kernel writes, 0 when the geometry is not covered
int sp2_wgrad_slots(const cnn_conv2d_desc* d) {
    Sp2Plan pl;
    return make_sp2_plan(d, &pl) ? pl.kblocks : 0;
}

int sp2_wgrad_launch(const cnn_conv2d_desc* d, const float* x, const float* dy, float* slabs, hipStream_t s) {
    Sp2Plan pl;
    if (!make_sp2_plan(d, &pl)) return fail(CNN_AMD_E_BADARG, "wgrad_sp2: geometry not covered");
    pl.p.x = x; pl.p.dy = dy; pl.p.slabs = slabs;
    return launch_any(pl.inst, pl, d, s, std::make_integer_sequence<int, kNumInst>());
}

}  // namespace cnn_amd
