// conv_stem.hip -- Conv2D::forward (cpu/src/conv2d.cpp:34-94) for a THIN input under a LARGE strided filter: Ci = 3, 7x7, stride 2,
// pad 3 (the stem of the ResNet-18-shaped stack, BASELINE configs[4]).  On the implicit GEMM this layer ran at 20 TFLOP/s: its K
// dimension is 3*49 = 147, which the channel-chunked tap loop pads to 8 channels per tap (2.7x the MFMA work) and gathers through
// a 9-float-wide row image.  Here the contraction is laid out the other way round:
//
//   GEMM  M = Co (two 32-row MFMA tiles per 64 channels), N = 32 consecutive output pixels of one output row, K = (ci, ky, kx) =
//   147 (+1 zero) in 74 steps of v_mfma_f32_32x32x2_f32.
//
//   * A (filters) lives in LDS for the whole kernel as A[k][co] (pitch 65: conflict-free for the transposing fill and for the reads);
//     a lane's operand of step s sits at a compile-time offset from its base.
//   * B (input) is the zero-padded image rows themselves: a workgroup stages the 13 input rows x 3 channels that 4 output rows need
//     (16-byte loads, 41 KB) and a lane gathers x[ci][2r+ky][2(32t+n)+kx-3] with one ds_read_b32 per step and pixel tile (stride-2
//     across lanes = 2-way banked, far from binding at one read per 64-cycle MFMA).
//   * A wave owns one output row and two adjacent 32-pixel tiles x both 32-channel tiles: 4 MFMAs per 4 LDS reads, 64 accumulators.
//   * Workgroups are persistent over (image, 4-row group) items in image order; two fit a CU (79 KB of LDS each), so one stages
//     while the other computes.
// Output: y (+ bias), optionally the ReLU output as well (relu.cpp:25).  Traffic: x is read ~1.6x (38 MB), y written once (205 MB at
// batch 64).  [gpu] batch 64: 212 us = 71 TFLOP/s (implicit GEMM: 750 us).  CNN_AMD_STEM_DBG: without the MFMA steps 93 us (the
// 205 MB of stores + staging), without re-staging 196 us -- the MFMA loop itself runs near its 110 us floor, but a workgroup's store
// burst + the next item's staging do not overlap the MFMAs of the other workgroup on the CU (both fall into step).  Tried: fetching
// the next item's rows into registers behind the MFMAs, LDS-only barriers and the stores fired last (so that they drain under the
// next item's MFMAs): 24 more live registers spill under the 128-register budget and the kernel got slower (227 us); the store
// phase itself only reaches ~2.2 TB/s (448-byte rows: every second 128-byte segment straddles two lines); streaming stores: -2 %.
#include "common.h"

using namespace cnn_amd;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SCI = 3, SK = 7, SS = 2, SPAD = 3;
constexpr int SKK = SCI * SK * SK;        // 147
constexpr int SSTEPS = (SKK + 1) / 2;     // 74 MFMA steps (K = 148, the last column is zero)
constexpr int SR = 4;                     // output rows per workgroup item
constexpr int SRIN = SS * (SR - 1) + SK;  // 13 input rows
constexpr int SNT = 2;                    // 32-pixel tiles per wave
constexpr int SWAVES = 8;                 // 4 rows x 2 tile pairs
constexpr int SCOLS = 32 * SNT * 2;       // 128 output columns per item (>= Wo; wider layers loop over column blocks)
constexpr int SLW = (SS * (SCOLS - 1) + SK + 1 + 3) / 4 * 4;  // staged row pitch: input columns -4 .. 2*127+3 (262), rounded to 16 bytes -> 264
static_assert(SLW % 4 == 0, "16-byte staging");
constexpr int SAP = 65;                   // pitch of A[k][co] (64 channels + 1)

struct StemParams {
    const float* x;
    const float* w;     // reference layout [Co][3][7][7]
    const float* bias;
    float* y;           // nullable when y_relu is set
    float* y_relu;      // nullable
    int B, H, W, Co, Ho, Wo;
    int row_groups, col_blocks, items;  // items = B * row_groups * col_blocks
    int dbg;            // CNN_AMD_STEM_DBG (tuning): 1 = rows staged for the first item only, 2 = no MFMA steps
};

// compile-time offset (floats) of K index k inside the staged image: (ci * SRIN + ky) * SLW + kx
// (measured: splitting a staged row into its even and odd columns -- conflict-free gathers instead of 2-way banked ones -- does not
// pay: one ds_read_b32 per 64-cycle MFMA leaves the LDS pipe mostly idle either way, and the staging stores get narrower)
__host__ __device__ constexpr int stem_tap_off(int k) {
    return k >= SKK ? 0 : ((k / (SK * SK)) * SRIN + (k % (SK * SK)) / SK) * SLW + (k % SK);
}

template <bool RELU_OUT>
__global__ __launch_bounds__(SWAVES * 64, 4) void conv_stem_fwd_kernel(const StemParams p) {
    extern __shared__ float lds[];
    float* const As = lds;                       // [148][SAP]
    float* const Xs = lds + (SKK + 1) * SAP;     // [3][13][264]
    float* const Bs = Xs + SCI * SRIN * SLW;     // [64]: this channel block's bias (the epilogue reads it from LDS: a global load there would
                                                 // wait -- vmcnt counts stores too -- for every store issued before it; round 6)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, kk = lane >> 5;
    const int co0 = blockIdx.y * 64;

    // ---- filters -> A[k][co] (once per workgroup); channels beyond Co and column 147 are zero
    for (int e = tid; e < 64 * (SKK + 1); e += SWAVES * 64) {
        const int co = e / (SKK + 1), k = e - co * (SKK + 1);
        As[k * SAP + co] = (k < SKK && co0 + co < p.Co) ? p.w[(size_t)(co0 + co) * SKK + k] : 0.f;
    }
    if (tid < 64) Bs[tid] = co0 + tid < p.Co ? p.bias[co0 + tid] : 0.f;

    const int r = wave >> 1, ct0 = (wave & 1) * SNT;   // this wave's output row inside the group, first of its pixel tiles
    const int a_lane = kk * SAP + n;                    // + (2s) * SAP + mt * 32
    const int b_lane = (SS * r) * SLW + SS * (32 * ct0 + n) + 1;  // column: 2*ox + kx - 3 + 4
    for (int it = blockIdx.x; it < p.items; it += gridDim.x) {
        const int cb = it % p.col_blocks, rg = (it / p.col_blocks) % p.row_groups, b = it / (p.col_blocks * p.row_groups);
        const int oy0 = rg * SR, ox0 = cb * SCOLS;
        const int iy0 = SS * oy0 - SPAD, ix0 = SS * ox0 - 4;  // staged element (row rr, column c) = x[iy0 + rr][ix0 + c]
        __syncthreads();  // (everybody is done with the previous item's rows; first pass: A is complete)
        // ---- stage: 3 x 13 rows x 66 16-byte chunks, zero outside the image (W % 4 == 0: a chunk is inside or outside as a whole)
        // (all of a thread's chunks are loaded before the first one is stored: one memory round trip per item instead of six)
        constexpr int NCH = SCI * SRIN * (SLW / 4), NIT = (NCH + SWAVES * 64 - 1) / (SWAVES * 64);
        if (!(p.dbg == 1 && it != (int)blockIdx.x)) {
            float4 v[NIT];
#pragma unroll
            for (int q = 0; q < NIT; ++q) {
                const int e = tid + q * SWAVES * 64;
                const int row = e / (SLW / 4), c4 = e - row * (SLW / 4);
                const int ci = row / SRIN, rr = row - ci * SRIN;
                const int iy = iy0 + rr, ix = ix0 + 4 * c4;
                v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < NCH && iy >= 0 && iy < p.H && ix >= 0 && ix + 3 < p.W) v[q] = *(const float4*)(p.x + (((size_t)b * SCI + ci) * p.H + iy) * p.W + ix);
            }
#pragma unroll
            for (int q = 0; q < NIT; ++q) {
                const int e = tid + q * SWAVES * 64;
                if (e < NCH) *(float4*)(Xs + e * 4) = v[q];  // (row * SLW + 4 * c4 == 4 * e: rows are SLW / 4 chunks long)
            }
        }
        __syncthreads();

        f32x16 acc[2][SNT];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < SNT; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][t][i] = 0.f;
        const float* const ap = As + a_lane;
        const float* const bp = Xs + b_lane;
        // software pipeline by one step: the four LDS reads of step s+1 are issued in front of the four MFMAs of step s (the
        // sched barriers keep hipcc from hoisting dozens of steps' reads: 161 registers and spills under the 128-register budget)
        float a_cur[2], b_cur[SNT];
        {
            const int off = kk ? stem_tap_off(1) : stem_tap_off(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) a_cur[mt] = ap[mt * 32];
#pragma unroll
            for (int t = 0; t < SNT; ++t) b_cur[t] = bp[off + SS * 32 * t];
        }
        if (p.dbg != 2)
#pragma unroll
        for (int s = 0; s < SSTEPS; ++s) {
            float a_nxt[2], b_nxt[SNT];
            if (s + 1 < SSTEPS) {
                const int off = kk ? stem_tap_off(2 * s + 3) : stem_tap_off(2 * s + 2);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) a_nxt[mt] = ap[(2 * s + 2) * SAP + mt * 32];
#pragma unroll
                for (int t = 0; t < SNT; ++t) b_nxt[t] = bp[off + SS * 32 * t];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int t = 0; t < SNT; ++t) acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mt], b_cur[t], acc[mt][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < SSTEPS) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) a_cur[mt] = a_nxt[mt];
#pragma unroll
                for (int t = 0; t < SNT; ++t) b_cur[t] = b_nxt[t];
            }
        }

        // ---- D[row = (i & 3) + 8 (i >> 2) + 4 kk][column = n] -> y[b][co][oy][ox]: 32 consecutive pixels per register and half-wave
        const int oy = oy0 + r;
        if (oy < p.Ho) {
            const int plane = p.Ho * p.Wo;
#pragma unroll
            for (int t = 0; t < SNT; ++t) {
                const int ox = ox0 + 32 * (ct0 + t) + n;
                if (ox < p.Wo) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const int cbase = co0 + mt * 32 + 4 * kk;
                        // (32-bit element offsets: the host checks B*Co*Ho*Wo < 2^31)
                        const int obase = (b * p.Co + cbase) * plane + oy * p.Wo + ox;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int dc = (i & 3) + 8 * (i >> 2);
                            if (cbase + dc < p.Co) {
                                const float v = acc[mt][t][i] + Bs[mt * 32 + 4 * kk + dc];  // conv2d.cpp:87: the bias is added to the finished sum
                                if (p.y) p.y[obase + dc * plane] = v;
                                if constexpr (RELU_OUT) p.y_relu[obase + dc * plane] = v >= 0.f ? v : 0.f;  // relu.cpp:25 (keeps -0.0, NaN -> 0)
                            }
                        }
                    }
                }
            }
        }
    }
}


// ---- weight / bias gradient of the same layer (cpu/src/conv2d.cpp:117-159) ---------------------------------------------------------
//   gw[co][ci][ky][kx] = sum_{b,oy,ox} dy[b][co][oy][ox] * x[b][ci][2oy+ky-3][2ox+kx-3],   gb[co] = sum dy[b][co][oy][ox]
// GEMM on v_mfma_f32_16x16x4_f32: M = the 147 taps + one "tap" that reads a row of ones (its sums are the bias gradient) in ten
// 16-row tiles, N = co in 16-column tiles, K = the pixels of ONE output row, four per step.  A workgroup item = (image, output
// row): it stages the 7 zero-padded input rows x 3 channels (the forward kernel's row image, 22 KB) and the row's deltas
// TRANSPOSED to [co][pixel] (pitch 132: 16-byte rows, 2-way banked reads) -- in HBM a pixel's 64 deltas are 64 cache lines apart.
//   A (lane = tap, k = pixel 4j + kq): x[tap_off + 2 (4j + kq) + 1] -- per-lane tap offset + an immediate 32 j;
//   B (lane = co, k = pixel):          dyT[co][4j + kq].
// Wave w owns co tile w & 3 and five of the ten tap tiles: 5 MFMAs per 6 LDS reads, 20 accumulator registers, kept across ALL items
// of the (persistent) workgroup; one [Co][148] slab per workgroup at the end, summed by reduce_slabs like every other weight gradient.
constexpr int GTAPS = SKK + 1;                 // 148: taps + the ones "tap"
static_assert((GTAPS + 15) / 16 == 10, "two groups of five tap tiles");
constexpr int GXROWS = SCI * SK;               // 21 staged rows (+ 1 row of ones)
constexpr int GDP = SCOLS + 4;                 // 132: pitch of the transposed delta row
constexpr int GWAVES = 8;

struct StemGradParams {
    const float* x;
    const float* dy;
    float* slabs;  // [gridDim.x][Co][148]
    int B, H, W, Co, Ho, Wo;
    int col_blocks, items;  // items = B * Ho * col_blocks
};

__global__ __launch_bounds__(GWAVES * 64, 4) void conv_stem_wgrad_kernel(const StemGradParams p) {
    extern __shared__ float lds[];
    float* const Xs = lds;                               // [21][264], then one row of 1.0f
    float* const Ds = lds + (GXROWS + 1) * SLW;          // [64][132]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kq = lane >> 4;
    const int co0 = blockIdx.y * 64;
    const int ct = wave & 3, tg = wave >> 2;
    for (int i = tid; i < SLW; i += GWAVES * 64) Xs[GXROWS * SLW + i] = 1.f;

    // A: this lane's tap of each of its five tiles -> offset of x[ci][ky][kx] in the staged image (row 0 = input row 2oy-3, column 0 =
    // input column -4); tap 147 and the padding taps behind it read the ones row
    int a_off[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int tap = 16 * (tg * 5 + i) + m;
        a_off[i] = (tap < SKK ? (tap / SK) * SLW + tap % SK : GXROWS * SLW) + 2 * kq + 1;  // row ci*7 + ky = tap / 7 of THIS image
    }
    const int b_off = (16 * ct + m) * GDP + kq;

    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int it = blockIdx.x; it < p.items; it += gridDim.x) {
        const int cb = it % p.col_blocks, oy = (it / p.col_blocks) % p.Ho, b = it / (p.col_blocks * p.Ho);
        const int ox0 = cb * SCOLS;
        const int npix = p.Wo - ox0 < SCOLS ? p.Wo - ox0 : SCOLS;
        const int iy0 = SS * oy - SPAD, ix0 = SS * ox0 - 4;
        __syncthreads();  // (the previous item's operands have been consumed)
        for (int e = tid; e < GXROWS * (SLW / 4); e += GWAVES * 64) {
            const int row = e / (SLW / 4), c4 = e - row * (SLW / 4);
            const int ci = row / SK, rr = row - ci * SK;
            const int iy = iy0 + rr, ix = ix0 + 4 * c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < p.H && ix >= 0 && ix + 3 < p.W) v = *(const float4*)(p.x + (((size_t)b * SCI + ci) * p.H + iy) * p.W + ix);
            *(float4*)(Xs + row * SLW + 4 * c4) = v;
        }
        // the row's deltas, transposed: [co][pixel], zero behind the row's end (Wo % 4 == 0) and for channels beyond Co
        for (int e = tid; e < 64 * (SCOLS / 4); e += GWAVES * 64) {
            const int co = e / (SCOLS / 4), c4 = e - co * (SCOLS / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (4 * c4 < npix && co0 + co < p.Co)
                v = *(const float4*)(p.dy + (((size_t)b * p.Co + co0 + co) * p.Ho + oy) * p.Wo + ox0 + 4 * c4);
            *(float4*)(Ds + co * GDP + 4 * c4) = v;
        }
        __syncthreads();
        const int steps = (npix + 3) >> 2;
        const float* ap = Xs;
        const float* bp = Ds + b_off;
        for (int j = 0; j < steps; ++j) {
            const float bv = bp[4 * j];
            float av[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) av[i] = ap[a_off[i] + 8 * j];
#pragma unroll
            for (int i = 0; i < 5; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv, acc[i], 0, 0, 0);
        }
    }
    // D[row = tap 4 kq + r][column = co m] -> slab[co][tap]
    float* slab = p.slabs + (size_t)blockIdx.x * p.Co * GTAPS;
    const int co = co0 + 16 * ct + m;
    if (co < p.Co) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tap = 16 * (tg * 5 + i) + 4 * kq + r;
                if (tap < GTAPS) slab[(size_t)co * GTAPS + tap] = acc[i][r];
            }
    }
}

}  // namespace

namespace cnn_amd {

#define CONV_TAG(d) "B%d Ci%d %dx%d Co%d k%d s%d p%d", (d)->B, (d)->Ci, (d)->H, (d)->W, (d)->Co, (d)->k, (d)->s, (d)->pad

bool stem_fwd_supported(const cnn_conv2d_desc* d) {
    if (d->Ci != SCI || d->k != SK || d->s != SS || d->pad != SPAD || d->W % 4 != 0 || d->Co < 1) return false;
    const OptVal e = CNN_OPT_VAL("STEM_FWD");
    return !(e && atoi(e) == 0);
}

// w: the reference layout [Co][3][7][7] (also what cnn_conv2d_prepare_filters keeps as this layer's forward image)
int stem_forward(const cnn_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y, float* y_relu, hipStream_t s) {
    StemParams p;
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.y_relu = y_relu;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Co = d->Co;
    p.Ho = cnn_conv2d_out_dim(d->H, SK, SS, SPAD);
    p.Wo = cnn_conv2d_out_dim(d->W, SK, SS, SPAD);
    p.row_groups = (p.Ho + SR - 1) / SR;
    p.col_blocks = (p.Wo + SCOLS - 1) / SCOLS;
    const long long items = (long long)p.B * p.row_groups * p.col_blocks;
    CNN_REQUIRE(items < (1ll << 31) && (long long)p.B * p.Co * p.Ho * p.Wo < (1ll << 31), "stem_forward: tensor too large for 32-bit offsets");
    p.items = (int)items;
    p.dbg = CNN_MEASURE_INT("STEM_DBG", 0);
    const size_t lds_bytes = ((size_t)(SKK + 1) * SAP + (size_t)SCI * SRIN * SLW + 64) * sizeof(float);
    static DeviceOnce attr_once[2];
    const int which = y_relu ? 1 : 0;
    if (attr_once[which].needed()) {
        if (y_relu)
            CNN_HIP_CHECK(hipFuncSetAttribute((const void*)conv_stem_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        else
            CNN_HIP_CHECK(hipFuncSetAttribute((const void*)conv_stem_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_once[which].mark();
    }
    const int co_blocks = (d->Co + 63) / 64;
    // two workgroups per CU; every workgroup gets the same number of items where that is possible
    long long gx = 2ll * num_cus() / co_blocks;
    if (gx < 1) gx = 1;
    if (gx > items) gx = items;
    if (const OptVal e = CNN_OPT_VAL("STEM_GRID")) gx = atoi(e) > 0 ? atoi(e) : gx;  // (tuning)
    const dim3 grid((unsigned)gx, (unsigned)co_blocks);
    if (y_relu)
        CNN_KLAUNCH(s, "conv_stem_fwd<3,7,2,3>+relu", (conv_stem_fwd_kernel<true><<<grid, SWAVES * 64, lds_bytes, s>>>(p)), CONV_TAG(d));
    else
        CNN_KLAUNCH(s, "conv_stem_fwd<3,7,2,3>", (conv_stem_fwd_kernel<false><<<grid, SWAVES * 64, lds_bytes, s>>>(p)), CONV_TAG(d));
    return CNN_AMD_OK;
}

// ---- weight gradient: slabs of [Co][148] floats ([147 filter sums | 1 bias sum] per output channel), one per workgroup column
int stem_wgrad_slots(const cnn_conv2d_desc* d) {
    if (d->Ci != SCI || d->k != SK || d->s != SS || d->pad != SPAD || d->W % 8 != 0 || d->Co < 1) return 0;
    const OptVal e = CNN_OPT_VAL("STEM_WGRAD");
    if (e && atoi(e) == 0) return 0;
    const int Ho = cnn_conv2d_out_dim(d->H, SK, SS, SPAD), Wo = cnn_conv2d_out_dim(d->W, SK, SS, SPAD);
    const long long items = (long long)d->B * Ho * ((Wo + SCOLS - 1) / SCOLS);
    if (items >= (1ll << 31)) return 0;
    const int co_blocks = (d->Co + 63) / 64;
    long long gx = 2ll * num_cus() / co_blocks;  // two workgroups per CU
    if (gx < 1) gx = 1;
    if (gx > items) gx = items;
    return (int)gx;
}

int stem_wgrad_launch(const cnn_conv2d_desc* d, const float* x, const float* dy, float* slabs, hipStream_t s) {
    const int slots = stem_wgrad_slots(d);
    CNN_REQUIRE(slots > 0, "stem_wgrad: geometry not covered");
    StemGradParams p;
    p.x = x; p.dy = dy; p.slabs = slabs;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Co = d->Co;
    p.Ho = cnn_conv2d_out_dim(d->H, SK, SS, SPAD);
    p.Wo = cnn_conv2d_out_dim(d->W, SK, SS, SPAD);
    p.col_blocks = (p.Wo + SCOLS - 1) / SCOLS;
    p.items = d->B * p.Ho * p.col_blocks;
    const size_t lds_bytes = ((size_t)(GXROWS + 1) * SLW + (size_t)64 * GDP) * sizeof(float);
    static DeviceOnce attr_once;
    if (attr_once.needed()) {
        CNN_HIP_CHECK(hipFuncSetAttribute((const void*)conv_stem_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_once.mark();
    }
    const dim3 grid((unsigned)slots, (unsigned)((d->Co + 63) / 64));
    CNN_KLAUNCH(s, "conv_stem_wgrad<3,7,2,3>", (conv_stem_wgrad_kernel<<<grid, GWAVES * 64, lds_bytes, s>>>(p)), CONV_TAG(d));
    return CNN_AMD_OK;
}

}  // namespace cnn_amd
