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
// burst + the next item's staging do not overlap the MFMAs of the other workgroup on the CU (both fall into step).  Next: issue the
// next item's rows as LDS-DMA behind the barrier, wait for THEM, then fire the stores and compute while they drain.
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
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, kk = lane >> 5;
    const int co0 = blockIdx.y * 64;

    // ---- filters -> A[k][co] (once per workgroup); channels beyond Co and column 147 are zero
    for (int e = tid; e < 64 * (SKK + 1); e += SWAVES * 64) {
        const int co = e / (SKK + 1), k = e - co * (SKK + 1);
        As[k * SAP + co] = (k < SKK && co0 + co < p.Co) ? p.w[(size_t)(co0 + co) * SKK + k] : 0.f;
    }

    const int r = wave >> 1, ct0 = (wave & 1) * SNT;   // this wave's output row inside the group, first of its pixel tiles
    const int a_lane = kk * SAP + n;                    // + (2s) * SAP + mt * 32
    const int b_lane = (SS * r) * SLW + SS * (32 * ct0 + n) + 1;  // column: 2*ox + kx - 3 + 4
    for (int it = blockIdx.x; it < p.items; it += gridDim.x) {
        const int cb = it % p.col_blocks, rg = (it / p.col_blocks) % p.row_groups, b = it / (p.col_blocks * p.row_groups);
        const int oy0 = rg * SR, ox0 = cb * SCOLS;
        const int iy0 = SS * oy0 - SPAD, ix0 = SS * ox0 - 4;  // staged element (row rr, column c) = x[iy0 + rr][ix0 + c]
        __syncthreads();  // (everybody is done with the previous item's rows; first pass: A is complete)
        // ---- stage: 3 x 13 rows x 66 16-byte chunks, zero outside the image (W % 4 == 0: a chunk is inside or outside as a whole)
        for (int e = tid; e < SCI * SRIN * (SLW / 4) && !(p.dbg == 1 && it != (int)blockIdx.x); e += SWAVES * 64) {
            const int row = e / (SLW / 4), c4 = e - row * (SLW / 4);
            const int ci = row / SRIN, rr = row - ci * SRIN;
            const int iy = iy0 + rr, ix = ix0 + 4 * c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < p.H && ix >= 0 && ix + 3 < p.W) v = *(const float4*)(p.x + (((size_t)b * SCI + ci) * p.H + iy) * p.W + ix);
            *(float4*)(Xs + row * SLW + 4 * c4) = v;
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
                                const float v = acc[mt][t][i] + p.bias[cbase + dc];  // conv2d.cpp:87: the bias is added to the finished sum
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

}  // namespace

namespace cnn_amd {

#define CONV_TAG(d) "B%d Ci%d %dx%d Co%d k%d s%d p%d", (d)->B, (d)->Ci, (d)->H, (d)->W, (d)->Co, (d)->k, (d)->s, (d)->pad

bool stem_fwd_supported(const cnn_conv2d_desc* d) {
    if (d->Ci != SCI || d->k != SK || d->s != SS || d->pad != SPAD || d->W % 4 != 0 || d->Co < 1) return false;
    const char* e = getenv("CNN_AMD_STEM_FWD");
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
    p.dbg = getenv("CNN_AMD_STEM_DBG") ? atoi(getenv("CNN_AMD_STEM_DBG")) : 0;
    const size_t lds_bytes = ((size_t)(SKK + 1) * SAP + (size_t)SCI * SRIN * SLW) * sizeof(float);
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
    long long gx = 2ll * kNumCU / co_blocks;
    if (gx < 1) gx = 1;
    if (gx > items) gx = items;
    if (const char* e = getenv("CNN_AMD_STEM_GRID")) gx = atoi(e) > 0 ? atoi(e) : gx;  // (tuning)
    const dim3 grid((unsigned)gx, (unsigned)co_blocks);
    if (y_relu)
        CNN_KLAUNCH(s, "conv_stem_fwd<3,7,2,3>+relu", (conv_stem_fwd_kernel<true><<<grid, SWAVES * 64, lds_bytes, s>>>(p)), CONV_TAG(d));
    else
        CNN_KLAUNCH(s, "conv_stem_fwd<3,7,2,3>", (conv_stem_fwd_kernel<false><<<grid, SWAVES * 64, lds_bytes, s>>>(p)), CONV_TAG(d));
    return CNN_AMD_OK;
}

}  // namespace cnn_amd
