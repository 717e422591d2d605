// conv_fwd_rd.hip -- "register-direct" Conv2D forward (cpu/src/conv2d.cpp:60-113) for the mid-size 3x3 layers of the
// reference net (stride 1 or 2, no padding, even Ci <= 64): the GEMM
//     y[co][pixel] = bias[co] + sum_{ci,kx,ky} w[co][ci][kx][ky] * x[ci][s*p + kx][s*q + ky]
// on v_mfma_f32_32x32x2_f32 with M = output channels and N = 32 consecutive output pixels (linear (b,p,q) order):
//   * A operand (filters): the workgroup's Co-slice of w sits in LDS for the whole kernel, laid out so that a step's
//     32 channels are 32 consecutive floats (conflict-free ds_read with immediate offsets);
//   * B operand (input): k-slot kg of the MFMA handles the input channels [kg*Ci/2, (kg+1)*Ci/2), so lane (pixel n, kg)
//     needs x[ci][s*p+kx][s*q + 0..2] -- THREE CONSECUTIVE floats per (ci,kx): one 12-byte load straight into the
//     operand registers feeds three MFMA steps; across lanes the addresses advance by s floats (coalesced).
//     All per-step offsets are wave-uniform (scalar base + one per-lane offset for the whole tile): no address VALU.
//   * no LDS staging of the input, no barriers after the filter upload; the loads are software-pipelined by hand one
//     channel group ahead of the MFMAs (RD_PIPE_FENCE as in conv_wgrad_rd.hip).
// The accumulators start from the bias, the epilogue stores rows of 32 consecutive pixels (128-byte segments) and,
// for the fused Conv+ReLU entry, max(y, 0) to the second tensor.
#include <cstdlib>

#include "common.h"

using namespace cnn_amd;

namespace cnn_amd {
bool igemm_preferred(const cnn_conv2d_desc* d, int mode);  // conv_igemm.hip (mode 0 forward, 1 data gradient)
}

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
struct __attribute__((packed, aligned(4))) f3u {
    float x, y, z;
};

#define RD_PIPE_FENCE(reg) asm volatile("" : "+v"(reg) : : "memory")

struct FwdRdParams {
    const float* x;
    const float* w;     // [Co][Ci][3][3]
    const float* img;   // prepared LDS images (fwd_rd_prepare_batch), one per channel group -- or null: transpose from w
    const float* bias;  // [Co]
    float* y;   // nullable when y2 is given (pre-activation not wanted)
    float* y2;  // relu(y), or null
    int B, H, W, Co, Ho, Wo, HoWo;
    int pixels, tiles;       // B*Ho*Wo, ceil(pixels / 32)
    unsigned m_howo, m_wo;   // magic multipliers
    int dbg;                 // CNN_AMD_FWD_RD_DBG=1: workgroup 0 prints its phase cycle counts
};

__device__ __forceinline__ int fdiv(int n, unsigned magic, int d) {
    if (d == 1) return n;
    int q = (int)__umulhi((unsigned)n, magic);
    if (q * d > n) --q;
    return q;
}

// S stride, CI input channels (even), MT 32-channel output tiles per wave, NW waves per workgroup, UC channels per group
template <int S, int CI, int MT, int NW, int UC>
__global__ __launch_bounds__(NW * 64) void conv_fwd_rd_kernel(const FwdRdParams p) {
    constexpr int CH = CI / 2;      // channels per k-slot
    constexpr int G = CH / UC;      // pipeline groups per tile
    constexpr int CB = MT * 32;     // output channels per workgroup
    constexpr int CBP = CB + 1;     // LDS row pitch (odd: the transposing upload below is bank-conflict free)
    static_assert(CH % UC == 0 && (CH / UC) % 2 == 0, "an even number of groups per tile (the two buffers swap per group)");
    extern __shared__ float lw[];   // [CI*9][CBP]: lw[k][c] = w[co0 + c][k]
    const int lane = threadIdx.x & 63, n = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int co0 = blockIdx.y * CB;
    const long long dbg_t0 = p.dbg ? clock64() : 0;

    constexpr int IMG = (CI * 9 * CBP + CB + 3) / 4 * 4;  // floats of one channel group's LDS image: filters, then bias
    if (p.img) {
        // prepared image: a straight 16-byte copy
        const float4* src = (const float4*)(p.img + (size_t)blockIdx.y * IMG);
        constexpr int N4 = IMG / 4, NTH = NW * 64, U = 5;
        for (int base = threadIdx.x; base < N4; base += NTH * U) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = src[base + u * NTH < N4 ? base + u * NTH : 0];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (base + u * NTH < N4) ((float4*)lw)[base + u * NTH] = v[u];
        }
    } else {
        // filter slice -> LDS, transposed; loads batched so that a workgroup pays a handful of L2 round trips, not one
        // per element (the slice is one contiguous piece of w)
        constexpr int TOT = CB * CI * 9, NTH = NW * 64, U = 9;
        static_assert(TOT % (NTH * U) == 0, "upload batches");
        const float* wsrc = p.w + (size_t)co0 * CI * 9;
        const int live_floats = (p.Co - co0 < CB ? p.Co - co0 : CB) * CI * 9;  // channels beyond Co read as 0
        for (int base = threadIdx.x; base < TOT; base += NTH * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = base + u * NTH;
                v[u] = wsrc[e < live_floats ? e : 0];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = base + u * NTH, cl = e / (CI * 9), r = e - cl * (CI * 9);
                lw[r * CBP + cl] = e < live_floats ? v[u] : 0.f;
            }
        }
        float* lbias = lw + CI * 9 * CBP;
        if (threadIdx.x < CB) lbias[threadIdx.x] = co0 + threadIdx.x < p.Co ? p.bias[co0 + threadIdx.x] : 0.f;
    }
    __syncthreads();
    f32x16 bias_r[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) bias_r[mt][r] = lw[CI * 9 * CBP + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg];

    const long long dbg_t1 = p.dbg ? clock64() : 0;
    const float* lwl = lw + (kg * CH * 9) * CBP + n;  // this lane's A column
    const unsigned plane = (unsigned)(p.H * p.W);
    const int tstep = gridDim.x * NW;

    auto locate = [&](int tile, unsigned& xoff, unsigned& yoff, bool& live) {
        const int pi = tile * 32 + n;
        live = pi < p.pixels;
        const int pic = live ? pi : p.pixels - 1;
        const int b = fdiv(pic, p.m_howo, p.HoWo), rem = pic - b * p.HoWo;
        const int pr = fdiv(rem, p.m_wo, p.Wo), q = rem - pr * p.Wo;
        xoff = (unsigned)((b * CI + kg * CH) * (int)plane + (S * pr) * p.W + S * q);
        yoff = (unsigned)((b * p.Co + co0 + 4 * kg) * p.HoWo + rem);
    };
    auto load_group = [&](f3u (&buf)[UC * 3], int g, unsigned xoff) {
#pragma unroll
        for (int u = 0; u < UC; ++u)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float* base = p.x + ((size_t)(g * UC + u) * plane + (size_t)kx * p.W);  // wave-uniform
                buf[u * 3 + kx] = *(const f3u*)(base + xoff);
            }
    };

    int tile = blockIdx.x * NW + wave;
    if (tile >= p.tiles) return;
    unsigned xoff, yoff, nxoff = 0, nyoff = 0;
    bool live, nlive = false;
    locate(tile, xoff, yoff, live);
    auto load_filters = [&](float (&a)[MT * UC * 9], int g) {
#pragma unroll
        for (int i = 0; i < UC * 9; ++i)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[i * MT + mt] = lwl[(g * UC * 9 + i) * CBP + mt * 32];
    };
    // input windows: a ring of NB groups, requested NB-1 groups (~1150 MFMA cycles each) ahead -- with one or two waves
    // per SIMD nothing else covers an HBM / Infinity-Cache miss; filter values: two buffers (LDS latency only)
    constexpr int NB = 4;
    static_assert(G % NB == 0, "the ring index must line up from tile to tile");
    f3u xb[NB][UC * 3];
    float ab[2][MT * UC * 9];
    {
        const bool more0 = tile + tstep < p.tiles;
        if (more0) locate(tile + tstep, nxoff, nyoff, nlive);
#pragma unroll
        for (int g = 0; g < NB - 1; ++g) {
            if (g < G) load_group(xb[g], g, xoff);
            else if (more0) load_group(xb[g], g - G, nxoff);
        }
    }
    load_filters(ab[0], 0);
    for (; tile < p.tiles; tile += tstep) {
        const bool more = tile + tstep < p.tiles;
        if (more) locate(tile + tstep, nxoff, nyoff, nlive);
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = bias_r[mt];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int bi = g & 1, ri = g % NB, gp = g + NB - 1;  // gp: the group requested now
            if (gp < G) load_group(xb[gp % NB], gp, xoff);
            else if (more) load_group(xb[gp % NB], gp - G, nxoff);
            load_filters(ab[bi ^ 1], g + 1 < G ? g + 1 : 0);
            RD_PIPE_FENCE(ab[bi][0]);
#pragma unroll
            for (int u = 0; u < UC; ++u)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const f3u v = xb[ri][u * 3 + kx];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const float bv = ky == 0 ? v.x : ky == 1 ? v.y : v.z;
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[bi][((u * 3 + kx) * 3 + ky) * MT + mt], bv, acc[mt], 0, 0, 0);
                    }
                }
        }
        const long long dbg_t2 = p.dbg ? clock64() : 0;
        // ---- epilogue: rows of 32 consecutive pixels
        if (live) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rowl = mt * 32 + (r & 3) + 8 * (r >> 2);  // (+ 4*kg is inside yoff)
                    if (co0 + rowl + 4 * kg < p.Co) {
                        const size_t o = (size_t)yoff + (size_t)rowl * p.HoWo;
                        const float v = acc[mt][r];
                        if (p.y) p.y[o] = v;
                        if (p.y2) p.y2[o] = v >= 0.f ? v : 0.f;  // relu.cpp:21-26 (keeps -0.0, NaN -> 0)
                    }
                }
        }
        if (p.dbg && threadIdx.x == 0 && (blockIdx.x | blockIdx.y) == 0)
            printf("conv_fwd_rd wg 0: upload %lld, to end of MFMAs %lld, stores issued %lld cycles after start\n", dbg_t1 - dbg_t0,
                   dbg_t2 - dbg_t0, clock64() - dbg_t0);
        xoff = nxoff;
        yoff = nyoff;
        live = nlive;
    }
}

// ---- small layers (conv_layer_3 / _4 of the reference net): a wave owns 16 output channels and keeps ALL their filters as
// MFMA A operands in registers --------------------------------------------------------------------------------------------
// With 13x13 or 6x6 output maps the kernel above has one or two 32-pixel tiles per wave: the LDS upload of the filter slice
// (5.6 us), the exposed load latency of a lone wave per SIMD and the 16-row epilogue are 3/4 of its 35 us.  Here
// v_mfma_f32_16x16x4_f32 runs M = 16 output channels (one slice per wave, Co/16 slices side by side in a workgroup so that
// their input reads hit L1), N = 16 consecutive output pixels, K = 4 input channels of one tap: k-slot k of step
// j = (c4, kx, ky) is channel 4 c4 + k, so a lane's input address is its pixel base + k planes + a wave-uniform offset, and
// one 12-byte row load feeds the three ky steps.  The CI*9/4 A registers (144 for CI = 64) are loaded once per wave from the
// prepared lane-major copy img[(slice*NA + j)*64 + lane] (m16f_filter_index), or gathered from the reference layout.
typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline int m16f_filter_index(int i, int Ci) {
    const int NA = Ci * 9 / 4;
    const int lane = i & 63, j = (i >> 6) % NA, slice = (i >> 6) / NA;
    return ((16 * slice + (lane & 15)) * Ci + 4 * (j / 9) + (lane >> 4)) * 9 + j % 9;
}

// NB: ring of granules (one c4: three 12-byte loads, nine MFMA steps per slice), NB-1 in flight; MS: 16-channel slices per wave
// (2 where the registers allow: every input load then feeds two MFMA chains and the per-group overhead is paid half as often)
template <int S, int CI, int NW, bool PREP, int NB, int MS>
__global__ __launch_bounds__(NW * 64) void conv_fwd_m16_kernel(const FwdRdParams p) {
    constexpr int C4 = CI / 4, NA = C4 * 9;
    static_assert(C4 % NB == 0, "static ring indices");
    const int lane = threadIdx.x & 63;
    const int n = lane & 15, k = lane >> 4;
    const int slices = (p.Co >> 4) / MS, parts = p.tiles;  // (wave slices of 16*MS channels; tiles: pixel partitions, set by the host)
    const int wid = xcd_swizzle(blockIdx.x, gridDim.x) * NW + (threadIdx.x >> 6);  // (neighbouring pixel partitions share input rows)
    const int slice = wid % slices, part = wid / slices;
    const int groups = (p.pixels + 15) >> 4;
    if (part >= parts || part >= groups) return;
    float wa[MS][NA], bs[MS][4];
#pragma unroll
    for (int m = 0; m < MS; ++m) {
        const int sl = slice * MS + m;  // 16-channel slice
        if constexpr (PREP) {
#pragma unroll
            for (int j = 0; j < NA; ++j) wa[m][j] = p.img[(sl * NA + j) * 64 + lane];
        } else {
#pragma unroll
            for (int j = 0; j < NA; ++j) wa[m][j] = p.w[((16 * sl + n) * CI + 4 * (j / 9) + k) * 9 + j % 9];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) bs[m][r] = p.bias[16 * sl + 4 * k + r];
    }
    const long long dbg_t0 = p.dbg ? clock64() : 0;
    long long dbg_t1 = 0;
    int dbg_groups = 0;
    const unsigned plane = (unsigned)(p.H * p.W);
    auto locate = [&](int g, unsigned& xoff, unsigned& yoff, bool& live) {
        const int pi = g * 16 + n;
        live = g < groups && pi < p.pixels;
        const int pic = live ? pi : p.pixels - 1;  // (dead lanes read the last pixel's window and store nothing)
        const int b = fdiv(pic, p.m_howo, p.HoWo), rem = pic - b * p.HoWo;
        const int pr = fdiv(rem, p.m_wo, p.Wo), q = rem - pr * p.Wo;
        xoff = (unsigned)((b * CI + k) * (int)plane + (S * pr) * p.W + S * q);
        yoff = (unsigned)((b * p.Co + 16 * MS * slice + 4 * k) * p.HoWo + rem);
    };
    // (12-byte loads through a pointer: hipcc lowers __builtin_amdgcn_raw_buffer_load_b96 to ONE dword load here)
    auto load_g = [&](f3u (&buf)[3], int c4, unsigned xoff) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float* base = p.x + ((size_t)(c4 * 4) * plane + (size_t)kx * p.W);  // wave-uniform
            buf[kx] = *(const f3u*)(base + xoff);
        }
    };
    unsigned xoff, yoff, nxoff, nyoff;
    bool live, nlive;
    locate(part, xoff, yoff, live);
    f3u xb[NB][3];
#pragma unroll
    for (int i = 0; i < NB - 1; ++i) load_g(xb[i], i, xoff);
    for (int g = part; g < groups; g += parts) {
        locate(g + parts, nxoff, nyoff, nlive);
        constexpr int NACC = 2;  // partial sums, used round-robin (a dependent MFMA waits for the previous write-back)
        f32x4 acc[MS][NACC];
#pragma unroll
        for (int m = 0; m < MS; ++m) {
            acc[m][0] = f32x4{bs[m][0], bs[m][1], bs[m][2], bs[m][3]};
#pragma unroll
            for (int a = 1; a < NACC; ++a) acc[m][a] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int c4 = 0; c4 < C4; ++c4) {
            const int gn = c4 + NB - 1;  // requested now (from c4 = C4-3 on: the first granules of the next group)
            load_g(xb[gn % NB], gn % C4, gn < C4 ? xoff : nxoff);
            RD_PIPE_FENCE(xb[c4 % NB][0].x);
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const f3u v = xb[c4 % NB][i / 3];
                const float bv = i % 3 == 0 ? v.x : i % 3 == 1 ? v.y : v.z;
                const int a = (c4 * 9 + i) % NACC;
#pragma unroll
                for (int m = 0; m < MS; ++m) {
                    acc[m][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[m][c4 * 9 + i], bv, acc[m][a], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (live) {
#pragma unroll
            for (int mr = 0; mr < MS * 4; ++mr) {
                const int m = mr >> 2, r = mr & 3;
                const float v = acc[m][0][r] + acc[m][1][r];
                const size_t o = (size_t)yoff + (size_t)(16 * m + r) * p.HoWo;
                if (p.y) p.y[o] = v;
                if (p.y2) p.y2[o] = v >= 0.f ? v : 0.f;  // relu.cpp:21-26 (keeps -0.0, NaN -> 0)
            }
        }
        xoff = nxoff;
        yoff = nyoff;
        live = nlive;
        if (p.dbg) {
            if (!dbg_groups) dbg_t1 = clock64();
            ++dbg_groups;
        }
    }
    if (p.dbg && threadIdx.x == 0 && blockIdx.x == 0)
        printf("conv_fwd_m16 wg 0: first group done %lld, all %d groups %lld cycles after the filter loads were issued\n", dbg_t1 - dbg_t0, dbg_groups,
               clock64() - dbg_t0);
}

inline unsigned magic_of(int d) { return (unsigned)((1ull << 32) / (unsigned)d) + 1u; }

struct FwdRdPlan {
    FwdRdParams p;
    int s, ci, mt, nw, cgroups, blocks_x;
    bool m16;  // conv_fwd_m16_kernel (small layers); its prepared image is a Co*Ci*9-float permutation of the filters
    int m16_blocks, m16_parts, m16_ms;
    size_t lds, img_floats;
};

bool make_plan(const cnn_conv2d_desc* d, FwdRdPlan* pl) {
    if (d->k != 3 || d->pad != 0 || (d->s != 1 && d->s != 2)) return false;
    if (d->Ci != 16 && d->Ci != 32 && d->Ci != 64) return false;
    if (const OptVal e = CNN_OPT_VAL("FWD_RD"))
        if (atoi(e) == 0) return false;
    FwdRdParams& p = pl->p;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Co = d->Co;
    p.Ho = cnn_conv2d_out_dim(d->H, 3, d->s, 0);
    p.Wo = cnn_conv2d_out_dim(d->W, 3, d->s, 0);
    if (p.Ho <= 0 || p.Wo <= 0) return false;
    p.HoWo = p.Ho * p.Wo;
    const long long pixels = (long long)d->B * p.HoWo;
    if (pixels >= (1ll << 30) || (long long)d->B * d->Ci * d->H * d->W >= (1ll << 31) || pixels * d->Co >= (1ll << 31)) return false;
    p.pixels = (int)pixels;
    p.tiles = (int)((pixels + 31) / 32);
    p.m_howo = magic_of(p.HoWo);
    p.m_wo = magic_of(p.Wo);
    p.dbg = CNN_MEASURE_INT("FWD_RD_DBG", 0);
    pl->s = d->s;
    pl->ci = d->Ci;
    const int mtiles = (d->Co + 31) / 32;
    // output tiles per wave: keep the filter slice within LDS and the grid above one wave per SIMD
    // two tiles per wave halve the input reads, but only pay once every wave slot still gets a few items
    int mt = mtiles >= 2 && (long long)p.tiles * ((mtiles + 1) / 2) >= 4 * 4 * num_cus() ? 2 : 1;
    if ((size_t)(mt * 32 + 1) * d->Ci * 9 * 4 > 96 * 1024) mt = 1;
    if (const OptVal e = CNN_OPT_VAL("FWD_RD_MT")) mt = atoi(e) == 2 && mtiles >= 2 ? 2 : 1;
    pl->mt = mt;
    pl->cgroups = (mtiles + mt - 1) / mt;
    pl->img_floats = ((size_t)(mt * 32 + 1) * d->Ci * 9 + mt * 32 + 3) / 4 * 4;  // filters + bias, whole float4s
    pl->lds = pl->img_floats * sizeof(float);
    pl->nw = pl->lds > 80 * 1024 ? 8 : 4;  // a filter slice that allows one workgroup per CU gets 8 waves
    const int env = CNN_OPT_INT("FWD_RD_BLOCKS", 0);
    long long bx = (env > 0 ? env : (pl->nw == 8 ? num_cus() : 2 * num_cus())) / pl->cgroups;
    const long long need = (p.tiles + pl->nw - 1) / pl->nw;
    if (bx > need) bx = need;
    if (bx < 1) bx = 1;
    pl->blocks_x = (int)bx;
    // small layers: fewer than four 32-pixel wave tiles per SIMD for the kernel above
    const bool m16_ok = d->Co % 16 == 0 && (long long)d->B * d->Ci * d->H * d->W < (1ll << 29);  // (32-bit buffer offsets)
    // ... and the thin Ci = 16 layers (conv_layer_2: 27.7 vs 33 us on the kernel above), unless CNN_AMD_FWD_M16_THIN=0
    const bool thin = d->Ci == 16 && d->Co % 32 == 0 && !((CNN_OPT_SET("FWD_M16_THIN") && CNN_OPT_INT("FWD_M16_THIN", 0) == 0));
    pl->m16 = m16_ok && ((long long)p.tiles * mtiles < 4 * 4 * num_cus() || thin);
    if (const OptVal e = CNN_OPT_VAL("FWD_M16")) pl->m16 = m16_ok && atoi(e) != 0;
    if (pl->m16) {
        // two slices per wave where 2 * Ci*9/4 filter registers fit beside the rest (Ci <= 32)
        pl->m16_ms = (d->Co % 32 == 0 && d->Ci <= 32 && !((CNN_OPT_SET("FWD_M16_MS") && CNN_OPT_INT("FWD_M16_MS", 0) == 1))) ? 2 : 1;
        const int slices = d->Co / 16 / pl->m16_ms, groups = (int)((pixels + 15) / 16);
        // waves per SIMD, measured: Ci >= 32: 1 beats 2-4 (the filter preamble is per wave); Ci = 16: 2 (29.8 / 27.7 / 33.5 us for 1 / 2 / 3)
        int per_simd = CNN_OPT_INT("FWD_M16_WAVES", (d->Ci == 16 ? 2 : 1));
        if (per_simd < 1 || per_simd > 8) per_simd = 1;
        int parts = per_simd * 4 * num_cus() / slices;
        if (parts < 1) parts = 1;
        if (parts > groups) parts = groups;
        pl->m16_parts = parts;
        pl->m16_blocks = (parts * slices + 3) / 4;
    }
    // the big-layer kernel only: cnn_conv2d_autotune may have measured the implicit GEMM faster for this geometry
    if (!pl->m16 && cnn_amd::igemm_preferred(d, 0)) return false;
    return true;
}

template <int S, int CI, int MT, int NW>
int launch(const FwdRdPlan& pl, hipStream_t s, const char* name, const cnn_conv2d_desc* d) {
    auto kern = conv_fwd_rd_kernel<S, CI, MT, NW, 2>;
    static DeviceOnce attr_once;
    if (attr_once.needed()) {
        CNN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_once.mark();
    }
    const dim3 grid(pl.blocks_x, pl.cgroups);
    CNN_KLAUNCH(s, name, (kern<<<grid, NW * 64, pl.lds, s>>>(pl.p)), "B%d Ci%d %dx%d Co%d k%d s%d p%d", d->B, d->Ci, d->H, d->W,
                d->Co, d->k, d->s, d->pad);
    return CNN_AMD_OK;
}

struct FwdRdPrepJob {
    int kind;  // 0: forward LDS image of this file | 1: data-gradient filter copy of conv_dgrad_rd.hip | 2: conv_fwd_m16_kernel's A operands
    const float* w;
    const float* bias;
    float* img;
    int Co, Ci, cb, img_floats, cgroups, tr;
};
struct FwdRdPrepBatch {
    FwdRdPrepJob job[12];
};

// one workgroup column per job.  kind 0: img[g][k*CBP + c] = w[g*cb + c][k], then the group's bias;
// kind 1: img[(co*9 + tap)*Ci + ci] = w[(co*Ci + ci)*9 + tap] (tr = 1), the m16 lane-major order (tr = 2) or a verbatim copy
__global__ __launch_bounds__(256) void fwd_rd_prepare_kernel(const FwdRdPrepBatch pb) {
    const FwdRdPrepJob j = pb.job[blockIdx.y];
    if (j.kind == 2) {  // conv_fwd_m16_kernel: lane-major A operands
        const int total = j.Co * j.Ci * 9;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) j.img[i] = j.w[m16f_filter_index(i, j.Ci)];
        return;
    }
    if (j.kind == 1) {
        const int total = j.Co * j.Ci * 9;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
            if (j.tr == 2) {  // lane-major MFMA A operands of the Ci = 16 data-gradient kernel
                // (Co = 128: two halves of 64 dy channels, [half][slice][j][lane])
                const int coh = j.Co > 64 ? j.Co / 2 : j.Co, na = coh * 9 / 4, slices = j.Ci / 16;
                const int jj = (i >> 6) % na, blk = (i >> 6) / na, slice = blk % slices, half = blk / slices;
                j.img[i] = j.w[half * coh * j.Ci * 9 + m16_filter_index(jj, i & 63, j.Ci, slice)];
            } else if (j.tr) {
                const int ci = i % j.Ci, r = i / j.Ci, tap = r % 9, co = r / 9;
                j.img[i] = j.w[((size_t)co * j.Ci + ci) * 9 + tap];
            } else {
                j.img[i] = j.w[i];
            }
        }
        return;
    }
    const int K = j.Ci * 9, cbp = j.cb + 1, total = j.cgroups * j.img_floats;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int g = i / j.img_floats, o = i - g * j.img_floats;
        float v = 0.f;
        if (o < K * cbp) {
            const int k = o / cbp, c = o - k * cbp, co = g * j.cb + c;
            if (c < j.cb && co < j.Co) v = j.w[(size_t)co * K + k];
        } else if (o < K * cbp + j.cb) {
            const int co = g * j.cb + (o - K * cbp);
            if (co < j.Co) v = j.bias[co];
        }
        j.img[i] = v;
    }
}

}  // namespace

namespace cnn_amd {

bool fwd_rd_supported(const cnn_conv2d_desc* d) {
    FwdRdPlan pl;
    return make_plan(d, &pl);
}
// covered by the small-layer (16x16x4, filters in registers) kernel: never handed to the implicit GEMM
bool fwd_rd_small(const cnn_conv2d_desc* d) {
    FwdRdPlan pl;
    return make_plan(d, &pl) && pl.m16;
}

// floats of the prepared filter images of a layer (0: not an RD layer)
size_t fwd_rd_prepared_floats(const cnn_conv2d_desc* d) {
    FwdRdPlan pl;
    if (!make_plan(d, &pl)) return 0;
    const size_t a = pl.img_floats * pl.cgroups, b = (size_t)d->Co * d->Ci * 9;  // (LDS images | the m16 permutation)
    return a > b ? a : b;
}

int dgrad_rd_prepare_layout(const cnn_conv2d_desc* d, int* transposed);  // conv_dgrad_rd.hip

// prepares, in ONE launch, the forward images of every layer this file covers and the data-gradient filter copies of the
// layers conv_dgrad_rd.hip covers; sets their bits in *fdone / *ddone
int rd_prepare_batch(int n, const cnn_conv2d_desc* descs, const float* const* w, const float* const* bias, void* const* fwd,
                     void* const* dgrad, hipStream_t s, unsigned* fdone, unsigned* ddone) {
    FwdRdPrepBatch pb;
    int jobs = 0;
    size_t most = 0;
    for (int i = 0; i < n && i < 6; ++i) {
        FwdRdPlan pl;
        if (fwd && fwd[i] && !(*fdone >> i & 1u) && make_plan(&descs[i], &pl)) {
            CNN_REQUIRE(w[i] && bias[i], "cnn_conv2d_prepare_filters: filters / bias of layer %d are null", i);
            FwdRdPrepJob& j = pb.job[jobs++];
            j.kind = pl.m16 ? 2 : 0; j.w = w[i]; j.bias = bias[i]; j.img = (float*)fwd[i];
            j.Co = descs[i].Co; j.Ci = descs[i].Ci; j.cb = pl.mt * 32; j.img_floats = (int)pl.img_floats; j.cgroups = pl.cgroups; j.tr = 0;
            if (pl.img_floats * pl.cgroups > most) most = pl.img_floats * pl.cgroups;
            *fdone |= 1u << i;
        }
        int tr = 0;
        if (dgrad && dgrad[i] && !(*ddone >> i & 1u) && dgrad_rd_prepare_layout(&descs[i], &tr)) {
            CNN_REQUIRE(w[i] != nullptr, "cnn_conv2d_prepare_filters: filters of layer %d are null", i);
            FwdRdPrepJob& j = pb.job[jobs++];
            j.kind = 1; j.w = w[i]; j.bias = nullptr; j.img = (float*)dgrad[i];
            j.Co = descs[i].Co; j.Ci = descs[i].Ci; j.cb = 0; j.img_floats = 0; j.cgroups = 0; j.tr = tr;
            const size_t nf = (size_t)descs[i].Co * descs[i].Ci * 9;
            if (nf > most) most = nf;
            *ddone |= 1u << i;
        }
    }
    if (jobs) {
        unsigned gx = (unsigned)((most + 255) / 256);
        if (gx > 512) gx = 512;
        CNN_KLAUNCH(s, "rd_prepare", (fwd_rd_prepare_kernel<<<dim3(gx, jobs), 256, 0, s>>>(pb)), "jobs=%d", jobs);
    }
    return CNN_AMD_OK;
}

// w == nullptr: `img` holds the prepared images
int fwd_rd_forward(const cnn_conv2d_desc* d, const float* x, const float* w, const float* img, const float* bias, float* y,
                   float* y_relu, hipStream_t s) {
    FwdRdPlan pl;
    if (!make_plan(d, &pl)) return fail(CNN_AMD_E_BADARG, "conv_fwd_rd: geometry not covered");
    pl.p.x = x; pl.p.w = w; pl.p.img = w ? nullptr : img; pl.p.bias = bias; pl.p.y = y; pl.p.y2 = y_relu;
    char name[64];
    if (pl.m16) {
        pl.p.tiles = pl.m16_parts;
        snprintf(name, sizeof(name), "conv_fwd_rd<%d,%d,m16>/fwd%s", d->s, d->Ci, y_relu ? "+relu" : "");
#define M16K(S_, CI_, PREP_, MS_)                                                                                                            \
    CNN_KLAUNCH(s, name, (launch_pub(conv_fwd_m16_kernel<S_, CI_, 4, PREP_, 4, MS_>, dim3(pl.m16_blocks), dim3(256), 0, s, pl.p)), "B%d Ci%d %dx%d Co%d k%d s%d p%d", \
                d->B, d->Ci, d->H, d->W, d->Co, d->k, d->s, d->pad)
#define M16(S_, CI_)                                                          \
    do {                                                                      \
        if (pl.m16_ms == 2 && CI_ <= 32) {                                    \
            if (pl.p.img) M16K(S_, (CI_ <= 32 ? CI_ : 32), true, 2);         \
            else M16K(S_, (CI_ <= 32 ? CI_ : 32), false, 2);                 \
        } else {                                                              \
            if (pl.p.img) M16K(S_, CI_, true, 1);                             \
            else M16K(S_, CI_, false, 1);                                     \
        }                                                                     \
    } while (0)
        if (d->s == 2) { if (d->Ci == 16) M16(2, 16); else if (d->Ci == 32) M16(2, 32); else M16(2, 64); }
        else { if (d->Ci == 16) M16(1, 16); else if (d->Ci == 32) M16(1, 32); else M16(1, 64); }
#undef M16K
#undef M16
        return CNN_AMD_OK;
    }
    snprintf(name, sizeof(name), "conv_fwd_rd<%d,%d,%d>/fwd%s", d->s, d->Ci, pl.mt, y_relu ? "+relu" : "");
#define GO(S_, CI_)                                                                                  \
    do {                                                                                             \
        if (pl.mt == 2 && pl.nw == 8) return launch<S_, CI_, 2, 8>(pl, s, name, d);                  \
        if (pl.mt == 2) return launch<S_, CI_, 2, 4>(pl, s, name, d);                                \
        if (pl.nw == 8) return launch<S_, CI_, 1, 8>(pl, s, name, d);                                \
        return launch<S_, CI_, 1, 4>(pl, s, name, d);                                                \
    } while (0)
    if (d->s == 2) {
        if (d->Ci == 16) GO(2, 16); else if (d->Ci == 32) GO(2, 32); else GO(2, 64);
    } else {
        if (d->Ci == 16) GO(1, 16); else if (d->Ci == 32) GO(1, 32); else GO(1, 64);
    }
#undef GO
    return CNN_AMD_OK;
}

}  // namespace cnn_amd
