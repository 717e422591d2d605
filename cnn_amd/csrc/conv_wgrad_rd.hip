// conv_wgrad_rd.hip -- "register-direct" Conv2D weight / bias gradient (cpu/src/conv2d.cpp:117-159) for 3x3 filters,
// stride 1 or 2, no padding: the same batched outer-product GEMM as conv_wgrad.hip
//     gw[co][n] = sum_{b,p,q} dy[b][co][p][q] * x[b][ci_n][s*p + kx_n][s*q + ky_n],   n = (ci,kx,ky),
// on v_mfma_f32_32x32x2_f32, but WITHOUT staging operands through LDS and without barriers in the main loop.
//
// The reduction index (pixels) may be visited in any order as long as the A and B operand of an MFMA step agree.  The
// pixels are cut into RUNS of RL (16 or 8) consecutive pixels of one output row; a chunk is two consecutive runs, k-slot
// kg of MFMA step t <-> pixel t of run 2*chunk + kg (the two runs may lie in different rows or images: all addressing is
// per lane).  With q0 the first pixel of the lane's run:  Lane (m = lane%32, kg = lane/32) of the A operand then needs dy[co_m][p][q0+16kg .. +15]:
// 16 CONSECUTIVE floats (four 16-byte loads), and lane (n, kg) of the B operand needs x[ci_n][s*p+kx_n][s*(q0+16kg+t)+ky_n]:
// for stride 2 every second float of a 32-float window (eight 16-byte loads; even / odd element chosen per lane).
// Every lane therefore streams its own short contiguous runs from L1/L2 straight into MFMA operand registers; tails of
// rows are handled by guarded loads (A = 0 for missing pixels, so whatever B holds there is multiplied by 0).
//   per 32-pixel chunk and wave: 4 + 8*NT 16-byte loads for 16*NT MFMAs (NT = 32-column tiles per wave).
// Column Ntot of the output is the fused bias gradient (B operand = 1).  The four waves of a workgroup split its chunk
// range and are summed in a fixed order through LDS at the end; reduce_slabs() (conv_wgrad.hip) adds the workgroups.
#include <cstdlib>

#include "common.h"

using namespace cnn_amd;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
struct __attribute__((packed, aligned(4))) f4u {
    float x, y, z, w;
};

struct RdParams {
    const float* x;
    const float* dy;
    float* slabs;  // [gridDim.x][Co][pitch]
    int B, Ci, H, W, Co, Ho, Wo;
    int Ntot, pitch;  // Ci*9, Ntot + 1 (the bias column)
    int rpr;          // runs per output row
    int runs_total;   // B * Ho * rpr
    int chunks_total, chunks_per_block;  // chunk = 2 runs
    unsigned m_rows, m_rpr;  // magic multipliers: run -> (image*Ho + p, segment) and -> image
};

__device__ __forceinline__ int fdiv(int n, unsigned magic, int d) {
    if (d == 1) return n;  // (2^32 / 1 does not fit the 32-bit magic)
    int q = (int)__umulhi((unsigned)n, magic);
    if (q * d > n) --q;
    return q;
}

// four consecutive floats, of which only the first `nvalid` exist (the rest read as 0)
// (returned by value: writing through a pointer into the caller's array keeps that array in scratch memory)
__device__ __forceinline__ f4u load4(const float* __restrict__ p, int nvalid) {
    f4u v;
    if (nvalid >= 4) {
        v = *(const f4u*)p;
    } else {
        v.x = nvalid > 0 ? p[0] : 0.f;
        v.y = nvalid > 1 ? p[1] : 0.f;
        v.z = nvalid > 2 ? p[2] : 0.f;
        v.w = 0.f;
    }
    return v;
}

template <int S, int NT, int RL>
__global__ __launch_bounds__(256) void wgrad_rd_kernel(const RdParams p) {
    constexpr int WL = S * RL;  // floats of x a lane needs per chunk
    __shared__ float red[32][NT * 32 + 1];
    const int lane = threadIdx.x & 63, m = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int co = blockIdx.z * 32 + m;
    const int nbase = blockIdx.y * NT * 32;

    // per-lane column description for every N tile: offset of the filter tap inside an image, window shift, parity
    int xoff[NT], shift[NT];
    bool par[NT], ones[NT], nvalid_col[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nbase + nt * 32 + m;
        const int nn = n < p.Ntot ? n : 0;
        const int ci = nn / 9, kx = (nn - ci * 9) / 3, ky = nn - ci * 9 - kx * 3;
        nvalid_col[nt] = n < p.Ntot;
        ones[nt] = n == p.Ntot;
        shift[nt] = S == 2 ? (ky == 2 ? 2 : 0) : ky;
        par[nt] = S == 2 && ky == 1;
        xoff[nt] = (ci * p.H + kx) * p.W + shift[nt];
    }

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    // this wave's contiguous chunk range
    const int c_lo = blockIdx.x * p.chunks_per_block;
    const int c_hi = c_lo + p.chunks_per_block < p.chunks_total ? c_lo + p.chunks_per_block : p.chunks_total;
    const int per_wave = (c_hi - c_lo + 3) / 4;
    const int w_lo = c_lo + wave * per_wave, w_hi = w_lo + per_wave < c_hi ? w_lo + per_wave : c_hi;
    const size_t img_x = (size_t)p.Ci * p.H * p.W;

    for (int ch = w_lo; ch < w_hi; ++ch) {
        const int run = 2 * ch + kg;  // this lane's run
        const bool rlive = run < p.runs_total;
        const int runc = rlive ? run : 0;
        const int rowi = fdiv(runc, p.m_rpr, p.rpr), seg = runc - rowi * p.rpr;  // rowi = b*Ho + pr
        const int b = fdiv(rowi, p.m_rows, p.Ho), pr = rowi - b * p.Ho;
        const int q0 = seg * RL;
        int npix = rlive ? p.Wo - q0 : 0;  // valid pixels of this lane's run
        npix = npix < 0 ? 0 : (npix > RL ? RL : npix);
        // ---- A operand: RL consecutive dy values of channel co
        float a[RL];
        {
            const bool rowok = co < p.Co;
            const float* src = p.dy + (((size_t)b * p.Co + (rowok ? co : 0)) * p.Ho + pr) * p.Wo + q0;
            const int nv = rowok ? npix : 0;
#pragma unroll
            for (int j = 0; j < RL / 4; ++j) {
                const f4u v = load4(src + 4 * j, nv - 4 * j);
                a[4 * j] = v.x; a[4 * j + 1] = v.y; a[4 * j + 2] = v.z; a[4 * j + 3] = v.w;
            }
        }
        const float* ximg = p.x + (size_t)b * img_x + (size_t)(S * pr) * p.W + S * q0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            // ---- B operand: the lane's window of its filter tap's input row
            float w[WL];
            const int col = S * q0 + shift[nt];
            int rem = p.W - col;  // floats left in the input row
            rem = (nvalid_col[nt] && npix > 0) ? (rem < 0 ? 0 : rem) : 0;
            const float* src = ximg + xoff[nt];
#pragma unroll
            for (int j = 0; j < WL / 4; ++j) {
                const f4u v = load4(src + 4 * j, rem - 4 * j);
                w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
            }
#pragma unroll
            for (int t = 0; t < RL; ++t) {
                float bv = S == 2 ? (par[nt] ? w[2 * t + 1] : w[2 * t]) : w[t];
                bv = ones[nt] ? 1.f : bv;
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bv, acc[nt], 0, 0, 0);
            }
        }
    }

    // ---- sum the four waves in a fixed order, then one slab per workgroup
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * kg;
                    float* dst = &red[row][nt * 32 + m];
                    *dst = (w == 0) ? acc[nt][r] : *dst + acc[nt][r];
                }
        }
        __syncthreads();
    }
    float* slab = p.slabs + (size_t)blockIdx.x * p.Co * p.pitch;
    for (int i = threadIdx.x; i < 32 * NT * 32; i += 256) {
        const int row = i / (NT * 32), col = i - row * (NT * 32);
        const int c2 = blockIdx.z * 32 + row, n2 = nbase + col;
        if (c2 < p.Co && n2 < p.pitch) slab[(size_t)c2 * p.pitch + n2] = red[row][col];
    }
}

inline unsigned magic_of(int d) { return (unsigned)((1ull << 32) / (unsigned)d) + 1u; }

struct RdPlan {
    RdParams p;
    int nt, ngroups, mtiles, kblocks, rl;
};

bool make_rd_plan(const cnn_conv2d_desc* d, RdPlan* pl) {
    if (d->k != 3 || d->pad != 0 || (d->s != 1 && d->s != 2)) return false;
    RdParams& p = pl->p;
    p.B = d->B; p.Ci = d->Ci; p.H = d->H; p.W = d->W; p.Co = d->Co;
    p.Ho = cnn_conv2d_out_dim(d->H, 3, d->s, 0);
    p.Wo = cnn_conv2d_out_dim(d->W, 3, d->s, 0);
    if (p.Ho <= 0 || p.Wo <= 0) return false;
    p.Ntot = d->Ci * 9;
    p.pitch = p.Ntot + 1;
    pl->rl = p.Wo <= 8 ? 8 : 16;
    p.rpr = (p.Wo + pl->rl - 1) / pl->rl;
    const long long runs = (long long)p.B * p.Ho * p.rpr;
    const long long chunks = (runs + 1) / 2;
    p.runs_total = (int)runs;
    if (runs >= (1ll << 30) || (long long)p.B * p.Ci * p.H * p.W >= (1ll << 40)) return false;
    p.chunks_total = (int)chunks;
    const int tiles = (p.pitch + 31) / 32;
    pl->nt = tiles >= 5 ? 5 : tiles;  // 5 x 16 accumulator registers per wave
    if (tiles > 5 && tiles % 5 != 0 && tiles % 4 == 0) pl->nt = 4;
    if (tiles > 5 && tiles % 5 != 0 && tiles % 4 != 0 && tiles % 3 == 0) pl->nt = 3;
    pl->ngroups = (tiles + pl->nt - 1) / pl->nt;
    pl->mtiles = (p.Co + 31) / 32;
    const int env = getenv("CNN_AMD_RD_BLOCKS") ? atoi(getenv("CNN_AMD_RD_BLOCKS")) : 0;
    long long want = (env > 0 ? env : 2 * kNumCU) / ((long long)pl->ngroups * pl->mtiles);
    if (want < 1) want = 1;
    if (want > chunks) want = chunks;
    p.chunks_per_block = (int)((chunks + want - 1) / want);
    pl->kblocks = (int)((chunks + p.chunks_per_block - 1) / p.chunks_per_block);
    p.m_rows = magic_of(p.Ho);
    p.m_rpr = magic_of(p.rpr);
    return true;
}

}  // namespace

namespace cnn_amd {

// number of partial slabs ([Co][Ci*9 + 1] floats each) the kernel writes, 0 when the geometry is not covered
int wgrad_rd_slots(const cnn_conv2d_desc* d) {
    RdPlan pl;
    return make_rd_plan(d, &pl) ? pl.kblocks : 0;
}

#define CONV_TAG(d) "B%d Ci%d %dx%d Co%d k%d s%d p%d", (d)->B, (d)->Ci, (d)->H, (d)->W, (d)->Co, (d)->k, (d)->s, (d)->pad

int wgrad_rd_launch(const cnn_conv2d_desc* d, const float* x, const float* dy, float* slabs, hipStream_t s) {
    RdPlan pl;
    if (!make_rd_plan(d, &pl)) return fail(CNN_AMD_E_BADARG, "wgrad_rd: geometry not covered");
    pl.p.x = x; pl.p.dy = dy; pl.p.slabs = slabs;
    const dim3 grid(pl.kblocks, pl.ngroups, pl.mtiles);
    char name[64];
    snprintf(name, sizeof(name), "wgrad_rd<%d,%d,%d>", d->s, pl.nt, pl.rl);
#define RD(S_, NT_)                                                                                              \
    do {                                                                                                         \
        if (pl.rl == 16) CNN_KLAUNCH(s, name, (wgrad_rd_kernel<S_, NT_, 16><<<grid, 256, 0, s>>>(pl.p)), CONV_TAG(d)); \
        else CNN_KLAUNCH(s, name, (wgrad_rd_kernel<S_, NT_, 8><<<grid, 256, 0, s>>>(pl.p)), CONV_TAG(d));         \
    } while (0)
    if (d->s == 2) {
        if (pl.nt == 5) RD(2, 5); else if (pl.nt == 4) RD(2, 4); else if (pl.nt == 3) RD(2, 3); else if (pl.nt == 2) RD(2, 2); else RD(2, 1);
    } else {
        if (pl.nt == 5) RD(1, 5); else if (pl.nt == 4) RD(1, 4); else if (pl.nt == 3) RD(1, 3); else if (pl.nt == 2) RD(1, 2); else RD(1, 1);
    }
#undef RD
    return CNN_AMD_OK;
}

}  // namespace cnn_amd
