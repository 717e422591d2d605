// conv_wgrad_rd.hip -- "register-direct" Conv2D weight / bias gradient (cpu/src/conv2d.cpp:117-159) for 3x3 filters,
// stride 1 or 2, no padding: the same batched outer-product GEMM as conv_wgrad.hip
//     gw[co][n] = sum_{b,p,q} dy[b][co][p][q] * x[b][ci_n][s*p + kx_n][s*q + ky_n],   n = (ci,kx,ky),
// on v_mfma_f32_32x32x2_f32, but WITHOUT staging operands through LDS and without barriers in the main loop.
//
// The reduction index (pixels) may be visited in any order as long as the A and B operand of an MFMA step agree.  The
// pixels are cut into RUNS of RL (16 or 8) consecutive pixels of one output row; a chunk is two consecutive runs, k-slot
// kg of MFMA step t <-> pixel t of run 2*chunk + kg (the two runs may lie in different rows or images: all addressing is
// per lane).  With q0 the first pixel of the lane's run, lane (m = lane%32, kg = lane/32) of the A operand needs
// dy[co_m][p][q0 .. q0+RL-1]: RL CONSECUTIVE floats (16-byte loads), and lane (n, kg) of the B operand needs
// x[ci_n][s*p+kx_n][s*(q0+t)+ky_n]: for stride 2 every second float of a 2*RL-float window (even / odd element chosen
// per lane).  Every lane therefore streams its own short contiguous windows from L1/L2 straight into MFMA operand
// registers:  per chunk and wave  RL/4 + S*RL/4*NT 16-byte loads for RL*NT MFMAs (NT = 32-column tiles per wave).
//
// Pipelined path (all but the last few chunks of the tensors): the loads are unconditional -- a window may run over the
// end of its row into whatever follows, those k-slots are masked to zero on both operands -- and software-pipelined by
// hand with two window buffers: the window of tile nt+1 (behind the last tile: the next chunk's A run and first window)
// is in flight while the MFMAs of tile nt issue.  Measured on the north-star shape: 8.44 M shader cycles per workgroup
// against 7.88 M cycles of pure MFMA issue (93 %); a per-tile ring of NT buffers was slower (9.12 M).
// Guarded path: element-wise guarded loads for the chunks whose windows could leave the allocation.
//
// The bias gradient (sum of dy) is accumulated on the VALU from the A registers (no ones-column: Ci*9 = 576 columns are
// exactly 18 tiles).  The four waves of a workgroup split its chunk range and are summed in a fixed order through LDS at
// the end; reduce_slabs() (conv_wgrad.hip) adds the workgroups' slabs.
#include <cstdlib>
#include <type_traits>

#include "common.h"

using namespace cnn_amd;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
struct __attribute__((packed, aligned(4))) f4u {
    float x, y, z, w;
};

// Loads may not cross this (it might write memory) and the MFMAs that consume `reg` may not rise above it: pins the
// hand-made software pipeline (instruction selection otherwise sinks every prefetch down to its first use, and
// __builtin_amdgcn_sched_barrier only binds the later machine scheduler).
#define RD_PIPE_FENCE(reg) asm volatile("" : "+v"(reg) : : "memory")

struct RdParams {
    const float* x;
    const float* dy;       // POOLED kernels: dpool, the delta of the 2x2 / stride-2 pool output behind this layer's ReLU
    const int* pmask;      // POOLED: the pool's argmax mask and forward output; dy[co][p][q] is rebuilt as
    const float* pooled;   //   (pmask[w] == co*Ho*Wo + p*Wo + q && !(pooled[w] <= 0)) ? dpool[w] : 0,  w = window (p/2, q/2)
    int PHo, PWo;
    float* slabs;  // [gridDim.x][Co][pitch]
    int B, Ci, H, W, Co, Ho, Wo;
    int Ntot, pitch;  // Ci*9, Ntot + 1 (column Ntot = bias gradient)
    int rpr;          // runs per output row
    int runs_total;   // B * Ho * rpr
    int chunks_total, chunks_per_block;  // chunk = 2 runs
    unsigned m_rows, m_rpr;  // magic multipliers: run -> (image*Ho + p, segment) and -> image
    int chunks_fast;  // chunks [0, chunks_fast) may over-read their windows without leaving x / dy (host-checked)
    int dbg;          // CNN_AMD_RD_DBG=9: workgroup 0 prints its shader-cycle count and the clock it ran at
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

template <int S, int NT, int RL, bool POOLED>
__global__ __launch_bounds__(256) void wgrad_rd_kernel(const RdParams p) {
    constexpr int WL = S * RL;  // floats of x a lane needs per chunk and tile
    __shared__ float red[32][NT * 32 + 1];  // (+1: bank padding; the column doubles as the bias-gradient slot)
    const int lane = threadIdx.x & 63, m = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int co = blockIdx.z * 32 + m;
    const int nbase = blockIdx.y * NT * 32;
    const int nt_live = (p.Ntot - nbase + 31) / 32;  // tiles of this group that hold at least one column (block-uniform)

    // per-lane column description for every N tile: offset of the filter tap inside an image, window shift, parity
    int xoff[NT], shift[NT];
    bool par[NT], nvalid_col[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nbase + nt * 32 + m;
        const int nn = n < p.Ntot ? n : 0;
        const int ci = nn / 9, kx = (nn - ci * 9) / 3, ky = nn - ci * 9 - kx * 3;
        nvalid_col[nt] = n < p.Ntot;
        shift[nt] = S == 2 ? (ky == 2 ? 2 : 0) : ky;
        par[nt] = S == 2 && ky == 1;
        xoff[nt] = (ci * p.H + kx) * p.W + shift[nt];
    }

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    float bsum = 0.f;  // this lane's share of sum(dy[co]) (both k-groups see disjoint pixels)
    const long long dbg_t0 = p.dbg == 9 ? clock64() : 0, dbg_w0 = p.dbg == 9 ? wall_clock64() : 0;

    // this wave's contiguous chunk range
    const int c_lo = blockIdx.x * p.chunks_per_block;
    const int c_hi = c_lo + p.chunks_per_block < p.chunks_total ? c_lo + p.chunks_per_block : p.chunks_total;
    const int per_wave = (c_hi - c_lo + 3) / 4;
    const int w_lo = c_lo + wave * per_wave, w_hi = w_lo + per_wave < c_hi ? w_lo + per_wave : c_hi;
    const size_t img_x = (size_t)p.Ci * p.H * p.W;

    auto add_bias = [&](const float (&a)[RL]) {
        float s4[RL / 4];
#pragma unroll
        for (int j = 0; j < RL / 4; ++j) s4[j] = (a[4 * j] + a[4 * j + 1]) + (a[4 * j + 2] + a[4 * j + 3]);
        float s = s4[0];
#pragma unroll
        for (int j = 1; j < RL / 4; ++j) s += s4[j];
        bsum += s;
    };

    // ---- pipelined path
    const int f_hi = w_hi < p.chunks_fast ? w_hi : p.chunks_fast;
    int s_lo = w_lo;
    if (w_lo < f_hi) {
        s_lo = f_hi;
        const int co_c = co < p.Co ? co : 0;
        unsigned cur_x = 0, nxt_x = 0, nxt_a = 0;
        int cur_nv = 0, nxt_nv = 0, cur_nb = 0, nxt_nb = 0;  // live pixels of the lane's run: A side (0 for co >= Co) / B side
        int cur_e = 0, nxt_e = 0;  // POOLED: flat index (within the sample) of the first pixel of the lane's run, channel co
        auto locate = [&](int ch, unsigned& aoff, unsigned& xb, int& nv, int& nb, int& eidx) {
            const int run = 2 * ch + kg;
            const bool rlive = run < p.runs_total;
            const int runc = rlive ? run : 0;
            const int rowi = fdiv(runc, p.m_rpr, p.rpr), seg = runc - rowi * p.rpr;
            const int b = fdiv(rowi, p.m_rows, p.Ho), pr = rowi - b * p.Ho;
            const int q0 = seg * RL;
            int npix = p.Wo - q0;
            npix = npix > RL ? RL : npix;
            nb = rlive ? npix : 0;
            nv = co < p.Co ? nb : 0;
            if constexpr (POOLED) {
                int inwin = 2 * p.PWo - q0;  // pixels of the run that lie inside a pooling window (none in an uncovered row)
                inwin = (pr >> 1) < p.PHo ? (inwin < 0 ? 0 : inwin) : 0;
                nv = nv < inwin ? nv : inwin;
                aoff = (unsigned)(((b * p.Co + co_c) * p.PHo + (pr >> 1)) * p.PWo + (q0 >> 1));
                eidx = (co_c * p.Ho + pr) * p.Wo + q0;
            } else {
                aoff = (unsigned)(((b * p.Co + co_c) * p.Ho + pr) * p.Wo + q0);
                eidx = 0;
            }
            xb = (unsigned)(b * (p.Ci * p.H * p.W) + (S * pr) * p.W + S * q0);
        };
        constexpr int NA = POOLED ? 3 * (RL / 8) : RL / 4;  // 16-byte pieces of the A side: dy run | dpool, mask, pooled of RL/2 windows
        auto load_a = [&](f4u (&buf)[NA], unsigned aoff) {
            if constexpr (POOLED) {
#pragma unroll
                for (int j = 0; j < RL / 8; ++j) {
                    buf[j] = *(const f4u*)(p.dy + aoff + 4 * j);
                    buf[RL / 8 + j] = *(const f4u*)((const float*)p.pmask + aoff + 4 * j);
                    buf[2 * (RL / 8) + j] = p.pooled ? *(const f4u*)(p.pooled + aoff + 4 * j) : f4u{1.f, 1.f, 1.f, 1.f};
                }
            } else {
#pragma unroll
                for (int j = 0; j < RL / 4; ++j) buf[j] = *(const f4u*)(p.dy + aoff + 4 * j);
            }
        };
        auto f4_at = [](const f4u& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; };
        // A registers of the current run: live pixels only, POOLED: MaxPool2D::backward + ReLU::backward on the fly
        auto build_a = [&](float (&a)[RL], const f4u (&buf)[NA]) {
#pragma unroll
            for (int t = 0; t < RL; ++t) {
                if constexpr (POOLED) {
                    const int w = t >> 1;
                    const float g = f4_at(buf[w / 4], w & 3), pl = f4_at(buf[2 * (RL / 8) + w / 4], w & 3);
                    const int mk = __builtin_bit_cast(int, f4_at(buf[RL / 8 + w / 4], w & 3));
                    a[t] = (t < cur_nv && mk == cur_e + t && !(pl <= 0.f)) ? g : 0.f;
                } else {
                    a[t] = t < cur_nv ? f4_at(buf[t / 4], t & 3) : 0.f;
                }
            }
        };
        // one tile's MFMAs: k-slot t <-> pixel t of the lane's run
        auto tile_mfma = [&](int nt, const float (&a)[RL], const f4u (&win)[WL / 4]) {
#pragma unroll
            for (int t = 0; t < RL; ++t) {
                const int e = S == 2 ? 2 * t : t;  // window element of pixel t (even phase)
                const f4u& q = win[e / 4];
                float bv;
                if (S == 2) {
                    const float ev = (e & 3) == 0 ? q.x : q.z, od = (e & 3) == 0 ? q.y : q.w;
                    bv = par[nt] ? od : ev;
                } else {
                    bv = (e & 3) == 0 ? q.x : (e & 3) == 1 ? q.y : (e & 3) == 2 ? q.z : q.w;
                }
                bv = t < cur_nb ? bv : 0.f;  // what lies behind the run is not the reference's to read (may be Inf / NaN)
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bv, acc[nt], 0, 0, 0);
            }
        };
        f4u abuf[NA], wb[2][WL / 4];
        {
            unsigned a0;
            locate(w_lo, a0, cur_x, cur_nv, cur_nb, cur_e);
            load_a(abuf, a0);
#pragma unroll
            for (int j = 0; j < WL / 4; ++j) wb[0][j] = *(const f4u*)(p.x + (cur_x + (unsigned)xoff[0]) + 4 * j);
        }
        auto body = [&](auto PC, int ch_next) {
            constexpr int P = decltype(PC)::value;
            locate(ch_next, nxt_a, nxt_x, nxt_nv, nxt_nb, nxt_e);
            float a[RL];
            build_a(a, abuf);
            add_bias(a);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int bi = (P + nt) & 1;
                if (nt + 1 < NT) {
#pragma unroll
                    for (int j = 0; j < WL / 4; ++j)
                        wb[bi ^ 1][j] = *(const f4u*)(p.x + (cur_x + (unsigned)xoff[nt + 1 < NT ? nt + 1 : 0]) + 4 * j);
                } else {
                    load_a(abuf, nxt_a);
#pragma unroll
                    for (int j = 0; j < WL / 4; ++j) wb[bi ^ 1][j] = *(const f4u*)(p.x + (nxt_x + (unsigned)xoff[0]) + 4 * j);
                }
                RD_PIPE_FENCE(a[0]);
                if (nt < nt_live) tile_mfma(nt, a, wb[bi]);
            }
            cur_x = nxt_x;
            cur_nv = nxt_nv;
            cur_nb = nxt_nb;
            cur_e = nxt_e;
        };
        int ch = w_lo;
        if (NT & 1) {  // an odd tile count flips the buffer parity from chunk to chunk
            for (; ch + 1 < f_hi; ch += 2) {
                body(std::integral_constant<int, 0>(), ch + 1);
                body(std::integral_constant<int, 1>(), ch + 2 < f_hi ? ch + 2 : ch + 1);
            }
            if (ch < f_hi) body(std::integral_constant<int, 0>(), ch);
        } else {
            for (; ch < f_hi; ++ch) body(std::integral_constant<int, 0>(), ch + 1 < f_hi ? ch + 1 : ch);
        }
    }

    // ---- guarded path: the few chunks at the very end of the tensors (and tensors too small for the pipelined path)
    for (int ch = s_lo; ch < w_hi; ++ch) {
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
        if constexpr (POOLED) {
            const bool rowok = co < p.Co && (pr >> 1) < p.PHo;
            const int cc = co < p.Co ? co : 0;
            int inwin = 2 * p.PWo - q0;
            inwin = inwin < 0 ? 0 : inwin;
            const int nv = rowok ? (npix < inwin ? npix : inwin) : 0;
            const int nwin = (nv + 1) >> 1;  // windows the run's live pixels touch
            const size_t o = (((size_t)b * p.Co + cc) * p.PHo + (rowok ? (pr >> 1) : 0)) * p.PWo + (q0 >> 1);
            const int e0 = (cc * p.Ho + pr) * p.Wo + q0;
#pragma unroll
            for (int j = 0; j < RL / 8; ++j) {
                const f4u g = load4(p.dy + o + 4 * j, nwin - 4 * j), mk = load4((const float*)p.pmask + o + 4 * j, nwin - 4 * j),
                          pl = p.pooled ? load4(p.pooled + o + 4 * j, nwin - 4 * j) : f4u{1.f, 1.f, 1.f, 1.f};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int t = 8 * j + i, w = i >> 1;
                    const float gv = w == 0 ? g.x : w == 1 ? g.y : w == 2 ? g.z : g.w, pv = w == 0 ? pl.x : w == 1 ? pl.y : w == 2 ? pl.z : pl.w;
                    const int mv = __builtin_bit_cast(int, w == 0 ? mk.x : w == 1 ? mk.y : w == 2 ? mk.z : mk.w);
                    a[t] = (t < nv && mv == e0 + t && !(pv <= 0.f)) ? gv : 0.f;
                }
            }
        } else {
            const bool rowok = co < p.Co;
            const float* src = p.dy + (((size_t)b * p.Co + (rowok ? co : 0)) * p.Ho + pr) * p.Wo + q0;
            const int nv = rowok ? npix : 0;
#pragma unroll
            for (int j = 0; j < RL / 4; ++j) {
                const f4u v = load4(src + 4 * j, nv - 4 * j);
                a[4 * j] = v.x; a[4 * j + 1] = v.y; a[4 * j + 2] = v.z; a[4 * j + 3] = v.w;
            }
        }
        add_bias(a);
        const float* ximg = p.x + (size_t)b * img_x + (size_t)(S * pr) * p.W + S * q0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (nt >= nt_live) continue;
            // ---- B operand: the lane's window of its filter tap's input row
            float w[WL];
            const int col = S * q0 + shift[nt];
            int rem = p.W - col;  // floats left in the input row
            rem = rem < S * npix ? rem : S * npix;  // floats behind the run's last pixel are never multiplied by a live A
            rem = nvalid_col[nt] ? (rem < 0 ? 0 : rem) : 0;
            const float* src = ximg + xoff[nt];
#pragma unroll
            for (int j = 0; j < WL / 4; ++j) {
                const f4u v = load4(src + 4 * j, rem - 4 * j);
                w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
            }
#pragma unroll
            for (int t = 0; t < RL; ++t) {
                const float bv = S == 2 ? (par[nt] ? w[2 * t + 1] : w[2 * t]) : w[t];
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bv, acc[nt], 0, 0, 0);
            }
        }
    }

    // ---- sum the four waves in a fixed order, then one slab per workgroup
    const float bsum2 = bsum + __shfl_xor(bsum, 32, 64);  // the two k-groups of channel co
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
            if (kg == 0) red[m][NT * 32] = (w == 0) ? bsum2 : red[m][NT * 32] + bsum2;
        }
        __syncthreads();
    }
    if (p.dbg == 9 && threadIdx.x == 0 && (blockIdx.x | blockIdx.y | blockIdx.z) == 0)
        printf("wgrad_rd block 0: %lld shader cycles in %lld ticks of 10 ns -> %.0f MHz\n", clock64() - dbg_t0, wall_clock64() - dbg_w0,
               (double)(clock64() - dbg_t0) / ((double)(wall_clock64() - dbg_w0) / 100.0));
    float* slab = p.slabs + (size_t)blockIdx.x * p.Co * p.pitch;
    for (int i = threadIdx.x; i < 32 * NT * 32; i += 256) {
        const int row = i / (NT * 32), col = i - row * (NT * 32);
        const int c2 = blockIdx.z * 32 + row, n2 = nbase + col;
        if (c2 < p.Co && n2 < p.Ntot) slab[(size_t)c2 * p.pitch + n2] = red[row][col];
    }
    if (blockIdx.y == 0 && threadIdx.x < 32 && blockIdx.z * 32 + threadIdx.x < p.Co)
        slab[(size_t)(blockIdx.z * 32 + threadIdx.x) * p.pitch + p.Ntot] = red[threadIdx.x][NT * 32];
}

inline unsigned magic_of(int d) { return (unsigned)((1ull << 32) / (unsigned)d) + 1u; }

struct RdPlan {
    RdParams p;
    int nt, ngroups, mtiles, kblocks, rl;
};

bool make_rd_plan(const cnn_conv2d_desc* d, RdPlan* pl, bool pooled = false) {
    if (d->k != 3 || d->pad != 0 || (d->s != 1 && d->s != 2)) return false;
    RdParams& p = pl->p;
    p.B = d->B; p.Ci = d->Ci; p.H = d->H; p.W = d->W; p.Co = d->Co;
    p.Ho = cnn_conv2d_out_dim(d->H, 3, d->s, 0);
    p.Wo = cnn_conv2d_out_dim(d->W, 3, d->s, 0);
    if (p.Ho <= 0 || p.Wo <= 0) return false;
    p.Ntot = d->Ci * 9;
    p.pitch = p.Ntot + 1;
    p.PHo = p.Ho / 2; p.PWo = p.Wo / 2;
    p.pmask = nullptr; p.pooled = nullptr;
    if (pooled && (p.PHo < 1 || p.PWo < 1 || d->Ci * 9 > 32)) return false;  // (POOLED kernels exist for one column tile)
    pl->rl = p.Wo <= 8 ? 8 : 16;
    p.rpr = (p.Wo + pl->rl - 1) / pl->rl;
    const long long runs = (long long)p.B * p.Ho * p.rpr;
    const long long chunks = (runs + 1) / 2;
    p.runs_total = (int)runs;
    if (runs >= (1ll << 30) || (long long)p.B * p.Ci * p.H * p.W >= (1ll << 40)) return false;
    p.chunks_total = (int)chunks;
    // tiles per wave: the largest of 6..3 that divides the tile count (no dead tiles, equal groups), else 5
    const int tiles = (p.Ntot + 31) / 32;
    const int nt_max = (d->s == 2 && pl->rl == 16) ? 5 : 6;  // (64-register window pairs: 6 tiles would not fit 256 VGPRs)
    int nt = tiles <= nt_max ? tiles : 0;
    for (int c = nt_max; !nt && c >= 3; --c)
        if (tiles % c == 0) nt = c;
    if (!nt) nt = 5;
    if (const char* e = getenv("CNN_AMD_RD_NT")) {
        const int v = atoi(e);
        if (v >= 1 && v <= nt_max) nt = v;
    }
    pl->nt = nt;
    pl->ngroups = (tiles + nt - 1) / nt;
    pl->mtiles = (p.Co + 31) / 32;
    const int env = getenv("CNN_AMD_RD_BLOCKS") ? atoi(getenv("CNN_AMD_RD_BLOCKS")) : 0;
    long long want = (env > 0 ? env : 2 * kNumCU) / ((long long)pl->ngroups * pl->mtiles);
    if (want < 1) want = 1;
    if (want > chunks) want = chunks;
    p.chunks_per_block = (int)((chunks + want - 1) / want);
    pl->kblocks = (int)((chunks + p.chunks_per_block - 1) / p.chunks_per_block);
    p.m_rows = magic_of(p.Ho);
    p.m_rpr = magic_of(p.rpr);
    // runs whose (over-reading) windows stay inside the tensors: addresses grow with the run index, so scan from the end
    const long long x_total = (long long)p.B * p.Ci * p.H * p.W;
    const long long dy_total = pooled ? (long long)p.B * p.Co * p.PHo * p.PWo : (long long)p.B * p.Co * p.Ho * p.Wo;
    long long r_unsafe = 0;
    if (x_total < (1ll << 31) && dy_total < (1ll << 31)) {
        const long long xoff_bound = ((long long)(p.Ci - 1) * p.H + 2) * p.W + 2;
        r_unsafe = runs;
        while (r_unsafe > 0) {
            const long long r = r_unsafe - 1, rowi = r / p.rpr, seg = r % p.rpr, b = rowi / p.Ho, pr = rowi % p.Ho;
            const long long xb = b * p.Ci * p.H * p.W + d->s * pr * p.W + d->s * seg * pl->rl;
            const long long ab = pooled ? ((b * p.Co + p.Co - 1) * p.PHo + (pr >> 1)) * p.PWo + (seg * pl->rl >> 1)
                                        : ((b * p.Co + p.Co - 1) * p.Ho + pr) * p.Wo + seg * pl->rl;
            if (xb + xoff_bound + d->s * pl->rl <= x_total && ab + (pooled ? pl->rl / 2 : pl->rl) <= dy_total) break;
            --r_unsafe;
        }
    }
    p.dbg = getenv("CNN_AMD_RD_DBG") ? atoi(getenv("CNN_AMD_RD_DBG")) : 0;
    p.chunks_fast = getenv("CNN_AMD_RD_SLOW") ? 0 : (int)(r_unsafe / 2);  // (CNN_AMD_RD_SLOW: tests force the guarded path)
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
#define RD(S_, NT_, RL_) CNN_KLAUNCH(s, name, (wgrad_rd_kernel<S_, NT_, RL_, false><<<grid, 256, 0, s>>>(pl.p)), CONV_TAG(d))
#define RD_NT(S_, RL_)                                                     \
    switch (pl.nt) {                                                       \
        case 1: RD(S_, 1, RL_); break;                                     \
        case 2: RD(S_, 2, RL_); break;                                     \
        case 3: RD(S_, 3, RL_); break;                                     \
        case 4: RD(S_, 4, RL_); break;                                     \
        case 5: RD(S_, 5, RL_); break;                                     \
        default: RD(S_, (S_ == 2 && RL_ == 16 ? 5 : 6), RL_); break;       \
    }
    if (d->s == 2 && pl.rl == 16) { RD_NT(2, 16) }
    else if (d->s == 2) { RD_NT(2, 8) }
    else if (pl.rl == 16) { RD_NT(1, 16) }
    else { RD_NT(1, 8) }
#undef RD_NT
#undef RD
    return CNN_AMD_OK;
}

// the same from the pooled domain (first block: Ci*9 <= 32 columns); 0 slots = not covered
int wgrad_rd_pooled_slots(const cnn_conv2d_desc* d) {
    RdPlan pl;
    if (d->s != 2 || !make_rd_plan(d, &pl, true) || pl.nt != 1 || pl.ngroups != 1) return 0;
    return pl.kblocks;
}

int wgrad_rd_launch_pooled(const cnn_conv2d_desc* d, const float* x, const float* dpool, const int32_t* mask, const float* pooled,
                           float* slabs, hipStream_t s) {
    RdPlan pl;
    if (wgrad_rd_pooled_slots(d) == 0 || !make_rd_plan(d, &pl, true)) return fail(CNN_AMD_E_BADARG, "wgrad_rd+pool: geometry not covered");
    pl.p.x = x; pl.p.dy = dpool; pl.p.pmask = mask; pl.p.pooled = pooled; pl.p.slabs = slabs;
    const dim3 grid(pl.kblocks, pl.ngroups, pl.mtiles);
    char name[64];
    snprintf(name, sizeof(name), "wgrad_rd<%d,%d,%d>+pool", d->s, pl.nt, pl.rl);
    if (pl.rl == 16) CNN_KLAUNCH(s, name, (wgrad_rd_kernel<2, 1, 16, true><<<grid, 256, 0, s>>>(pl.p)), CONV_TAG(d));
    else CNN_KLAUNCH(s, name, (wgrad_rd_kernel<2, 1, 8, true><<<grid, 256, 0, s>>>(pl.p)), CONV_TAG(d));
    return CNN_AMD_OK;
}

}  // namespace cnn_amd
