// conv_wgrad_rd.hip -- "register-direct" Conv2D weight / bias gradient (cpu/src/conv2d.cpp:117-159) for 3x3 filters,
// stride 1 or 2, padding 0 (the reference) or 1 (the VGG / ResNet-shaped stacks): the same batched outer-product GEMM as
// conv_wgrad.hip
//     gw[co][n] = sum_{b,p,q} dy[b][co][p][q] * x[b][ci_n][s*p + kx_n - pad][s*q + ky_n - pad],   n = (ci,kx,ky),
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
// Padding (PAD = 1): the windows keep their addresses -- a tap row above / below the image is the last / first row of the
// neighbouring channel plane (mapped memory) -- and every (lane, tile, chunk) gets a live range [lo, hi) of k-slots: empty
// for a row outside the image, clipped where the window starts one column left of the image or runs over its right edge;
// one compare + select per MFMA step like the pad-0 kernels' `t < nb`.  Only output row 0 (and the first run of row 1) of
// image 0 could read in front of the allocation (x[0][0][-1][..], x[0][0][0][-1]): those chunks take the guarded path
// (RdParams::chunks_head), like the last rows of the tensor do for both paddings.
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
    int chunks_fast;  // chunks [chunks_head, chunks_fast) may over-read their windows without leaving x / dy (host-checked)
    int chunks_head;  // chunks [0, chunks_head) could read in FRONT of x (padding): guarded path
    // FLAT (stride 1, pad 1, Wo == W): runs are cut from the FLATTENED pixel index of an image instead of row by row -- both
    // operands stay contiguous across a row end (dy row p+1 follows row p, and so does the input window) -- so a 28- or 56-wide
    // row no longer wastes the tail of its last run (87.5 % -> 100 % live k-slots).  Then Ho = 1, Wo = H*W for the addressing
    // and W / H / m_w (magic of W) describe the real rows for the masks.
    int flat;
    unsigned m_w;
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

template <int S, int NT, int RL, bool POOLED, int PAD, bool FLAT = false>
__device__ __forceinline__ void wgrad_rd_body(const RdParams& p) {
    static_assert(!FLAT || (S == 1 && PAD == 1), "flattened runs: stride 1, pad 1 (runs of 16, or of 8 for rows of 9 .. 15 pixels: RL <= W)");
    static_assert(PAD == 0 || (PAD == 1 && !POOLED), "padding 0 or 1; the pooled-domain first block is unpadded");
    constexpr int WL = S * RL;  // floats of x a lane needs per chunk and tile
    __shared__ float red[32][NT * 32 + 1];  // (+1: bank padding; the column doubles as the bias-gradient slot)
    const int lane = threadIdx.x & 63, m = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int co = blockIdx.z * 32 + m;
    const int nbase = blockIdx.y * NT * 32;
    const int nt_live = (p.Ntot - nbase + 31) / 32;  // tiles of this group that hold at least one column (block-uniform)

    // per-lane column description for every N tile: offset of the filter tap inside an image, window shift, parity
    // column of pixel t in the input row: S*(q0+t) + cc, cc = ky - PAD; row: S*pr + dk, dk = kx - PAD.
    // (kx, ky) of all NT tiles live in ONE register, 4 bits per tile (the padded kernels sit at the 256-register limit of two
    // waves per SIMD: per-tile arrays for them cost 12 registers and the second wave)
    int xoff[NT], shift[NT];
    bool par[NT], nvalid_col[NT];
    unsigned taps = 0;
    auto dk_of = [&](int nt) { return (int)((taps >> (4 * nt)) & 3u) - PAD; };
    auto cc_of = [&](int nt) { return (int)((taps >> (4 * nt + 2)) & 3u) - PAD; };
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nbase + nt * 32 + m;
        const int nn = n < p.Ntot ? n : 0;
        const int ci = nn / 9, kx = (nn - ci * 9) / 3, ky = nn - ci * 9 - kx * 3;
        nvalid_col[nt] = n < p.Ntot;
        taps |= (unsigned)(kx | (ky << 2)) << (4 * nt);
        const int c = ky - PAD;
        // stride 2: the window starts at the even column 2*q0 + shift and pixel t is its element 2t (+1 when c is odd)
        shift[nt] = S == 2 ? (c < 0 ? -2 : (c == 2 ? 2 : 0)) : c;
        par[nt] = S == 2 && (c & 1);
        xoff[nt] = (ci * p.H + kx - PAD) * p.W + shift[nt];
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

    // ---- guarded path (defined below the pipelined one): chunks [lo, hi) of this wave
    auto guarded = [&](int g_lo, int g_hi) {
    for (int ch = g_lo; ch < g_hi; ++ch) {
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
            const int cc_ = co < p.Co ? co : 0;
            int inwin = 2 * p.PWo - q0;
            inwin = inwin < 0 ? 0 : inwin;
            const int nv = rowok ? (npix < inwin ? npix : inwin) : 0;
            const int nwin = (nv + 1) >> 1;  // windows the run's live pixels touch
            const size_t o = (((size_t)b * p.Co + cc_) * p.PHo + (rowok ? (pr >> 1) : 0)) * p.PWo + (q0 >> 1);
            const int e0 = (cc_ * p.Ho + pr) * p.Wo + q0;
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
            if constexpr (PAD > 0) {
                // element-wise: pixel t reads column S*(q0+t) + cc of row S*pr + dk, zero outside the image
                const int cc_nt = cc_of(nt);
                const bool rowv = nvalid_col[nt] && (unsigned)(S * pr + dk_of(nt)) < (unsigned)p.H;
                const float* src = ximg + (ptrdiff_t)(xoff[nt] - shift[nt]);  // (row, column S*q0) of the tap's input row
#pragma unroll
                for (int t = 0; t < RL; ++t) {
                    bool ok;
                    if constexpr (FLAT) {  // flattened runs: pixel t is (pt, qt) of the image
                        const int pt = fdiv(q0 + t, p.m_w, p.W), qt = q0 + t - pt * p.W;
                        ok = nvalid_col[nt] && t < npix && (unsigned)(pt + dk_of(nt)) < (unsigned)p.H && (unsigned)(qt + cc_nt) < (unsigned)p.W;
                    } else {
                        const int col = S * (q0 + t) + cc_nt;
                        ok = rowv && t < npix && (unsigned)col < (unsigned)p.W;
                    }
                    const float v = ok ? src[S * t + cc_nt] : 0.f;
                    w[S == 2 ? 2 * t + (par[nt] ? 1 : 0) : t] = v;
                    if (S == 2) w[2 * t + (par[nt] ? 0 : 1)] = 0.f;
                }
            } else {
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
            }
#pragma unroll
            for (int t = 0; t < RL; ++t) {
                const float bv = S == 2 ? (par[nt] ? w[2 * t + 1] : w[2 * t]) : w[t];
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bv, acc[nt], 0, 0, 0);
            }
        }
    }
    };

    // ---- head: the chunk(s) whose windows could start in front of x (PAD only)
    const int h_hi = w_hi < p.chunks_head ? w_hi : p.chunks_head;
    if (w_lo < h_hi) guarded(w_lo, h_hi);
    const int p_lo = w_lo > h_hi ? w_lo : h_hi;

    // ---- pipelined path
    const int f_hi = w_hi < p.chunks_fast ? w_hi : p.chunks_fast;
    int s_lo = p_lo;
    if (p_lo < f_hi) {
        s_lo = f_hi;
        const int co_c = co < p.Co ? co : 0;
        unsigned cur_x = 0, nxt_x = 0, nxt_a = 0;
        int cur_nv = 0, nxt_nv = 0, cur_nb = 0, nxt_nb = 0;  // live pixels of the lane's run: A side (0 for co >= Co) / B side
        int cur_e = 0, nxt_e = 0;  // POOLED: flat index (within the sample) of the first pixel of the lane's run, channel co
        unsigned cur_pq = 0, nxt_pq = 0;  // PAD: output row (<< 16) | first column of the lane's run
        auto locate = [&](int ch, unsigned& aoff, unsigned& xb, int& nv, int& nb, int& eidx, unsigned& pq_out) {
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
            pq_out = ((unsigned)pr << 16) | (unsigned)q0;
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
            // live k-slots of this lane's window: a range [lo, lo + span) -- or, with flattened runs, one bit per pixel of the run
            unsigned vm = 0;
            int lo = 0, span = cur_nb;
            if constexpr (PAD > 0) {
                const int dk = dk_of(nt), cc = cc_of(nt);
                if constexpr (FLAT) {
                    // flattened runs: the run starts at pixel (p0, q0) and may cross ONE row end (RL <= W)
                    const int f0 = (int)(cur_pq & 0xffffu);
                    const int p0 = fdiv(f0, p.m_w, p.W), q0 = f0 - p0 * p.W;
                    const int wrap = p.W - q0;  // first k-slot of the next row
                    const unsigned first = wrap >= RL ? (1u << RL) - 1u : (1u << wrap) - 1u;
                    const bool r0 = (unsigned)(p0 + dk) < (unsigned)p.H, r1 = (unsigned)(p0 + 1 + dk) < (unsigned)p.H;
                    const unsigned rows = (r0 ? first : 0u) | (r1 ? (((1u << RL) - 1u) & ~first) : 0u);
                    // the one pixel whose window column leaves the image: column 0 for cc = -1, column W-1 for cc = +1
                    unsigned bad = 0u;
                    if (cc < 0) bad = q0 == 0 ? 1u : (wrap < RL ? 1u << wrap : 0u);
                    else if (cc > 0) bad = wrap - 1 < RL ? 1u << (wrap - 1) : 0u;
                    vm = ((1u << cur_nb) - 1u) & rows & ~bad;
                } else {
                    // a tap row outside the image still reads MAPPED memory (the neighbouring channel / image; the first and last
                    // rows of the whole tensor are the guarded path's) and is masked as a whole
                    const int c0 = S * (int)(cur_pq & 0xffffu) + cc;         // column of pixel 0
                    lo = c0 < 0 ? 1 : 0;                                     // (c0 >= -PAD = -1)
                    int hi = S == 1 ? p.W - c0 : (p.W - c0 + 1) >> 1;        // pixels whose column is < W
                    hi = hi < cur_nb ? hi : cur_nb;
                    const bool rowv = (unsigned)(S * (int)(cur_pq >> 16) + dk) < (unsigned)p.H;
                    span = rowv ? hi - lo : 0;
                }
            }
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
                // what lies behind the run (or outside the image) is not the reference's to read (may be Inf / NaN)
                if constexpr (FLAT) bv = (vm & (1u << t)) ? bv : 0.f;
                else if constexpr (PAD > 0) bv = (unsigned)(t - lo) < (unsigned)span ? bv : 0.f;
                else bv = t < cur_nb ? bv : 0.f;
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bv, acc[nt], 0, 0, 0);
            }
        };
        f4u abuf[NA], wb[2][WL / 4];
        {
            unsigned a0;
            locate(p_lo, a0, cur_x, cur_nv, cur_nb, cur_e, cur_pq);
            load_a(abuf, a0);
#pragma unroll
            for (int j = 0; j < WL / 4; ++j) wb[0][j] = *(const f4u*)(p.x + (cur_x + (unsigned)xoff[0]) + 4 * j);
        }
        auto body = [&](auto PC, int ch_next) {
            constexpr int P = decltype(PC)::value;
            locate(ch_next, nxt_a, nxt_x, nxt_nv, nxt_nb, nxt_e, nxt_pq);
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
            cur_pq = nxt_pq;
        };
        int ch = p_lo;
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

    // ---- tail: the few chunks at the very end of the tensors (and tensors too small for the pipelined path)
    guarded(s_lo, w_hi);

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

template <int S, int NT, int RL, bool POOLED>
__global__ __launch_bounds__(256) void wgrad_rd_kernel(const RdParams p) {
    wgrad_rd_body<S, NT, RL, POOLED, 0>(p);
}
// The padded variants carry a few more live values per tile; without a register budget hipcc takes 290 registers for six
// tiles and the kernel drops to ONE wave per SIMD (62 instead of 100+ TFLOP/s): two waves per SIMD = 256 registers, enforced.
template <int S, int NT, int RL, bool FLAT>
__global__ __launch_bounds__(256, 2) void wgrad_rd_kernel_p1(const RdParams p) {
    wgrad_rd_body<S, NT, RL, false, 1, FLAT>(p);
}

inline unsigned magic_of(int d) { return (unsigned)((1ull << 32) / (unsigned)d) + 1u; }

struct RdPlan {
    RdParams p;
    int nt, ngroups, mtiles, kblocks, rl;
};

bool make_rd_plan(const cnn_conv2d_desc* d, RdPlan* pl, bool pooled = false) {
    if (d->k != 3 || d->pad < 0 || d->pad > 1 || (d->s != 1 && d->s != 2)) return false;
    if (pooled && d->pad != 0) return false;
    RdParams& p = pl->p;
    p.B = d->B; p.Ci = d->Ci; p.H = d->H; p.W = d->W; p.Co = d->Co;
    p.Ho = cnn_conv2d_out_dim(d->H, 3, d->s, d->pad);
    p.Wo = cnn_conv2d_out_dim(d->W, 3, d->s, d->pad);
    if (p.Ho <= 0 || p.Wo <= 0) return false;
    // flattened runs (see RdParams::flat): worth it when the rows do not divide into whole runs of 16
    p.flat = 0;
    p.m_w = magic_of(d->W);
    // (measurement switch RD_RL8: rows that are whole runs of 8 but not of 16 -- 56, 24, 40 -- as runs of 8 instead of flattened runs of 16)
    const bool rl8_rows = !pooled && d->s == 1 && d->W > 8 && d->W % 8 == 0 && d->W % 16 != 0 && CNN_OPT_INT("RD_RL8", 0) != 0;
    // (round 4, measurement switch RD_FLAT8=1) rows of 9 .. 15 pixels (the 14x14 layers) as flattened runs of 8 -- 25 runs of 8 cover an image's 196
    // pixels (98 % live k-slots) where one run of 16 per row of 14 leaves 12.5 % dead.  Measured: 512 -> 512 14x14 at batch 128 1 420 -> 1 372 us,
    // 256 -> 256 14x14 at batch 64 209 -> 221 us (twice the chunks, each with half the MFMAs behind its loads): off by default
    const bool flat8 = !pooled && d->s == 1 && d->pad == 1 && d->W > 8 && d->W < 16 && d->W % 8 != 0 && CNN_OPT_INT("RD_FLAT8", 0) != 0 && !CNN_OPT_SET("RD_NOFLAT");
    if (flat8 || (!rl8_rows && !pooled && d->s == 1 && d->pad == 1 && d->W >= 16 && d->W % 16 != 0 && (long long)d->H * d->W < 65536 && !CNN_OPT_SET("RD_NOFLAT"))) {
        p.flat = 1;
        p.Wo = d->H * d->W;  // one "row" per image
        p.Ho = 1;
    }
    p.Ntot = d->Ci * 9;
    p.pitch = p.Ntot + 1;
    p.PHo = p.Ho / 2; p.PWo = p.Wo / 2;
    p.pmask = nullptr; p.pooled = nullptr;
    if (pooled && (p.PHo < 1 || p.PWo < 1 || d->Ci * 9 > 32)) return false;  // (POOLED kernels exist for one column tile)
    pl->rl = (p.Wo <= 8 || rl8_rows || flat8) ? 8 : 16;
    p.rpr = (p.Wo + pl->rl - 1) / pl->rl;
    const long long runs = (long long)p.B * p.Ho * p.rpr;
    const long long chunks = (runs + 1) / 2;
    p.runs_total = (int)runs;
    if (runs >= (1ll << 30) || (long long)p.B * p.Ci * p.H * p.W >= (1ll << 40)) return false;
    p.chunks_total = (int)chunks;
    // tiles per wave: the largest of 6..3 that divides the tile count (no dead tiles, equal groups), else 5
    const int tiles = (p.Ntot + 31) / 32;
    // (stride 2, runs of 16: 64-register window pairs -- 6 tiles do not fit 256 VGPRs, and with padding not 5 either)
    const int nt_max = (d->s == 2 && pl->rl == 16) ? (d->pad ? 4 : 5) : 6;
    int nt = tiles <= nt_max ? tiles : 0;
    for (int c = nt_max; !nt && c >= 3; --c)
        if (tiles % c == 0) nt = c;
    if (!nt) nt = 5;
    if (const OptVal e = CNN_OPT_VAL("RD_NT")) {
        const int v = atoi(e);
        if (v >= 1 && v <= nt_max) nt = v;
    }
    pl->nt = nt;
    pl->ngroups = (tiles + nt - 1) / nt;
    pl->mtiles = (p.Co + 31) / 32;
    const int env = CNN_OPT_INT("RD_BLOCKS", 0);
    const long long colblocks = (long long)pl->ngroups * pl->mtiles, slots = env > 0 ? env : 2 * num_cus();
    long long want = slots / colblocks;
    if (want < 1) want = 1;
    // (round 4) long pixel ranges: one round of ~500 workgroups that each run for milliseconds ends with the slowest of them -- three rounds of
    // shorter ones balance themselves.  256 -> 256 56x56 at batch 128: 5 014 -> 4 714 us, 128 -> 256 56x56: 2 510 -> 2 406 us
    // (tools/probes/wgrad_blocks.py, SWEEP=xcd); the batch-64 layers of the ResNet-shaped stack (< 100 chunks per workgroup) keep one round
    // (not where that means > 64 pixel ranges -- 64 -> 128 112x112: 128 slabs, a two-stage reduction, 2 376 against 2 350 us)
    if (env <= 0 && chunks / want >= 1024 && 3 * slots / colblocks <= 64 && !CNN_OPT_SET("RD_ONE_ROUND")) want = 3 * slots / colblocks;
    if (env <= 0 && colblocks * want * 8 < slots * 7) {
        // wide layers (Ci*9/32/NT column groups x Co/32 row tiles is already comparable to the chip): pick the split whose
        // workgroup count fills whole rounds of the resident slots best (384 column blocks alone would leave a quarter idle)
        double best = 0;
        for (long long k = 1; k <= 8; ++k) {
            const long long blocks = colblocks * k, rounds = (blocks + slots - 1) / slots;
            const double fill = (double)blocks / (double)(rounds * slots);
            // (equal fill: the finer split as long as a workgroup keeps >= 100 chunks -- 512 -> 512 14x14 at batch 128: 1 451 -> 1 387 us)
            if (fill > best + 1e-9 || (fill > best - 1e-9 && chunks / k >= 100 && !CNN_OPT_SET("RD_ONE_ROUND"))) { best = fill; want = k; }
        }
    }
    if (want > chunks) want = chunks;
    p.chunks_per_block = (int)((chunks + want - 1) / want);
    pl->kblocks = (int)((chunks + p.chunks_per_block - 1) / p.chunks_per_block);
    // (measured and dropped: a split with kblocks % 8 == 0 puts the colblocks workgroups of one pixel range -- same x / dy data,
    // linear ids kblocks apart -- on ONE XCD: conv_layer_4 then fetches 17 MB instead of 84 MB, and the step gets 4 % SLOWER: the
    // operands fit the Infinity Cache, and eight L2s pulling them in parallel beat one L2 serving twelve workgroups)
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
    p.dbg = CNN_MEASURE_INT("RD_DBG", 0);
    p.chunks_fast = CNN_OPT_SET("RD_SLOW") ? 0 : (int)(r_unsafe / 2);  // (CNN_AMD_RD_SLOW: tests force the guarded path)
    // padded layers: the first run of output rows 0 and 1 of image 0 reads input row 0 from column -1 (or -2): x[-1] lies in
    // front of the allocation, so the chunks up to run `rpr` (first run of row 1) take the guarded path
    p.chunks_head = d->pad > 0 ? (p.flat ? ((d->W + 1) / pl->rl + 2) / 2 + 1 : p.rpr / 2 + 1) : 0;
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
    snprintf(name, sizeof(name), d->pad ? (pl.p.flat ? "wgrad_rd<%d,%d,%d,p1,flat>" : "wgrad_rd<%d,%d,%d,p1>") : "wgrad_rd<%d,%d,%d>", d->s, pl.nt, pl.rl);
#define RD(S_, NT_, RL_)                                                                                                   \
    do {                                                                                                                   \
        if (d->pad == 0) CNN_KLAUNCH(s, name, (wgrad_rd_kernel<S_, NT_, RL_, false><<<grid, 256, 0, s>>>(pl.p)), CONV_TAG(d)); \
        else if (S_ == 1 && pl.p.flat)                                                                                     \
            CNN_KLAUNCH(s, name, (wgrad_rd_kernel_p1<S_, NT_, RL_, (S_ == 1)><<<grid, 256, 0, s>>>(pl.p)), CONV_TAG(d)); \
        else CNN_KLAUNCH(s, name, (wgrad_rd_kernel_p1<S_, NT_, RL_, false><<<grid, 256, 0, s>>>(pl.p)), CONV_TAG(d));        \
    } while (0)
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
