// conv_wgrad_win.hip -- weight / bias gradient (cpu/src/conv2d.cpp:117-159) of the thin first layer
// Conv2D(3 -> 16, 3x3, stride 2, pad 0) (alexnet.cpp:12), from the delta of its output (POOLED = 0) or straight from the pooled
// domain of the Conv2D -> ReLU -> MaxPool2D(2,2) block (POOLED = 1 / 2: dy = ReLU'(MaxPool'(dpool)) rebuilt on the fly, see
// cnn_conv2d_backward_weight_pooled2 in include/cnn_amd.h).  HBM-bound: x (B*3*H*W) + dy (or dpool + mask [+ pooled]) are read
// once, 16 x 28 sums come out.  It replaces conv_direct.hip's packed VALU kernel, which was bound by LOAD INSTRUCTIONS (27
// row segments of 256 B per 64 pixels) at 2.0 - 2.3 TB/s.
//
//   gw[co][(ci,kx,ky)] = sum_{b,p,q} dy[b][co][p][q] * x[b][ci][2p+kx][2q+ky]         gb[co] = sum dy[b][co][p][q]
//
// as a GEMM on v_mfma_f32_16x16x4_f32: M = 16 = Co exactly, N = the 27 (ci,kx,ky) columns in two 16-wide tiles, K = pixels.
// The unit of work is a STRIP: one row of 2x2 pixel windows (= pooling windows) of one image, up to 32 windows wide: conv
// rows 2wr, 2wr+1 <- input rows 4wr .. 4wr+4.  A wave owns a contiguous range of strips (vertically neighbouring strips
// share an input row: L2) and, per strip,
//   * moves the 3 x 5 input row segments (<= 132 floats each) and the strip's delta operands HBM -> LDS with
//     global_load_lds (16-byte DMA for x: 8 instructions per strip instead of 90+ load instructions), into the second of two
//     private buffers while it computes on the first -- no VGPR round trip, no barrier, no other wave involved;
//   * runs groups of 4 windows: k-slot k of an MFMA step is window 4g+k, the four steps of a group are the window's pixels
//     (pr,pc).  A operand (lane = co, k): the pixel's delta; pooled domain: (mask == flat index of the pixel) ? dpool : 0 from
//     ONE dpool / mask value per window, staged window-major ([window][co]: conflict-free).  B operand (lane = column, k):
//     x[ci][4wr + 2pr + kx][4(4g+k) + 2pc + ky], read as ds_read2_b32 (pc = 0 | 1) at a per-lane base + an immediate.
// Four accumulators (tile x pr) keep dependent MFMAs two issues apart.  The waves of a workgroup are summed in a fixed order
// through LDS into one 16 x 28 slab per workgroup ([27 weight sums | bias sum]); reduce_slabs() (conv_wgrad.hip) adds the slabs
// and divides.  Plain and pooled variants visit the same strips in the same order with the same operand values, so the fused
// and unfused paths stay bit-identical (tests/test_gpu_parity.py).
#include <cstdlib>
#include <type_traits>

#include "common.h"

using namespace cnn_amd;

#ifndef CNN_WIN_EXPERIMENT
#define CNN_WIN_EXPERIMENT 0
#endif
namespace {
#if CNN_WIN_EXPERIMENT == 6  // (an A/B build of this file only: per-phase cycle counters; never in a shipped library)
__device__ unsigned long long g_prof[256 * 8][8];
#endif
constexpr int kExp = CNN_WIN_EXPERIMENT;  // timing experiments only (3: no LDS operand reads, 4: no MFMAs); 0 in the product

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CI = 3, CO = 16, XR = 5;  // input rows per window row: 4wr .. 4wr+4
constexpr int kSW = 16;                 // windows per strip (at most): four groups of four
constexpr int kGroups = kSW / 4;
constexpr int XP = 4 * kSW + 4;         // floats per staged input row segment: columns 4*wc_lo .. 4*wc_lo + 4*SW (+ pad to 16 B)
constexpr int XROWS = CI * XR;          // 15 staged rows ...
constexpr int XBUF = (XROWS + 1) * XP;  // ... + one row of 1.0f: the "column" whose sums are the bias gradient (never overwritten)
constexpr int kWaves = 8;               // two waves per SIMD: one wave's LDS / VALU latency hides behind the other's MFMAs
constexpr int kProd = 4;                // SPEC: one extra DMA-only wave per SIMD (see the kernel)
// delta operands of a strip: plain 2 rows x 2*SW columns x 16 channels; pooled domain 2 (dpool, mask) or 3 (+ pooled) tensors
// x SW windows x 16 channels
__host__ __device__ constexpr int abuf_floats(int pooled) {
    return pooled == 0 ? 2 * 2 * kSW * CO : (pooled == 3 ? kSW * CO + 64 : (pooled == 1 ? 3 : 2) * kSW * CO);  // (3: dpool + 64 dwords of mask bytes)
}
__host__ __device__ constexpr int buf_floats(int pooled) { return XBUF + abuf_floats(pooled); }
// strips in flight per wave: the train step's variant (POOLED = 2, 6.4 KB per strip) affords three buffers in 160 KB of LDS
// (two strips on their way while one is consumed), the others two
__host__ __device__ constexpr int num_bufs(int pooled) { return pooled >= 2 ? 3 : 2; }

struct WinParams {
    const float* x;
    const float* dy;       // POOLED: dpool
    const int* pmask;
    const float* pooled;
    float* slabs;          // [gridDim.x][16][28]
    int B, H, W, Ho, Wo;
    int PHo, PWo;          // pooled-domain size (Ho/2, Wo/2)
    int pitch8;            // POOLED == 3: bytes per row of the packed mask (pool_mask_pitch)
    int WR, WC;            // window grid: ceil(Ho/2) x ceil(Wo/2)
    int nseg, SW;          // column segments per window row, windows per segment (multiple of 4, <= kSW)
    int strips_total, strips_per_wave;
    int dbg;  // CNN_AMD_WIN_DBG (tuning): 1 = no MFMA groups, 2 = no staging
    int lockstep;  // the waves of a workgroup start every strip together (only when every wave has the same number of strips)
    int spec;      // host: launch the wave-specialised instance (same precondition)
    int spec_slack;  // (tuning) strips a producer may run ahead of the OTHER SIMDs' consumers beyond the buffer depth
};

// POOLED: 0 dy | 1 pooled domain (dpool, mask, pooled) | 2 pooled domain with dpool already ReLU-masked (pooled not read)
//         | 3 = 2 with the packed one-byte mask (include/cnn_amd.h, CNN_CONV2D_POOL_MASK_PACKED): the four windows of a lane's group
//           are ONE dword, moved by a 4-byte DMA into dword 16 * group + co behind the strip's dpool values
// DMA16: x rows are 16-byte aligned (W % 4 == 0, base aligned): 16-byte DMA; otherwise 4 bytes per lane
// SPEC ("wave specialisation"): a wave that is blocked while the memory pipeline takes its DMA burst cannot issue MFMAs, and with
// every wave doing both jobs the two phases barely overlapped (staging-only 52 us + MFMA-only 48 us ~ the 70 us measured).  With SPEC
// the eight MFMA waves never issue a DMA: four extra waves (one per SIMD, each feeding the two MFMA waves of its SIMD) run stage()
// for them and hand strips over through LDS flags -- ready[c][buffer] = strip index once its DMA has landed (the producer's counted
// s_waitcnt), done[c] = strips consumed (buffer free).  A producer never runs more than NBUF strips ahead of the SLOWEST consumer of
// the workgroup, which also keeps the eight runs close enough for the shared delta lines to stay in L2 (the job of the lock-step
// barrier in the non-SPEC kernel).
template <int POOLED, bool DMA16, bool SPEC>
__global__ __launch_bounds__((kWaves + (SPEC ? kProd : 0)) * 64) void conv_wgrad_win_kernel(const WinParams p) {
    constexpr int BUF = buf_floats(POOLED), NBUF = num_bufs(POOLED);
    constexpr int NT = POOLED == 1 ? 3 : 2;  // pooled-domain tensors
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = lane & 15, k = lane >> 4;  // MFMA lane coordinates: (row / column index, k-slot)
    float* const wbuf = lds + (wave < kWaves ? wave : 0) * NBUF * BUF;
    if (wave < kWaves) {
#pragma unroll
        for (int nb = 0; nb < NBUF; ++nb)
            for (int i = lane; i < XP; i += 64) wbuf[nb * BUF + XROWS * XP + i] = 1.f;  // the ones row of every buffer
    }

    // ---- per-lane constants of the B operand: column (ci,kx,ky) of tile t.  Column 27 reads the ones row (its sums are the
    // bias gradient: D[co][27] = sum of the deltas); columns 28..31 repeat it (discarded).
    int bbase[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int col = n0 + 16 * t;
        const int cc = col < 27 ? col : 26;
        const int ci = cc / 9, kx = (cc - ci * 9) / 3, ky = cc - ci * 9 - kx * 3;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) bbase[t][pr] = col < 27 ? (ci * XR + 2 * pr + kx) * XP + 4 * k + ky : XROWS * XP + 4 * k;
    }
    const int co = n0;  // A operand: this lane's output channel

    f32x4 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) acc[t][pr] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int gw = blockIdx.x * kWaves + (wave < kWaves ? wave : 0);
    const int s_lo = gw * p.strips_per_wave;
    const int s_hi = s_lo + p.strips_per_wave < p.strips_total ? s_lo + p.strips_per_wave : p.strips_total;

    // strip s -> (image b, column segment, window row wr): consecutive strips walk DOWN one column segment of one image.
    // A wave decodes its first strip once and then ADVANCES two cursors (the strip being staged, the strip being computed):
    // re-deriving everything from s cost ~300 scalar / vector instructions per strip -- more issue time than the strip's 32 MFMAs.
    struct Cur {
        int s, b, seg, wr;
    };
    auto decode = [&](int s) {
        Cur c;
        c.s = s;
        c.b = s / (p.nseg * p.WR);
        const int r = s - c.b * (p.nseg * p.WR);
        c.seg = r / p.WR;
        c.wr = r - c.seg * p.WR;
        return c;
    };
    auto advance = [&](Cur& c) {
        ++c.s;
        if (++c.wr == p.WR) {
            c.wr = 0;
            if (++c.seg == p.nseg) {
                c.seg = 0;
                ++c.b;
            }
        }
    };
    auto seg_geom = [&](int seg, int& wc_lo, int& sw) {
        wc_lo = seg * p.SW;
        sw = p.WC - wc_lo < p.SW ? p.WC - wc_lo : p.SW;
    };

    // per-lane DMA source offsets (floats) of the current staging COLUMN (image-independent, window-row-independent): x relative
    // to the strip's first element x[b][0][4wr][4wc_lo], the pooled-domain operands relative to T[b][0][wr][wc_lo].  Columns clipped
    // by the image's right edge are clamped here; strips clipped by its bottom edge take the slow path below.
    constexpr int CHX = DMA16 ? 4 : 1;
    constexpr int PER_ROW = XP / CHX;
    constexpr int TOTALX = XROWS * PER_ROW;
    constexpr int NX = (TOTALX + 63) / 64;
    struct ColState {
        unsigned gx[NX];
        unsigned ga;
        unsigned gm;  // POOLED == 3: byte offset of this lane's mask dword
        int col_seg;
    };
    // per DMA instruction i: this lane's chunk is in staged row r5 == 0 / exists at all (lane constants)
    unsigned row0_bits = 0, live_bits = 0;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int f = 64 * i + lane;
        const int row = f / PER_ROW;
        if (f < TOTALX) live_bits |= 1u << i;
        if (row % XR == 0) row0_bits |= 1u << i;
    }
    auto setup_column = [&](int seg, ColState& cs) {
        int wc_lo, sw;
        seg_geom(seg, wc_lo, sw);
        const int ncols = (p.W - 4 * wc_lo) < (4 * sw + 4) ? (p.W - 4 * wc_lo) : (4 * sw + 4);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int f = 64 * i + lane;
            const int row = f / PER_ROW, c = (f - row * PER_ROW) * CHX;
            const int ci = row / XR, r5 = row - ci * XR;
            cs.gx[i] = (unsigned)((ci * p.H + r5) * p.W + (c < ncols ? c : 0));
        }
        cs.ga = 0;
        cs.gm = 0;
        if constexpr (POOLED) {
            const int nv = p.PWo - wc_lo < sw ? p.PWo - wc_lo : sw;
            const int g = lane >> 4;
            cs.ga = (unsigned)((co * p.PHo) * p.PWo + (4 * g < nv ? 4 * g : 0));
            if constexpr (POOLED == 3) cs.gm = (unsigned)((co * p.PHo) * p.pitch8 + (4 * g < nv ? 4 * g : 0));
        }
        cs.col_seg = seg;
    };

    // ---- HBM -> LDS for one strip (asynchronous: completion = vmcnt).  Every instruction is issued by every strip (the wait
    // below counts instructions): a lane whose element lies outside the image / the row re-reads a valid element of the same
    // row instead, and the compute path ignores that slot (EDGE).  Returns true for the ONE strip kind that stages its delta
    // operands with 4-byte instead of 16-byte DMA (see below).
    auto stage = [&](const Cur& cu, float* buf, ColState& cs, int first_strip) -> bool {
        const int s = cu.s, b = cu.b, wr = cu.wr;
        int wc_lo, sw;
        seg_geom(cu.seg, wc_lo, sw);
        const bool chained = s > first_strip && wr > 0;  // the consumer took strip s-1 = (b, same segment, wr-1) just before
        if (cs.col_seg != cu.seg) setup_column(cu.seg, cs);
        const int nrows = (p.H - 4 * wr) < XR ? (p.H - 4 * wr) : XR;
        // x: rows 4wr .. 4wr+4 of the three channels, columns 4*wc_lo .. 4*wc_lo + 4*sw (inclusive), clipped to the image
        if (nrows == XR) {
            const float* src = p.x + ((size_t)b * CI * p.H + 4 * wr) * p.W + 4 * wc_lo;  // wave-uniform
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                if ((live_bits >> i & 1u) && !(chained && (row0_bits >> i & 1u))) {
                    if constexpr (DMA16) __builtin_amdgcn_global_load_lds((gbl_void_ptr)(src + cs.gx[i]), (lds_void_ptr)(buf + 64 * i * CHX), 16, 0, 0);
                    else __builtin_amdgcn_global_load_lds((gbl_void_ptr)(src + cs.gx[i]), (lds_void_ptr)(buf + 64 * i * CHX), 4, 0, 0);
                }
            }
        } else {
            const int ncols = (p.W - 4 * wc_lo) < (4 * sw + 4) ? (p.W - 4 * wc_lo) : (4 * sw + 4);
            const float* src = p.x + ((size_t)b * CI * p.H + 4 * wr) * p.W + 4 * wc_lo;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                const int f = 64 * i + lane;
                const int row = f / PER_ROW, c = (f - row * PER_ROW) * CHX;  // staged row (ci*5 + r5), first column of the chunk
                const int ci = row / XR, r5 = row - ci * XR;
                const int rr = r5 < nrows ? r5 : 0, cq = c < ncols ? c : 0;  // (outside the image: a valid element instead)
                // (lanes behind the last row would overwrite the ones row; row 0 of a strip that continues the previous one
                // -- same image, same column segment, next window row -- is that strip's row 4: handed over LDS -> LDS)
                if (f < TOTALX && !(chained && r5 == 0)) {
                    if constexpr (DMA16)
                        __builtin_amdgcn_global_load_lds((gbl_void_ptr)(src + ((size_t)ci * p.H + rr) * p.W + cq), (lds_void_ptr)(buf + 64 * i * CHX), 16, 0, 0);
                    else
                        __builtin_amdgcn_global_load_lds((gbl_void_ptr)(src + ((size_t)ci * p.H + rr) * p.W + cq), (lds_void_ptr)(buf + 64 * i * CHX), 4, 0, 0);
                }
            }
        }
        float* const abuf = buf + XBUF;
        if constexpr (POOLED) {
            // one dpool / mask (/ pooled) value per window and channel; per tensor: float index (group*16 + co)*4 + window % 4.
            // Fast path: lane (co = lane & 15, g = lane >> 4) moves the 4 windows of its group with ONE 16-byte DMA per tensor
            // (rows of the pooled domain follow each other in memory, so a chunk that runs over its row end reads the next row's
            // first elements: valid memory, ignored slots) -- except in the very last row of the tensors, where it would leave the
            // allocation: that strip moves single windows (lane = 4*co + window % 4, one instruction per group).
            const bool rowok = wr < p.PHo;
            const int nv = rowok ? (p.PWo - wc_lo < sw ? p.PWo - wc_lo : sw) : 0;
            const bool last_row = b == p.B - 1 && wr >= p.PHo - 1 && wc_lo + kSW > p.PWo;
            if (!last_row && rowok) {
                const size_t abase = (((size_t)b * CO) * p.PHo + wr) * p.PWo + wc_lo;  // wave-uniform; ga: this lane's (co, group)
                __builtin_amdgcn_global_load_lds((gbl_void_ptr)(p.dy + abase + cs.ga), (lds_void_ptr)(abuf), 16, 0, 0);
                if constexpr (POOLED == 3) {
                    const size_t mbase = (((size_t)b * CO) * p.PHo + wr) * p.pitch8 + wc_lo;  // bytes; rows and segments start 4-byte aligned
                    __builtin_amdgcn_global_load_lds((gbl_void_ptr)((const char*)p.pmask + mbase + cs.gm), (lds_void_ptr)(abuf + kSW * CO), 4, 0, 0);
                    return false;
                }
                __builtin_amdgcn_global_load_lds((gbl_void_ptr)((const float*)p.pmask + abase + cs.ga), (lds_void_ptr)(abuf + kSW * CO), 16, 0, 0);
                if constexpr (POOLED == 1) __builtin_amdgcn_global_load_lds((gbl_void_ptr)(p.pooled + abase + cs.ga), (lds_void_ptr)(abuf + 2 * kSW * CO), 16, 0, 0);
                return false;
            }
            if (!last_row) {  // a window row outside the pooled domain: any valid row will do (the compute path ignores it)
                const int g = lane >> 4;
                const size_t off = (((size_t)b * CO + co) * p.PHo) * p.PWo + wc_lo + (4 * g < nv ? 4 * g : 0);
                __builtin_amdgcn_global_load_lds((gbl_void_ptr)(p.dy + off), (lds_void_ptr)(abuf), 16, 0, 0);
                if constexpr (POOLED == 3) {
                    const size_t off8 = (((size_t)b * CO + co) * p.PHo) * p.pitch8 + wc_lo + (4 * g < nv ? 4 * g : 0);
                    __builtin_amdgcn_global_load_lds((gbl_void_ptr)((const char*)p.pmask + off8), (lds_void_ptr)(abuf + kSW * CO), 4, 0, 0);
                    return false;
                }
                __builtin_amdgcn_global_load_lds((gbl_void_ptr)((const float*)p.pmask + off), (lds_void_ptr)(abuf + kSW * CO), 16, 0, 0);
                if constexpr (POOLED == 1) __builtin_amdgcn_global_load_lds((gbl_void_ptr)(p.pooled + off), (lds_void_ptr)(abuf + 2 * kSW * CO), 16, 0, 0);
                return false;
            }
            const int c4 = lane >> 2, e = lane & 3;
            const size_t rowbase = (((size_t)b * CO + c4) * p.PHo + (rowok ? wr : 0)) * p.PWo + wc_lo;
            if constexpr (POOLED == 3) {  // (the packed mask's allocation carries 64 bytes of slack: its dword never leaves it)
                const int g4 = lane >> 4;
                const size_t off8 = (((size_t)b * CO + co) * p.PHo + (rowok ? wr : 0)) * p.pitch8 + wc_lo + (4 * g4 < nv ? 4 * g4 : 0);
                __builtin_amdgcn_global_load_lds((gbl_void_ptr)((const char*)p.pmask + off8), (lds_void_ptr)(abuf + kSW * CO), 4, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < kGroups; ++g) {
                const size_t off = rowbase + (4 * g + e < nv ? 4 * g + e : 0);
                __builtin_amdgcn_global_load_lds((gbl_void_ptr)(p.dy + off), (lds_void_ptr)(abuf + 64 * g), 4, 0, 0);
                if constexpr (POOLED != 3)
                    __builtin_amdgcn_global_load_lds((gbl_void_ptr)((const float*)p.pmask + off), (lds_void_ptr)(abuf + kSW * CO + 64 * g), 4, 0, 0);
                if constexpr (POOLED == 1) __builtin_amdgcn_global_load_lds((gbl_void_ptr)(p.pooled + off), (lds_void_ptr)(abuf + 2 * kSW * CO + 64 * g), 4, 0, 0);
            }
            return true;
        } else {
            // dy rows 2wr, 2wr+1, columns 2*wc_lo .. 2*wc_lo + 2*sw - 1, column-major per row: float index (pr*2*SW + column) * 16 + co
            const int j = lane >> 4;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int prow = 2 * wr + pr;
                const bool rowok = prow < p.Ho;
                const size_t rowbase = (((size_t)b * CO + co) * p.Ho + (rowok ? prow : 0)) * p.Wo + 2 * wc_lo;
                const int nv = rowok ? (p.Wo - 2 * wc_lo < 2 * sw ? p.Wo - 2 * wc_lo : 2 * sw) : 0;
#pragma unroll
                for (int i = 0; i < 2 * kSW / 4; ++i) {
                    const int c = 4 * i + j;
                    __builtin_amdgcn_global_load_lds((gbl_void_ptr)(p.dy + rowbase + (c < nv ? c : 0)), (lds_void_ptr)(abuf + (pr * 2 * kSW + 4 * i) * CO), 4, 0, 0);
                }
            }
            return false;
        }
    };
    // DMA instructions per strip: what `s_waitcnt vmcnt(N)` has to leave in flight when the NEXT strip is already on its way
    constexpr int N_FAST = NX + (POOLED ? NT : 2 * (2 * kSW / 4)), N_SLOW = NX + (POOLED == 3 ? kGroups + 1 : NT * kGroups);

    // ---- the strip's MFMA groups.  EDGE: the strip touches the last window row / column of the layer, where a window's pixels
    // may lie outside the output (delta 0, and their x values are not the reference's to read: both operands are forced to 0);
    // everywhere else every pixel of every window is live and the selects are compiled out.
    struct Ops {
        float a0, a1, a2;   // POOLED: dpool, mask bits, pooled | plain: unused
        float av[2][2];     // plain: the four deltas of the window
        float bv[2][2][2];  // [tile][pr][pc]
    };
    auto compute = [&](auto EDGE_C, int wr, int wc_lo, int sw, const float* buf, bool mid_barrier) {
        constexpr bool EDGE = decltype(EDGE_C)::value;
        const float* const xb = buf;
        const float* const ab = buf + XBUF;
        const int groups = (sw + 3) >> 2;
        const bool rowv0 = 2 * wr < p.Ho, rowv1 = 2 * wr + 1 < p.Ho;
        int nv;  // windows of this strip that carry a delta at all (pooled: inside the pooled domain)
        if constexpr (POOLED) nv = wr < p.PHo ? (p.PWo - wc_lo < sw ? p.PWo - wc_lo : sw) : 0;
        else nv = sw;
        const int e_lane = co * p.Ho * p.Wo + 2 * wr * p.Wo + 2 * (wc_lo + k);  // flat index of pixel (0,0) of window 4*0 + k
        auto load_ops = [&](int g, Ops& o) {
            const int w = 4 * g + k;
            if constexpr (kExp == 3) {  // (experiment: MFMA issue alone)
                o.a0 = o.a1 = o.a2 = (float)lane;
                for (int t = 0; t < 2; ++t) for (int pr = 0; pr < 2; ++pr) o.bv[t][pr][0] = o.bv[t][pr][1] = (float)(lane + g);
                for (int pr = 0; pr < 2; ++pr) for (int pc = 0; pc < 2; ++pc) o.av[pr][pc] = 1.f;
                return;
            }
            if constexpr (POOLED) {
                o.a0 = ab[64 * g + 4 * co + k];
                if constexpr (POOLED == 3) o.a1 = ab[kSW * CO + 16 * g + co];  // the group's four mask bytes of this channel
                else o.a1 = ab[kSW * CO + 64 * g + 4 * co + k];
                if constexpr (POOLED == 1) o.a2 = ab[2 * kSW * CO + 64 * g + 4 * co + k];
                (void)w;
            } else {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) o.av[pr][pc] = ab[(pr * 2 * kSW + 2 * w + pc) * CO + co];
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const float* src = xb + bbase[t][pr] + 16 * g;
                    o.bv[t][pr][0] = src[0];
                    o.bv[t][pr][1] = src[2];
                }
        };
        auto run_group = [&](int g, const Ops& o) {
            const int w = 4 * g + k;  // this lane's window inside the strip
            const bool winv = w < nv;
            const int q0 = 2 * (wc_lo + w);  // conv column of the window's pc = 0 pixel
            float av[2][2];
            if constexpr (POOLED) {
                float dp = o.a0;
                if constexpr (POOLED == 1) dp = (o.a2 <= 0.f) ? 0.f : dp;  // ReLU::backward in the pooled domain (relu.cpp:37)
                if (EDGE) dp = winv ? dp : 0.f;                             // (a window outside the pooled domain: stale LDS)
                if constexpr (POOLED == 3) {
                    const int code = (__builtin_bit_cast(int, o.a1) >> (8 * k)) & 0xff;  // 2 * row + column of the maximum | 0x80: ReLU-dead
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                        for (int pc = 0; pc < 2; ++pc) av[pr][pc] = (code == 2 * pr + pc) ? dp : 0.f;  // MaxPool2D::backward (pool2d.cpp:105)
                } else {
                    const int d = __builtin_bit_cast(int, o.a1) - (e_lane + 8 * g);  // mask - flat index of pixel (0,0)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                        for (int pc = 0; pc < 2; ++pc) av[pr][pc] = (d == pr * p.Wo + pc) ? dp : 0.f;  // MaxPool2D::backward (pool2d.cpp:105)
                }
            } else {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        const bool ok = !EDGE || (winv && (pr ? rowv1 : rowv0) && q0 + pc < p.Wo);
                        av[pr][pc] = ok ? o.av[pr][pc] : 0.f;
                    }
            }
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
                    const bool ok = !EDGE || (winv && (pr ? rowv1 : rowv0) && q0 + pc < p.Wo);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        if constexpr (kExp == 4) acc[t][pr][0] += av[pr][pc] * o.bv[t][pr][pc];  // (experiment: everything but the MFMAs)
                        else acc[t][pr] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[pr][pc], ok ? o.bv[t][pr][pc] : 0.f, acc[t][pr], 0, 0, 0);
                    }
                }
        };
        // software pipeline: the LDS reads of group g+1 are issued in front of the MFMAs of group g
        Ops ops[2];
        load_ops(0, ops[0]);
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
            if (g < groups) {
                if (g + 1 < kGroups && g + 1 < groups) load_ops(g + 1, ops[(g + 1) & 1]);
                run_group(g, ops[g & 1]);
            }
            if (g == 1 && mid_barrier) __builtin_amdgcn_s_barrier();  // (skewed lockstep, see the strip loop)
        }
    };

    // LDS -> LDS: strip s+1 continues strip s (same image, same column segment, next window row): its input row 0 is this strip's
    // row 4 (3 channels x 17 16-byte chunks; its DMA leaves row 0 alone) instead of a second trip to L2 / HBM (the re-read rows
    // were 25 % extra fetch traffic).  Inline asm: for a compiler-visible LDS access the waitcnt pass inserts s_waitcnt vmcnt(0)
    // -- it cannot tell that the in-flight LDS-DMA writes land elsewhere -- which drained the whole prefetch pipeline once per strip.
    auto hand_over = [&](const float* from, float* to) {
        if (lane < CI * (XP / 4)) {
            const int ci = lane / (XP / 4), ch = lane - ci * (XP / 4);
            const unsigned a_from = (unsigned)(uintptr_t)(lds_void_ptr)(from + (ci * XR + 4) * XP + 4 * ch);
            const unsigned a_to = (unsigned)(uintptr_t)(lds_void_ptr)(to + (ci * XR) * XP + 4 * ch);
            f32x4 v;
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tds_write_b128 %2, %0\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(v)
                         : "v"(a_from), "v"(a_to)
                         : "memory");
        }
    };
    auto edge_strip = [&](int wr, int wc_lo, int sw) {
        return 2 * wr + 1 >= p.Ho || 2 * (wc_lo + ((sw + 3) & ~3)) > p.Wo || 4 * wr + XR > p.H || 4 * (wc_lo + sw) + 4 > p.W ||
               (POOLED && (wr >= p.PHo || wc_lo + ((sw + 3) & ~3) > p.PWo));
    };

    if constexpr (SPEC) {
        // (host: every wave of the grid has exactly strips_per_wave strips)
        const int n = p.strips_per_wave;
        int* const flags = reinterpret_cast<int*>(lds + kWaves * NBUF * BUF);  // ready[kWaves][NBUF], done[kWaves]
        if (threadIdx.x < kWaves * NBUF) flags[threadIdx.x] = -1;
        if (threadIdx.x < kWaves) flags[kWaves * NBUF + threadIdx.x] = 0;
        __syncthreads();
        const unsigned flags_lds = (unsigned)(uintptr_t)(lds_void_ptr)flags;
        if (wave >= kWaves) {
            // ---------------- producer: DMA for MFMA waves cA and cB (the two on this wave's SIMD) ----------------
            const int cA = wave - kWaves, cB = cA + kProd;
            const int fA = (blockIdx.x * kWaves + cA) * n, fB = (blockIdx.x * kWaves + cB) * n;
            Cur ca = decode(fA), cb = decode(fB);
            ColState stA, stB;
            stA.col_seg = stB.col_seg = -1;
            float* const bufA = lds + cA * NBUF * BUF;
            float* const bufB = lds + cB * NBUF * BUF;
            // (flag traffic in inline asm: a compiler-visible LDS access behind LDS-DMA gets an s_waitcnt vmcnt(0) in front)
            auto lds_load = [&](unsigned addr) {
                int v;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
                return v;
            };
            auto publish = [&](int c, int strip) {
                if (lane == 0) {
                    const unsigned addr = flags_lds + 4u * (unsigned)(c * NBUF + strip % NBUF);
                    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(strip) : "memory");
                }
            };
            int buf = 0;
            for (int r = 0; r < n; ++r) {
                if (r >= NBUF) {
                    // buffer r % NBUF is free once strip r - NBUF has been consumed -- by EVERY consumer of the workgroup
                    while (true) {
                        const int who = lane & (kWaves - 1);
                        int d = lds_load(flags_lds + 4u * (unsigned)(kWaves * NBUF + who));
                        if (who != cA && who != cB) d += p.spec_slack;  // (own consumers: the buffer; the others: drift only)
#pragma unroll
                        for (int o = 1; o < kWaves; o <<= 1) {
                            const int e = __shfl_xor(d, o, 64);
                            d = e < d ? e : d;
                        }
                        if (__builtin_amdgcn_readfirstlane(d) >= r - NBUF + 1) break;
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
                const bool slA = stage(ca, bufA + buf * BUF, stA, fA);
                advance(ca);
                const bool slB = stage(cb, bufB + buf * BUF, stB, fB);
                advance(cb);
                if (r >= 1) {  // the strips of round r-1 have landed once at most this round's instructions are outstanding
                    if (slA && slB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * N_SLOW) : "memory");
                    else if (slA || slB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_FAST + N_SLOW) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * N_FAST) : "memory");
                    publish(cA, r - 1);
                    publish(cB, r - 1);
                }
                buf = buf + 1 == NBUF ? 0 : buf + 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            publish(cA, n - 1);
            publish(cB, n - 1);
        } else {
            // ---------------- consumer: MFMA groups only ----------------
            volatile int* const ready = flags + wave * NBUF;
            volatile int* const done = flags + kWaves * NBUF + wave;
            Cur cc = decode(s_lo);
            int cur = 0;
            for (int i = 0; i < n; ++i, advance(cc)) {
                while (ready[cur] != i) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");  // (the operand reads below stay behind the flag)
                int wc_lo, sw;
                seg_geom(cc.seg, wc_lo, sw);
                if (p.dbg != 1) {
                    if (edge_strip(cc.wr, wc_lo, sw)) compute(std::true_type(), cc.wr, wc_lo, sw, wbuf + cur * BUF, false);
                    else compute(std::false_type(), cc.wr, wc_lo, sw, wbuf + cur * BUF, false);
                }
                const int nxt = cur + 1 == NBUF ? 0 : cur + 1;
                if (i + 1 < n && cc.wr + 1 < p.WR) hand_over(wbuf + cur * BUF, wbuf + nxt * BUF);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every LDS read of this buffer is done: hand it back
                if (lane == 0) *done = i + 1;
                cur = nxt;
            }
        }
    } else if (s_lo < s_hi) {
        // NBUF - 1 strips ahead: strip s is consumed from buffer s % NBUF while s+1 (.. s+NBUF-1) are on their way
        bool slow_next = false;  // kind of the most recently staged strip that is still in flight behind strip s
        Cur sc = decode(s_lo), cc = sc;  // staging cursor / compute cursor
        ColState own;
        own.col_seg = -1;
        stage(sc, wbuf, own, s_lo);
        advance(sc);
        if (NBUF == 3 && sc.s < s_hi && p.dbg < 2) {
            slow_next = stage(sc, wbuf + BUF, own, s_lo);
            advance(sc);
        }
        int cur = 0;
        unsigned long long tp[6] = {0, 0, 0, 0, 0, 0}, tq = 0;
        auto tick = [&](int i) {
            if constexpr (kExp == 6) {
                const unsigned long long now = __builtin_readcyclecounter();
                if (i >= 0) tp[i] += now - tq;
                tq = now;
            }
        };
        for (; cc.s < s_hi; advance(cc)) {
            const int s = cc.s, wr = cc.wr;
            tick(-1);
            // lockstep: one workgroup barrier per strip keeps the eight waves -- eight neighbouring runs of the image -- within a strip
            // of each other, so the delta rows' 128-byte lines are fetched once (L2 hit for the other waves).  Skewed (2): the second
            // wave of every SIMD takes ITS barrier half-way through its MFMAs, so that one wave's MFMAs cover the other's DMA issue /
            // LDS latency instead of both waves of a SIMD stalling and computing together.
            const bool late = p.lockstep == 2 && wave >= kWaves / 2;
            if (p.lockstep && !late) __builtin_amdgcn_s_barrier();
            tick(0);
            // strip s has landed once at most the instructions of strip s+1 are outstanding
            if (NBUF == 3 && s + 1 < s_hi && p.dbg < 2) {
                if (slow_next) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_SLOW) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_FAST) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            tick(1);
            int wc_lo, sw;
            seg_geom(cc.seg, wc_lo, sw);
            int nb = cur + NBUF - 1;  // the buffer of strip s-1, which the next stage() call re-uses
            nb = nb >= NBUF ? nb - NBUF : nb;
            if (s + NBUF - 1 < s_hi && p.dbg < 2) {
                if (NBUF == 2) {  // (two buffers: the staging cursor runs one strip ahead, not two)
                    slow_next = stage(sc, wbuf + nb * BUF, own, s_lo);
                    advance(sc);
                } else {
                    slow_next = stage(sc, wbuf + nb * BUF, own, s_lo);
                    advance(sc);
                }
            }
            tick(2);
            if (p.dbg != 1) {
                const bool edge = edge_strip(wr, wc_lo, sw);
                if (edge) compute(std::true_type(), wr, wc_lo, sw, wbuf + cur * BUF, late);
                else compute(std::false_type(), wr, wc_lo, sw, wbuf + cur * BUF, late);
            } else if (late) {
                __builtin_amdgcn_s_barrier();
            }
            if constexpr (kExp == 6) asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[1][0]), "v"(acc[1][1]));
            tick(3);
            const int nxt = cur + 1 == NBUF ? 0 : cur + 1;
            if (s + 1 < s_hi && wr + 1 < p.WR && p.dbg < 2) hand_over(wbuf + cur * BUF, wbuf + nxt * BUF);  // (behind this strip's MFMAs)
            cur = nxt;
            tick(4);
        }
#if CNN_WIN_EXPERIMENT == 6
        if (lane == 0 && blockIdx.x < 256)
            for (int i = 0; i < 5; ++i) g_prof[blockIdx.x * 8 + wave][i] = tp[i];
#endif
    }

    // ---- the eight waves in a fixed order -> one slab per workgroup
    __syncthreads();
    float(*red)[CO][33] = reinterpret_cast<float(*)[CO][33]>(lds);  // [wave][co][column | 27: bias]   (the staging buffers are dead)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = acc[t][0][r] + acc[t][1][r];
            const int col = n0 + 16 * t;
            if (col < 28 && wave < kWaves) red[wave][4 * k + r][col] = v;  // D[row = 4*(lane>>4) + r][column = lane & 15]
        }
    __syncthreads();
    for (int i = threadIdx.x; i < CO * 28; i += blockDim.x) {
        const int c2 = i / 28, col = i - c2 * 28;
        float v = red[0][c2][col];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) v += red[w][c2][col];
        p.slabs[(size_t)blockIdx.x * CO * 28 + i] = v;
    }
}

}  // namespace

namespace cnn_amd {

#define CONV_TAG(d) "B%d Ci%d %dx%d Co%d k%d s%d p%d", (d)->B, (d)->Ci, (d)->H, (d)->W, (d)->Co, (d)->k, (d)->s, (d)->pad

namespace {
bool make_win_params(const cnn_conv2d_desc* d, WinParams* p, int* grid) {
    if (d->Ci != 3 || d->Co != 16 || d->k != 3 || d->s != 2 || d->pad != 0) return false;
    const OptVal e = CNN_OPT_VAL("WG_WIN");
    if (e && atoi(e) == 0) return false;
    p->B = d->B; p->H = d->H; p->W = d->W;
    p->Ho = cnn_conv2d_out_dim(d->H, 3, 2, 0);
    p->Wo = cnn_conv2d_out_dim(d->W, 3, 2, 0);
    if (p->Ho < 1 || p->Wo < 1) return false;
    if ((long long)CO * p->Ho * p->Wo >= (1ll << 31)) return false;  // (the pool mask is an int32 flat index)
    p->PHo = p->Ho / 2; p->PWo = p->Wo / 2;
    p->pitch8 = pool_mask_pitch(p->PWo);
    p->WR = (p->Ho + 1) / 2; p->WC = (p->Wo + 1) / 2;
    p->nseg = (p->WC + kSW - 1) / kSW;
    p->SW = (((p->WC + p->nseg - 1) / p->nseg) + 3) / 4 * 4;  // balanced segments, whole groups of 4 windows
    const long long strips = (long long)d->B * p->WR * p->nseg;
    if (strips >= (1ll << 30)) return false;
    p->strips_total = (int)strips;
    // one workgroup (8 waves, 135 KB of LDS) per CU
    long long g = (strips + kWaves - 1) / kWaves;
    if (g > num_cus()) g = num_cus();
    *grid = (int)g;
    p->strips_per_wave = (int)((strips + g * kWaves - 1) / (g * kWaves));
    p->dbg = CNN_MEASURE_INT("WIN_DBG", 0);
    {
        // default on: the eight waves of a workgroup walk eight neighbouring column-segment runs; in step they share the delta rows'
        // 128-byte lines in L2 (PMC: 1.37x -> 1.02x of the algorithmic fetch).  Needs the same trip count in every wave.
        const OptVal e = CNN_OPT_VAL("WIN_LOCKSTEP");
        p->lockstep = ((!e || atoi(e) != 0) && strips == (long long)g * kWaves * p->strips_per_wave) ? (e ? atoi(e) : 1) : 0;
        const OptVal sp = CNN_OPT_VAL("WIN_SPEC");
        p->spec = ((!sp || atoi(sp) != 0) && strips == (long long)g * kWaves * p->strips_per_wave && p->dbg < 2) ? 1 : 0;
        p->spec_slack = CNN_OPT_INT("WIN_SLACK", 0);
    }
    return true;
}

template <int POOLED, bool DMA16, bool SPEC>
int launch_win3(const cnn_conv2d_desc* d, WinParams& p, int grid, hipStream_t s, const char* name) {
    const size_t lds_bytes = (size_t)kWaves * num_bufs(POOLED) * buf_floats(POOLED) * sizeof(float) + (SPEC ? 256 : 0);
    static DeviceOnce attr_once;  // (per template instance)
    if (attr_once.needed()) {
        CNN_HIP_CHECK(hipFuncSetAttribute((const void*)conv_wgrad_win_kernel<POOLED, DMA16, SPEC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)lds_bytes));
        attr_once.mark();
    }
    constexpr int threads = (kWaves + (SPEC ? kProd : 0)) * 64;
    CNN_KLAUNCH(s, name, (conv_wgrad_win_kernel<POOLED, DMA16, SPEC><<<grid, threads, lds_bytes, s>>>(p)), CONV_TAG(d));
#if CNN_WIN_EXPERIMENT == 6
    if constexpr (!SPEC) {  // per-phase cycle counters of the strip loop (timing experiments only)
        static unsigned long long h[256 * 8][8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prof), sizeof(h));
        const char* names[5] = {"barrier", "dma wait", "stage issue", "compute", "hand-over"};
        for (int w = 0; w < 8; ++w) {
            fprintf(stderr, "[win prof %s] wave %d:", name, w);
            for (int i = 0; i < 5; ++i) {
                double sum = 0;
                for (int g = 0; g < grid && g < 256; ++g) sum += (double)h[g * 8 + w][i];
                fprintf(stderr, "  %s %.0f", names[i], sum / (grid < 256 ? grid : 256) / p.strips_per_wave);
            }
            fprintf(stderr, "  (cycles per strip)\n");
        }
    }
#endif
    return CNN_AMD_OK;
}

template <int POOLED>
int launch_win(const cnn_conv2d_desc* d, WinParams& p, int grid, hipStream_t s, const char* name) {
    const bool dma16 = (p.W % 4 == 0) && (reinterpret_cast<uintptr_t>(p.x) % 16 == 0);
    // (wave specialisation needs the three-buffer ring: with two, a producer cannot run ahead -- measured 150 vs 119 us)
    if (dma16) return (p.spec && num_bufs(POOLED) == 3) ? launch_win3<POOLED, true, POOLED >= 2>(d, p, grid, s, name) : launch_win3<POOLED, true, false>(d, p, grid, s, name);
    return launch_win3<POOLED, false, false>(d, p, grid, s, name);  // (4-byte DMA: too many instructions per strip for counted waits on pairs)
}
}  // namespace

// number of 16 x 28 slabs (= workgroups) the window kernel writes; 0 = geometry not covered / switched off (CNN_AMD_WG_WIN=0)
int win_wgrad_slots(const cnn_conv2d_desc* d) {
    WinParams p;
    int grid = 0;
    return make_win_params(d, &p, &grid) ? grid : 0;
}

// pooled == nullptr && mask != nullptr: dpool is pre-masked; mask == nullptr: dy is the materialised delta of the conv output
int win_wgrad_launch(const cnn_conv2d_desc* d, const float* x, const float* dy, const int32_t* mask, const float* pooled, float* slabs,
                     hipStream_t s) {
    WinParams p;
    int grid = 0;
    if (!make_win_params(d, &p, &grid)) return fail(CNN_AMD_E_BADARG, "conv_wgrad_win: geometry not covered");
    p.x = x; p.dy = dy; p.pmask = mask; p.pooled = pooled; p.slabs = slabs;
    if (d->flags & CNN_CONV2D_POOL_MASK_PACKED) {
        CNN_REQUIRE(mask != nullptr && pooled == nullptr, "conv_wgrad_win: packed pool mask: mask must be set, pooled must be NULL");
        return launch_win<3>(d, p, grid, s, "conv_wgrad_win<3,16,3,2>+poolm8");
    }
    if (mask == nullptr) return launch_win<0>(d, p, grid, s, "conv_wgrad_win<3,16,3,2>");
    if (pooled != nullptr) return launch_win<1>(d, p, grid, s, "conv_wgrad_win<3,16,3,2>+pool");
    return launch_win<2>(d, p, grid, s, "conv_wgrad_win<3,16,3,2>+poolm");
}

}  // namespace cnn_amd
