// conv_wgrad.hip -- Conv2D weight / bias gradient (cpu/src/conv2d.cpp:117-159) as a split-K GEMM on the fp32
// matrix cores:      gw[co][(ci,kx,ky)] = (1/divisor) * sum_{b,p,q} dy[b,co,p,q] * x[b,ci,p*s+kx-pad,q*s+ky-pad]
//   M = Co, N = Ci*k*k (already the reference's [Co][Ci][k][k] layout), K = B*Ho*Wo output pixels.
// MFMA operands: A[i=co][k=pixel] from a dy tile, B[k=pixel][j=(ci,tap)] gathered from the staged input rows --
// the "batched outer-product reduction" of north_star; im2col never exists in memory.
// The pixel axis is split over workgroups (and, for small M*N, over the waves of a workgroup); every split writes
// its partial [Co][N] slab and a second kernel adds the slabs in a FIXED order and divides once: deterministic, no
// atomics.  (The reference divides each sample's sum by B and accumulates, :148; algebraically (1/B)*sum.)
// Bias gradient (conv2d.cpp:153-157) is a per-channel two-stage reduction of dy.
#include <cstdlib>

#include "common.h"

using namespace cnn_amd;

namespace cnn_amd {
size_t igemm_workspace_floats(const cnn_conv2d_desc* d);  // conv_igemm.hip
int direct_wgrad_slots(const cnn_conv2d_desc* d);          // conv_direct.hip: thin first layer, packed VALU kernel
int direct_conv_wgrad(const cnn_conv2d_desc* d, const float* x, const float* dy, float* slabs, hipStream_t s);
bool direct_conv_pool_supported(const cnn_conv2d_desc* d);
int direct_conv_wgrad_pooled(const cnn_conv2d_desc* d, const float* x, const float* dpool, const int32_t* mask, const float* pooled,
                             float* slabs, hipStream_t s);
int direct_first_layer_finish(const cnn_conv2d_desc* d, const float* slabs, int nslots, float divisor, float* gw, float* gb, float* w,
                              float* bias, float lr, float grad_scale, void* fwd_img, void* dgrad_img, float* w_keep, float* bias_keep,
                              hipStream_t s);
int stem_wgrad_slots(const cnn_conv2d_desc* d);  // conv_stem.hip: 3 -> Co, 7x7, stride 2, pad 3
int os_wgrad_slots(const cnn_conv2d_desc* d);    // conv_wgrad_os.hip: the reference net's small 3x3 / stride-2 layers, output-stationary
int os_wgrad_launch(const cnn_conv2d_desc* d, const float* x, const float* dy, float* slabs, hipStream_t s);
int c11_wgrad_slots(const cnn_conv2d_desc* d);   // conv_1x1.hip: 1x1 convolutions (stride 1 / 2): split-K GEMM over the sub-sampled plane
int c11_wgrad_launch(const cnn_conv2d_desc* d, const float* x, const float* dy, float* slabs, hipStream_t s);
int stem_wgrad_launch(const cnn_conv2d_desc* d, const float* x, const float* dy, float* slabs, hipStream_t s);
int sp_wgrad_slots(const cnn_conv2d_desc* d);  // conv_wgrad_sp.hip: small planes (7x7 .. 56x56), 3x3 / stride 1 / pad 1, LDS-staged output-stationary
int sp_wgrad_launch(const cnn_conv2d_desc* d, const float* x, const float* dy, float* slabs, hipStream_t s);
int wgrad_rd_slots(const cnn_conv2d_desc* d);  // conv_wgrad_rd.hip: register-direct MFMA kernel (3x3, stride 1/2, pad 0)
int wgrad_rd_launch(const cnn_conv2d_desc* d, const float* x, const float* dy, float* slabs, hipStream_t s);
int wgrad_rd_pooled_slots(const cnn_conv2d_desc* d);
int wgrad_rd_launch_pooled(const cnn_conv2d_desc* d, const float* x, const float* dpool, const int32_t* mask, const float* pooled,
                           float* slabs, hipStream_t s);
}

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WgradParams {
    const float* x;
    const float* dy;
    float* part;  // [nslots][Co][Ntot]
    int B, Ci, H, W, Co, k, s, pad, Ho, Wo;
    int R, QC, QCP;    // chunk = R output rows x QC output columns (QCP = QC rounded up to the MFMA k-step)
    int nrc, ncc;      // chunks per image along rows / columns
    int DROW;          // LDS floats per dy channel (R*QCP, padded for banking)
    int XR, LWc, CHS;  // staged input rows per channel, their pitch, channel stride
    int CIB;           // channel slots staged per workgroup
    int Ntot;          // Ci*k*k
    long long chunks_total;
    int chunks_per_split;
    int rwd_shift, rwx_shift;  // log2(lanes per staged dy / x row)
    int dbg;                   // ablation bits (CNN_AMD_DBG, tuning only): 1 no dy staging, 2 no x staging, 4 no MFMA
    int fuse_bias;             // column Ntot of the output tile accumulates sum(dy) (B operand = 1): the bias gradient
    int pitch;                 // floats per output row of a partial slab (Ntot, or Ntot + 1 when fuse_bias)
};

template <int MF>
struct Mfma;
template <>
struct Mfma<32> {
    typedef f32x16 type;
    static constexpr int kRegs = 16, kStep = 2;
    __device__ static __forceinline__ type run(float a, float b, type c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ int row(int reg, int lh) { return (reg & 3) + 8 * (reg >> 2) + 4 * lh; }
};
template <>
struct Mfma<16> {
    typedef f32x4 type;
    static constexpr int kRegs = 4, kStep = 4;
    __device__ static __forceinline__ type run(float a, float b, type c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ int row(int reg, int lh) { return 4 * lh + reg; }
};

// MA x NB MFMA tiles per wave, WM x WN waves tile the (co, n) block, WK waves split the chunk's rows.
// NARROW (compile time): every staged row fits one 64-lane load -> 8 rows x 1 load in flight per lane; otherwise 4 x 4.
// PC (compile time): producer/consumer wave specialisation.  The workgroup has 2*NWAVES waves; waves [0,NWAVES) only
// issue MFMAs on chunk c while waves [NWAVES,2*NWAVES) only stage chunk c+1 into the second LDS buffer; one barrier per
// chunk.  (Two ordinary workgroups per CU were measured to run in lockstep: staging and compute times simply added up.)
template <int MF, int MA, int NB, int WM, int WN, int WK, bool NARROW, bool PC>
__global__ __launch_bounds__(64 * WM * WN * WK * (PC ? 2 : 1), PC ? 2 : 2) void wgrad_kernel(const WgradParams p) {
    using M_ = Mfma<MF>;
    constexpr int NWAVES = WM * WN * WK;  // waves per role
    constexpr int MTB = MF * MA * WM;
    constexpr int NTB = MF * NB * WN;
    constexpr int KSTEP = M_::kStep;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int buf_floats = MTB * p.DROW + p.CIB * p.CHS;  // one {dy tile, input rows} pair (PC uses two)
    float* Ds = smem;                              // [MTB][DROW]
    float* Xs = smem + (size_t)MTB * p.DROW;       // [CIB][CHS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = PC && wave_all >= NWAVES;                // wave-uniform role
    const int wave = producer ? wave_all - NWAVES : wave_all;      // index within the role
    const int wk = wave / (WM * WN), wmn = wave % (WM * WN), wm = wmn / WN, wn = wmn % WN;
    const int li = lane & (MF - 1), lh = lane / MF;
    const int split = blockIdx.x, nblk = blockIdx.y, mblk = blockIdx.z;
    const int kk2 = p.k * p.k;
    const int ci_first = (nblk * NTB) / kk2;

    int a_off[MA], b_off[NB];
    bool ones[NB];  // this lane's column is the fused bias-gradient column
#pragma unroll
    for (int ma = 0; ma < MA; ++ma) a_off[ma] = ((wm * MA + ma) * MF + li) * p.DROW + lh;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        int n = nblk * NTB + (wn * NB + nb) * MF + li;
        ones[nb] = p.fuse_bias && n == p.Ntot;
        if (n >= p.Ntot) n = nblk * NTB;  // padded column: reads something valid, never stored
        const int ci = n / kk2, tap = n - ci * kk2, kx = tap / p.k, ky = tap - kx * p.k;
        b_off[nb] = (ci - ci_first) * p.CHS + kx * p.LWc + ky + lh * p.s;
    }

    typename M_::type acc[MA][NB];
#pragma unroll
    for (int ma = 0; ma < MA; ++ma)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < M_::kRegs; ++r) acc[ma][nb][r] = 0.f;

    const long long ch_begin = (long long)split * p.chunks_per_split;
    long long ch_end = ch_begin + p.chunks_per_split;
    if (ch_end > p.chunks_total) ch_end = p.chunks_total;
    const int per_img = p.nrc * p.ncc;

    // stage(ch, dst buffer) and compute(ch, src buffer) as lambdas over the chunk geometry
    struct ChunkGeo { int b, p0, q0, Rv, QCv; };
    auto chunk_geo = [&](long long ch) {
        ChunkGeo g;
        g.b = (int)(ch / per_img);
        const int rem = (int)(ch - (long long)g.b * per_img);
        const int rc = rem / p.ncc, cq = rem - rc * p.ncc;
        g.p0 = rc * p.R; g.q0 = cq * p.QC;
        g.Rv = (p.Ho - g.p0 < p.R) ? p.Ho - g.p0 : p.R;
        g.QCv = (p.Wo - g.q0 < p.QC) ? p.Wo - g.q0 : p.QC;
        return g;
    };
    auto stage_chunk = [&](const ChunkGeo& g, float* Ds, float* Xs) {
        const int b = g.b, p0 = g.p0, q0 = g.q0, Rv = g.Rv, QCv = g.QCv;
        (void)b; (void)p0; (void)q0; (void)Rv; (void)QCv;
        constexpr int kUn = NARROW ? 8 : 4, kMaxC = NARROW ? 1 : 4;
        // ---- stage dy tile rows (zero beyond the valid columns / rows / channels); narrow rows share a wave-wide
        //      load, kUn loads are in flight per lane before the first LDS store ----
        if (!(p.dbg & 1)) {
            const int RW = 1 << p.rwd_shift, RPI = 64 >> p.rwd_shift;
            const int sub = lane >> p.rwd_shift, col0 = lane & (RW - 1);
            const int total = MTB * p.R;
            for (int cb = 0; cb < p.QCP; cb += 64 * kMaxC) {
                for (int rb = wave * RPI; rb < total; rb += NWAVES * RPI * kUn) {
                    float v[kUn][kMaxC];
#pragma unroll
                    for (int u = 0; u < kUn; ++u) {
                        const int ridx = rb + u * NWAVES * RPI + sub;
                        const int col_ = ridx / p.R, r = ridx - col_ * p.R;
                        const int co = mblk * MTB + col_;
                        const bool live = (ridx < total) && (co < p.Co) && (r < Rv);
                        const float* g =
                            p.dy + (((size_t)b * p.Co + (live ? co : 0)) * p.Ho + p0 + (live ? r : 0)) * p.Wo + q0;
#pragma unroll
                        for (int ci = 0; ci < kMaxC; ++ci) {
                            const int c = cb + col0 + ci * 64;
                            if (ci == 0 || RW == 64) v[u][ci] = (live && c < QCv) ? g[c] : 0.f;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kUn; ++u) {
                        const int ridx = rb + u * NWAVES * RPI + sub;
                        const int col_ = ridx / p.R, r = ridx - col_ * p.R;
                        float* d = Ds + col_ * p.DROW + r * p.QCP;
#pragma unroll
                        for (int ci = 0; ci < kMaxC; ++ci) {
                            const int c = cb + col0 + ci * 64;
                            if ((ci == 0 || RW == 64) && ridx < total && c < p.QCP) d[c] = v[u][ci];
                        }
                    }
                }
            }
        }
        // ---- stage input rows (zero outside the image: padding and the k-step overreach) ----
        if (!(p.dbg & 2)) {
            const int RW = 1 << p.rwx_shift, RPI = 64 >> p.rwx_shift;
            const int sub = lane >> p.rwx_shift, col0 = lane & (RW - 1);
            const int total = p.CIB * p.XR;
            const int w0 = q0 * p.s - p.pad;
            for (int cb = 0; cb < p.LWc; cb += 64 * kMaxC) {
                for (int rb = wave * RPI; rb < total; rb += NWAVES * RPI * kUn) {
                    float v[kUn][kMaxC];
#pragma unroll
                    for (int u = 0; u < kUn; ++u) {
                        const int ridx = rb + u * NWAVES * RPI + sub;
                        const int slot = ridx / p.XR, xr = ridx - slot * p.XR;
                        const int ci_ = ci_first + slot;
                        const int h = p0 * p.s + xr - p.pad;
                        const bool live = (ridx < total) && (ci_ < p.Ci) && (h >= 0) && (h < p.H);
                        const float* g = p.x + (((size_t)b * p.Ci + (live ? ci_ : 0)) * p.H + (live ? h : 0)) * p.W;
#pragma unroll
                        for (int ci = 0; ci < kMaxC; ++ci) {
                            const int c = cb + col0 + ci * 64;
                            const int w = w0 + c;
                            if (ci == 0 || RW == 64) v[u][ci] = (live && w >= 0 && w < p.W) ? g[w] : 0.f;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kUn; ++u) {
                        const int ridx = rb + u * NWAVES * RPI + sub;
                        const int slot = ridx / p.XR, xr = ridx - slot * p.XR;
                        float* d = Xs + slot * p.CHS + xr * p.LWc;
#pragma unroll
                        for (int ci = 0; ci < kMaxC; ++ci) {
                            const int c = cb + col0 + ci * 64;
                            if ((ci == 0 || RW == 64) && ridx < total && c < p.LWc) d[c] = v[u][ci];
                        }
                    }
                }
            }
        }
    };
    auto compute_chunk = [&](const ChunkGeo& g, const float* Ds, const float* Xs) {
        const int b = g.b, p0 = g.p0, q0 = g.q0, Rv = g.Rv, QCv = g.QCv;
        (void)b; (void)p0; (void)q0; (void)Rv; (void)QCv;
        // ---- MFMA: k-steps walk the chunk's pixels ----
        for (int r = wk; r < Rv && !(p.dbg & 4); r += WK) {
            const int arow = r * p.QCP, brow = r * p.s * p.LWc;
            // software pipelined by one k-step: the LDS reads of step q+KSTEP are issued above the MFMAs of step q
            float a_cur[MA], b_cur[NB];
#pragma unroll
            for (int ma = 0; ma < MA; ++ma) a_cur[ma] = Ds[a_off[ma] + arow];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float xv = Xs[b_off[nb] + brow];
                b_cur[nb] = ones[nb] ? 1.f : xv;  // dy is 0 on padded pixels, so the ones column sums exactly dy
            }
            for (int q = 0; q < p.QCP; q += KSTEP) {
                const int qn = (q + KSTEP < p.QCP) ? q + KSTEP : 0;  // last step: harmless re-read of step 0
                float a_nxt[MA], b_nxt[NB];
#pragma unroll
                for (int ma = 0; ma < MA; ++ma) a_nxt[ma] = Ds[a_off[ma] + arow + qn];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float xv = Xs[b_off[nb] + brow + qn * p.s];
                    b_nxt[nb] = ones[nb] ? 1.f : xv;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ma = 0; ma < MA; ++ma)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[ma][nb] = M_::run(a_cur[ma], b_cur[nb], acc[ma][nb]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ma = 0; ma < MA; ++ma) a_cur[ma] = a_nxt[ma];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) b_cur[nb] = b_nxt[nb];
            }
        }
    };

    if constexpr (!PC) {
        for (long long ch = ch_begin; ch < ch_end; ++ch) {
            const ChunkGeo g = chunk_geo(ch);
            __syncthreads();
            stage_chunk(g, Ds, Xs);
            __syncthreads();
            compute_chunk(g, Ds, Xs);
        }
    } else {
        // Separate loops per role (their live ranges must not overlap: the MFMA role alone needs ~200 VGPRs).  Both roles
        // execute exactly the same number of barriers: one after the first chunk is staged, then one per chunk.
        if (producer) {
            if (ch_begin < ch_end) stage_chunk(chunk_geo(ch_begin), Ds, Xs);
            __syncthreads();
            for (long long ch = ch_begin; ch < ch_end; ++ch) {
                float* Dnxt = smem + (int)((ch - ch_begin + 1) & 1) * buf_floats;
                if (ch + 1 < ch_end) stage_chunk(chunk_geo(ch + 1), Dnxt, Dnxt + MTB * p.DROW);
                __syncthreads();  // chunk ch+1 is staged / every MFMA wave is done with chunk ch's buffer
            }
            return;
        }
        __syncthreads();
        for (long long ch = ch_begin; ch < ch_end; ++ch) {
            const float* Dcur = smem + (int)((ch - ch_begin) & 1) * buf_floats;
            compute_chunk(chunk_geo(ch), Dcur, Dcur + MTB * p.DROW);
            __syncthreads();
        }
    }

    // ---- write this (split, wk) slab ----
    float* out = p.part + ((size_t)split * WK + wk) * p.Co * p.pitch;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = nblk * NTB + (wn * NB + nb) * MF + li;
        if (n >= p.pitch) continue;
#pragma unroll
        for (int ma = 0; ma < MA; ++ma) {
            const int cbase = mblk * MTB + (wm * MA + ma) * MF;
#pragma unroll
            for (int r = 0; r < M_::kRegs; ++r) {
                const int co = cbase + M_::row(r, lh);
                if (co < p.Co) out[(size_t)co * p.pitch + n] = acc[ma][nb][r];
            }
        }
    }
}

// Deterministic slab reduction: out[g][i] = sum over slots of group g of in[slot][i] (optionally / divisor).
// A workgroup owns 32 consecutive elements; its 8 slot-lanes walk the group's slots with independent loads and
// are then combined through LDS in a fixed order, so the result does not depend on scheduling.
constexpr int kRedElems = 32, kRedLanes = 8;
// Jobs with MANY small slabs (conv_layer_2 of the reference net: 512 slabs of 4 640 floats) gave each of their few workgroups a chain
// of 64 dependent-ish loads per thread -- 40 us of the side stream in the train step.  Such a job runs with 32 slot-lanes x 8
// elements per workgroup instead of 8 x 32: four times the workgroups, a quarter of the chain.  The rule is part of the summation
// order (lane sl adds slots sl, sl + L, ...; then the L partial sums in lane order), so every reduction path applies the same one
// (first_layer_finish's slabs have n = 448 and never qualify).
constexpr int kRedWideLanes = 32;
__host__ __device__ inline int red_lanes(int nslots, size_t n) { return (nslots > 256 && n >= 1024) ? kRedWideLanes : kRedLanes; }
__global__ __launch_bounds__(kRedElems * kRedLanes) void slab_reduce(const float* __restrict__ in,
                                                                     float* __restrict__ out, int nslots, size_t n,
                                                                     int per_group, float divisor, int final_stage,
                                                                     int split_n, float* __restrict__ out_b, int lanes) {
    __shared__ float red[kRedWideLanes][kRedElems];
    const int E = (kRedElems * kRedLanes) / lanes;  // elements per workgroup: 32 (8 lanes) or 8 (32 lanes)
    const int e = threadIdx.x % E, sl = threadIdx.x / E;
    const size_t i = (size_t)blockIdx.x * E + e;
    const int g = blockIdx.y;
    const int s_begin = g * per_group;
    const int s_end = (s_begin + per_group < nslots) ? s_begin + per_group : nslots;
    float acc = 0.f;
    if (i < n) {
        int s = s_begin + sl;
        for (; s + 3 * lanes < s_end; s += 4 * lanes) {
            const float v0 = in[(size_t)s * n + i], v1 = in[(size_t)(s + lanes) * n + i];
            const float v2 = in[(size_t)(s + 2 * lanes) * n + i], v3 = in[(size_t)(s + 3 * lanes) * n + i];
            acc += v0; acc += v1; acc += v2; acc += v3;
        }
        for (; s < s_end; s += lanes) acc += in[(size_t)s * n + i];
    }
    red[sl][e] = acc;
    __syncthreads();
    if (sl == 0 && i < n) {
        float t = 0.f;
        for (int k = 0; k < lanes; ++k) t += red[k][e];
        if (final_stage && split_n > 0) {  // slab rows are [split_n weight gradients | 1 bias gradient]
            const size_t row = i / (split_n + 1), col = i - row * (split_n + 1);
            if (col < (size_t)split_n) out[row * split_n + col] = t / divisor;
            else if (out_b) out_b[row] = t / divisor;
        } else {
            out[(size_t)g * n + i] = final_stage ? t / divisor : t;
        }
    }
}

// ---- deferred final reductions ---------------------------------------------------------------------------------------------
// Conv2D::backward with defer_join (conv_backward.hip) runs the weight-gradient kernels of several layers on one side stream.
// Their 6 us slab reductions take 20-90 us each when they share the chip with a large kernel and used to sit BETWEEN the
// layers' gradient kernels; between wgrad_defer_reduce(true) and the join they are only recorded, and
// wgrad_flush_reduces() runs them all in ONE launch just before the join event (same summation order: bit-identical).
struct RedJob {
    const float* in;
    float* out;
    float* out_b;
    int nslots, split_n;
    unsigned n;
    float divisor;
};
constexpr int kMaxRedJobs = 8;
struct RedBatch {
    RedJob job[kMaxRedJobs];
};
struct PendingReduces {
    RedBatch batch;
    int count = 0;
    bool defer = false;
};
// one list per (host thread, device): the recorded jobs belong to that device's side stream (conv_backward.hip keeps one side
// stream per thread and device), so a thread that alternates between devices never flushes device A's pointers on device B
PendingReduces& pending_reduces() {
    static thread_local PendingReduces p[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return p[dev % 16];
}

__global__ __launch_bounds__(kRedElems * kRedLanes) void slab_reduce_batch(const RedBatch rb) {
    __shared__ float red[kRedWideLanes][kRedElems];
    const RedJob j = rb.job[blockIdx.y];
    const int L = red_lanes(j.nslots, j.n), E = (kRedElems * kRedLanes) / L;
    if ((size_t)blockIdx.x * E >= j.n) return;
    const int e = threadIdx.x % E, sl = threadIdx.x / E;
    const size_t i = (size_t)blockIdx.x * E + e, n = j.n;
    float acc = 0.f;
    if (i < n) {  // (the loop of slab_reduce's final stage: same order, same rounding)
        int s = sl;
        for (; s + 3 * L < j.nslots; s += 4 * L) {
            const float v0 = j.in[(size_t)s * n + i], v1 = j.in[(size_t)(s + L) * n + i];
            const float v2 = j.in[(size_t)(s + 2 * L) * n + i], v3 = j.in[(size_t)(s + 3 * L) * n + i];
            acc += v0; acc += v1; acc += v2; acc += v3;
        }
        for (; s < j.nslots; s += L) acc += j.in[(size_t)s * n + i];
    }
    red[sl][e] = acc;
    __syncthreads();
    if (sl == 0 && i < n) {
        float t = 0.f;
        for (int k = 0; k < L; ++k) t += red[k][e];
        if (j.split_n > 0) {
            const size_t row = i / (j.split_n + 1), col = i - row * (j.split_n + 1);
            if (col < (size_t)j.split_n) j.out[row * j.split_n + col] = t / j.divisor;
            else if (j.out_b) j.out_b[row] = t / j.divisor;
        } else {
            j.out[i] = t / j.divisor;
        }
    }
}

// the same with FOUR consecutive elements per thread (16-byte loads, a quarter of the workgroups): every element still sees its slots
// in the same order -> bit-identical.  The wide layers of the stacks reduce 2.4 M-element slabs: the scalar kernel launched 590 k
// workgroups of one load per thread for a batch of eight such jobs and ran at 0.6 TB/s (dispatch-bound).
__global__ __launch_bounds__(kRedElems * kRedLanes) void slab_reduce_batch_v4(const RedBatch rb) {
    __shared__ float4 red[kRedWideLanes][kRedElems];
    const RedJob j = rb.job[blockIdx.y];
    const int kRedLanes = red_lanes(j.nslots, j.n), kRedElems = (::kRedElems * ::kRedLanes) / kRedLanes;  // (per job, see red_lanes)
    if ((size_t)blockIdx.x * kRedElems * 4 >= j.n) return;
    const int e = threadIdx.x % kRedElems, sl = threadIdx.x / kRedElems;
    const size_t i = ((size_t)blockIdx.x * kRedElems + e) * 4, n = j.n;  // (n % 4 == 0: all four elements exist or none)
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        int s = sl;
        // jobs with hundreds of slabs (conv_layer_2: 512 x 18 KB) were a chain of 16 dependent rounds of four loads, each round a
        // memory round trip next to an HBM-bound kernel (48 us in the train step): sixteen loads in flight, added in the same order
        for (; s + 15 * kRedLanes < j.nslots; s += 16 * kRedLanes) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = *(const float4*)(j.in + (size_t)(s + u * kRedLanes) * n + i);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
            }
        }
        for (; s + 3 * kRedLanes < j.nslots; s += 4 * kRedLanes) {
            const float4 v0 = *(const float4*)(j.in + (size_t)s * n + i), v1 = *(const float4*)(j.in + (size_t)(s + kRedLanes) * n + i);
            const float4 v2 = *(const float4*)(j.in + (size_t)(s + 2 * kRedLanes) * n + i), v3 = *(const float4*)(j.in + (size_t)(s + 3 * kRedLanes) * n + i);
            acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
            acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
            acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
            acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
        }
        for (; s < j.nslots; s += kRedLanes) {
            const float4 v = *(const float4*)(j.in + (size_t)s * n + i);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[sl][e] = acc;
    __syncthreads();
    if (sl == 0 && i < n) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < kRedLanes; ++k) {
            const float4 v = red[k][e];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const size_t ii = i + c;
            if (j.split_n > 0) {
                const size_t row = ii / (j.split_n + 1), col = ii - row * (j.split_n + 1);
                if (col < (size_t)j.split_n) j.out[row * j.split_n + col] = tv[c] / j.divisor;
                else if (j.out_b) j.out_b[row] = tv[c] / j.divisor;
            } else {
                j.out[ii] = tv[c] / j.divisor;
            }
        }
    }
}

// (round 5) The same sums, ONE thread per four consecutive elements and no LDS: the thread walks the slots itself, eight partial sums in
// the order of the eight slot-lanes above (partial k adds slots k, k + 8, ...; then the partials in order 0..7) -- bit-identical to
// slab_reduce / slab_reduce_batch(_v4) for every job whose red_lanes() is 8.  The stacks' weight-gradient slabs (4 - 256 slabs of
// 0.15 - 9.4 MB) ran at 0.5 - 2 TB/s through the workgroup-per-32-elements kernels: with 4 slabs half of a workgroup's threads held no
// slot, every workgroup ended in a barrier and an LDS pass, and the output index cost four 64-bit divisions per thread.
__global__ __launch_bounds__(256) void slab_reduce_flat(const RedBatch rb) {
    const RedJob j = rb.job[blockIdx.y];
    const unsigned n4 = j.n / 4;
    const float4* const in4 = (const float4*)j.in;
    for (unsigned q = blockIdx.x * 256 + threadIdx.x; q < n4; q += gridDim.x * 256) {
        float4 part[kRedLanes];
#pragma unroll
        for (int k = 0; k < kRedLanes; ++k) part[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s0 = 0; s0 < j.nslots; s0 += kRedLanes) {
            float4 v[kRedLanes];
#pragma unroll
            for (int u = 0; u < kRedLanes; ++u)
                if (s0 + u < j.nslots) v[u] = in4[(size_t)(s0 + u) * n4 + q];
#pragma unroll
            for (int u = 0; u < kRedLanes; ++u)
                if (s0 + u < j.nslots) {
                    part[u].x += v[u].x; part[u].y += v[u].y; part[u].z += v[u].z; part[u].w += v[u].w;
                }
        }
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < kRedLanes; ++k) {
            t.x += part[k].x; t.y += part[k].y; t.z += part[k].z; t.w += part[k].w;
        }
        const float tv[4] = {t.x / j.divisor, t.y / j.divisor, t.z / j.divisor, t.w / j.divisor};
        const unsigned i = q * 4;
        if (j.split_n > 0) {  // slab rows are [split_n weight gradients | 1 bias gradient]
            const unsigned w = (unsigned)j.split_n + 1;
            unsigned row = i / w, col = i - row * w;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (col < (unsigned)j.split_n) j.out[(size_t)row * j.split_n + col] = tv[c];
                else if (j.out_b) j.out_b[row] = tv[c];
                if (++col == w) { col = 0; ++row; }
            }
        } else {
            *(float4*)(j.out + i) = make_float4(tv[0], tv[1], tv[2], tv[3]);
        }
    }
}
// a job the flat kernel takes: whole float4s, aligned, the 8-lane summation order
inline bool flat_job(const RedJob& j) {
    return j.n % 4 == 0 && reinterpret_cast<uintptr_t>(j.in) % 16 == 0 && red_lanes(j.nslots, j.n) == kRedLanes &&
           (j.split_n > 0 || reinterpret_cast<uintptr_t>(j.out) % 16 == 0);
}

int flush_reduces(hipStream_t s) {
    PendingReduces& p = pending_reduces();
    if (p.count == 0) return CNN_AMD_OK;
    bool v4 = !((CNN_OPT_SET("REDUCE_SCALAR") && CNN_OPT_INT("REDUCE_SCALAR", 0) != 0));
    for (int i = 0; i < p.count; ++i)
        if (p.batch.job[i].n % 4 != 0 || reinterpret_cast<uintptr_t>(p.batch.job[i].in) % 16 != 0) v4 = false;
    unsigned most = 0;  // workgroups along x: the job that needs the most (each job's elements per workgroup follow red_lanes)
    for (int i = 0; i < p.count; ++i) {
        const RedJob& j = p.batch.job[i];
        const unsigned per = (unsigned)((kRedElems * kRedLanes) / red_lanes(j.nslots, j.n)) * (v4 ? 4u : 1u);
        const unsigned need = (j.n + per - 1) / per;
        most = need > most ? need : most;
    }
    const int jobs = p.count;
    bool flat = !CNN_OPT_SET("REDUCE_OLD") && !CNN_OPT_SET("REDUCE_SCALAR");
    unsigned most4 = 0;
    for (int i = 0; i < jobs; ++i) {
        flat = flat && flat_job(p.batch.job[i]);
        most4 = p.batch.job[i].n / 4 > most4 ? p.batch.job[i].n / 4 : most4;
    }
    p.count = 0;
    if (flat) {
        unsigned gx = (most4 + 255) / 256;
        if (gx > 8192) gx = 8192;
        if (gx < 1) gx = 1;
        CNN_KLAUNCH(s, "slab_reduce/batch", (slab_reduce_flat<<<dim3(gx, jobs), 256, 0, s>>>(p.batch)), "jobs=%d", jobs);
        return CNN_AMD_OK;
    }
    if (v4) {
        CNN_KLAUNCH(s, "slab_reduce/batch", (slab_reduce_batch_v4<<<dim3(most, jobs), kRedElems * kRedLanes, 0, s>>>(p.batch)), "jobs=%d", jobs);
        return CNN_AMD_OK;
    }
    CNN_KLAUNCH(s, "slab_reduce/batch", (slab_reduce_batch<<<dim3(most, jobs), kRedElems * kRedLanes, 0, s>>>(p.batch)),
                "jobs=%d", jobs);
    return CNN_AMD_OK;
}

// sums `nslots` slabs of n floats held in `slabs` into dst (divided by divisor); `tmp` holds >= ceil(nslots/64)*n
int reduce_slabs(hipStream_t s, const float* slabs, int nslots, size_t n, float* tmp, float* dst, float divisor,
                 const char* tag, int split_n = 0, float* dst_b = nullptr) {
    PendingReduces& pend = pending_reduces();
    if (pend.defer && !(nslots > 512 || (nslots > 64 && n > 65536)) && n < (1ull << 31)) {
        if (pend.count == kMaxRedJobs)
            if (int rc = flush_reduces(s)) return rc;
        RedJob& j = pend.batch.job[pend.count++];
        j.in = slabs; j.out = dst; j.out_b = dst_b; j.nslots = nslots; j.split_n = split_n; j.n = (unsigned)n; j.divisor = divisor;
        return CNN_AMD_OK;
    }
    // two stages only when one workgroup per 32 elements would walk too many slabs (small slabs are launch-bound: one
    // launch of up to 512 slots x 8 slot-lanes beats two)
    if (nslots > 512 || (nslots > 64 && n > 65536)) {
        const int per = 64, groups = (nslots + per - 1) / per;
        const unsigned gx = (unsigned)((n + kRedElems - 1) / kRedElems);
        CNN_KLAUNCH(s, "slab_reduce/stage1",
                    (slab_reduce<<<dim3(gx, groups), kRedElems * kRedLanes, 0, s>>>(slabs, tmp, nslots, n, per, 1.f, 0, 0, nullptr, kRedLanes)), "%s", tag);
        slabs = tmp;
        nslots = groups;
    }
    if (n < (1ull << 31) && !CNN_OPT_SET("REDUCE_OLD") && !CNN_OPT_SET("REDUCE_SCALAR")) {
        RedBatch one;
        RedJob& j = one.job[0];
        j.in = slabs; j.out = dst; j.out_b = dst_b; j.nslots = nslots; j.split_n = split_n; j.n = (unsigned)n; j.divisor = divisor;
        if (flat_job(j)) {
            unsigned gx = (unsigned)((n / 4 + 255) / 256);
            if (gx > 8192) gx = 8192;
            if (gx < 1) gx = 1;
            CNN_KLAUNCH(s, "slab_reduce/final", (slab_reduce_flat<<<dim3(gx, 1), 256, 0, s>>>(one)), "%s", tag);
            return CNN_AMD_OK;
        }
    }
    const int lanes = red_lanes(nslots, n), elems = (kRedElems * kRedLanes) / lanes;
    CNN_KLAUNCH(s, "slab_reduce/final",
                (slab_reduce<<<dim3((unsigned)((n + elems - 1) / elems), 1), kRedElems * kRedLanes, 0, s>>>(slabs, dst, nslots, n, nslots, divisor, 1, split_n, dst_b, lanes)), "%s", tag);
    return CNN_AMD_OK;
}

// bias gradient stage 1: partial[g][co] = sum over this group's samples of sum_pq dy[b][co][pq]
constexpr int kBiasBlock = 256;
__global__ __launch_bounds__(kBiasBlock) void bias_grad_partial(const float* __restrict__ dy,
                                                                float* __restrict__ partial, int B, int Co, int P,
                                                                int groups) {
    __shared__ float red[kBiasBlock / 64];
    const int co = blockIdx.x, g = blockIdx.y;
    const int per = (B + groups - 1) / groups;
    const int bb = g * per, be = (bb + per < B) ? bb + per : B;
    float s = 0.f;
    for (int b = bb; b < be; ++b) {
        const float* d = dy + ((size_t)b * Co + co) * P;
        for (int i = threadIdx.x; i < P; i += kBiasBlock) s += d[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < kBiasBlock / 64; ++w) t += red[w];
        partial[(size_t)g * Co + co] = t;
    }
}
// ---- host planning -----------------------------------------------------------------------------------------
enum { W_128x288 = 0, W_64x320, W_32x160, W_16x32, W_64x64, W_64x128, W_64x192, W_128x96, W_32x64 };

struct WPlan {
    int cfg, MF, MTB, NTB, WK, threads;
    WgradParams p;
    size_t lds_bytes;
    int nsplit, nslots;
    unsigned gy, gz;
    int bias_groups;
    int pc;  // producer/consumer kernel variant (two LDS buffers)
    size_t part_floats, bias_floats, tmp_floats;
};

constexpr size_t kLdsBudget = 78 * 1024;  // two workgroups per CU: the second one computes while this one stages

int make_wplan(const char* who, const cnn_conv2d_desc* d, WPlan* pl) {
    WgradParams& p = pl->p;
    p = WgradParams();
    p.B = d->B; p.Ci = d->Ci; p.H = d->H; p.W = d->W; p.Co = d->Co; p.k = d->k; p.s = d->s; p.pad = d->pad;
    p.Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, d->pad);
    p.Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, d->pad);
    CNN_REQUIRE(p.Ho > 0 && p.Wo > 0, "%s: empty output", who);
    p.Ntot = d->Ci * d->k * d->k;
    const long long kpix = (long long)p.B * p.Ho * p.Wo;  // the GEMM's K
    // short-K problems (the reference net's last layers): narrower tiles mean fewer / smaller split-K slabs, and an N tile
    // that does not divide Ci*k*k leaves a spare column for the fused bias gradient (measured on conv_layer_3 / _4)
    if (p.Co > 64 && kpix <= 16384) { pl->cfg = W_64x320; pl->MF = 32; pl->MTB = 64; pl->NTB = 320; pl->WK = 1; pl->threads = 256; }
    else if (p.Co > 64) { pl->cfg = W_128x288; pl->MF = 32; pl->MTB = 128; pl->NTB = 288; pl->WK = 1; pl->threads = 256; }
    else if (p.Co > 32 && kpix <= 65536 && p.Ntot <= 384) { pl->cfg = W_64x192; pl->MF = 32; pl->MTB = 64; pl->NTB = 192; pl->WK = 1; pl->threads = 256; }
    else if (p.Co > 32) { pl->cfg = W_64x320; pl->MF = 32; pl->MTB = 64; pl->NTB = 320; pl->WK = 1; pl->threads = 256; }
    else if (p.Co > 16) { pl->cfg = W_32x160; pl->MF = 32; pl->MTB = 32; pl->NTB = 160; pl->WK = 4; pl->threads = 256; }
    else { pl->cfg = W_16x32; pl->MF = 16; pl->MTB = 16; pl->NTB = 32; pl->WK = 4; pl->threads = 256; }
    if (const OptVal ov = CNN_OPT_VAL("WGRAD_CFG")) {  // tuning override
        const int c = atoi(ov);
        if (c == W_64x64 && p.Co > 32) { pl->cfg = c; pl->MF = 32; pl->MTB = 64; pl->NTB = 64; pl->WK = 1; }
        if (c == W_64x128 && p.Co > 32) { pl->cfg = c; pl->MF = 32; pl->MTB = 64; pl->NTB = 128; pl->WK = 1; }
        if (c == W_64x192 && p.Co > 32) { pl->cfg = c; pl->MF = 32; pl->MTB = 64; pl->NTB = 192; pl->WK = 1; }
        if (c == W_128x96 && p.Co > 64) { pl->cfg = c; pl->MF = 32; pl->MTB = 128; pl->NTB = 96; pl->WK = 1; }
        if (c == W_32x64 && p.Co > 16) { pl->cfg = c; pl->MF = 32; pl->MTB = 32; pl->NTB = 64; pl->WK = 2; }
        if (c == W_64x320 && p.Co > 32) { pl->cfg = c; pl->MF = 32; pl->MTB = 64; pl->NTB = 320; pl->WK = 1; }
        if (c == W_32x160 && p.Co > 16) { pl->cfg = c; pl->MF = 32; pl->MTB = 32; pl->NTB = 160; pl->WK = 4; }
    }
    const int kstep = pl->MF == 32 ? 2 : 4;
    // wave specialisation pays when the MFMA phase is long enough to hide a chunk's staging (the 128-channel tile)
    // (measured on MI355X: no gain -- with half the waves loading, the latency-bound staging takes twice as long -- so the
    //  variant stays opt-in for experiments: CNN_AMD_WGRAD_PC=1)
    pl->pc = CNN_OPT_INT("WGRAD_PC", 0);
    const int nbuf = pl->pc ? 2 : 1;
    const size_t lds_budget = CNN_OPT_SET("WGRAD_LDS") ? (size_t)CNN_OPT_INT("WGRAD_LDS", 0) * 1024
                                                          : (pl->pc ? 150 * 1024 : kLdsBudget);
    const int kk2 = d->k * d->k;
    p.CIB = (pl->NTB + kk2 - 2) / kk2 + 1;
    if (p.CIB > p.Ci) p.CIB = p.Ci;
    // pick the chunk: as many whole output rows as fit the LDS budget (a multiple of WK), else a slice of one row
    auto drow_for = [&](int R, int QCP) {
        int drow = R * QCP;
        if (pl->MF == 32) drow |= 1;                     // odd stride: 32 channels hit 32 banks
        else drow += ((2 - drow % 32) + 32) % 32;        // == 2 (mod 32): 16 channels x 2 k-lanes per half-wave
        return drow;
    };
    auto lds_for = [&](int R, int QC) {
        const int QCP = (QC + kstep - 1) / kstep * kstep;
        const int XR = (R - 1) * d->s + d->k, LWc = (QCP - 1) * d->s + d->k;
        return nbuf * ((size_t)pl->MTB * drow_for(R, QCP) + (size_t)p.CIB * (XR * LWc + 1)) * sizeof(float);
    };
    int R = 1, QC = p.Wo;
    if (lds_for(1, p.Wo) <= lds_budget) {
        const int target_pixels = 512;  // enough k-steps per barrier; more only costs LDS
        while (R < p.Ho && (R + pl->WK) * p.Wo <= target_pixels + p.Wo && lds_for(R + 1, p.Wo) <= lds_budget) ++R;
        if (pl->WK > 1 && R >= pl->WK) R = R / pl->WK * pl->WK;
    } else {
        while (QC > kstep && lds_for(1, QC) > lds_budget) QC = (QC + 1) / 2;
        CNN_REQUIRE(lds_for(1, QC) <= lds_budget, "%s: Ci*k*k tile does not fit LDS (k=%d)", who, d->k);
    }
    p.R = R; p.QC = QC; p.QCP = (QC + kstep - 1) / kstep * kstep;
    p.nrc = (p.Ho + R - 1) / R; p.ncc = (p.Wo + QC - 1) / QC;
    p.DROW = drow_for(R, p.QCP);
    p.XR = (R - 1) * d->s + d->k; p.LWc = (p.QCP - 1) * d->s + d->k; p.CHS = p.XR * p.LWc + 1;
    pl->lds_bytes = nbuf * ((size_t)pl->MTB * p.DROW + (size_t)p.CIB * p.CHS) * sizeof(float);
    CNN_REQUIRE(pl->lds_bytes <= 160 * 1024, "%s: LDS plan %zu B too large", who, pl->lds_bytes);
    p.rwd_shift = 0;
    while ((1 << p.rwd_shift) < p.QCP && p.rwd_shift < 6) ++p.rwd_shift;
    p.rwx_shift = 0;
    while ((1 << p.rwx_shift) < p.LWc && p.rwx_shift < 6) ++p.rwx_shift;
    p.dbg = CNN_MEASURE_INT("DBG", 0);
    p.chunks_total = (long long)p.B * p.nrc * p.ncc;
    pl->gy = (unsigned)((p.Ntot + pl->NTB - 1) / pl->NTB);
    pl->gz = (unsigned)((p.Co + pl->MTB - 1) / pl->MTB);
    int bpc = (int)((160 * 1024) / pl->lds_bytes);
    if (bpc < 1) bpc = 1;
    if (bpc > 8) bpc = 8;
    long long want = (long long)num_cus() * bpc / ((long long)pl->gy * pl->gz);
    if (want < 1) want = 1;
    if (want > p.chunks_total) want = p.chunks_total;
    p.chunks_per_split = (int)((p.chunks_total + want - 1) / want);
    pl->nsplit = (int)((p.chunks_total + p.chunks_per_split - 1) / p.chunks_per_split);
    pl->nslots = pl->nsplit * pl->WK;
    p.fuse_bias = (p.Ntot % pl->NTB) != 0 ? 1 : 0;  // a spare padded column exists in the last N block
    p.pitch = p.Ntot + p.fuse_bias;
    pl->gy = (unsigned)((p.pitch + pl->NTB - 1) / pl->NTB);
    pl->part_floats = (size_t)pl->nslots * p.Co * p.pitch;
    pl->bias_groups = p.B < 64 ? p.B : 64;
    pl->bias_floats = (size_t)pl->bias_groups * p.Co;
    pl->tmp_floats = (size_t)((pl->nslots + 63) / 64) * p.Co * p.pitch;
    return CNN_AMD_OK;
}

#define CONV_TAG(d) "B%d Ci%d %dx%d Co%d k%d s%d p%d", (d)->B, (d)->Ci, (d)->H, (d)->W, (d)->Co, (d)->k, (d)->s, (d)->pad

template <int MF, int MA, int NB, int WM, int WN, int WK, bool NARROW, bool PC>
int launch_w2(const WPlan& pl, hipStream_t s, const cnn_conv2d_desc* d, int* nsplit_used) {
    auto kern = wgrad_kernel<MF, MA, NB, WM, WN, WK, NARROW, PC>;
    constexpr int kThreads = 64 * WM * WN * WK * (PC ? 2 : 1);
    static DeviceOnce attr_once;
    if (pl.lds_bytes > 48 * 1024 && attr_once.needed()) {
        CNN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_once.mark();
    }
    // split count = exactly the number of workgroups the chip holds at once (registers + LDS): a larger grid would run a
    // second, partly empty round.  The plan's nsplit is the upper bound the workspace was sized for.
    static thread_local size_t occ_lds = (size_t)-1;
    static thread_local int occ = 0;
    if (occ_lds != pl.lds_bytes) {
        int n = 0;
        CNN_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, kThreads, pl.lds_bytes));
        occ = n < 1 ? 1 : n;
        occ_lds = pl.lds_bytes;
    }
    WgradParams q = pl.p;
    long long want = (long long)occ * num_cus() / ((long long)pl.gy * pl.gz);
    if (want < 1) want = 1;
    if (want > pl.nsplit) want = pl.nsplit;
    if (want > q.chunks_total) want = q.chunks_total;
    q.chunks_per_split = (int)((q.chunks_total + want - 1) / want);
    const int nsplit = (int)((q.chunks_total + q.chunks_per_split - 1) / q.chunks_per_split);
    *nsplit_used = nsplit;
    char name[96];
    snprintf(name, sizeof(name), "wgrad_kernel<%d,%d,%d,%d,%d,%d>%s", MF, MA, NB, WM, WN, WK, PC ? "/pc" : "");
    CNN_KLAUNCH(s, name, (kern<<<dim3(nsplit, pl.gy, pl.gz), kThreads, pl.lds_bytes, s>>>(q)), CONV_TAG(d));
    return CNN_AMD_OK;
}

template <int MF, int MA, int NB, int WM, int WN, int WK>
int launch_w(const WPlan& pl, hipStream_t s, const cnn_conv2d_desc* d, int* nsplit_used) {
    const bool narrow = pl.p.QCP <= 64 && pl.p.LWc <= 64;
    if (pl.pc) return narrow ? launch_w2<MF, MA, NB, WM, WN, WK, true, true>(pl, s, d, nsplit_used)
                             : launch_w2<MF, MA, NB, WM, WN, WK, false, true>(pl, s, d, nsplit_used);
    return narrow ? launch_w2<MF, MA, NB, WM, WN, WK, true, false>(pl, s, d, nsplit_used)
                  : launch_w2<MF, MA, NB, WM, WN, WK, false, false>(pl, s, d, nsplit_used);
}

// the register-direct kernel (conv_wgrad_rd.hip) takes every geometry it covers except the thin first layer (packed VALU
// kernel); CNN_AMD_WGRAD_RD=0 sends those layers through the LDS-staged kernel below instead (A/B measurements)
// The thin first layer (3 -> 16 channels) stays on the packed VALU kernel unless CNN_AMD_WG_POOL_RD=1: the MFMA kernel is
// faster on a materialised delta (102 vs 123 us at batch 256) but slower from the pooled domain (196 vs 145 us: three
// 16-byte loads per window triple from 32 different channel planes per wave), and that is the variant the train step uses;
// both variants switch together so that the fused and unfused paths keep identical summation orders.
bool first_layer_rd() {
    const OptVal e = CNN_OPT_VAL("WG_POOL_RD");
    return e && atoi(e) != 0;
}
bool rd_wanted(const cnn_conv2d_desc* d) {
    if (direct_wgrad_slots(d) > 0 && !first_layer_rd()) return false;
    const OptVal e = CNN_OPT_VAL("WGRAD_RD");
    if (e && atoi(e) == 0) return false;
    return wgrad_rd_slots(d) > 0;
}

int check_desc(const char* who, const cnn_conv2d_desc* d) {
    CNN_REQUIRE(d != nullptr, "%s: desc is null", who);
    CNN_REQUIRE(d->B > 0 && d->Ci > 0 && d->H > 0 && d->W > 0 && d->Co > 0 && d->k > 0 && d->s > 0 && d->pad >= 0,
                "%s: bad desc B=%d Ci=%d H=%d W=%d Co=%d k=%d s=%d pad=%d", who, d->B, d->Ci, d->H, d->W, d->Co, d->k,
                d->s, d->pad);
    CNN_REQUIRE(d->H + 2 * d->pad >= d->k && d->W + 2 * d->pad >= d->k, "%s: kernel %d larger than padded input", who,
                d->k);
    CNN_REQUIRE((d->flags & ~CNN_CONV2D_POOL_MASK_PACKED) == 0, "%s: unknown desc flags 0x%x", who, (unsigned)d->flags);
    return CNN_AMD_OK;
}

}  // namespace

namespace cnn_amd {
// conv_backward.hip: record (true) / launch immediately (false) the final slab reductions of this thread's weight gradients
void wgrad_defer_reduce(bool on) {
    const bool off = (CNN_OPT_SET("NO_DEFER_REDUCE") && CNN_OPT_INT("NO_DEFER_REDUCE", 0) != 0);
    pending_reduces().defer = on && !off;
}
int wgrad_flush_reduces(hipStream_t s) { return flush_reduces(s); }
}  // namespace cnn_amd

extern "C" {

size_t cnn_conv2d_workspace_bytes(const cnn_conv2d_desc* d) {
    if (check_desc("cnn_conv2d_workspace_bytes", d)) return 0;
    static thread_local DescMemo memo;
    size_t known = 0;
    if (memo.find(d, &known)) return known;
    WPlan pl;
    size_t wg = 0;
    if (make_wplan("cnn_conv2d_workspace_bytes", d, &pl) == CNN_AMD_OK) wg = pl.part_floats + pl.bias_floats + pl.tmp_floats;
    const size_t ig = igemm_workspace_floats(d);
    const int ds = direct_wgrad_slots(d);
    const size_t dw = ds ? (size_t)(ds + (ds + 63) / 64) * 16 * 28 : 0;  // slabs + stage-1 scratch of the direct kernel
    size_t m = wg > ig ? wg : ig;
    if (dw > m) m = dw;
    const int rs = wgrad_rd_slots(d);
    const size_t rw = rs ? (size_t)(rs + (rs + 63) / 64) * d->Co * (d->Ci * 9 + 1) : 0;
    if (rw > m) m = rw;
    const int rps = wgrad_rd_pooled_slots(d);
    const size_t rpw = rps ? (size_t)(rps + (rps + 63) / 64) * d->Co * (d->Ci * 9 + 1) : 0;
    if (rpw > m) m = rpw;
    const int sts = stem_wgrad_slots(d);
    const size_t stw = sts ? (size_t)(sts + (sts + 63) / 64) * d->Co * 148 : 0;
    if (stw > m) m = stw;
    const int oss = os_wgrad_slots(d);
    const size_t osw = oss ? (size_t)(oss + (oss + 63) / 64) * d->Co * (d->Ci * 9 + 1) : 0;
    if (osw > m) m = osw;
    const int sps = sp_wgrad_slots(d);
    const size_t spw = sps ? (size_t)(sps + (sps + 63) / 64) * d->Co * (d->Ci * 9 + 1) : 0;
    if (spw > m) m = spw;
    const int c1s = c11_wgrad_slots(d);
    const size_t c1w = c1s ? (size_t)(c1s + (c1s + 63) / 64) * d->Co * (d->Ci + 1) : 0;
    if (c1w > m) m = c1w;
    memo.put(d, (m + 64) * sizeof(float));
    return (m + 64) * sizeof(float);
}

/* weight / bias gradient of the first block when its convolution delta exists only as (dpool, mask, pooled) */
int cnn_conv2d_backward_weight_pooled2(const cnn_conv2d_desc* d, const float* x, const float* dpool, const int32_t* mask,
                                       const float* pooled, float* gw, float* gb, float divisor, void* ws, size_t ws_bytes,
                                       void* stream) {
    if (int rc = check_desc("cnn_conv2d_backward_weight_pooled2", d)) return rc;
    CNN_REQUIRE(x && dpool && mask && gw, "cnn_conv2d_backward_weight_pooled2: null pointer");
    CNN_REQUIRE(divisor != 0.f, "cnn_conv2d_backward_weight_pooled2: divisor is 0");
    const int ds = direct_wgrad_slots(d);
    CNN_REQUIRE(ds > 0 && direct_conv_pool_supported(d), "cnn_conv2d_backward_weight_pooled2: geometry not covered");
    const size_t n = 16 * 28, need_d = (size_t)(ds + (ds + 63) / 64) * n * sizeof(float);
    CNN_REQUIRE(ws != nullptr && ws_bytes >= need_d, "cnn_conv2d_backward_weight_pooled2: workspace too small (%zu < %zu bytes)", ws_bytes,
                need_d);
    hipStream_t sd = as_stream(stream);
    // MFMA register-direct kernel (CNN_AMD_WG_POOL_RD=0: the packed VALU kernel): same slab layout, [16][27 | 1]
    const int rs = (first_layer_rd() && !(d->flags & CNN_CONV2D_POOL_MASK_PACKED)) ? wgrad_rd_pooled_slots(d) : 0;  // (packed mask: window kernel only)
    if (rs > 0 && ws_bytes >= (size_t)(rs + (rs + 63) / 64) * n * sizeof(float)) {
        if (int rc = wgrad_rd_launch_pooled(d, x, dpool, mask, pooled, (float*)ws, sd)) return rc;
        char tagr[160];
        snprintf(tagr, sizeof(tagr), CONV_TAG(d));
        return reduce_slabs(sd, (const float*)ws, rs, n, (float*)ws + (size_t)rs * n, gw, divisor, tagr, 27, gb);
    }
    if (int rc = direct_conv_wgrad_pooled(d, x, dpool, mask, pooled, (float*)ws, sd)) return rc;
    char tagd[160];
    snprintf(tagd, sizeof(tagd), CONV_TAG(d));
    return reduce_slabs(sd, (const float*)ws, ds, n, (float*)ws + (size_t)ds * n, gw, divisor, tagd, 27, gb);
}

int cnn_conv2d_backward_weight_pooled2_sgd(const cnn_conv2d_desc* d, const float* x, const float* dpool, const int32_t* mask,
                                           const float* pooled, float* gw, float* gb, float divisor, float* w, float* bias, float lr,
                                           float grad_scale, void* fwd_prepared, void* dgrad_prepared, void* ws, size_t ws_bytes,
                                           void* stream) {
    return cnn_conv2d_backward_weight_pooled2_sgd_keep(d, x, dpool, mask, pooled, gw, gb, divisor, w, bias, lr, grad_scale, fwd_prepared,
                                                       dgrad_prepared, nullptr, nullptr, ws, ws_bytes, stream);
}

int cnn_conv2d_backward_weight_pooled2_sgd_keep(const cnn_conv2d_desc* d, const float* x, const float* dpool, const int32_t* mask,
                                                const float* pooled, float* gw, float* gb, float divisor, float* w, float* bias, float lr,
                                                float grad_scale, void* fwd_prepared, void* dgrad_prepared, float* w_previous,
                                                float* bias_previous, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_desc("cnn_conv2d_backward_weight_pooled2_sgd", d)) return rc;
    CNN_REQUIRE(x && dpool && mask && gw && gb && w && bias, "cnn_conv2d_backward_weight_pooled2_sgd: null pointer");
    CNN_REQUIRE(divisor != 0.f, "cnn_conv2d_backward_weight_pooled2_sgd: divisor is 0");
    const int ds = direct_wgrad_slots(d);
    CNN_REQUIRE(ds > 0 && direct_conv_pool_supported(d), "cnn_conv2d_backward_weight_pooled2_sgd: geometry not covered");
    const size_t n = 16 * 28;
    hipStream_t sd = as_stream(stream);
    int slots = ds;
    const int rs = (first_layer_rd() && !(d->flags & CNN_CONV2D_POOL_MASK_PACKED)) ? wgrad_rd_pooled_slots(d) : 0;  // (packed mask: window kernel only)
    if (rs > 0 && ws != nullptr && ws_bytes >= (size_t)rs * n * sizeof(float)) {
        slots = rs;
        if (int rc = wgrad_rd_launch_pooled(d, x, dpool, mask, pooled, (float*)ws, sd)) return rc;
    } else {
        CNN_REQUIRE(ws != nullptr && ws_bytes >= (size_t)ds * n * sizeof(float),
                    "cnn_conv2d_backward_weight_pooled2_sgd: workspace too small (%zu < %zu bytes)", ws_bytes, (size_t)ds * n * sizeof(float));
        if (int rc = direct_conv_wgrad_pooled(d, x, dpool, mask, pooled, (float*)ws, sd)) return rc;
    }
    // (more than 512 slabs would go through a two-stage reduction in the unfused path: a different summation order)
    CNN_REQUIRE(slots <= 512, "cnn_conv2d_backward_weight_pooled2_sgd: %d slabs", slots);
    return direct_first_layer_finish(d, (const float*)ws, slots, divisor, gw, gb, w, bias, lr, grad_scale, fwd_prepared, dgrad_prepared,
                                     w_previous, bias_previous, sd);
}

int cnn_amd_flush_reduces(void* stream) { return flush_reduces(as_stream(stream)); }

int cnn_conv2d_backward_weight(const cnn_conv2d_desc* d, const float* x, const float* dy, float* gw, float* gb,
                               float divisor, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_desc("cnn_conv2d_backward_weight", d)) return rc;
    CNN_REQUIRE(x && dy && gw, "cnn_conv2d_backward_weight: null pointer");
    CNN_REQUIRE(divisor != 0.f, "cnn_conv2d_backward_weight: divisor is 0");
    CNN_REQUIRE(ws != nullptr, "cnn_conv2d_backward_weight: workspace is null");
    if (const int ds = rd_wanted(d) ? 0 : direct_wgrad_slots(d)) {
        const size_t n = 16 * 28, need_d = (size_t)(ds + (ds + 63) / 64) * n * sizeof(float);
        if (ws_bytes >= need_d) {
            hipStream_t sd = as_stream(stream);
            if (int rc = direct_conv_wgrad(d, x, dy, (float*)ws, sd)) return rc;
            char tagd[160];
            snprintf(tagd, sizeof(tagd), CONV_TAG(d));
            float gb_dummy_unused = 0.f;
            (void)gb_dummy_unused;
            return reduce_slabs(sd, (const float*)ws, ds, n, (float*)ws + (size_t)ds * n, gw, divisor, tagd, 27, gb);
        }
    }
    if (const int c1s = c11_wgrad_slots(d)) {
        const size_t n = (size_t)d->Co * (d->Ci + 1), need_c = (size_t)(c1s + (c1s + 63) / 64) * n * sizeof(float);
        if (ws_bytes >= need_c) {
            hipStream_t sc = as_stream(stream);
            if (int rc = c11_wgrad_launch(d, x, dy, (float*)ws, sc)) return rc;
            char tagc[160];
            snprintf(tagc, sizeof(tagc), CONV_TAG(d));
            return reduce_slabs(sc, (const float*)ws, c1s, n, (float*)ws + (size_t)c1s * n, gw, divisor, tagc, d->Ci, gb);
        }
    }
    // (the LDS-staged small-plane kernels first: for the reference net's stride-2 shapes they apply only when asked for, WGRAD_SP2=2)
    if (const int sps = sp_wgrad_slots(d)) {
        const size_t n = (size_t)d->Co * (d->Ci * 9 + 1), need_p = (size_t)(sps + (sps + 63) / 64) * n * sizeof(float);
        if (ws_bytes >= need_p) {
            hipStream_t sp = as_stream(stream);
            if (int rc = sp_wgrad_launch(d, x, dy, (float*)ws, sp)) return rc;
            char tagp[160];
            snprintf(tagp, sizeof(tagp), CONV_TAG(d));
            return reduce_slabs(sp, (const float*)ws, sps, n, (float*)ws + (size_t)sps * n, gw, divisor, tagp, d->Ci * 9, gb);
        }
    }
    if (const int oss = os_wgrad_slots(d)) {
        const size_t n = (size_t)d->Co * (d->Ci * 9 + 1), need_o = (size_t)(oss + (oss + 63) / 64) * n * sizeof(float);
        if (ws_bytes >= need_o) {
            hipStream_t so = as_stream(stream);
            if (int rc = os_wgrad_launch(d, x, dy, (float*)ws, so)) return rc;
            char tago[160];
            snprintf(tago, sizeof(tago), CONV_TAG(d));
            return reduce_slabs(so, (const float*)ws, oss, n, (float*)ws + (size_t)oss * n, gw, divisor, tago, d->Ci * 9, gb);
        }
    }
    if (rd_wanted(d)) {
        const int rs = wgrad_rd_slots(d);
        const size_t n = (size_t)d->Co * (d->Ci * 9 + 1), need_r = (size_t)(rs + (rs + 63) / 64) * n * sizeof(float);
        if (rs > 0 && ws_bytes >= need_r) {
            hipStream_t sr = as_stream(stream);
            if (int rc = wgrad_rd_launch(d, x, dy, (float*)ws, sr)) return rc;
            char tagr[160];
            snprintf(tagr, sizeof(tagr), CONV_TAG(d));
            return reduce_slabs(sr, (const float*)ws, rs, n, (float*)ws + (size_t)rs * n, gw, divisor, tagr, d->Ci * 9, gb);
        }
    }
    if (const int sts = stem_wgrad_slots(d)) {
        const size_t n = (size_t)d->Co * 148, need_s = (size_t)(sts + (sts + 63) / 64) * n * sizeof(float);
        if (ws_bytes >= need_s) {
            hipStream_t ss = as_stream(stream);
            if (int rc = stem_wgrad_launch(d, x, dy, (float*)ws, ss)) return rc;
            char tags[160];
            snprintf(tags, sizeof(tags), CONV_TAG(d));
            return reduce_slabs(ss, (const float*)ws, sts, n, (float*)ws + (size_t)sts * n, gw, divisor, tags, 147, gb);
        }
    }
    WPlan pl;
    if (int rc = make_wplan("cnn_conv2d_backward_weight", d, &pl)) return rc;
    const size_t need = (pl.part_floats + pl.bias_floats + pl.tmp_floats) * sizeof(float);
    if (ws_bytes < need)
        return fail(CNN_AMD_E_WORKSPACE, "cnn_conv2d_backward_weight: workspace %zu B < %zu B", ws_bytes, need);
    hipStream_t s = as_stream(stream);
    pl.p.x = x; pl.p.dy = dy; pl.p.part = (float*)ws;
    int rc, nsplit_used = pl.nsplit;
    switch (pl.cfg) {
        case W_128x288: rc = launch_w<32, 1, 9, 4, 1, 1>(pl, s, d, &nsplit_used); break;
        case W_64x320: rc = launch_w<32, 1, 5, 2, 2, 1>(pl, s, d, &nsplit_used); break;
        case W_32x160: rc = launch_w<32, 1, 5, 1, 1, 4>(pl, s, d, &nsplit_used); break;
        case W_64x64: rc = launch_w<32, 1, 1, 2, 2, 1>(pl, s, d, &nsplit_used); break;
        case W_64x128: rc = launch_w<32, 1, 2, 2, 2, 1>(pl, s, d, &nsplit_used); break;
        case W_64x192: rc = launch_w<32, 1, 3, 2, 2, 1>(pl, s, d, &nsplit_used); break;
        case W_128x96: rc = launch_w<32, 1, 3, 4, 1, 1>(pl, s, d, &nsplit_used); break;
        case W_32x64: rc = launch_w<32, 1, 1, 1, 2, 2>(pl, s, d, &nsplit_used); break;
        default: rc = launch_w<16, 1, 2, 1, 1, 4>(pl, s, d, &nsplit_used); break;
    }
    if (rc) return rc;
    pl.nslots = nsplit_used * pl.WK;
    const size_t n = (size_t)pl.p.Co * pl.p.pitch;
    char tag[160];
    snprintf(tag, sizeof(tag), CONV_TAG(d));
    float* bpart = (float*)ws + pl.part_floats;
    float* tmp = bpart + pl.bias_floats;
    if (pl.p.fuse_bias) {  // weight and bias gradients come out of the same slabs
        if (int rc2 = reduce_slabs(s, (const float*)ws, pl.nslots, n, tmp, gw, divisor, tag, pl.p.Ntot, gb)) return rc2;
        return CNN_AMD_OK;
    }
    if (int rc2 = reduce_slabs(s, (const float*)ws, pl.nslots, n, tmp, gw, divisor, tag)) return rc2;
    if (gb) {
        CNN_KLAUNCH(s, "bias_grad_partial",
                    (bias_grad_partial<<<dim3(pl.p.Co, pl.bias_groups), kBiasBlock, 0, s>>>(dy, bpart, pl.p.B, pl.p.Co,
                                                                                           pl.p.Ho * pl.p.Wo, pl.bias_groups)),
                    CONV_TAG(d));
        if (int rc2 = reduce_slabs(s, bpart, pl.bias_groups, (size_t)pl.p.Co, tmp, gb, divisor, tag)) return rc2;
    }
    return CNN_AMD_OK;
}

}  // extern "C"
