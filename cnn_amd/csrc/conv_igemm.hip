// conv_igemm.hip -- Conv2D forward and data-gradient as ONE implicit-GEMM kernel on the gfx950 fp32 matrix
// cores (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate, 157.3 TFLOP/s).
//
//   acc[m][n] = sum_{c < C} sum_{(tr,tc) < (TR,TC)} A[(tr,tc)][c][m] * X[b][c][u*su + r0 + tr][v*su + c0 + tc]
//   n <-> (b,u,v) flattened over the whole batch, X reads outside the image are 0.
//
//   forward  (conv2d.cpp:69-92):  X = x, C = Ci, M = Co, taps = k x k, su = s, r0 = c0 = -pad,
//                                 A[t][ci][co] = w[co][ci][kx][ky];  epilogue adds bias, writes NCHW.
//   dgrad    (conv2d.cpp:168-199, re-expressed as a gather): X = dy, C = Co, grid = ceil(H/s) x ceil(W/s), su = 1,
//                                 M = s*s*Ci "virtual channels" (one per output-parity class (ph,pw) and ci),
//                                 taps = the <= ceil(k/s)^2 window of dy each class reads,
//                                 A[t][co][(ph,pw,ci)] = w[co][ci][kx][ky] for the tap that class uses, else 0;
//                                 epilogue scatters class (ph,pw) to dx[.., u*s+ph, v*s+pw] (every dx element is
//                                 written exactly once -> no memset, no atomics; uncovered rows/cols get 0).
//
// Data movement (per workgroup, per chunk of CK channels):
//   * the input ROWS the pixel tile needs are copied HBM -> LDS as whole NCHW rows (coalesced 4*XW-byte runs),
//     laid out [ck][lds_row][LW] with zero pad columns / zero rows where the window leaves the image;
//   * the filter slab for the chunk is one contiguous block (weights are pre-arranged by igemm_prep_weights into
//     [mblock][chunk][tap][ck][MT]) copied with 16-byte loads;
//   * MFMA B operands are ds_read_b32 gathers from the row image (im2col never exists in memory), A operands are
//     conflict-free ds_read_b32 of the slab.
// Roofline: MFMA-bound for the 64->128 112x112 shape (190 FLOP/B); HBM-bound for Ci = 3.
#include <cstdlib>

#include <map>
#include <mutex>

#include "common.h"

using namespace cnn_amd;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct IgemmParams {
    const float* X;
    const float* A;     // prepared weights
    const float* bias;  // [M] or nullptr
    float* Y;
    float* Y2;          // nullable. forward: second output relu(Y) (the ReLU layer behind the convolution, relu.cpp:25);
                        // dgrad: INPUT, the output of the ReLU layer in front -- dx is stored as (Y2 <= 0 ? 0 : dx), relu.cpp:38
    int B, C, XH, XW;   // input tensor
    int U, V;           // output grid per image
    int su;             // grid stride in input coordinates
    int TR, TC, r0, c0; // tap window and its origin relative to (u*su, v*su)
    int M;              // valid (virtual) output channels
    int nchunk;         // ceil(C / CK)
    int LW, padL;       // LDS row pitch (floats) and left zero pad
    int nrows_max;      // LDS rows per channel (upper bound over all tiles)
    int chs;            // LDS channel stride (floats)
    long long N;        // B*U*V
    int mode;           // 0 = forward NCHW store (+bias), 1 = dgrad parity scatter
    int s_out, c_out, OH, OW;  // mode 1: stride, real channel count Ci, dx height/width
    int ntiles;         // number of pixel tiles (grid.x)
    int rw_shift;       // log2(lanes per staged row): narrow rows share one wave-wide load
    int need_zero;      // the row image has pad columns / out-of-image rows -> zero it once
    int run_mode;       // DMA kernel: rows of a channel are one contiguous 16-byte-aligned run
    int row_tail;       // run_mode 3 with XW % 4 != 0: valid floats in the LAST 16-byte unit of an input row (0: rows are whole units)
    int dbg;            // ablation bits (CNN_AMD_DBG, tuning only): 1 no X DMA, 2 no A DMA, 4 no MFMA, 8 no stores
    int tc_inv;         // 65536 / TC + 1: t / TC == (t * tc_inv) >> 16 for the small tap indices used here
    int ncls;           // dgrad: number of output-parity classes with a tap mask below (0: every tap is used by every row)
    unsigned cls_mask[16];  // dgrad: bit t set <=> class cls = m / c_out has a filter tap at window position t
    int ksplit;         // DMA kernel: blockIdx.z = one of ksplit contiguous chunk ranges; partial sums go to Y + z * zstride (split_reduce adds them)
    long long zstride;  // floats per partial output tensor
};

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

template <int MF>
struct Acc;
template <>
struct Acc<32> {
    typedef f32x16 type;
    static constexpr int kRegs = 16;
    static constexpr int kStep = 2;
    __device__ static __forceinline__ type mfma(float a, float b, type c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    // C/D layout: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
    __device__ static __forceinline__ int row(int reg, int lh) { return (reg & 3) + 8 * (reg >> 2) + 4 * lh; }
};
template <>
struct Acc<16> {
    typedef f32x4 type;
    static constexpr int kRegs = 4;
    static constexpr int kStep = 4;
    __device__ static __forceinline__ type mfma(float a, float b, type c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // C/D layout: col = lane & 15, row = 4*(lane >> 4) + reg
    __device__ static __forceinline__ int row(int reg, int lh) { return 4 * lh + reg; }
};


// ---- pieces shared by the single-buffered and the DMA double-buffered kernels ---------------------------------

// Accumulators start at the bias (forward) or 0: the epilogue is then a pure store (no loads on the store path).
template <int MF, int MA, int NB>
__device__ __forceinline__ void init_acc(typename Acc<MF>::type (&acc)[MA][NB], const IgemmParams& p, int mbase_wave,
                                         int lh, bool with_bias = true) {
    using A_ = Acc<MF>;
#pragma unroll
    for (int ma = 0; ma < MA; ++ma) {
        float bv[A_::kRegs];
#pragma unroll
        for (int r = 0; r < A_::kRegs; ++r) {
            const int m = mbase_wave + ma * MF + A_::row(r, lh);
            bv[r] = (with_bias && p.bias != nullptr && m < p.M) ? p.bias[m] : 0.f;
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < A_::kRegs; ++r) acc[ma][nb][r] = bv[r];
    }
}

// One chunk of CK channels: for every tap, CK/KSTEP MFMA k-steps.  Software pipelined by one k-step: the LDS reads of
// step s+1 are issued before the MFMAs of step s, so the matrix pipe never waits on ds_read latency.
// A operands walk the slab linearly ([tap][ck][MT] is contiguous: +KSTEP*MT floats per step); B operands are gathers
// from the row image at pix_off + tap offset + channel offset.
template <int MF, int MA, int NB, int CK, int MT>
__device__ __forceinline__ void compute_chunk(const float* __restrict__ As, const float* __restrict__ Xs,
                                              const int (&pix_off)[NB], int a_lane,
                                              typename Acc<MF>::type (&acc)[MA][NB], const IgemmParams& p) {
    using A_ = Acc<MF>;
    constexpr int KSTEP = A_::kStep, S = CK / KSTEP;
    const float* a_ptr = As + a_lane;
    float a_cur[MA], b_cur[NB];
#pragma unroll
    for (int ma = 0; ma < MA; ++ma) a_cur[ma] = a_ptr[ma * MF];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) b_cur[nb] = Xs[pix_off[nb]];
    const int T = p.TR * p.TC;
    int tr = 0, tc = 0;
    for (int t = 0; t < T; ++t) {
        const int tap_off = tr * p.LW + tc;
        int ntc = tc + 1, ntr = tr;
        if (ntc == p.TC) { ntc = 0; ++ntr; }
        const int tap_next = (t + 1 < T) ? ntr * p.LW + ntc : 0;  // after the last tap: a harmless re-read
#pragma unroll
        for (int c2 = 0; c2 < S; ++c2) {
            float a_nxt[MA], b_nxt[NB];
            a_ptr += KSTEP * MT;  // (one step past the slab on the very last step: still inside this LDS buffer)
            const int boff = (c2 + 1 < S) ? tap_off + (c2 + 1) * KSTEP * p.chs : tap_next;
#pragma unroll
            for (int ma = 0; ma < MA; ++ma) a_nxt[ma] = a_ptr[ma * MF];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) b_nxt[nb] = Xs[pix_off[nb] + boff];
            // keep the reads of step s+1 ABOVE the MFMAs of step s (hipcc otherwise sinks them next to their use and
            // every MFMA pair waits a full LDS round trip)
#ifndef CNN_NO_PIN
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int ma = 0; ma < MA; ++ma)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[ma][nb] = A_::mfma(a_cur[ma], b_cur[nb], acc[ma][nb]);
#ifndef CNN_NO_PIN
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int ma = 0; ma < MA; ++ma) a_cur[ma] = a_nxt[ma];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) b_cur[nb] = b_nxt[nb];
        }
        tr = ntr;
        tc = ntc;
    }
}

// Epilogue.  forward: y[b][m][pixel] (each (reg, half-wave) writes 32 / 16 consecutive pixels of one channel);
// dgrad: virtual channel m = (ph*s+pw)*Ci + ci of grid pixel (u,v) -> dx[b][ci][u*s+ph][v*s+pw].
template <int MF, int MA, int NB, bool R2>
__device__ __forceinline__ void store_tile(const typename Acc<MF>::type (&acc)[MA][NB], const IgemmParams& p,
                                           long long n_first, long long n1, int mbase_wave, int li, int lh, float* const Yb) {  // Yb: p.Y, or this split-K range's partial tensor
    using A_ = Acc<MF>;
    const long long UV = (long long)p.U * p.V;
    const bool full_m = (mbase_wave + MA * MF <= p.M);  // wave-uniform: no per-row bounds checks in the common case
    if constexpr (R2 && MF == 16 && NB >= 7) {
        // wide tiles (16x16x4 blocks: 4 accumulator registers each), data gradient + ReLU': the group-of-kRegs batching below would be 26
        // dependent rounds of 4 mask loads per lane -- here the masks of TWO column blocks x MA row blocks (16 values) are requested at once
        if (p.mode != MODE_FWD) {
            constexpr int NBG = 2;
#pragma unroll
            for (int nb0 = 0; nb0 < NB; nb0 += NBG) {
                size_t off[NBG][MA][4];
                float mk[NBG][MA][4];
                unsigned okm = 0;
#pragma unroll
                for (int g = 0; g < NBG; ++g) {
                    if (nb0 + g < NB) {
                        const long long n = n_first + (nb0 + g) * MF + li;
                        const bool okn = n <= n1;
                        const long long nn = okn ? n : n1;
                        const int b = (int)(nn / UV);
                        const int rem = (int)(nn - b * UV);
                        const int u = rem / p.V, v = rem - u * p.V;
#pragma unroll
                        for (int ma = 0; ma < MA; ++ma)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int m = mbase_wave + ma * MF + A_::row(j, lh);
                                const int cls = m / p.c_out, ci = m - cls * p.c_out;
                                const int ph = cls / p.s_out;
                                const int h = u * p.s_out + ph, w = v * p.s_out + (cls - ph * p.s_out);
                                const bool ok = okn && (full_m || m < p.M) && h < p.OH && w < p.OW;
                                off[g][ma][j] = ok ? (((size_t)b * p.c_out + ci) * p.OH + h) * p.OW + w : 0;
                                okm |= (unsigned)ok << ((g * MA + ma) * 4 + j);
                            }
                    }
                }
#pragma unroll
                for (int g = 0; g < NBG; ++g)
                    if (nb0 + g < NB)
#pragma unroll
                        for (int ma = 0; ma < MA; ++ma)
#pragma unroll
                            for (int j = 0; j < 4; ++j) mk[g][ma][j] = p.Y2[off[g][ma][j]];
#pragma unroll
                for (int g = 0; g < NBG; ++g)
                    if (nb0 + g < NB)
#pragma unroll
                        for (int ma = 0; ma < MA; ++ma)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if ((okm >> ((g * MA + ma) * 4 + j)) & 1u)
                                    Yb[off[g][ma][j]] = (mk[g][ma][j] <= 0.f) ? 0.f : acc[ma][nb0 + g][j];
            }
            return;
        }
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const long long n = n_first + nb * MF + li;
        if (n > n1) continue;
        const int b = (int)(n / UV);
        const int rem = (int)(n - b * UV);
        if (p.mode == MODE_FWD) {
            // (round 4) Y == nullptr with R2: only the ReLU output is wanted (cnn_conv2d_relu_only_supported: nothing in a train step reads
            // the pre-activation tensor of a Conv2D -> ReLU pair; on the HBM-bound first layer of the VGG-shaped stack it is half of 3.3 GB)
            const bool wy = !R2 || Yb != nullptr;  // wave-uniform
            float* out = (wy ? Yb : p.Y2) + ((size_t)b * p.M + mbase_wave) * UV + rem;
            // fused ReLU::forward: the second output tensor has the same layout, so its address is the first one's plus a
            // wave-uniform distance (keeps the epilogue's register footprint that of the plain store)
            // (R2 is a compile-time variant: as a run-time branch the extra stores cost every kernel 32 VGPRs)
            const ptrdiff_t d2 = (R2 && wy) ? (p.Y2 - Yb) : 0;
#pragma unroll
            for (int ma = 0; ma < MA; ++ma) {
#pragma unroll
                for (int g = 0; g < A_::kRegs / 4; ++g) {
                    const int row0 = ma * MF + A_::row(4 * g, lh);  // rows row0 .. row0+3 are regs 4g .. 4g+3
                    float* o = out + (size_t)row0 * UV;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (full_m || mbase_wave + row0 + j < p.M) {
                            const float v = acc[ma][nb][4 * g + j];
                            if (wy) o[(size_t)j * UV] = v;
                            if constexpr (R2) o[(size_t)j * UV + d2] = v >= 0.f ? v : 0.f;
                        }
                }
            }
        } else {
            const int u = rem / p.V, v = rem - u * p.V;
#pragma unroll
            for (int ma = 0; ma < MA; ++ma) {
                if constexpr (R2) {
                    // round 4: every mask value of a tile is REQUESTED before the first one is used.  The per-element form below compiled
                    // to load -> wait -> store under a divergent branch, 16 dependent round trips per tile and lane (the ISA of the
                    // north-star tile: "L w S b" x 64): data gradient + ReLU' 2 672 vs 2 358 us plain on 64 -> 128 112x112 at batch 128.
                    // Lanes without a destination read element 0 (always valid) and store nothing.  Same values: bit-identical.
                    // (in groups of 16, or of 8 for the two-tile instances that live within 128 VGPRs, see igemm_dma_min_waves)
                    constexpr int G = (MA * NB <= 2 && A_::kRegs > 8) ? 8 : A_::kRegs;
#pragma unroll
                    for (int r0 = 0; r0 < A_::kRegs; r0 += G) {
                        size_t off[G];
                        float mk[G];
                        unsigned okm = 0;
#pragma unroll
                        for (int j = 0; j < G; ++j) {
                            const int m = mbase_wave + ma * MF + A_::row(r0 + j, lh);
                            const int cls = m / p.c_out, ci = m - cls * p.c_out;
                            const int ph = cls / p.s_out;
                            const int h = u * p.s_out + ph, w = v * p.s_out + (cls - ph * p.s_out);
                            const bool ok = (full_m || m < p.M) && h < p.OH && w < p.OW;
                            off[j] = ok ? (((size_t)b * p.c_out + ci) * p.OH + h) * p.OW + w : 0;
                            okm |= (unsigned)ok << j;
                        }
#pragma unroll
                        for (int j = 0; j < G; ++j) mk[j] = p.Y2[off[j]];
#pragma unroll
                        for (int j = 0; j < G; ++j)
                            if ((okm >> j) & 1u) Yb[off[j]] = (mk[j] <= 0.f) ? 0.f : acc[ma][nb][r0 + j];  // fused ReLU::backward of the layer in front
                    }
                } else {
#pragma unroll
                for (int r = 0; r < A_::kRegs; ++r) {
                    const int m = mbase_wave + ma * MF + A_::row(r, lh);
                    if (full_m || m < p.M) {
                        const int cls = m / p.c_out, ci = m - cls * p.c_out;
                        const int ph = cls / p.s_out;
                        const int h = u * p.s_out + ph, w = v * p.s_out + (cls - ph * p.s_out);
                        if (h < p.OH && w < p.OW) {
                            const size_t o = (((size_t)b * p.c_out + ci) * p.OH + h) * p.OW + w;
                            Yb[o] = acc[ma][nb][r];
                        }
                    }
                }
                }
            }
        }
    }
}

// MF: MFMA tile edge; MA x NB tiles per wave; WM x WN waves per workgroup; CK channels per LDS chunk.
// NARROW selects the row-staging scheme at compile time (keeping both in one kernel costs ~50 VGPRs of occupancy).
template <int MF, int MA, int NB, int WM, int WN, int CK, bool NARROW, bool R2>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu((R2 && MA * NB == 2 && WM * WN == 4) ? 3 : 1)))
void igemm_kernel(const IgemmParams p) {  // (round 4: the fused-ReLU epilogue of the 2-tile, 4-wave instances took 180 - 196 VGPRs against 148 - 164)
    using A_ = Acc<MF>;
    constexpr int NWAVES = WM * WN;
    constexpr int NT = 64 * NWAVES;
    constexpr int MT = MF * MA * WM;
    constexpr int NPIX = MF * NB * WN;
    constexpr int KSTEP = A_::kStep;
    static_assert(CK % KSTEP == 0, "chunk must hold whole MFMA k-steps");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = p.TR * p.TC;
    float* As = smem;                                   // [T][CK][MT]
    float* Xs = smem + (size_t)T * CK * MT;             // [CK][chs]
    int* rowsrc = (int*)(Xs + (size_t)CK * p.chs);      // [nrows_max]: (b*C*XH + xrow) or -1

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & (MF - 1);
    const int lh = lane / MF;  // k index inside one MFMA step

    // XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8), so give each XCD a
    // contiguous run of pixel tiles -> neighbouring tiles (shared halo rows) hit the same L2.
    int tile;
    {
        const int nt = p.ntiles, id = blockIdx.x;
        const int q = nt / kNumXCD, r = nt % kNumXCD, xcd = id % kNumXCD, k = id / kNumXCD;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int mb = blockIdx.y;

    // pixel indices fit int32 (make_plan enforces N < 2^31), which keeps the divisions below cheap
    const int UV = p.U * p.V;
    const int n0 = tile * NPIX;
    const int n1 = (n0 + NPIX <= (int)p.N ? n0 + NPIX : (int)p.N) - 1;
    const int b0 = n0 / UV;
    const int u0 = (n0 - b0 * UV) / p.V;
    const int b1 = n1 / UV;
    const int u1 = (n1 - b1 * UV) / p.V;
    const int nseg = b1 - b0 + 1;
    const int nrows0 = ((nseg == 1 ? u1 : p.U - 1) - u0) * p.su + p.TR;
    const int full = (p.U - 1) * p.su + p.TR;
    const int nrows = (nseg == 1) ? nrows0 : nrows0 + (nseg - 2) * full + u1 * p.su + p.TR;

    // ---- one-time LDS setup: the row table, then zero ONLY what staging never writes (pad columns and rows
    //      outside the image); those stay 0 for every chunk because the row -> image mapping is chunk-invariant ----
    for (int r = tid; r < nrows; r += NT) {
        int b, xrow;
        if (r < nrows0) {
            b = b0;
            xrow = u0 * p.su + p.r0 + r;
        } else {
            const int rr = r - nrows0;
            b = b0 + 1 + rr / full;
            xrow = p.r0 + rr % full;
        }
        rowsrc[r] = (xrow >= 0 && xrow < p.XH) ? (b * p.C * p.XH + xrow) : -1;
    }
    if (p.need_zero) {
        __syncthreads();
        const int npad = p.LW - p.XW;
        for (int ck = 0; ck < CK; ++ck) {
            float* xch = Xs + ck * p.chs;
            for (int r = wave; r < nrows; r += NWAVES) {
                float* d = xch + r * p.LW;
                if (rowsrc[r] < 0) {
                    for (int col = lane; col < p.LW; col += 64) d[col] = 0.f;
                } else {
                    for (int j = lane; j < npad; j += 64) d[j < p.padL ? j : j + p.XW] = 0.f;
                }
            }
        }
    }

    // ---- per-lane pixel -> LDS offset of its window origin (B operand column = this lane's pixel) ----
    int pix_off[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        int n = n0 + (wn * NB + nb) * MF + li;
        if (n > n1) n = n1;  // clamp: results of padded lanes are never stored
        const int b = n / UV;
        const int rem = n - b * UV;
        const int u = rem / p.V, v = rem - u * p.V;
        const int lrow = (b == b0) ? (u - u0) * p.su : nrows0 + (b - b0 - 1) * full + u * p.su;
        pix_off[nb] = lrow * p.LW + v * p.su + p.c0 + p.padL + lh * p.chs;
    }
    const int a_lane = lh * MT + wm * MA * MF + li;

    typename A_::type acc[MA][NB];
    const int mbase_wave = mb * MT + wm * MA * MF;
    init_acc<MF, MA, NB>(acc, p, mbase_wave, lh);

    const float4* Ag = (const float4*)(p.A + (size_t)mb * p.nchunk * T * CK * MT);
    const int a_vec = T * CK * MT / 4;

    for (int cc = 0; cc < p.nchunk; ++cc) {
        __syncthreads();  // previous chunk's reads are done (and, first time, the zero fill / row table landed)
        // ---- stage the filter slab: one contiguous block; every thread has 8 16-byte loads in flight
        //      before the first LDS store ----
        if (!(p.dbg & 2)) {
            // (named registers instead of an array: hipcc leaves a 128-byte private array in scratch memory)
            const float4* src = Ag + (size_t)cc * a_vec;
            float4* dst = (float4*)As;
            for (int i0 = tid; i0 < a_vec; i0 += NT * 8) {
                float4 t0, t1, t2, t3, t4, t5, t6, t7;
#define CNN_LD(u) if (i0 + u * NT < a_vec) t##u = src[i0 + u * NT];
#define CNN_ST(u) if (i0 + u * NT < a_vec) dst[i0 + u * NT] = t##u;
                CNN_LD(0) CNN_LD(1) CNN_LD(2) CNN_LD(3) CNN_LD(4) CNN_LD(5) CNN_LD(6) CNN_LD(7)
                CNN_ST(0) CNN_ST(1) CNN_ST(2) CNN_ST(3) CNN_ST(4) CNN_ST(5) CNN_ST(6) CNN_ST(7)
#undef CNN_LD
#undef CNN_ST
            }
        }
        // ---- stage input rows: a wave covers 64 >> rw_shift rows per load instruction (lanes along the row),
        //      kUn such instructions are issued back to back before the first LDS store.  Used for rows wider than 64
        //      floats, where one channel already provides a full batch of loads. ----
        if constexpr (!NARROW) if (!(p.dbg & 1)) {
            constexpr int kUn = 4, kMaxC = 4;
            const int RW = 1 << p.rw_shift, RPI = 64 >> p.rw_shift;
            const int sub = lane >> p.rw_shift, col0 = lane & (RW - 1);
            for (int ck = 0; ck < CK; ++ck) {
                const int c = cc * CK + ck;
                float* xdst = Xs + ck * p.chs + p.padL;
                if (c < p.C) {
                    const float* gch = p.X + (size_t)c * p.XH * p.XW;
                    for (int cb = 0; cb < p.XW; cb += 64 * kMaxC) {
                        for (int rb = wave * RPI; rb < nrows; rb += NWAVES * RPI * kUn) {
                            float v[kUn][kMaxC];
                            int src[kUn];
#pragma unroll
                            for (int u = 0; u < kUn; ++u) {
                                const int r = rb + u * NWAVES * RPI + sub;
                                src[u] = (r < nrows) ? rowsrc[r] : -1;
                            }
#pragma unroll
                            for (int u = 0; u < kUn; ++u) {
                                const float* g = gch + (size_t)(src[u] < 0 ? 0 : src[u]) * p.XW;
#pragma unroll
                                for (int ci = 0; ci < kMaxC; ++ci) {
                                    const int col = cb + col0 + ci * 64;
                                    if ((ci == 0 || RW == 64) && src[u] >= 0 && col < p.XW) v[u][ci] = g[col];
                                }
                            }
#pragma unroll
                            for (int u = 0; u < kUn; ++u) {
                                const int r = rb + u * NWAVES * RPI + sub;
                                float* d = xdst + r * p.LW;
#pragma unroll
                                for (int ci = 0; ci < kMaxC; ++ci) {
                                    const int col = cb + col0 + ci * 64;
                                    if ((ci == 0 || RW == 64) && src[u] >= 0 && col < p.XW) d[col] = v[u][ci];
                                }
                            }
                        }
                    }
                } else if (cc == p.nchunk - 1) {  // channel padding of the last chunk: must be finite
                    for (int r = wave; r < nrows; r += NWAVES) {
                        float* d = xdst + r * p.LW;
                        for (int col = lane; col < p.XW; col += 64) d[col] = 0.f;
                    }
                }
            }
        }
        // ---- stage input rows (narrow images).  Unit = one wave-wide load = (channel, group of 64>>rw_shift rows, 64-column block);
        //      units of ALL channels of the chunk are dealt round-robin to the waves and kU of them are in flight per lane
        //      before the first LDS store, so a chunk costs one or two memory round trips, not one per channel/row. ----
        if constexpr (NARROW) if (!(p.dbg & 1)) {
            constexpr int kU = 16;
            const int RPI = 64 >> p.rw_shift;
            const int sub = lane >> p.rw_shift, col0 = lane & ((1 << p.rw_shift) - 1);
            const int ngroups = (nrows + RPI - 1) / RPI;
            const int ncb = (p.rw_shift == 6) ? (p.XW + 63) / 64 : 1;
            const int per_ch = ngroups * ncb;
            const int cvalid = (p.C - cc * CK < CK) ? p.C - cc * CK : CK;  // real channels in this chunk
            const int NU = cvalid * per_ch;
            const float* gchunk = p.X + (size_t)cc * CK * p.XH * p.XW;
            for (int u0 = wave; u0 < NU; u0 += NWAVES * kU) {
                float v[kU];
                int ldo[kU];  // LDS float offset of the element this lane stages in slot j, or -1
#pragma unroll
                for (int j = 0; j < kU; ++j) {
                    const int uid = u0 + j * NWAVES;  // wave-uniform
                    ldo[j] = -1;
                    if (uid < NU) {
                        const int ck = uid / per_ch, rem = uid - ck * per_ch;
                        const int rg = rem / ncb, cbi = rem - rg * ncb;
                        const int r = rg * RPI + sub, col = cbi * 64 + col0;
                        const int src = (r < nrows) ? rowsrc[r] : -1;
                        if (src >= 0 && col < p.XW) {
                            v[j] = gchunk[((size_t)src + (size_t)ck * p.XH) * p.XW + col];
                            ldo[j] = ck * p.chs + p.padL + r * p.LW + col;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < kU; ++j)
                    if (ldo[j] >= 0) Xs[ldo[j]] = v[j];
            }
            if (cvalid < CK) {  // channel padding of the last chunk: must be finite
                for (int ck = cvalid; ck < CK; ++ck)
                    for (int r = wave; r < nrows; r += NWAVES) {
                        float* d = Xs + ck * p.chs + p.padL + r * p.LW;
                        for (int col = lane; col < p.XW; col += 64) d[col] = 0.f;
                    }
            }
        }
        __syncthreads();
        if (!(p.dbg & 4)) compute_chunk<MF, MA, NB, CK, MT>(As, Xs, pix_off, a_lane, acc, p);
    }

    if (!(p.dbg & 8)) store_tile<MF, MA, NB, R2>(acc, p, n0 + wn * NB * MF, n1, mbase_wave, li, lh, p.Y);
}

// ---- double-buffered variant: HBM -> LDS by DMA (global_load_lds), one barrier per chunk -------------------------
// Same math and row image as igemm_kernel, specialised for the MFMA-bound shapes (32x32x2 tiles, 8 channels per
// chunk):
//   * chunk c+1's row image and filter slab are written straight into the second LDS buffer by
//     global_load_lds_dword / _dwordx4 (no VGPR round trip; EXEC-masked lanes do not write -- probed on MI355X,
//     tools/probes/glds_probe.cpp) while the MFMAs consume chunk c; __syncthreads() drains the DMA queue (vmcnt(0))
//     exactly where the next chunk is needed;
//   * when the image needs no padding and rows are 16-byte multiples, the rows of one channel form ONE contiguous run
//     in HBM and in LDS and are moved 1 KiB per wave-instruction;
//   * the filter slab is stored [tap][k-half][m][4 k-steps] ("A4"), so a lane fetches its A operands for a whole tap
//     (4 MFMA k-steps) with one conflict-free ds_read_b128; operands of tap t+1 are read before the 4*MA*NB MFMAs of
//     tap t are issued.
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;

//   * XM > 0 ("whole-image" staging, small images): instead of the rows a tile needs, the COMPLETE CK-channel block of
//     every image the tile touches is staged -- in NCHW that is ONE contiguous run of CK*XH*XW floats per image, so a
//     chunk costs a handful of DMA instructions with trivial addressing (row staging of 6..27-float rows is bound by
//     instruction issue, not by bytes).  LDS image = [image][ck][XH*XW]; there are no pad columns / zero rows: XM == 2
//     masks the B operands of taps that leave the image instead (per-lane row / column bit masks).
// round 4: the fused-ReLU epilogue (R2) of the whole-image M = 64 tiles (MA = 2, NB = 1, four waves: the 14x14 layers of the VGG- /
// ResNet-shaped stacks) cost 32 VGPRs more than the plain one -- 144 / 156 instead of 112 / 124: three workgroups per CU instead of four,
// fwd+relu 1 935 us against fwd 1 405 us on 512 -> 512 14x14 at batch 128 (tools/probes/relu_epilogue.py).  The main loop does not need
// them: these instances are held to four waves per SIMD (<= 128 VGPRs) and hipcc schedules the epilogue within that.
// (the bound is the occupancy class the PLAIN instance compiles to: whole-image tiles with up to four k-steps per tap 108 - 124 VGPRs =
// four waves per SIMD; with eight k-steps, and the row-image tiles, 148 - 168 = three)
template <int MA, int NB, int WM, int WN, int S, int XM, bool R2>
constexpr int igemm_dma_min_waves() {
    return (R2 && MA * NB == 2 && WM * WN == 4) ? (XM != 0 ? (S <= 4 ? 4 : 3) : (S <= 2 ? 3 : 1)) : 1;  // (row-image tiles with S > 2: 176 - 200 plain)
}
template <int MF, int MA, int NB, int WM, int WN, int S, int XM, bool R2>  // S = k-steps per tap = channels per chunk / KSTEP
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(igemm_dma_min_waves<MA, NB, WM, WN, S, XM, R2>())))
void igemm_dma_kernel(const IgemmParams p) {
    using A_ = Acc<MF>;
    constexpr int KSTEP = A_::kStep, CK = KSTEP * S;
    typedef float avec_t __attribute__((ext_vector_type(S)));
    constexpr int NWAVES = WM * WN;
    constexpr int NT = 64 * NWAVES;
    constexpr int MT = MF * MA * WM;
    constexpr int NPIX = MF * NB * WN;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = p.TR * p.TC;
    const int a_floats = T * CK * MT;
    // one {slab, row image} pair; chs % 4 == 0.  XM != 0: nrows_max holds the number of images a tile can touch
    const int buf_floats = a_floats + (XM != 0 ? p.nrows_max : 1) * CK * p.chs;
    int* rowsrc = (int*)(smem + 2 * (size_t)buf_floats);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & (MF - 1), lh = lane / MF;

    int tile;
    {
        const int nt = p.ntiles, id = blockIdx.x;
        const int q = nt / kNumXCD, r = nt % kNumXCD, xcd = id % kNumXCD, k = id / kNumXCD;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int mb = blockIdx.y;
    // all pixel indices fit int32 here (the host only picks this kernel when N < 2^31)
    const int UV = p.U * p.V;
    const int n0 = tile * NPIX;
    const int n1 = (n0 + NPIX <= (int)p.N ? n0 + NPIX : (int)p.N) - 1;
    const int b0 = n0 / UV;
    const int u0 = (n0 - b0 * UV) / p.V;
    const int b1 = n1 / UV;
    const int u1 = (n1 - b1 * UV) / p.V;
    const int nseg = b1 - b0 + 1;
    const int nrows0 = ((nseg == 1 ? u1 : p.U - 1) - u0) * p.su + p.TR;
    const int full = (p.U - 1) * p.su + p.TR;
    const int nrows = (nseg == 1) ? nrows0 : nrows0 + (nseg - 2) * full + u1 * p.su + p.TR;

    if constexpr (XM == 0) {
    for (int r = tid; r < nrows; r += NT) {
        int b, xrow;
        if (r < nrows0) {
            b = b0;
            xrow = u0 * p.su + p.r0 + r;
        } else {
            const int rr = r - nrows0;
            b = b0 + 1 + rr / full;
            xrow = p.r0 + rr % full;
        }
        rowsrc[r] = (xrow >= 0 && xrow < p.XH) ? (b * p.C * p.XH + xrow) : -1;
    }
    __syncthreads();
    }
    if (XM == 0 && p.need_zero) {  // pad columns and out-of-image rows of BOTH buffers: never touched by the DMA
        const int npad = p.LW - p.XW;
        for (int bi = 0; bi < 2; ++bi)
            for (int ck = 0; ck < CK; ++ck) {
                float* xch = smem + bi * buf_floats + a_floats + ck * p.chs;
                for (int r = wave; r < nrows; r += NWAVES) {
                    float* d = xch + r * p.LW;
                    if (rowsrc[r] < 0) {
                        for (int col = lane; col < p.LW; col += 64) d[col] = 0.f;
                    } else {
                        for (int j = lane; j < npad; j += 64) d[j < p.padL ? j : j + p.XW] = 0.f;
                    }
                }
            }
    }

    int pix_off[NB];
    unsigned rmask[NB], cmask[NB];  // XM == 2: bit t set <=> window row / column t of this lane's pixel is inside the image
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        int n = n0 + (wn * NB + nb) * MF + li;
        if (n > n1) n = n1;
        const int b = n / UV;
        const int rem = n - b * UV;
        const int u = rem / p.V, v = rem - u * p.V;
        if constexpr (XM == 0) {
            const int lrow = (b == b0) ? (u - u0) * p.su : nrows0 + (b - b0 - 1) * full + u * p.su;
            pix_off[nb] = lrow * p.LW + v * p.su + p.c0 + p.padL + lh * p.chs;
        } else {
            const int xr = u * p.su + p.r0, xc = v * p.su + p.c0;
            pix_off[nb] = (b - b0) * CK * p.chs + xr * p.XW + xc + lh * p.chs;
            rmask[nb] = cmask[nb] = 0;
            if constexpr (XM >= 2) {
                for (int t = 0; t < p.TR; ++t) rmask[nb] |= (unsigned)(xr + t >= 0 && xr + t < p.XH) << t;
                for (int t = 0; t < p.TC; ++t) cmask[nb] |= (unsigned)(xc + t >= 0 && xc + t < p.XW) << t;
            }
        }
    }
    (void)rmask; (void)cmask;
    const int a_lane = (lh * MT + wm * MA * MF + li) * S;  // slab layout: [tap][k-lane][m][S k-steps]

    typename A_::type acc[MA][NB];
    const int mbase_wave = mb * MT + wm * MA * MF;
    // split-K (round 4, small planes at batch 64: a 512 -> 512 7x7 layer has 32 wide tiles for 256 CUs): workgroup z of ksplit takes the
    // chunks [c_lo, c_hi); range 0 starts from the bias, the others from 0, each stores its own partial tensor
    const int zk = (int)blockIdx.z;
    const int c_lo = (int)((long long)p.nchunk * zk / p.ksplit), c_hi = (int)((long long)p.nchunk * (zk + 1) / p.ksplit);
    init_acc<MF, MA, NB>(acc, p, mbase_wave, lh, zk == 0);

    // taps each 32/16-row MFMA tile of this wave needs (wave-uniform); umask = their union
    unsigned tmask[MA], umask = 0;
#pragma unroll
    for (int ma = 0; ma < MA; ++ma) {
        unsigned m_ = (T >= 32) ? ~0u : ((1u << T) - 1u);
        if (XM == 3 && p.ncls > 0) {
            const int m_lo = mbase_wave + ma * MF;
            int m_hi = m_lo + MF - 1;
            if (m_hi > p.M - 1) m_hi = p.M - 1;
            m_ = 0;
            if (m_lo <= m_hi)
                for (int cls = m_lo / p.c_out; cls <= m_hi / p.c_out; ++cls) m_ |= p.cls_mask[cls];
        }
        tmask[ma] = __builtin_amdgcn_readfirstlane(m_);
        umask |= tmask[ma];
    }
    if (umask == 0) umask = 1;  // a wave entirely beyond M: keep the loop structure, its tiles are never stored

    const float* Ag = p.A + (size_t)mb * p.nchunk * a_floats;
    const int a_vec = a_floats / 4;

    // ---- this wave's share of a chunk's ROW-IMAGE DMA, decoded ONCE (what changes from chunk to chunk is only the channel base
    // cc*CK*XH*XW).  run_mode 1 / 2: every channel's rows of an image segment are one contiguous run.  run_mode 3 (padded layers
    // with XW % 4 == 0: every 3x3 layer of the VGG / ResNet-shaped stacks): the LDS rows have a 4-float left pad and a pitch that is a
    // multiple of 4, so a channel's row image is a sequence of 16-byte units of which unit 0 and the last one(s) of every row are pad
    // (zeroed once, never written again) -- ONE 16-byte DMA instruction covers 64 consecutive LDS units = several rows, each lane
    // fetching its own unit from HBM.  Before: one 4-byte DMA instruction per (channel, row), each behind an LDS read of the row
    // table and 64-bit address arithmetic -- 7 instructions and ~3 400 - 4 900 cycles per wave and chunk right behind the barrier,
    // where no wave has MFMAs in flight (the filter slab's 4.5 instructions: 800 cycles): 16 % of the 256->256 56x56 layer.
    constexpr int MAXD = 8;
    int d_lds[MAXD];        // float offset of the instruction's LDS destination inside the row image (wave-uniform)
    unsigned d_src[MAXD];   // this lane's source element relative to the chunk's first channel plane, ~0u: lane inactive
    unsigned d_tail = 0;    // bit i: this lane's unit of instruction i is the ragged LAST unit of an input row (row_tail != 0)
    int nd = 0;
    bool desc_ok = false;
    if (p.run_mode != 0 && (unsigned long long)p.B * p.C * p.XH * p.XW < (1ull << 32)) {
        desc_ok = true;
        auto record = [&](int lds_off, unsigned src, bool tail_unit = false) {
            if (nd < MAXD) {
#pragma unroll
                for (int i = 0; i < MAXD; ++i)
                    if (i == nd) {
                        d_lds[i] = lds_off;
                        d_src[i] = src;
                    }
                if (tail_unit) d_tail |= 1u << nd;
                ++nd;
            } else {
                desc_ok = false;
            }
        };
        if constexpr (XM != 0) {
            // whole images (small layers): the CK-channel block of image b0 + img is one contiguous run in HBM and in LDS
            const int HW = p.chs, unit = (p.run_mode == 1) ? 4 : 1;
            const int runu = CK * HW / unit, per_img = (runu + 63) / 64;
            for (int j = wave; j < nseg * per_img; j += NWAVES) {
                const int img = j / per_img, part = j - img * per_img;
                const int idx = part * 64 + lane;
                record(img * CK * HW + part * 64 * unit, idx < runu ? (unsigned)((size_t)(b0 + img) * p.C * HW) + (unsigned)idx * unit : ~0u);
            }
        } else if (p.run_mode == 3) {
            const int upr = p.LW >> 2, total = nrows * upr, per_ch = (total + 63) / 64, dcols = (p.XW + 3) >> 2, pl4 = p.padL >> 2;
            for (int j = wave; j < CK * per_ch; j += NWAVES) {
                const int ck = j / per_ch, part = j - ck * per_ch;
                const int u = part * 64 + lane;
                const int rl = u / upr, cu = u - rl * upr;
                int b, xrow;
                if (rl < nrows0) {
                    b = b0;
                    xrow = u0 * p.su + p.r0 + rl;
                } else {
                    const int rr = rl - nrows0;
                    b = b0 + 1 + rr / full;
                    xrow = p.r0 + rr % full;
                }
                const bool on = u < total && cu >= pl4 && cu < pl4 + dcols && xrow >= 0 && xrow < p.XH;
                const unsigned src = (unsigned)(((size_t)b * p.C * p.XH + xrow) * p.XW) + (unsigned)ck * (unsigned)(p.XH * p.XW) + (unsigned)(4 * (cu - pl4));
                record(ck * p.chs + part * 256, on ? src : ~0u, on && p.row_tail != 0 && cu == pl4 + dcols - 1);
            }
        } else {
            const int unit = (p.run_mode == 1) ? 4 : 1;
            int lrow0 = 0;
            for (int sg = 0; sg < nseg; ++sg) {
                const int nr = (sg == 0) ? nrows0 : ((sg == nseg - 1) ? u1 * p.su + p.TR : full);
                const int xrow0 = (sg == 0) ? u0 * p.su + p.r0 : p.r0;
                const int runu = nr * p.XW / unit, per_ch = (runu + 63) / 64;
                const unsigned gseg = (unsigned)(((size_t)(b0 + sg) * p.C * p.XH + xrow0) * p.XW);
                for (int j = wave; j < CK * per_ch; j += NWAVES) {
                    const int ck = j / per_ch, part = j - ck * per_ch;
                    const int idx = part * 64 + lane;
                    record(ck * p.chs + lrow0 * p.LW + part * 64 * unit,
                           idx < runu ? gseg + (unsigned)ck * (unsigned)(p.XH * p.XW) + (unsigned)idx * unit : ~0u);
                }
                lrow0 += nr;
            }
        }
        nd = __builtin_amdgcn_readfirstlane(nd);
    }

    // ---- DMA of one chunk into buffer bi, issued as one burst right after the barrier (measured: spreading the issue
    //      over the taps costs more in decode math than it hides) ----
    auto issue_dma = [&](int cc, int bi) {
        float* Abuf = smem + bi * buf_floats;
        float* Xbuf = Abuf + a_floats;
        const float* asrc = Ag + (size_t)cc * a_floats;
        if (!(p.dbg & 2)) {
#pragma nounroll
            for (int i = wave; i * 64 < a_vec; i += NWAVES) {
                const int idx = i * 64 + lane;
                if (idx < a_vec)
                    __builtin_amdgcn_global_load_lds((gbl_void_ptr)(asrc + (size_t)idx * 4), (lds_void_ptr)(Abuf + i * 256), 16, 0, 0);
            }
        }
        if (p.dbg & 1) return;
        if (desc_ok && cc * CK + CK <= p.C && !(p.dbg & 64)) {  // (no channel padding in this chunk; 64: A/B switch)
            const float* cbase = p.X + (size_t)cc * CK * p.XH * p.XW;
            // (ragged rows: the last unit of the tensor's very LAST row is fetched float by float in fix_tails -- a 16-byte DMA
            // there would read past the end of the allocation)
            const unsigned last_unit = p.row_tail ? (unsigned)((size_t)p.B * p.C * p.XH * p.XW - (size_t)cc * CK * p.XH * p.XW) - (unsigned)p.row_tail : ~0u;
#pragma unroll
            for (int i = 0; i < MAXD; ++i) {
                if (i < nd && d_src[i] != ~0u && d_src[i] != last_unit) {
                    if (p.run_mode == 2)
                        __builtin_amdgcn_global_load_lds((gbl_void_ptr)(cbase + d_src[i]), (lds_void_ptr)(Xbuf + d_lds[i]), 4, 0, 0);
                    else
                        __builtin_amdgcn_global_load_lds((gbl_void_ptr)(cbase + d_src[i]), (lds_void_ptr)(Xbuf + d_lds[i]), 16, 0, 0);
                }
            }
            return;
        }
        if constexpr (XM != 0) {
            // whole images: the CK-channel block of image b0+img is one contiguous run in HBM and in LDS
            const int HW = p.chs;  // == XH*XW
            const int cvalid = (p.C - cc * CK < CK) ? p.C - cc * CK : CK;
            const int unit = (p.run_mode == 1) ? 4 : 1;
            const int runu = cvalid * HW / unit;
            const int per_img = (runu + 63) / 64;
            const float* g0 = p.X + ((size_t)b0 * p.C + (size_t)cc * CK) * HW;
#pragma nounroll
            for (int j = wave; j < nseg * per_img; j += NWAVES) {
                const int img = j / per_img, part = j - img * per_img;
                const int idx = part * 64 + lane;
                const float* g = g0 + (size_t)img * p.C * HW;
                float* d = Xbuf + img * CK * HW + part * 64 * unit;
                if (idx < runu) {
                    if (p.run_mode == 1)
                        __builtin_amdgcn_global_load_lds((gbl_void_ptr)(g + (size_t)idx * 4), (lds_void_ptr)d, 16, 0, 0);
                    else
                        __builtin_amdgcn_global_load_lds((gbl_void_ptr)(g + idx), (lds_void_ptr)d, 4, 0, 0);
                }
            }
            if (cvalid < CK) {  // channel padding of the last chunk must be finite
                for (int img = 0; img < nseg; ++img)
                    for (int i = cvalid * HW + tid; i < CK * HW; i += NT) Xbuf[img * CK * HW + i] = 0.f;
            }
            return;
        }
        if (p.run_mode == 1 || p.run_mode == 2) {
            // every channel's rows of one image segment are one contiguous run (HBM and LDS): moved 1 KiB per instruction
            // when rows are 16-byte multiples (run_mode 1), 256 B per instruction otherwise (run_mode 2)
            const int unit = (p.run_mode == 1) ? 4 : 1;  // floats per lane
            int lrow0 = 0;
#pragma nounroll
            for (int sg = 0; sg < nseg; ++sg) {
                const int nr = (sg == 0) ? nrows0 : ((sg == nseg - 1) ? u1 * p.su + p.TR : full);
                const int xrow0 = (sg == 0) ? u0 * p.su + p.r0 : p.r0;
                const int runu = nr * p.XW / unit;     // lane-units per channel run
                const int per_ch = (runu + 63) / 64;   // wave-instructions per channel
                const size_t gbase = ((size_t)(b0 + sg) * p.C * p.XH + xrow0) * p.XW;
#pragma nounroll
                for (int j = wave; j < CK * per_ch; j += NWAVES) {
                    const int ck = j / per_ch, part = j - ck * per_ch;
                    const int c = cc * CK + ck;
                    const int idx = part * 64 + lane;
                    float* d = Xbuf + ck * p.chs + lrow0 * p.LW + part * 64 * unit;
                    if (c < p.C) {
                        const float* g = p.X + gbase + (size_t)c * p.XH * p.XW;
                        if (idx < runu) {
                            if (p.run_mode == 1)
                                __builtin_amdgcn_global_load_lds((gbl_void_ptr)(g + (size_t)idx * 4), (lds_void_ptr)d, 16, 0, 0);
                            else
                                __builtin_amdgcn_global_load_lds((gbl_void_ptr)(g + idx), (lds_void_ptr)d, 4, 0, 0);
                        }
                    } else if (idx < runu) {
                        if (p.run_mode == 1) *(float4*)(d + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
                        else d[lane] = 0.f;
                    }
                }
                lrow0 += nr;
            }
            return;
        }
#pragma nounroll
        for (int ck = 0; ck < CK; ++ck) {
            const int c = cc * CK + ck;
            float* xch = Xbuf + ck * p.chs + p.padL;
            if (c < p.C) {
                const float* gch = p.X + (size_t)c * p.XH * p.XW;
#pragma nounroll
                for (int r = wave; r < nrows; r += NWAVES) {
                    const int src = __builtin_amdgcn_readfirstlane(rowsrc[r]);
                    if (src >= 0) {
                        const float* g = gch + (size_t)src * p.XW;
                        float* d = xch + r * p.LW;
#pragma nounroll
                        for (int cb = 0; cb < p.XW; cb += 64)
                            if (cb + lane < p.XW)
                                __builtin_amdgcn_global_load_lds((gbl_void_ptr)(g + cb + lane), (lds_void_ptr)(d + cb), 4, 0, 0);
                    }
                }
            } else {  // channel padding of the last chunk must be finite: plain LDS stores into the idle buffer
                for (int r = wave; r < nrows; r += NWAVES) {
                    float* d = xch + r * p.LW;
                    for (int col = lane; col < p.XW; col += 64) d[col] = 0.f;
                }
            }
        }
    };

    // ragged rows (row_tail != 0): the units this wave fetched for chunk cc have landed (its own vmcnt(0)); put zeros back into the
    // pad columns behind the row's last valid float (they received the floats that follow the row in memory)
    auto fix_tails = [&](int cc) {
        float* Xbuf = smem + (cc & 1) * buf_floats + a_floats;
        if (!(desc_ok && cc * CK + CK <= p.C) || (p.dbg & (1 | 64))) return;
        const unsigned last_unit = (unsigned)((size_t)p.B * p.C * p.XH * p.XW - (size_t)cc * CK * p.XH * p.XW) - (unsigned)p.row_tail;
#pragma unroll
        for (int i = 0; i < MAXD; ++i) {
            if (i < nd && (d_tail >> i & 1u)) {
                float* u = Xbuf + d_lds[i] + lane * 4;
                if (d_src[i] == last_unit) {
                    const float* g = p.X + (size_t)cc * CK * p.XH * p.XW + d_src[i];
                    for (int t = 0; t < p.row_tail; ++t) u[t] = g[t];
                }
                for (int t = p.row_tail; t < 4; ++t) u[t] = 0.f;
            }
        }
    };

    issue_dma(c_lo, c_lo & 1);
    for (int cc = c_lo; cc < c_hi; ++cc) {
        // Every wave first waits for ITS OWN outstanding LDS-DMA (chunk cc), then the barrier publishes all of them and
        // guarantees every wave is done reading the other buffer.  The explicit wait is required: hipcc (ROCm 7.2) hoists
        // its own vmcnt(0) out of this loop, leaving the in-loop s_barrier unprotected (caught by tools/det_check.py).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (XM == 0 && p.row_tail != 0) fix_tails(cc);
        __syncthreads();
        if (cc + 1 < c_hi) issue_dma(cc + 1, (cc + 1) & 1);
        if (p.dbg & 4) continue;
        const float* As = smem + (cc & 1) * buf_floats;
        const float* Xs = As + a_floats;
        if constexpr (XM != 3 && NB >= 7) {
        // ---- wide pixel tiles (NB = 7 / 13 blocks of 16 pixels, round 4): B operands pipelined by one K-STEP instead of one tap -- 2 * NB
        //      registers instead of 2 * NB * S.  Every accumulator still sees (tap, k-step) in the same order as below.
        avec_t a_cur[MA];
        float b_cur[NB];
#pragma unroll
        for (int ma = 0; ma < MA; ++ma) a_cur[ma] = *(const avec_t*)(As + a_lane + ma * MF * S);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const bool ok = (XM < 2) || ((rmask[nb] & cmask[nb] & 1u) != 0);
            const float xv = Xs[pix_off[nb]];
            b_cur[nb] = ok ? xv : 0.f;
        }
        int tr = 0, tc = 0;
        for (int t = 0; t < T; ++t) {
            int ntc = tc + 1, ntr = tr;
            if (ntc == p.TC) { ntc = 0; ++ntr; }
            const bool last = (t + 1 == T);
            const int tap_cur = tr * p.LW + tc, tap_next = last ? 0 : ntr * p.LW + ntc;  // (after the last tap: a harmless re-read)
            const float* a_next = As + (last ? 0 : (t + 1) * KSTEP * MT * S) + a_lane;
            avec_t a_nxt[MA];
#pragma unroll
            for (int ma = 0; ma < MA; ++ma) a_nxt[ma] = *(const avec_t*)(a_next + ma * MF * S);
#pragma unroll
            for (int c2 = 0; c2 < S; ++c2) {
                float b_nxt[NB];
                const bool same = (c2 + 1 < S);
                const int boff = same ? tap_cur + (c2 + 1) * KSTEP * p.chs : tap_next;
                const int mr = same ? tr : ntr, mc = same ? tc : ntc;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const bool ok = (XM < 2) || (!same && last) || (((rmask[nb] >> mr) & (cmask[nb] >> mc) & 1u) != 0);
                    const float xv = Xs[pix_off[nb] + boff];
                    b_nxt[nb] = ok ? xv : 0.f;
                }
                __builtin_amdgcn_sched_barrier(0);  // reads of the next k-step stay above the MFMAs of this one
#pragma unroll
                for (int ma = 0; ma < MA; ++ma)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[ma][nb] = A_::mfma(a_cur[ma][c2], b_cur[nb], acc[ma][nb]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) b_cur[nb] = b_nxt[nb];
            }
#pragma unroll
            for (int ma = 0; ma < MA; ++ma) a_cur[ma] = a_nxt[ma];
            tr = ntr;
            tc = ntc;
        }
        } else
        // ---- MFMA, software pipelined by one tap ----
        if constexpr (XM != 3) {
        avec_t a_cur[MA];
        float b_cur[NB][S];
#pragma unroll
        for (int ma = 0; ma < MA; ++ma) a_cur[ma] = *(const avec_t*)(As + a_lane + ma * MF * S);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const bool ok = (XM < 2) || ((rmask[nb] & cmask[nb] & 1u) != 0);
#pragma unroll
            for (int c2 = 0; c2 < S; ++c2) {
                const float xv = Xs[pix_off[nb] + c2 * KSTEP * p.chs];
                b_cur[nb][c2] = ok ? xv : 0.f;
            }
        }
        int tr = 0, tc = 0;
        for (int t = 0; t < T; ++t) {
            int ntc = tc + 1, ntr = tr;
            if (ntc == p.TC) { ntc = 0; ++ntr; }
            const bool last = (t + 1 == T);
            const int tap_next = last ? 0 : ntr * p.LW + ntc;
            const float* a_next = As + (last ? 0 : (t + 1) * KSTEP * MT * S) + a_lane;
            avec_t a_nxt[MA];
            float b_nxt[NB][S];
#pragma unroll
            for (int ma = 0; ma < MA; ++ma) a_nxt[ma] = *(const avec_t*)(a_next + ma * MF * S);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const bool ok = (XM < 2) || last || (((rmask[nb] >> ntr) & (cmask[nb] >> ntc) & 1u) != 0);
#pragma unroll
                for (int c2 = 0; c2 < S; ++c2) {
                    const float xv = Xs[pix_off[nb] + tap_next + c2 * KSTEP * p.chs];
                    b_nxt[nb][c2] = ok ? xv : 0.f;
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // reads of tap t+1 stay above the MFMAs of tap t
#pragma unroll
            for (int c2 = 0; c2 < S; ++c2)
#pragma unroll
                for (int ma = 0; ma < MA; ++ma)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[ma][nb] = A_::mfma(a_cur[ma][c2], b_cur[nb][c2], acc[ma][nb]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ma = 0; ma < MA; ++ma) a_cur[ma] = a_nxt[ma];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int c2 = 0; c2 < S; ++c2) b_cur[nb][c2] = b_nxt[nb][c2];
            tr = ntr;
            tc = ntc;
        }
        } else {
        // ---- MFMA, software pipelined by one tap.  Only the taps some tile of this wave uses are visited, and a tile skips
        //      the taps its parity class has no filter element for (dgrad, stride > 1: 7 of 16 (class, tap) pairs of a
        //      3x3 / stride-2 filter are structurally zero) ----
        avec_t a_cur[MA];
        float b_cur[NB][S];
        int t = __builtin_ctz(umask);
        {
            const int tr = (t * p.tc_inv) >> 16, tc = t - tr * p.TC;
            const int tap_off = tr * p.LW + tc;
            const float* a_p = As + t * KSTEP * MT * S + a_lane;
#pragma unroll
            for (int ma = 0; ma < MA; ++ma) a_cur[ma] = *(const avec_t*)(a_p + ma * MF * S);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const bool ok = (XM < 2) || (((rmask[nb] >> tr) & (cmask[nb] >> tc) & 1u) != 0);
#pragma unroll
                for (int c2 = 0; c2 < S; ++c2) {
                    const float xv = Xs[pix_off[nb] + tap_off + c2 * KSTEP * p.chs];
                    b_cur[nb][c2] = ok ? xv : 0.f;
                }
            }
        }
        while (true) {
            const unsigned rest = umask & ((~1u) << t);
            const bool last = (rest == 0);
            const int tn = last ? t : __builtin_ctz(rest);  // after the last tap: a harmless re-read
            const int ntr = (tn * p.tc_inv) >> 16, ntc = tn - ntr * p.TC;
            const int tap_next = ntr * p.LW + ntc;
            const float* a_next = As + tn * KSTEP * MT * S + a_lane;
            avec_t a_nxt[MA];
            float b_nxt[NB][S];
#pragma unroll
            for (int ma = 0; ma < MA; ++ma) a_nxt[ma] = *(const avec_t*)(a_next + ma * MF * S);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const bool ok = (XM < 2) || (((rmask[nb] >> ntr) & (cmask[nb] >> ntc) & 1u) != 0);
#pragma unroll
                for (int c2 = 0; c2 < S; ++c2) {
                    const float xv = Xs[pix_off[nb] + tap_next + c2 * KSTEP * p.chs];
                    b_nxt[nb][c2] = ok ? xv : 0.f;
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // reads of the next tap stay above the MFMAs of this one
#pragma unroll
            for (int ma = 0; ma < MA; ++ma) {
                if ((tmask[ma] >> t) & 1u) {
#pragma unroll
                    for (int c2 = 0; c2 < S; ++c2)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
                            acc[ma][nb] = A_::mfma(a_cur[ma][c2], b_cur[nb][c2], acc[ma][nb]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ma = 0; ma < MA; ++ma) a_cur[ma] = a_nxt[ma];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int c2 = 0; c2 < S; ++c2) b_cur[nb][c2] = b_nxt[nb][c2];
            if (last) break;
            t = tn;
        }
        }
        }
    if (!(p.dbg & 8)) store_tile<MF, MA, NB, R2>(acc, p, (long long)n0 + (long long)wn * NB * MF, n1, mbase_wave, li, lh, p.Y + (long long)zk * p.zstride);
}

// ---- weight preparation: [Co][Ci][k][k]  ->  A[mblock][chunk][tap][ck][MT] (zero padded) -------------------
struct PrepParams {
    const float* w;
    float* A;
    int Co, Ci, k, s, pad;
    int mode;
    int C, M, TR, TC, r0, c0;
    int CK, MT, nchunk, nmb;
    int a4;  // S > 0: slab layout [tap][k-lane][m][S k-steps] (DMA kernel) instead of [tap][ck][m]
    int kstep;  // MFMA k per instruction (2 for 32x32x2, 4 for 16x16x4)
};

// IDX: the index arithmetic's type -- unsigned where the image has fewer than 2^31 elements (every layer of the three workloads): the
// chain of divisions by run-time values in 64 bits cost ~700 instructions per element (the 256 -> 512 stride-2 layer of the ResNet-shaped
// stack: 206 us of the side stream for 2.4 M elements, beside the HBM-bound BatchNorm2D pass of the main stream)
template <class IDX>
__device__ void igemm_prep_body_t(const PrepParams& q, IDX tid, IDX nt, IDX total) {
    const int T = q.TR * q.TC;
    for (IDX idx = tid; idx < total; idx += nt) {
        IDX r = idx;
        int mm, ck, t, cc, mb;
        if (q.a4) {  // [mb][cc][tap][lh][mm][c2]  with ck = 2*c2 + lh (32x32x2: k index = lane >> 5)
            const int c2 = (int)(r % q.a4); r /= q.a4;
            mm = (int)(r % q.MT); r /= q.MT;
            const int lh = (int)(r % q.kstep); r /= q.kstep;
            ck = q.kstep * c2 + lh;
        } else {     // [mb][cc][tap][ck][mm]
            mm = (int)(r % q.MT); r /= q.MT;
            ck = (int)(r % q.CK); r /= q.CK;
        }
        t = (int)(r % T); r /= T;
        cc = (int)(r % q.nchunk);
        mb = (int)(r / q.nchunk);
        const int m = mb * q.MT + mm, c = cc * q.CK + ck;
        const int tr = t / q.TC, tc = t % q.TC;
        float v = 0.f;
        if (m < q.M && c < q.C) {
            if (q.mode == MODE_FWD) {
                v = q.w[(((size_t)m * q.Ci + c) * q.k + tr) * q.k + tc];
            } else {
                // virtual channel m = (ph*s + pw)*Ci + ci reads dy at grid offset d = r0 + tr, which is tap
                // kx = (ph+pad)%s + s*j with j = (ph+pad)/s - d   (valid when 0 <= j and kx < k)
                const int cls = m / q.Ci, ci = m % q.Ci;
                const int ph = cls / q.s, pw = cls % q.s;
                const int jr = (ph + q.pad) / q.s - (q.r0 + tr), jc = (pw + q.pad) / q.s - (q.c0 + tc);
                const int kx = (ph + q.pad) % q.s + q.s * jr, ky = (pw + q.pad) % q.s + q.s * jc;
                if (jr >= 0 && jc >= 0 && kx < q.k && ky < q.k) v = q.w[(((size_t)c * q.Ci + ci) * q.k + kx) * q.k + ky];
            }
        }
        q.A[idx] = v;
    }
}

__device__ void igemm_prep_body(const PrepParams& q, long long tid, long long nt) {
    const long long total = (long long)q.nmb * q.nchunk * (q.TR * q.TC) * q.CK * q.MT;
    if (total + nt < (1ll << 31)) igemm_prep_body_t<unsigned>(q, (unsigned)tid, (unsigned)nt, (unsigned)total);
    else igemm_prep_body_t<long long>(q, tid, nt, total);
}

__global__ void igemm_prep_weights(const PrepParams q) {
    igemm_prep_body(q, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
}

// the re-layouts of several layers / modes in ONE launch (cnn_conv2d_prepare_filters): blockIdx.y = job
struct PrepBatch {
    int n;
    PrepParams q[12];
};
__global__ void igemm_prep_batch(const PrepBatch pb) {
    igemm_prep_body(pb.q[blockIdx.y], (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
}

// ---- split-K: y = P[0] + P[1] + ... + P[Z-1] in that order (P[0] carries the bias), with the epilogue the unsplit kernel would have fused:
// forward: y (nullable) and relu(y) (nullable); data gradient: dx = (mask <= 0) ? 0 : sum when the ReLU output in front is given
__global__ __launch_bounds__(256) void split_reduce(const float* __restrict__ part, long long zstride, int Z, size_t n4, size_t n, float* __restrict__ y,
                                                    float* __restrict__ y2, int mode) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = *(const float4*)(part + 4 * i);
        for (int z = 1; z < Z; ++z) {
            const float4 w = *(const float4*)(part + (size_t)z * zstride + 4 * i);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        if (mode == MODE_FWD) {
            if (y != nullptr) *(float4*)(y + 4 * i) = v;
            if (y2 != nullptr) *(float4*)(y2 + 4 * i) = make_float4(v.x >= 0.f ? v.x : 0.f, v.y >= 0.f ? v.y : 0.f, v.z >= 0.f ? v.z : 0.f, v.w >= 0.f ? v.w : 0.f);
        } else {
            if (y2 != nullptr) {
                const float4 m = *(const float4*)(y2 + 4 * i);
                v = make_float4(m.x <= 0.f ? 0.f : v.x, m.y <= 0.f ? 0.f : v.y, m.z <= 0.f ? 0.f : v.z, m.w <= 0.f ? 0.f : v.w);
            }
            *(float4*)(y + 4 * i) = v;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - 4 * n4)) {  // (tensors whose size is not a multiple of 4)
        const size_t i = 4 * n4 + threadIdx.x;
        float v = part[i];
        for (int z = 1; z < Z; ++z) v += part[(size_t)z * zstride + i];
        if (mode == MODE_FWD) {
            if (y != nullptr) y[i] = v;
            if (y2 != nullptr) y2[i] = v >= 0.f ? v : 0.f;
        } else {
            y[i] = (y2 != nullptr && y2[i] <= 0.f) ? 0.f : v;
        }
    }
}

// ---- host-side planning ------------------------------------------------------------------------------------
struct Plan {
    int cfg;  // which instantiation
    int MT, NPIX, CK, MF;
    IgemmParams p;
    PrepParams q;
    size_t lds_bytes;
    size_t a_floats;
    unsigned grid_x, grid_y;
    int dma;  // double-buffered DMA-staged kernel
    int xm;   // DMA kernel staging mode: 0 rows, 1 whole images, 2 whole images + masked taps
    int ksplit;        // > 1: split-K, partial tensors behind the filter image in the workspace (see IgemmParams::ksplit)
    size_t out_floats;  // floats of the output tensor (= one partial tensor)
};

enum { CFG_M128_L = 100, CFG_M128 = 0, CFG_M128_S, CFG_M64, CFG_M64_S, CFG_M32, CFG_M32_S, CFG_M16_CK4, CFG_M16_CK16, CFG_M16_CK4_L, CFG_M16_CK8, CFG_M16_CK8_L, CFG_D_M128 = 200, CFG_D_M64, CFG_D_M64W4, CFG_D_M128W4, CFG_D_M128W4N2, CFG_D_M128W4_C4, CFG_D_M128_C4, CFG_D_M64W4_C4, CFG_D_M64W4N1_C4,
       CFG_D16_C4 /*209*/, CFG_D16_C4_L, CFG_D16_C16, CFG_D16_C16_L, CFG_D16_C8, CFG_D16_C8_L,
       CFG_D_M32 /*215*/, CFG_D_M32_C4, CFG_D_M64N1 /*217*/, CFG_D_M128S /*218*/, CFG_D_M128S_C4, CFG_D_M64S /*220*/, CFG_D_M64S_C4,
       CFG_D_M32_C16 /*222*/, CFG_D_M64S_C16, CFG_D_M128S_C16, CFG_D_M64W4N1_C16, CFG_D_M64W4N1_C8 /*226*/, CFG_D_M64N2W8 /*227*/,
       CFG_W26_M64_N4 /*228*/, CFG_W26_M128_N2 /*229*/,
       CFG_W26_M128_N2_K2 /*230*/, CFG_W26_M128_N2_K4, CFG_W26_M128_N2_K8, CFG_W26_M64_N4_K2 /*233*/, CFG_W26_M64_N4_K4, CFG_W26_M64_N4_K8 /*235*/,
       CFG_M64_S_C16 = 20, CFG_M64_S_C32, CFG_M128_S_C16, CFG_M128_S_C32, CFG_M32_S_C16 /*24*/, CFG_M64_S_C4 /*25*/, CFG_M128_S_C4 /*26*/,
       CFG_M32_S_C4 /*27*/ };

// taps of dy one parity class reads: offsets d in [e - J + 1, e], e = (ph+pad)/s, J = #taps kx = kx0 + s*j < k
void dgrad_window(int k, int s, int pad, int* r0, int* TR) {
    int dmin = 1 << 30, dmax = -(1 << 30);
    for (int ph = 0; ph < s; ++ph) {
        const int kx0 = (ph + pad) % s, e = (ph + pad) / s;
        if (kx0 >= k) continue;
        const int J = (k - kx0 + s - 1) / s;
        if (e - J + 1 < dmin) dmin = e - J + 1;
        if (e > dmax) dmax = e;
    }
    *r0 = dmin;
    *TR = dmax - dmin + 1;
}

// ---- measured tile choice (cnn_conv2d_autotune) ---------------------------------------------------------------------------
// The rules in make_plan were tuned on the reference net's layers; on other geometries (the VGG / ResNet-shaped stacks) a
// different tile is often 1.3 - 2.6x faster (tools/sweep_igemm.py).  cnn_conv2d_autotune times a short candidate list ONCE per
// (geometry, mode) with scratch buffers of its own and pins the winner for this process; make_plan consults that table.
struct TuneKey {
    int v[9];
    bool operator<(const TuneKey& o) const {
        for (int i = 0; i < 9; ++i)
            if (v[i] != o.v[i]) return v[i] < o.v[i];
        return false;
    }
};
TuneKey tune_key(const cnn_conv2d_desc* d, int mode) { return TuneKey{{d->B, d->Ci, d->H, d->W, d->Co, d->k, d->s, d->pad, mode}}; }
std::mutex& tune_mutex() {
    static std::mutex m;
    return m;
}
std::map<TuneKey, int>& tune_table() {
    static std::map<TuneKey, int> t;
    return t;
}
// geometries the register-direct kernels cover but the implicit GEMM runs faster (measured by cnn_conv2d_autotune)
std::map<TuneKey, bool>& prefer_table() {
    static std::map<TuneKey, bool> t;
    return t;
}
thread_local int g_forced_cfg = -1;  // >= 0: make_plan must use exactly this configuration (the tuner's probe runs)
// candidates: the rule-based default (-1) plus the tiles that won somewhere in tools/sweep_igemm.py
// 227 (round 3): 64 output rows x 512 pixels per workgroup, every wave a 2 x 2 block of 32x32 MFMA tiles -- for M = 64 (the data
// gradient of a 64-channel layer: the north-star shape's) the 2 x 1 tiles of 201 stage twice the bytes per MFMA that the M = 128
// forward tile does; 2 x 2 restores the forward kernel's ratio of DMA bytes and LDS operand reads to MFMAs.  202: its 4-wave sibling.
// 228 / 229 (round 4, "wide" tiles): every wave owns 32 output rows x 13 blocks of 16 pixels (two 16x16x4 MFMA row blocks x 13 column blocks; 104
// accumulator registers), eight waves = 64 x 832 or 128 x 416 per workgroup.  (a) 62 % more outputs per workgroup than 128 x 256 for the same filter slab:
// fewer staged bytes and fewer chunk barriers per MFMA; (b) 13 x 16 = 208 divides the 49 * 2^k pixel counts of 7 / 14 / 28 / 56 / 112-wide
// layers almost evenly: 64 -> 64 56x56 at batch 64 is 242 workgroups (one round of 256 CUs) instead of 392 (136 CUs run two, 120 one).
// 230 - 235: the wide tiles with the channel range split in 2 / 4 / 8 (planes of 7x7 .. 28x28 at batch 64 give 32 .. 122 wide tiles for 256 CUs)
const int kTuneCandidates[] = {-1, 200, 201, 206, 207, 208, 226, 222, 224, 215, 216, 219, 225, 0, 1, 22, 227, 202, 228, 229, 230, 231, 232, 233, 234, 235};  // (207: 4-channel chunks, the 3 -> 64 first layer of the VGG-shaped stack: 1 059 -> 853 us)

int make_plan(const char* who, const cnn_conv2d_desc* d, int mode, Plan* pl, bool allow_dma = true, int shrink = 0) {
    const int Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, d->pad), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, d->pad);
    CNN_REQUIRE(Ho > 0 && Wo > 0, "%s: empty output", who);
    IgemmParams& p = pl->p;
    PrepParams& q = pl->q;
    p = IgemmParams();
    q = PrepParams();
    p.B = d->B;
    p.mode = mode;
    pl->ksplit = 1;
    if (mode == MODE_FWD) {
        p.C = d->Ci; p.XH = d->H; p.XW = d->W; p.U = Ho; p.V = Wo; p.su = d->s;
        p.TR = p.TC = d->k; p.r0 = p.c0 = -d->pad; p.M = d->Co;
    } else {
        p.C = d->Co; p.XH = Ho; p.XW = Wo;
        p.U = (d->H + d->s - 1) / d->s; p.V = (d->W + d->s - 1) / d->s; p.su = 1;
        dgrad_window(d->k, d->s, d->pad, &p.r0, &p.TR);
        p.c0 = p.r0; p.TC = p.TR;
        p.M = d->s * d->s * d->Ci;
        p.s_out = d->s; p.c_out = d->Ci; p.OH = d->H; p.OW = d->W;
    }
    p.N = (long long)p.B * p.U * p.V;
    CNN_REQUIRE((long long)p.B * p.C * p.XH < (1ll << 31), "%s: B*C*H exceeds int32 row index", who);
    CNN_REQUIRE(p.N < (1ll << 31) - 4096, "%s: too many output pixels (B*Ho*Wo must stay below 2^31)", who);

    // tile choice: the widest pixel tile that still gives every CU a couple of workgroups
    auto blocks_for = [&](int MT, int NPIX) { return ((p.N + NPIX - 1) / NPIX) * ((p.M + MT - 1) / MT); };
    const long long kWantBlocks = 2 * num_cus();
    // forward without padding: the DMA kernel can move whole multi-row runs -> worth it even for small images
    const bool unpadded = (mode == MODE_FWD && d->pad == 0);
    const bool dma_ok = allow_dma && p.TR * p.TC <= 9 && p.N < (1ll << 31) - 4096;
    // small images (whole-image staging applies, see igemm_dma_kernel XM): tiles sized for enough workgroups at the
    // batch sizes of the reference net; measured on conv_layer_3 / conv_layer_4 (alexnet.cpp:19,22) forward and dgrad
    // (only when the large MFMA tiles below would leave CUs idle: a 28x28 layer at batch 128 is a big GEMM)
    const bool small_img = dma_ok && p.XH * p.XW <= 1024 && p.M > 32 && !CNN_OPT_SET("IGEMM_NOIMG") &&
                           !(p.M > 64 && blocks_for(128, 256) >= 2 * num_cus()) && !(p.M <= 64 && blocks_for(64, 128) >= 8 * num_cus());
    if (small_img && p.M > 64 && mode == MODE_FWD) { pl->cfg = CFG_D_M64S; pl->MF = 32; pl->MT = 64; pl->NPIX = 64; pl->CK = 8; }
    else if (small_img && p.M > 64 && p.C >= 128) { pl->cfg = CFG_D_M64S_C16; pl->MF = 32; pl->MT = 64; pl->NPIX = 64; pl->CK = 16; }
    else if (small_img && p.M > 64) { pl->cfg = CFG_D_M32; pl->MF = 32; pl->MT = 32; pl->NPIX = 128; pl->CK = 8; }
    else if (small_img && p.C >= 16) { pl->cfg = CFG_D_M64W4N1_C4; pl->MF = 32; pl->MT = 64; pl->NPIX = 128; pl->CK = 4; }
    else if (p.M > 64) {
        pl->MF = 32; pl->MT = 128; pl->CK = 8;
        if (allow_dma && blocks_for(128, 256) >= 2 * num_cus() && p.TR * p.TC <= 9 && p.N < (1ll << 31) - 1024) { pl->cfg = CFG_D_M128; pl->NPIX = 256; }
        else if (blocks_for(128, 128) >= kWantBlocks) { pl->cfg = CFG_M128; pl->NPIX = 128; }
        else if (dma_ok && unpadded && blocks_for(128, 64) >= num_cus() / 2) { pl->cfg = CFG_D_M128S; pl->NPIX = 64; }
        else { pl->cfg = CFG_M128_S; pl->NPIX = 64; }
    } else if (p.M > 32) {
        pl->MF = 32; pl->MT = 64; pl->CK = 8;
        if (allow_dma && blocks_for(64, 128) >= 4 * num_cus() && p.TR * p.TC <= 9 && p.N < (1ll << 31) - 1024 && p.C >= 16) {
            pl->cfg = CFG_D_M64W4N1_C4; pl->NPIX = 128; pl->CK = 4;
        } else if (blocks_for(64, 256) >= kWantBlocks) { pl->cfg = CFG_M64; pl->NPIX = 256; }
        else { pl->cfg = CFG_M64_S; pl->NPIX = 64; }
    } else if (p.M > 16) {
        pl->MF = 32; pl->MT = 32; pl->CK = 8;
        if (dma_ok && unpadded && blocks_for(32, 128) >= kWantBlocks && blocks_for(32, 512) < 4 * kWantBlocks) {
            pl->cfg = CFG_D_M32_C4; pl->NPIX = 128; pl->CK = 4;
        } else if (blocks_for(32, 512) >= kWantBlocks) { pl->cfg = CFG_M32; pl->NPIX = 512; }
        else { pl->cfg = CFG_M32_S; pl->NPIX = 128; }
    } else if (p.C <= 4) { pl->cfg = CFG_M16_CK4; pl->MF = 16; pl->MT = 16; pl->NPIX = 256; pl->CK = 4; }
    else if (allow_dma && p.N < (1ll << 31) - 1024 && p.TR * p.TC <= 9) { pl->cfg = CFG_D16_C4; pl->MF = 16; pl->MT = 16; pl->NPIX = 256; pl->CK = 4; }
    else { pl->cfg = CFG_M16_CK16; pl->MF = 16; pl->MT = 16; pl->NPIX = 256; pl->CK = 16; }

    // LDS fallback chain (large filters / wide rows): narrower pixel tile, then 4-channel chunks
    if (shrink >= 1) {
        const bool c4 = shrink >= 2;
        if (p.M > 64) { pl->cfg = c4 ? CFG_M128_S_C4 : CFG_M128_S; pl->MF = 32; pl->MT = 128; pl->NPIX = 64; pl->CK = c4 ? 4 : 8; }
        else if (p.M > 32) { pl->cfg = c4 ? CFG_M64_S_C4 : CFG_M64_S; pl->MF = 32; pl->MT = 64; pl->NPIX = 64; pl->CK = c4 ? 4 : 8; }
        else if (p.M > 16) { pl->cfg = c4 ? CFG_M32_S_C4 : CFG_M32_S; pl->MF = 32; pl->MT = 32; pl->NPIX = 128; pl->CK = c4 ? 4 : 8; }
        else { pl->cfg = CFG_M16_CK4; pl->MF = 16; pl->MT = 16; pl->NPIX = 256; pl->CK = 4; }
    }
    // overrides: CNN_AMD_IGEMM_CFG=<cfg id> (debug only) > the tuner's probe > the tuner's pinned choice for this geometry
    int override_cfg = -1;
    if (const OptVal ov = CNN_OPT_VAL("IGEMM_CFG")) override_cfg = atoi(ov);
    else if (g_forced_cfg >= 0) override_cfg = g_forced_cfg;
    else if (shrink == 0 && allow_dma) {
        std::lock_guard<std::mutex> lk(tune_mutex());
        auto it = tune_table().find(tune_key(d, mode));
        if (it != tune_table().end()) override_cfg = it->second;
    }
    const bool pinned = override_cfg >= 0;
    if (pinned) {
        const int c = override_cfg;
        struct { int cfg, MF, MT, NPIX, CK, Z; } tab[] = {
            {CFG_M128_L, 32, 128, 256, 8}, {CFG_M128, 32, 128, 128, 8}, {CFG_M128_S, 32, 128, 64, 8},
            {CFG_M64, 32, 64, 256, 8}, {CFG_M64_S, 32, 64, 64, 8}, {CFG_M32, 32, 32, 512, 8}, {CFG_M32_S, 32, 32, 128, 8},
            {CFG_M16_CK4, 16, 16, 256, 4}, {CFG_M16_CK16, 16, 16, 256, 16}, {CFG_M16_CK4_L, 16, 16, 512, 4},
            {CFG_M16_CK8, 16, 16, 256, 8}, {CFG_M16_CK8_L, 16, 16, 512, 8},
            {CFG_D_M128, 32, 128, 256, 8}, {CFG_D_M64, 32, 64, 256, 8}, {CFG_D_M64W4, 32, 64, 256, 8},
            {CFG_D_M128W4, 32, 128, 128, 8}, {CFG_D_M128W4N2, 32, 128, 256, 8}, {CFG_D_M128W4_C4, 32, 128, 128, 4},
            {CFG_D_M128_C4, 32, 128, 256, 4}, {CFG_D_M64W4_C4, 32, 64, 256, 4}, {CFG_D_M64W4N1_C4, 32, 64, 128, 4},
            {CFG_D16_C4, 16, 16, 256, 4}, {CFG_D16_C4_L, 16, 16, 512, 4}, {CFG_D16_C16, 16, 16, 256, 16},
            {CFG_D16_C16_L, 16, 16, 512, 16}, {CFG_D16_C8, 16, 16, 256, 8}, {CFG_D16_C8_L, 16, 16, 512, 8},
            {CFG_D_M32, 32, 32, 128, 8}, {CFG_D_M32_C4, 32, 32, 128, 4}, {CFG_D_M64N1, 32, 64, 128, 8},
            {CFG_D_M128S, 32, 128, 64, 8}, {CFG_D_M128S_C4, 32, 128, 64, 4}, {CFG_D_M64S, 32, 64, 64, 8}, {CFG_D_M64S_C4, 32, 64, 64, 4},
            {CFG_D_M32_C16, 32, 32, 128, 16}, {CFG_D_M64S_C16, 32, 64, 64, 16}, {CFG_D_M128S_C16, 32, 128, 64, 16},
            {CFG_D_M64W4N1_C16, 32, 64, 128, 16}, {CFG_D_M64W4N1_C8, 32, 64, 128, 8}, {CFG_D_M64N2W8, 32, 64, 512, 8},
            {CFG_W26_M64_N4, 16, 64, 832, 8}, {CFG_W26_M128_N2, 16, 128, 416, 8},
            {CFG_W26_M128_N2_K2, 16, 128, 416, 8, 2}, {CFG_W26_M128_N2_K4, 16, 128, 416, 8, 4}, {CFG_W26_M128_N2_K8, 16, 128, 416, 8, 8},
            {CFG_W26_M64_N4_K2, 16, 64, 832, 8, 2}, {CFG_W26_M64_N4_K4, 16, 64, 832, 8, 4}, {CFG_W26_M64_N4_K8, 16, 64, 832, 8, 8},
            {CFG_M64_S_C16, 32, 64, 64, 16}, {CFG_M64_S_C32, 32, 64, 64, 32}, {CFG_M128_S_C16, 32, 128, 64, 16},
            {CFG_M128_S_C32, 32, 128, 64, 32}, {CFG_M32_S_C16, 32, 32, 128, 16}, {CFG_M64_S_C4, 32, 64, 64, 4},
            {CFG_M128_S_C4, 32, 128, 64, 4}, {CFG_M32_S_C4, 32, 32, 128, 4}};
        for (auto& t : tab)
            if (t.cfg == c && p.M <= ((p.M + t.MT - 1) / t.MT) * t.MT && (t.MT >= 32 || p.M <= 16)) {
                if (t.Z > 1) {
                    // split-K tiles: only where the plain tile leaves CUs without a workgroup and the split fills about one round, every
                    // range keeps >= 2 chunks, and the partial tensors stay small (they live in the caller's workspace / prepared buffer)
                    const long long blocks = blocks_for(t.MT, t.NPIX), chunks = (p.C + t.CK - 1) / t.CK;
                    const long long outf = mode == MODE_FWD ? p.N * p.M : (long long)p.B * d->Ci * d->H * d->W;
                    if (!(blocks < num_cus() && blocks * t.Z <= num_cus() + num_cus() / 4 && blocks * t.Z > num_cus() / 2 && chunks >= 2 * t.Z &&
                          outf * t.Z <= (64ll << 20) && dma_ok))
                        continue;
                    pl->ksplit = t.Z;
                }
                pl->cfg = t.cfg; pl->MF = t.MF; pl->MT = t.MT; pl->NPIX = t.NPIX; pl->CK = t.CK;
            }
    }
    p.nchunk = (p.C + pl->CK - 1) / pl->CK;
    p.padL = p.c0 < 0 ? -p.c0 : 0;
    int padR = (p.V - 1) * p.su + p.TC - 1 + p.c0 - (p.XW - 1);
    if (padR < 0) padR = 0;
    p.LW = p.padL + p.XW + padR;
    // DMA configurations on padded layers whose rows are 16-byte multiples: 4-float left pad, pitch a multiple of 4 floats -> the row
    // image is staged by 16-byte DMA instructions that span several rows (run_mode 3 of igemm_dma_kernel)
    const bool dma_cfg = pl->cfg >= CFG_D_M128;
    // (round 3) rows that are NOT whole 16-byte units (the north-star data gradient: dy rows of 110 floats, pad 2) take the same
    // path: the last unit of a row then carries XW % 4 valid floats and up to three floats of whatever follows the row in memory,
    // which land in the right-pad columns -- the wave that issued the unit zeroes them again once its DMA has landed (the pad
    // columns must read as zeros).  IGEMM_RAGGED_ROWS=0: the old per-row 4-byte DMA for such layers.
    const bool ragged_ok = p.XW % 4 == 0 || CNN_OPT_INT("IGEMM_RAGGED_ROWS", 1) != 0;
    const bool vecrows = dma_cfg && ragged_ok && p.XW >= 4 && p.padL > 0 && p.padL <= 4 && !((CNN_OPT_SET("IGEMM_NOVECROWS") && CNN_OPT_INT("IGEMM_NOVECROWS", 0) != 0));
    if (vecrows) {
        p.padL = 4;
        p.LW = (4 + p.XW + padR + 3) & ~3;
    }
    long long out_rows = (pl->NPIX + p.V - 2) / p.V + 1;
    if (out_rows > (long long)p.B * p.U) out_rows = (long long)p.B * p.U;
    long long nseg = (out_rows + p.U - 2) / p.U + 1;
    if (nseg > p.B) nseg = p.B;
    p.nrows_max = (int)(out_rows * p.su + nseg * (p.TR > p.su ? p.TR - p.su : 0));
    p.chs = p.nrows_max * p.LW + (p.TR - 1) * 0;  // window never leaves the rows staged for its own pixel
    // the last pixel's window may run up to TC-1 floats past its row end only inside the padded pitch -> in range.
    if (pl->MF == 16) {  // de-conflict the four k-groups of a 16x16x4 B read (see profiles/NOTEBOOK.md "LDS banking")
        const int want = (p.su & 1) ? 16 : 17;
        p.chs += ((want - p.chs % 32) + 32) % 32;
    }
    const int T = p.TR * p.TC;
    pl->dma = pl->cfg >= CFG_D_M128;
    if (pl->dma) p.chs = (p.chs + 3) & ~3;  // 16-byte aligned buffers for the dwordx4 DMA
    pl->a_floats = (size_t)((p.M + pl->MT - 1) / pl->MT) * p.nchunk * T * pl->CK * pl->MT;
    pl->lds_bytes = ((size_t)T * pl->CK * pl->MT + (size_t)pl->CK * p.chs) * sizeof(float) * (pl->dma ? 2 : 1) +
                    (size_t)p.nrows_max * 4;
    const int need_zero = (p.padL > 0 || padR > 0 || p.r0 < 0 || (p.U - 1) * p.su + p.r0 + p.TR - 1 > p.XH - 1) ? 1 : 0;
    p.need_zero = need_zero;
    pl->xm = 0;
    // whole-image staging for small images (see igemm_dma_kernel): CNN_AMD_IGEMM_XM=0 disables (tuning only)
    {
        const int HW = p.XH * p.XW;
        const int UVp = p.U * p.V;
        long long nimg = (pl->NPIX + UVp - 2) / UVp + 1;
        if (nimg > p.B) nimg = p.B;
        const bool img_cfg = pl->cfg == CFG_D_M64W4N1_C4 || pl->cfg == CFG_D_M64W4N1_C8 || pl->cfg == CFG_D_M64W4N1_C16 ||
                             pl->cfg == CFG_D_M32 || pl->cfg == CFG_D_M32_C4 || pl->cfg == CFG_D_M32_C16 ||
                             pl->cfg == CFG_D_M128S || pl->cfg == CFG_D_M128S_C4 || pl->cfg == CFG_D_M128S_C16 ||
                             pl->cfg == CFG_D_M64S || pl->cfg == CFG_D_M64S_C4 || pl->cfg == CFG_D_M64S_C16;  // (the wide tiles were measured with whole-image staging too: 14x14 k/4 192 vs 172 us, 7x7 k/8 193 vs 188 us -- rows stay)
        const OptVal xe = CNN_OPT_VAL("IGEMM_XM");
        const size_t lds = 2 * ((size_t)T * pl->CK * pl->MT + (size_t)nimg * pl->CK * HW) * sizeof(float);
        if (img_cfg && HW <= 1024 && lds <= 160 * 1024 && (long long)p.B * p.C * HW < (1ll << 31) && !(xe && atoi(xe) == 0)) {
            pl->xm = p.need_zero ? 2 : 1;
            p.padL = 0;
            p.LW = p.XW;
            p.chs = HW;
            p.nrows_max = (int)nimg;  // images per tile (the row table is not used in this mode)
            pl->lds_bytes = lds;
            const bool vec = (p.C % pl->CK == 0) && ((pl->CK * HW) % 4 == 0) && (((long long)p.C * HW) % 4 == 0);
            p.run_mode = vec ? 1 : 2;
        }
    }
    if (pl->lds_bytes > 160 * 1024 && pl->dma && allow_dma && !pinned)
        return make_plan(who, d, mode, pl, false);  // two buffers do not fit: single-buffered kernel
    if (pl->lds_bytes > 160 * 1024 && shrink < 2 && !pinned)
        return make_plan(who, d, mode, pl, false, shrink + 1);
    CNN_REQUIRE(pl->lds_bytes <= 160 * 1024, "%s: tile needs %zu B of LDS (> 160 KiB): k=%d W=%d not supported", who,
                pl->lds_bytes, d->k, d->W);
    p.rw_shift = 0;
    while ((1 << p.rw_shift) < p.XW && p.rw_shift < 6) ++p.rw_shift;
    p.dbg = CNN_MEASURE_INT("DBG", 0);
    p.ntiles = (int)((p.N + pl->NPIX - 1) / pl->NPIX);
    pl->grid_x = (unsigned)p.ntiles;
    pl->grid_y = (unsigned)((p.M + pl->MT - 1) / pl->MT);

    p.tc_inv = 65536 / p.TC + 1;
    p.ncls = 0;
    if (mode == MODE_DGRAD && d->s > 1 && d->s * d->s <= 16 && T <= 32) {
        // class (ph,pw) reads window position t = (tr,tc) iff tap kx = (ph+pad)%s + s*jr, jr = (ph+pad)/s - (r0+tr), exists
        auto uses = [&](int phase, int tt, int org) {
            const int j = (phase + d->pad) / d->s - (org + tt);
            return j >= 0 && (phase + d->pad) % d->s + d->s * j < d->k;
        };
        p.ncls = d->s * d->s;
        for (int cls = 0; cls < p.ncls; ++cls) {
            unsigned mk = 0;
            for (int tr = 0; tr < p.TR; ++tr)
                for (int tc = 0; tc < p.TC; ++tc)
                    if (uses(cls / d->s, tr, p.r0) && uses(cls % d->s, tc, p.c0)) mk |= 1u << (tr * p.TC + tc);
            p.cls_mask[cls] = mk;
        }
    }
    if (pl->xm == 2 && p.ncls > 0 && pl->CK >= 8 && !CNN_OPT_SET("NO_TAPSKIP")) pl->xm = 3;  // (with 2 k-steps per tap the branches cost more than the skipped MFMAs)
    if (pl->xm != 3) p.ncls = 0;
    pl->out_floats = mode == MODE_FWD ? (size_t)p.N * p.M : (size_t)p.B * d->Ci * d->H * d->W;
    p.ksplit = pl->ksplit;
    p.zstride = pl->ksplit > 1 ? (long long)((pl->out_floats + 63) / 64 * 64) : 0;
    q.Co = d->Co; q.Ci = d->Ci; q.k = d->k; q.s = d->s; q.pad = d->pad; q.mode = mode;
    q.C = p.C; q.M = p.M; q.TR = p.TR; q.TC = p.TC; q.r0 = p.r0; q.c0 = p.c0;
    q.CK = pl->CK; q.MT = pl->MT; q.nchunk = p.nchunk; q.nmb = (int)pl->grid_y; q.kstep = pl->MF == 32 ? 2 : 4;
    q.a4 = pl->dma ? pl->CK / q.kstep : 0;
    if (pl->xm == 0) p.run_mode = (pl->dma && !p.need_zero && p.LW == p.XW) ? ((p.XW % 4 == 0) ? 1 : 2) : ((pl->dma && vecrows) ? 3 : 0);
    p.row_tail = (pl->xm == 0 && p.run_mode == 3) ? p.XW % 4 : 0;
    return CNN_AMD_OK;
}

#define CONV_TAG(d) "B%d Ci%d %dx%d Co%d k%d s%d p%d", (d)->B, (d)->Ci, (d)->H, (d)->W, (d)->Co, (d)->k, (d)->s, (d)->pad

template <int MF, int MA, int NB, int WM, int WN, int CK, bool NARROW, bool R2>
int launch_cfg3(const Plan& pl, hipStream_t s, const cnn_conv2d_desc* d) {
    auto kern = igemm_kernel<MF, MA, NB, WM, WN, CK, NARROW, R2>;
    static thread_local size_t max_set = 0;
    if (pl.lds_bytes > 48 * 1024 && pl.lds_bytes > max_set) {
        CNN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        max_set = 160 * 1024;
    }
    char name[96];
    snprintf(name, sizeof(name), "igemm_kernel<%d,%d,%d,%d,%d,%d>%s", MF, MA, NB, WM, WN, CK,
             pl.p.mode == MODE_FWD ? (R2 ? "/fwd+relu" : "/fwd") : (R2 ? "/dgrad+relu" : "/dgrad"));
    CNN_KLAUNCH(s, name, (kern<<<dim3(pl.grid_x, pl.grid_y), 64 * WM * WN, pl.lds_bytes, s>>>(pl.p)), CONV_TAG(d));
    return CNN_AMD_OK;
}

template <int MF, int MA, int NB, int WM, int WN, int CK, bool NARROW>
int launch_cfg2(const Plan& pl, hipStream_t s, const cnn_conv2d_desc* d) {
    return pl.p.Y2 != nullptr ? launch_cfg3<MF, MA, NB, WM, WN, CK, NARROW, true>(pl, s, d)
                              : launch_cfg3<MF, MA, NB, WM, WN, CK, NARROW, false>(pl, s, d);
}

template <int MF, int MA, int NB, int WM, int WN, int S, int XM, bool R2>
int launch_dma_xm2(const Plan& pl, hipStream_t s, const cnn_conv2d_desc* d) {
    auto kern = igemm_dma_kernel<MF, MA, NB, WM, WN, S, XM, R2>;
    static DeviceOnce attr_once;
    if (attr_once.needed()) {
        CNN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_once.mark();
    }
    char name[96];
    char ks[16] = "";
    if (pl.ksplit > 1) snprintf(ks, sizeof(ks), ",k/%d", pl.ksplit);
    snprintf(name, sizeof(name), "igemm_dma_kernel<%d,%d,%d,%d,%d,%d%s%s>%s", MF, MA, NB, WM, WN, S,
             XM == 0 ? "" : (XM == 1 ? ",img" : (XM == 2 ? ",img+mask" : ",img+mask+skip")), ks,
             pl.p.mode == MODE_FWD ? (R2 ? "/fwd+relu" : "/fwd") : (R2 ? "/dgrad+relu" : "/dgrad"));
    CNN_KLAUNCH(s, name, (kern<<<dim3(pl.grid_x, pl.grid_y, pl.ksplit), 64 * WM * WN, pl.lds_bytes, s>>>(pl.p)), CONV_TAG(d));
    return CNN_AMD_OK;
}

template <int MF, int MA, int NB, int WM, int WN, int S, int XM>
int launch_dma_xm(const Plan& pl, hipStream_t s, const cnn_conv2d_desc* d) {
    if (pl.p.Y2 != nullptr) return launch_dma_xm2<MF, MA, NB, WM, WN, S, XM, true>(pl, s, d);
    return launch_dma_xm2<MF, MA, NB, WM, WN, S, XM, false>(pl, s, d);
}

// IMG: this tile shape also has the whole-image staging instantiations (the small-image configs)
template <int MF, int MA, int NB, int WM, int WN, int S, bool IMG = false>
int launch_dma(const Plan& pl, hipStream_t s, const cnn_conv2d_desc* d) {
    if constexpr (IMG) {
        if (pl.xm == 1) return launch_dma_xm<MF, MA, NB, WM, WN, S, 1>(pl, s, d);
        if (pl.xm == 2) return launch_dma_xm<MF, MA, NB, WM, WN, S, 2>(pl, s, d);
        if (pl.xm == 3) return launch_dma_xm<MF, MA, NB, WM, WN, S, 3>(pl, s, d);
    }
    if (pl.xm != 0) return fail(CNN_AMD_E_BADARG, "internal: whole-image staging not instantiated for cfg %d", pl.cfg);
    return launch_dma_xm<MF, MA, NB, WM, WN, S, 0>(pl, s, d);
}

template <int MF, int MA, int NB, int WM, int WN, int CK>
int launch_cfg(const Plan& pl, hipStream_t s, const cnn_conv2d_desc* d) {
    return pl.p.XW <= 64 ? launch_cfg2<MF, MA, NB, WM, WN, CK, true>(pl, s, d)
                         : launch_cfg2<MF, MA, NB, WM, WN, CK, false>(pl, s, d);
}

// prepared: `ws` already holds the re-arranged filters (cnn_conv2d_prepare_filters); w is then unused
int run_plan(Plan& pl, const cnn_conv2d_desc* d, const float* X, const float* w, const float* bias, float* Y, float* Y2,
             void* ws, size_t ws_bytes, hipStream_t s, const char* who, bool prepared = false) {
    CNN_REQUIRE(ws != nullptr, "%s: workspace is null", who);
    const size_t a_al = (pl.a_floats + 63) / 64 * 64;  // (split-K: the partial tensors start 256-byte aligned behind the filter image)
    const size_t need = pl.ksplit > 1 ? a_al + (size_t)pl.ksplit * (size_t)pl.p.zstride : pl.a_floats;
    if (ws_bytes < need * sizeof(float))
        return fail(CNN_AMD_E_WORKSPACE, "%s: workspace %zu B < %zu B", who, ws_bytes, need * sizeof(float));
    if (!prepared) {
        pl.q.w = w;
        pl.q.A = (float*)ws;
        const long long total = (long long)pl.a_floats;
        unsigned pg = (unsigned)((total + 255) / 256);
        if (pg > 4096) pg = 4096;
        CNN_KLAUNCH(s, pl.p.mode == MODE_FWD ? "igemm_prep_weights/fwd" : "igemm_prep_weights/dgrad",
                    (igemm_prep_weights<<<pg, 256, 0, s>>>(pl.q)), CONV_TAG(d));
    }
    pl.p.X = X; pl.p.A = (const float*)ws; pl.p.bias = bias; pl.p.Y = Y; pl.p.Y2 = Y2;
    if (pl.ksplit > 1) {
        // the ranges write partial tensors (plain epilogue), split_reduce adds them in range order and applies the fused epilogue
        float* part = (float*)ws + a_al;
        CNN_REQUIRE(reinterpret_cast<uintptr_t>(part) % 16 == 0 && (Y == nullptr || reinterpret_cast<uintptr_t>(Y) % 16 == 0) &&
                        (Y2 == nullptr || reinterpret_cast<uintptr_t>(Y2) % 16 == 0), "%s: split-K needs 16-byte aligned tensors", who);
        pl.p.Y = part; pl.p.Y2 = nullptr;
        int rc;
        switch (pl.cfg) {
            case CFG_W26_M128_N2_K2: case CFG_W26_M128_N2_K4: case CFG_W26_M128_N2_K8: rc = launch_dma<16, 2, 13, 4, 2, 2>(pl, s, d); break;
            default: rc = launch_dma<16, 2, 13, 2, 4, 2>(pl, s, d); break;
        }
        if (rc) return rc;
        const size_t n = pl.out_floats, n4 = n / 4;
        size_t g = (n4 + 255) / 256;
        if (g > 4096) g = 4096;
        if (g < 1) g = 1;
        CNN_KLAUNCH(s, pl.p.mode == MODE_FWD ? (Y2 ? "split_reduce/fwd+relu" : "split_reduce/fwd") : (Y2 ? "split_reduce/dgrad+relu" : "split_reduce/dgrad"),
                    (split_reduce<<<(unsigned)g, 256, 0, s>>>(part, pl.p.zstride, pl.ksplit, n4, n, Y, Y2, pl.p.mode)), CONV_TAG(d));
        return CNN_AMD_OK;
    }
    switch (pl.cfg) {
        case CFG_D_M128: return launch_dma<32, 4, 1, 1, 8, 4>(pl, s, d);
        case CFG_D_M64: return launch_dma<32, 2, 1, 1, 8, 4>(pl, s, d);
        case CFG_D_M64W4: return launch_dma<32, 2, 2, 1, 4, 4>(pl, s, d);
        case CFG_D_M128W4: return launch_dma<32, 4, 1, 1, 4, 4>(pl, s, d);
        case CFG_D_M128W4N2: return launch_dma<32, 4, 2, 1, 4, 4>(pl, s, d);
        case CFG_D_M128W4_C4: return launch_dma<32, 4, 1, 1, 4, 2>(pl, s, d);
        case CFG_D_M128_C4: return launch_dma<32, 4, 1, 1, 8, 2>(pl, s, d);
        case CFG_D_M64W4_C4: return launch_dma<32, 2, 2, 1, 4, 2>(pl, s, d);
        case CFG_D_M64W4N1_C4: return launch_dma<32, 2, 1, 1, 4, 2, true>(pl, s, d);
        case CFG_D_M32: return launch_dma<32, 1, 1, 1, 4, 4, true>(pl, s, d);
        case CFG_D_M32_C4: return launch_dma<32, 1, 1, 1, 4, 2, true>(pl, s, d);
        case CFG_D_M64N1: return launch_dma<32, 2, 1, 1, 4, 4>(pl, s, d);
        case CFG_D_M128S: return launch_dma<32, 2, 1, 2, 2, 4, true>(pl, s, d);
        case CFG_D_M128S_C4: return launch_dma<32, 2, 1, 2, 2, 2, true>(pl, s, d);
        case CFG_D_M64S: return launch_dma<32, 1, 1, 2, 2, 4, true>(pl, s, d);
        case CFG_D_M64S_C4: return launch_dma<32, 1, 1, 2, 2, 2, true>(pl, s, d);
        case CFG_D_M32_C16: return launch_dma<32, 1, 1, 1, 4, 8, true>(pl, s, d);
        case CFG_D_M64S_C16: return launch_dma<32, 1, 1, 2, 2, 8, true>(pl, s, d);
        case CFG_D_M128S_C16: return launch_dma<32, 2, 1, 2, 2, 8, true>(pl, s, d);
        case CFG_D_M64W4N1_C16: return launch_dma<32, 2, 1, 1, 4, 8, true>(pl, s, d);
        case CFG_D_M64W4N1_C8: return launch_dma<32, 2, 1, 1, 4, 4, true>(pl, s, d);
        case CFG_D_M64N2W8: return launch_dma<32, 2, 2, 1, 8, 4>(pl, s, d);
        case CFG_W26_M64_N4: case CFG_W26_M64_N4_K2: case CFG_W26_M64_N4_K4: case CFG_W26_M64_N4_K8: return launch_dma<16, 2, 13, 2, 4, 2>(pl, s, d);
        case CFG_W26_M128_N2: case CFG_W26_M128_N2_K2: case CFG_W26_M128_N2_K4: case CFG_W26_M128_N2_K8: return launch_dma<16, 2, 13, 4, 2, 2>(pl, s, d);
        case CFG_M64_S_C16: return launch_cfg<32, 1, 1, 2, 2, 16>(pl, s, d);
        case CFG_M64_S_C32: return launch_cfg<32, 1, 1, 2, 2, 32>(pl, s, d);
        case CFG_M128_S_C16: return launch_cfg<32, 2, 1, 2, 2, 16>(pl, s, d);
        case CFG_M128_S_C32: return launch_cfg<32, 2, 1, 2, 2, 32>(pl, s, d);
        case CFG_M32_S_C16: return launch_cfg<32, 1, 1, 1, 4, 16>(pl, s, d);
        case CFG_M64_S_C4: return launch_cfg<32, 1, 1, 2, 2, 4>(pl, s, d);
        case CFG_M128_S_C4: return launch_cfg<32, 2, 1, 2, 2, 4>(pl, s, d);
        case CFG_M32_S_C4: return launch_cfg<32, 1, 1, 1, 4, 4>(pl, s, d);
        case CFG_D16_C4: return launch_dma<16, 1, 4, 1, 4, 1>(pl, s, d);
        case CFG_D16_C4_L: return launch_dma<16, 1, 8, 1, 4, 1>(pl, s, d);
        case CFG_D16_C16: return launch_dma<16, 1, 4, 1, 4, 4>(pl, s, d);
        case CFG_D16_C16_L: return launch_dma<16, 1, 8, 1, 4, 4>(pl, s, d);
        case CFG_D16_C8: return launch_dma<16, 1, 4, 1, 4, 2>(pl, s, d);
        case CFG_D16_C8_L: return launch_dma<16, 1, 8, 1, 4, 2>(pl, s, d);
        case CFG_M128_L: return launch_cfg<32, 4, 2, 1, 4, 8>(pl, s, d);
        case CFG_M128: return launch_cfg<32, 4, 1, 1, 4, 8>(pl, s, d);
        case CFG_M128_S: return launch_cfg<32, 2, 1, 2, 2, 8>(pl, s, d);
        case CFG_M64: return launch_cfg<32, 2, 2, 1, 4, 8>(pl, s, d);
        case CFG_M64_S: return launch_cfg<32, 1, 1, 2, 2, 8>(pl, s, d);
        case CFG_M32: return launch_cfg<32, 1, 4, 1, 4, 8>(pl, s, d);
        case CFG_M32_S: return launch_cfg<32, 1, 1, 1, 4, 8>(pl, s, d);
        case CFG_M16_CK4: return launch_cfg<16, 1, 4, 1, 4, 4>(pl, s, d);
        case CFG_M16_CK4_L: return launch_cfg<16, 1, 8, 1, 4, 4>(pl, s, d);
        case CFG_M16_CK8: return launch_cfg<16, 1, 4, 1, 4, 8>(pl, s, d);
        case CFG_M16_CK8_L: return launch_cfg<16, 1, 8, 1, 4, 8>(pl, s, d);
        default: return launch_cfg<16, 1, 4, 1, 4, 16>(pl, s, d);
    }
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
bool direct_conv_supported(const cnn_conv2d_desc* d);  // conv_direct.hip: thin first layers bypass the implicit GEMM
int direct_conv_forward(const cnn_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y, float* y_relu,
                        void* ws, size_t ws_bytes, hipStream_t s, bool prepared);
int direct_conv_dgrad(const cnn_conv2d_desc* d, const float* dy, const float* w, float* dx, void* ws, size_t ws_bytes,
                      hipStream_t s, bool prepared);
int direct_prepare_batch(int n, const cnn_conv2d_desc* descs, const float* const* w, const float* const* bias,
                         void* const* fwd, void* const* dgrad, hipStream_t s, unsigned* fwd_done, unsigned* dgrad_done);
bool direct_prepared_fwd_ok(const cnn_conv2d_desc* d);
bool direct_prepared_dgrad_ok(const cnn_conv2d_desc* d);
bool direct_conv_pool_supported(const cnn_conv2d_desc* d);  // conv_direct.hip: Conv -> ReLU -> MaxPool(2,2) in one kernel
size_t direct_pool_mask_bytes(const cnn_conv2d_desc* d);
int direct_pool_mask_unpack(const cnn_conv2d_desc* d, const void* packed, int32_t* mask, hipStream_t s);
int direct_conv_pool_forward(const cnn_conv2d_desc* d, const float* x, const float* w, const float* bias, float* pooled,
                             int32_t* mask, void* ws, size_t ws_bytes, hipStream_t s, bool prepared);
int direct_conv_dgrad_pooled(const cnn_conv2d_desc* d, const float* dpool, const int32_t* mask, const float* pooled, const float* w,
                             float* dx, void* ws, size_t ws_bytes, hipStream_t s, bool prepared);
bool dgrad_rd_supported(const cnn_conv2d_desc* d);  // conv_dgrad_rd.hip: register-direct data gradient, 3x3 stride 2, Co 64 / 128
size_t dgrad_rd_prepared_floats(const cnn_conv2d_desc* d);
int dgrad_rd_backward_data(const cnn_conv2d_desc* d, const float* dy, const float* w, const float* img, const float* relu_below,
                           float* dx, void* ws, size_t ws_bytes, hipStream_t s);
size_t rows_workspace_floats(const cnn_conv2d_desc* d, int mode);  // conv_rows.hip (round 5): LDS-staged 3x3 / stride-1 forward and data gradient
int rows_prepare(const cnn_conv2d_desc* d, int mode, const float* w, float* image, hipStream_t s);
int rows_prepare_batch(int n, const cnn_conv2d_desc* const* d, const int* mode, const float* const* w, float* const* image, hipStream_t s);                                   // of wide planes
int rows_run(const cnn_conv2d_desc* d, int mode, const float* in, const float* image, const float* bias, float* out, float* out_relu,
             const float* relu_below, hipStream_t s);
bool fwd_rd_supported(const cnn_conv2d_desc* d);  // conv_fwd_rd.hip: register-direct forward of the mid-size 3x3 layers
size_t fwd_rd_prepared_floats(const cnn_conv2d_desc* d);
int rd_prepare_batch(int n, const cnn_conv2d_desc* descs, const float* const* w, const float* const* bias, void* const* fwd,
                     void* const* dgrad, hipStream_t s, unsigned* fdone, unsigned* ddone);
int fwd_rd_forward(const cnn_conv2d_desc* d, const float* x, const float* w, const float* img, const float* bias, float* y,
                   float* y_relu, hipStream_t s);
bool fwd_rd_small(const cnn_conv2d_desc* d);           // conv_fwd_rd.hip: the small-layer kernel (never replaced)
bool stem_fwd_supported(const cnn_conv2d_desc* d);     // conv_stem.hip: Ci = 3, 7x7, stride 2, pad 3 forward on its own MFMA kernel
int stem_forward(const cnn_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y, float* y_relu, hipStream_t s);
bool thin_dgrad_supported(const cnn_conv2d_desc* d);   // conv_dgrad_thin.hip: VALU data gradient of thin (Ci = 3) stride-1 layers
int thin_dgrad(const cnn_conv2d_desc* d, const float* dy, const float* w, const float* packed, const float* relu_below, float* dx, hipStream_t s);
size_t thin_dgrad_packed_floats(const cnn_conv2d_desc* d);  // > 0: the layer's data gradient reads a packed filter image (the 7x7 stem) ...
int thin_dgrad_pack(const cnn_conv2d_desc* d, const float* w, float* image, hipStream_t s);  // ... made by this
bool c11_supported(const cnn_conv2d_desc* d);          // conv_1x1.hip: 1x1 convolutions (stride 1 / 2) as plain LDS-tiled GEMMs
int c11_forward(const cnn_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y, float* y_relu, hipStream_t s);
int c11_backward_data(const cnn_conv2d_desc* d, const float* dy, const float* w, const float* relu_below, float* dx, hipStream_t s);
bool pk_dgrad_s2_supported(const cnn_conv2d_desc* d);  // conv_direct.hip: packed VALU dgrad for small stride-2 layers
size_t pk_dgrad_s2_workspace_floats(const cnn_conv2d_desc* d);
int pk_dgrad_s2(const cnn_conv2d_desc* d, const float* dy, const float* w, float* dx, void* ws, hipStream_t s, bool prepared,
                const float* relu_below);
// scratch floats the forward / dgrad plans need (used by cnn_conv2d_workspace_bytes in conv_wgrad.hip)
bool igemm_preferred(const cnn_conv2d_desc* d, int mode) {
    std::lock_guard<std::mutex> lk(tune_mutex());
    auto it = prefer_table().find(tune_key(d, mode));
    return it != prefer_table().end() && it->second;
}

size_t igemm_workspace_floats(const cnn_conv2d_desc* d) {
    static thread_local DescMemo memo;  // (independent of what the tuner has pinned: every candidate is planned regardless)
    size_t n = 0;
    if (memo.find(d, &n)) return n;
    // the largest re-arranged filter image any tile the tuner may pin would need (the caller sizes its buffers once)
    for (int c : kTuneCandidates)
        for (int mode = 0; mode < 2; ++mode) {
            Plan pl;
            g_forced_cfg = c;
            const int rc = make_plan("ws", d, mode, &pl);
            g_forced_cfg = -1;
            if (rc != CNN_AMD_OK || (c >= 0 && pl.cfg != c)) continue;
            const size_t need = pl.ksplit > 1 ? (pl.a_floats + 63) / 64 * 64 + (size_t)pl.ksplit * (size_t)pl.p.zstride : pl.a_floats;
            if (need > n) n = need;
        }
    if (direct_conv_supported(d) && n < 1024) n = 1024;  // packed filter copies of the direct kernels (conv_direct.hip)
    if (c11_supported(d) && n < (size_t)d->Co * d->Ci) n = (size_t)d->Co * d->Ci;  // (conv_1x1.hip: the prepared image is a verbatim copy)
    if (pk_dgrad_s2_supported(d) && n < pk_dgrad_s2_workspace_floats(d)) n = pk_dgrad_s2_workspace_floats(d);
    if (fwd_rd_prepared_floats(d) > n) n = fwd_rd_prepared_floats(d);
    if (dgrad_rd_prepared_floats(d) > n) n = dgrad_rd_prepared_floats(d);
    if (thin_dgrad_supported(d) && thin_dgrad_packed_floats(d) > n) n = thin_dgrad_packed_floats(d);
    for (int mode = 0; mode < 2; ++mode)
        if (rows_workspace_floats(d, mode) > n) n = rows_workspace_floats(d, mode);
    memo.put(d, n);
    return n;
}
}  // namespace cnn_amd

extern "C" {

static int conv2d_forward_impl(const char* who, const cnn_conv2d_desc* d, const float* x, const float* w, const float* bias,
                               float* y, float* y_relu, void* ws, size_t ws_bytes, void* stream, bool prepared = false) {
    if (int rc = check_desc(who, d)) return rc;
    // y may be NULL when only the ReLU output is wanted and the layer runs on the register-direct forward kernel
    CNN_REQUIRE(x && (w || prepared) && bias && (y || (y_relu && !direct_conv_supported(d) && !stem_fwd_supported(d))), "%s: null pointer", who);
    if (direct_conv_supported(d)) return direct_conv_forward(d, x, w, bias, y, y_relu, ws, ws_bytes, as_stream(stream), prepared);
    if (c11_supported(d))  // (its "prepared" image is a verbatim copy of w)
        return c11_forward(d, x, prepared ? (const float*)ws : w, bias, y, y_relu, as_stream(stream));
    // (round 5) wide 3x3 / stride-1 layers: the row kernel (conv_rows.hip); the workspace / prepared buffer holds its filter image
    if (rows_workspace_floats(d, MODE_FWD) > 0 && ws != nullptr && ws_bytes >= rows_workspace_floats(d, MODE_FWD) * sizeof(float) &&
        (reinterpret_cast<uintptr_t>(ws) & 15) == 0) {
        if (!prepared)
            if (int rc = rows_prepare(d, MODE_FWD, w, (float*)ws, as_stream(stream))) return rc;
        return rows_run(d, MODE_FWD, x, (const float*)ws, bias, y, y_relu, nullptr, as_stream(stream));
    }
    if (fwd_rd_supported(d))
        return fwd_rd_forward(d, x, prepared ? nullptr : w, prepared ? (const float*)ws : nullptr, bias, y, y_relu, as_stream(stream));
    if (stem_fwd_supported(d) && (y || y_relu))  // (its "prepared" image is a verbatim copy of w)
        return stem_forward(d, x, prepared ? (const float*)ws : w, bias, y, y_relu, as_stream(stream));
    Plan pl;
    if (int rc = make_plan(who, d, MODE_FWD, &pl)) return rc;
    return run_plan(pl, d, x, w, bias, y, y_relu, ws, ws_bytes, as_stream(stream), who, prepared);
}

// relu_below (nullable): output of the ReLU layer whose input gradient dx is -- fuses that layer's backward pass
static int conv2d_backward_data_impl(const char* who, const cnn_conv2d_desc* d, const float* dy, const float* w, float* dx,
                                     void* ws, size_t ws_bytes, void* stream, bool prepared, const float* relu_below = nullptr) {
    if (int rc = check_desc(who, d)) return rc;
    CNN_REQUIRE(dy && (w || prepared) && dx, "%s: null pointer", who);
    if (direct_conv_supported(d)) {  // the first-layer kernels have no masked epilogue: same result from the ReLU kernel
        const int rc = direct_conv_dgrad(d, dy, w, dx, ws, ws_bytes, as_stream(stream), prepared);
        if (rc || !relu_below) return rc;
        return cnn_relu_backward(relu_below, dx, (size_t)d->B * d->Ci * d->H * d->W, stream);
    }
    if (c11_supported(d))  // (its "prepared" image is a verbatim copy of w)
        return c11_backward_data(d, dy, prepared ? (const float*)ws : w, relu_below, dx, as_stream(stream));
    if (thin_dgrad_supported(d)) {
        const size_t pk = thin_dgrad_packed_floats(d);
        if (pk == 0)  // (the "prepared" image of such a layer is a verbatim copy of w)
            return thin_dgrad(d, dy, prepared ? (const float*)ws : w, nullptr, relu_below, dx, as_stream(stream));
        if (prepared) return thin_dgrad(d, dy, nullptr, (const float*)ws, relu_below, dx, as_stream(stream));  // (prepared image = the packed one)
        if (ws == nullptr || ws_bytes < pk * sizeof(float))
            return thin_dgrad(d, dy, w, nullptr, relu_below, dx, as_stream(stream));  // (no room for the packed image: the scalar-operand kernel)
        if (int rc = thin_dgrad_pack(d, w, (float*)ws, as_stream(stream))) return rc;
        return thin_dgrad(d, dy, nullptr, (const float*)ws, relu_below, dx, as_stream(stream));
    }
    if (rows_workspace_floats(d, MODE_DGRAD) > 0 && ws != nullptr && ws_bytes >= rows_workspace_floats(d, MODE_DGRAD) * sizeof(float) &&
        (reinterpret_cast<uintptr_t>(ws) & 15) == 0) {
        if (!prepared)
            if (int rc = rows_prepare(d, MODE_DGRAD, w, (float*)ws, as_stream(stream))) return rc;
        return rows_run(d, MODE_DGRAD, dy, (const float*)ws, nullptr, dx, nullptr, relu_below, as_stream(stream));
    }
    if (dgrad_rd_supported(d))
        return dgrad_rd_backward_data(d, dy, prepared ? nullptr : w, prepared ? (const float*)ws : nullptr, relu_below, dx,
                                      prepared ? nullptr : ws, prepared ? 0 : ws_bytes, as_stream(stream));
    if (pk_dgrad_s2_supported(d) && ws != nullptr && ws_bytes >= pk_dgrad_s2_workspace_floats(d) * sizeof(float))
        return pk_dgrad_s2(d, dy, w, dx, ws, as_stream(stream), prepared, relu_below);
    Plan pl;
    if (int rc = make_plan(who, d, MODE_DGRAD, &pl)) return rc;
    return run_plan(pl, d, dy, w, nullptr, dx, const_cast<float*>(relu_below), ws, ws_bytes, as_stream(stream), who, prepared);
}

// floats of scratch one measurement needs: x, y, w, the implicit GEMM's workspace / prepared image, bias + per-channel scratch
static size_t autotune_scratch_floats(const cnn_conv2d_desc* d, size_t* nx, size_t* ny, size_t* nw, size_t* na) {
    const int Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, d->pad), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, d->pad);
    *nx = ((size_t)d->B * d->Ci * d->H * d->W + 63) / 64 * 64;
    *ny = ((size_t)d->B * d->Co * Ho * Wo + 63) / 64 * 64;
    *nw = ((size_t)d->Co * d->Ci * d->k * d->k + 63) / 64 * 64;
    *na = (igemm_workspace_floats(d) + 64 + 63) / 64 * 64;
    return *nx + *ny + *nw + *na + ((size_t)(d->Co + d->Ci) + 63) / 64 * 64;
}

// does cnn_conv2d_autotune measure anything for this geometry in this mode (the specialised kernels keep their layers; a geometry is
// measured once per process)?
static bool autotune_applies(const cnn_conv2d_desc* d, int mode) {
    if (direct_conv_supported(d) || c11_supported(d)) return false;
    if (rows_workspace_floats(d, mode) > 0) return false;  // (conv_rows.hip has no tile to choose)
    const bool rd_fwd = mode == MODE_FWD && fwd_rd_supported(d);
    const bool rd_dgrad = mode == MODE_DGRAD && dgrad_rd_supported(d);
    if (rd_fwd && fwd_rd_small(d)) return false;
    if (mode == MODE_FWD && !rd_fwd && stem_fwd_supported(d)) return false;
    if (rd_dgrad && d->s != 1) return false;
    if (mode == MODE_DGRAD && !rd_dgrad && (pk_dgrad_s2_supported(d) || thin_dgrad_supported(d))) return false;
    std::lock_guard<std::mutex> lk(tune_mutex());
    return tune_table().count(tune_key(d, mode)) == 0;
}
static bool autotune_enabled() {
    if (const OptVal e = CNN_OPT_VAL("IGEMM_AUTOTUNE"))
        if (atoi(e) == 0) return false;
    return !CNN_OPT_SET("IGEMM_CFG");
}

// 0: nothing to measure for this geometry (cnn_conv2d_autotune_ws then needs no scratch and returns at once)
size_t cnn_conv2d_autotune_workspace_bytes(const cnn_conv2d_desc* d) {
    if (check_desc("cnn_conv2d_autotune_workspace_bytes", d)) return 0;
    if (!autotune_enabled() || (!autotune_applies(d, MODE_FWD) && !autotune_applies(d, MODE_DGRAD))) return 0;
    size_t nx, ny, nw, na;
    return autotune_scratch_floats(d, &nx, &ny, &nw, &na) * sizeof(float);
}

// scratch == NULL: buffers of the library's own for the duration of the call (the older entry point, cnn_conv2d_autotune)
static int autotune_impl(const cnn_conv2d_desc* d, void* scratch, size_t scratch_bytes, void* stream) {
    if (int rc = check_desc("cnn_conv2d_autotune", d)) return rc;
    if (const OptVal e = CNN_OPT_VAL("IGEMM_AUTOTUNE"))
        if (atoi(e) == 0) return CNN_AMD_OK;
    if (CNN_OPT_SET("IGEMM_CFG")) return CNN_AMD_OK;
    hipStream_t s = as_stream(stream);
    for (int mode = 0; mode < 2; ++mode) {
        // geometries that never reach the implicit GEMM in this mode
        if (direct_conv_supported(d) || c11_supported(d)) continue;
        if (rows_workspace_floats(d, mode) > 0) continue;  // (conv_rows.hip has no tile to choose)
        // register-direct kernels for BIG layers (stride-1 / stride-2 3x3, pad 0, Ci <= 64; stride-1 data gradient) are measured
        // against the implicit GEMM below; the small-layer kernels, the first-layer kernels and the stem keep their layers
        const bool rd_fwd = mode == MODE_FWD && fwd_rd_supported(d);
        const bool rd_dgrad = mode == MODE_DGRAD && dgrad_rd_supported(d);
        if (rd_fwd && fwd_rd_small(d)) continue;
        if (mode == MODE_FWD && !rd_fwd && stem_fwd_supported(d)) continue;
        if (rd_dgrad && d->s != 1) continue;
        if (mode == MODE_DGRAD && !rd_dgrad && (pk_dgrad_s2_supported(d) || thin_dgrad_supported(d))) continue;
        {
            std::lock_guard<std::mutex> lk(tune_mutex());
            if (tune_table().count(tune_key(d, mode))) continue;
        }
        size_t nx, ny, nw, na;
        const size_t total = autotune_scratch_floats(d, &nx, &ny, &nw, &na);
        float *bx = nullptr, *by = nullptr, *bw = nullptr, *ba = nullptr, *bb = nullptr;
        float* own = nullptr;
        auto release = [&]() {
            if (own) (void)hipFree(own);
        };
        if (scratch != nullptr) {
            CNN_REQUIRE(scratch_bytes >= total * sizeof(float), "cnn_conv2d_autotune_ws: scratch %zu B < %zu B", scratch_bytes, total * sizeof(float));
            bx = (float*)scratch;
        } else {
            if (hipMalloc(&own, total * sizeof(float)) != hipSuccess) {
                (void)hipGetLastError();
                return CNN_AMD_OK;  // no room to measure: keep the rule-based choice
            }
            bx = own;
        }
        by = bx + nx; bw = by + ny; ba = bw + nw; bb = ba + na;
        // (zeros: MFMA / LDS / DMA timing does not depend on the values)
        (void)hipMemsetAsync(bx, 0, nx * 4, s); (void)hipMemsetAsync(by, 0, ny * 4, s); (void)hipMemsetAsync(bw, 0, nw * 4, s);
        (void)hipMemsetAsync(bb, 0, (size_t)(d->Co + d->Ci) * 4, s);
        const float* X = mode == MODE_FWD ? bx : by;  // forward reads x, the data gradient reads dy
        float* Y = mode == MODE_FWD ? by : bx;
        // (round 5) the data gradient is measured WITH the fused ReLU' epilogue: in a train step nearly every data gradient carries the
        // mask of the layer below, and its loads move the ranking (64 x 256 -> 256 at 14x14: the 64-wide tile 197 us plain, 220 us masked;
        // the split-K wide tile 198 us either way).  The output tensor doubles as the mask: every element is read, then written, by the
        // same lane (igemm epilogue, split_reduce).  TUNE_MASKED=0: the plain epilogue as before.
        float* Ymask = mode == MODE_DGRAD && CNN_OPT_INT("TUNE_MASKED", 1) != 0 ? Y : nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {  // (nothing to measure with: keep the rules)
            (void)hipGetLastError();
            if (e0) (void)hipEventDestroy(e0);
            release();
            return CNN_AMD_OK;
        }
        float rd_ms = 1e30f;
        if (rd_fwd || rd_dgrad) {
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(e0, s);
                const int rc = rd_fwd ? fwd_rd_forward(d, X, bw, nullptr, bb, Y, nullptr, s)
                                      : dgrad_rd_backward_data(d, X, bw, nullptr, nullptr, Y, ba, na * 4, s);
                (void)hipEventRecord(e1, s);
                float t = 1e30f;
                if (rc == CNN_AMD_OK && hipEventSynchronize(e1) == hipSuccess) (void)hipEventElapsedTime(&t, e0, e1);
                else (void)hipGetLastError();
                if (rep > 0 && t < rd_ms) rd_ms = t;  // (best of three behind a warm-up, like the candidates below)
            }
        }
        int best = -1;
        float best_ms = 1e30f;
        const float margin = 1.f - 0.001f * (float)CNN_OPT_INT("TUNE_MARGIN", 0);  // (per mille a later candidate has to win by: 30 until the timings became best-of-three; VGG-shaped step 2 156-2 161 at 30, 2 165-2 172 at 0)
        const int excluded = CNN_OPT_INT("TUNE_EXCLUDE", -2);  // (measurement switch: one candidate the tuner must not pick)
        for (int c : kTuneCandidates) {
            if (c == excluded) continue;
            Plan pl;
            g_forced_cfg = c;
            int rc = make_plan("cnn_conv2d_autotune", d, mode, &pl);
            if (rc == CNN_AMD_OK && c >= 0 && pl.cfg != c) rc = CNN_AMD_E_BADARG;  // not applicable to this geometry
            float ms = 1e30f;
            if (rc == CNN_AMD_OK) {
                rc = run_plan(pl, d, X, bw, bb, Y, Ymask, ba, na * 4, s, "cnn_conv2d_autotune");  // warm-up (first-use setup, filter image)
                // best of three runs on the filter image the warm-up left behind (a train step prepares the images apart from the
                // convolutions; one run each used to decide 3 - 5 % differences by the box's noise)
                for (int rep = 0; rep < 3 && rc == CNN_AMD_OK; ++rep) {
                    (void)hipEventRecord(e0, s);
                    rc = run_plan(pl, d, X, bw, bb, Y, Ymask, ba, na * 4, s, "cnn_conv2d_autotune", true);
                    (void)hipEventRecord(e1, s);
                    float t = 1e30f;
                    if (rc == CNN_AMD_OK && hipEventSynchronize(e1) == hipSuccess) (void)hipEventElapsedTime(&t, e0, e1);
                    if (t < ms) ms = t;
                }
            }
            g_forced_cfg = -1;
            if (rc != CNN_AMD_OK) {
                (void)hipGetLastError();
                continue;
            }
            // the rule-based default keeps its place unless something is clearly faster (noise: a few per cent)
            if (best_ms > 1e29f || ms < best_ms * margin) {
                best = c;
                best_ms = ms;
            }
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        release();
        if (rd_fwd || rd_dgrad) {
            // the register-direct kernel keeps its layer unless the implicit GEMM is clearly faster
            std::lock_guard<std::mutex> lk(tune_mutex());
            prefer_table()[tune_key(d, mode)] = best_ms < rd_ms * 0.97f;
        }
        if (best >= 0) {
            std::lock_guard<std::mutex> lk(tune_mutex());
            tune_table()[tune_key(d, mode)] = best;
        } else {
            std::lock_guard<std::mutex> lk(tune_mutex());
            tune_table()[tune_key(d, mode)] = -1;  // measured: the default stays
        }
    }
    return CNN_AMD_OK;
}

int cnn_conv2d_autotune(const cnn_conv2d_desc* d, void* stream) { return autotune_impl(d, nullptr, 0, stream); }

int cnn_conv2d_autotune_ws(const cnn_conv2d_desc* d, void* scratch, size_t scratch_bytes, void* stream) {
    if (int rc = check_desc("cnn_conv2d_autotune_ws", d)) return rc;
    if (cnn_conv2d_autotune_workspace_bytes(d) == 0) return CNN_AMD_OK;  // (nothing to measure)
    CNN_REQUIRE(scratch != nullptr, "cnn_conv2d_autotune_ws: null scratch");
    return autotune_impl(d, scratch, scratch_bytes, stream);
}

// what cnn_conv2d_autotune pinned for this geometry, as four integers a caller can ship to other processes:
// [0] forward tile, [1] data-gradient tile (-1: the rule-based default was measured best; CNN_TUNE_NONE: never measured),
// [2] / [3] the implicit GEMM replaces the register-direct forward / data-gradient kernel (0 / 1; CNN_TUNE_NONE: never measured)
int cnn_conv2d_tune_export(const cnn_conv2d_desc* d, int32_t out[4]) {
    if (int rc = check_desc("cnn_conv2d_tune_export", d)) return rc;
    CNN_REQUIRE(out != nullptr, "cnn_conv2d_tune_export: null pointer");
    std::lock_guard<std::mutex> lk(tune_mutex());
    for (int mode = 0; mode < 2; ++mode) {
        const auto t = tune_table().find(tune_key(d, mode));
        out[mode] = t == tune_table().end() ? CNN_TUNE_NONE : t->second;
        const auto p = prefer_table().find(tune_key(d, mode));
        out[2 + mode] = p == prefer_table().end() ? CNN_TUNE_NONE : (p->second ? 1 : 0);
    }
    return CNN_AMD_OK;
}

// pins another process' choices for this geometry (replicas of a data-parallel job then run the same kernels: bit-identical local
// arithmetic on every rank); entries equal to CNN_TUNE_NONE leave the table alone.  Call it where cnn_conv2d_autotune would be called.
int cnn_conv2d_tune_import(const cnn_conv2d_desc* d, const int32_t in[4]) {
    if (int rc = check_desc("cnn_conv2d_tune_import", d)) return rc;
    CNN_REQUIRE(in != nullptr, "cnn_conv2d_tune_import: null pointer");
    std::lock_guard<std::mutex> lk(tune_mutex());
    for (int mode = 0; mode < 2; ++mode) {
        if (in[mode] != CNN_TUNE_NONE) tune_table()[tune_key(d, mode)] = in[mode];
        if (in[2 + mode] != CNN_TUNE_NONE) prefer_table()[tune_key(d, mode)] = in[2 + mode] != 0;
    }
    return CNN_AMD_OK;
}

int cnn_conv2d_forward(const cnn_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y,
                       void* ws, size_t ws_bytes, void* stream) {
    return conv2d_forward_impl("cnn_conv2d_forward", d, x, w, bias, y, nullptr, ws, ws_bytes, stream);
}

int cnn_conv2d_forward_relu(const cnn_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y,
                            float* y_relu, void* ws, size_t ws_bytes, void* stream) {
    CNN_REQUIRE(y_relu, "cnn_conv2d_forward_relu: null pointer");
    return conv2d_forward_impl("cnn_conv2d_forward_relu", d, x, w, bias, y, y_relu, ws, ws_bytes, stream);
}

int cnn_conv2d_backward_data(const cnn_conv2d_desc* d, const float* dy, const float* w, float* dx, void* ws,
                             size_t ws_bytes, void* stream) {
    return conv2d_backward_data_impl("cnn_conv2d_backward_data", d, dy, w, dx, ws, ws_bytes, stream, false);
}

int cnn_conv2d_relu_only_supported(const cnn_conv2d_desc* d) {
    if (check_desc("cnn_conv2d_relu_only_supported", d)) return 0;
    // (round 4: the implicit GEMM's fused-ReLU epilogue too -- every kernel family behind cnn_conv2d_forward except the thin first layers')
    return (!direct_conv_supported(d) && !stem_fwd_supported(d)) ? 1 : 0;
}

/* ---- Conv2D -> ReLU -> MaxPool2D(2,2) ---- */
int cnn_conv2d_relu_maxpool2_supported(const cnn_conv2d_desc* d) {
    if (check_desc("cnn_conv2d_relu_maxpool2_supported", d)) return 0;
    return direct_conv_pool_supported(d) ? 1 : 0;
}

int cnn_conv2d_pool_mask_packed_supported(const cnn_conv2d_desc* d) {
    if (check_desc("cnn_conv2d_pool_mask_packed_supported", d)) return 0;
    return direct_pool_mask_packed_ok(d) ? 1 : 0;
}
size_t cnn_conv2d_pool_mask_bytes(const cnn_conv2d_desc* d) {
    if (check_desc("cnn_conv2d_pool_mask_bytes", d)) return 0;
    return direct_pool_mask_bytes(d);
}
int cnn_conv2d_pool_mask_unpack(const cnn_conv2d_desc* d, const void* packed, int32_t* mask, void* stream) {
    if (int rc = check_desc("cnn_conv2d_pool_mask_unpack", d)) return rc;
    CNN_REQUIRE(packed && mask, "cnn_conv2d_pool_mask_unpack: null pointer");
    return direct_pool_mask_unpack(d, packed, mask, as_stream(stream));
}

int cnn_conv2d_relu_maxpool2_forward(const cnn_conv2d_desc* d, const float* x, const float* w, const float* bias, float* pooled,
                                     int32_t* mask, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_desc("cnn_conv2d_relu_maxpool2_forward", d)) return rc;
    CNN_REQUIRE(x && w && bias && pooled, "cnn_conv2d_relu_maxpool2_forward: null pointer");
    return direct_conv_pool_forward(d, x, w, bias, pooled, mask, ws, ws_bytes, as_stream(stream), false);
}

int cnn_conv2d_relu_maxpool2_forward_prepared(const cnn_conv2d_desc* d, const float* x, const void* prepared_fwd, float* pooled,
                                              int32_t* mask, void* stream) {
    if (int rc = check_desc("cnn_conv2d_relu_maxpool2_forward_prepared", d)) return rc;
    CNN_REQUIRE(x && prepared_fwd && pooled, "cnn_conv2d_relu_maxpool2_forward_prepared: null pointer");
    return direct_conv_pool_forward(d, x, nullptr, nullptr, pooled, mask, (void*)prepared_fwd, cnn_conv2d_prepared_bytes(d),
                                    as_stream(stream), true);
}

int cnn_conv2d_backward_data_pooled2(const cnn_conv2d_desc* d, const float* dpool, const int32_t* mask, const float* pooled,
                                     const float* w, float* dx, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_desc("cnn_conv2d_backward_data_pooled2", d)) return rc;
    CNN_REQUIRE(dpool && mask && w && dx, "cnn_conv2d_backward_data_pooled2: null pointer");
    return direct_conv_dgrad_pooled(d, dpool, mask, pooled, w, dx, ws, ws_bytes, as_stream(stream), false);
}

int cnn_conv2d_backward_data_pooled2_prepared(const cnn_conv2d_desc* d, const float* dpool, const int32_t* mask, const float* pooled,
                                              const void* prepared_dgrad, float* dx, void* stream) {
    if (int rc = check_desc("cnn_conv2d_backward_data_pooled2_prepared", d)) return rc;
    CNN_REQUIRE(dpool && mask && prepared_dgrad && dx, "cnn_conv2d_backward_data_pooled2_prepared: null pointer");
    return direct_conv_dgrad_pooled(d, dpool, mask, pooled, nullptr, dx, (void*)prepared_dgrad, cnn_conv2d_prepared_bytes(d),
                                    as_stream(stream), true);
}

/* ---- filter preparation hoisted out of the per-layer calls ---- */
size_t cnn_conv2d_prepared_bytes(const cnn_conv2d_desc* d) {
    if (check_desc("cnn_conv2d_prepared_bytes", d)) return 0;
    return (igemm_workspace_floats(d) + 64) * sizeof(float);
}

int cnn_conv2d_prepare_filters(int n, const cnn_conv2d_desc* descs, const float* const* w, const float* const* bias,
                               void* const* fwd, void* const* dgrad, void* stream) {
    CNN_REQUIRE(n > 0 && n <= 6 && descs && w && bias, "cnn_conv2d_prepare_filters: n=%d (1..6 layers per call)", n);
    hipStream_t s = as_stream(stream);
    unsigned fdone = 0, ddone = 0;
    if (int rc = direct_prepare_batch(n, descs, w, bias, fwd, dgrad, s, &fdone, &ddone)) return rc;
    {  // (round 5) the row kernel's layers: the per-layer entry points look for it before the register-direct kernels too (direct_prepare_batch resets the masks: it goes first); one launch for all of them
        const cnn_conv2d_desc* rd[12];
        int rmode[12], rn = 0;
        const float* rw[12];
        float* rimg[12];
        for (int i = 0; i < n; ++i) {
            if (check_desc("cnn_conv2d_prepare_filters", &descs[i]) || direct_conv_supported(&descs[i]) || c11_supported(&descs[i])) continue;
            for (int mode = 0; mode < 2; ++mode) {
                void* out = mode == MODE_FWD ? (fwd ? fwd[i] : nullptr) : (dgrad ? dgrad[i] : nullptr);
                if (!out || rows_workspace_floats(&descs[i], mode) == 0 || (reinterpret_cast<uintptr_t>(out) & 15) != 0) continue;
                CNN_REQUIRE(w[i] != nullptr, "cnn_conv2d_prepare_filters: filters of layer %d are null", i);
                rd[rn] = &descs[i]; rmode[rn] = mode; rw[rn] = w[i]; rimg[rn] = (float*)out; ++rn;
                (mode == MODE_FWD ? fdone : ddone) |= 1u << i;
            }
        }
        if (rn > 0)
            if (int rc = rows_prepare_batch(rn, rd, rmode, rw, rimg, s)) return rc;
    }
    if (int rc = rd_prepare_batch(n, descs, w, bias, fwd, dgrad, s, &fdone, &ddone)) return rc;
    PrepBatch pb;
    pb.n = 0;
    long long most = 0;
    for (int i = 0; i < n; ++i) {
        if (int rc = check_desc("cnn_conv2d_prepare_filters", &descs[i])) return rc;
        CNN_REQUIRE(w[i] != nullptr, "cnn_conv2d_prepare_filters: filters of layer %d are null", i);
        for (int mode = 0; mode < 2; ++mode) {
            void* out = mode == MODE_FWD ? (fwd ? fwd[i] : nullptr) : (dgrad ? dgrad[i] : nullptr);
            if (!out || ((mode == MODE_FWD ? fdone : ddone) >> i & 1u)) continue;
            if (!direct_conv_supported(&descs[i]) && c11_supported(&descs[i])) {
                // conv_1x1.hip reads the reference layout in both directions: its prepared images are verbatim copies
                CNN_HIP_CHECK(hipMemcpyAsync(out, w[i], sizeof(float) * (size_t)descs[i].Co * descs[i].Ci, hipMemcpyDeviceToDevice, s));
                continue;
            }
            if (mode == MODE_FWD && !direct_conv_supported(&descs[i]) && !fwd_rd_supported(&descs[i]) && stem_fwd_supported(&descs[i])) {
                // conv_stem.hip reads the reference layout: its prepared image is a verbatim copy
                CNN_HIP_CHECK(hipMemcpyAsync(out, w[i], sizeof(float) * (size_t)descs[i].Co * descs[i].Ci * descs[i].k * descs[i].k,
                                             hipMemcpyDeviceToDevice, s));
                continue;
            }
            if (mode == MODE_DGRAD && !direct_conv_supported(&descs[i]) && thin_dgrad_supported(&descs[i])) {
                if (thin_dgrad_packed_floats(&descs[i]) > 0) {  // the 7x7 stem: packed rows (conv_dgrad_thin_s2_pk7)
                    if (int rc = thin_dgrad_pack(&descs[i], w[i], (float*)out, s)) return rc;
                    continue;
                }
                // conv_dgrad_thin.hip reads the reference layout: its prepared image is a verbatim copy
                CNN_HIP_CHECK(hipMemcpyAsync(out, w[i], sizeof(float) * (size_t)descs[i].Co * descs[i].Ci * descs[i].k * descs[i].k,
                                             hipMemcpyDeviceToDevice, s));
                continue;
            }
            CNN_REQUIRE(!direct_conv_supported(&descs[i]) && !(mode == MODE_DGRAD && pk_dgrad_s2_supported(&descs[i]) && !dgrad_rd_supported(&descs[i])),
                        "cnn_conv2d_prepare_filters: layer %d has no prepared path for this mode", i);
            Plan pl;
            if (int rc = make_plan("cnn_conv2d_prepare_filters", &descs[i], mode, &pl)) return rc;
            pl.q.w = w[i];
            pl.q.A = (float*)out;
            pb.q[pb.n++] = pl.q;
            if ((long long)pl.a_floats > most) most = (long long)pl.a_floats;
        }
    }
    if (pb.n > 0) {
        unsigned pg = (unsigned)((most + 255) / 256);
        if (pg > 1024) pg = 1024;
        CNN_KLAUNCH(s, "igemm_prep_batch", (igemm_prep_batch<<<dim3(pg, pb.n), 256, 0, s>>>(pb)), "jobs=%d", pb.n);
    }
    return CNN_AMD_OK;
}

int cnn_conv2d_forward_prepared(const cnn_conv2d_desc* d, const float* x, const void* prepared_fwd, const float* bias, float* y,
                                float* y_relu, void* stream) {
    CNN_REQUIRE(prepared_fwd != nullptr, "cnn_conv2d_forward_prepared: null pointer");
    if (int rc = check_desc("cnn_conv2d_forward_prepared", d)) return rc;
    CNN_REQUIRE(!direct_conv_supported(d) || direct_prepared_fwd_ok(d), "cnn_conv2d_forward_prepared: no prepared path for this layer");
    return conv2d_forward_impl("cnn_conv2d_forward_prepared", d, x, nullptr, bias, y, y_relu, (void*)prepared_fwd,
                               cnn_conv2d_prepared_bytes(d), stream, true);
}

int cnn_conv2d_backward_data_relu(const cnn_conv2d_desc* d, const float* dy, const float* w, const float* relu_below, float* dx,
                                  void* ws, size_t ws_bytes, void* stream) {
    CNN_REQUIRE(relu_below, "cnn_conv2d_backward_data_relu: null pointer");
    return conv2d_backward_data_impl("cnn_conv2d_backward_data_relu", d, dy, w, dx, ws, ws_bytes, stream, false, relu_below);
}

int cnn_conv2d_backward_data_relu_prepared(const cnn_conv2d_desc* d, const float* dy, const void* prepared_dgrad,
                                           const float* relu_below, float* dx, void* stream) {
    CNN_REQUIRE(prepared_dgrad != nullptr && relu_below != nullptr, "cnn_conv2d_backward_data_relu_prepared: null pointer");
    if (int rc = check_desc("cnn_conv2d_backward_data_relu_prepared", d)) return rc;
    CNN_REQUIRE(!direct_conv_supported(d) || direct_prepared_dgrad_ok(d),
                "cnn_conv2d_backward_data_relu_prepared: no prepared path for this layer");
    return conv2d_backward_data_impl("cnn_conv2d_backward_data_relu_prepared", d, dy, nullptr, dx, (void*)prepared_dgrad,
                                     cnn_conv2d_prepared_bytes(d), stream, true, relu_below);
}

int cnn_conv2d_backward_data_prepared(const cnn_conv2d_desc* d, const float* dy, const void* prepared_dgrad, float* dx,
                                      void* stream) {
    CNN_REQUIRE(prepared_dgrad != nullptr, "cnn_conv2d_backward_data_prepared: null pointer");
    if (int rc = check_desc("cnn_conv2d_backward_data_prepared", d)) return rc;
    CNN_REQUIRE(!direct_conv_supported(d) || direct_prepared_dgrad_ok(d),
                "cnn_conv2d_backward_data_prepared: no prepared path for this layer");
    return conv2d_backward_data_impl("cnn_conv2d_backward_data_prepared", d, dy, nullptr, dx, (void*)prepared_dgrad,
                                     cnn_conv2d_prepared_bytes(d), stream, true);
}

}  // extern "C"
