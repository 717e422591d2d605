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

// MF: MFMA tile edge; MA x NB tiles per wave; WM x WN waves per workgroup; CK channels per LDS chunk.
template <int MF, int MA, int NB, int WM, int WN, int CK>
__global__ __launch_bounds__(64 * WM * WN) void igemm_kernel(const IgemmParams p) {
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

    const long long UV = (long long)p.U * p.V;
    const long long n0 = (long long)tile * NPIX;
    const long long n1 = (n0 + NPIX <= p.N ? n0 + NPIX : p.N) - 1;
    const int b0 = (int)(n0 / UV);
    const int u0 = (int)((n0 - b0 * UV) / p.V);
    const int b1 = (int)(n1 / UV);
    const int u1 = (int)((n1 - b1 * UV) / p.V);
    const int nseg = b1 - b0 + 1;
    const int nrows0 = ((nseg == 1 ? u1 : p.U - 1) - u0) * p.su + p.TR;
    const int full = (p.U - 1) * p.su + p.TR;
    const int nrows = (nseg == 1) ? nrows0 : nrows0 + (nseg - 2) * full + u1 * p.su + p.TR;

    // ---- one-time LDS setup: zero the row image (pad columns / out-of-image rows stay 0 for every chunk) ----
    if (p.need_zero)
        for (int i = tid; i < CK * p.chs; i += NT) Xs[i] = 0.f;
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

    // ---- per-lane pixel -> LDS offset of its window origin (B operand column = this lane's pixel) ----
    int pix_off[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        long long n = n0 + (wn * NB + nb) * MF + li;
        if (n > n1) n = n1;  // clamp: results of padded lanes are never stored
        const int b = (int)(n / UV);
        const int rem = (int)(n - b * UV);
        const int u = rem / p.V, v = rem - u * p.V;
        const int lrow = (b == b0) ? (u - u0) * p.su : nrows0 + (b - b0 - 1) * full + u * p.su;
        pix_off[nb] = lrow * p.LW + v * p.su + p.c0 + p.padL + lh * p.chs;
    }
    const int a_lane = lh * MT + wm * MA * MF + li;

    typename A_::type acc[MA][NB];
#pragma unroll
    for (int ma = 0; ma < MA; ++ma)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < A_::kRegs; ++r) acc[ma][nb][r] = 0.f;

    const float4* Ag = (const float4*)(p.A + (size_t)mb * p.nchunk * T * CK * MT);
    const int a_vec = T * CK * MT / 4;

    for (int cc = 0; cc < p.nchunk; ++cc) {
        __syncthreads();  // previous chunk's reads are done (and, first time, the zero fill / row table landed)
        // ---- stage the filter slab: one contiguous block, 4 independent 16-byte loads in flight per thread ----
        {
            const float4* src = Ag + (size_t)cc * a_vec;
            float4* dst = (float4*)As;
            for (int i0 = tid; i0 < a_vec; i0 += NT * 4) {
                float4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + u * NT < a_vec) t[u] = src[i0 + u * NT];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + u * NT < a_vec) dst[i0 + u * NT] = t[u];
            }
        }
        // ---- stage input rows: a wave covers 64 >> rw_shift rows per load instruction (lanes along the row),
        //      kUn such instructions are issued back to back before the first LDS store ----
        {
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
        __syncthreads();
        // ---- MFMA over taps x channel steps ----
        for (int tr = 0; tr < p.TR; ++tr) {
            for (int tc = 0; tc < p.TC; ++tc) {
                const int tap_off = tr * p.LW + tc;
                const float* a_tap = As + (tr * p.TC + tc) * CK * MT + a_lane;
#pragma unroll
                for (int c2 = 0; c2 < CK / KSTEP; ++c2) {
                    float a[MA], b[NB];
#pragma unroll
                    for (int ma = 0; ma < MA; ++ma) a[ma] = a_tap[c2 * KSTEP * MT + ma * MF];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) b[nb] = Xs[pix_off[nb] + tap_off + c2 * KSTEP * p.chs];
#pragma unroll
                    for (int ma = 0; ma < MA; ++ma)
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) acc[ma][nb] = A_::mfma(a[ma], b[nb], acc[ma][nb]);
                }
            }
        }
    }

    // ---- epilogue ----
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const long long n = n0 + (wn * NB + nb) * MF + li;
        if (n > n1) continue;
        const int b = (int)(n / UV);
        const int rem = (int)(n - b * UV);
        if (p.mode == MODE_FWD) {
            float* out = p.Y + (size_t)b * p.M * UV + rem;
#pragma unroll
            for (int ma = 0; ma < MA; ++ma) {
                const int mbase = mb * MT + (wm * MA + ma) * MF;
#pragma unroll
                for (int r = 0; r < A_::kRegs; ++r) {
                    const int m = mbase + A_::row(r, lh);
                    if (m < p.M) out[(size_t)m * UV] = acc[ma][nb][r] + (p.bias ? p.bias[m] : 0.f);
                }
            }
        } else {
            const int u = rem / p.V, v = rem - u * p.V;
#pragma unroll
            for (int ma = 0; ma < MA; ++ma) {
                const int mbase = mb * MT + (wm * MA + ma) * MF;
#pragma unroll
                for (int r = 0; r < A_::kRegs; ++r) {
                    const int m = mbase + A_::row(r, lh);
                    if (m < p.M) {
                        const int cls = m / p.c_out, ci = m - cls * p.c_out;
                        const int h = u * p.s_out + cls / p.s_out, w = v * p.s_out + cls % p.s_out;
                        if (h < p.OH && w < p.OW)
                            p.Y[(((size_t)b * p.c_out + ci) * p.OH + h) * p.OW + w] = acc[ma][nb][r];
                    }
                }
            }
        }
    }
}

// ---- weight preparation: [Co][Ci][k][k]  ->  A[mblock][chunk][tap][ck][MT] (zero padded) -------------------
struct PrepParams {
    const float* w;
    float* A;
    int Co, Ci, k, s, pad;
    int mode;
    int C, M, TR, TC, r0, c0;
    int CK, MT, nchunk, nmb;
};

__global__ void igemm_prep_weights(const PrepParams q) {
    const int T = q.TR * q.TC;
    const long long total = (long long)q.nmb * q.nchunk * T * q.CK * q.MT;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        long long r = idx;
        const int mm = (int)(r % q.MT); r /= q.MT;
        const int ck = (int)(r % q.CK); r /= q.CK;
        const int t = (int)(r % T); r /= T;
        const int cc = (int)(r % q.nchunk);
        const int mb = (int)(r / q.nchunk);
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

// ---- host-side planning ------------------------------------------------------------------------------------
struct Plan {
    int cfg;  // which instantiation
    int MT, NPIX, CK, MF;
    IgemmParams p;
    PrepParams q;
    size_t lds_bytes;
    size_t a_floats;
    unsigned grid_x, grid_y;
};

enum { CFG_M128 = 0, CFG_M128_S, CFG_M64, CFG_M64_S, CFG_M32, CFG_M32_S, CFG_M16_CK4, CFG_M16_CK16 };

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

int make_plan(const char* who, const cnn_conv2d_desc* d, int mode, Plan* pl) {
    const int Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, d->pad), Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, d->pad);
    CNN_REQUIRE(Ho > 0 && Wo > 0, "%s: empty output", who);
    IgemmParams& p = pl->p;
    PrepParams& q = pl->q;
    p = IgemmParams();
    q = PrepParams();
    p.B = d->B;
    p.mode = mode;
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
    CNN_REQUIRE(p.N / 16 < (1ll << 30), "%s: too many output pixels", who);

    // tile choice: the widest pixel tile that still gives every CU a couple of workgroups
    auto blocks_for = [&](int MT, int NPIX) { return ((p.N + NPIX - 1) / NPIX) * ((p.M + MT - 1) / MT); };
    const long long kWantBlocks = 2 * kNumCU;
    if (p.M > 64) {
        pl->MF = 32; pl->MT = 128; pl->CK = 8;
        if (blocks_for(128, 128) >= kWantBlocks) { pl->cfg = CFG_M128; pl->NPIX = 128; }
        else { pl->cfg = CFG_M128_S; pl->NPIX = 64; }
    } else if (p.M > 32) {
        pl->MF = 32; pl->MT = 64; pl->CK = 8;
        if (blocks_for(64, 256) >= kWantBlocks) { pl->cfg = CFG_M64; pl->NPIX = 256; }
        else { pl->cfg = CFG_M64_S; pl->NPIX = 64; }
    } else if (p.M > 16) {
        pl->MF = 32; pl->MT = 32; pl->CK = 8;
        if (blocks_for(32, 512) >= kWantBlocks) { pl->cfg = CFG_M32; pl->NPIX = 512; }
        else { pl->cfg = CFG_M32_S; pl->NPIX = 128; }
    } else if (p.C <= 4) { pl->cfg = CFG_M16_CK4; pl->MF = 16; pl->MT = 16; pl->NPIX = 256; pl->CK = 4; }
    else { pl->cfg = CFG_M16_CK16; pl->MF = 16; pl->MT = 16; pl->NPIX = 256; pl->CK = 16; }

    p.nchunk = (p.C + pl->CK - 1) / pl->CK;
    p.padL = p.c0 < 0 ? -p.c0 : 0;
    int padR = (p.V - 1) * p.su + p.TC - 1 + p.c0 - (p.XW - 1);
    if (padR < 0) padR = 0;
    p.LW = p.padL + p.XW + padR;
    long long out_rows = (pl->NPIX + p.V - 2) / p.V + 1;
    if (out_rows > (long long)p.B * p.U) out_rows = (long long)p.B * p.U;
    long long nseg = (out_rows + p.U - 2) / p.U + 1;
    if (nseg > p.B) nseg = p.B;
    p.nrows_max = (int)(out_rows * p.su + nseg * (p.TR > p.su ? p.TR - p.su : 0));
    p.chs = p.nrows_max * p.LW + (p.TR - 1) * 0;  // window never leaves the rows staged for its own pixel
    // the last pixel's window may run up to TC-1 floats past its row end only inside the padded pitch -> in range.
    if (pl->MF == 16) {  // de-conflict the four k-groups of a 16x16x4 B read (see DESIGN.md "LDS banking")
        const int want = (p.su & 1) ? 16 : 17;
        p.chs += ((want - p.chs % 32) + 32) % 32;
    }
    const int T = p.TR * p.TC;
    pl->a_floats = (size_t)((p.M + pl->MT - 1) / pl->MT) * p.nchunk * T * pl->CK * pl->MT;
    pl->lds_bytes = ((size_t)T * pl->CK * pl->MT + (size_t)pl->CK * p.chs) * sizeof(float) + (size_t)p.nrows_max * 4;
    CNN_REQUIRE(pl->lds_bytes <= 160 * 1024, "%s: tile needs %zu B of LDS (> 160 KiB): k=%d W=%d not supported", who,
                pl->lds_bytes, d->k, d->W);
    p.rw_shift = 0;
    while ((1 << p.rw_shift) < p.XW && p.rw_shift < 6) ++p.rw_shift;
    p.need_zero = (p.padL > 0 || padR > 0 || p.r0 < 0 || (p.U - 1) * p.su + p.r0 + p.TR - 1 > p.XH - 1) ? 1 : 0;
    p.ntiles = (int)((p.N + pl->NPIX - 1) / pl->NPIX);
    pl->grid_x = (unsigned)p.ntiles;
    pl->grid_y = (unsigned)((p.M + pl->MT - 1) / pl->MT);

    q.Co = d->Co; q.Ci = d->Ci; q.k = d->k; q.s = d->s; q.pad = d->pad; q.mode = mode;
    q.C = p.C; q.M = p.M; q.TR = p.TR; q.TC = p.TC; q.r0 = p.r0; q.c0 = p.c0;
    q.CK = pl->CK; q.MT = pl->MT; q.nchunk = p.nchunk; q.nmb = (int)pl->grid_y;
    return CNN_AMD_OK;
}

#define CONV_TAG(d) "B%d Ci%d %dx%d Co%d k%d s%d p%d", (d)->B, (d)->Ci, (d)->H, (d)->W, (d)->Co, (d)->k, (d)->s, (d)->pad

template <int MF, int MA, int NB, int WM, int WN, int CK>
int launch_cfg(const Plan& pl, hipStream_t s, const cnn_conv2d_desc* d) {
    auto kern = igemm_kernel<MF, MA, NB, WM, WN, CK>;
    static thread_local size_t max_set = 0;
    if (pl.lds_bytes > 48 * 1024 && pl.lds_bytes > max_set) {
        CNN_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        max_set = 160 * 1024;
    }
    char name[96];
    snprintf(name, sizeof(name), "igemm_kernel<%d,%d,%d,%d,%d,%d>%s", MF, MA, NB, WM, WN, CK,
             pl.p.mode == MODE_FWD ? "/fwd" : "/dgrad");
    CNN_KLAUNCH(s, name, (kern<<<dim3(pl.grid_x, pl.grid_y), 64 * WM * WN, pl.lds_bytes, s>>>(pl.p)), CONV_TAG(d));
    return CNN_AMD_OK;
}

int run_plan(Plan& pl, const cnn_conv2d_desc* d, const float* X, const float* w, const float* bias, float* Y, void* ws,
             size_t ws_bytes, hipStream_t s, const char* who) {
    CNN_REQUIRE(ws != nullptr, "%s: workspace is null", who);
    if (ws_bytes < pl.a_floats * sizeof(float))
        return fail(CNN_AMD_E_WORKSPACE, "%s: workspace %zu B < %zu B", who, ws_bytes, pl.a_floats * sizeof(float));
    pl.q.w = w;
    pl.q.A = (float*)ws;
    const long long total = (long long)pl.a_floats;
    unsigned pg = (unsigned)((total + 255) / 256);
    if (pg > 4096) pg = 4096;
    CNN_KLAUNCH(s, pl.p.mode == MODE_FWD ? "igemm_prep_weights/fwd" : "igemm_prep_weights/dgrad",
                (igemm_prep_weights<<<pg, 256, 0, s>>>(pl.q)), CONV_TAG(d));
    pl.p.X = X; pl.p.A = (const float*)ws; pl.p.bias = bias; pl.p.Y = Y;
    switch (pl.cfg) {
        case CFG_M128: return launch_cfg<32, 4, 1, 1, 4, 8>(pl, s, d);
        case CFG_M128_S: return launch_cfg<32, 2, 1, 2, 2, 8>(pl, s, d);
        case CFG_M64: return launch_cfg<32, 2, 2, 1, 4, 8>(pl, s, d);
        case CFG_M64_S: return launch_cfg<32, 1, 1, 2, 2, 8>(pl, s, d);
        case CFG_M32: return launch_cfg<32, 1, 4, 1, 4, 8>(pl, s, d);
        case CFG_M32_S: return launch_cfg<32, 1, 1, 1, 4, 8>(pl, s, d);
        case CFG_M16_CK4: return launch_cfg<16, 1, 4, 1, 4, 4>(pl, s, d);
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
    return CNN_AMD_OK;
}

}  // namespace

namespace cnn_amd {
// scratch floats the forward / dgrad plans need (used by cnn_conv2d_workspace_bytes in conv_wgrad.hip)
size_t igemm_workspace_floats(const cnn_conv2d_desc* d) {
    Plan a, b;
    size_t n = 0;
    if (make_plan("ws", d, MODE_FWD, &a) == CNN_AMD_OK) n = a.a_floats;
    if (make_plan("ws", d, MODE_DGRAD, &b) == CNN_AMD_OK && b.a_floats > n) n = b.a_floats;
    return n;
}
}  // namespace cnn_amd

extern "C" {

int cnn_conv2d_forward(const cnn_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y,
                       void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_desc("cnn_conv2d_forward", d)) return rc;
    CNN_REQUIRE(x && w && bias && y, "cnn_conv2d_forward: null pointer");
    Plan pl;
    if (int rc = make_plan("cnn_conv2d_forward", d, MODE_FWD, &pl)) return rc;
    return run_plan(pl, d, x, w, bias, y, ws, ws_bytes, as_stream(stream), "cnn_conv2d_forward");
}

int cnn_conv2d_backward_data(const cnn_conv2d_desc* d, const float* dy, const float* w, float* dx, void* ws,
                             size_t ws_bytes, void* stream) {
    if (int rc = check_desc("cnn_conv2d_backward_data", d)) return rc;
    CNN_REQUIRE(dy && w && dx, "cnn_conv2d_backward_data: null pointer");
    Plan pl;
    if (int rc = make_plan("cnn_conv2d_backward_data", d, MODE_DGRAD, &pl)) return rc;
    return run_plan(pl, d, dy, w, nullptr, dx, ws, ws_bytes, as_stream(stream), "cnn_conv2d_backward_data");
}

}  // extern "C"
