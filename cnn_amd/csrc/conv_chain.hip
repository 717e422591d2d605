// conv_chain.hip -- the back half of the reference net as TWO sample-resident kernels (round 4).
//
// Behind the pool-fused first block every tensor of the reference net (alexnet.cpp:12-31) is small PER SAMPLE -- 16x55x55, 32x27x27,
// 64x13x13, 128x6x6 floats -- and every layer of the forward chain (conv2d.cpp:69-92, relu.cpp:21-26, linear.cpp:33-43, func.cpp:16-73)
// and of the data-gradient chain (linear.cpp:73-90, relu.cpp:35-40, conv2d.cpp:168-199) is independent across samples.  The per-layer
// kernels (conv_fwd_rd.hip / conv_dgrad_rd.hip: 16x16x4 MFMA tiles, a wave's 16-channel filter slice held in registers as the A
// operand) spend as long in launch gaps, filter preambles, ramp-up and drain as in their MFMA loops on these layers: 9 launches, 237 of
// the step's 419 us in round 3 for ~65 us of arithmetic.  Here ONE workgroup takes ONE sample through the whole chain:
//
//   chain_fwd_kernel<N>:  [Conv2D+ReLU] x N  ->  LinearLayer -> softmax / cross-entropy -> delta -> d(linear input) (+ ReLU')
//   chain_bwd_kernel<N>:  data gradients of the N convolutions, last to first, each with the ReLU' of the layer in front
//
// with a workgroup barrier where the per-layer path has a kernel boundary.  The sample's intermediate tensors are written to HBM exactly
// once (the weight gradients and Layer::get_output() need them) and read back by the next phase of the SAME compute unit out of its
// L1 / L2 -- workgroup-scope visibility, no grid-wide synchronisation anywhere.  Every phase is the wave-level loop of the per-layer
// kernel with the pixel groups of one sample dealt to the workgroup's 8 waves: same products, same accumulation order -- the results are
// BIT-IDENTICAL to the per-layer path (tests/test_gpu_parity.py compares them bit for bit and against the oracle).
// Channels: the chain ends at 64 -> 128 and halves towards the front: N = 1: 64->128 | N = 2: 32->64->128 | N = 3: 16->32->64->128;
// every convolution 3x3, stride 2, no padding (the reference's constructor defaults, architectures.h:69); LinearLayer -> 3 classes.
#include <cfloat>
#include <cstdlib>

#include "common.h"

using namespace cnn_amd;

namespace cnn_amd {
bool fwd_rd_small(const cnn_conv2d_desc* d);                                // conv_fwd_rd.hip: served by the m16 forward kernel
int dgrad_rd_prepare_layout(const cnn_conv2d_desc* d, int* transposed);    // conv_dgrad_rd.hip: 2 = the m16 lane-major image
}

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) f3u {
    float x, y, z;
};
struct __attribute__((packed, aligned(4))) w3 {
    float a, b, c;
};
#define CH_PIPE_FENCE(reg) asm volatile("" : "+v"(reg) : : "memory")

constexpr int kChainWaves = 8;             // waves per workgroup = per sample (two per SIMD)
constexpr int kChainThreads = kChainWaves * 64;
constexpr unsigned kOOB = 0x7ffffffcu;     // buffer offset behind every tensor: loads give 0, stores are dropped

__device__ __forceinline__ int fdivm(int n, unsigned magic, int d) {
    int q = (int)__umulhi((unsigned)n, magic);
    if (q * d > n) --q;
    return q;
}

// ---- one Conv2D + ReLU of ONE sample: conv_fwd_m16_kernel's wave loop (conv_fwd_rd.hip) --------------------------------------------
// xi [CI][H][W] -> yi = relu(conv) [Co][Ho][Wo]; img = the layer's prepared lane-major filters (m16f_filter_index), bias [Co].
// A wave owns MS 16-channel slices for the whole phase (their Ci*9/4 filter values per slice are its A operands) and every
// (Co/16/MS)-th ... the pixel groups (16 consecutive output pixels) are dealt round-robin to the waves that share a slice.
template <int CI, int MS>
__device__ __forceinline__ void chain_fwd_phase(const float* __restrict__ xi, const float* __restrict__ img, const float* __restrict__ bias,
                                                float* __restrict__ yi, int H, int W, int Co, int Ho, int Wo, unsigned m_wo, int wave,
                                                int lane) {
    constexpr int C4 = CI / 4, NA = C4 * 9, NB = 4;
    static_assert(C4 % NB == 0, "static ring indices");
    const int n = lane & 15, k = lane >> 4;
    const int slices = (Co >> 4) / MS;
    const int parts = kChainWaves / slices;
    const int slice = wave % slices, part = wave / slices;
    const int HoWo = Ho * Wo;
    const int groups = (HoWo + 15) >> 4;
    if (part >= parts || part >= groups) return;
    float wa[MS][NA], bs[MS][4];
#pragma unroll
    for (int m = 0; m < MS; ++m) {
        const int sl = slice * MS + m;
#pragma unroll
        for (int j = 0; j < NA; ++j) wa[m][j] = img[(sl * NA + j) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) bs[m][r] = bias[16 * sl + 4 * k + r];
    }
    const unsigned plane = (unsigned)(H * W);
    auto locate = [&](int g, unsigned& xoff, unsigned& yoff, bool& live) {
        const int pi = g * 16 + n;
        live = g < groups && pi < HoWo;
        const int pic = live ? pi : HoWo - 1;  // (dead lanes read the last pixel's window and store nothing)
        const int pr = fdivm(pic, m_wo, Wo), q = pic - pr * Wo;
        xoff = (unsigned)(k * (int)plane + (2 * pr) * W + 2 * q);
        yoff = (unsigned)((16 * MS * slice + 4 * k) * HoWo + pic);
    };
    auto load_g = [&](f3u(&buf)[3], int c4, unsigned xoff) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float* base = xi + ((size_t)(c4 * 4) * plane + (size_t)kx * W);  // wave-uniform
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
        f32x4 acc[MS][2];  // two partial sums, used alternately (a dependent MFMA waits for the previous write-back)
#pragma unroll
        for (int m = 0; m < MS; ++m) {
            acc[m][0] = f32x4{bs[m][0], bs[m][1], bs[m][2], bs[m][3]};
            acc[m][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int c4 = 0; c4 < C4; ++c4) {
            const int gn = c4 + NB - 1;  // requested now (from c4 = C4-3 on: the first granules of the next group)
            load_g(xb[gn % NB], gn % C4, gn < C4 ? xoff : nxoff);
            CH_PIPE_FENCE(xb[c4 % NB][0].x);
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const f3u v = xb[c4 % NB][i / 3];
                const float bv = i % 3 == 0 ? v.x : i % 3 == 1 ? v.y : v.z;
                const int a = (c4 * 9 + i) % 2;
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
                yi[(size_t)yoff + (size_t)(16 * m + r) * HoWo] = v >= 0.f ? v : 0.f;  // relu.cpp:21-26 (keeps -0.0, NaN -> 0)
            }
        }
        xoff = nxoff;
        yoff = nyoff;
        live = nlive;
    }
}

// ---- the same phase with the filter slice held in HALVES (lean chain: <= 168 VGPRs per wave, see chain_fwd_kernel) -------------------
// MS = 1; a wave keeps the two partial sums of ALL its pixel groups (at most GMAX) in registers and walks them twice: first with
// the filter values of channels [0, CI/2), then with those of [CI/2, CI).  Every accumulator still receives its MFMA steps in the order
// of chain_fwd_phase (step (c4, tap) feeds partial sum (9 c4 + tap) mod 2, c4 ascending): bit-identical.
template <int CI, int GMAX>
__device__ __forceinline__ void chain_fwd_phase_split(const float* __restrict__ xi, const float* __restrict__ img, const float* __restrict__ bias,
                                                      float* __restrict__ yi, int H, int W, int Co, int Ho, int Wo, unsigned m_wo, int wave,
                                                      int lane) {
    constexpr int C4 = CI / 4, NA = C4 * 9, HC4 = C4 / 2, HNA = HC4 * 9, NB = 4;
    static_assert(HC4 % NB == 0, "static ring indices");
    const int n = lane & 15, k = lane >> 4;
    const int slices = Co >> 4;
    const int parts = kChainWaves / slices;
    const int slice = wave % slices, part = wave / slices;
    const int HoWo = Ho * Wo;
    const int groups = (HoWo + 15) >> 4;
    if (part >= parts || part >= groups) return;
    const int mine = (groups - part + parts - 1) / parts;  // pixel groups of this wave: part, part + parts, ... (<= GMAX, checked by the host)
    const unsigned plane = (unsigned)(H * W);
    unsigned xoff[GMAX], yoff[GMAX];
    bool live[GMAX];
#pragma unroll
    for (int j = 0; j < GMAX; ++j) {
        const int g = part + j * parts;
        const int pi = g * 16 + n;
        live[j] = j < mine && pi < HoWo;
        const int pic = live[j] ? pi : HoWo - 1;
        const int pr = fdivm(pic, m_wo, Wo), q = pic - pr * Wo;
        xoff[j] = (unsigned)(k * (int)plane + (2 * pr) * W + 2 * q);
        yoff[j] = (unsigned)((16 * slice + 4 * k) * HoWo + pic);
    }
    f32x4 acc[GMAX][2];
#pragma unroll
    for (int j = 0; j < GMAX; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][0][r] = bias[16 * slice + 4 * k + r];
        acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto load_g = [&](f3u(&buf)[3], int c4, unsigned xo) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float* base = xi + ((size_t)(c4 * 4) * plane + (size_t)kx * W);  // wave-uniform
            buf[kx] = *(const f3u*)(base + xo);
        }
    };
    static_assert((HC4 * 9) % 2 == 0, "the partial sum a step feeds must not depend on the half");
#pragma nounroll
    for (int h = 0; h < 2; ++h) {  // (a real loop: unrolled, hipcc fetches the second half's 72 filter values early and spills)
        float wa[HNA];
#pragma unroll
        for (int j = 0; j < HNA; ++j) wa[j] = img[(slice * NA + h * HNA + j) * 64 + lane];
        f3u xb[NB][3];
#pragma unroll
        for (int i = 0; i < NB - 1; ++i) load_g(xb[i], h * HC4 + i, xoff[0]);
#pragma unroll
        for (int j = 0; j < GMAX; ++j) {
            if (j < mine) {  // (wave-uniform)
                const unsigned nxo = xoff[j + 1 < GMAX ? j + 1 : j];  // (behind the last group: a harmless re-read)
#pragma unroll
                for (int c = 0; c < HC4; ++c) {
                    const int gn = c + NB - 1;
                    load_g(xb[gn % NB], h * HC4 + gn % HC4, gn < HC4 ? xoff[j] : nxo);
                    CH_PIPE_FENCE(xb[c % NB][0].x);
#pragma unroll
                    for (int i = 0; i < 9; ++i) {
                        const f3u v = xb[c % NB][i / 3];
                        const float bv = i % 3 == 0 ? v.x : i % 3 == 1 ? v.y : v.z;
                        const int a = (c * 9 + i) % 2;
                        acc[j][a] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c * 9 + i], bv, acc[j][a], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < GMAX; ++j) {
        if (live[j]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[j][0][r] + acc[j][1][r];
                yi[(size_t)yoff[j] + (size_t)r * HoWo] = v >= 0.f ? v : 0.f;  // relu.cpp:21-26
            }
        }
    }
}

// func.cpp:6-12
__device__ __forceinline__ float clamped_exp_c(float v) {
    if (v >= 88.f) return FLT_MAX;
    if (v <= -50.f) return 0.f;
    return expf(v);
}
__device__ __forceinline__ float wave_sum_c(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ---- the head of ONE sample: linear_fwd_softmax_xent<true, 2> (linear.hip) on the first 256 threads, term by term ------------------
// logits = x W + bias (linear.cpp:33-43), softmax / cross-entropy / delta = probs - onehot (func.cpp:16-73),
// dx = (x <= 0) ? 0 : delta W^T (linear.cpp:73-90, relu.cpp:38).  out == 3.  Every thread of the workgroup must call it (barriers).
struct HeadShared {
    float part[4][8];
    float logit[8];
    float dl[8];
};
template <bool LEAN = false>  // LEAN: half as many loads in flight, nothing kept in registers for the dx pass (x and W are read again)
__device__ __forceinline__ void chain_head_phase(const float* __restrict__ xb, const float* __restrict__ w, const float* __restrict__ bias,
                                                 const int32_t* __restrict__ labels, float* __restrict__ y, float* __restrict__ probs,
                                                 float* __restrict__ delta, float* __restrict__ loss_terms, float* __restrict__ dxb, int in,
                                                 int out_rt, int b, HeadShared& sh) {
    constexpr int kBlock = 256, out = 3, UK = 18, kOutTile = 8;  // (out_rt == 3: the run-time copy keeps the generic loops' code shape)
    constexpr int U = LEAN ? 9 : UK, KEEP = LEAN ? 1 : UK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool worker = tid < kBlock;
    float xk[KEEP];
    w3 wk[KEEP];
    bool kept = false;
    float acc[3] = {0.f, 0.f, 0.f};
    if (worker) {
        int i = tid;
        for (; i + (U - 1) * kBlock < in; i += U * kBlock) {  // (ascending i: the sums do not depend on U)
            float xv[U];
            w3 wv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xv[u] = xb[i + u * kBlock];
                wv[u] = *reinterpret_cast<const w3*>(w + (size_t)(i + u * kBlock) * 3);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc[0] = __builtin_fmaf(xv[u], wv[u].a, acc[0]);
                acc[1] = __builtin_fmaf(xv[u], wv[u].b, acc[1]);
                acc[2] = __builtin_fmaf(xv[u], wv[u].c, acc[2]);
            }
            if (in == UK * kBlock) {
                kept = true;
                if constexpr (!LEAN) {
#pragma unroll
                    for (int u = 0; u < UK; ++u) {
                        xk[u] = xv[u];
                        wk[u] = wv[u];
                    }
                }
            }
        }
#pragma unroll 6
        for (; i < in; i += kBlock) {
            const float xv = xb[i];
            const float* wr = w + (size_t)i * out;
#pragma unroll
            for (int j = 0; j < out; ++j) acc[j] = __builtin_fmaf(xv, wr[j], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < out; ++j) {
            const float s = wave_sum_c(acc[j]);
            if (lane == 0) sh.part[wave][j] = s;
        }
    }
    __syncthreads();
    if (tid < out) {
        float s = 0.f;
        for (int wv = 0; wv < kBlock / 64; ++wv) s += sh.part[wv][tid];
        s += bias[tid];
        y[(size_t)b * out + tid] = s;
        sh.logit[tid] = s;
    }
    __syncthreads();
    if (tid == 0) {  // the reference's sequential per-sample arithmetic
        float mx = sh.logit[0];
        for (int i = 1; i < out; ++i)
            if (sh.logit[i] > mx) mx = sh.logit[i];
        float sum = 0.f;
        for (int i = 0; i < out; ++i) sum += clamped_exp_c(sh.logit[i] - mx);
        const int label = labels[b];
        float term = 0.f;
        for (int i = 0; i < out; ++i) {
            float pr = clamped_exp_c(sh.logit[i] - mx) / sum;
            if (isnan(pr)) pr = 0.f;
            const float yv = (i == label) ? 1.f : 0.f;
            if (probs) probs[(size_t)b * out + i] = pr;
            delta[(size_t)b * out + i] = pr - yv;
            sh.dl[i] = pr - yv;
            term += logf(pr) * yv;
        }
        loss_terms[b] = term;
    }
    __syncthreads();
    if (!worker) return;
    const float d0 = sh.dl[0], d1 = sh.dl[1], d2 = sh.dl[2];
    if (kept) {  // in == 18 * 256: no loads at all (LEAN: x and the W rows once more, same expression)
#pragma unroll
        for (int u = 0; u < UK; ++u) {
            const float xu = LEAN ? xb[tid + u * kBlock] : xk[LEAN ? 0 : u];
            const w3 wu = LEAN ? *reinterpret_cast<const w3*>(w + (size_t)(tid + u * kBlock) * 3) : wk[LEAN ? 0 : u];
            float sj = 0.f;  // (linear_bwd_fused's expression, term by term)
            sj += d0 * wu.a;
            sj += d1 * wu.b;
            sj += d2 * wu.c;
            dxb[tid + u * kBlock] = (xu <= 0.f) ? 0.f : sj;
        }
    } else {
        for (int i2 = tid; i2 < in; i2 += kBlock) {  // (linear_fwd_softmax_xent's generic loop, statement by statement)
            const float* wr = w + (size_t)i2 * out_rt;
            float sj = 0.f;
#pragma unroll
            for (int j = 0; j < kOutTile; ++j)
                if (j < out_rt) sj += sh.dl[j] * wr[j];
            dxb[i2] = (xb[i2] <= 0.f) ? 0.f : sj;
        }
    }
}

struct ChainFwdParams {
    const float* x;          // input of the first convolution [B][C0][H0][W0]
    const float* img[3];     // prepared forward filters per layer (front to back)
    const float* bias[3];
    float* a[3];             // ReLU outputs per layer
    int H[4], W[4];          // H[l], W[l]: input of layer l; H[N], W[N]: output of the last one
    unsigned m_wo[3];        // magic multiplier of each layer's output width
    const float* lin_w;      // [in][3], bias behind it
    const float* lin_b;
    const int32_t* labels;
    float* logits;
    float* probs;            // nullable
    float* delta;
    float* loss_terms;
    float* dx_head;          // d(linear input), masked by the last ReLU
    int lin_in, lin_out;
};

template <int N, bool DBG = false>
__global__ __launch_bounds__(kChainThreads) void chain_fwd_kernel(const ChainFwdParams p) {
    __shared__ HeadShared sh;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int C0 = 128 >> N;
    long long t[6] = {0, 0, 0, 0, 0, 0};  // (DBG: s_memtime stamps of wave 0 at the phase boundaries)
    if (DBG) t[0] = clock64();
    const float* xin = p.x + (size_t)b * C0 * p.H[0] * p.W[0];
    if constexpr (N >= 3) {
        float* out = p.a[N - 3] + (size_t)b * 32 * p.H[N - 2] * p.W[N - 2];
        chain_fwd_phase<16, 2>(xin, p.img[N - 3], p.bias[N - 3], out, p.H[N - 3], p.W[N - 3], 32, p.H[N - 2], p.W[N - 2], p.m_wo[N - 3], wave, lane);
        if (DBG) t[1] = clock64();
        __syncthreads();
        xin = out;
    }
    if constexpr (N >= 2) {
        float* out = p.a[N - 2] + (size_t)b * 64 * p.H[N - 1] * p.W[N - 1];
        if (DBG) t[2] = clock64();
        chain_fwd_phase<32, 2>(xin, p.img[N - 2], p.bias[N - 2], out, p.H[N - 2], p.W[N - 2], 64, p.H[N - 1], p.W[N - 1], p.m_wo[N - 2], wave, lane);
        if (DBG) t[3] = clock64();
        __syncthreads();
        xin = out;
    }
    float* a_last = p.a[N - 1] + (size_t)b * 128 * p.H[N] * p.W[N];
    if (DBG) t[4] = clock64();
    chain_fwd_phase<64, 1>(xin, p.img[N - 1], p.bias[N - 1], a_last, p.H[N - 1], p.W[N - 1], 128, p.H[N], p.W[N], p.m_wo[N - 1], wave, lane);
    if (DBG) t[5] = clock64();
    __syncthreads();
    const long long th = DBG ? clock64() : 0;
    chain_head_phase(a_last, p.lin_w, p.lin_b, p.labels, p.logits, p.probs, p.delta, p.loss_terms, p.dx_head + (size_t)b * p.lin_in, p.lin_in,
                     p.lin_out, b, sh);
    if (DBG && (b == 0 || b == 100) && lane == 0 && (wave == 0 || wave == 7))
        printf("chain_fwd<%d> wg %d wave %d: phase A %lld (+barrier %lld) | B %lld (+barrier %lld) | C %lld (+barrier %lld) | head %lld | total %lld ticks\n", N, b,
               wave, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], th - t[5], (long long)clock64() - th, (long long)clock64() - t[0]);
}

// The LEAN forward chain: the same phases within 168 VGPRs per wave (three waves per SIMD's worth: the workgroup's two leave a third of
// every compute unit's register file to other kernels -- two workgroups of the deferred first-layer data gradient, 88 VGPRs x 4 waves
// each, fit beside it).  conv_layer_3's waves own ONE 16-channel slice (72 filter registers) instead of two, conv_layer_4's hold their
// slice in halves (chain_fwd_phase_split).  Same results, bit for bit.
template <int N>
__global__ __launch_bounds__(kChainThreads) __attribute__((amdgpu_waves_per_eu(3, 3))) void chain_fwd_lean_kernel(const ChainFwdParams p) {
    __shared__ HeadShared sh;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int C0 = 128 >> N;
    const float* xin = p.x + (size_t)b * C0 * p.H[0] * p.W[0];
    if constexpr (N >= 3) {
        float* out = p.a[N - 3] + (size_t)b * 32 * p.H[N - 2] * p.W[N - 2];
        chain_fwd_phase<16, 2>(xin, p.img[N - 3], p.bias[N - 3], out, p.H[N - 3], p.W[N - 3], 32, p.H[N - 2], p.W[N - 2], p.m_wo[N - 3], wave, lane);
        __syncthreads();
        xin = out;
    }
    if constexpr (N >= 2) {
        float* out = p.a[N - 2] + (size_t)b * 64 * p.H[N - 1] * p.W[N - 1];
        chain_fwd_phase<32, 1>(xin, p.img[N - 2], p.bias[N - 2], out, p.H[N - 2], p.W[N - 2], 64, p.H[N - 1], p.W[N - 1], p.m_wo[N - 2], wave, lane);
        __syncthreads();
        xin = out;
    }
    float* a_last = p.a[N - 1] + (size_t)b * 128 * p.H[N] * p.W[N];
    chain_fwd_phase_split<64, 3>(xin, p.img[N - 1], p.bias[N - 1], a_last, p.H[N - 1], p.W[N - 1], 128, p.H[N], p.W[N], p.m_wo[N - 1], wave, lane);
    __syncthreads();
    chain_head_phase<true>(a_last, p.lin_w, p.lin_b, p.labels, p.logits, p.probs, p.delta, p.loss_terms, p.dx_head + (size_t)b * p.lin_in, p.lin_in,
                     p.lin_out, b, sh);
}

// ---- data gradient of ONE stride-2 convolution for ONE sample: conv_dgrad_m16_s2_kernel's wave loop (conv_dgrad_rd.hip) -------------
// dyi [CO*KS][Ho][Wo] -> dxi [Ci][H][W] (= (mask <= 0) ? 0 : dx when maski is given: the ReLU::backward of the layer in front).
// A grid pixel (u, v) owns dx[.][2u | 2u+1][2v | 2v+1]; K = 4 dy channels of one filter tap per MFMA step; a wave holds the 2.25*CO
// filter values of its 16-input-channel slice (and, KS = 2, of its half of the dy channels) as A operands.  KS = 2: slices * 2 must be
// the workgroup's 8 waves; the second half hands its partial sums to the first through `red` (one barrier per pixel group).
template <int CO, int KS>
__device__ __forceinline__ void chain_dgrad_phase(const float* __restrict__ dyi, const float* __restrict__ wimg, const float* maski,
                                                  float* dxi, int Ci, int H, int W, int Ho, int Wo, unsigned m_v, int wave, int lane,
                                                  float* red) {
    constexpr int C4 = CO / 4, NA = CO * 9 / 4, NB = 4;
    static_assert(C4 % NB == 0, "static ring indices");
    const int n = lane & 15, k = lane >> 4;
    const int slices = Ci >> 4;
    const int parts = kChainWaves / (slices * KS);  // (KS = 2: 1)
    const int slice = wave % slices, half = KS > 1 ? (wave / slices) % KS : 0, part = wave / (slices * KS);
    const int U = (H + 1) / 2, V = (W + 1) / 2, UV = U * V;
    const int groups = (UV + 15) >> 4;
    if (KS == 1 && (part >= parts || part >= groups)) return;  // (KS = 2: every wave takes part in the barriers below)
    float wa[NA];  // [c4*9 + tap]
#pragma unroll
    for (int j = 0; j < NA; ++j) wa[j] = wimg[((half * slices + slice) * NA + j) * 64 + lane];
    const int plane = Ho * Wo;
    const size_t hw = (size_t)H * W;
    const unsigned chs = (unsigned)hw * 4u;
    const bool odd = (W & 1) != 0;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)dyi, 0, (int)((unsigned)CO * KS * plane * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)dxi, 0, (int)((unsigned)Ci * (unsigned)hw * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t mrs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(maski ? maski : dxi), 0, (int)((unsigned)Ci * (unsigned)hw * 4u), 0x00020000);
    struct Loc {
        unsigned o0, o1;  // dy byte offsets of this lane's pair in rows u, u-1 (channel k of a granule), or out of range: reads 0
        bool ca, cb;      // column case: a: pair = (v-1, v) | b: v = 0, pair = (0, 1) | neither: v = Wo, pair = (Wo-2, Wo-1)
        unsigned x0, x1;  // dx byte offsets of rows 2u, 2u+1 at column 2v, channel 4k (or out of range: no store)
        bool w1;          // column 2v+1 exists (false only in the last column of an odd W)
    };
    auto locate = [&](int g, Loc& L) {
        const int pix = g * 16 + n;
        const bool live = g < groups && pix < UV;
        const int pp = live ? pix : 0;
        const int u = fdivm(pp, m_v, V);
        const int v = pp - u * V;
        L.cb = v == 0;
        L.ca = !L.cb && v < Wo;
        const int cs = L.ca ? v - 1 : (L.cb ? 0 : Wo - 2);  // (v <= Wo always: V = Wo + 1)
        const unsigned base = (unsigned)((half * CO + k) * plane + u * Wo + cs) * 4u;
        L.o0 = (live && u < Ho) ? base : kOOB;
        L.o1 = (live && u >= 1) ? base - (unsigned)Wo * 4u : kOOB;
        const unsigned xb = (unsigned)(((size_t)16 * slice + 4 * k) * hw + (size_t)(2 * u) * W + 2 * v) * 4u;
        L.x0 = live ? xb : kOOB;
        L.x1 = (live && 2 * u + 1 < H) ? xb + (unsigned)W * 4u : kOOB;
        L.w1 = 2 * v + 1 < W;
    };
    auto load_g = [&](v2f(&buf)[2], int c4, const Loc& L) {
        const int so = c4 * 4 * plane * 4;
        buf[0] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)L.o0, so, 0));
        buf[1] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)L.o1, so, 0));
    };
    Loc cur, nxt;
    locate(part, cur);
    v2f ring[NB][2];
#pragma unroll
    for (int i = 0; i < NB - 1; ++i) load_g(ring[i], i, cur);
    for (int g = part; g < groups; g += parts) {
        locate(g + parts, nxt);  // (behind the last group: every offset out of range)
        v2f mk[4][2];
        if (maski && half == 0) {
            const unsigned m0 = (cur.w1 || cur.x0 == kOOB) ? cur.x0 : cur.x0 - 4u, m1 = (cur.w1 || cur.x1 == kOOB) ? cur.x1 : cur.x1 - 4u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mk[r][0] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(mrs, (int)m0, (int)(r * chs), 0));
                mk[r][1] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(mrs, (int)m1, (int)(r * chs), 0));
            }
        }
        f32x4 acc[4];  // classes (ph,pw) = 00, 01, 10, 11; register r of lane (n, k) = channel 4k + r of pixel n
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c4 = 0; c4 < C4; ++c4) {
            const int gn = c4 + NB - 1;
            load_g(ring[gn % NB], gn % C4, gn < C4 ? cur : nxt);
            CH_PIPE_FENCE(ring[c4 % NB][0]);
            const v2f r0 = ring[c4 % NB][0], r1 = ring[c4 % NB][1];
            const float d00 = cur.ca ? r0.y : (cur.cb ? r0.x : 0.f), d01 = cur.ca ? r0.x : (cur.cb ? 0.f : r0.y);  // D[0][jc]
            const float d10 = cur.ca ? r1.y : (cur.cb ? r1.x : 0.f), d11 = cur.ca ? r1.x : (cur.cb ? 0.f : r1.y);  // D[1][jc]
            const float* a = &wa[c4 * 9];
#define CH_STEP(ACC, A_, B_)                                           \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(A_, B_, ACC, 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0)
            CH_STEP(acc[1], a[1], d00);
            CH_STEP(acc[0], a[0], d00);
            CH_STEP(acc[2], a[3], d00);
            CH_STEP(acc[0], a[2], d01);
            CH_STEP(acc[3], a[4], d00);
            CH_STEP(acc[0], a[6], d10);
            CH_STEP(acc[1], a[7], d10);
            CH_STEP(acc[0], a[8], d11);
            CH_STEP(acc[2], a[5], d01);
#undef CH_STEP
        }
        if constexpr (KS > 1) {
            // red[2][slices = 4][16][64]: two buffers, alternating per pixel group: one barrier per group
            float* rb = red + (size_t)(((g - part) / parts) & 1) * (4 * 16 * 64);
            if (half > 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) rb[(slice * 16 + c * 4 + r) * 64 + lane] = acc[c][r];
            }
            __syncthreads();
            if (half > 0) {
                cur = nxt;
                continue;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[c][r] += rb[(slice * 16 + c * 4 + r) * 64 + lane];
        }
        // epilogue, branch-free per lane: buffer stores whose offset is out of range for dead lanes / the row behind the tensor
        const unsigned pp0 = cur.w1 ? cur.x0 : kOOB, pp1 = cur.w1 ? cur.x1 : kOOB;  // (pw = 0,1) pairs
        const unsigned ss0 = cur.w1 ? kOOB : cur.x0, ss1 = cur.w1 ? kOOB : cur.x1;  // single element, last column of an odd W
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                float v0 = acc[ph * 2][r], v1 = acc[ph * 2 + 1][r];
                if (maski) {
                    const float k0 = cur.w1 ? mk[r][ph].x : mk[r][ph].y;
                    v0 = (k0 <= 0.f) ? 0.f : v0;
                    v1 = (mk[r][ph].y <= 0.f) ? 0.f : v1;
                }
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v2f{v0, v1}), xrs, (int)(ph ? pp1 : pp0), (int)(r * chs), 0);
                if (odd) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), xrs, (int)(ph ? ss1 : ss0), (int)(r * chs), 0);
            }
        cur = nxt;
    }
}

struct ChainBwdParams {
    const float* dy;          // delta of the LAST convolution's output (ReLU' applied) [B][128][H[N]][W[N]]
    const float* img[3];      // prepared data-gradient filters per layer (front to back)
    const float* mask[3];     // the ReLU output that is layer l's input (nullable: no ReLU in front, e.g. a pooled-domain delta)
    float* dx[3];             // delta of layer l's input
    int H[4], W[4];
    unsigned m_v[3];          // magic multiplier of each layer's grid width (W[l] + 1) / 2
};

template <int N>
__global__ __launch_bounds__(kChainThreads) void chain_bwd_kernel(const ChainBwdParams p) {
    __shared__ float red[2 * 4 * 16 * 64];  // 32 KB: the split-K hand-over of the 64 -> 128 layer
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        constexpr int l = N - 1;
        const float* dy = p.dy + (size_t)b * 128 * p.H[N] * p.W[N];
        const size_t in = (size_t)b * 64 * p.H[l] * p.W[l];
        chain_dgrad_phase<64, 2>(dy, p.img[l], p.mask[l] ? p.mask[l] + in : nullptr, p.dx[l] + in, 64, p.H[l], p.W[l], p.H[N], p.W[N], p.m_v[l], wave,
                                 lane, red);
    }
    if constexpr (N >= 2) {
        __syncthreads();
        constexpr int l = N - 2;
        const float* dy = p.dx[l + 1] + (size_t)b * 64 * p.H[l + 1] * p.W[l + 1];
        const size_t in = (size_t)b * 32 * p.H[l] * p.W[l];
        chain_dgrad_phase<64, 1>(dy, p.img[l], p.mask[l] ? p.mask[l] + in : nullptr, p.dx[l] + in, 32, p.H[l], p.W[l], p.H[l + 1], p.W[l + 1],
                                 p.m_v[l], wave, lane, red);
    }
    if constexpr (N >= 3) {
        __syncthreads();
        constexpr int l = N - 3;
        const float* dy = p.dx[l + 1] + (size_t)b * 32 * p.H[l + 1] * p.W[l + 1];
        const size_t in = (size_t)b * 16 * p.H[l] * p.W[l];
        chain_dgrad_phase<32, 1>(dy, p.img[l], p.mask[l] ? p.mask[l] + in : nullptr, p.dx[l] + in, 16, p.H[l], p.W[l], p.H[l + 1], p.W[l + 1],
                                 p.m_v[l], wave, lane, red);
    }
}

inline unsigned magic_of(int d) { return (unsigned)((1ull << 32) / (unsigned)d) + 1u; }

// the chain's geometry: n layers, channels (128 >> n) -> ... -> 128, each 3x3 / stride 2 / pad 0 on the previous one's output, served by
// the m16 kernels of the per-layer path (so that the prepared filter images are the ones this file reads)
bool chain_geometry_ok(int n, const cnn_conv2d_desc* d) {
    if (n < 1 || n > 3 || d == nullptr) return false;
    for (int l = 0; l < n; ++l) {
        const cnn_conv2d_desc& c = d[l];
        const int ci = 128 >> (n - l);
        if (c.Ci != ci || c.Co != 2 * ci || c.k != 3 || c.s != 2 || c.pad != 0 || c.B != d[0].B || c.B < 1) return false;
        if (c.H < 5 || c.W < 5) return false;  // (the data gradient's column pairs need Wo >= 2)
        if (l > 0 && (c.H != cnn_conv2d_out_dim(d[l - 1].H, 3, 2, 0) || c.W != cnn_conv2d_out_dim(d[l - 1].W, 3, 2, 0))) return false;
        if ((long long)c.Ci * c.H * c.W >= (1ll << 27)) return false;  // (32-bit byte offsets inside one sample)
        int tr = 0;
        if (!fwd_rd_small(&c) || !dgrad_rd_prepare_layout(&c, &tr) || tr != 2) return false;
    }
    return true;
}

}  // namespace

extern "C" {

int cnn_conv_chain_supported(int n, const cnn_conv2d_desc* descs, int lin_in, int lin_out) {
    if (!chain_geometry_ok(n, descs)) return 0;
    if (const OptVal e = CNN_OPT_VAL("NO_CHAIN"))
        if (atoi(e) != 0) return 0;
    if (lin_out == 0 && lin_in == 0) return 1;  // (the data-gradient chain alone)
    const cnn_conv2d_desc& last = descs[n - 1];
    return (lin_out == 3 && lin_in == 128 * cnn_conv2d_out_dim(last.H, 3, 2, 0) * cnn_conv2d_out_dim(last.W, 3, 2, 0)) ? 1 : 0;
}

int cnn_conv_chain_forward_loss_prepared(int n, const cnn_conv2d_desc* descs, const float* x, const void* const* prepared_fwd,
                                         const float* const* bias, float* const* y_relu, const float* lin_w, const float* lin_bias,
                                         const int32_t* labels, float* logits, float* probs, float* delta, float* loss_terms, float* dx_head,
                                         int lin_in, int lin_out, void* stream) {
    CNN_REQUIRE(descs && x && prepared_fwd && bias && y_relu && lin_w && lin_bias && labels && logits && delta && loss_terms && dx_head,
                "cnn_conv_chain_forward_loss_prepared: null pointer");
    CNN_REQUIRE(cnn_conv_chain_supported(n, descs, lin_in, lin_out) != 0, "cnn_conv_chain_forward_loss_prepared: chain not covered (n=%d in=%d out=%d)", n,
                lin_in, lin_out);
    ChainFwdParams p{};
    p.x = x;
    for (int l = 0; l < n; ++l) {
        CNN_REQUIRE(prepared_fwd[l] && bias[l] && y_relu[l], "cnn_conv_chain_forward_loss_prepared: null pointer (layer %d)", l);
        p.img[l] = (const float*)prepared_fwd[l];
        p.bias[l] = bias[l];
        p.a[l] = y_relu[l];
        p.H[l] = descs[l].H;
        p.W[l] = descs[l].W;
        p.m_wo[l] = magic_of(cnn_conv2d_out_dim(descs[l].W, 3, 2, 0));
    }
    p.H[n] = cnn_conv2d_out_dim(descs[n - 1].H, 3, 2, 0);
    p.W[n] = cnn_conv2d_out_dim(descs[n - 1].W, 3, 2, 0);
    p.lin_w = lin_w; p.lin_b = lin_bias; p.labels = labels; p.logits = logits; p.probs = probs; p.delta = delta; p.loss_terms = loss_terms;
    p.dx_head = dx_head; p.lin_in = lin_in; p.lin_out = lin_out;
    hipStream_t s = as_stream(stream);
    const int B = descs[0].B;
    char name[48];
    snprintf(name, sizeof(name), "conv_chain_fwd<%d>+head", n);
#define CHAIN_F(N_) \
    CNN_KLAUNCH(s, name, (launch_pub(chain_fwd_kernel<N_>, dim3(B), dim3(kChainThreads), 0, s, p)), "B%d C%d %dx%d", B, descs[0].Ci, descs[0].H, descs[0].W)
    // CHAIN_LEAN=1 (measurement switch): the <= 168-VGPR instance; its last layer keeps the sums of all pixel groups of a wave in registers
    // (at most 3 groups: output maps up to 48 pixels)
    if (CNN_OPT_INT("CHAIN_LEAN", 0) != 0 && p.H[n] * p.W[n] <= 48) {
        snprintf(name, sizeof(name), "conv_chain_fwd_lean<%d>+head", n);
#define CHAIN_L(N_) \
    CNN_KLAUNCH(s, name, (launch_pub(chain_fwd_lean_kernel<N_>, dim3(B), dim3(kChainThreads), 0, s, p)), "B%d C%d %dx%d", B, descs[0].Ci, descs[0].H, descs[0].W)
        if (n == 1) CHAIN_L(1); else if (n == 2) CHAIN_L(2); else CHAIN_L(3);
#undef CHAIN_L
        return CNN_AMD_OK;
    }
    if (CNN_OPT_INT("CHAIN_DBG", 0) != 0 && n == 3) {
        CNN_KLAUNCH(s, name, (launch_pub(chain_fwd_kernel<3, true>, dim3(B), dim3(kChainThreads), 0, s, p)), "B%d C%d %dx%d", B, descs[0].Ci, descs[0].H, descs[0].W);
        return CNN_AMD_OK;
    }
    if (n == 1) CHAIN_F(1); else if (n == 2) CHAIN_F(2); else CHAIN_F(3);
#undef CHAIN_F
    return CNN_AMD_OK;
}

int cnn_conv_chain_backward_data_prepared(int n, const cnn_conv2d_desc* descs, const float* dy_last, const void* const* prepared_dgrad,
                                          const float* const* relu_below, float* const* dx, void* stream) {
    CNN_REQUIRE(descs && dy_last && prepared_dgrad && relu_below && dx, "cnn_conv_chain_backward_data_prepared: null pointer");
    CNN_REQUIRE(cnn_conv_chain_supported(n, descs, 0, 0) != 0, "cnn_conv_chain_backward_data_prepared: chain not covered (n=%d)", n);
    ChainBwdParams p{};
    p.dy = dy_last;
    for (int l = 0; l < n; ++l) {
        CNN_REQUIRE(prepared_dgrad[l] && dx[l], "cnn_conv_chain_backward_data_prepared: null pointer (layer %d)", l);
        p.img[l] = (const float*)prepared_dgrad[l];
        p.mask[l] = relu_below[l];
        p.dx[l] = dx[l];
        p.H[l] = descs[l].H;
        p.W[l] = descs[l].W;
        p.m_v[l] = magic_of((descs[l].W + 1) / 2);
    }
    p.H[n] = cnn_conv2d_out_dim(descs[n - 1].H, 3, 2, 0);
    p.W[n] = cnn_conv2d_out_dim(descs[n - 1].W, 3, 2, 0);
    hipStream_t s = as_stream(stream);
    const int B = descs[0].B;
    char name[48];
    snprintf(name, sizeof(name), "conv_chain_dgrad<%d>", n);
#define CHAIN_B(N_) \
    CNN_KLAUNCH(s, name, (launch_pub(chain_bwd_kernel<N_>, dim3(B), dim3(kChainThreads), 0, s, p)), "B%d C%d %dx%d", B, descs[0].Ci, descs[0].H, descs[0].W)
    if (n == 1) CHAIN_B(1); else if (n == 2) CHAIN_B(2); else CHAIN_B(3);
#undef CHAIN_B
    return CNN_AMD_OK;
}

}  // extern "C"
