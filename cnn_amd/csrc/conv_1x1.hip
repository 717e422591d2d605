// conv_1x1.hip -- dedicated kernels for 1x1 convolutions with stride 1 / 2 (round 4): the downsample shapes of the ResNet-18-shaped
// stack (BASELINE configs[4]; the reference reaches kernel_size 1 only with -DNDEBUG, cpu/src/conv2d.cpp:14).  With one tap the three
// loop nests of Conv2D (conv2d.cpp:69-92 forward, :168-199 data gradient, :117-159 weight / bias gradient) are plain GEMMs over the
// SUB-SAMPLED input plane xs[b][ci][p][q] = x[b][ci][s*p][s*q]:
//     forward        y [co][n]  = bias[co] + sum_ci w[co][ci] * xs[ci][n]                      n = (b, p, q)
//     data gradient  dxs[ci][n] = sum_co w[co][ci] * dy[co][n];  dx = 0 off the sampled grid, (relu_below <= 0 ? 0 : dxs) on it
//     weight grad.   gw[co][ci] = (sum_n dy[co][n] * xs[ci][n]) / divisor,  gb[co] = (sum_n dy[co][n]) / divisor
// The implicit GEMM (conv_igemm.hip) pays a chunk barrier per 8 channels of ONE tap and the generic weight-gradient kernel streams a
// 9-tap tile shape: 24 / 9.5 / 4.3-9.6 TFLOP/s on 128 -> 256, 28x28, stride 2 at batch 64 (profiles/NOTEBOOK.md section 9).  Here: K-chunked LDS
// GEMMs on v_mfma_f32_16x16x4_f32, 8 waves per workgroup, the next chunk's operands prefetched into registers while the current chunk's
// MFMAs run (one barrier pair per chunk), operand tiles padded so that both MFMA operand reads are bank-conflict free.
#include "common.h"

using namespace cnn_amd;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 8, kThreads = kWaves * 64;
constexpr int kKC = 32;          // K elements per chunk
constexpr int kPA = kKC + 2;     // pitch of the A tile [m][k]: lanes (m, k) hit banks 2m + k -- distinct per half-wave
constexpr int kMT = 16 * kWaves; // rows (M) per workgroup: one 16-row slice per wave

__device__ __forceinline__ int fdivm(int n, unsigned magic, int d) {
    if (d == 1) return n;
    int q = (int)__umulhi((unsigned)n, magic);
    if (q * d > n) --q;
    return q;
}

struct C11Params {
    const float* a_src;   // forward: w [Co][Ci] | data gradient: w [Co][Ci] (read transposed)
    const float* b_src;   // forward: x | data gradient: dy
    const float* bias;    // forward (nullable)
    const float* mask;    // data gradient: relu_below (nullable), same shape as dx
    float* out;           // forward: y (nullable when out2 is given) | data gradient: dx (zero-filled by the caller)
    float* out2;          // forward: relu(y) (nullable)
    int B, Ci, H, W, Co, Ho, Wo, S;
    int M, K, pixels;     // GEMM sizes: rows, contraction length, columns (= B * Ho * Wo)
    unsigned m_howo, m_wo;
};

// D[M][pixels] = A[M][K] * Bm[K][pixels]; DGRAD = false: forward, true: data gradient.  NT = 16-pixel column tiles per workgroup.
// grid.x = pixel tiles, grid.y = row blocks of 128.  LDS: two stages of { As[128][kPA], Bs[kKC][PB] }.
template <bool DGRAD, int NT>
__global__ __launch_bounds__(kThreads) void c11_gemm_kernel(const C11Params p) {
    constexpr int NPX = 16 * NT, PB = NPX + 16;  // (PB = 16 mod 32: lanes (n, k) hit banks 16 (k & 1) + n)
    constexpr int A_PER = kMT * kKC / kThreads;  // 8 A elements per thread and chunk
    constexpr int B_PER = kKC * NPX / kThreads;  // 2 (NT = 2) or 4 (NT = 4)
    __shared__ float As[2][kMT * kPA];
    __shared__ float Bs[2][kKC * PB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * kMT, n0 = blockIdx.x * NPX;
    const int HoWo = p.Ho * p.Wo;
    // this thread's B column (pixel) is the same in every chunk: decode it once
    const int bi = tid % NPX, bk0 = tid / NPX;  // rows bk0, bk0 + kThreads / NPX, ...
    const int pix = n0 + bi;
    const bool pix_ok = pix < p.pixels;
    const int pc = pix_ok ? pix : 0;
    const int pb = fdivm(pc, p.m_howo, HoWo), prem = pc - pb * HoWo;
    const int pp = fdivm(prem, p.m_wo, p.Wo), pq = prem - pp * p.Wo;
    size_t b_off;   // offset of (b, channel 0, this pixel) in the B source
    size_t b_chs;   // channel stride of the B source
    if (DGRAD) {
        b_off = (size_t)pb * p.Co * HoWo + prem;
        b_chs = (size_t)HoWo;
    } else {
        b_off = ((size_t)pb * p.Ci * p.H + (size_t)p.S * pp) * p.W + (size_t)p.S * pq;
        b_chs = (size_t)p.H * p.W;
    }
    float ra[A_PER], rb[B_PER];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int e = tid + i * kThreads;
            float v = 0.f;
            if (DGRAD) {  // A[m = ci][k = co] = w[co][ci]: consecutive threads walk ci (coalesced), stored transposed
                const int m = e % kMT, kk = e / kMT;
                if (m0 + m < p.M && k0 + kk < p.K) v = p.a_src[(size_t)(k0 + kk) * p.Ci + m0 + m];
            } else {      // A[m = co][k = ci] = w[co][ci]: consecutive threads walk ci
                const int kk = e % kKC, m = e / kKC;
                if (m0 + m < p.M && k0 + kk < p.K) v = p.a_src[(size_t)(m0 + m) * p.Ci + k0 + kk];
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int kk = bk0 + i * (kThreads / NPX);
            rb[i] = (pix_ok && k0 + kk < p.K) ? p.b_src[b_off + (size_t)(k0 + kk) * b_chs] : 0.f;
        }
    };
    auto commit = [&](int st) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int e = tid + i * kThreads;
            const int m = DGRAD ? e % kMT : e / kKC, kk = DGRAD ? e / kMT : e % kKC;
            As[st][m * kPA + kk] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) Bs[st][(bk0 + i * (kThreads / NPX)) * PB + bi] = rb[i];
    };
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int chunks = (p.K + kKC - 1) / kKC;
    fetch(0);
    commit(0);
    __syncthreads();
    for (int c = 0; c < chunks; ++c) {
        const int st = c & 1;
        if (c + 1 < chunks) fetch((c + 1) * kKC);  // in flight under this chunk's MFMAs
        const float* a = &As[st][(16 * wave + n) * kPA + kq];
        const float* bsm = &Bs[st][kq * PB + n];
#pragma unroll
        for (int j = 0; j < kKC / 4; ++j) {
            const float av = a[4 * j];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bsm[(4 * j) * PB + 16 * t], acc[t], 0, 0, 0);
        }
        if (c + 1 < chunks) {
            commit(st ^ 1);  // (the other stage: last read one iteration ago, before the barrier below)
            __syncthreads();
        }
    }
    // D tile of a wave: rows 16 wave + 4 kq + r, column n of tile t
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int px = n0 + 16 * t + n;
        if (px >= p.pixels) continue;
        const int b = fdivm(px, p.m_howo, HoWo), rem = px - b * HoWo;
        const int pr = fdivm(rem, p.m_wo, p.Wo), q = rem - pr * p.Wo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * wave + 4 * kq + r;
            if (m >= p.M) continue;
            if (DGRAD) {
                const size_t o = (((size_t)b * p.Ci + m) * p.H + (size_t)p.S * pr) * p.W + (size_t)p.S * q;
                float v = acc[t][r];
                if (p.mask && p.mask[o] <= 0.f) v = 0.f;  // relu.cpp:38
                p.out[o] = v;
            } else {
                const size_t o = ((size_t)b * p.Co + m) * HoWo + rem;
                const float v = acc[t][r] + (p.bias ? p.bias[m] : 0.f);
                if (p.out) p.out[o] = v;
                if (p.out2) p.out2[o] = v >= 0.f ? v : 0.f;  // relu.cpp:21-26
            }
        }
    }
}

// ---- weight / bias gradient: slab[slot][co][ci | bias] = sum over the slot's pixels ---------------------------------------------------
struct C11WParams {
    const float* x;
    const float* dy;
    float* slabs;     // [slots][Co][Ci + 1]
    int B, Ci, H, W, Co, Ho, Wo, S;
    int pixels, per_slot;  // B * Ho * Wo; pixels per slot (a multiple of kKC)
    unsigned m_howo, m_wo;
};

// grid.x = slots (split-K over pixels), grid.y = blocks of 128 output channels, grid.z = blocks of 128 input channels.
// A[m = co][k = pixel] = dy, Bm[k = pixel][n = ci] = xs; a wave owns one 16-co slice and all (<= 8) ci tiles of the block.
template <int NT>  // ci tiles per workgroup: Ci block = 16 * NT (<= 128)
__global__ __launch_bounds__(kThreads) void c11_wgrad_kernel(const C11WParams p) {
    constexpr int NCI = 16 * NT, PB = NCI + 16;
    constexpr int A_PER = kMT * kKC / kThreads;   // 8
    constexpr int B_PER = kKC * NCI / kThreads;   // NT
    __shared__ float As[kMT * kPA];  // (one stage: with 8 ci tiles two would not fit the 64 KB of static LDS)
    __shared__ float Bs[kKC * PB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int co0 = blockIdx.y * kMT, ci0 = blockIdx.z * NCI;
    const int HoWo = p.Ho * p.Wo;
    const int k_begin = blockIdx.x * p.per_slot, k_end = min(p.pixels, k_begin + p.per_slot);
    // every thread stages the SAME pixel column kk = tid % kKC of every chunk: its (b, p, q) advances by kKC pixels per chunk
    const int kk = tid % kKC, row0 = tid / kKC;  // rows row0, row0 + 16, ...
    float ra[A_PER], rb[B_PER];
    auto fetch = [&](int k0) {
        const int pix = k0 + kk;
        const bool ok = pix < k_end;
        const int pc = ok ? pix : 0;
        const int b = fdivm(pc, p.m_howo, HoWo), rem = pc - b * HoWo;
        const int pr = fdivm(rem, p.m_wo, p.Wo), q = rem - pr * p.Wo;
        const float* dyb = p.dy + (size_t)b * p.Co * HoWo + rem;
        const float* xb = p.x + ((size_t)b * p.Ci * p.H + (size_t)p.S * pr) * p.W + (size_t)p.S * q;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int m = row0 + i * (kThreads / kKC);
            ra[i] = (ok && co0 + m < p.Co) ? dyb[(size_t)(co0 + m) * HoWo] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int c = row0 + i * (kThreads / kKC);
            rb[i] = (ok && ci0 + c < p.Ci) ? xb[(size_t)(ci0 + c) * p.H * p.W] : 0.f;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) As[(row0 + i * (kThreads / kKC)) * kPA + kk] = ra[i];
#pragma unroll
        for (int i = 0; i < B_PER; ++i) Bs[kk * PB + row0 + i * (kThreads / kKC)] = rb[i];
    };
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const int chunks = (k_end - k_begin + kKC - 1) / kKC;
    if (chunks > 0) fetch(k_begin);
    for (int c = 0; c < chunks; ++c) {
        commit();
        __syncthreads();
        if (c + 1 < chunks) fetch(k_begin + (c + 1) * kKC);  // in flight under this chunk's MFMAs
        const float* a = &As[(16 * wave + n) * kPA + kq];
        const float* bsm = &Bs[kq * PB + n];
#pragma unroll
        for (int j = 0; j < kKC / 4; ++j) {
            const float av = a[4 * j];
            bsum += av;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bsm[(4 * j) * PB + 16 * t], acc[t], 0, 0, 0);
        }
        __syncthreads();  // (every wave is done reading before the next chunk is committed)
    }
    const int pitch = p.Ci + 1;
    float* slab = p.slabs + (size_t)blockIdx.x * p.Co * pitch;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int ci = ci0 + 16 * t + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + 16 * wave + 4 * kq + r;
            if (co < p.Co && ci < p.Ci) slab[(size_t)co * pitch + ci] = acc[t][r];
        }
    }
    if (blockIdx.z == 0) {  // bias gradient: lane (m, kq) summed its co's pixels k = kq mod 4; the four k-groups in a fixed order
        float v = bsum;
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        const int co = co0 + 16 * wave + n;
        if (kq == 0 && co < p.Co) slab[(size_t)co * pitch + p.Ci] = v;
    }
}

inline unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)d) + 1u; }

bool c11_ok(const cnn_conv2d_desc* d) {
    if (d->k != 1 || d->pad != 0 || (d->s != 1 && d->s != 2)) return false;
    if (d->Ci % 16 != 0 || d->Co % 16 != 0 || d->Ci < 32 || d->Co < 32) return false;
    const long long Ho = cnn_conv2d_out_dim(d->H, 1, d->s, 0), Wo = cnn_conv2d_out_dim(d->W, 1, d->s, 0);
    if (Ho < 1 || Wo < 1 || (long long)d->B * Ho * Wo >= (1ll << 30)) return false;
    if (const OptVal e = CNN_OPT_VAL("CONV_1X1"))
        if (atoi(e) == 0) return false;
    return true;
}

void fill_geometry(const cnn_conv2d_desc* d, C11Params* p) {
    p->B = d->B; p->Ci = d->Ci; p->H = d->H; p->W = d->W; p->Co = d->Co; p->S = d->s;
    p->Ho = cnn_conv2d_out_dim(d->H, 1, d->s, 0);
    p->Wo = cnn_conv2d_out_dim(d->W, 1, d->s, 0);
    p->pixels = d->B * p->Ho * p->Wo;
    p->m_howo = magic_of(p->Ho * p->Wo);
    p->m_wo = magic_of(p->Wo);
}

}  // namespace

namespace cnn_amd {

bool c11_supported(const cnn_conv2d_desc* d) { return c11_ok(d); }

// column tiles per workgroup: 64-pixel tiles unless that leaves compute units without a workgroup
static int pick_nt(int pixels, int row_blocks) { return ((long long)((pixels + 63) / 64) * row_blocks >= num_cus()) ? 4 : 2; }

int c11_forward(const cnn_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y, float* y_relu, hipStream_t s) {
    CNN_REQUIRE(c11_ok(d) && x && w && (y || y_relu), "conv_1x1 forward: geometry not covered / null pointer");
    C11Params p{};
    fill_geometry(d, &p);
    p.a_src = w; p.b_src = x; p.bias = bias; p.out = y; p.out2 = y_relu; p.mask = nullptr;
    p.M = d->Co; p.K = d->Ci;
    const int rb = (p.M + kMT - 1) / kMT, nt = pick_nt(p.pixels, rb);
    const dim3 grid((unsigned)((p.pixels + 16 * nt - 1) / (16 * nt)), (unsigned)rb);
    const char* name = y_relu ? "conv_1x1/fwd+relu" : "conv_1x1/fwd";
    if (nt == 4) {
        CNN_KLAUNCH(s, name, (c11_gemm_kernel<false, 4><<<grid, kThreads, 0, s>>>(p)), "B%d Ci%d %dx%d Co%d k1 s%d p0", d->B, d->Ci, d->H, d->W, d->Co, d->s);
    } else {
        CNN_KLAUNCH(s, name, (c11_gemm_kernel<false, 2><<<grid, kThreads, 0, s>>>(p)), "B%d Ci%d %dx%d Co%d k1 s%d p0", d->B, d->Ci, d->H, d->W, d->Co, d->s);
    }
    return CNN_AMD_OK;
}

int c11_backward_data(const cnn_conv2d_desc* d, const float* dy, const float* w, const float* relu_below, float* dx, hipStream_t s) {
    CNN_REQUIRE(c11_ok(d) && dy && w && dx, "conv_1x1 data gradient: geometry not covered / null pointer");
    C11Params p{};
    fill_geometry(d, &p);
    p.a_src = w; p.b_src = dy; p.bias = nullptr; p.out = dx; p.out2 = nullptr; p.mask = relu_below;
    p.M = d->Ci; p.K = d->Co;
    // positions off the sampled grid (and rows / columns behind the last window) get no contribution: conv2d.cpp:168 zero-fills
    if (d->s != 1) {
        CNN_HIP_CHECK(hipMemsetAsync(dx, 0, sizeof(float) * (size_t)d->B * d->Ci * d->H * d->W, s));
        publish_mark_stale(s);
    }
    const int rb = (p.M + kMT - 1) / kMT, nt = pick_nt(p.pixels, rb);
    const dim3 grid((unsigned)((p.pixels + 16 * nt - 1) / (16 * nt)), (unsigned)rb);
    const char* name = relu_below ? "conv_1x1/dgrad+relu" : "conv_1x1/dgrad";
    if (nt == 4) {
        CNN_KLAUNCH(s, name, (c11_gemm_kernel<true, 4><<<grid, kThreads, 0, s>>>(p)), "B%d Ci%d %dx%d Co%d k1 s%d p0", d->B, d->Ci, d->H, d->W, d->Co, d->s);
    } else {
        CNN_KLAUNCH(s, name, (c11_gemm_kernel<true, 2><<<grid, kThreads, 0, s>>>(p)), "B%d Ci%d %dx%d Co%d k1 s%d p0", d->B, d->Ci, d->H, d->W, d->Co, d->s);
    }
    return CNN_AMD_OK;
}

// split-K slabs of the weight gradient (0: geometry not covered)
int c11_wgrad_slots(const cnn_conv2d_desc* d) {
    if (!c11_ok(d)) return 0;
    const long long pixels = (long long)d->B * cnn_conv2d_out_dim(d->H, 1, d->s, 0) * cnn_conv2d_out_dim(d->W, 1, d->s, 0);
    const int blocks = ((d->Co + kMT - 1) / kMT) * ((d->Ci + 127) / 128);
    long long slots = (2 * num_cus() + blocks - 1) / blocks;  // about two workgroups per compute unit
    const long long most = (pixels + 4 * kKC - 1) / (4 * kKC);  // at least four chunks per slot
    if (slots > most) slots = most;
    if (slots > 256) slots = 256;
    if (slots < 1) slots = 1;
    return (int)slots;
}

int c11_wgrad_launch(const cnn_conv2d_desc* d, const float* x, const float* dy, float* slabs, hipStream_t s) {
    const int slots = c11_wgrad_slots(d);
    CNN_REQUIRE(slots > 0 && x && dy && slabs, "conv_1x1 weight gradient: geometry not covered / null pointer");
    C11WParams p{};
    p.x = x; p.dy = dy; p.slabs = slabs;
    p.B = d->B; p.Ci = d->Ci; p.H = d->H; p.W = d->W; p.Co = d->Co; p.S = d->s;
    p.Ho = cnn_conv2d_out_dim(d->H, 1, d->s, 0);
    p.Wo = cnn_conv2d_out_dim(d->W, 1, d->s, 0);
    p.pixels = d->B * p.Ho * p.Wo;
    p.per_slot = ((p.pixels + slots - 1) / slots + kKC - 1) / kKC * kKC;
    p.m_howo = magic_of(p.Ho * p.Wo);
    p.m_wo = magic_of(p.Wo);
    const int ci_blocks = (d->Ci + 127) / 128;
    const int nt = d->Ci >= 128 ? 8 : (d->Ci + 15) / 16;  // 16-channel tiles per ci block
    const dim3 grid((unsigned)slots, (unsigned)((d->Co + kMT - 1) / kMT), (unsigned)ci_blocks);
#define C11W(NT_) \
    CNN_KLAUNCH(s, "conv_1x1/wgrad", (c11_wgrad_kernel<NT_><<<grid, kThreads, 0, s>>>(p)), "B%d Ci%d %dx%d Co%d k1 s%d p0 slabs%d", d->B, d->Ci, d->H, d->W, d->Co, d->s, slots)
    switch (nt) {
        case 2: C11W(2); break;
        case 3: C11W(3); break;
        case 4: C11W(4); break;
        case 5: C11W(5); break;
        case 6: C11W(6); break;
        case 7: C11W(7); break;
        default: C11W(8); break;
    }
#undef C11W
    return CNN_AMD_OK;
}

}  // namespace cnn_amd
