// conv_im2col.hip -- FUNCTIONAL FALLBACK for Conv2D, kept only so the MFMA implicit-GEMM kernels have an
// independent on-device cross-check (north_star: "im2col materialised only as a functional fallback for parity
// checks").  im2col/col2im + a plain LDS-tiled fp32 VALU GEMM; not tuned, never on the bench path.
//   col[b][(ci,kx,ky)][(p,q)] = x[b][ci][p*s+kx-pad][q*s+ky-pad]   (0 outside the image)
//   fwd : y[b]    = W[Co x K] * col[b][K x P] + bias          (conv2d.cpp:69-92)
//   wgrad: gw     = (sum_b dy[b][Co x P] * col[b]^T) / divisor (conv2d.cpp:120-151), gb likewise (:153-157)
//   dgrad: dcol[b]= W^T * dy[b]; dx = col2im(dcol) in gather form (conv2d.cpp:168-199)
#include "common.h"

using namespace cnn_amd;

namespace {

struct Geo {
    int B, Ci, H, W, Co, k, s, pad, Ho, Wo, K, P;
};

Geo make_geo(const cnn_conv2d_desc* d) {
    Geo g;
    g.B = d->B; g.Ci = d->Ci; g.H = d->H; g.W = d->W; g.Co = d->Co; g.k = d->k; g.s = d->s; g.pad = d->pad;
    g.Ho = cnn_conv2d_out_dim(d->H, d->k, d->s, d->pad);
    g.Wo = cnn_conv2d_out_dim(d->W, d->k, d->s, d->pad);
    g.K = d->Ci * d->k * d->k;
    g.P = g.Ho * g.Wo;
    return g;
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

__global__ void im2col_kernel(const float* __restrict__ x, float* __restrict__ col, Geo g, int b0, int nb) {
    const long long total = (long long)nb * g.K * g.P;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(t % g.P);
        const long long r = t / g.P;
        const int kk = (int)(r % g.K);
        const int bb = (int)(r / g.K);
        const int ci = kk / (g.k * g.k), tap = kk % (g.k * g.k), kx = tap / g.k, ky = tap % g.k;
        const int h = (p / g.Wo) * g.s + kx - g.pad, w = (p % g.Wo) * g.s + ky - g.pad;
        float v = 0.f;
        if (h >= 0 && h < g.H && w >= 0 && w < g.W) v = x[(((size_t)(b0 + bb) * g.Ci + ci) * g.H + h) * g.W + w];
        col[t] = v;
    }
}

// dx[b][ci][h][w] = sum over taps and output pixels that read this element of dcol (gather, deterministic)
__global__ void col2im_kernel(const float* __restrict__ dcol, float* __restrict__ dx, Geo g, int b0, int nb) {
    const long long total = (long long)nb * g.Ci * g.H * g.W;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int w = (int)(t % g.W);
        long long r = t / g.W;
        const int h = (int)(r % g.H);
        r /= g.H;
        const int ci = (int)(r % g.Ci);
        const int bb = (int)(r / g.Ci);
        float acc = 0.f;
        for (int kx = 0; kx < g.k; ++kx) {
            const int ph = h + g.pad - kx;
            if (ph < 0 || ph % g.s) continue;
            const int p = ph / g.s;
            if (p >= g.Ho) continue;
            for (int ky = 0; ky < g.k; ++ky) {
                const int qw = w + g.pad - ky;
                if (qw < 0 || qw % g.s) continue;
                const int q = qw / g.s;
                if (q >= g.Wo) continue;
                const int kk = (ci * g.k + kx) * g.k + ky;
                acc += dcol[((size_t)bb * g.K + kk) * g.P + p * g.Wo + q];
            }
        }
        dx[(((size_t)(b0 + bb) * g.Ci + ci) * g.H + h) * g.W + w] = acc;
    }
}

// C[bt][m][n] = sum_k A(bt,m,k) * B(bt,k,n) (+ rowbias[m]); A/B addressed by element strides.
constexpr int kT = 32, kKT = 16;
__global__ __launch_bounds__(256) void sgemm_ref(const float* __restrict__ A, long long a_bs, long long a_rs,
                                                 long long a_cs, const float* __restrict__ Bm, long long b_bs,
                                                 long long b_rs, long long b_cs, float* __restrict__ C, long long c_bs,
                                                 int M, int N, int Kd, const float* __restrict__ rowbias) {
    __shared__ float As[kKT][kT + 1];
    __shared__ float Bs[kKT][kT + 1];
    const int bt = blockIdx.z;
    const float* Ab = A + bt * a_bs;
    const float* Bb = Bm + bt * b_bs;
    float* Cb = C + bt * c_bs;
    const int m0 = blockIdx.y * kT, n0 = blockIdx.x * kT;
    const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    for (int k0 = 0; k0 < Kd; k0 += kKT) {
        for (int t = threadIdx.x; t < kKT * kT; t += 256) {
            const int kk = t / kT, mm = t % kT;
            As[kk][mm] = (m0 + mm < M && k0 + kk < Kd) ? Ab[(m0 + mm) * a_rs + (k0 + kk) * a_cs] : 0.f;
            Bs[kk][mm] = (n0 + mm < N && k0 + kk < Kd) ? Bb[(k0 + kk) * b_rs + (n0 + mm) * b_cs] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kKT; ++kk) {
            const float a0 = As[kk][ty], a1 = As[kk][ty + 16], b0 = Bs[kk][tx], b1 = Bs[kk][tx + 16];
            acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[1][0] += a1 * b0; acc[1][1] += a1 * b1;
        }
        __syncthreads();
    }
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + ty + 16 * i, n = n0 + tx + 16 * j;
            if (m < M && n < N) Cb[(size_t)m * N + n] = acc[i][j] + (rowbias ? rowbias[m] : 0.f);
        }
}

// acc[i] (+)= sum_{t<nt} part[t][i]  in ascending t (deterministic)
__global__ void accumulate_parts(const float* __restrict__ part, float* __restrict__ acc, int nt, size_t n,
                                 int first) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = first ? 0.f : acc[i];
        for (int t = 0; t < nt; ++t) s += part[(size_t)t * n + i];
        acc[i] = s;
    }
}
__global__ void scale_div(const float* __restrict__ in, float* __restrict__ out, size_t n, float divisor) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = in[i] / divisor;
}

// gb[o] = (sum_b sum_pq dy[b][o][pq]) / divisor -- one workgroup per channel, fixed-order tree
__global__ __launch_bounds__(256) void bias_grad_ref(const float* __restrict__ dy, float* __restrict__ gb, int B,
                                                     int Co, int P, float divisor) {
    __shared__ float red[256];
    const int o = blockIdx.x;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* d = dy + ((size_t)b * Co + o) * P;
        for (int i = threadIdx.x; i < P; i += 256) s += d[i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) gb[o] = red[0] / divisor;
}

// workspace carve: [col: bc*K*P][parts: bc*Co*K][acc: Co*K]; bc = images per chunk
struct Carve {
    int bc;
    float *col, *parts, *acc;
};
int carve(const char* who, const Geo& g, void* ws, size_t ws_bytes, Carve* c) {
    CNN_REQUIRE(ws != nullptr, "%s: workspace is null", who);
    const size_t per_img = ((size_t)g.K * g.P + (size_t)g.Co * g.K) * sizeof(float);
    const size_t fixed = (size_t)g.Co * g.K * sizeof(float) + 256;
    if (ws_bytes < fixed + per_img)
        return fail(CNN_AMD_E_WORKSPACE, "%s: workspace %zu B < minimum %zu B", who, ws_bytes, fixed + per_img);
    size_t bc = (ws_bytes - fixed) / per_img;
    if (bc > (size_t)g.B) bc = g.B;
    if (bc > 4096) bc = 4096;
    c->bc = (int)bc;
    c->col = (float*)ws;
    c->parts = c->col + bc * g.K * g.P;
    c->acc = c->parts + bc * g.Co * g.K;
    return CNN_AMD_OK;
}

inline unsigned grid1d(long long n) {
    long long b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

}  // namespace

extern "C" {

size_t cnn_conv2d_im2col_workspace_bytes(const cnn_conv2d_desc* d) {
    if (!d) return 0;
    const Geo g = make_geo(d);
    const size_t per_img = ((size_t)g.K * g.P + (size_t)g.Co * g.K) * sizeof(float);
    const size_t fixed = (size_t)g.Co * g.K * sizeof(float) + 256;
    // aim for <= 1 GiB of column buffer, at least one image
    size_t bc = (size_t)1 << 30;
    bc = bc / per_img;
    if (bc < 1) bc = 1;
    if (bc > (size_t)g.B) bc = g.B;
    return fixed + bc * per_img;
}

int cnn_conv2d_forward_im2col(const cnn_conv2d_desc* d, const float* x, const float* w, const float* bias, float* y,
                              void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_desc("cnn_conv2d_forward_im2col", d)) return rc;
    CNN_REQUIRE(x && w && bias && y, "cnn_conv2d_forward_im2col: null pointer");
    const Geo g = make_geo(d);
    Carve c;
    if (int rc = carve("cnn_conv2d_forward_im2col", g, ws, ws_bytes, &c)) return rc;
    hipStream_t s = as_stream(stream);
    for (int b0 = 0; b0 < g.B; b0 += c.bc) {
        const int nb = (g.B - b0 < c.bc) ? g.B - b0 : c.bc;
        im2col_kernel<<<grid1d((long long)nb * g.K * g.P), 256, 0, s>>>(x, c.col, g, b0, nb);
        CNN_LAUNCH_CHECK();
        dim3 grid(ceil_div(g.P, kT), ceil_div(g.Co, kT), nb);
        sgemm_ref<<<grid, 256, 0, s>>>(w, 0, g.K, 1, c.col, (long long)g.K * g.P, g.P, 1,
                                       y + (size_t)b0 * g.Co * g.P, (long long)g.Co * g.P, g.Co, g.P, g.K, bias);
        CNN_LAUNCH_CHECK();
    }
    publish_mark_stale(s);  // (plain <<<>>> launches: not tracked by CNN_KLAUNCH)
    return CNN_AMD_OK;
}

int cnn_conv2d_backward_weight_im2col(const cnn_conv2d_desc* d, const float* x, const float* dy, float* gw, float* gb,
                                      float divisor, void* ws, size_t ws_bytes, void* stream) {
    if (int rc = check_desc("cnn_conv2d_backward_weight_im2col", d)) return rc;
    CNN_REQUIRE(x && dy && gw, "cnn_conv2d_backward_weight_im2col: null pointer");
    const Geo g = make_geo(d);
    Carve c;
    if (int rc = carve("cnn_conv2d_backward_weight_im2col", g, ws, ws_bytes, &c)) return rc;
    hipStream_t s = as_stream(stream);
    const size_t n = (size_t)g.Co * g.K;
    for (int b0 = 0; b0 < g.B; b0 += c.bc) {
        const int nb = (g.B - b0 < c.bc) ? g.B - b0 : c.bc;
        im2col_kernel<<<grid1d((long long)nb * g.K * g.P), 256, 0, s>>>(x, c.col, g, b0, nb);
        CNN_LAUNCH_CHECK();
        dim3 grid(ceil_div(g.K, kT), ceil_div(g.Co, kT), nb);
        // A = dy[b] (Co x P), B = col[b]^T (P x K): element (p,kk) at col[kk*P + p]
        sgemm_ref<<<grid, 256, 0, s>>>(dy + (size_t)b0 * g.Co * g.P, (long long)g.Co * g.P, g.P, 1, c.col,
                                       (long long)g.K * g.P, 1, g.P, c.parts, (long long)n, g.Co, g.K, g.P, nullptr);
        CNN_LAUNCH_CHECK();
        accumulate_parts<<<grid1d((long long)n), 256, 0, s>>>(c.parts, c.acc, nb, n, b0 == 0);
        CNN_LAUNCH_CHECK();
    }
    scale_div<<<grid1d((long long)n), 256, 0, s>>>(c.acc, gw, n, divisor);
    CNN_LAUNCH_CHECK();
    if (gb) {
        bias_grad_ref<<<g.Co, 256, 0, s>>>(dy, gb, g.B, g.Co, g.P, divisor);
        CNN_LAUNCH_CHECK();
    }
    publish_mark_stale(s);  // (plain <<<>>> launches: not tracked by CNN_KLAUNCH)
    return CNN_AMD_OK;
}

int cnn_conv2d_backward_data_im2col(const cnn_conv2d_desc* d, const float* dy, const float* w, float* dx, void* ws,
                                    size_t ws_bytes, void* stream) {
    if (int rc = check_desc("cnn_conv2d_backward_data_im2col", d)) return rc;
    CNN_REQUIRE(dy && w && dx, "cnn_conv2d_backward_data_im2col: null pointer");
    const Geo g = make_geo(d);
    Carve c;
    if (int rc = carve("cnn_conv2d_backward_data_im2col", g, ws, ws_bytes, &c)) return rc;
    hipStream_t s = as_stream(stream);
    for (int b0 = 0; b0 < g.B; b0 += c.bc) {
        const int nb = (g.B - b0 < c.bc) ? g.B - b0 : c.bc;
        dim3 grid(ceil_div(g.P, kT), ceil_div(g.K, kT), nb);
        // A = W^T (K x Co): element (kk,co) at w[co*K + kk]; B = dy[b] (Co x P); C = dcol[b] (K x P)
        sgemm_ref<<<grid, 256, 0, s>>>(w, 0, 1, g.K, dy + (size_t)b0 * g.Co * g.P, (long long)g.Co * g.P, g.P, 1, c.col,
                                       (long long)g.K * g.P, g.K, g.P, g.Co, nullptr);
        CNN_LAUNCH_CHECK();
        col2im_kernel<<<grid1d((long long)nb * g.Ci * g.H * g.W), 256, 0, s>>>(c.col, dx, g, b0, nb);
        CNN_LAUNCH_CHECK();
    }
    publish_mark_stale(s);  // (plain <<<>>> launches: not tracked by CNN_KLAUNCH)
    return CNN_AMD_OK;
}

}  // extern "C"
