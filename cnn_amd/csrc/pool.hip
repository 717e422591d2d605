// pool.hip -- MaxPool2D forward / backward (cpu/src/pool2d.cpp:53-87, 96-107).
// HBM-bound, one wavefront per image row so that a wave's loads/stores are one contiguous run of the NCHW row;
// no per-element integer division (row -> (plane,h) is wave-uniform scalar arithmetic).
//   fwd algorithmic bytes: 4*C*(H*W + 2*Ho*Wo) per sample (input + output + int32 mask)
//   bwd algorithmic bytes: 4*C*(2*Ho*Wo + H*W) per sample (delta + mask + dx written once, no separate memset)
#include <cstdint>
#include <cstdlib>

#include "common.h"

using namespace cnn_amd;

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWave * kWavesPerBlock;

// One wave per OUTPUT row (plane, ho).  Window scan order and the strict '<' are the reference's
// (pool2d.cpp:67-75): max starts at window[0]; a later element replaces it only if max < comp, so the first
// maximum wins ties, a NaN in window[0] is never replaced and a later NaN never wins, -0.0/+0.0 tie.
template <int K, int STEP>  // 0 = runtime value
__global__ __launch_bounds__(kBlock) void maxpool_fwd_rows(const float* __restrict__ x, float* __restrict__ y,
                                                           int32_t* __restrict__ mask, long long n_rows, int C,
                                                           int H, int W, int Ho, int Wo, int k_rt, int step_rt) {
    const int k = K ? K : k_rt;
    const int step = STEP ? STEP : step_rt;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    for (long long r = (long long)blockIdx.x * kWavesPerBlock + wave; r < n_rows;
         r += (long long)gridDim.x * kWavesPerBlock) {
        const long long plane = r / Ho;  // = b*C + c
        const int ho = (int)(r - plane * Ho);
        const int c = (int)(plane % C);
        const float* xrow = x + plane * H * W + (size_t)(ho * step) * W;
        float* yrow = y + r * Wo;
        int32_t* mrow = mask ? mask + r * Wo : nullptr;
        const int mbase = c * H * W + ho * step * W;
        for (int wo = lane; wo < Wo; wo += kWave) {
            const float* win = xrow + wo * step;
            float best = win[0];
            int best_off = 0;
#pragma unroll
            for (int di = 0; di < k; ++di) {
#pragma unroll
                for (int dj = 0; dj < k; ++dj) {
                    if (di == 0 && dj == 0) continue;
                    const float comp = win[di * W + dj];
                    if (best < comp) {
                        best = comp;
                        best_off = di * W + dj;
                    }
                }
            }
            yrow[wo] = best;
            if (mrow) mrow[wo] = mbase + wo * step + best_off;
        }
    }
}

// One wave per INPUT row (plane, h): dx[h][w] = dy of the highest-index window whose recorded argmax is this
// element, else 0 -- the gather form of "dx = 0; for i ascending: dx[mask[i]] = dy[i]" (pool2d.cpp:96-107).
// With k <= step (the only configuration the reference net uses) at most one window covers an element.
// `pooled` (nullable) fuses the ReLU::backward of the layer in front of the pool (relu.cpp:35-40): the pool's input IS
// that ReLU's output, and at an argmax position its value is the pooled output itself, so "output <= 0 ? 0 : delta" needs
// only pooled[window]; everywhere else the delta is 0 either way.
template <int K, int STEP>
__global__ __launch_bounds__(kBlock) void maxpool_bwd_rows(const float* __restrict__ dy,
                                                           const int32_t* __restrict__ mask,
                                                           const float* __restrict__ pooled, float* __restrict__ dx,
                                                           long long n_rows, int C, int H, int W, int Ho, int Wo,
                                                           int k_rt, int step_rt) {
    const int k = K ? K : k_rt;
    const int step = STEP ? STEP : step_rt;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    for (long long r = (long long)blockIdx.x * kWavesPerBlock + wave; r < n_rows;
         r += (long long)gridDim.x * kWavesPerBlock) {
        const long long plane = r / H;
        const int h = (int)(r - plane * H);
        const int c = (int)(plane % C);
        // windows (rows) that contain h: ho*step <= h <= ho*step + k - 1
        int ho_hi = h / step;
        if (ho_hi > Ho - 1) ho_hi = Ho - 1;
        int ho_lo = (h - k + 1 + step - 1);
        ho_lo = ho_lo <= 0 ? 0 : ho_lo / step;
        const int32_t self_row = c * H * W + h * W;
        const float* dplane = dy + plane * Ho * Wo;
        const float* pplane = pooled ? pooled + plane * Ho * Wo : nullptr;
        const int32_t* mplane = mask + plane * Ho * Wo;
        float* out = dx + r * W;
        for (int w = lane; w < W; w += kWave) {
            int wo_hi = w / step;
            if (wo_hi > Wo - 1) wo_hi = Wo - 1;
            int wo_lo = (w - k + 1 + step - 1);
            wo_lo = wo_lo <= 0 ? 0 : wo_lo / step;
            float v = 0.f;
            bool found = false;
            for (int ho = ho_hi; ho >= ho_lo && !found; --ho)
                for (int wo = wo_hi; wo >= wo_lo; --wo) {
                    if ((mplane[ho * Wo + wo] & 0x7fffffff) == self_row + w) {  // (bit 31: see cnn_conv2d_relu_maxpool2_forward)
                        v = dplane[ho * Wo + wo];
                        if (pplane && pplane[ho * Wo + wo] <= 0.f) v = 0.f;
                        found = true;
                        break;
                    }
                }
            out[w] = v;
        }
    }
}

// k = step = 2 (the reference net's only pool): one wave per OUTPUT row writes BOTH input rows it covers, so dy and
// mask are read exactly once; rows/cols past the last window (H or W odd) are zero-filled by the same waves.
__global__ __launch_bounds__(kBlock) void maxpool_bwd_k2s2(const float* __restrict__ dy, const int32_t* __restrict__ mask,
                                                           const float* __restrict__ pooled, float* __restrict__ dx,
                                                           long long n_units, int C, int H, int W, int Ho, int Wo) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int extra = H - 2 * Ho;          // 0 or 1 uncovered row at the bottom
    const int upp = Ho + extra;            // units per plane
    for (long long u = (long long)blockIdx.x * kWavesPerBlock + wave; u < n_units;
         u += (long long)gridDim.x * kWavesPerBlock) {
        const long long plane = u / upp;
        const int r = (int)(u - plane * upp);
        const int c = (int)(plane % C);
        float* xplane = dx + plane * H * W;
        if (r >= Ho) {  // bottom row no window covers
            float* row = xplane + (size_t)(2 * Ho + (r - Ho)) * W;
            for (int w = lane; w < W; w += kWave) row[w] = 0.f;
            continue;
        }
        const float* drow = dy + (plane * Ho + r) * Wo;
        const int32_t* mrow = mask + (plane * Ho + r) * Wo;
        const float* prow = pooled ? pooled + (plane * Ho + r) * Wo : nullptr;
        float* row0 = xplane + (size_t)(2 * r) * W;
        float* row1 = row0 + W;
        const int32_t base = c * H * W + 2 * r * W;
        for (int wo = lane; wo < Wo; wo += kWave) {
            float d = drow[wo];
            if (prow && prow[wo] <= 0.f) d = 0.f;  // fused ReLU::backward (relu.cpp:38) of the layer in front
            const int off = (mrow[wo] & 0x7fffffff) - base - 2 * wo;  // 0, 1, W or W+1 (bit 31: see cnn_conv2d_relu_maxpool2_forward)
            row0[2 * wo] = (off == 0) ? d : 0.f;
            row0[2 * wo + 1] = (off == 1) ? d : 0.f;
            row1[2 * wo] = (off == W) ? d : 0.f;
            row1[2 * wo + 1] = (off == W + 1) ? d : 0.f;
        }
        for (int w = 2 * Wo + lane; w < W; w += kWave) {  // right-hand columns no window covers
            row0[w] = 0.f;
            row1[w] = 0.f;
        }
    }
}

// Narrow planes (Wo <= 32: the 56x56 / 28x28 / 14x14 pools of the VGG-shaped stack): one output row fills only a few of a
// wavefront's lanes, so the wavefront is cut into 64 / LPR parts of LPR = 2^shift lanes and every part takes its own row (rows of
// one plane are adjacent in memory, so a wavefront's accesses stay contiguous runs).  Same window order / strict '<' / mask
// encoding as maxpool_fwd_rows<2,2> and the same outputs as maxpool_bwd_k2s2, element for element.
__global__ __launch_bounds__(kBlock) void maxpool_fwd_k2s2_packed(const float* __restrict__ x, float* __restrict__ y,
                                                                  int32_t* __restrict__ mask, unsigned n_rows, int C, int H,
                                                                  int W, int Ho, int Wo, int shift) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int lpr = 1 << shift, sub = lane >> shift, l = lane & (lpr - 1);
    const unsigned rpw = (unsigned)(kWave >> shift);
    for (unsigned g = blockIdx.x * kWavesPerBlock + wave; g * rpw < n_rows; g += gridDim.x * kWavesPerBlock) {
        const unsigned r = g * rpw + sub;
        if (r >= n_rows) continue;
        const unsigned plane = r / (unsigned)Ho;
        const int ho = (int)(r - plane * Ho);
        const int c = (int)(plane % (unsigned)C);
        const float* xrow = x + (size_t)plane * H * W + (size_t)(ho * 2) * W;
        float* yrow = y + (size_t)r * Wo;
        int32_t* mrow = mask ? mask + (size_t)r * Wo : nullptr;
        const int mbase = c * H * W + ho * 2 * W;
        for (int wo = l; wo < Wo; wo += lpr) {
            const float* win = xrow + wo * 2;
            float best = win[0];
            int best_off = 0;
            float comp = win[1];
            if (best < comp) { best = comp; best_off = 1; }
            comp = win[W];
            if (best < comp) { best = comp; best_off = W; }
            comp = win[W + 1];
            if (best < comp) { best = comp; best_off = W + 1; }
            yrow[wo] = best;
            if (mrow) mrow[wo] = mbase + wo * 2 + best_off;
        }
    }
}

__global__ __launch_bounds__(kBlock) void maxpool_bwd_k2s2_packed(const float* __restrict__ dy, const int32_t* __restrict__ mask,
                                                                  const float* __restrict__ pooled, float* __restrict__ dx,
                                                                  unsigned n_units, int C, int H, int W, int Ho, int Wo,
                                                                  int shift) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int lpr = 1 << shift, sub = lane >> shift, l = lane & (lpr - 1);
    const unsigned rpw = (unsigned)(kWave >> shift);
    const int extra = H - 2 * Ho;
    const unsigned upp = (unsigned)(Ho + extra);
    for (unsigned g = blockIdx.x * kWavesPerBlock + wave; g * rpw < n_units; g += gridDim.x * kWavesPerBlock) {
        const unsigned u = g * rpw + sub;
        if (u >= n_units) continue;
        const unsigned plane = u / upp;
        const int r = (int)(u - plane * upp);
        const int c = (int)(plane % (unsigned)C);
        float* xplane = dx + (size_t)plane * H * W;
        if (r >= Ho) {
            float* row = xplane + (size_t)(2 * Ho + (r - Ho)) * W;
            for (int w = l; w < W; w += lpr) row[w] = 0.f;
            continue;
        }
        const size_t orow = ((size_t)plane * Ho + r) * Wo;
        const float* drow = dy + orow;
        const int32_t* mrow = mask + orow;
        const float* prow = pooled ? pooled + orow : nullptr;
        float* row0 = xplane + (size_t)(2 * r) * W;
        float* row1 = row0 + W;
        const int32_t base = c * H * W + 2 * r * W;
        for (int wo = l; wo < Wo; wo += lpr) {
            float d = drow[wo];
            if (prow && prow[wo] <= 0.f) d = 0.f;
            const int off = (mrow[wo] & 0x7fffffff) - base - 2 * wo;
            row0[2 * wo] = (off == 0) ? d : 0.f;
            row0[2 * wo + 1] = (off == 1) ? d : 0.f;
            row1[2 * wo] = (off == W) ? d : 0.f;
            row1[2 * wo + 1] = (off == W + 1) ? d : 0.f;
        }
        for (int w = 2 * Wo + l; w < W; w += lpr) {
            row0[w] = 0.f;
            row1[w] = 0.f;
        }
    }
}

// lanes per row of the packed kernels: the smallest power of two >= Wo; 0 = rows are wide enough for one wavefront each
inline int packed_shift(int Wo, long long units) {
    const bool off = CNN_OPT_SET("POOL_NO_PACK");  // (A/B switch)
    if (off || Wo > 32 || units >= (1ll << 31) - 64) return -1;
    int sh = 0;
    while ((1 << sh) < Wo) ++sh;
    return sh;
}

inline unsigned row_grid(long long rows) {
    long long need = (rows + kWavesPerBlock - 1) / kWavesPerBlock;
    long long cap = (long long)num_cus() * 16;
    return (unsigned)(need < 1 ? 1 : (need > cap ? cap : need));
}

int check_geom(const char* who, int B, int C, int H, int W, int k, int step) {
    CNN_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && k > 0 && step > 0, "%s: bad dims B=%d C=%d H=%d W=%d k=%d step=%d",
                who, B, C, H, W, k, step);
    CNN_REQUIRE(H >= k && W >= k, "%s: window %d larger than %dx%d input", who, k, H, W);
    CNN_REQUIRE((long long)C * H * W < (1ll << 31), "%s: C*H*W overflows the int32 mask", who);
    return CNN_AMD_OK;
}

}  // namespace

extern "C" {

int cnn_maxpool2d_forward(const float* x, float* y, int32_t* mask, int B, int C, int H, int W, int k, int step,
                          void* stream) {
    CNN_REQUIRE(x && y, "cnn_maxpool2d_forward: null pointer");
    if (int rc = check_geom("cnn_maxpool2d_forward", B, C, H, W, k, step)) return rc;
    const int Ho = cnn_maxpool2d_out_dim(H, k, step), Wo = cnn_maxpool2d_out_dim(W, k, step);
    const long long rows = (long long)B * C * Ho;
    hipStream_t s = as_stream(stream);
#define POOL_TAG "B%d C%d %dx%d k%d step%d mask%d", B, C, H, W, k, step, mask ? 1 : 0
    const int psh = (k == 2 && step == 2) ? packed_shift(Wo, rows) : -1;
    if (psh >= 0) {
        const long long groups = (rows + (kWave >> psh) - 1) / (kWave >> psh);
        CNN_KLAUNCH(s, "maxpool_fwd_k2s2_packed",
                    (maxpool_fwd_k2s2_packed<<<row_grid(groups), kBlock, 0, s>>>(x, y, mask, (unsigned)rows, C, H, W, Ho, Wo, psh)), POOL_TAG);
    } else if (k == 2 && step == 2)
        CNN_KLAUNCH(s, "maxpool_fwd_rows<2,2>",
                    (maxpool_fwd_rows<2, 2><<<row_grid(rows), kBlock, 0, s>>>(x, y, mask, rows, C, H, W, Ho, Wo, k, step)), POOL_TAG);
    else if (k == 3 && step == 2)
        CNN_KLAUNCH(s, "maxpool_fwd_rows<3,2>",
                    (maxpool_fwd_rows<3, 2><<<row_grid(rows), kBlock, 0, s>>>(x, y, mask, rows, C, H, W, Ho, Wo, k, step)), POOL_TAG);
    else
        CNN_KLAUNCH(s, "maxpool_fwd_rows<0,0>",
                    (maxpool_fwd_rows<0, 0><<<row_grid(rows), kBlock, 0, s>>>(x, y, mask, rows, C, H, W, Ho, Wo, k, step)), POOL_TAG);
    return CNN_AMD_OK;
}

static int maxpool_backward_impl(const char* who, const float* dy, const int32_t* mask, const float* pooled, float* dx,
                                 int B, int C, int H, int W, int k, int step, void* stream) {
    CNN_REQUIRE(dy && mask && dx, "%s: null pointer", who);
    if (int rc = check_geom(who, B, C, H, W, k, step)) return rc;
    const int Ho = cnn_maxpool2d_out_dim(H, k, step), Wo = cnn_maxpool2d_out_dim(W, k, step);
    const long long rows = (long long)B * C * H;
    hipStream_t s = as_stream(stream);
    if (k == 2 && step == 2) {
        const long long units = (long long)B * C * (Ho + (H - 2 * Ho));
        const int psh = packed_shift(Wo, units);
        if (psh >= 0) {
            const long long groups = (units + (kWave >> psh) - 1) / (kWave >> psh);
            CNN_KLAUNCH(s, pooled ? "maxpool_bwd_k2s2_packed+relu" : "maxpool_bwd_k2s2_packed",
                        (maxpool_bwd_k2s2_packed<<<row_grid(groups), kBlock, 0, s>>>(dy, mask, pooled, dx, (unsigned)units, C, H, W, Ho, Wo, psh)),
                        POOL_TAG);
            return CNN_AMD_OK;
        }
        CNN_KLAUNCH(s, pooled ? "maxpool_bwd_k2s2+relu" : "maxpool_bwd_k2s2",
                    (maxpool_bwd_k2s2<<<row_grid(units), kBlock, 0, s>>>(dy, mask, pooled, dx, units, C, H, W, Ho, Wo)), POOL_TAG);
    } else
        CNN_KLAUNCH(s, pooled ? "maxpool_bwd_rows<0,0>+relu" : "maxpool_bwd_rows<0,0>",
                    (maxpool_bwd_rows<0, 0><<<row_grid(rows), kBlock, 0, s>>>(dy, mask, pooled, dx, rows, C, H, W, Ho, Wo, k, step)),
                    POOL_TAG);
    return CNN_AMD_OK;
}

int cnn_maxpool2d_backward(const float* dy, const int32_t* mask, float* dx, int B, int C, int H, int W, int k,
                           int step, void* stream) {
    return maxpool_backward_impl("cnn_maxpool2d_backward", dy, mask, nullptr, dx, B, C, H, W, k, step, stream);
}

int cnn_maxpool2d_backward_relu(const float* dy, const int32_t* mask, const float* pooled, float* dx, int B, int C, int H,
                                int W, int k, int step, void* stream) {
    CNN_REQUIRE(pooled, "cnn_maxpool2d_backward_relu: null pointer");
    return maxpool_backward_impl("cnn_maxpool2d_backward_relu", dy, mask, pooled, dx, B, C, H, W, k, step, stream);
}

}  // extern "C"
