// batchnorm.hip -- BatchNorm2D forward / backward (cpu/src/batchnorm2d.cpp:24-95, 98-158); SURVEY.md 8(f) row n1.
// HBM-bound per-channel reductions over the B strided H*W planes of an NCHW tensor plus an elementwise pass.
//
// Work decomposition: a UNIT is up to kSeg consecutive floats of one (b, c) plane; a wavefront owns whole units, so
// the plane lookup is wave-uniform scalar arithmetic and a wave's loads are one contiguous run.  Planes start at
// arbitrary 4-byte offsets (H*W is odd for every conv output of the reference net), so a unit is walked in 16-byte
// SLOTS aligned to the tensor base: interior slots are one dwordx4 access, the two edge slots fall back to
// per-component guarded accesses.  Grid = (G, C): block (g, c) reduces units g*4+w, +4G, ... of channel c into
// partial sums part[c][g][k]; the consumer kernel re-reduces the G partials with a fixed lane/xor tree in every
// block, so results are deterministic and no finalize launch is needed.
//
//   forward, training: 2 launches  (sums around a pilot value -> mean, var | apply)  8 + 4 = 12 B / element
//                                  (round 5: one statistics pass, bn_stats_pilot; x read 2x, y written once.  The two-pass form
//                                   -- sum -> mean | sum (x-u)^2 -> var, 16 B / element -- stays behind CNN_AMD_BN_TWO_PASS=1)
//   forward, eval:     1 launch   (apply with the moving statistics)               8 B / element
//   backward:          2 launches (4 sums | in-place dx)                           8 + 12 = 20 B / element
// The reference's normed_input buffer (batchnorm2d.cpp:38,71) is NOT materialised: (x-u)*var_inv is recomputed
// from x and the saved batch statistics, which is the same arithmetic and saves a 4 B/element write + read.
#include <cstdint>
#include <cstdlib>

#include "common.h"

using namespace cnn_amd;

namespace {

constexpr int kWaves = 4;
constexpr int kBlock = kWave * kWaves;
constexpr int kSeg = 2048;   // floats per unit (multiple of 4)
constexpr int kMaxG = 256;   // partial-sum slots per channel (<= 4 per lane in the re-reduction)

struct Geo {
    int B, C, HW, nseg, G;
    long long units;  // per channel = B * nseg
    // small planes (H*W <= 60: the 7x7 / 6x6 / 3x3 ends of the stacks): a plane fills only a few of a wavefront's 64 slots, so a
    // wavefront is cut into P PARTS of 64/P lanes and each part walks its own unit; P = 1 for everything larger.
    int P, part_shift;  // 64/P = 1 << part_shift
};

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// deterministic sum of part[0..G) (stride `stride` floats), identical in every block that calls it
__device__ inline float sum_partials(const float* __restrict__ part, int G, int stride, int lane) {
    float v = 0.f;
    for (int g = lane; g < G; g += kWave) v += part[(size_t)g * stride];
    return wave_sum(v);
}

// walk the unit's 16-byte slots; f4(slot_float_index) handles a full slot, f1(float_index) a single element
template <class F4, class F1>
__device__ inline void walk_unit(long long g0, long long g1, int lane, int stride, F4&& f4, F1&& f1) {
    const long long slot0 = g0 >> 2;
    const int nslots = (int)(((g1 + 3) >> 2) - slot0);
    for (int t = lane; t < nslots; t += stride) {
        const long long e = (slot0 + t) << 2;
        if (e >= g0 && e + 4 <= g1) {
            f4(e);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (e + k >= g0 && e + k < g1) f1(e + k);
        }
    }
}

// unit u of channel c -> [g0, g1) as float indices from the tensor base
__device__ inline void unit_range(const Geo& q, int c, long long u, long long& g0, long long& g1) {
    const int b = q.nseg == 1 ? (int)u : (int)(u / q.nseg);
    const int sg = (int)(u - (long long)b * q.nseg);
    const long long base = ((long long)b * q.C + c) * q.HW;
    const int s0 = sg * kSeg;
    const int s1 = s0 + kSeg < q.HW ? s0 + kSeg : q.HW;
    g0 = base + s0;
    g1 = base + s1;
}

// block-level: sum NS per-lane accumulators over the 4 waves -> part[(c*G+g)*NS + k]
template <int NS>
__device__ inline void block_store_partials(float (&acc)[NS], float* __restrict__ part, int G) {
    __shared__ float red[kWaves][NS];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const float v = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NS) {
        float v = red[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) v += red[w][threadIdx.x];
        part[((size_t)blockIdx.y * G + blockIdx.x) * NS + threadIdx.x] = v;
    }
}

// MODE 0: sum x            (batchnorm2d.cpp:48-55)
// MODE 1: sum (x - u)^2    (batchnorm2d.cpp:57-63); u = (sum of part_in) / L, block g == 0 also publishes saved_mean
// MODE 2: like 1, but around mean = gsum[c] / count with gsum the ALL-REDUCED channel sums of a sharded batch (sync-BN)
template <int MODE>
__global__ __launch_bounds__(kBlock) void bn_stats(const float* __restrict__ x, const float* __restrict__ part_in,
                                                   float* __restrict__ part_out, float* __restrict__ saved_mean,
                                                   Geo q, float count) {
    const int c = blockIdx.y, g = blockIdx.x;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int wpart = lane >> q.part_shift, stride = 1 << q.part_shift, sub = lane & (stride - 1);
    float u = 0.f;
    if (MODE == 1) {
        u = sum_partials(part_in + (size_t)c * q.G, q.G, 1, lane) / (float)((long long)q.B * q.HW);
        if (g == 0 && threadIdx.x == 0) saved_mean[c] = u;
    }
    if (MODE == 2) u = part_in[c] / count;
    float acc[1] = {0.f};
    for (long long un = ((long long)g * kWaves + wave) * q.P + wpart; un < q.units; un += (long long)q.G * kWaves * q.P) {
        long long g0, g1;
        unit_range(q, c, un, g0, g1);
        walk_unit(
            g0, g1, sub, stride,
            [&](long long e) {
                const float4 v = *(const float4*)(x + e);
                if (MODE == 0) {
                    acc[0] += (v.x + v.y) + (v.z + v.w);
                } else {
                    const float a = v.x - u, b = v.y - u, cc = v.z - u, d = v.w - u;
                    acc[0] += (a * a + b * b) + (cc * cc + d * d);
                }
            },
            [&](long long e) {
                const float v = x[e];
                acc[0] += MODE == 0 ? v : (v - u) * (v - u);  // MODE 1 / 2
            });
    }
    block_store_partials<1>(acc, part_out, q.G);
}

// ONE pass instead of two (round 5): both batch statistics from a single read of x, around a per-channel PILOT value p = the first
// element of the channel (any member of the distribution will do): S1 = sum (x - p), S2 = sum (x - p)^2, then
//     mean = p + S1/L        var = S2/L - (S1/L)^2           (batchnorm2d.cpp:46-61 computes the same two numbers in two passes)
// The textbook one-pass formula E[x^2] - E[x]^2 cancels catastrophically when |mean| >> sigma; shifted by a sample of the channel
// |S1/L| is O(sigma) whatever the channel's offset, and the subtraction costs a few ulps (measured against the oracle: <= 2e-6
// tensor-normalised on y, tests/test_gpu_batchnorm.py, the offset cases included).  CNN_AMD_BN_TWO_PASS=1 keeps the two-pass kernels.
// The pilot value is PUBLISHED (pilot_out[c], written by the channel's first workgroup): bn_apply takes it from there, not from x -- the
// API allows y == x / y_relu == x, and an apply pass that re-read x[c*HW] could find it already overwritten by the workgroup that owns
// that element.
__global__ __launch_bounds__(kBlock) void bn_stats_pilot(const float* __restrict__ x, float* __restrict__ part_out, float* __restrict__ pilot_out, Geo q) {
    const int c = blockIdx.y, g = blockIdx.x;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int wpart = lane >> q.part_shift, stride = 1 << q.part_shift, sub = lane & (stride - 1);
    const float p = x[(size_t)c * q.HW];
    if (g == 0 && threadIdx.x == 0) pilot_out[c] = p;
    float acc[2] = {0.f, 0.f};
    for (long long un = ((long long)g * kWaves + wave) * q.P + wpart; un < q.units; un += (long long)q.G * kWaves * q.P) {
        long long g0, g1;
        unit_range(q, c, un, g0, g1);
        walk_unit(
            g0, g1, sub, stride,
            [&](long long e) {
                const float4 v = *(const float4*)(x + e);
                const float a = v.x - p, b = v.y - p, cc = v.z - p, d = v.w - p;
                acc[0] += (a + b) + (cc + d);
                acc[1] += (a * a + b * b) + (cc * cc + d * d);
            },
            [&](long long e) {
                const float a = x[e] - p;
                acc[0] += a;
                acc[1] += a * a;
            });
    }
    block_store_partials<2>(acc, part_out, q.G);
}

struct BnApply {
    const float* x;
    float* y;
    float* y_relu;  // RELU kernels: the output of the ReLU layer behind this one (relu.cpp:25), written by the same pass
    const float* gamma;
    const float* beta;
    float* moving_mean;
    float* moving_var;
    float* saved_mean;  // training: input (published by bn_stats<1>); unused in eval
    float* saved_var;   // training: output
    const float* part;  // training: partial sums of (x-u)^2
    float eps, momentum;
    int training;
    const float* gsum_x;   // sync-BN: all-reduced channel sums of x and of (x-mean)^2, with the global element count
    const float* gsum_sq;
    float count;
    const float* part2;  // training, one-pass statistics: partial sums (S1, S2) around the channel's pilot value (bn_stats_pilot)
    const float* pilot;  // ... and the pilot values themselves [C]
};

// y = gamma * ((x - u) * var_inv) + beta  (batchnorm2d.cpp:69-77 / 84-92), moving statistics (:79-80)
__device__ __forceinline__ float bn_relu(float v) { return v >= 0.f ? v : 0.f; }  // == elementwise.hip relu_f (relu.cpp:25)

// the channel's mean / variance of this pass (every workgroup of the channel computes the same two numbers from the same partials in the same
// order), and -- workgroup 0 of the channel -- the saved / moving statistics (batchnorm2d.cpp:64-66, 79-80)
__device__ __forceinline__ void bn_channel_stats(const BnApply& a, const Geo& q, int c, int g, int lane, float& u, float& var) {
#pragma clang fp contract(off)
    if (a.training) {
        if (a.gsum_x != nullptr) {
            u = a.gsum_x[c] / a.count;
            var = a.gsum_sq[c] / a.count;
            if (g == 0 && threadIdx.x == 0) a.saved_mean[c] = u;
        } else if (a.part2 != nullptr) {
            const float L = (float)((long long)q.B * q.HW);
            const float d = sum_partials(a.part2 + (size_t)c * q.G * 2, q.G, 2, lane) / L;
            const float m2 = sum_partials(a.part2 + (size_t)c * q.G * 2 + 1, q.G, 2, lane) / L;
            u = a.pilot[c] + d;
            var = m2 - d * d;
            var = var > 0.f ? var : 0.f;
            if (g == 0 && threadIdx.x == 0) a.saved_mean[c] = u;
        } else {
            u = a.saved_mean[c];
            var = sum_partials(a.part + (size_t)c * q.G, q.G, 1, lane) / (float)((long long)q.B * q.HW);
        }
        if (g == 0 && threadIdx.x == 0) {
            a.saved_var[c] = var;
            a.moving_mean[c] = (1.f - a.momentum) * a.moving_mean[c] + a.momentum * u;
            a.moving_var[c] = (1.f - a.momentum) * a.moving_var[c] + a.momentum * var;
        }
    } else {
        u = a.moving_mean[c];
        var = a.moving_var[c];
    }
}

template <bool RELU>
__global__ __launch_bounds__(kBlock) void bn_apply(BnApply a, Geo q) {
#pragma clang fp contract(off)
    const int c = blockIdx.y, g = blockIdx.x;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int wpart = lane >> q.part_shift, stride = 1 << q.part_shift, sub = lane & (stride - 1);
    float u, var;
    bn_channel_stats(a, q, c, g, lane, u, var);
    const float var_inv = 1.f / sqrtf(var + a.eps);
    const float gm = a.gamma[c], bt = a.beta[c];
    for (long long un = ((long long)g * kWaves + wave) * q.P + wpart; un < q.units; un += (long long)q.G * kWaves * q.P) {
        long long g0, g1;
        unit_range(q, c, un, g0, g1);
        walk_unit(
            g0, g1, sub, stride,
            [&](long long e) {
                const float4 v = *(const float4*)(a.x + e);
                float4 o;
                o.x = gm * ((v.x - u) * var_inv) + bt;
                o.y = gm * ((v.y - u) * var_inv) + bt;
                o.z = gm * ((v.z - u) * var_inv) + bt;
                o.w = gm * ((v.w - u) * var_inv) + bt;
                if (!RELU || a.y != nullptr) *(float4*)(a.y + e) = o;
                if (RELU) {
                    o.x = bn_relu(o.x); o.y = bn_relu(o.y); o.z = bn_relu(o.z); o.w = bn_relu(o.w);
                    *(float4*)(a.y_relu + e) = o;
                }
            },
            [&](long long e) {
                const float o = gm * ((a.x[e] - u) * var_inv) + bt;
                if (!RELU || a.y != nullptr) a.y[e] = o;
                if (RELU) a.y_relu[e] = bn_relu(o);
            });
    }
}

// (round 6) BatchNorm2D -> ReLU -> MaxPool2D(2, 2) forward in ONE apply pass: a lane owns a pooling window -- two floats of row 2 ph and two
// of row 2 ph + 1 -- normalises them (bn_apply's expression), applies ReLU (relu.cpp:25) and takes the window's maximum in the pool's scan
// order with its strict '<' (pool2d.cpp:60-83: the first maximum wins; mask = flat index into the sample).  The normalised tensor and the
// ReLU output are written only when asked for (a train step reads neither: BatchNorm2D::backward re-computes from x, the pooled-domain
// backward pass below takes ReLU' from the pooled value).  Values are elementwise functions of x and the channel's statistics: bit-identical
// to bn_apply -> maxpool_fwd.  H, W even.
__global__ __launch_bounds__(kBlock) void bn_apply_pool(BnApply a, Geo q, float* __restrict__ pooled, int32_t* __restrict__ mask, int H, int W) {
#pragma clang fp contract(off)
    const int c = blockIdx.y, g = blockIdx.x;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    float u, var;
    bn_channel_stats(a, q, c, g, lane, u, var);
    const float var_inv = 1.f / sqrtf(var + a.eps);
    const float gm = a.gamma[c], bt = a.beta[c];
    const int PH = H / 2, PW = W / 2;
    const long long rows = (long long)q.B * PH;
    for (long long un = (long long)g * kWaves + wave; un < rows; un += (long long)q.G * kWaves) {
        const int b = (int)(un / PH), ph = (int)(un - (long long)b * PH);
        const long long plane = (long long)b * q.C + c;
        const long long r0 = plane * q.HW + (long long)(2 * ph) * W;
        const int mbase = c * q.HW + 2 * ph * W;
        for (int pw = lane; pw < PW; pw += kWave) {
            const float2 t0 = *(const float2*)(a.x + r0 + 2 * pw), t1 = *(const float2*)(a.x + r0 + W + 2 * pw);
            float2 o0, o1;
            o0.x = gm * ((t0.x - u) * var_inv) + bt;
            o0.y = gm * ((t0.y - u) * var_inv) + bt;
            o1.x = gm * ((t1.x - u) * var_inv) + bt;
            o1.y = gm * ((t1.y - u) * var_inv) + bt;
            if (a.y != nullptr) {
                *(float2*)(a.y + r0 + 2 * pw) = o0;
                *(float2*)(a.y + r0 + W + 2 * pw) = o1;
            }
            o0.x = bn_relu(o0.x); o0.y = bn_relu(o0.y); o1.x = bn_relu(o1.x); o1.y = bn_relu(o1.y);
            if (a.y_relu != nullptr) {
                *(float2*)(a.y_relu + r0 + 2 * pw) = o0;
                *(float2*)(a.y_relu + r0 + W + 2 * pw) = o1;
            }
            float best = o0.x;
            int off = 0;
            if (best < o0.y) { best = o0.y; off = 1; }
            if (best < o1.x) { best = o1.x; off = W; }
            if (best < o1.y) { best = o1.y; off = W + 1; }
            const long long at = (plane * PH + ph) * PW + pw;
            pooled[at] = best;
            if (mask != nullptr) mask[at] = mbase + 2 * pw + off;
        }
    }
}

// (round 6) BatchNorm2D <- ReLU <- MaxPool2D(2, 2): the delta at the normalisation's output REBUILT from the pooled domain instead of read --
// MaxPool2D::backward (pool2d.cpp:100-107: a window's delta goes to the element its mask names) and ReLU::backward (relu.cpp:37: 0 where the
// ReLU output -- at a window's maximum that IS the pooled value -- is <= 0), element for element what maxpool_bwd_k2s2+relu (pool.hip)
// writes.  The two backward kernels walk the SAME units / slots / lanes with the same arithmetic as with a materialised delta, so the
// channel sums and dx are bit-identical to the three-kernel sequence; what goes away is the 4 B / element the pool's backward writes and the
// 8 B / element the two passes read back (the stem of the ResNet-shaped stack: 205 MB each).  H even, W % 4 == 0.
struct PoolDelta {
    const float* dpool;
    const int32_t* mask;
    const float* pooled;
    int W, PW, PHPW;   // input row length, pooled row length, pooled plane size
    unsigned wmagic;   // ceil(2^32 / W): hw / W == umulhi(hw, wmagic) for hw < H * W
};
// the four deltas of the aligned slot at element hw (a multiple of 4) of plane `plane` = b * C + c; cbase = c * H * W (the mask holds
// indices into the SAMPLE, pool2d.cpp:81)
__device__ inline float4 pool_delta4(const PoolDelta& s, long long plane, int cbase, int hw) {
    const int h = (int)__umulhi((unsigned)hw, s.wmagic), w = hw - h * s.W;
    const long long wi = plane * s.PHPW + (long long)(h >> 1) * s.PW + (w >> 1);
    const float2 dp = *(const float2*)(s.dpool + wi);
    const int2 mk = *(const int2*)(s.mask + wi);
    const float2 po = *(const float2*)(s.pooled + wi);
    const float d0 = po.x <= 0.f ? 0.f : dp.x, d1 = po.y <= 0.f ? 0.f : dp.y;
    const int f = cbase + hw, m0 = mk.x & 0x7fffffff, m1 = mk.y & 0x7fffffff;
    return make_float4(m0 == f ? d0 : 0.f, m0 == f + 1 ? d0 : 0.f, m1 == f + 2 ? d1 : 0.f, m1 == f + 3 ? d1 : 0.f);
}
__device__ inline float pool_delta1(const PoolDelta& s, long long plane, int cbase, int hw) {
    const int h = (int)__umulhi((unsigned)hw, s.wmagic), w = hw - h * s.W;
    const long long wi = plane * s.PHPW + (long long)(h >> 1) * s.PW + (w >> 1);
    const float d = s.pooled[wi] <= 0.f ? 0.f : s.dpool[wi];
    return (s.mask[wi] & 0x7fffffff) == cbase + hw ? d : 0.f;
}

// partial sums of the backward pass, per channel (batchnorm2d.cpp:118-146):
//   [0] sum dy*norm   [1] sum dy   [2] sum (dy*gamma)*(x-u)*(-0.5)*var_inv^3   [3] sum (x-u)
template <bool POOLED>
__global__ __launch_bounds__(kBlock) void bn_bwd_stats_t(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ saved_mean,
                                                         const float* __restrict__ saved_var, float* __restrict__ part,
                                                         float eps, Geo q, PoolDelta pd) {
    const int c = blockIdx.y, g = blockIdx.x;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int wpart = lane >> q.part_shift, stride = 1 << q.part_shift, sub = lane & (stride - 1);
    const float u = saved_mean[c];
    const float var_inv = 1.f / sqrtf(saved_var[c] + eps);
    const float var_inv_3 = var_inv * var_inv * var_inv;
    const float gm = gamma[c];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    auto one = [&](float xv, float d) {
        const float xc = xv - u;
        acc[0] += d * (xc * var_inv);
        acc[1] += d;
        acc[2] += (d * gm) * xc * -0.5f * var_inv_3;
        acc[3] += xc;
    };
    for (long long un = ((long long)g * kWaves + wave) * q.P + wpart; un < q.units; un += (long long)q.G * kWaves * q.P) {
        long long g0, g1;
        unit_range(q, c, un, g0, g1);
        const long long plane = (q.nseg == 1 ? un : un / q.nseg) * q.C + c, pbase = plane * q.HW;  // (POOLED: the unit's plane)
        walk_unit(
            g0, g1, sub, stride,
            [&](long long e) {
                const float4 v = *(const float4*)(x + e);
                const float4 d = POOLED ? pool_delta4(pd, plane, c * q.HW, (int)(e - pbase)) : *(const float4*)(dy + e);
                one(v.x, d.x);
                one(v.y, d.y);
                one(v.z, d.z);
                one(v.w, d.w);
            },
            [&](long long e) { one(x[e], POOLED ? pool_delta1(pd, plane, c * q.HW, (int)(e - pbase)) : dy[e]); });
    }
    block_store_partials<4>(acc, part, q.G);
}

// dx = (dy*gamma)*var_inv + inv*2*(x-u) + u_g/L, in place on dy (batchnorm2d.cpp:148-155); POOLED: dy rebuilt from the pooled domain
// (pool_delta4), dx written to `dy` (nothing is read there)
template <bool POOLED>
__global__ __launch_bounds__(kBlock) void bn_bwd_apply_t(const float* __restrict__ x, float* __restrict__ dy,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ saved_mean,
                                                         const float* __restrict__ saved_var,
                                                         const float* __restrict__ part, float* __restrict__ ggamma,
                                                         float* __restrict__ gbeta, float eps, Geo q,
                                                         const float* __restrict__ gsums, float count, PoolDelta pd) {
#pragma clang fp contract(off)
    const int c = blockIdx.y, g = blockIdx.x;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int wpart = lane >> q.part_shift, stride = 1 << q.part_shift, sub = lane & (stride - 1);
    // gsums (sync-BN): the four ALL-REDUCED channel sums [C][4] and the global element count replace the local ones
    const float L = gsums ? count : (float)((long long)q.B * q.HW);
    const float u = saved_mean[c];
    const float var_inv = 1.f / sqrtf(saved_var[c] + eps);
    const float gm = gamma[c];
    const float* pc = part + (size_t)c * q.G * 4;
    const float s_gg = gsums ? gsums[c * 4 + 0] : sum_partials(pc + 0, q.G, 4, lane);
    const float s_gb = gsums ? gsums[c * 4 + 1] : sum_partials(pc + 1, q.G, 4, lane);
    const float var_g = gsums ? gsums[c * 4 + 2] : sum_partials(pc + 2, q.G, 4, lane);
    const float s_xc = gsums ? gsums[c * 4 + 3] : sum_partials(pc + 3, q.G, 4, lane);
    const float inv = var_g / L;
    // u_g = sum [ (dy*gamma)*(-var_inv) + inv*(-2)*(x-u) ]  (batchnorm2d.cpp:139-146), from the channel sums
    const float u_g = (s_gb * gm) * (-var_inv) + inv * -2.f * s_xc;
    const float u_term = u_g / L;
    if (g == 0 && threadIdx.x == 0) {
        ggamma[c] = s_gg;
        gbeta[c] = s_gb;
    }
    const float inv2 = inv * 2.f;
    for (long long un = ((long long)g * kWaves + wave) * q.P + wpart; un < q.units; un += (long long)q.G * kWaves * q.P) {
        long long g0, g1;
        unit_range(q, c, un, g0, g1);
        const long long plane = (q.nseg == 1 ? un : un / q.nseg) * q.C + c, pbase = plane * q.HW;  // (POOLED: the unit's plane)
        walk_unit(
            g0, g1, sub, stride,
            [&](long long e) {
                const float4 v = *(const float4*)(x + e);
                float4 d = POOLED ? pool_delta4(pd, plane, c * q.HW, (int)(e - pbase)) : *(float4*)(dy + e);
                d.x = (d.x * gm) * var_inv + inv2 * (v.x - u) + u_term;
                d.y = (d.y * gm) * var_inv + inv2 * (v.y - u) + u_term;
                d.z = (d.z * gm) * var_inv + inv2 * (v.z - u) + u_term;
                d.w = (d.w * gm) * var_inv + inv2 * (v.w - u) + u_term;
                *(float4*)(dy + e) = d;
            },
            [&](long long e) {
                const float d = POOLED ? pool_delta1(pd, plane, c * q.HW, (int)(e - pbase)) : dy[e];
                dy[e] = (d * gm) * var_inv + inv2 * (x[e] - u) + u_term;
            });
    }
}

// out[c*NS + k] = sum over the G partial slots of channel c (same fixed tree as the in-kernel re-reduction)
__global__ __launch_bounds__(kWave) void bn_reduce_partials(const float* __restrict__ part, float* __restrict__ out, int G,
                                                            int NS) {
    const int c = blockIdx.x, lane = threadIdx.x;
    for (int k = 0; k < NS; ++k) {
        const float v = sum_partials(part + (size_t)c * G * NS + k, G, NS, lane);
        if (lane == 0) out[c * NS + k] = v;
    }
}

// ---- one workgroup per channel: the whole channel (B planes of H*W floats) fits LDS ----
// The deep end of a stack (7x7 / 14x14 planes at batch 64: 12 - 50 KB per channel) is launch-latency bound on the general path
// (3 forward / 2 backward launches of a few microseconds each, every one waiting for its predecessor): here a channel is read ONCE
// into LDS, the statistics are block reductions in a fixed order (lane tree, then the 16 waves in order) and the apply pass runs
// from LDS -- one launch per direction, 8 B / element forward and 12 B / element backward.
// threads per channel workgroup: 1024 (round 2) or 256 (round 4, the default).  In a train step these kernels run on the compute stream
// while the side stream's weight-gradient workgroups occupy the CUs: a 1024-thread workgroup has to wait until a CU has 16 free wave slots
// and its registers at once (74 - 76 us in situ for a 6 - 13 MB tensor), a 256-thread one slips in beside them.  The summation order
// depends on the count (lane e mod T, wave tree, waves in order): results of the two differ in the last bits.
constexpr int kChanThreadsMax = 1024;
constexpr int kChanMaxElems = 16384;  // B*H*W floats of one channel: 64 KB forward, 2 x 64 KB backward

template <int NS, int kChanWaves>
__device__ inline void chan_block_sum(float (&v)[NS], float* __restrict__ red) {  // red: [kChanWaves][NS]; result in every thread
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const float w = wave_sum(v[k]);
        if (lane == 0) red[wave * NS + k] = w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        float t = red[k];
        for (int w = 1; w < kChanWaves; ++w) t += red[w * NS + k];
        v[k] = t;
    }
    __syncthreads();
}

__device__ inline size_t chan_addr(const Geo& q, int c, unsigned e) {  // element e of channel c -> float index in the tensor
    const unsigned b = e / (unsigned)q.HW;
    return ((size_t)b * q.C + c) * q.HW + (e - b * (unsigned)q.HW);
}

// training forward (batchnorm2d.cpp:46-80): mean | biased variance around it | y, moving statistics
template <bool RELU, int kChanThreads>
__global__ __launch_bounds__(kChanThreads) void bn_fwd_channel(BnApply a, Geo q) {
#pragma clang fp contract(off)
    constexpr int kChanWaves = kChanThreads / kWave;
    extern __shared__ float chan[];  // [n] x
    __shared__ float red[kChanWaves];
    const int c = blockIdx.x;
    const unsigned n = (unsigned)(q.B * q.HW);
    float acc[1] = {0.f};
    for (unsigned e = threadIdx.x; e < n; e += kChanThreads) {
        const float v = a.x[chan_addr(q, c, e)];
        chan[e] = v;
        acc[0] += v;
    }
    chan_block_sum<1, kChanWaves>(acc, red);  // (its barrier also publishes chan[])
    const float u = acc[0] / (float)n;
    acc[0] = 0.f;
    for (unsigned e = threadIdx.x; e < n; e += kChanThreads) {
        const float d = chan[e] - u;
        acc[0] += d * d;
    }
    chan_block_sum<1, kChanWaves>(acc, red);
    const float var = acc[0] / (float)n;
    if (threadIdx.x == 0) {
        a.saved_mean[c] = u;
        a.saved_var[c] = var;
        a.moving_mean[c] = (1.f - a.momentum) * a.moving_mean[c] + a.momentum * u;
        a.moving_var[c] = (1.f - a.momentum) * a.moving_var[c] + a.momentum * var;
    }
    const float var_inv = 1.f / sqrtf(var + a.eps);
    const float gm = a.gamma[c], bt = a.beta[c];
    for (unsigned e = threadIdx.x; e < n; e += kChanThreads) {
        const size_t at = chan_addr(q, c, e);
        const float o = gm * ((chan[e] - u) * var_inv) + bt;
        if (!RELU || a.y != nullptr) a.y[at] = o;
        if (RELU) a.y_relu[at] = bn_relu(o);
    }
}

// backward (batchnorm2d.cpp:118-155): the four channel sums of bn_bwd_stats, then bn_bwd_apply's dx in place on dy
template <int kChanThreads>
__global__ __launch_bounds__(kChanThreads) void bn_bwd_channel(const float* __restrict__ x, float* __restrict__ dy,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ saved_mean,
                                                               const float* __restrict__ saved_var, float* __restrict__ ggamma,
                                                               float* __restrict__ gbeta, float eps, Geo q) {
#pragma clang fp contract(off)
    constexpr int kChanWaves = kChanThreads / kWave;
    extern __shared__ float chan[];  // [n] x - mean, [n] dy
    __shared__ float red[kChanWaves * 4];
    const int c = blockIdx.x;
    const unsigned n = (unsigned)(q.B * q.HW);
    float* const xc_s = chan;
    float* const d_s = chan + n;
    const float u = saved_mean[c];
    const float var_inv = 1.f / sqrtf(saved_var[c] + eps);
    const float var_inv_3 = var_inv * var_inv * var_inv;
    const float gm = gamma[c];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (unsigned e = threadIdx.x; e < n; e += kChanThreads) {
        const size_t at = chan_addr(q, c, e);
        const float xc = x[at] - u, d = dy[at];
        xc_s[e] = xc;
        d_s[e] = d;
        acc[0] += d * (xc * var_inv);
        acc[1] += d;
        acc[2] += (d * gm) * xc * -0.5f * var_inv_3;
        acc[3] += xc;
    }
    chan_block_sum<4, kChanWaves>(acc, red);
    const float L = (float)n;
    const float inv = acc[2] / L;
    const float u_g = (acc[1] * gm) * (-var_inv) + inv * -2.f * acc[3];
    const float u_term = u_g / L;
    if (threadIdx.x == 0) {
        ggamma[c] = acc[0];
        gbeta[c] = acc[1];
    }
    const float inv2 = inv * 2.f;
    for (unsigned e = threadIdx.x; e < n; e += kChanThreads)
        dy[chan_addr(q, c, e)] = (d_s[e] * gm) * var_inv + inv2 * xc_s[e] + u_term;
}

// BN_CHAN_THREADS=1024: the round-2 workgroup size of the channel kernels (the default; 256 measured: see kChanThreadsMax)
// (round 6) the same kernel with the channel held in REGISTERS instead of LDS: thread t keeps elements t, t + T, t + 2T, ... (at most EPT of
// them) of x - mean and dy, so the element -> thread map, every thread's summation order and the block reduction are those of
// bn_bwd_channel<T> -- bit-identical results -- but the workgroup needs 2 x EPT + ~20 registers per thread and 64 B of LDS instead of
// 2 x B*H*W floats (100 KB for a 14x14 plane at batch 64).  In the train step these kernels run on the compute stream while the side
// stream's weight-gradient workgroups (conv_wgrad_sp.hip: 150 KB of LDS each) sit on the CUs: the LDS version could only start on a CU
// none of them occupied -- 92 - 120 us in the step for a 6 - 13 MB tensor that takes 10 us alone.
template <int kChanThreads, int EPT>
__global__ __launch_bounds__(kChanThreads) void bn_bwd_channel_reg(const float* __restrict__ x, float* __restrict__ dy,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ saved_mean,
                                                                   const float* __restrict__ saved_var, float* __restrict__ ggamma,
                                                                   float* __restrict__ gbeta, float eps, Geo q) {
#pragma clang fp contract(off)
    constexpr int kChanWaves = kChanThreads / kWave;
    __shared__ float red[kChanWaves * 4];
    const int c = blockIdx.x;
    const unsigned n = (unsigned)(q.B * q.HW);
    const float u = saved_mean[c];
    const float var_inv = 1.f / sqrtf(saved_var[c] + eps);
    const float var_inv_3 = var_inv * var_inv * var_inv;
    const float gm = gamma[c];
    float xc_r[EPT], d_r[EPT];
    // every load of the thread is issued before the first one is used
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const unsigned e = threadIdx.x + (unsigned)i * kChanThreads;
        const size_t at = chan_addr(q, c, e < n ? e : 0);
        xc_r[i] = x[at];
        d_r[i] = dy[at];
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const unsigned e = threadIdx.x + (unsigned)i * kChanThreads;
        if (e < n) {
            const float xc = xc_r[i] - u, d = d_r[i];
            xc_r[i] = xc;
            acc[0] += d * (xc * var_inv);
            acc[1] += d;
            acc[2] += (d * gm) * xc * -0.5f * var_inv_3;
            acc[3] += xc;
        }
    }
    chan_block_sum<4, kChanWaves>(acc, red);
    const float L = (float)n;
    const float inv = acc[2] / L;
    const float u_g = (acc[1] * gm) * (-var_inv) + inv * -2.f * acc[3];
    const float u_term = u_g / L;
    if (threadIdx.x == 0) {
        ggamma[c] = acc[0];
        gbeta[c] = acc[1];
    }
    const float inv2 = inv * 2.f;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const unsigned e = threadIdx.x + (unsigned)i * kChanThreads;
        if (e < n) dy[chan_addr(q, c, e)] = (d_r[i] * gm) * var_inv + inv2 * xc_r[i] + u_term;
    }
}

int chan_threads() { return CNN_OPT_INT("BN_CHAN_THREADS", 1024) >= kChanThreadsMax ? 1024 : 256; }

// the channel kernels take a layer when a channel fits LDS and there are enough channels to spread over the chip
bool channel_path(int B, int C, int H, int W) {
    const bool off = CNN_OPT_SET("BN_NO_CHANNEL");  // (A/B switch)
    return !off && (long long)B * H * W <= kChanMaxElems && C >= 32;
}

int make_geo(int B, int C, int H, int W, Geo* q) {
    CNN_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "batchnorm2d: bad geometry B=%d C=%d H=%d W=%d", B, C, H, W);
    CNN_REQUIRE((long long)H * W < (1ll << 30), "batchnorm2d: plane too large");
    q->B = B;
    q->C = C;
    q->HW = H * W;
    q->nseg = (q->HW + kSeg - 1) / kSeg;
    q->units = (long long)B * q->nseg;
    // ~2048 workgroups over the chip (8 per CU), at most one wave per unit, at most kMaxG partial slots
    q->P = 1;
    q->part_shift = 6;
    if (q->nseg == 1)  // a plane spans at most HW/4 + 2 slots
        while (q->P < 8 && q->HW / 4 + 2 <= (kWave >> 1) / q->P) q->P *= 2, --q->part_shift;
    long long G = (2048 + C - 1) / C;
    const long long gmax = (q->units + kWaves * q->P - 1) / (kWaves * q->P);
    if (G > gmax) G = gmax;
    if (G > kMaxG) G = kMaxG;
    if (G < 1) G = 1;
    q->G = (int)G;
    CNN_REQUIRE(C <= 65535, "batchnorm2d: more than 65535 channels");
    return CNN_AMD_OK;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

#define BN_TAG "B%d C%d %dx%d", B, C, H, W

extern "C" {

size_t cnn_batchnorm2d_workspace_bytes(int B, int C, int H, int W) {
    Geo q;
    if (make_geo(B, C, H, W, &q) != CNN_AMD_OK) return 0;
    return (size_t)C * q.G * 4 * sizeof(float) * 2;  // two partial-sum arenas (ping / pong)
}

static int bn_forward_impl(const float* x, float* y, float* y_relu, const float* gamma, const float* beta, float* moving_mean,
                           float* moving_var, float* saved_mean, float* saved_var, int B, int C, int H, int W, float eps,
                           float momentum, int training, void* workspace, size_t workspace_bytes, void* stream, float* pooled = nullptr,
                           int32_t* pool_mask = nullptr) {
    Geo q;
    int rc = make_geo(B, C, H, W, &q);
    if (rc != CNN_AMD_OK) return rc;
    // (round 4) y == NULL with y_relu: only the ReLU output is written (nothing in a train step reads the normalised tensor itself)
    // (round 6) pooled: ReLU and MaxPool2D(2, 2) in the same apply pass; then neither y nor y_relu has to be written
    CNN_REQUIRE(x && (y || y_relu || pooled) && gamma && beta && moving_mean && moving_var, "cnn_batchnorm2d_forward: null pointer");
    CNN_REQUIRE(aligned16(x) && aligned16(y) && aligned16(y_relu), "cnn_batchnorm2d_forward: x / y / y_relu must be 16-byte aligned");
    hipStream_t s = as_stream(stream);
    const dim3 grid(q.G, C);
    BnApply a{x, y, y_relu, gamma, beta, moving_mean, moving_var, saved_mean, saved_var, nullptr, eps, momentum, training ? 1 : 0,
              nullptr, nullptr, 0.f, nullptr, nullptr};
    if (training) {
        CNN_REQUIRE(saved_mean && saved_var, "cnn_batchnorm2d_forward: training needs saved_mean / saved_var");
        CNN_REQUIRE(workspace && workspace_bytes >= cnn_batchnorm2d_workspace_bytes(B, C, H, W),
                    "cnn_batchnorm2d_forward: workspace too small (%zu bytes)", workspace_bytes);
        if (channel_path(B, C, H, W) && pooled == nullptr) {
            const size_t lds = (size_t)B * H * W * sizeof(float);
            static DeviceOnce attr_once;
            if (attr_once.needed()) {
                const int lim = (int)(kChanMaxElems * sizeof(float));
                CNN_HIP_CHECK(hipFuncSetAttribute((const void*)bn_fwd_channel<false, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
                CNN_HIP_CHECK(hipFuncSetAttribute((const void*)bn_fwd_channel<true, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
                CNN_HIP_CHECK(hipFuncSetAttribute((const void*)bn_fwd_channel<false, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
                CNN_HIP_CHECK(hipFuncSetAttribute((const void*)bn_fwd_channel<true, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
                attr_once.mark();
            }
            const bool wide = chan_threads() == 1024;
            if (y_relu) {
                if (wide) CNN_KLAUNCH(s, "bn_fwd_channel+relu", (bn_fwd_channel<true, 1024><<<C, 1024, lds, s>>>(a, q)), BN_TAG);
                else CNN_KLAUNCH(s, "bn_fwd_channel+relu", (bn_fwd_channel<true, 256><<<C, 256, lds, s>>>(a, q)), BN_TAG);
            } else {
                if (wide) CNN_KLAUNCH(s, "bn_fwd_channel", (bn_fwd_channel<false, 1024><<<C, 1024, lds, s>>>(a, q)), BN_TAG);
                else CNN_KLAUNCH(s, "bn_fwd_channel", (bn_fwd_channel<false, 256><<<C, 256, lds, s>>>(a, q)), BN_TAG);
            }
            return CNN_AMD_OK;
        }
        float* p0 = (float*)workspace;
        float* p1 = p0 + (size_t)C * q.G * 4;
        if (CNN_OPT_INT("BN_TWO_PASS", 0) != 0) {
            CNN_KLAUNCH(s, "bn_stats<0>", (bn_stats<0><<<grid, kBlock, 0, s>>>(x, nullptr, p0, nullptr, q, 0.f)), BN_TAG);
            CNN_KLAUNCH(s, "bn_stats<1>", (bn_stats<1><<<grid, kBlock, 0, s>>>(x, p0, p1, saved_mean, q, 0.f)), BN_TAG);
            a.part = p1;
        } else {
            CNN_KLAUNCH(s, "bn_stats_pilot", (bn_stats_pilot<<<grid, kBlock, 0, s>>>(x, p0, p1, q)), BN_TAG);
            a.part2 = p0;
            a.pilot = p1;  // (the second arena is free in this path: C <= C * G * 4)
        }
    }
    if (pooled)
        CNN_KLAUNCH(s, "bn_apply+relu+pool", (bn_apply_pool<<<grid, kBlock, 0, s>>>(a, q, pooled, pool_mask, H, W)), BN_TAG);
    else if (y_relu)
        CNN_KLAUNCH(s, "bn_apply+relu", (bn_apply<true><<<grid, kBlock, 0, s>>>(a, q)), BN_TAG);
    else
        CNN_KLAUNCH(s, "bn_apply", (bn_apply<false><<<grid, kBlock, 0, s>>>(a, q)), BN_TAG);
    return CNN_AMD_OK;
}

int cnn_batchnorm2d_forward_relu_pool_supported(int B, int C, int H, int W) {
    Geo q;
    // (channels that fit one workgroup are normalised by the channel-resident kernel -- statistics and apply in one launch from LDS: the
    //  pool stays its own kernel there)
    return make_geo(B, C, H, W, &q) == CNN_AMD_OK && !channel_path(B, C, H, W) && H >= 2 && H % 2 == 0 && W >= 2 && W % 2 == 0 &&
                   (long long)C * H * W < (1ll << 31)
               ? 1 : 0;
}

int cnn_batchnorm2d_forward_relu_pool(const float* x, float* y, float* y_relu, float* pooled, int32_t* pool_mask, const float* gamma,
                                      const float* beta, float* moving_mean, float* moving_var, float* saved_mean, float* saved_var, int B,
                                      int C, int H, int W, float eps, float momentum, int training, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    CNN_REQUIRE(pooled != nullptr, "cnn_batchnorm2d_forward_relu_pool: null pooled");
    CNN_REQUIRE(cnn_batchnorm2d_forward_relu_pool_supported(B, C, H, W), "cnn_batchnorm2d_forward_relu_pool: not supported for %dx%dx%dx%d", B, C, H, W);
    CNN_REQUIRE(((uintptr_t)x & 7) == 0 && ((uintptr_t)y & 7) == 0 && ((uintptr_t)y_relu & 7) == 0, "cnn_batchnorm2d_forward_relu_pool: unaligned tensor");
    return bn_forward_impl(x, y, y_relu, gamma, beta, moving_mean, moving_var, saved_mean, saved_var, B, C, H, W, eps, momentum, training, workspace,
                           workspace_bytes, stream, pooled, pool_mask);
}

int cnn_batchnorm2d_forward(const float* x, float* y, const float* gamma, const float* beta, float* moving_mean,
                            float* moving_var, float* saved_mean, float* saved_var, int B, int C, int H, int W, float eps,
                            float momentum, int training, void* workspace, size_t workspace_bytes, void* stream) {
    return bn_forward_impl(x, y, nullptr, gamma, beta, moving_mean, moving_var, saved_mean, saved_var, B, C, H, W, eps, momentum,
                           training, workspace, workspace_bytes, stream);
}

int cnn_batchnorm2d_forward_relu(const float* x, float* y, float* y_relu, const float* gamma, const float* beta,
                                 float* moving_mean, float* moving_var, float* saved_mean, float* saved_var, int B, int C, int H,
                                 int W, float eps, float momentum, int training, void* workspace, size_t workspace_bytes,
                                 void* stream) {
    CNN_REQUIRE(y_relu != nullptr, "cnn_batchnorm2d_forward_relu: null y_relu");
    return bn_forward_impl(x, y, y_relu, gamma, beta, moving_mean, moving_var, saved_mean, saved_var, B, C, H, W, eps, momentum,
                           training, workspace, workspace_bytes, stream);
}

int cnn_batchnorm2d_backward(const float* x, float* dy, const float* gamma, const float* saved_mean,
                             const float* saved_var, float* ggamma, float* gbeta, int B, int C, int H, int W, float eps,
                             void* workspace, size_t workspace_bytes, void* stream) {
    Geo q;
    int rc = make_geo(B, C, H, W, &q);
    if (rc != CNN_AMD_OK) return rc;
    CNN_REQUIRE(x && dy && gamma && saved_mean && saved_var && ggamma && gbeta, "cnn_batchnorm2d_backward: null pointer");
    CNN_REQUIRE(aligned16(x) && aligned16(dy), "cnn_batchnorm2d_backward: x / dy must be 16-byte aligned");
    CNN_REQUIRE(workspace && workspace_bytes >= cnn_batchnorm2d_workspace_bytes(B, C, H, W),
                "cnn_batchnorm2d_backward: workspace too small (%zu bytes)", workspace_bytes);
    hipStream_t s = as_stream(stream);
    const dim3 grid(q.G, C);
    if (channel_path(B, C, H, W)) {
        const size_t lds = (size_t)B * H * W * sizeof(float) * 2;
        static DeviceOnce attr_once;
        if (attr_once.needed()) {
            const int lim = (int)(kChanMaxElems * sizeof(float) * 2);
            CNN_HIP_CHECK(hipFuncSetAttribute((const void*)bn_bwd_channel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            CNN_HIP_CHECK(hipFuncSetAttribute((const void*)bn_bwd_channel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
            attr_once.mark();
        }
        // (round 6) register-resident channels: same results bit for bit, no LDS to wait for beside the weight gradients
        if (chan_threads() == 1024 && CNN_OPT_INT("BN_BWD_LDS", 0) == 0) {
            const long long nel = (long long)B * H * W;
            if (nel <= 1024 * 4)
                CNN_KLAUNCH(s, "bn_bwd_channel_reg", (bn_bwd_channel_reg<1024, 4><<<C, 1024, 0, s>>>(x, dy, gamma, saved_mean, saved_var, ggamma, gbeta, eps, q)), BN_TAG);
            else if (nel <= 1024 * 8)
                CNN_KLAUNCH(s, "bn_bwd_channel_reg", (bn_bwd_channel_reg<1024, 8><<<C, 1024, 0, s>>>(x, dy, gamma, saved_mean, saved_var, ggamma, gbeta, eps, q)), BN_TAG);
            else
                CNN_KLAUNCH(s, "bn_bwd_channel_reg", (bn_bwd_channel_reg<1024, 16><<<C, 1024, 0, s>>>(x, dy, gamma, saved_mean, saved_var, ggamma, gbeta, eps, q)), BN_TAG);
            return CNN_AMD_OK;
        }
        if (chan_threads() == 1024)
            CNN_KLAUNCH(s, "bn_bwd_channel",
                        (bn_bwd_channel<1024><<<C, 1024, lds, s>>>(x, dy, gamma, saved_mean, saved_var, ggamma, gbeta, eps, q)), BN_TAG);
        else
            CNN_KLAUNCH(s, "bn_bwd_channel",
                        (bn_bwd_channel<256><<<C, 256, lds, s>>>(x, dy, gamma, saved_mean, saved_var, ggamma, gbeta, eps, q)), BN_TAG);
        return CNN_AMD_OK;
    }
    float* part = (float*)workspace;
    CNN_KLAUNCH(s, "bn_bwd_stats",
                (bn_bwd_stats_t<false><<<grid, kBlock, 0, s>>>(x, dy, gamma, saved_mean, saved_var, part, eps, q, PoolDelta{})), BN_TAG);
    CNN_KLAUNCH(s, "bn_bwd_apply",
                (bn_bwd_apply_t<false><<<grid, kBlock, 0, s>>>(x, dy, gamma, saved_mean, saved_var, part, ggamma, gbeta, eps, q, nullptr, 0.f, PoolDelta{})),
                BN_TAG);
    return CNN_AMD_OK;
}

int cnn_batchnorm2d_backward_pooled_supported(int B, int C, int H, int W) {
    Geo q;
    // (channels that fit one workgroup take the channel-resident kernel in cnn_batchnorm2d_backward: another summation order -- the pooled
    //  form would not be bit-identical to the sequence it replaces there, and has nothing to save: 19 against 18 B / element)
    return make_geo(B, C, H, W, &q) == CNN_AMD_OK && !channel_path(B, C, H, W) && H >= 2 && H % 2 == 0 && W % 4 == 0 &&
                   (long long)C * H * W < (1ll << 31)
               ? 1 : 0;
}

int cnn_batchnorm2d_backward_pooled(const float* x, const float* dpool, const int32_t* mask, const float* pooled, float* dx,
                                    const float* gamma, const float* saved_mean, const float* saved_var, float* ggamma, float* gbeta,
                                    int B, int C, int H, int W, float eps, void* workspace, size_t workspace_bytes, void* stream) {
    Geo q;
    int rc = make_geo(B, C, H, W, &q);
    if (rc != CNN_AMD_OK) return rc;
    CNN_REQUIRE(x && dpool && mask && pooled && dx && gamma && saved_mean && saved_var && ggamma && gbeta, "cnn_batchnorm2d_backward_pooled: null pointer");
    CNN_REQUIRE(cnn_batchnorm2d_backward_pooled_supported(B, C, H, W), "cnn_batchnorm2d_backward_pooled: H must be even and W a multiple of 4 (%dx%d)", H, W);
    CNN_REQUIRE(aligned16(x) && aligned16(dx) && ((uintptr_t)dpool & 7) == 0 && ((uintptr_t)mask & 7) == 0 && ((uintptr_t)pooled & 7) == 0,
                "cnn_batchnorm2d_backward_pooled: x / dx must be 16-byte, the pooled-domain tensors 8-byte aligned");
    CNN_REQUIRE(workspace && workspace_bytes >= cnn_batchnorm2d_workspace_bytes(B, C, H, W),
                "cnn_batchnorm2d_backward_pooled: workspace too small (%zu bytes)", workspace_bytes);
    hipStream_t s = as_stream(stream);
    const dim3 grid(q.G, C);
    const PoolDelta pd{dpool, mask, pooled, W, W / 2, (H / 2) * (W / 2), (unsigned)(((1ull << 32) + (unsigned)W - 1) / (unsigned)W)};
    float* part = (float*)workspace;
    CNN_KLAUNCH(s, "bn_bwd_stats+pool", (bn_bwd_stats_t<true><<<grid, kBlock, 0, s>>>(x, nullptr, gamma, saved_mean, saved_var, part, eps, q, pd)), BN_TAG);
    CNN_KLAUNCH(s, "bn_bwd_apply+pool",
                (bn_bwd_apply_t<true><<<grid, kBlock, 0, s>>>(x, dx, gamma, saved_mean, saved_var, part, ggamma, gbeta, eps, q, nullptr, 0.f, pd)), BN_TAG);
    return CNN_AMD_OK;
}

/* ---- the same arithmetic with the batch sharded over data-parallel ranks (sync-BN): see include/cnn_amd.h ---- */
int cnn_batchnorm2d_partial_sums(const float* x, const float* sum_x, float count, float* out, int B, int C, int H, int W,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    Geo q;
    int rc = make_geo(B, C, H, W, &q);
    if (rc != CNN_AMD_OK) return rc;
    CNN_REQUIRE(x && out && aligned16(x), "cnn_batchnorm2d_partial_sums: null / unaligned pointer");
    CNN_REQUIRE(sum_x == nullptr || count > 0.f, "cnn_batchnorm2d_partial_sums: count must be positive");
    CNN_REQUIRE(workspace && workspace_bytes >= cnn_batchnorm2d_workspace_bytes(B, C, H, W),
                "cnn_batchnorm2d_partial_sums: workspace too small (%zu bytes)", workspace_bytes);
    hipStream_t s = as_stream(stream);
    const dim3 grid(q.G, C);
    float* part = (float*)workspace;
    if (sum_x == nullptr)
        CNN_KLAUNCH(s, "bn_stats<0>", (bn_stats<0><<<grid, kBlock, 0, s>>>(x, nullptr, part, nullptr, q, 0.f)), BN_TAG);
    else
        CNN_KLAUNCH(s, "bn_stats<2>", (bn_stats<2><<<grid, kBlock, 0, s>>>(x, sum_x, part, nullptr, q, count)), BN_TAG);
    CNN_KLAUNCH(s, "bn_reduce_partials", (bn_reduce_partials<<<C, kWave, 0, s>>>(part, out, q.G, 1)), BN_TAG);
    return CNN_AMD_OK;
}

static int bn_forward_from_sums_impl(const float* x, float* y, float* y_relu, const float* gamma, const float* beta,
                                     float* moving_mean, float* moving_var, float* saved_mean, float* saved_var,
                                     const float* sum_x, const float* sum_sq, float count, int B, int C, int H, int W, float eps,
                                     float momentum, void* stream) {
    Geo q;
    int rc = make_geo(B, C, H, W, &q);
    if (rc != CNN_AMD_OK) return rc;
    CNN_REQUIRE(x && (y || y_relu) && gamma && beta && moving_mean && moving_var && saved_mean && saved_var && sum_x && sum_sq,
                "cnn_batchnorm2d_forward_from_sums: null pointer");
    CNN_REQUIRE(aligned16(x) && aligned16(y) && aligned16(y_relu) && count > 0.f,
                "cnn_batchnorm2d_forward_from_sums: unaligned x / y / y_relu or count <= 0");
    hipStream_t s = as_stream(stream);
    BnApply a{x, y, y_relu, gamma, beta, moving_mean, moving_var, saved_mean, saved_var, nullptr, eps, momentum, 1, sum_x, sum_sq, count};
    if (y_relu)
        CNN_KLAUNCH(s, "bn_apply+relu/sync", (bn_apply<true><<<dim3(q.G, C), kBlock, 0, s>>>(a, q)), BN_TAG);
    else
        CNN_KLAUNCH(s, "bn_apply/sync", (bn_apply<false><<<dim3(q.G, C), kBlock, 0, s>>>(a, q)), BN_TAG);
    return CNN_AMD_OK;
}

int cnn_batchnorm2d_forward_from_sums(const float* x, float* y, const float* gamma, const float* beta, float* moving_mean,
                                      float* moving_var, float* saved_mean, float* saved_var, const float* sum_x,
                                      const float* sum_sq, float count, int B, int C, int H, int W, float eps, float momentum,
                                      void* stream) {
    return bn_forward_from_sums_impl(x, y, nullptr, gamma, beta, moving_mean, moving_var, saved_mean, saved_var, sum_x, sum_sq, count,
                                     B, C, H, W, eps, momentum, stream);
}

int cnn_batchnorm2d_forward_from_sums_relu(const float* x, float* y, float* y_relu, const float* gamma, const float* beta,
                                           float* moving_mean, float* moving_var, float* saved_mean, float* saved_var,
                                           const float* sum_x, const float* sum_sq, float count, int B, int C, int H, int W,
                                           float eps, float momentum, void* stream) {
    CNN_REQUIRE(y_relu != nullptr, "cnn_batchnorm2d_forward_from_sums_relu: null y_relu");
    return bn_forward_from_sums_impl(x, y, y_relu, gamma, beta, moving_mean, moving_var, saved_mean, saved_var, sum_x, sum_sq, count,
                                     B, C, H, W, eps, momentum, stream);
}

int cnn_batchnorm2d_backward_sums(const float* x, const float* dy, const float* gamma, const float* saved_mean,
                                  const float* saved_var, float* sums4, int B, int C, int H, int W, float eps, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    Geo q;
    int rc = make_geo(B, C, H, W, &q);
    if (rc != CNN_AMD_OK) return rc;
    CNN_REQUIRE(x && dy && gamma && saved_mean && saved_var && sums4, "cnn_batchnorm2d_backward_sums: null pointer");
    CNN_REQUIRE(aligned16(x) && aligned16(dy), "cnn_batchnorm2d_backward_sums: x / dy must be 16-byte aligned");
    CNN_REQUIRE(workspace && workspace_bytes >= cnn_batchnorm2d_workspace_bytes(B, C, H, W),
                "cnn_batchnorm2d_backward_sums: workspace too small (%zu bytes)", workspace_bytes);
    hipStream_t s = as_stream(stream);
    float* part = (float*)workspace;
    CNN_KLAUNCH(s, "bn_bwd_stats",
                (bn_bwd_stats_t<false><<<dim3(q.G, C), kBlock, 0, s>>>(x, dy, gamma, saved_mean, saved_var, part, eps, q, PoolDelta{})), BN_TAG);
    CNN_KLAUNCH(s, "bn_reduce_partials", (bn_reduce_partials<<<C, kWave, 0, s>>>(part, sums4, q.G, 4)), BN_TAG);
    return CNN_AMD_OK;
}

int cnn_batchnorm2d_backward_from_sums(const float* x, float* dy, const float* gamma, const float* saved_mean,
                                       const float* saved_var, const float* sums4, float count, float* ggamma, float* gbeta,
                                       int B, int C, int H, int W, float eps, void* stream) {
    Geo q;
    int rc = make_geo(B, C, H, W, &q);
    if (rc != CNN_AMD_OK) return rc;
    CNN_REQUIRE(x && dy && gamma && saved_mean && saved_var && sums4 && ggamma && gbeta,
                "cnn_batchnorm2d_backward_from_sums: null pointer");
    CNN_REQUIRE(aligned16(x) && aligned16(dy) && count > 0.f, "cnn_batchnorm2d_backward_from_sums: unaligned x / dy or count <= 0");
    hipStream_t s = as_stream(stream);
    CNN_KLAUNCH(s, "bn_bwd_apply/sync",
                (bn_bwd_apply_t<false><<<dim3(q.G, C), kBlock, 0, s>>>(x, dy, gamma, saved_mean, saved_var, sums4, ggamma, gbeta, eps, q,
                                                                       sums4, count, PoolDelta{})),
                BN_TAG);
    return CNN_AMD_OK;
}

}  // extern "C"
