// conv_wgrad_os.hip -- output-stationary Conv2D weight / bias gradient (cpu/src/conv2d.cpp:117-159) for the SMALL 3x3 / stride-2 /
// pad-0 layers of the reference net (conv_layer_2..4: 16->32 @55, 32->64 @27, 64->128 @13; alexnet.cpp:14-22).
//
// Why another kernel: the register-direct kernel (conv_wgrad_rd.hip) streams both operands from L1 into MFMA registers with one
// 16-byte window per lane -- 64 different cache lines per load instruction.  On the wide layers the 32x32 tiles amortise that; on
// these three the texture path's line rate is the bound (41 / 32 / 36 us against MFMA floors of 13 / 12 / 10 us), and because the
// reduction dimension (pixels) is split over 512 workgroups, 28 MB of partial slabs go through slab_reduce.  Here the operands are
// staged through LDS with fully coalesced loads and the accumulators stay put:
//
//   GEMM  M = filter taps (ci, ky, kx) of a 16-channel block = 144 -> 9 tiles, N = 32 output channels -> 2 tiles, K = pixels, on
//   v_mfma_f32_16x16x4_f32; a workgroup of 6 waves (3 x 2, three tap tiles x one channel tile per wave) owns that 144 x 32 block of
//   the gradient (blockIdx.y / .z pick the channel blocks) and walks UNITS of R output rows of one image:
//   * x: the 2R+1 input rows of the unit are ONE contiguous run per channel -> LDS as they lie (Xs[ci][(2R+1)*W]); dy: the R output
//     rows are one contiguous run per output channel -> Ds[co][DP] (DP = 4 mod 32: conflict-free k-group reads), zero tail.
//   * k-slot kq of step j is pixel f = 4j + kq of the unit (flattened over the R rows); a lane's A operand is
//     Xs[tap offset + 2*(f / Wo)*W + 2*(f % Wo)] (the per-lane pixel offsets of all steps are computed once), its B operand
//     Ds[co][f].  Pixels behind the unit's last valid one are zeroed on BOTH operands.
//   * the next unit's operands are fetched into registers before the MFMA loop of the current one and written to LDS behind it.
//   The bias gradient (sum of dy) is accumulated on the VALU from the B registers by the waves of tap block 0.
// Output: slabs[blockIdx.x][Co][Ci*9 + 1] like the register-direct kernel (reduce_slabs adds them in a fixed order).
#include <cstdlib>

#include "common.h"

using namespace cnn_amd;

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct OsParams {
    const float* x;
    const float* dy;
    float* slabs;  // [gridDim.x][CO][CI*9 + 1]
    int B, units;  // units = B * ceil(Ho / R)
    int dbg;       // CNN_AMD_OS_DBG (tuning): 1 = operands fetched for the first unit only, 2 = no MFMA loop
};

constexpr int kCib = 16, kCob = 32;          // channel block of a workgroup: 144 taps x 32 output channels = 9 x 2 tiles
constexpr int kWm = 3, kWn = 2, kTm = 3;     // waves (tap groups x channel tiles), tap tiles per wave
constexpr int kThreads = 64 * kWm * kWn;

template <int CI, int CO, int H, int W, int R>
struct OsGeo {
    static constexpr int Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1;
    static constexpr int XR = 2 * R + 1, XN = XR * W;   // input rows of a unit, floats per channel
    static constexpr int PX = R * Wo, STEPS = (PX + 3) / 4;
    static constexpr int DP = ((4 * STEPS - 4 + 31) / 32) * 32 + 4;
    static constexpr int UPI = (Ho + R - 1) / R;
    static constexpr int XE = kCib * XN, DE = kCob * PX;
    static constexpr int TPX = kThreads / kCib, TPD = kThreads / kCob;  // staging threads per input / output channel
    static constexpr int NXR = (XN + TPX - 1) / TPX, NDR = (PX + TPD - 1) / TPD;
    static constexpr int PITCH = CI * 9 + 1;
    static constexpr size_t lds_bytes = (size_t)(XE + kCob * DP) * sizeof(float);
    static_assert(CI % kCib == 0 && CO % kCob == 0, "channel blocks");
    static_assert(DP >= 4 * STEPS && DP % 32 == 4, "dy pitch");
    static_assert(2 * (R - 1) + 2 < XR && 2 * (Wo - 1) + 2 < W, "window inside the staged rows");
};

template <int CI, int CO, int H, int W, int R>
__global__ __launch_bounds__(kThreads, 3) void conv_wgrad_os_kernel(const OsParams p) {
    using G = OsGeo<CI, CO, H, W, R>;
    extern __shared__ float lds[];
    float* const Xs = lds;            // [kCib][XN]
    float* const Ds = lds + G::XE;    // [kCob][DP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kq = lane >> 4;
    const int wm = wave % kWm, wn = wave / kWm;
    const int ci0 = blockIdx.y * kCib, co0 = blockIdx.z * kCob;

    // zero tails of the dy rows (never overwritten)
    for (int e = tid; e < kCob * (G::DP - G::PX); e += kThreads) {
        const int co = e / (G::DP - G::PX), off = e - co * (G::DP - G::PX);
        Ds[co * G::DP + G::PX + off] = 0.f;
    }

    // A: offset of this lane's tap inside the staged channel block, for each of its tap tiles
    int a_off[kTm];
#pragma unroll
    for (int i = 0; i < kTm; ++i) {
        const int tap = 16 * (wm * kTm + i) + m;  // (ci_local, ky, kx) = (tap / 9, tap % 9 / 3, tap % 3)
        a_off[i] = (tap / 9) * G::XN + ((tap % 9) / 3) * W + tap % 3;
    }
    const int b_off = (16 * wn + m) * G::DP + kq;
    // pixel f = 4j + kq of the unit -> offset of its window's first element (clamped behind the last pixel: masked below)
    int xo[G::STEPS];
#pragma unroll
    for (int j = 0; j < G::STEPS; ++j) {
        int f = 4 * j + kq;
        f = f < G::PX ? f : G::PX - 1;
        xo[j] = 2 * (f / G::Wo) * W + 2 * (f % G::Wo);
    }

    f32x4 acc[kTm];
#pragma unroll
    for (int i = 0; i < kTm; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    // staging: kThreads / 16 = 24 threads per input channel, kThreads / 32 = 12 per output channel; a thread's elements are TPX / TPD
    // floats apart, so its global and LDS addresses are one base + compile-time offsets
    float xr[G::NXR], dr[G::NDR];
    const int sci = tid / G::TPX, sxo = tid - sci * G::TPX;
    const int sco = tid / G::TPD, sdo = tid - sco * G::TPD;
    float* const xs_w = Xs + sci * G::XN + sxo;
    float* const ds_w = Ds + sco * G::DP + sdo;
    auto fetch = [&](int u) {
        const int b = u / G::UPI, r0 = (u - b * G::UPI) * R;
        const float* xb = p.x + ((size_t)b * CI + ci0 + sci) * (H * W) + 2 * r0 * W + sxo;
        const float* db = p.dy + ((size_t)b * CO + co0 + sco) * (G::Ho * G::Wo) + r0 * G::Wo + sdo;
        // (rows under the image / behind the last output row -- a unit that ends behind the image -- are staged as zeros)
        const int xlim = ((H - 2 * r0) * W < G::XN ? (H - 2 * r0) * W : G::XN) - sxo;
        const int dlim = ((G::Ho - r0) * G::Wo < G::PX ? (G::Ho - r0) * G::Wo : G::PX) - sdo;
#pragma unroll
        for (int i = 0; i < G::NXR; ++i) xr[i] = i * G::TPX < xlim ? xb[i * G::TPX] : 0.f;
#pragma unroll
        for (int i = 0; i < G::NDR; ++i) dr[i] = i * G::TPD < dlim ? db[i * G::TPD] : 0.f;
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < G::NXR; ++i)
            if (sxo + i * G::TPX < G::XN) xs_w[i * G::TPX] = xr[i];
#pragma unroll
        for (int i = 0; i < G::NDR; ++i)
            if (sdo + i * G::TPD < G::PX) ds_w[i * G::TPD] = dr[i];
    };

    int u = blockIdx.x;
    if (u < p.units) fetch(u);
    for (; u < p.units; u += gridDim.x) {
        __syncthreads();  // the previous unit's operands have been consumed
        commit();
        __syncthreads();
        const int un = u + gridDim.x;
        if (un < p.units && p.dbg != 1) fetch(un);  // in flight under the MFMA loop
        if (p.dbg == 2) continue;
        const int r0 = (u % G::UPI) * R;
        const int rows = G::Ho - r0 < R ? G::Ho - r0 : R;
        const int pxv = rows * G::Wo;  // valid pixels of this unit
#pragma unroll
        for (int j = 0; j < G::STEPS; ++j) {
            const float bv = Ds[b_off + 4 * j];  // (0 behind the last valid pixel: staged as zeros)
            const bool live = 4 * j + kq < pxv;
            float av[kTm];
#pragma unroll
            for (int i = 0; i < kTm; ++i) {
                const float v = Xs[a_off[i] + xo[j]];
                av[i] = live ? v : 0.f;
            }
#pragma unroll
            for (int i = 0; i < kTm; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv, acc[i], 0, 0, 0);
            bsum += bv;
        }
    }

    // D[row = tap 4 kq + r][column = co m] -> slab[co][tap]
    float* slab = p.slabs + (size_t)blockIdx.x * CO * G::PITCH;
    const int co = co0 + 16 * wn + m;
#pragma unroll
    for (int i = 0; i < kTm; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tap = ci0 * 9 + 16 * (wm * kTm + i) + 4 * kq + r;
            slab[(size_t)co * G::PITCH + tap] = acc[i][r];
        }
    if (blockIdx.y == 0 && wm == 0) {  // bias gradient: the four k-groups of a channel, in a fixed order
        float v = bsum;
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (kq == 0) slab[(size_t)co * G::PITCH + CI * 9] = v;
    }
}

struct OsShape {
    int Ci, Co, H, W, slots;
};
// slabs per layer: (Ci/16) * (Co/32) workgroups share one slab; 768 / 512 / 512 workgroups in all
constexpr OsShape kShapes[] = {{16, 32, 55, 55, 512}, {32, 64, 27, 27, 128}, {64, 128, 13, 13, 32}};

const OsShape* find_shape(const cnn_conv2d_desc* d) {
    const bool off = (CNN_OPT_SET("WGRAD_OS") && CNN_OPT_INT("WGRAD_OS", 0) == 0);
    if (off || d->k != 3 || d->s != 2 || d->pad != 0 || d->B < 1) return nullptr;
    const int mask = CNN_OPT_INT("WGRAD_OS_MASK", 5);  // bit l = conv_layer_{l+2}; conv_layer_3 (33.7 vs 32.7 us alone) stays on the register-direct kernel: in the step 5 measured >= 7 > 3 ~ 0
    int bit = 1;
    for (const OsShape& s : kShapes) {
        if ((mask & bit) && d->Ci == s.Ci && d->Co == s.Co && d->H == s.H && d->W == s.W) return &s;
        bit <<= 1;
    }
    return nullptr;
}

template <int CI, int CO, int H, int W, int R>
int launch_os(const cnn_conv2d_desc* d, const float* x, const float* dy, float* slabs, int slots, hipStream_t s) {
    using G = OsGeo<CI, CO, H, W, R>;
    OsParams p;
    p.x = x; p.dy = dy; p.slabs = slabs; p.B = d->B;
    p.units = d->B * G::UPI;
    p.dbg = CNN_MEASURE_INT("OS_DBG", 0);
    static DeviceOnce attr_once;
    if (attr_once.needed()) {
        CNN_HIP_CHECK(hipFuncSetAttribute((const void*)conv_wgrad_os_kernel<CI, CO, H, W, R>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)G::lds_bytes));
        attr_once.mark();
    }
    const dim3 grid((unsigned)slots, CI / kCib, CO / kCob);
    char name[48];
    snprintf(name, sizeof(name), "conv_wgrad_os<%d,%d>", CI, CO);
    CNN_KLAUNCH(s, name, (conv_wgrad_os_kernel<CI, CO, H, W, R><<<grid, kThreads, G::lds_bytes, s>>>(p)),
                "B%d Ci%d %dx%d Co%d k3 s2 p0 slabs%d", d->B, CI, H, W, CO, slots);
    return CNN_AMD_OK;
}

}  // namespace

namespace cnn_amd {

// number of slabs os_wgrad_launch() writes (0: geometry not covered); every slab is fully written
int os_wgrad_slots(const cnn_conv2d_desc* d) {
    const OsShape* sh = find_shape(d);
    if (!sh) return 0;
    int slots = sh->slots;
    if (const OptVal e = CNN_OPT_VAL("OS_SLABS")) {
        const int v = atoi(e);
        if (v > 0) slots = v;
    }
    // (tuning, per layer: OS_SLABS2 / OS_SLABS3 / OS_SLABS4 = slabs of conv_layer_2 / _3 / _4)
    const OptVal per = sh->H == 55 ? CNN_OPT_VAL("OS_SLABS2") : (sh->H == 27 ? CNN_OPT_VAL("OS_SLABS3") : CNN_OPT_VAL("OS_SLABS4"));
    if (per && atoi(per) > 0) slots = atoi(per);
    const int upi = sh->H == 55 ? 9 : (sh->H == 27 ? 2 : 1);
    const long long units = (long long)d->B * upi;
    if (units >= (1ll << 30)) return 0;
    return slots < units ? slots : (int)units;
}

int os_wgrad_launch(const cnn_conv2d_desc* d, const float* x, const float* dy, float* slabs, hipStream_t s) {
    const int slots = os_wgrad_slots(d);
    CNN_REQUIRE(slots > 0, "os_wgrad_launch: geometry not covered");
    if (d->Ci == 16) return launch_os<16, 32, 55, 55, 3>(d, x, dy, slabs, slots, s);
    if (d->Ci == 32) return launch_os<32, 64, 27, 27, 7>(d, x, dy, slabs, slots, s);
    return launch_os<64, 128, 13, 13, 6>(d, x, dy, slabs, slots, s);
}

}  // namespace cnn_amd
