// Probe: what the fp32 MFMA pipe sustains with no memory traffic at all (the practical ceiling the conv kernels are
// measured against), and the shader clock it runs at while doing so.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak_probe.cpp -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS, int VALU>
__global__ __launch_bounds__(256) void k32(float* out, int iters, long long* clk) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f + 1.f;
    const int sel = out[0] != 0.f;  // unknown to the compiler
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                float bv = b;
                if (VALU >= 1) bv = (u < sel + 20) ? bv : 0.f;
                if (VALU >= 2) bv = sel ? 1.f : bv;
                if (VALU >= 3) bv = (threadIdx.x & sel) ? a : bv;
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[c], 0, 0, 0);
            }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.f) out[1] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int CHAINS>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 4; ++r) acc[c][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f + 1.f;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 4; ++r) s += acc[c][r];
    if (s == 12345.f) out[1] = s;
}

template <typename F>
static float time_ms(F f) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    f();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out; long long* clk; long long h[2];
    (void)hipMalloc(&out, 64); (void)hipMemset(out, 0, 64);
    (void)hipMalloc(&clk, 16);
    const int iters = 20000;
    for (int blocks : {256, 512, 1024}) {
        const double fl32 = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 2;
        float ms;
        ms = time_ms([&] { k32<1, 0><<<blocks, 256>>>(out, iters, clk); });
        (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        printf("blocks %4d  32x32x2 1 chain          : %8.3f ms  %7.2f TFLOP/s   clock64 %lld ticks, wall_clock64 %lld ticks (100 MHz) -> %.0f MHz\n", blocks, ms,
               fl32 / ms * 1e-9, h[0], h[1], (double)h[0] / ((double)h[1] / 100.0));
        ms = time_ms([&] { k32<4, 0><<<blocks, 256>>>(out, iters / 4, clk); });
        printf("blocks %4d  32x32x2 4 chains         : %8.3f ms  %7.2f TFLOP/s\n", blocks, ms, fl32 / ms * 1e-9);
        ms = time_ms([&] { k32<1, 1><<<blocks, 256>>>(out, iters, clk); });
        printf("blocks %4d  32x32x2 1 chain + 2 VALU : %8.3f ms  %7.2f TFLOP/s\n", blocks, ms, fl32 / ms * 1e-9);
        ms = time_ms([&] { k32<1, 3><<<blocks, 256>>>(out, iters, clk); });
        printf("blocks %4d  32x32x2 1 chain + 5 VALU : %8.3f ms  %7.2f TFLOP/s\n", blocks, ms, fl32 / ms * 1e-9);
        const double fl16 = (double)blocks * 4 * iters * 16 * 2.0 * 16 * 16 * 4;
        ms = time_ms([&] { k16<1><<<blocks, 256>>>(out, iters); });
        printf("blocks %4d  16x16x4 1 chain          : %8.3f ms  %7.2f TFLOP/s\n", blocks, ms, fl16 / ms * 1e-9);
        ms = time_ms([&] { k16<4><<<blocks, 256>>>(out, iters / 4); });
        printf("blocks %4d  16x16x4 4 chains         : %8.3f ms  %7.2f TFLOP/s\n", blocks, ms, fl16 / ms * 1e-9);
    }
    return 0;
}
