// micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 with the dependency pattern of conv_wgrad_win (4 accumulators, reuse distance 2)
// build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int DIST>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, float b) {
    f4 acc[4] = {};
    float av = a + threadIdx.x, bv = b + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j % DIST] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j % DIST], 0, 0, 0);
    }
    f4 s = acc[0] + acc[1] + acc[2] + acc[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 4; waves <= 8; waves += 4)
        for (int dist = 1; dist <= 4; dist *= 2) {
            const int iters = 28 * 8;  // x32 MFMAs
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (dist == 1) k<1><<<256, waves * 64>>>(out, iters, 1.f, 2.f);
                if (dist == 2) k<2><<<256, waves * 64>>>(out, iters, 1.f, 2.f);
                if (dist == 4) k<4><<<256, waves * 64>>>(out, iters, 1.f, 2.f);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double mf = 256.0 * waves * iters * 32;
                if (rep == 2) printf("waves/WG %d dist %d: %.1f us, %.1f cycles@2.4GHz per MFMA per SIMD, %.1f TF\n", waves, dist, ms * 1e3,
                                     ms * 1e-3 * 2.4e9 / (iters * 32.0 * waves / 4), mf * 2048 / (ms * 1e-3) / 1e12);
            }
        }
    return 0;
}
