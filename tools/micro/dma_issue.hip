// dma_issue.hip -- how long does a wave take to ISSUE a burst of global_load_lds (LDS-DMA) instructions, as a function of how many
// are already outstanding on the CU?  One workgroup of W waves; every wave issues N 1-KiB DMA instructions (64 lanes x 16 B) from
// a cold (never touched) or hot (just read) region and records s_memtime after each.  Output: per wave, cycles from the first issue
// to the completion of issue i, and to the final s_waitcnt vmcnt(0).
// build: hipcc --offload-arch=gfx950 -O3 -o dma_issue.bin dma_issue.hip ; run: ./dma_issue.bin [waves=8] [n=8] [hot=0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* gbl_void_ptr;
constexpr int MAXN = 32;
__global__ void k(const float* __restrict__ src, unsigned long long* out, int n, int hot) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const float* base = src + ((size_t)blockIdx.x * nw + wave) * n * 256 + (size_t)lane * 4;
    if (hot) {
        float acc = 0;
        for (int i = 0; i < n; ++i) acc += base[i * 256];
        if (acc == 123.456f) out[0] = 1;
        __syncthreads();
    }
    unsigned long long t[MAXN + 2];
    float* dst = lds + wave * n * 256;
    t[0] = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < MAXN; ++i) {
        if (i < n) {
            __builtin_amdgcn_global_load_lds((gbl_void_ptr)(base + i * 256), (lds_void_ptr)(dst + i * 256), 16, 0, 0);
            t[i + 1] = __builtin_readcyclecounter();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t[MAXN + 1] = __builtin_readcyclecounter();
    if (lane == 0) {
        unsigned long long* o = out + ((size_t)blockIdx.x * nw + wave) * (MAXN + 2);
        for (int i = 0; i <= MAXN + 1; ++i) o[i] = (i <= n || i == MAXN + 1) ? t[i] - t[0] : 0;
    }
}
int main(int argc, char** argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 8, n = argc > 2 ? atoi(argv[2]) : 8, hot = argc > 3 ? atoi(argv[3]) : 0;
    const int blocks = argc > 4 ? atoi(argv[4]) : 1;
    float* src;
    unsigned long long* out;
    const size_t floats = (size_t)blocks * waves * n * 256 + 1024;
    hipMalloc(&src, floats * 4 * 2);
    hipMemset(src, 0, floats * 4 * 2);
    hipMalloc(&out, (size_t)blocks * waves * (MAXN + 2) * 8);
    hipDeviceSynchronize();
    // flush caches: touch a big buffer
    float* junk; hipMalloc(&junk, 512u << 20); hipMemset(junk, 1, 512u << 20); hipDeviceSynchronize();
    hipLaunchKernelGGL(k, dim3(blocks), dim3(waves * 64), waves * n * 1024, 0, src, out, n, hot);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)blocks * waves * (MAXN + 2));
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    printf("waves=%d n=%d hot=%d blocks=%d (s_memtime ticks since the wave's first issue)\n", waves, n, hot, blocks);
    for (int w = 0; w < waves; ++w) {
        printf("wave %d issue:", w);
        for (int i = 1; i <= n; ++i) printf(" %llu", h[(size_t)w * (MAXN + 2) + i]);
        printf("  | all landed: %llu\n", h[(size_t)w * (MAXN + 2) + MAXN + 1]);
    }
    return 0;
}
