// load_issue.hip -- the register-staged alternative to tools/micro/dma_issue.hip: N plain 16-byte global loads per wave (compile
// time N = 8), then ONE timestamp (all issued), s_waitcnt vmcnt(0) (all landed), N ds_write_b128 (LDS filled).
// build: hipcc --offload-arch=gfx950 -O3 -o load_issue.bin load_issue.hip ; run: ./load_issue.bin [waves=8] [hot=0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int N = 8;
__global__ void kreg(const float* __restrict__ src, unsigned long long* out, int hot) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const float4* base = (const float4*)(src + ((size_t)blockIdx.x * nw + wave) * N * 256) + lane;
    if (hot) {
        float acc = 0;
        for (int i = 0; i < N; ++i) acc += base[i * 64].x;
        if (acc == 123.456f) out[0] = 1;
        __syncthreads();
    }
    float4* dst = (float4*)(lds + wave * N * 256) + lane;
    const unsigned long long t0 = __builtin_readcyclecounter();
    asm volatile("" ::: "memory");
    const float4 v0 = base[0], v1 = base[64], v2 = base[128], v3 = base[192], v4 = base[256], v5 = base[320], v6 = base[384], v7 = base[448];
    asm volatile("" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();  // issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();  // landed
    dst[0] = v0; dst[64] = v1; dst[128] = v2; dst[192] = v3; dst[256] = v4; dst[320] = v5; dst[384] = v6; dst[448] = v7;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t3 = __builtin_readcyclecounter();  // in LDS
    if (lane == 0) {
        unsigned long long* o = out + ((size_t)blockIdx.x * nw + wave) * 4;
        o[0] = t1 - t0; o[1] = t2 - t0; o[2] = t3 - t2; o[3] = (unsigned long long)(lds[wave * N * 256 + 5] == 77.f);
    }
}
int main(int argc, char** argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 8, hot = argc > 2 ? atoi(argv[2]) : 0;
    float* src; unsigned long long* out;
    const size_t floats = (size_t)waves * N * 256 + 1024;
    hipMalloc(&src, floats * 4); hipMemset(src, 0, floats * 4);
    hipMalloc(&out, waves * 4 * 8);
    float* junk; hipMalloc(&junk, 512u << 20); hipMemset(junk, 1, 512u << 20); hipDeviceSynchronize();
    hipLaunchKernelGGL(kreg, dim3(1), dim3(waves * 64), waves * N * 1024, 0, src, out, hot);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(waves * 4);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    printf("waves=%d n=%d hot=%d: per wave  issued | landed | ds_write x%d\n", waves, N, hot, N);
    for (int w = 0; w < waves; ++w) printf("wave %d: %llu | %llu | %llu\n", w, h[w * 4], h[w * 4 + 1], h[w * 4 + 2]);
    return 0;
}
