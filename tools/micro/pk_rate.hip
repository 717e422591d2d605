// pk_rate.hip -- issue rate of v_pk_fma_f32 against v_fma_f32 on gfx950, by operand kind: cycles per instruction and wave with W waves per
// SIMD (blockDim = 256 * W: one workgroup per CU).  Variants: 0 v_fma_f32 vgpr x vgpr; 1 v_fma_f32 sgpr x vgpr; 2 v_pk_fma_f32 vgpr pairs;
// 3 v_pk_fma_f32 sgpr pair x vgpr pair; 4 v_pk_fma_f32 sgpr pair x BROADCAST vgpr (op_sel_hi:[1,0,1]); 5 like 4 with 2 accumulators only
// (dependent chains of length N / 2).   build: hipcc --offload-arch=gfx950 -O3 -o pk_rate.bin pk_rate.hip ; run: ./pk_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int REP = 8192, UN = 16;
template <int V>
__global__ void k(float* out, unsigned long long* cyc, float s0, float s1) {
    v2f a[8];
    float x = threadIdx.x * 0.001f, y = x + 1.f;
    for (int i = 0; i < 8; ++i) a[i] = v2f{x + i, y + i};
    v2f b = v2f{x * 0.5f, y * 0.25f};
    v2f sp = v2f{s0, s1};
    asm volatile("" : "+s"(sp));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            v2f& c = a[V == 5 ? u % 2 : u % 8];
            if (V == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c.x) : "v"(b.x), "v"(b.y));
            if (V == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c.x) : "s"(sp.x), "v"(b.y));
            if (V == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(b), "v"(a[(u + 1) % 8]));
            if (V == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c) : "s"(sp), "v"(b));
            if (V == 4 || V == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(c) : "s"(sp), "v"(b));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xffff] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) & 0xfff] = t1 - t0;
}
template <int V>
void run(const char* name) {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 2048 * 4);
    hipMalloc(&cyc, 256 * 32 * 8);
    for (int w : {1, 2, 4}) {
        k<V><<<256, 256 * w>>>(out, cyc, 1.0001f, 0.9999f);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<V><<<256, 256 * w>>>(out, cyc, 1.0001f, 0.9999f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[4];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        const double instr_per_simd = 1.0 * w * REP * UN;  // per SIMD: one workgroup per CU, w waves per SIMD
        printf("%-44s waves/SIMD %d: s_memtime ticks per instr (wave 0) %.2f   wall ns per instr per SIMD %.3f\n", name, w, (double)h[0] / (REP * UN),
               ms * 1e6 / instr_per_simd);
    }
}
int main() {
    run<0>("v_fma_f32 vgpr,vgpr");
    run<1>("v_fma_f32 sgpr,vgpr");
    run<2>("v_pk_fma_f32 vgpr pairs");
    run<3>("v_pk_fma_f32 sgpr pair, vgpr pair");
    run<4>("v_pk_fma_f32 sgpr pair, broadcast vgpr");
    run<5>("v_pk_fma_f32 sgpr, bcast, 2 accumulators");
    return 0;
}
