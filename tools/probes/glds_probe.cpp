// Probe: semantics of __builtin_amdgcn_global_load_lds on gfx950 (exec masking, per-lane source, dest = base + lane*size)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__global__ void probe(const float* src, float* out, int nactive) {
    __shared__ float lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = -1.f;
    __syncthreads();
    const int lane = threadIdx.x;
    if (lane < nactive) {
        // every active lane reads src[100 + 2*lane]; dest base = &lds[8]
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + 100 + 2 * lane), (lds_ptr_t)(&lds[8]), 4, 0, 0);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
    __syncthreads();
    // 16-byte variant: lane l reads src[4*l .. 4*l+3] reversed order of lanes, dest base = &lds[0]
    if (lane < 16) {
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + 4 * (15 - lane)), (lds_ptr_t)(&lds[64]), 16, 0, 0);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[256 + i] = lds[i];
}

int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 4096 * 4); hipMalloc(&o, 512 * 4);
    hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, o, 48);
    std::vector<float> r(512);
    hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
    printf("4B variant, 48 active lanes, base lds[8]:\n");
    for (int i = 0; i < 80; ++i) printf("%g ", r[i]);
    printf("\n16B variant, 16 active lanes (reversed sources), base lds[64]:\n");
    for (int i = 256 + 56; i < 256 + 140; ++i) printf("%g ", r[i]);
    printf("\n");
    return 0;
}
