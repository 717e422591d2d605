// Probe: what does buffer_load_dword ... lds (LDS-DMA through a buffer descriptor) write for lanes whose offset is out
// of range?  Build: hipcc --offload-arch=gfx950 -O2 -o buflds_probe.bin tools/probes/buflds_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void k(const float* p, float* o, int nbytes) {
    __shared__ float buf[64];
    buf[threadIdx.x] = -7.f;  // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
    const unsigned voff = (threadIdx.x & 1) ? 0x7ffffffcu : threadIdx.x * 4u;  // odd lanes out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)&buf[0], 4, (int)voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    o[threadIdx.x] = buf[threadIdx.x];
}
int main() {
    float h[64], *d, *o;
    for (int i = 0; i < 64; ++i) h[i] = 100.f + i;
    hipMalloc(&d, 256); hipMalloc(&o, 256);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, 256);
    hipMemcpy(h, o, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i) printf("lane %d -> %g\n", i, h[i]);
    return 0;
}
