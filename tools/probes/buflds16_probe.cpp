// Probe: buffer_load_dwordx4 ... lds (16 bytes per lane) from source addresses that are only 4-byte aligned, and what a lane whose 16 bytes
// straddle the end of the buffer receives.  Build: hipcc --offload-arch=gfx950 -O2 -o buflds16_probe.bin tools/probes/buflds16_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void k(const float* p, float* o, int nbytes, int shift_floats) {
    __shared__ __attribute__((aligned(16))) float buf[256];
    for (int i = threadIdx.x; i < 256; i += 64) buf[i] = -7.f;  // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
    const unsigned voff = threadIdx.x * 16u + shift_floats * 4u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)&buf[0], 16, (int)voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) o[i] = buf[i];
}
int main() {
    float h[300], *d, *o;
    for (int i = 0; i < 300; ++i) h[i] = 100.f + i;
    (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&o, 1024);
    (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        k<<<1, 64>>>(d, o, 254 * 4, shift);  // 254 floats in range: the last lanes straddle / leave the buffer
        float r[256];
        (void)hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) {
            const int src = i + shift;
            const float want = src < 254 ? 100.f + src : 0.f;
            if (r[i] != want) ++bad;
        }
        printf("shift %d floats: %d of 256 floats differ from (in range ? data : 0); lane 0: %g %g %g %g  lane 63: %g %g %g %g\n", shift, bad, r[0], r[1], r[2],
               r[3], r[252], r[253], r[254], r[255]);
    }
    return 0;
}
