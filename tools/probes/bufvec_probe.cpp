// Probe: multi-dword raw buffer loads (b64 / b128) with a stride-0 descriptor, at 4-byte-aligned offsets.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(const float* p, float* o, int nbytes, int shift, int flags) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, flags);
    const int voff = threadIdx.x * 16 + shift * 4;
    v4i a = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
    v2i b = __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0);
    float* out = o + threadIdx.x * 6;
    out[0] = __builtin_bit_cast(float, a.x); out[1] = __builtin_bit_cast(float, a.y);
    out[2] = __builtin_bit_cast(float, a.z); out[3] = __builtin_bit_cast(float, a.w);
    out[4] = __builtin_bit_cast(float, b.x); out[5] = __builtin_bit_cast(float, b.y);
}
int main() {
    const int N = 64 * 4 + 8;
    float h[N], *d, *o, r[64 * 6];
    for (int i = 0; i < N; ++i) h[i] = (float)i;
    (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&o, sizeof(r));
    (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    for (int flags : {0x00020000, 0x00027000, 0x00077FAC, 0x00027FAC, 0x00024FAC})
    for (int shift = 0; shift < 2; ++shift) {
        k<<<1, 64>>>(d, o, 64 * 16, shift, flags);   // num_records = 1024 bytes: the last lanes partly out of range when shifted
        (void)hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
        for (int l : {1, 63}) printf("flags %08x shift %d lane %2d: b128 %g %g %g %g | b64 %g %g\n", flags, shift, l, r[l*6], r[l*6+1], r[l*6+2], r[l*6+3], r[l*6+4], r[l*6+5]);
    }
    return 0;
}
