// abi.hip -- version / error / memory-helper entry points of include/cnn_amd.h
#include <cstring>

#include "common.h"

namespace cnn_amd {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace cnn_amd

using namespace cnn_amd;

extern "C" {

int cnn_amd_abi_version(void) { return CNN_AMD_ABI_VERSION; }
const char* cnn_amd_last_error(void) { return error_buffer(); }

const char* cnn_amd_device_arch(void) {
    static thread_local char arch[256];
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        (void)hipGetLastError();
        snprintf(arch, sizeof(arch), "no HIP device visible");
        return arch;
    }
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
        snprintf(arch, sizeof(arch), "hipGetDeviceProperties failed");
        return arch;
    }
    // gcnArchName looks like "gfx950:sramecc+:xnack-"
    snprintf(arch, sizeof(arch), "%s", p.gcnArchName);
    char* colon = strchr(arch, ':');
    if (colon) *colon = 0;
    return arch;
}

int cnn_conv2d_out_dim(int in, int k, int s, int pad) { return (in + 2 * pad - k) / s + 1; }
int cnn_maxpool2d_out_dim(int in, int k, int step) { return (in - k) / step + 1; }

int cnn_device_alloc(void** ptr, size_t bytes) {
    CNN_REQUIRE(ptr != nullptr, "cnn_device_alloc: ptr is null");
    CNN_HIP_CHECK(hipMalloc(ptr, bytes ? bytes : 16));
    return CNN_AMD_OK;
}
int cnn_device_free(void* ptr) {
    if (ptr) CNN_HIP_CHECK(hipFree(ptr));
    return CNN_AMD_OK;
}
int cnn_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
    if (bytes == 0) return CNN_AMD_OK;
    CNN_REQUIRE(dst && src, "cnn_memcpy_h2d: null pointer");
    CNN_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return CNN_AMD_OK;
}
int cnn_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
    if (bytes == 0) return CNN_AMD_OK;
    CNN_REQUIRE(dst && src, "cnn_memcpy_d2h: null pointer");
    CNN_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return CNN_AMD_OK;
}
int cnn_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
    if (bytes == 0) return CNN_AMD_OK;
    CNN_REQUIRE(dst && src, "cnn_memcpy_d2d: null pointer");
    CNN_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return CNN_AMD_OK;
}
int cnn_memset_zero(void* dst, size_t bytes, void* stream) {
    if (bytes == 0) return CNN_AMD_OK;
    CNN_REQUIRE(dst != nullptr, "cnn_memset_zero: null pointer");
    CNN_HIP_CHECK(hipMemsetAsync(dst, 0, bytes, as_stream(stream)));
    return CNN_AMD_OK;
}
int cnn_stream_synchronize(void* stream) {
    CNN_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    return CNN_AMD_OK;
}

}  // extern "C"
