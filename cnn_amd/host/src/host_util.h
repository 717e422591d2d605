// host_util.h -- internal helpers of the C++ host layer
#pragma once
#include <cstdio>
#include <cstdlib>

#include "cnn_amd.h"

namespace cnn_amd_host {
inline void must(int rc, const char* what) {
    if (rc != 0) {  // the reference has no error channel (assert only): abort with the ABI's message
        std::fprintf(stderr, "cnn_amd host: %s failed (%d): %s\n", what, rc, cnn_amd_last_error());
        std::abort();
    }
}
inline void* dev_alloc(size_t bytes) {
    void* p = nullptr;
    must(cnn_device_alloc(&p, bytes), "cnn_device_alloc");
    return p;
}
}  // namespace cnn_amd_host
