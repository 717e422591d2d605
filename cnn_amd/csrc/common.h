// common.h -- shared host-side helpers for the libcnn_amd.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "cnn_amd.h"

namespace cnn_amd {

// last-error text (per thread), surfaced through cnn_amd_last_error()
char* error_buffer();
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

#define CNN_HIP_CHECK(expr)                                                                         \
    do {                                                                                            \
        hipError_t e__ = (expr);                                                                    \
        if (e__ != hipSuccess)                                                                      \
            return ::cnn_amd::fail(CNN_AMD_E_HIP + (int)e__, "%s failed: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

// every kernel launch is followed by this: catches bad launch configs without synchronising
#define CNN_LAUNCH_CHECK() CNN_HIP_CHECK(hipGetLastError())

// Optional per-kernel timing with HIP events recorded on the launch stream (cnn_amd_kernel_timing_* in the ABI).
// mode 0 = off (one predictable branch per launch), 1 = every kernel, 2 = only records whose key contains `filter`.
bool ktimer_active();
void ktimer_begin(hipStream_t s, const char* kernel, const char* fmt, ...);
void ktimer_end(hipStream_t s);

// launch `expr` (a kernel<<<...>>>(...) expression), timed when profiling is on; `...` = printf-style geometry tag
#define CNN_KLAUNCH(stream, kernel_name, expr, ...)                                         \
    do {                                                                                    \
        const bool t__ = ::cnn_amd::ktimer_active();                                        \
        if (t__) ::cnn_amd::ktimer_begin(stream, kernel_name, __VA_ARGS__);                 \
        expr;                                                                               \
        if (t__) ::cnn_amd::ktimer_end(stream);                                             \
        CNN_LAUNCH_CHECK();                                                                 \
    } while (0)

#define CNN_REQUIRE(cond, ...)                                             \
    do {                                                                   \
        if (!(cond)) return ::cnn_amd::fail(CNN_AMD_E_BADARG, __VA_ARGS__); \
    } while (0)

// conv_dgrad_rd.hip's Ci = 16 kernel keeps the whole filter bank as MFMA A operands: register j (0 .. 2.25*Co - 1) of lane l
// holds w[m16_filter_index(j, l, Co)] ([Co][16][3][3] layout).  The prepared copy is img[j*64 + l] (one coalesced load per j).
__host__ __device__ inline int m16_filter_index(int j, int lane, int Co) {
    const int n = lane & 15, k = lane >> 4;
    int co, tap;
    if (j < Co) { co = j; tap = (2 * (k >> 1)) * 3 + 2 * (k & 1); }                                      // class (0,0): k = 2jr + jc
    else if (j < Co + Co / 2) { co = 2 * (j - Co) + (k >> 1); tap = (2 * (k & 1)) * 3 + 1; }           // class (0,1): k = (channel, jr)
    else if (j < 2 * Co) { co = 2 * (j - Co - Co / 2) + (k >> 1); tap = 3 + 2 * (k & 1); }            // class (1,0): k = (channel, jc)
    else { co = 4 * (j - 2 * Co) + k; tap = 4; }                                                        // class (1,1): k = channel
    return (co * 16 + n) * 9 + tap;
}

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kNumCU = 256;        // MI355X
constexpr int kNumXCD = 8;

inline unsigned ceil_div(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// grid for HBM-bound streaming kernels: enough workgroups to fill 256 CUs x 8, grid-stride the rest
inline unsigned stream_grid(size_t work_items, int block) {
    size_t need = (work_items + block - 1) / block;
    size_t cap = (size_t)kNumCU * 8;
    return (unsigned)(need < 1 ? 1 : (need > cap ? cap : need));
}

}  // namespace cnn_amd
