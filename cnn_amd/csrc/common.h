// common.h -- shared host-side helpers for the libcnn_amd.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>

#include "cnn_amd.h"

namespace cnn_amd {

// last-error text (per thread), surfaced through cnn_amd_last_error()
char* error_buffer();
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Memo of a planning result per (desc, option-table generation), per host thread: the size queries walk every tile candidate of the
// implicit GEMM (34 plans) and are called by the prepared entry points and the host layers on EVERY launch -- 800+ plans per train step
// of the reference net, most of the 0.36 ms the host needed to enqueue a 0.39 ms step.
unsigned options_generation();
int num_cus();
struct DescMemo {
    struct Entry {
        cnn_conv2d_desc d;
        unsigned gen;
        int cus;  // (the planners size their grids by the CURRENT device's CU count: a thread that changes devices must not hit)
        bool used;
        size_t value;
    };
    Entry e[16] = {};
    int next = 0;
    bool find(const cnn_conv2d_desc* d, size_t* out) const {
        const unsigned gen = options_generation();
        const int cus = num_cus();
        for (const Entry& x : e)
            if (x.used && x.gen == gen && x.cus == cus && x.d.B == d->B && x.d.Ci == d->Ci && x.d.H == d->H && x.d.W == d->W && x.d.Co == d->Co && x.d.k == d->k &&
                x.d.s == d->s && x.d.pad == d->pad && x.d.flags == d->flags) {
                *out = x.value;
                return true;
            }
        return false;
    }
    void put(const cnn_conv2d_desc* d, size_t v) {
        e[next] = Entry{*d, options_generation(), num_cus(), true, v};
        next = (next + 1) % 16;
    }
};

// packed pool mask (include/cnn_amd.h, CNN_CONV2D_POOL_MASK_PACKED): bytes per pooled row (rows start 4-byte aligned: the window
// kernel moves four windows' bytes per 4-byte LDS-DMA)
__host__ __device__ inline int pool_mask_pitch(int PWo) { return (PWo + 3) & ~3; }
bool direct_pool_mask_packed_ok(const cnn_conv2d_desc* d);  // conv_direct.hip

// ---- measurement switches -------------------------------------------------------------------------------
// The A/B switches of the kernels' planners live in ONE process-wide table (abi.hip): filled once from the CNN_AMD_* variables of the
// environment when the library is first used, changed afterwards only through cnn_amd_set_option() (include/cnn_amd.h).  A launch
// path never calls getenv: a site holds a static Option whose cached value is re-read from the table only when the table's
// generation counter has moved (one relaxed atomic load per query otherwise).  Names are the variable names without "CNN_AMD_".
unsigned options_generation();
bool option_lookup(const char* name, long long* ival, double* dval);  // false: not set
class Option {
public:
    explicit Option(const char* n) : name_(n) {}
    bool is_set() { refresh(); return set_.load(std::memory_order_relaxed); }
    int as_int(int dflt) { refresh(); return set_.load(std::memory_order_relaxed) ? (int)ival_.load(std::memory_order_relaxed) : dflt; }
    double as_double(double dflt) { refresh(); return set_.load(std::memory_order_relaxed) ? dval_.load(std::memory_order_relaxed) : dflt; }
private:
    void refresh() {
        const unsigned g = options_generation();
        if (gen_.load(std::memory_order_acquire) == g) return;
        long long iv = 0;
        double dv = 0;
        const bool s = option_lookup(name_, &iv, &dv);
        ival_.store(iv, std::memory_order_relaxed);
        dval_.store(dv, std::memory_order_relaxed);
        set_.store(s, std::memory_order_relaxed);
        gen_.store(g, std::memory_order_release);
    }
    const char* name_;
    std::atomic<unsigned> gen_{0};  // (the table's generation starts at 1)
    std::atomic<bool> set_{false};
    std::atomic<long long> ival_{0};
    std::atomic<double> dval_{0};
};
// one static Option per call site
#define CNN_OPT(name) ([]() -> ::cnn_amd::Option& { static ::cnn_amd::Option o__(name); return o__; }())
#define CNN_OPT_SET(name) (CNN_OPT(name).is_set())
#define CNN_OPT_INT(name, dflt) (CNN_OPT(name).as_int(dflt))
// "is it set, and to which integer" in one value: `const OptVal e = CNN_OPT_VAL("X"); if (e && atoi(e) == 0) ...`
struct OptVal {
    bool set;
    int v;
    explicit operator bool() const { return set; }
};
inline int atoi(const OptVal& o) { return o.v; }
inline OptVal opt_val(Option& o) { return OptVal{o.is_set(), o.as_int(0)}; }
#define CNN_OPT_VAL(name) (::cnn_amd::opt_val(CNN_OPT(name)))

// MEASUREMENT-ONLY switches: ablations that leave out a kernel's stores / DMAs / MFMAs, per-phase cycle printers, an
// LDS request override, a data-parallel step without its exchange -- anything that changes RESULTS or exists for a timing experiment.
// They exist only in the measurement build (make -C cnn_amd/csrc measure -> libcnn_amd_measure.so, -DCNN_AMD_MEASURE, used by tools/);
// in the product library every such query is the compile-time constant `dflt`, cnn_amd_set_option() rejects the names and the
// CNN_AMD_* environment is not consulted for them (abi.hip: kMeasureOnly; tests/test_abi_exports.py).
#ifdef CNN_AMD_MEASURE
#define CNN_MEASURE_INT(name, dflt) CNN_OPT_INT(name, dflt)
#else
#define CNN_MEASURE_INT(name, dflt) (dflt)
#endif
bool measure_only_option(const char* name);  // (abi.hip) is `name` one of them

// compute units of the current device (hipDeviceProp_t::multiProcessorCount, cached per device; 256 on MI355X)
int num_cus();

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

// ---- "published" kernels: a cross-stream dependency without a marker packet on the producing stream -------------------------
// hipEventRecord between two kernels of one stream costs that stream ~4.5 us (measured, tools/micro/event_gap.hip: the marker
// packet has to retire before the next dispatch); an event attached to the producing kernel's OWN dispatch packet
// (hipExtLaunchKernelGGL's stopEvent) costs ~1.7 us and releases the waiting stream ~4 us sooner.  cnn_amd_publish_next_kernel(stream)
// arms the calling thread: the next kernel the library launches on `stream` carries the event -- through launch_pub() at the launch
// sites that sit in front of a fork in the train step, through a plain hipEventRecord behind any other launch (same meaning,
// old cost).  Consumers: cnn_amd_wait_published(), and the fork of cnn_conv2d_backward*(defer_join = 2).
struct PublishState {
    hipEvent_t ev[2] = {nullptr, nullptr};
    int cur = 0, device = -1;
    bool armed = false;   // the next launch on `stream` publishes
    bool valid = false;   // ev[cur] marks the completion of the last published kernel ...
    bool stale = false;   // ... but the library has launched something else on that stream since
    hipStream_t stream = nullptr;
};
PublishState& publish_state();                   // per host thread (abi.hip)
hipEvent_t publish_take(hipStream_t s);          // armed for s: the event to attach (state -> published), else nullptr
int publish_after_launch(hipStream_t s);         // bookkeeping behind EVERY launch (fallback record / staleness)
void publish_mark_stale(hipStream_t s);         // something that is NOT a library kernel was queued on s (copies, waits, collectives)
inline bool publish_busy() {
    const PublishState& p = publish_state();
    return p.armed || (p.valid && !p.stale);
}

// kernel launch that can carry the published event (use inside CNN_KLAUNCH instead of kernel<<<...>>>(...))
template <typename... KArgs, typename... Args>
inline void launch_pub(void (*kern)(KArgs...), dim3 grid, dim3 block, unsigned shmem, hipStream_t s, Args&&... args) {
    hipEvent_t ev = publish_take(s);
    if (ev) hipExtLaunchKernelGGL(kern, grid, block, shmem, s, nullptr, ev, 0, static_cast<KArgs>(args)...);
    else hipLaunchKernelGGL(kern, grid, block, shmem, s, static_cast<KArgs>(args)...);
}

// launch `expr` (a kernel<<<...>>>(...) expression), timed when profiling is on; `...` = printf-style geometry tag
#define CNN_KLAUNCH(stream, kernel_name, expr, ...)                                         \
    do {                                                                                    \
        const bool t__ = ::cnn_amd::ktimer_active();                                        \
        if (t__) ::cnn_amd::ktimer_begin(stream, kernel_name, __VA_ARGS__);                 \
        expr;                                                                               \
        if (t__) ::cnn_amd::ktimer_end(stream);                                             \
        CNN_LAUNCH_CHECK();                                                                 \
        if (::cnn_amd::publish_busy())                                                      \
            if (int prc__ = ::cnn_amd::publish_after_launch(stream)) return prc__;          \
    } while (0)

#define CNN_REQUIRE(cond, ...)                                             \
    do {                                                                   \
        if (!(cond)) return ::cnn_amd::fail(CNN_AMD_E_BADARG, __VA_ARGS__); \
    } while (0)

// conv_dgrad_rd.hip's m16 kernel keeps the filters of a 16-input-channel slice as MFMA A operands: register j = c4*9 + tap
// (0 .. 2.25*Co - 1) of lane l = (ci = l & 15, k = l >> 4) holds w[4*c4 + k][16*slice + ci][tap] ([Co][Ci][3][3] layout).
// The prepared copy is img[(slice*NA + j)*64 + l], NA = 2.25*Co (one coalesced load per j).
__host__ __device__ inline int m16_filter_index(int j, int lane, int Ci, int slice) {
    return ((4 * (j / 9) + (lane >> 4)) * Ci + 16 * slice + (lane & 15)) * 9 + j % 9;
}

// "has this per-function attribute been set on the CURRENT device yet" (hipFuncSetAttribute is per device): one atomic flag per
// device and call site.  (racing threads may both set it: setting the attribute twice is
// harmless, never setting it on a second device -- or launching before it is set -- is not): needed() until mark().
struct DeviceOnce {
    std::atomic<unsigned char> done[64];
    static int device() {
        int dev = 0;
        return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) ? dev : -1;
    }
    bool needed() const {
        const int dev = device();
        return dev < 0 || done[dev].load(std::memory_order_acquire) == 0;
    }
    void mark() {
        const int dev = device();
        if (dev >= 0) done[dev].store(1, std::memory_order_release);
    }
};

// SGD on one element (alexnet.cpp:62-65, conv2d.cpp / linear.cpp update_gradients):
// p - lr*(g*scale) with every product/sum rounded separately: the reference is built without FMA
// (x86-64 -O2, CMakeLists.txt:5), so w -= lr*g is mul-then-sub; the fp-contract pragma stops hipcc fusing it
// (HIP's __fmul_rn/__fsub_rn are plain operators and do get contracted).
__device__ __forceinline__ float sgd_one(float p, float g, float lr, float scale, bool scaled) {
#pragma clang fp contract(off)
    const float gs = scaled ? g * scale : g;
    const float step = lr * gs;
    return p - step;
}

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kNumXCD = 8;

// Workgroups are dealt round-robin to the 8 XCDs (id % 8), each with its own L2: neighbouring tiles of an image -- which share
// halo rows -- would sit on eight different L2s and fetch those rows eight times.  This gives XCD x the x-th CONTIGUOUS eighth of
// the logical ids instead (same bijection as the implicit GEMM's tile map).  nb = gridDim.x.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned bid, unsigned nb) {
#ifdef CNN_NO_XCD_SWIZZLE  // (A/B builds only)
    return bid;
#endif
    const unsigned q = nb / kNumXCD, r = nb % kNumXCD, xcd = bid % kNumXCD, k = bid / kNumXCD;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

inline unsigned ceil_div(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// grid for HBM-bound streaming kernels: enough workgroups to fill 256 CUs x 8, grid-stride the rest
inline unsigned stream_grid(size_t work_items, int block) {
    size_t need = (work_items + block - 1) / block;
    size_t cap = (size_t)num_cus() * 8;
    return (unsigned)(need < 1 ? 1 : (need > cap ? cap : need));
}

}  // namespace cnn_amd
