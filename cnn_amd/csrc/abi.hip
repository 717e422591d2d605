// abi.hip -- version / error / memory-helper entry points of include/cnn_amd.h
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

extern char** environ;  // (POSIX; read once by the option table below)

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

// ---- measurement switches (common.h) ---------------------------------------------------------------------------------------
// Names that exist only in the measurement build (common.h, CNN_MEASURE_INT): ablations that change results, cycle printers, overrides.
static const char* const kMeasureOnly[] = {"DBG",    "ROWS_DBG",   "ROWS_LDS",     "WIN_DBG", "OS_DBG",    "STEM_DBG",
                                           "FWD_RD_DBG", "DGRAD_RD_DBG", "RD_DBG",  "SP_DBG",    "S2_DBG", "DP_SKIP_EXCHANGE"};
bool measure_only_option(const char* name) {
    for (const char* m : kMeasureOnly)
        if (strcmp(name, m) == 0) return true;
    return false;
}
#ifdef CNN_AMD_MEASURE
static constexpr bool kMeasureBuild = true;
#else
static constexpr bool kMeasureBuild = false;
#endif
namespace {
struct OptionTable {
    std::mutex mu;
    std::map<std::string, std::string> values;
    std::atomic<unsigned> generation{1};
    bool env_loaded = false;
    void load_env_locked() {
        if (env_loaded) return;
        env_loaded = true;
        for (char** e = ::environ; e && *e; ++e) {
            if (strncmp(*e, "CNN_AMD_", 8) != 0) continue;
            const char* eq = strchr(*e, '=');
            if (!eq) continue;
            const std::string key(*e + 8, (size_t)(eq - (*e + 8)));
            if (!kMeasureBuild && measure_only_option(key.c_str())) continue;  // (the product library cannot be armed through the environment)
            if (!values.count(key)) values[key] = eq + 1;  // (cnn_amd_set_option before the first query wins)
        }
    }
};
OptionTable& option_table() {
    static OptionTable t;
    return t;
}
}  // namespace
unsigned options_generation() { return option_table().generation.load(std::memory_order_acquire); }
bool option_lookup(const char* name, long long* ival, double* dval) {
    OptionTable& t = option_table();
    std::lock_guard<std::mutex> lk(t.mu);
    t.load_env_locked();
    auto it = t.values.find(name);
    if (it == t.values.end()) return false;
    *ival = atoll(it->second.c_str());
    *dval = atof(it->second.c_str());
    return true;
}

int num_cus() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n > 0) return n;
    hipDeviceProp_t prop;
    n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cached[dev].store(n, std::memory_order_relaxed);
    return n;
}

// ---- published kernels (common.h) ---------------------------------------------------------------------------------------------
namespace {
// one state per (host thread, device), like the side streams of conv_backward.hip: a thread that alternates between devices keeps
// both event pairs instead of leaking one per switch.  The slot is re-selected where a device query is affordable (arming);
// launches use the slot selected last.
struct PublishSlots {
    PublishState st[16];
    int cur = 0;
};
PublishSlots& publish_slots() {
    static thread_local PublishSlots s;
    return s;
}
}  // namespace
PublishState& publish_state() {
    PublishSlots& s = publish_slots();
    return s.st[s.cur];
}
namespace {
int publish_events(PublishState*& out) {
    int dev = 0;
    CNN_HIP_CHECK(hipGetDevice(&dev));
    PublishSlots& s = publish_slots();
    s.cur = (dev >= 0 ? dev : 0) % 16;
    PublishState& p = s.st[s.cur];
    if (p.ev[0] == nullptr || p.device != dev) {
        CNN_HIP_CHECK(hipEventCreateWithFlags(&p.ev[0], hipEventDisableTiming));
        CNN_HIP_CHECK(hipEventCreateWithFlags(&p.ev[1], hipEventDisableTiming));
        p.device = dev;
        p.valid = false;
    }
    out = &p;
    return CNN_AMD_OK;
}
thread_local bool g_just_published = false;
}  // namespace

hipEvent_t publish_take(hipStream_t s) {
    PublishState& p = publish_state();
    if (!p.armed || p.stream != s) return nullptr;
    p.cur ^= 1;
    p.armed = false;
    p.valid = true;
    p.stale = false;
    g_just_published = true;
    return p.ev[p.cur];
}

void publish_mark_stale(hipStream_t s) {
    PublishState& p = publish_state();
    if (p.valid && p.stream == s) p.stale = true;
    g_just_published = false;
}

int publish_after_launch(hipStream_t s) {
    PublishState& p = publish_state();
    if (p.armed && p.stream == s) {  // launched by a site without launch_pub(): the marker packet after all
        p.cur ^= 1;
        CNN_HIP_CHECK(hipEventRecord(p.ev[p.cur], s));
        p.armed = false;
        p.valid = true;
        p.stale = false;
        g_just_published = false;
        return CNN_AMD_OK;
    }
    if (p.valid && p.stream == s) {
        if (g_just_published) g_just_published = false;
        else p.stale = true;
    }
    return CNN_AMD_OK;
}

// ---- per-kernel event timing -----------------------------------------------------------------------------
namespace {
struct KRecord {
    std::string key;
    hipEvent_t e0, e1;
};
struct KTimer {
    int mode = 0;
    int every = 1;       // mode 2: bracket only every `every`-th matching launch (an event pair costs the stream two ~6 us bubbles)
    long long seen = 0;
    std::string filter;
    std::vector<KRecord> recs;
    unsigned long long epoch = 1;  // moves when recs is cleared
    std::mutex mu;
};
// the record a begin() of THIS host thread opened and its end() closes: per thread (ADVICE r5 -- one host thread per rank in one
// process interleaves begin / end pairs; a process-wide "last record" would put e1 on another rank's stream), valid for one epoch
struct KOpen {
    unsigned long long epoch = 0;
    long long idx = -1;
};
thread_local KOpen k_open;
KTimer& kt() {
    static KTimer t;
    return t;
}
}  // namespace

bool ktimer_active() { return kt().mode != 0; }

void ktimer_begin(hipStream_t s, const char* kernel, const char* fmt, ...) {
    KTimer& t = kt();
    char tag[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tag, sizeof(tag), fmt, ap);
    va_end(ap);
    std::string key = std::string(kernel) + "|" + tag;
    std::lock_guard<std::mutex> lk(t.mu);
    k_open.idx = -1;
    if (t.mode == 2 && key.find(t.filter) == std::string::npos) return;
    if (t.mode == 2 && (t.seen++ % t.every) != 0) return;
    KRecord r;
    r.key = key;
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    (void)hipEventRecord(r.e0, s);
    t.recs.push_back(r);
    k_open.epoch = t.epoch;
    k_open.idx = (long long)t.recs.size() - 1;
}

void ktimer_end(hipStream_t s) {
    KTimer& t = kt();
    std::lock_guard<std::mutex> lk(t.mu);
    if (k_open.idx < 0 || k_open.epoch != t.epoch || k_open.idx >= (long long)t.recs.size()) return;
    (void)hipEventRecord(t.recs[(size_t)k_open.idx].e1, s);
    k_open.idx = -1;
}

}  // namespace cnn_amd

using namespace cnn_amd;

extern "C" {

int cnn_amd_abi_version(void) { return CNN_AMD_ABI_VERSION; }

int cnn_amd_timing_span_begin(void* stream, const char* name) {
    CNN_REQUIRE(name != nullptr, "cnn_amd_timing_span_begin: null name");
    if (ktimer_active()) ktimer_begin(as_stream(stream), name, "span");
    return CNN_AMD_OK;
}
int cnn_amd_timing_span_end(void* stream) {
    if (ktimer_active()) ktimer_end(as_stream(stream));
    return CNN_AMD_OK;
}
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

int cnn_amd_kernel_timing_enable(int mode, const char* filter) {
    KTimer& t = kt();
    std::lock_guard<std::mutex> lk(t.mu);
    CNN_REQUIRE(mode >= 0 && mode <= 2, "cnn_amd_kernel_timing_enable: mode %d", mode);
    t.mode = mode;
    t.filter = filter ? filter : "";
    t.seen = 0;
    return CNN_AMD_OK;
}

int cnn_amd_kernel_timing_sampling(int every) {
    KTimer& t = kt();
    std::lock_guard<std::mutex> lk(t.mu);
    CNN_REQUIRE(every >= 1, "cnn_amd_kernel_timing_sampling: every = %d", every);
    t.every = every;
    return CNN_AMD_OK;
}

// Synchronises the device, then writes one line per distinct key: "<key>\t<launches>\t<total_ms>\n" and clears
// the records.  Returns the number of bytes needed (call again with a larger buffer if > cap; records are kept).
long long cnn_amd_kernel_timing_report(char* buf, size_t cap) {
    KTimer& t = kt();
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    std::lock_guard<std::mutex> lk(t.mu);
    std::map<std::string, std::pair<long long, double>> agg;
    std::vector<std::string> order;
    for (auto& r : t.recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
        auto it = agg.find(r.key);
        if (it == agg.end()) {
            order.push_back(r.key);
            agg[r.key] = {1, ms};
        } else {
            it->second.first += 1;
            it->second.second += ms;
        }
    }
    std::string out;
    char line[512];
    for (auto& k : order) {
        snprintf(line, sizeof(line), "%s\t%lld\t%.6f\n", k.c_str(), agg[k].first, agg[k].second);
        out += line;
    }
    if (out.size() + 1 > cap || !buf) return (long long)out.size() + 1;
    memcpy(buf, out.c_str(), out.size() + 1);
    for (auto& r : t.recs) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    t.recs.clear();
    ++t.epoch;
    return (long long)out.size() + 1;
}

// (a stride of 0 is a caller's bug, not a reason to raise SIGFPE in its process: 0 outputs)
int cnn_conv2d_out_dim(int in, int k, int s, int pad) { return s > 0 ? (in + 2 * pad - k) / s + 1 : 0; }
int cnn_maxpool2d_out_dim(int in, int k, int step) { return step > 0 ? (in - k) / step + 1 : 0; }

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
    publish_mark_stale(as_stream(stream));
    return CNN_AMD_OK;
}
int cnn_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
    if (bytes == 0) return CNN_AMD_OK;
    CNN_REQUIRE(dst && src, "cnn_memcpy_d2h: null pointer");
    CNN_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    publish_mark_stale(as_stream(stream));
    return CNN_AMD_OK;
}
int cnn_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
    if (bytes == 0) return CNN_AMD_OK;
    CNN_REQUIRE(dst && src, "cnn_memcpy_d2d: null pointer");
    CNN_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    publish_mark_stale(as_stream(stream));
    return CNN_AMD_OK;
}
int cnn_memset_zero(void* dst, size_t bytes, void* stream) {
    if (bytes == 0) return CNN_AMD_OK;
    CNN_REQUIRE(dst != nullptr, "cnn_memset_zero: null pointer");
    CNN_HIP_CHECK(hipMemsetAsync(dst, 0, bytes, as_stream(stream)));
    publish_mark_stale(as_stream(stream));
    return CNN_AMD_OK;
}
int cnn_stream_synchronize(void* stream) {
    CNN_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    return CNN_AMD_OK;
}
int cnn_stream_create(void** stream) {
    CNN_REQUIRE(stream != nullptr, "cnn_stream_create: null pointer");
    hipStream_t s = nullptr;
    CNN_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return CNN_AMD_OK;
}
int cnn_stream_create_priority(void** stream, int level) {
    CNN_REQUIRE(stream != nullptr, "cnn_stream_create_priority: null pointer");
    int least = 0, greatest = 0;  // (numerically: greatest priority = the smallest number)
    CNN_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t s = nullptr;
    CNN_HIP_CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, level < 0 ? greatest : (level > 0 ? least : 0)));
    *stream = s;
    return CNN_AMD_OK;
}
int cnn_stream_destroy(void* stream) {
    if (stream) CNN_HIP_CHECK(hipStreamDestroy(as_stream(stream)));
    return CNN_AMD_OK;
}
int cnn_event_create(void** event) {
    CNN_REQUIRE(event != nullptr, "cnn_event_create: null pointer");
    hipEvent_t e = nullptr;
    CNN_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event = e;
    return CNN_AMD_OK;
}
int cnn_event_destroy(void* event) {
    if (event) CNN_HIP_CHECK(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return CNN_AMD_OK;
}
int cnn_event_record(void* event, void* stream) {
    CNN_REQUIRE(event != nullptr, "cnn_event_record: null event");
    CNN_HIP_CHECK(hipEventRecord(reinterpret_cast<hipEvent_t>(event), as_stream(stream)));
    return CNN_AMD_OK;
}
int cnn_stream_wait_event(void* stream, void* event) {
    CNN_REQUIRE(event != nullptr, "cnn_stream_wait_event: null event");
    CNN_HIP_CHECK(hipStreamWaitEvent(as_stream(stream), reinterpret_cast<hipEvent_t>(event), 0));
    publish_mark_stale(as_stream(stream));  // (a fork taken from an earlier published kernel would miss this dependency)
    return CNN_AMD_OK;
}
int cnn_stream_wait_event_local(void* stream, void* event) {
    CNN_REQUIRE(event != nullptr, "cnn_stream_wait_event_local: null event");
    CNN_HIP_CHECK(hipStreamWaitEvent(as_stream(stream), reinterpret_cast<hipEvent_t>(event), 0));
    return CNN_AMD_OK;
}
int cnn_host_alloc_pinned(void** ptr, size_t bytes) {
    CNN_REQUIRE(ptr != nullptr, "cnn_host_alloc_pinned: ptr is null");
    CNN_HIP_CHECK(hipHostMalloc(ptr, bytes ? bytes : 16, hipHostMallocDefault));
    return CNN_AMD_OK;
}
int cnn_host_free_pinned(void* ptr) {
    if (ptr) CNN_HIP_CHECK(hipHostFree(ptr));
    return CNN_AMD_OK;
}

// ---- input staging -------------------------------------------------------------------------------------------------
namespace {
struct Stager {
    size_t bytes = 0;
    int depth = 0, next = 0;
    hipStream_t copy = nullptr;
    std::vector<void*> host, dev;
    // u8 staging (cnn_batch_stager_create_u8): `dev` holds the uploaded bytes, `dev_f32` the converted batch submit() hands out
    std::vector<void*> dev_f32;
    float* lut = nullptr;  // device: lut[v] = v * 1.f / 255, computed on the host
    int B = 0, H = 0, W = 0;
    std::vector<hipEvent_t> uploaded, consumed;
    std::vector<char> in_use;  // consumed[i] has been recorded at least once
};

// Tensor3D::read_from_opencv_mat (data_format.cpp:13-23) for a whole batch: interleaved bytes [B][H*W][3] -> planar fp32 [B][3][H*W].
// A thread converts four consecutive pixels: 12 bytes in (three 4-byte loads), one 16-byte store into each channel plane.  The table
// lives in LDS (no arithmetic on the device: the host's v * 1.f / 255, bit for bit).
__global__ __launch_bounds__(256) void u8hwc_to_f32chw(const unsigned* __restrict__ src, float* __restrict__ dst, const float* __restrict__ lut_g,
                                                       int hw4, long long quads_total) {
    __shared__ float lut[256];
    lut[threadIdx.x] = lut_g[threadIdx.x];
    __syncthreads();
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < quads_total; q += (long long)gridDim.x * 256) {
        const long long b = q / hw4;
        const int i4 = (int)(q - b * hw4);  // quad index inside the image
        const unsigned w0 = src[3 * q], w1 = src[3 * q + 1], w2 = src[3 * q + 2];  // bytes p0c0 p0c1 p0c2 p1c0 | p1c1 p1c2 p2c0 p2c1 | p2c2 p3c0 p3c1 p3c2
        float4 c0, c1, c2;
        c0.x = lut[w0 & 255u];         c1.x = lut[(w0 >> 8) & 255u];  c2.x = lut[(w0 >> 16) & 255u];
        c0.y = lut[w0 >> 24];          c1.y = lut[w1 & 255u];         c2.y = lut[(w1 >> 8) & 255u];
        c0.z = lut[(w1 >> 16) & 255u]; c1.z = lut[w1 >> 24];          c2.z = lut[w2 & 255u];
        c0.w = lut[(w2 >> 8) & 255u];  c1.w = lut[(w2 >> 16) & 255u]; c2.w = lut[w2 >> 24];
        float* d = dst + (size_t)b * 12 * hw4 + 4 * (size_t)i4;  // image b, channel 0, pixel 4*i4
        *(float4*)d = c0;
        *(float4*)(d + 4 * (size_t)hw4) = c1;
        *(float4*)(d + 8 * (size_t)hw4) = c2;
    }
}
// ... and for images whose pixel count is no multiple of four (round 6: any H, W): one thread per pixel, three byte loads, three stores
__global__ __launch_bounds__(256) void u8hwc_to_f32chw_px(const unsigned char* __restrict__ src, float* __restrict__ dst, const float* __restrict__ lut_g,
                                                          int hw, long long px_total) {
    __shared__ float lut[256];
    lut[threadIdx.x] = lut_g[threadIdx.x];
    __syncthreads();
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < px_total; q += (long long)gridDim.x * 256) {
        const long long b = q / hw;
        const int i = (int)(q - b * hw);
        float* d = dst + (size_t)b * 3 * hw + i;
        d[0] = lut[src[3 * q]];
        d[(size_t)hw] = lut[src[3 * q + 1]];
        d[2 * (size_t)hw] = lut[src[3 * q + 2]];
    }
}
}  // namespace

int cnn_batch_stager_create(void** stager, size_t batch_bytes, int depth) {
    CNN_REQUIRE(stager && batch_bytes > 0 && depth >= 2 && depth <= 16, "cnn_batch_stager_create: bad arguments (depth 2..16)");
    Stager* st = new Stager();
    st->bytes = batch_bytes;
    st->depth = depth;
    st->host.assign(depth, nullptr); st->dev.assign(depth, nullptr);
    st->uploaded.assign(depth, nullptr); st->consumed.assign(depth, nullptr);
    st->in_use.assign(depth, 0);
    *stager = st;
    CNN_HIP_CHECK(hipStreamCreateWithFlags(&st->copy, hipStreamNonBlocking));
    for (int i = 0; i < depth; ++i) {
        CNN_HIP_CHECK(hipHostMalloc(&st->host[i], batch_bytes, hipHostMallocDefault));
        CNN_HIP_CHECK(hipMalloc(&st->dev[i], batch_bytes));
        CNN_HIP_CHECK(hipEventCreateWithFlags(&st->uploaded[i], hipEventDisableTiming));
        CNN_HIP_CHECK(hipEventCreateWithFlags(&st->consumed[i], hipEventDisableTiming));
    }
    return CNN_AMD_OK;
}

int cnn_batch_stager_create_u8(void** stager, int B, int H, int W, int depth) {
    CNN_REQUIRE(stager && B > 0 && H > 0 && W > 0 && depth >= 2 && depth <= 16, "cnn_batch_stager_create_u8: bad arguments (depth 2..16)");
    Stager* st = new Stager();
    st->bytes = (size_t)B * H * W * 3;
    st->depth = depth;
    st->B = B; st->H = H; st->W = W;
    st->host.assign(depth, nullptr); st->dev.assign(depth, nullptr); st->dev_f32.assign(depth, nullptr);
    st->uploaded.assign(depth, nullptr); st->consumed.assign(depth, nullptr);
    st->in_use.assign(depth, 0);
    *stager = st;
    CNN_HIP_CHECK(hipStreamCreateWithFlags(&st->copy, hipStreamNonBlocking));
    float table[256];
    for (int v = 0; v < 256; ++v) table[v] = (unsigned char)v * 1.f / 255;  // data_format.cpp:19-21, the reference's own expression
    CNN_HIP_CHECK(hipMalloc((void**)&st->lut, sizeof(table)));
    CNN_HIP_CHECK(hipMemcpy(st->lut, table, sizeof(table), hipMemcpyHostToDevice));
    for (int i = 0; i < depth; ++i) {
        CNN_HIP_CHECK(hipHostMalloc(&st->host[i], st->bytes, hipHostMallocDefault));
        CNN_HIP_CHECK(hipMalloc(&st->dev[i], st->bytes));
        CNN_HIP_CHECK(hipMalloc(&st->dev_f32[i], st->bytes * sizeof(float)));
        CNN_HIP_CHECK(hipEventCreateWithFlags(&st->uploaded[i], hipEventDisableTiming));
        CNN_HIP_CHECK(hipEventCreateWithFlags(&st->consumed[i], hipEventDisableTiming));
    }
    return CNN_AMD_OK;
}

int cnn_batch_stager_destroy(void* stager) {
    Stager* st = static_cast<Stager*>(stager);
    if (!st) return CNN_AMD_OK;
    if (st->copy) (void)hipStreamSynchronize(st->copy);
    if (st->lut) (void)hipFree(st->lut);
    for (void* p : st->dev_f32)
        if (p) (void)hipFree(p);
    for (int i = 0; i < st->depth; ++i) {
        if (st->host[i]) (void)hipHostFree(st->host[i]);
        if (st->dev[i]) (void)hipFree(st->dev[i]);
        if (st->uploaded[i]) (void)hipEventDestroy(st->uploaded[i]);
        if (st->consumed[i]) (void)hipEventDestroy(st->consumed[i]);
    }
    if (st->copy) (void)hipStreamDestroy(st->copy);
    delete st;
    return CNN_AMD_OK;
}

int cnn_batch_stager_acquire(void* stager, void** pinned_host, int* slot) {
    Stager* st = static_cast<Stager*>(stager);
    CNN_REQUIRE(st && pinned_host && slot, "cnn_batch_stager_acquire: null pointer");
    const int i = st->next;
    st->next = (st->next + 1) % st->depth;
    // the previous upload out of this slot must have left the host buffer, and its consumer must be done with the device one
    CNN_HIP_CHECK(hipEventSynchronize(st->uploaded[i]));
    if (st->in_use[i]) CNN_HIP_CHECK(hipEventSynchronize(st->consumed[i]));
    *pinned_host = st->host[i];
    *slot = i;
    return CNN_AMD_OK;
}

int cnn_batch_stager_submit(void* stager, int slot, void** device_ptr) {
    Stager* st = static_cast<Stager*>(stager);
    CNN_REQUIRE(st && device_ptr && slot >= 0 && slot < st->depth, "cnn_batch_stager_submit: bad arguments");
    if (st->in_use[slot]) CNN_HIP_CHECK(hipStreamWaitEvent(st->copy, st->consumed[slot], 0));
    CNN_HIP_CHECK(hipMemcpyAsync(st->dev[slot], st->host[slot], st->bytes, hipMemcpyHostToDevice, st->copy));
    if (st->lut != nullptr) {  // bytes -> the fp32 planar batch, behind the copy on the same stream
        const int hw = st->H * st->W;
        if (hw % 4 == 0) {
            const int hw4 = hw / 4;
            const long long quads = (long long)st->B * hw4;
            u8hwc_to_f32chw<<<stream_grid((size_t)quads, 256), 256, 0, st->copy>>>((const unsigned*)st->dev[slot], (float*)st->dev_f32[slot], st->lut,
                                                                                   hw4, quads);
        } else {
            const long long px = (long long)st->B * hw;
            u8hwc_to_f32chw_px<<<stream_grid((size_t)px, 256), 256, 0, st->copy>>>((const unsigned char*)st->dev[slot], (float*)st->dev_f32[slot],
                                                                                   st->lut, hw, px);
        }
        CNN_LAUNCH_CHECK();
    }
    CNN_HIP_CHECK(hipEventRecord(st->uploaded[slot], st->copy));
    *device_ptr = st->lut != nullptr ? st->dev_f32[slot] : st->dev[slot];
    return CNN_AMD_OK;
}

int cnn_batch_stager_wait(void* stager, int slot, void* stream) {
    Stager* st = static_cast<Stager*>(stager);
    CNN_REQUIRE(st && slot >= 0 && slot < st->depth, "cnn_batch_stager_wait: bad arguments");
    CNN_HIP_CHECK(hipStreamWaitEvent(as_stream(stream), st->uploaded[slot], 0));
    publish_mark_stale(as_stream(stream));
    return CNN_AMD_OK;
}

int cnn_batch_stager_release(void* stager, int slot, void* stream) {
    Stager* st = static_cast<Stager*>(stager);
    CNN_REQUIRE(st && slot >= 0 && slot < st->depth, "cnn_batch_stager_release: bad arguments");
    CNN_HIP_CHECK(hipEventRecord(st->consumed[slot], as_stream(stream)));
    st->in_use[slot] = 1;
    return CNN_AMD_OK;
}

int cnn_amd_set_option(const char* name, const char* value) {
    CNN_REQUIRE(name != nullptr && name[0] != 0, "cnn_amd_set_option: empty name");
    if (strncmp(name, "CNN_AMD_", 8) == 0) name += 8;
    if (!kMeasureBuild && value != nullptr && measure_only_option(name))
        return fail(CNN_AMD_E_BADARG, "cnn_amd_set_option: %s is a measurement-only switch (it changes results or prints timings); it exists in "
                                      "libcnn_amd_measure.so (make -C cnn_amd/csrc measure), not in the product library", name);
    OptionTable& t = option_table();
    std::lock_guard<std::mutex> lk(t.mu);
    t.load_env_locked();
    if (value) t.values[name] = value;
    else t.values.erase(name);
    t.generation.fetch_add(1, std::memory_order_acq_rel);
    return CNN_AMD_OK;
}

int cnn_amd_measure_build(void) { return kMeasureBuild ? 1 : 0; }

int cnn_amd_get_option(const char* name, char* value_out, size_t cap) {
    CNN_REQUIRE(name != nullptr, "cnn_amd_get_option: null name");
    if (strncmp(name, "CNN_AMD_", 8) == 0) name += 8;
    OptionTable& t = option_table();
    std::lock_guard<std::mutex> lk(t.mu);
    t.load_env_locked();
    auto it = t.values.find(name);
    if (it == t.values.end()) return 1;
    if (value_out && cap) snprintf(value_out, cap, "%s", it->second.c_str());
    return CNN_AMD_OK;
}

int cnn_amd_publish_next_kernel(void* stream) {
    PublishState* p = nullptr;
    if (int rc = publish_events(p)) return rc;
    p->armed = true;
    p->stream = as_stream(stream);
    return CNN_AMD_OK;
}

int cnn_amd_published_is_last(void* stream) {
    const PublishState& p = publish_state();
    return (p.valid && !p.armed && !p.stale && p.stream == as_stream(stream)) ? 1 : 0;
}

int cnn_amd_wait_published(void* stream) {
    PublishState& p = publish_state();
    CNN_REQUIRE(p.valid && !p.armed, "cnn_amd_wait_published: %s", p.armed ? "the armed kernel has not been launched yet" : "nothing has been published");
    CNN_HIP_CHECK(hipStreamWaitEvent(as_stream(stream), p.ev[p.cur], 0));
    return CNN_AMD_OK;
}

}  // extern "C"
