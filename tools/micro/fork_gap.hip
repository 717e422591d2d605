// micro-benchmark (round 4): what a cross-stream FORK costs the producing stream when the producer has really written memory.
//   main: A (writes `mb` MB) -> [tell the side stream] -> B (dependent on A)        side: wait -> C (reads what A wrote, checks it)
// modes: 0 hipEventRecord between A and B | 1 event on A's dispatch packet (hipExtLaunchKernelGGL stopEvent) | 2 no fork at all
//        3 hipStreamWriteValue32 behind A + hipStreamWaitValue32 on the side stream
//        4 A's LAST workgroup writes the flag itself (device-scope fences), hipStreamWaitValue32 on the side stream: NOTHING between A and B
// prints the A-end -> B-start gap on the main stream, when C started, and whether C saw all of A's data.
// build: hipcc --offload-arch=gfx950 -O3 fork_gap.hip -o fork_gap ; run: ./fork_gap [MB written by A, default 24]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void produce(float* buf, size_t n, float value, unsigned long long* stamp, int slot, unsigned* counter, unsigned* flag, unsigned flag_value) {
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2 * slot] = wall_clock64();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = value;
    if (flag != nullptr) {  // mode 4: the last WAVE to finish publishes (no workgroup barrier: waves may leave a kernel at different points)
        __threadfence();
        if ((threadIdx.x & 63) == 0) {
            const unsigned done = atomicAdd(counter, 1u);
            if (done == gridDim.x * (blockDim.x / 64) - 1) {
                *counter = 0;
                stamp[2 * slot + 1] = wall_clock64();
                __threadfence();
                __hip_atomic_store(flag, flag_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    } else {
        __syncthreads();
        if (threadIdx.x == 0) atomicMax(&stamp[2 * slot + 1], wall_clock64());
    }
}
// mode 5: A writes its output with WRITE-THROUGH stores (agent-scope atomic stores: global_store ... sc1), every wave waits for its own
// stores (vmcnt(0)) and counts itself done; the last wave sets the flag.  No cache write-back anywhere inside the kernel.
__global__ void produce_wt(float* buf, size_t n, float value, unsigned long long* stamp, int slot, unsigned* counter, unsigned* flag, unsigned flag_value) {
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2 * slot] = wall_clock64();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __hip_atomic_store(&buf[i], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) {
        const unsigned done = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x * (blockDim.x / 64) - 1) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            stamp[2 * slot + 1] = wall_clock64();
            __hip_atomic_store(flag, flag_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void consume(const float* buf, size_t n, float value, unsigned long long* stamp, int slot, unsigned* bad) {
    if (threadIdx.x == 0) atomicMin(&stamp[2 * slot], wall_clock64());
    unsigned wrong = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) wrong += buf[i] != value;
    if (wrong) atomicAdd(bad, wrong);
    if (threadIdx.x == 0) atomicMax(&stamp[2 * slot + 1], wall_clock64());
}

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? (size_t)atoi(argv[1]) : 24;
    const size_t n = mb * 1024 * 1024 / 4;
    float* buf;
    unsigned long long* stamp;
    unsigned *bad, *counter, *flag;
    CK(hipMalloc(&buf, n * 4));
    CK(hipMalloc(&stamp, 64 * 8));
    CK(hipMalloc(&bad, 4));
    CK(hipMalloc(&counter, 4));
    CK(hipMemset(counter, 0, 4));
    CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
    CK(hipMemset(flag, 0, 8));
    hipStream_t main_s, side;
    CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const char* names[6] = {"hipEventRecord between A and B", "stopEvent on A's dispatch packet", "no fork", "hipStreamWriteValue32 + hipStreamWaitValue32",
                            "flag written by A's last wave (fences) + hipStreamWaitValue32", "write-through stores + flag by A's last wave + WaitValue32"};
    unsigned seq = 0;
    // event flavours for modes 0 / 1 (argv[2]): 0 DisableTiming | 1 + DisableSystemFence | 2 + ReleaseToDevice | 3 both
    const int flavour = argc > 2 ? atoi(argv[2]) : 0;
    if (flavour) {
        CK(hipEventDestroy(ev));
        CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | ((flavour & 1) ? hipEventDisableSystemFence : 0) |
                                            ((flavour & 2) ? hipEventReleaseToDevice : 0)));
        printf("event flavour %d\n", flavour);
    }
    for (int mode = 0; mode < (flavour ? 2 : 6); ++mode) {
        double gap = 0, cstart = 0, adur = 0;
        unsigned total_bad = 0;
        const int reps = 20;
        for (int r = 0; r < reps + 2; ++r) {
            ++seq;
            const float value = (float)(seq % 1000) + 0.5f;
            unsigned long long init[8] = {0, 0, ~0ull, 0, ~0ull, 0, 0, 0};
            CK(hipMemcpy(stamp, init, sizeof(init), hipMemcpyHostToDevice));
            CK(hipMemset(bad, 0, 4));
            CK(hipDeviceSynchronize());
            if (mode == 1)
                hipExtLaunchKernelGGL(produce, dim3(1024), dim3(256), 0, main_s, nullptr, ev, 0, buf, n, value, stamp, 0, counter, (unsigned*)nullptr, 0u);
            else if (mode == 4)
                produce<<<1024, 256, 0, main_s>>>(buf, n, value, stamp, 0, counter, flag, seq);
            else if (mode == 5)
                produce_wt<<<1024, 256, 0, main_s>>>(buf, n, value, stamp, 0, counter, flag, seq);
            else
                produce<<<1024, 256, 0, main_s>>>(buf, n, value, stamp, 0, counter, nullptr, 0u);
            if (mode == 0) CK(hipEventRecord(ev, main_s));
            if (mode == 3) CK(hipStreamWriteValue32(main_s, flag, seq, 0));
            if (mode == 0 || mode == 1) CK(hipStreamWaitEvent(side, ev, 0));
            if (mode == 3 || mode == 4 || mode == 5) CK(hipStreamWaitValue32(side, flag, seq, hipStreamWaitValueGte, 0xffffffffu));
            if (mode != 2) consume<<<256, 256, 0, side>>>(buf, n, value, stamp, 2, bad);
            consume<<<1024, 256, 0, main_s>>>(buf, n, value, stamp, 1, bad);  // B: depends on A through stream order
            CK(hipDeviceSynchronize());
            unsigned long long h[8];
            unsigned hb = 0;
            CK(hipMemcpy(h, stamp, sizeof(h), hipMemcpyDeviceToHost));
            CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            if (r >= 2) {
                gap += (double)((long long)h[2] - (long long)h[1]) / 100.0;
                adur += (double)((long long)h[1] - (long long)h[0]) / 100.0;
                if (mode != 2) cstart += (double)((long long)h[4] - (long long)h[1]) / 100.0;
                total_bad += hb;
            }
        }
        printf("%-62s A->B gap %6.2f us   C starts %6.2f us after A's end   A runs %6.2f us   stale reads %u\n", names[mode], gap / reps, cstart / reps,
               adur / reps, total_bad);
    }
    // the JOIN direction: main: A -> [wait for something on the side stream that finished long ago] -> B
    for (int jm = 0; jm < 3; ++jm) {
        double gap = 0;
        const int reps = 20;
        for (int r = 0; r < reps + 2; ++r) {
            ++seq;
            const float value = (float)(seq % 1000) + 0.5f;
            unsigned long long init[8] = {0, 0, ~0ull, 0, ~0ull, 0, 0, 0};
            CK(hipMemcpy(stamp, init, sizeof(init), hipMemcpyHostToDevice));
            // the side stream's work, finished before main starts
            produce<<<64, 256, 0, side>>>(buf, 1024, value, stamp, 3, counter, jm == 2 ? flag : nullptr, seq);
            if (jm == 1) CK(hipEventRecord(ev, side));
            CK(hipDeviceSynchronize());
            produce<<<1024, 256, 0, main_s>>>(buf, n, value, stamp, 0, counter, nullptr, 0u);
            if (jm == 1) CK(hipStreamWaitEvent(main_s, ev, 0));
            if (jm == 2) CK(hipStreamWaitValue32(main_s, flag, seq, hipStreamWaitValueGte, 0xffffffffu));
            consume<<<1024, 256, 0, main_s>>>(buf, n, value, stamp, 1, bad);
            CK(hipDeviceSynchronize());
            unsigned long long h[8];
            CK(hipMemcpy(h, stamp, sizeof(h), hipMemcpyDeviceToHost));
            if (r >= 2) gap += (double)((long long)h[2] - (long long)h[1]) / 100.0;
        }
        const char* jn[3] = {"join: nothing between A and B", "join: hipStreamWaitEvent (event long complete) between A and B", "join: hipStreamWaitValue32 (flag long set) between A and B"};
        printf("%-62s A->B gap %6.2f us\n", jn[jm], gap / reps);
    }
    return 0;
}
