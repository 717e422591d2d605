// micro-benchmark: bubble that a cross-stream dependency leaves on the PRODUCING stream.
//   main: A -> [publish A's completion to the side stream] -> B         side: wait -> C
// (1) hipEventRecord between A and B   (2) the event attached to A's dispatch packet (hipExtLaunchKernelGGL stopEvent)   (3) no side stream
// build: hipcc --offload-arch=gfx950 -O3 event_gap.hip -o event_gap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(unsigned long long* stamp, int slot, long long ticks) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2 * slot] = t0;
    while ((long long)(wall_clock64() - t0) < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) stamp[2 * slot + 1] = wall_clock64();
}
int main() {
    unsigned long long* stamp;
    CK(hipMalloc(&stamp, 64 * 8));
    hipStream_t main_s, side;
    CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const long long ticks = 3000;  // 100 MHz wall clock: 30 us
    for (int mode = 0; mode < 4; ++mode) {
        double gap = 0;
        const int reps = 20;
        for (int r = 0; r < reps + 2; ++r) {
            if (mode == 1) {
                hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, main_s, nullptr, ev, 0, stamp, 0, ticks);
            } else {
                spin<<<256, 256, 0, main_s>>>(stamp, 0, ticks);
                if (mode == 0 || mode == 3) CK(hipEventRecord(ev, main_s));
            }
            if (mode != 2) {
                CK(hipStreamWaitEvent(side, ev, 0));
                spin<<<64, 256, 0, side>>>(stamp, 2, ticks);
            }
            spin<<<256, 256, 0, main_s>>>(stamp, 1, ticks);
            if (mode == 3) {  // + join back
                CK(hipEventRecord(ev, side));
                CK(hipStreamWaitEvent(main_s, ev, 0));
                spin<<<256, 256, 0, main_s>>>(stamp, 3, ticks);
            }
            CK(hipDeviceSynchronize());
            unsigned long long h[8];
            CK(hipMemcpy(h, stamp, sizeof(h), hipMemcpyDeviceToHost));
            if (r >= 2) gap += (double)(h[2] - h[1]) / 100.0;  // us between A's end and B's start
            if (r == reps + 1 && mode != 2) printf("   (C started %.1f us after A ended)\n", (double)((long long)h[4] - (long long)h[1]) / 100.0);
            if (r == reps + 1 && mode == 3) printf("   (D started %.1f us after max(B, C) ended)\n", (double)((long long)h[6] - (long long)(h[3] > h[5] ? h[3] : h[5])) / 100.0);
        }
        const char* names[4] = {"hipEventRecord between A and B", "stopEvent on A's dispatch (hipExtLaunchKernelGGL)", "no cross-stream dependency", "record + join back"};
        printf("%-55s A->B gap %.2f us\n", names[mode], gap / reps);
    }
    return 0;
}
