// LD_PRELOAD shim: who calls hipGetDevice?  Histogram of return addresses (resolved with dladdr) printed at exit.
//   gcc -shared -fPIC -O2 -o /tmp/getdev_shim.so tools/probes/getdev_shim.c -ldl
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <execinfo.h>
typedef int (*fn_t)(int*);
static fn_t real;
static void* addr[4096][3];
static long cnt[4096];
static int n;
static void dump(void) {
    for (int i = 0; i < n; ++i) {
        if (cnt[i] < 50) continue;
        fprintf(stderr, "[getdev] %8ld calls:", cnt[i]);
        for (int k = 0; k < 3; ++k) {
            Dl_info di;
            if (addr[i][k] && dladdr(addr[i][k], &di) && di.dli_sname) fprintf(stderr, "  <- %s+%ld", di.dli_sname, (long)((char*)addr[i][k] - (char*)di.dli_saddr));
            else if (addr[i][k] && dladdr(addr[i][k], &di) && di.dli_fname) fprintf(stderr, "  <- %s@%lx", di.dli_fname, (long)((char*)addr[i][k] - (char*)di.dli_fbase));
        }
        fprintf(stderr, "\n");
    }
}
int hipGetDevice(int* d) {
    if (!real) {
        real = (fn_t)dlsym(RTLD_NEXT, "hipGetDevice");
        if (!real) {  // (the HIP runtime was dlopen()ed by somebody else: RTLD_NEXT does not see it)
            void* h = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("/opt/rocm/lib/libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
            real = h ? (fn_t)dlsym(h, "hipGetDevice") : 0;
        }
        if (!real) { fprintf(stderr, "[getdev] no hipGetDevice\n"); abort(); }
        atexit(dump);
    }
    void* bt[5];
    int m = backtrace(bt, 5);
    void* a0 = m > 1 ? bt[1] : 0, *a1 = m > 2 ? bt[2] : 0, *a2 = m > 3 ? bt[3] : 0;
    int i;
    for (i = 0; i < n; ++i) if (addr[i][0] == a0 && addr[i][1] == a1 && addr[i][2] == a2) break;
    if (i == n && n < 4096) { addr[i][0] = a0; addr[i][1] = a1; addr[i][2] = a2; cnt[i] = 0; ++n; }
    if (i < 4096) ++cnt[i];
    return real(d);
}
