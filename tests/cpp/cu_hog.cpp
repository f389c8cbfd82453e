// Test helper (tests/test_gpu_recovery.py): ANOTHER process that keeps compute units busy for a while -- what a second MD process, a profiler
// or a monitoring job on the same GPU does to a kernel whose blocks must be resident together (the one-launch step's in-kernel rendezvous).
//   cu_hog <blocks> <seconds> [<delay>]   prints "ready" once HIP is up, waits for a line on stdin (and <delay> seconds more), launches <blocks> blocks of 1024 threads with all the LDS
//                                 a block may have (160 KB) each (nothing else fits next to one of them on its CU) that spin for <seconds>, prints "launched",
//                                 and "done" when they have ended.
// Build: hipcc --offload-arch=gfx950 -O2 -o cu_hog cu_hog.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
__global__ void __launch_bounds__(1024) hog(long long ticks, unsigned* sink) {
    extern __shared__ char lds[];
    lds[threadIdx.x] = (char) threadIdx.x;
    const long long t0 = (long long) wall_clock64();                  // 100 MHz
    while ((long long) wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (threadIdx.x == 0 && lds[5] == 77) sink[0] = 1;
}
int main(int argc, char** argv) {
    const int blocks = argc > 1 ? std::atoi(argv[1]) : 64;
    const double seconds = argc > 2 ? std::atof(argv[2]) : 0.5;
    unsigned* sink = nullptr;
    if (hipMalloc((void**) &sink, 64) != hipSuccess) { std::printf("no device\n"); return 3; }
    // as much LDS as a block may have (160 KB on gfx950): nothing that needs LDS of its own fits next to such a block on its CU
    int lds = 0;
    for (int kb = 160; kb >= 64 && !lds; kb -= 2) {
        if (hipFuncSetAttribute((const void*) hog, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024) != hipSuccess) { (void) hipGetLastError(); continue; }
        hipLaunchKernelGGL(hog, 1, 1024, kb * 1024, 0, 1000LL, sink);           // (module load, first-launch set-up)
        if (hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess) lds = kb * 1024;
    }
    if (!lds) { std::printf("no LDS size accepted\n"); return 3; }
    std::printf("ready %d\n", lds / 1024); std::fflush(stdout);
    char line[64];
    if (!std::fgets(line, sizeof line, stdin)) return 2;
    if (argc > 3) usleep((useconds_t) (std::atof(argv[3]) * 1e6));
    hipLaunchKernelGGL(hog, blocks, 1024, lds, 0, (long long) (seconds * 1e8), sink);
    std::printf("launched\n"); std::fflush(stdout);
    const hipError_t e = hipDeviceSynchronize();
    std::printf(e == hipSuccess ? "done\n" : "failed\n"); std::fflush(stdout);
    return e == hipSuccess ? 0 : 1;
}
