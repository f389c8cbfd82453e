// Probe: which shader clock do SHORT kernels see?  Every wave runs n dependent fp64 FMAs and records s_memtime ticks and the 100 MHz
// wall clock around them; the kernel is launched back to back (reps launches) and the LAST launch is reported, for a long kernel
// (one launch of ~1 ms) and for ~3 us kernels in a stream of thousands, with 1 wave on the chip and with 256 blocks x 8 waves.
// Build: hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e)); return 3; } } while (0)
__global__ void k(double* out, long long* rec, int n, double a, double b) {
    double x = 1.0 + threadIdx.x * 1e-9;
    const long long t0 = __builtin_readcyclecounter();
    const long long w0 = wall_clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) x = fma(x, a, b);
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long w1 = wall_clock64();
    if (x == 123.0) out[0] = x;
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) { rec[0] = t1 - t0; rec[1] = w1 - w0; }
}
int main() {
    double* out; long long* rec;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&rec, 16));
    hipStream_t s; CK(hipStreamCreate(&s));
    struct Case { const char* name; int blocks, threads, n, reps; } cases[] = {
        {"1 wave, one long launch (~1.3 ms)", 1, 64, 32768, 1},
        {"256x512, one long launch", 256, 512, 32768, 1},
        {"1 wave, 3 us kernels x 20000", 1, 64, 64, 20000},
        {"256x512, 3 us kernels x 20000 (60 ms busy)", 256, 512, 64, 20000},
        {"256x512, 3 us kernels x 200000 (0.6 s busy)", 256, 512, 64, 200000},
        {"256x64, 3 us kernels x 200000", 256, 64, 64, 200000},
    };
    for (auto& c : cases) {
        CK(hipStreamSynchronize(s));
        auto h0 = std::chrono::steady_clock::now();
        for (int r = 0; r < c.reps; r++) hipLaunchKernelGGL(k, c.blocks, c.threads, 0, s, out, rec, c.n, 0.999999, 1e-7);
        CK(hipStreamSynchronize(s));
        const double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count() / c.reps;
        long long v[2]; CK(hipMemcpy(v, rec, 16, hipMemcpyDeviceToHost));
        std::printf("%-48s: %8lld ticks in %8.2f us of the wall clock -> %.2f ticks/ns; %5.2f ticks per fma; host: %.2f us per launch\n", c.name, v[0], v[1] * 0.01,
                    (double) v[0] / (v[1] * 10.0), (double) v[0] / (16.0 * c.n), host_us);
    }
    return 0;
}
