// Probe: latency of dependent fp64 operations for ONE wave on an otherwise idle SIMD (what the thermostat wave of kernel B is).
// Build: hipcc --offload-arch=gfx950 -O3 -o dpchain_probe dpchain_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e)); return 3; } } while (0)
template <int MODE>
__global__ void k(double* out, long long* cyc, double x0, int n) {
    double x = x0 + threadIdx.x * 1e-9, y = x0 * 0.5, z = x0 * 0.25;
    const long long t0 = __builtin_readcyclecounter();
    const long long w0 = wall_clock64();
    for (int i = 0; i < n; i += 32) {
#pragma unroll
      for (int u = 0; u < 32; u++) {
        if (MODE == 0) { x = fma(x, 0.999999, 1e-7); }                                    // 1 dependent FMA per iteration
        if (MODE == 1) { x = fma(x, 0.999999, 1e-7); y = fma(y, 0.999998, 1e-7); z = fma(z, 0.999997, 1e-7); }   // 3 independent chains
        if (MODE == 2) { x = x * 0.999999; x = x + 1e-7; }                                 // mul then add (no contraction)
        if (MODE == 3) { float f = (float) x; f = fmaf(f, 0.999999f, 1e-7f); x = f; }      // cvt + f32 fma + cvt
      }
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long w1 = wall_clock64();
    out[threadIdx.x] = x + y + z;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
int main() {
    double* out; long long* cyc;
    CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&cyc, 16));
    const int n = 4096;
    const char* names[4] = {"1 dependent fma_f64", "3 independent fma_f64 chains", "dependent mul_f64 + add_f64", "cvt + fma_f32 + cvt"};
    for (int mode = 0; mode < 4; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, 1, 64, 0, 0, out, cyc, 1.0, n);
            if (mode == 1) hipLaunchKernelGGL(k<1>, 1, 64, 0, 0, out, cyc, 1.0, n);
            if (mode == 2) hipLaunchKernelGGL(k<2>, 1, 64, 0, 0, out, cyc, 1.0, n);
            if (mode == 3) hipLaunchKernelGGL(k<3>, 1, 64, 0, 0, out, cyc, 1.0, n);
            CK(hipDeviceSynchronize());
        }
        long long c[2]; CK(hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost));
        std::printf("%-30s: %6.1f counter ticks / iteration, %6.2f ns / iteration (100 MHz wall clock) -> counter runs at %.2f GHz\n", names[mode], (double) c[0] / n, (double) c[1] * 10.0 / n, (double) c[0] / ((double) c[1] * 10.0));
    }
    return 0;
}
