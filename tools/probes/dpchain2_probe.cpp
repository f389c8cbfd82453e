// Probe: dependent-op latency of v_mul_f64 / v_add_f64 / v_fma_f64 and of an NH-chain-shaped update for ONE wave, alone on its CU and
// with seven waves of the same block parked at a barrier (what the thermostat wave of kernel B sees).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o dpchain2_probe dpchain2_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e)); return 3; } } while (0)
__device__ __forceinline__ double ex7(double x) {
    const double x2 = x * x;
    const double p01 = x + 1.0, p23 = fma(x, 1.0 / 6.0, 0.5), p45 = fma(x, 1.0 / 120.0, 1.0 / 24.0), p67 = fma(x, 1.0 / 5040.0, 1.0 / 720.0);
    const double x4 = x2 * x2, q0 = fma(x2, p23, p01), q1 = fma(x2, p67, p45);
    return fma(x4, q1, q0);
}
template <int MODE>
__global__ void k(double* out, long long* cyc, double x0, int n, double a, double b) {
    if (threadIdx.x >= 64) { __syncthreads(); return; }
    double x = x0 + threadIdx.x * 1e-9;
    double ed0 = 1e-3 * x, ed1 = 2e-3 * x, ed2 = 3e-3 * x, edd0 = 1e-2, edd1 = 2e-2, edd2 = 3e-2, factor = 1.0;
    const double dt8 = 1.25e-4, dt4 = 2.5e-4, dt2 = 5e-4, ke2 = 1000.0 * x, tgt = 999.0, im = 1e-3, kT = 2.7, m0 = 10, m1 = 11;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 32; u++) x = x * a;
        }
        if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 32; u++) x = x + b;
        }
        if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 32; u++) x = fma(x, a, b);
        }
        if (MODE == 3) {      // one NH-chain update of three links (shape of propagate_preloaded<3>, tail link zero)
            double e;
            e = 1.0; ed2 *= e; ed2 += edd2 * dt4; ed2 *= e;
            e = ex7(-dt8 * ed2); ed1 *= e; ed1 += edd1 * dt4; ed1 *= e;
            e = ex7(-dt8 * ed1); ed0 *= e; ed0 += edd0 * dt4; ed0 *= e;
            factor *= ex7(-dt2 * ed0);
            edd0 = (ke2 * factor * factor - tgt) * im;
            ed0 *= e; ed0 += edd0 * dt4; ed0 *= e;
            e = ex7(-dt8 * ed2); ed1 *= e; edd1 = (m0 * ed0 * ed0 - kT) * im; ed1 += edd1 * dt4; ed1 *= e;
            e = 1.0; ed2 *= e; edd2 = (m1 * ed1 * ed1 - kT) * im; ed2 += edd2 * dt4; ed2 *= e;
            x = factor;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x + ed0 + ed1 + ed2;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    if (blockDim.x > 64) __syncthreads();
}
int main() {
    double* out; long long* cyc;
    CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&cyc, 16));
    const char* names[4] = {"32 dependent mul_f64", "32 dependent add_f64", "32 dependent fma_f64", "one 3-link chain update"};
    for (int threads : {64, 512})
    for (int mode = 0; mode < 4; mode++) {
        const int n = mode == 3 ? 1 : 128;
        for (int rep = 0; rep < 3; rep++) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, 1, threads, 0, 0, out, cyc, 1.0, n, 0.999999, 1e-7);
            if (mode == 1) hipLaunchKernelGGL(k<1>, 1, threads, 0, 0, out, cyc, 1.0, n, 0.999999, 1e-7);
            if (mode == 2) hipLaunchKernelGGL(k<2>, 1, threads, 0, 0, out, cyc, 1.0, n, 0.999999, 1e-7);
            if (mode == 3) hipLaunchKernelGGL(k<3>, 1, threads, 0, 0, out, cyc, 1.0, n, 0.999999, 1e-7);
            CK(hipDeviceSynchronize());
        }
        long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        std::printf("%d threads  %-26s: %8.1f ticks per %s\n", threads, names[mode], (double) c / n / (mode == 3 ? 1 : 32), mode == 3 ? "update (single cold pass)" : "op");
    }
    // the chain update repeated (warm)
    for (int threads : {64, 512}) {
        hipLaunchKernelGGL(k<3>, 1, threads, 0, 0, out, cyc, 1.0, 64, 0.999999, 1e-7);
        CK(hipDeviceSynchronize());
        long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        std::printf("%d threads  chain update, 64 in a row: %8.1f ticks per update\n", threads, (double) c / 64);
    }
    return 0;
}
