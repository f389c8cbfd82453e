// Latency / issue rate of fp64 VALU operations for ONE wave on a CU (the situation of the in-kernel constraint solver and of the
// thermostat wave): dependent chains vs independent streams of v_fma_f64 / v_mul_f64 / v_add_f64, the IEEE division, v_rcp_f64,
// and an LDS round trip.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o dp_latency_probe dp_latency_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 512
template <int MODE>
__global__ void probe(double* out, long long* cyc, double seed) {
    double a = seed + threadIdx.x * 1e-9, b = seed * 0.5, c = 1.0000001, d = seed * 0.25, e = seed * 0.125;
    __shared__ double lds[64];
    lds[threadIdx.x] = a;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; i++) {
        if (MODE == 0) a = fma(a, c, b);                                        // dependent fma
        if (MODE == 1) a = a * c;                                               // dependent mul
        if (MODE == 2) a = a + b;                                               // dependent add
        if (MODE == 3) { a = fma(a, c, b); b = fma(b, c, d); d = fma(d, c, e); e = fma(e, c, a * 0); }   // 4 streams (loosely coupled)
        if (MODE == 4) a = 1.0 / (a + 3.0);                                     // dependent IEEE division
        if (MODE == 5) a = __builtin_amdgcn_rcp(a + 3.0);                       // dependent rcp
        if (MODE == 6) { lds[threadIdx.x] = a; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); a = lds[(threadIdx.x + 1) & 63] + 1.0; }   // LDS round trip + 1 add
        if (MODE == 7) { a = a * c; a = a + b; }                                // mul then add (no contraction)
    }
    const long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a + b + d + e;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    const char* names[8] = {"dependent v_fma_f64", "dependent v_mul_f64", "dependent v_add_f64", "4 independent fma streams (per fma)", "dependent IEEE 1/x (+add)", "dependent v_rcp_f64 (+add)", "LDS write -> read neighbour (+add)", "dependent mul+add pair"};
    for (int rep = 0; rep < 2; rep++)
    for (int m = 0; m < 8; m++) {
        switch (m) {
            case 0: hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5); break;
            case 1: hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5); break;
            case 2: hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5); break;
            case 3: hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5); break;
            case 4: hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5); break;
            case 5: hipLaunchKernelGGL(probe<5>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5); break;
            case 6: hipLaunchKernelGGL(probe<6>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5); break;
            default: hipLaunchKernelGGL(probe<7>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5); break;
        }
        long long h = 0; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        const double per = (double) h / N / (m == 3 ? 5.0 : 1.0);
        if (rep) printf("%-40s %7.1f shader-clock cycles per iteration\n", names[m], per);
    }
    return 0;
}
