// Probe (round 6): what does COLD instruction fetch cost a single wave that runs straight-line dependent fp64 code once per launch -- the
// situation of the one-launch step's thermostat wave between "all words held" and "scale factors released" (vv_device.inc: thermostat_tail)?
//   * body<K>: K dependent v_fma_f64 (8 bytes each: K / 8 instruction-cache lines of 64 B), straight line;
//   * the wave runs it `reps` times in a loop that is NOT unrolled (same addresses every pass): pass 0 is cold, passes 1.. are hot;
//   * `helper` variant: ANOTHER wave of the same block runs the body first while wave 0 sleeps -- is the instruction cache warm for wave 0 then?
//     (what a tile wave of the block could do for the thermostat wave while it waits for the release anyway);
//   * 256 blocks of 64 or 512 threads (the other waves parked at a barrier), one block per CU like the one-launch step.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o icache_probe icache_probe.cpp        Run: ./icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e)); return 3; } } while (0)

template <int K>
__device__ __forceinline__ double body(double x, double c, double d) {
#pragma unroll
    for (int i = 0; i < K; i++) x = __builtin_fma(x, c, d);      // K dependent fp64 FMAs on register operands: K 8-byte instructions, nothing else
    return x;
}
__device__ __forceinline__ long long now() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long t = (long long) __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return t;
}
// mode 0: wave 0 alone runs the passes; mode 1: wave 1 runs ONE pass of the very same code (same addresses: one loop, entered by both waves) while wave 0
// sleeps ~4000 clocks, then wave 0 runs its passes
template <int K>
__global__ void __launch_bounds__(512) k(double* out, long long* cyc, int reps, int mode, double c, double d) {
    const int w = threadIdx.x >> 6;
    double x = threadIdx.x * 1e-6;
    const bool runs = w == 0 || (w == 1 && mode == 1);
    if (runs) {
        if (w == 0 && mode == 1) for (int i = 0; i < 64; i++) __builtin_amdgcn_s_sleep(1);      // 64 x 64 clocks
        const int n = w == 0 ? reps : 1;
        long long t[6];
#pragma unroll
        for (int i = 0; i < 6; i++) t[i] = 0;
        t[0] = now();
#pragma clang loop unroll(disable)
        for (int r = 0; r < n; r++) {
            x = body<K>(x, c, d);
            asm volatile("" : "+v"(x));
            const long long tt = now();
            if (r == 0) t[1] = tt; else if (r == 1) t[2] = tt; else if (r == 2) t[3] = tt; else if (r == 3) t[4] = tt;
        }
        out[blockIdx.x * 1024 + threadIdx.x] = x;
        if (threadIdx.x == 0) for (int i = 0; i < 4; i++) cyc[blockIdx.x * 4 + i] = t[i + 1] - t[i];
    }
    if (blockDim.x > 64) __syncthreads();
}
template <int K>
static int run(double* out, long long* cyc) {
    for (int mode = 0; mode < 2; mode++)
        for (int blocks : {1, 256})
            for (int threads : {128, 512}) {
                for (int rep = 0; rep < 3; rep++) {      // the LAST launch is reported (the code is in L2 by then, as in a replayed graph)
                    hipLaunchKernelGGL(k<K>, blocks, threads, 0, 0, out, cyc, 4, mode, 0.999, 1.0e-3);
                    CK(hipDeviceSynchronize());
                }
                long long c[256 * 4]; CK(hipMemcpy(c, cyc, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost));
                long long mn[4] = {1 << 30, 1 << 30, 1 << 30, 1 << 30}, mx[4] = {0, 0, 0, 0}; double av[4] = {0, 0, 0, 0};
                for (int b = 0; b < blocks; b++) for (int i = 0; i < 4; i++) { mn[i] = c[b * 4 + i] < mn[i] ? c[b * 4 + i] : mn[i]; mx[i] = c[b * 4 + i] > mx[i] ? c[b * 4 + i] : mx[i]; av[i] += (double) c[b * 4 + i] / blocks; }
                std::printf("K = %4d dependent fp64 FMAs (%3d lines), %s, %3d blocks x %3d threads: clocks per pass, 4 passes in a row (avg [min..max] over blocks): "
                            "%.0f [%lld..%lld] | %.0f [%lld..%lld] | %.0f | %.0f\n", K, K / 8, mode ? "another wave of the block ran it first" : "wave 0 alone                          ",
                            blocks, threads, av[0], mn[0], mx[0], av[1], mn[1], mx[1], av[2], av[3]);
            }
    return 0;
}
int main() {
    double* out; long long* cyc;
    CK(hipMalloc(&out, sizeof(double) * 1024 * 256)); CK(hipMalloc(&cyc, sizeof(long long) * 256 * 4));
    if (run<64>(out, cyc)) return 3;
    if (run<128>(out, cyc)) return 3;
    if (run<256>(out, cyc)) return 3;
    if (run<512>(out, cyc)) return 3;
    return 0;
}
