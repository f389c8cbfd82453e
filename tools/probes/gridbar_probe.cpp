// Probe: what does an in-kernel "all blocks have added their partial sums" rendezvous cost on MI355X, compared with
// ending the kernel and starting the next one?  Fused: every block adds {count:16 | value:48} words into S slots per quantity with
// fire-and-forget atomics; one wave per block then polls the slots (agent-scope loads) until the counts add up to the grid size --
// at which moment it also holds the totals.  Split: kernel 1 ends after the atomics, kernel 2 folds the slots and reloads its data.
// Build: hipcc --offload-arch=gfx950 -O2 -o gridbar_probe gridbar_probe.cpp ; run: ./gridbar_probe [blocks] [slots]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e)); std::exit(3); } } while (0)
typedef unsigned long long u64;

__device__ __forceinline__ double wave_sum(double x) { for (int o = 32; o; o >>= 1) x += __shfl_xor(x, o, 64); return x; }

template <bool FUSED>
__global__ void __launch_bounds__(320) k1(const double4* in, double4* out, u64* acc, u64* acc_next, int S, int n, unsigned* status) {
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    __shared__ double red[8][3];
    __shared__ double sh_factor;
    const bool poll_wave = FUSED && wib == nw - 1;
    const int tile = blockIdx.x * (FUSED ? nw - 1 : nw) + wib;
    const int i = tile * 64 + lane;
    double4 v = {0, 0, 0, 0};
    if (!poll_wave && i < n) v = in[i];
    double q[3] = {v.x * v.x, v.y * v.y, v.z * v.z};
    for (int k = 0; k < 3; k++) { double s = wave_sum(q[k]); if (lane == 0) red[wib][k] = poll_wave ? 0 : s; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double s = 0;
        for (int w = 0; w < nw; w++) s += red[w][threadIdx.x];
        const u64 word = (1ull << 48) | ((u64) (long long) (s * 1024.0) & ((1ull << 48) - 1));
        atomicAdd(&acc[threadIdx.x * S + (blockIdx.x % S)], word);
    }
    if (!FUSED) { if (!poll_wave && i < n) out[i] = v; return; }
    if (poll_wave) {
        const long long t0 = wall_clock64();
        u64 tot[3];
        for (;;) {
            bool done = true;
            for (int k = 0; k < 3; k++) {
                u64 s = 0;
                for (int j = lane; j < S; j += 64) s += __hip_atomic_load(&acc[k * S + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
                tot[k] = s;
                done = done && (s >> 48) == (u64) gridDim.x;
            }
            if (done) break;
            if (wall_clock64() - t0 > 100000000LL) { status[0] = 1; break; }
        }
        if (lane == 0) sh_factor = 1.0 + 1e-9 * (double) ((tot[0] + tot[1] + tot[2]) & 0xFFFF);
        if (blockIdx.x == 0) for (int j = lane; j < 3 * S; j += 64) acc_next[j] = 0;
    }
    __syncthreads();
    if (!poll_wave && i < n) { const double f = sh_factor; v.x *= f; v.y *= f; v.z *= f; out[i] = v; }
}

__global__ void __launch_bounds__(320) k2(const double4* in, double4* out, const u64* acc, u64* acc_next, int S, int n) {
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    __shared__ double sh_factor;
    const bool fold_wave = wib == nw - 1;
    const int i = (blockIdx.x * (nw - 1) + wib) * 64 + lane;
    double4 v = {0, 0, 0, 0};
    if (!fold_wave && i < n) v = in[i];
    if (fold_wave) {
        u64 t = 0;
        for (int k = 0; k < 3; k++) { u64 s = 0; for (int j = lane; j < S; j += 64) s += acc[k * S + j]; for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64); t += s; }
        if (lane == 0) sh_factor = 1.0 + 1e-9 * (double) (t & 0xFFFF);
        if (blockIdx.x == 0) for (int j = lane; j < 3 * S; j += 64) acc_next[j] = 0;
    }
    __syncthreads();
    if (!fold_wave && i < n) { const double f = sh_factor; v.x *= f; v.y *= f; v.z *= f; out[i] = v; }
}

int main(int argc, char** argv) {
    const int tiles = argc > 1 ? std::atoi(argv[1]) : 1752, S = argc > 2 ? std::atoi(argv[2]) : 64;
    const int n = tiles * 64;
    double4 *a, *b; u64* acc; unsigned* status;
    CK(hipMalloc(&a, n * sizeof(double4))); CK(hipMalloc(&b, n * sizeof(double4)));
    CK(hipMalloc(&acc, 2 * 3 * S * sizeof(u64))); CK(hipMemset(acc, 0, 2 * 3 * S * sizeof(u64)));
    CK(hipMalloc(&status, 4)); CK(hipMemset(status, 0, 4));
    std::vector<double4> h(n, double4{0.5, 0.25, 0.125, 1.0});
    CK(hipMemcpy(a, h.data(), n * sizeof(double4), hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int gridF = (tiles + 3) / 4, gridS1 = (tiles + 4) / 5;
    for (int mode = 0; mode < 2; mode++) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int it = 0; it < 100; it++) {
            u64* cur = acc + (it & 1) * 3 * S; u64* nxt = acc + ((it & 1) ^ 1) * 3 * S;
            if (mode == 0) hipLaunchKernelGGL(k1<true>, gridF, 320, 0, s, (it & 1) ? b : a, (it & 1) ? a : b, cur, nxt, S, n, status);
            else { hipLaunchKernelGGL(k1<false>, gridS1, 320, 0, s, (it & 1) ? b : a, (it & 1) ? a : b, cur, nxt, S, n, status);
                   hipLaunchKernelGGL(k2, gridF, 320, 0, s, (it & 1) ? a : b, (it & 1) ? b : a, cur, nxt, S, n); }
        }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 3; w++) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < 20; r++) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned st; CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
        std::printf("%s: tiles %d slots %d: %.2f us per iteration%s\n", mode == 0 ? "fused (in-kernel rendezvous)" : "split (two kernels)      ", tiles, S, ms * 1e3 / 2000, st ? "  [TIMEOUT]" : "");
    }
    return 0;
}
