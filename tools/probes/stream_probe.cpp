// Probe: what bandwidth do the access shapes of the fused kernels reach at 8.9 M particles, without any arithmetic?
//   copy16      float4 in -> float4 out, one element per lane per iteration (the guide's 6.3 TB/s reference shape)
//   copy32      double4 (32 B per lane) in -> out                                   (velm)
//   copy32_idx  the same through an int2 slot table (identity), i.e. a dependent load in front of the data  (our layout)
//   copy32_x2   32 B per lane, two tiles per wave in flight                         (more bytes in flight per wave)
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e)); std::exit(3); } } while (0)

__global__ void __launch_bounds__(256) copy16(const float4* in, float4* out, size_t n) {
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void __launch_bounds__(256) copy32(const double4* in, double4* out, size_t n) {
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void __launch_bounds__(256) copy32_idx(const double4* in, double4* out, const int2* slots, size_t n) {
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) { const int a = slots[i].x; out[a] = in[a]; }
}
__global__ void __launch_bounds__(256) copy32_idx_x2(const double4* in, double4* out, const int2* slots, size_t n) {
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += 2 * stride) {
        const size_t j = i + stride;
        const int a = slots[i].x, b = j < n ? slots[j].x : -1;
        const double4 va = in[a];
        double4 vb = {0, 0, 0, 0};
        if (b >= 0) vb = in[b];
        out[a] = va;
        if (b >= 0) out[b] = vb;
    }
}
__global__ void __launch_bounds__(256) kickish(const double4* velm, double4* vout, const long long* force, const int2* slots, size_t n, size_t padded) {
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        const int a = slots[i].x;
        double4 v = velm[a];
        v.x += 1e-12 * v.w * force[a]; v.y += 1e-12 * v.w * force[a + padded]; v.z += 1e-12 * v.w * force[a + 2 * padded];
        vout[a] = v;
    }
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? std::atol(argv[1]) : 8880000;
    const int grid = argc > 2 ? std::atoi(argv[2]) : 2048;
    double4 *a, *b; int2* slots; long long* force;
    CK(hipMalloc(&a, n * 32)); CK(hipMalloc(&b, n * 32)); CK(hipMalloc(&slots, n * 8)); CK(hipMalloc(&force, n * 24));
    CK(hipMemset(a, 0, n * 32)); CK(hipMemset(b, 0, n * 32)); CK(hipMemset(force, 0, n * 24));
    std::vector<int2> h(n); for (size_t i = 0; i < n; i++) h[i] = int2{(int) i, 0};
    CK(hipMemcpy(slots, h.data(), n * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, double bytes, auto launch) {
        for (int w = 0; w < 3; w++) launch();
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 20; r++) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%-14s grid %5d: %8.1f us  %6.2f TB/s\n", name, grid, ms * 1e3 / 20, bytes / (ms * 1e-3 / 20) / 1e12);
    };
    run("copy16", 2.0 * n * 32, [&] { hipLaunchKernelGGL(copy16, grid, 256, 0, 0, (const float4*) a, (float4*) b, 2 * n); });
    run("copy32", 2.0 * n * 32, [&] { hipLaunchKernelGGL(copy32, grid, 256, 0, 0, a, b, n); });
    run("copy32_idx", n * 72.0, [&] { hipLaunchKernelGGL(copy32_idx, grid, 256, 0, 0, a, b, slots, n); });
    run("copy32_idx_x2", n * 72.0, [&] { hipLaunchKernelGGL(copy32_idx_x2, grid, 256, 0, 0, a, b, slots, n); });
    run("kickish", n * 96.0, [&] { hipLaunchKernelGGL(kickish, grid, 256, 0, 0, a, b, force, slots, n, n); });
    return 0;
}
