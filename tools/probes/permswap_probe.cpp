// What v_permlane32_swap / v_permlane16_swap (new in gfx950) do to two registers, lane by lane, and the four-quantity wave reduction
// built on them (vv_device.inc: wave_reduce4): totals of a, b, c, d in lanes 15, 31, 47, 63.
// Build: hipcc --offload-arch=gfx950 -O3 -o permswap_probe permswap_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[128 + threadIdx.x] = q[0]; out[192 + threadIdx.x] = q[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[4] = {"swap32 vdst'", "swap32 src0'", "swap16 vdst'", "swap16 src0'"};
    for (int j = 0; j < 4; j++) { std::printf("%s:", names[j]); for (int i = 0; i < 64; i += 8) std::printf(" [%d]=%u", i, h[64 * j + i]); std::printf("\n"); }
    return 0;
}
