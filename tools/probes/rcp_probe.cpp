// Probe: accuracy of v_rcp_f64 + n Newton steps against the IEEE quotient.  hipcc --offload-arch=gfx950 -O2 -o rcp_probe rcp_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* r0, double* r1, double* r2, double* q, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = x[i];
    double r = __builtin_amdgcn_rcp(a);
    r0[i] = r;
    r = fma(fma(-a, r, 1.0), r, r); r1[i] = r;
    r = fma(fma(-a, r, 1.0), r, r); r2[i] = r;
    q[i] = 1.0 / a;
}
int main() {
    const int n = 1 << 20;
    std::vector<double> h(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = std::ldexp(1.0 + (double) (s >> 11) / 9007199254740992.0, (int) (s % 41) - 20); }
    double *x, *r0, *r1, *r2, *q;
    hipMalloc(&x, n * 8); hipMalloc(&r0, n * 8); hipMalloc(&r1, n * 8); hipMalloc(&r2, n * 8); hipMalloc(&q, n * 8);
    hipMemcpy(x, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, n / 256, 256, 0, 0, x, r0, r1, r2, q, n);
    std::vector<double> a0(n), a1(n), a2(n), aq(n);
    hipMemcpy(a0.data(), r0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(a1.data(), r1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(a2.data(), r2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(aq.data(), q, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0, eq = 0;
    for (int i = 0; i < n; i++) {
        const double t = 1.0 / h[i];
        e0 = std::fmax(e0, std::fabs(a0[i] - t) / t); e1 = std::fmax(e1, std::fabs(a1[i] - t) / t);
        e2 = std::fmax(e2, std::fabs(a2[i] - t) / t); eq = std::fmax(eq, std::fabs(aq[i] - t) / t);
    }
    std::printf("max relative error vs host 1/x: rcp %.3e | +1 Newton %.3e | +2 Newton %.3e | device 1.0/x %.3e   (ulp = 1.1e-16)\n", e0, e1, e2, eq);
    return 0;
}
