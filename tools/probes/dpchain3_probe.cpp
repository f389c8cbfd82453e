// Probe: the thermostat wave's chain code (copied from vv_kernels.hip: chain_exp_small / propagate_preloaded<3, true>) timed on its own,
// with and without s_setprio 3, cold and warm.   Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o dpchain3_probe dpchain3_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e)); return 3; } } while (0)
struct ChainLaneBlock { double eta_mass[4], inv_eta_mass[4]; double nkbt, kT, acc_inv_scale, active; double dt2, dt4, dt8, pad_; };
struct ChainRegs { double eta[4], eta_dot[5], eta_dotdot[4]; };
__device__ __forceinline__ double chain_exp_small(double x, unsigned& max_hi) {
    const unsigned hi = (unsigned) __double2hiint(x) & 0x7FFFFFFFu;
    max_hi = hi > max_hi ? hi : max_hi;
    const double x2 = x * x;
    const double p01 = x + 1.0, p23 = fma(x, 1.0 / 6.0, 0.5), p45 = fma(x, 1.0 / 120.0, 1.0 / 24.0), p67 = fma(x, 1.0 / 5040.0, 1.0 / 720.0);
    const double x4 = x2 * x2, q0 = fma(x2, p23, p01), q1 = fma(x2, p67, p45);
    return fma(x4, q1, q0);
}
template <int NC>
__device__ __forceinline__ double propagate(int loops, const ChainLaneBlock& lc, double ke2, ChainRegs& r, unsigned& max_hi) {
    auto ex = [&](double x) { return chain_exp_small(x, max_hi); };
    const bool tail_zero = !__any(r.eta_dot[NC] != 0);
    double factor = 1.0;
    if (lc.active == 0) return factor;
    const double ke2_target = lc.nkbt;
    double expfac = 1.0;
    const double dt2 = lc.dt2, dt4 = lc.dt4, dt8 = lc.dt8, kT = lc.kT;
    r.eta_dotdot[0] = (ke2 - ke2_target) * lc.inv_eta_mass[0];
    for (int iloop = 0; iloop < loops; iloop++) {
#pragma unroll
        for (int ich = NC - 1; ich >= 0; ich--) {
            expfac = (tail_zero && ich == NC - 1) ? 1.0 : ex(-dt8 * r.eta_dot[ich + 1]);
            r.eta_dot[ich] *= expfac; r.eta_dot[ich] += r.eta_dotdot[ich] * dt4; r.eta_dot[ich] *= expfac;
        }
        factor *= ex(-dt2 * r.eta_dot[0]);
#pragma unroll
        for (int ich = 0; ich < NC; ich++) r.eta[ich] += dt2 * r.eta_dot[ich];
        r.eta_dotdot[0] = (ke2 * factor * factor - ke2_target) * lc.inv_eta_mass[0];
        r.eta_dot[0] *= expfac; r.eta_dot[0] += r.eta_dotdot[0] * dt4; r.eta_dot[0] *= expfac;
#pragma unroll
        for (int ich = 1; ich < NC; ich++) {
            expfac = (tail_zero && ich == NC - 1) ? 1.0 : ex(-dt8 * r.eta_dot[ich + 1]);
            r.eta_dot[ich] *= expfac;
            r.eta_dotdot[ich] = (lc.eta_mass[ich - 1] * r.eta_dot[ich - 1] * r.eta_dot[ich - 1] - kT) * lc.inv_eta_mass[ich];
            r.eta_dot[ich] += r.eta_dotdot[ich] * dt4; r.eta_dot[ich] *= expfac;
        }
    }
    return factor;
}
template <int PRIO>
__global__ void k(double* out, long long* cyc, const ChainLaneBlock* lcp, const ChainRegs* st, int loops, int reps, double ke2, const double4* big, int nbig, double4* sink) {
    if (threadIdx.x >= 64) {
        if (big) {        // tile-wave stand-in: a few 32-byte loads per lane from a large array, some fp64 work on them, a store; then park
            const size_t i = ((size_t) blockIdx.x * blockDim.x + threadIdx.x) % (size_t) nbig;
            double4 a = big[i], b = big[(i * 7 + 13) % nbig], c = big[(i * 3 + 5) % nbig];
            double x = a.x * b.y + c.z, y = a.y * b.z + c.x;
            for (int u = 0; u < 40; u++) { x = x * 0.999 + y; y = y * 1.001 - x; }
            sink[i] = double4{x, y, a.w, b.w};
        }
        __syncthreads();
        return;
    }
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    const int cg = threadIdx.x < 3 ? threadIdx.x : 2;
    ChainRegs cr = st[cg];
    const ChainLaneBlock lc = lcp[cg];
    double f = 0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    long long t[5];
    t[0] = __builtin_readcyclecounter();
    for (int i = 0; i < reps; i++) {
        unsigned mh = 0;
        f += propagate<3>(loops, lc, ke2 + i, cr, mh);
        if (__any(mh > 0x3F900000u)) f += 1;
        if (i < 4) t[i + 1] = __builtin_readcyclecounter();
    }
    out[threadIdx.x] = f + cr.eta[0] + cr.eta_dot[1] + cr.eta_dot[2];
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) for (int i = 0; i < 4; i++) cyc[i] = t[i + 1] - t[i];
    if (blockDim.x > 64) __syncthreads();
}
int main() {
    double* out; long long* cyc; ChainLaneBlock* lc; ChainRegs* st; double4* big; double4* sink; const int nbig = 1 << 20;
    CK(hipMalloc(&big, sizeof(double4) * nbig)); CK(hipMalloc(&sink, sizeof(double4) * nbig)); CK(hipMemset(big, 0, sizeof(double4) * nbig));
    CK(hipMalloc(&out, 64 * 8)); CK(hipMalloc(&cyc, 64)); CK(hipMalloc(&lc, 3 * sizeof(ChainLaneBlock))); CK(hipMalloc(&st, 3 * sizeof(ChainRegs)));
    ChainLaneBlock h[3] = {}; ChainRegs s[3] = {};
    for (int g = 0; g < 3; g++) {
        for (int i = 0; i < 4; i++) { h[g].eta_mass[i] = i ? 0.0277 : 4570.0; h[g].inv_eta_mass[i] = 1.0 / h[g].eta_mass[i]; s[g].eta_dot[i] = 0.01 * (i + 1); s[g].eta_dotdot[i] = 0.1; }
        h[g].nkbt = 457000; h[g].kT = 2.77; h[g].active = 1; h[g].dt2 = 5e-4; h[g].dt4 = 2.5e-4; h[g].dt8 = 1.25e-4;
        s[g].eta_dot[3] = 0; s[g].eta_dot[4] = 0;
    }
    CK(hipMemcpy(lc, h, sizeof(h), hipMemcpyHostToDevice)); CK(hipMemcpy(st, s, sizeof(s), hipMemcpyHostToDevice));
    for (int withmem = 0; withmem < 2; withmem++)
    for (int blocks : {1, 256})
    for (int prio = 0; prio < 2; prio++)
        for (int threads : {64, 512}) {
            for (int rep = 0; rep < 3; rep++) {
                if (prio) hipLaunchKernelGGL(k<1>, blocks, threads, 0, 0, out, cyc, lc, st, 1, 4, 457100.0, withmem ? big : nullptr, nbig, sink);
                else hipLaunchKernelGGL(k<0>, blocks, threads, 0, 0, out, cyc, lc, st, 1, 4, 457100.0, withmem ? big : nullptr, nbig, sink);
                CK(hipDeviceSynchronize());
            }
            long long c[4]; CK(hipMemcpy(c, cyc, 32, hipMemcpyDeviceToHost));
            std::printf("%s, %4d blocks, setprio %d, %3d threads: chain update ticks, 4 in a row: %lld %lld %lld %lld\n", withmem ? "other waves load + compute" : "other waves idle", blocks, prio * 3, threads, c[0], c[1], c[2], c[3]);
        }
    return 0;
}
