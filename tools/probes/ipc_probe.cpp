// Probe used while designing the multi-GPU mailbox: two processes share an uncached device allocation through
// hipIpc and ping-pong through it from kernels.  Build: hipcc --offload-arch=gfx950 -O2 -o ipc_probe ipc_probe.cpp
// probe: can two processes share an uncached device allocation through hipIpc* and ping-pong through it from kernels?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <sys/wait.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("[%d] %s -> %s\n", getpid(), #x, hipGetErrorString(e)); std::exit(3); } } while (0)

__global__ void writer(unsigned long long* box, unsigned int seq, int k) {
    if (threadIdx.x < 16) __hip_atomic_store(&box[k * 16 + threadIdx.x], ((unsigned long long) seq << 32) | (threadIdx.x + 100u * k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void reader(unsigned long long* box, unsigned int seq, int k, unsigned int* out, long long* cycles) {
    long long t0 = wall_clock64();
    unsigned long long v;
    int spins = 0;
    do { v = __hip_atomic_load(&box[k * 16 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); spins++; }
    while ((unsigned int) (v >> 32) != seq && wall_clock64() - t0 < 200000000LL);   // 2 s at 100 MHz
    out[threadIdx.x] = (unsigned int) v;
    if (threadIdx.x == 0) { cycles[0] = wall_clock64() - t0; cycles[1] = spins; }
}

int main(int argc, char** argv) {
    int mode = argc > 1 ? std::atoi(argv[1]) : 0;      // 0: uncached, 1: fine-grained, 2: plain hipMalloc
    int p2c[2], c2p[2];
    pipe(p2c); pipe(c2p);
    pid_t pid = fork();
    const bool parent = pid != 0;
    CK(hipSetDevice(0));
    unsigned long long* mine = nullptr;
    if (mode == 2) CK(hipMalloc(&mine, 4096));
    else CK(hipExtMallocWithFlags((void**) &mine, 4096, mode == 0 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained));
    CK(hipMemset(mine, 0, 4096));
    CK(hipDeviceSynchronize());
    hipIpcMemHandle_t h, ho;
    CK(hipIpcGetMemHandle(&h, mine));
    write(parent ? p2c[1] : c2p[1], &h, sizeof h);
    read(parent ? c2p[0] : p2c[0], &ho, sizeof ho);
    unsigned long long* other = nullptr;
    CK(hipIpcOpenMemHandle((void**) &other, ho, hipIpcMemLazyEnablePeerAccess));
    unsigned int* out; long long* cyc;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 16));
    // ping-pong 200 rounds: each side writes into the OTHER's box (slot = own id) and waits on its own box for the peer
    const int me = parent ? 0 : 1, peer = 1 - me;
    long long tot = 0, spins = 0;
    for (unsigned int seq = 1; seq <= 200; seq++) {
        hipLaunchKernelGGL(writer, 1, 64, 0, 0, other, seq, me);
        hipLaunchKernelGGL(reader, 1, 16, 0, 0, mine, seq, peer, out, cyc);
        CK(hipDeviceSynchronize());
        unsigned int ho_[16]; long long c[2];
        CK(hipMemcpy(ho_, out, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost));
        for (int i = 0; i < 16; i++) if (ho_[i] != i + 100u * peer) { std::printf("[%s] seq %u word %d = %u WRONG (timeout?)\n", parent ? "parent" : "child", seq, i, ho_[i]); std::exit(4); }
        if (seq > 20) { tot += c[0]; spins += c[1]; }
    }
    std::printf("[%s] mode %d OK: 200 rounds, reader avg wait %.2f us, %.1f spins\n", parent ? "parent" : "child", mode, tot / 180.0 / 100.0, spins / 180.0);
    CK(hipIpcCloseMemHandle(other));
    if (parent) { int st; waitpid(pid, &st, 0); return WEXITSTATUS(st); }
    return 0;
}
