// Probe (round 5, TUNING_LOG section 13): what an in-kernel rendezvous of G co-resident blocks costs on MI355X, by mechanism.
// Every block's wave 0 publishes W 8-byte words {seq:2 | payload:62} (flag and data in ONE atomic word, the LL idea) and the blocks
// then need the sum of all blocks' words.  Variants:
//   a2a   all-to-all: every block's wave 0 polls all G x W words itself
//   hop2  block 0 collects all words, folds, publishes W totals on one line, everybody polls that line
//   cnt   plain stores + release fence + atomic counter; pollers wait for the counter, then load the values
// each on (uc) an uncached allocation with system-scope atomics and (ag) ordinary hipMalloc memory with agent-scope atomics.
// Per block the 100 MHz wall clock is read when the block publishes and when it holds the totals; reported: the slowest block's
// "totals held" minus the LAST block's publish (the pure rendezvous latency) and minus the FIRST block's start (what a kernel sees).
// Build: hipcc --offload-arch=gfx950 -O2 -o rendezvous_probe rendezvous_probe.cpp ; run: ./rendezvous_probe [blocks] [words] [threads]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e)); std::exit(3); } } while (0)
typedef unsigned long long u64;

enum { V_A2A = 0, V_HOP2 = 1, V_CNT = 2, V_SPLIT = 3 };      // V_SPLIT: all-to-all, the block's waves poll one row (or two) each

template <bool SYS>
__device__ __forceinline__ void st(u64* p, u64 v) {
    if (SYS) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool SYS>
__device__ __forceinline__ u64 ld(const u64* p) {
    return SYS ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ long long wave_sum_ll(long long x) { for (int o = 32; o; o >>= 1) x += __shfl_xor(x, o, 64); return x; }

// box: [W][256] words (quantity-major, slot = block index: the layout of the product's accumulators); bcast: [W] words; stamps: [G][4]
template <int VARIANT, bool SYS, int W, int R>
__global__ void __launch_bounds__(512) rendezvous(u64* box, u64* bcast, u64* counter, u64* plain, unsigned seq, long long* stamps, const double4* data, double4* out, int work, unsigned* status) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int G = gridDim.x;
    __shared__ long long sh_tot[16];
    const long long t_in = wall_clock64();
    // some per-wave work in front (a tile's loads + arithmetic), so that blocks arrive with the skew a real kernel has
    double4 v = data[(size_t) (blockIdx.x * (blockDim.x >> 6) + wib) * 64 + lane];
    for (int i = 0; i < work; i++) { v.x = v.x * 1.0000001 + v.y; v.y = v.y * 0.9999999 + v.z; v.z = v.z * 1.0000001 + v.x; }
    __syncthreads();
    if (VARIANT == V_SPLIT) {
        // wave 0 publishes (as in the other variants), then wave w polls rows w, w + nwaves, ...: 4 loads per row per lane, a sum per row, LDS
        __shared__ long long sh_row[16];
        __shared__ long long sh_t[2];
        const int nwv = blockDim.x >> 6;
        const u64 tag = (u64) (seq & 3u) << 62;
        const long long mine = (long long) (blockIdx.x + 1) * 1000 + (long long) (v.x != 12345.0);
        if (wib == 0) {
            if (lane == 0) sh_t[0] = wall_clock64();
            if (lane < W * R) st<SYS>(&box[(lane / W) * (16 * 256) + (lane % W) * 256 + blockIdx.x], tag | ((u64) (mine + lane % W) & 0x3FFFFFFFFFFFFFFFull));
        }
        const u64* mybox = box + (blockIdx.x % R) * (16 * 256);
        const long long t0 = wall_clock64();
        bool timeout = false;
        for (int k = wib; k < W; k += nwv) {
            u64 x[4];
            for (;;) {
#pragma unroll
                for (int j = 0; j < 4; j++) x[j] = ld<SYS>(&mybox[k * 256 + lane + 64 * j]);
                bool all = true;
#pragma unroll
                for (int j = 0; j < 4; j++) if (lane + 64 * j < G && (x[j] >> 62) != (u64) (seq & 3u)) all = false;
                if (!__any(!all)) break;
                if (wall_clock64() - t0 > 20000000LL) { timeout = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            long long part = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) if (lane + 64 * j < G) part += (long long) (x[j] << 2) >> 2;
            part = wave_sum_ll(part);
            if (lane == 0) sh_row[k] = part;
        }
        __syncthreads();
        if (wib == 0 && lane == 0) {
            const long long t_got = wall_clock64();
            sh_tot[0] = sh_row[0];
            stamps[blockIdx.x * 4 + 0] = t_in; stamps[blockIdx.x * 4 + 1] = sh_t[0]; stamps[blockIdx.x * 4 + 2] = t_got; stamps[blockIdx.x * 4 + 3] = sh_row[0];
            if (timeout) status[0] = 1;
        }
    } else
    if (wib == 0) {
        const u64 tag = (u64) (seq & 3u) << 62;
        const long long mine = (long long) (blockIdx.x + 1) * 1000 + (long long) (v.x != 12345.0);      // payload: depends on the work
        const long long t_pub = wall_clock64();
        long long tot[W];
        bool timeout = false;
        // R replicas of the words (one store instruction, W x R lanes): poller b reads replica b % R, so that every line is polled by 1 / R of the blocks
        if (lane < W * R) st<SYS>(&box[(lane / W) * (16 * 256) + (lane % W) * 256 + blockIdx.x], tag | ((u64) (mine + lane % W) & 0x3FFFFFFFFFFFFFFFull));
        const u64* mybox = box + (blockIdx.x % R) * (16 * 256);
        const long long t0 = wall_clock64();
        if (VARIANT == V_A2A || (VARIANT == V_HOP2 && blockIdx.x == 0)) {
            // every slot of every quantity requested in ONE batch (4 W loads in flight per lane), then checked; a round that finds a
            // word missing is repeated as a whole
            u64 x[W][4];
            for (;;) {
#pragma unroll
                for (int k = 0; k < W; k++)
#pragma unroll
                    for (int j = 0; j < 4; j++) x[k][j] = ld<SYS>(&mybox[k * 256 + lane + 64 * j]);
                bool all = true;
#pragma unroll
                for (int k = 0; k < W; k++)
#pragma unroll
                    for (int j = 0; j < 4; j++) if (lane + 64 * j < G && (x[k][j] >> 62) != (u64) (seq & 3u)) all = false;
                if (!__any(!all)) break;
                if (wall_clock64() - t0 > 20000000LL) { timeout = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int k = 0; k < W; k++) {
                long long part = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) if (lane + 64 * j < G) part += (long long) (x[k][j] << 2) >> 2;
                tot[k] = wave_sum_ll(part);
            }
            if (VARIANT == V_HOP2) {
                long long t = 0;
#pragma unroll
                for (int k = 0; k < W; k++) if (lane == k) t = tot[k];
                if (lane < W) st<SYS>(&bcast[lane], tag | ((u64) t & 0x3FFFFFFFFFFFFFFFull));
            }
        }
        if (VARIANT == V_HOP2 && blockIdx.x != 0) {
            u64 x = 0;
            for (;;) {
                if (lane < W) x = ld<SYS>(&bcast[lane]);
                const bool ok = lane >= W || (x >> 62) == (u64) (seq & 3u);
                if (!__any(!ok)) break;
                if (wall_clock64() - t0 > 20000000LL) { timeout = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int k = 0; k < W; k++) tot[k] = __shfl((long long) (x << 2) >> 2, k, 64);
        }
        const long long t_got = wall_clock64();
        if (lane == 0) {
            sh_tot[0] = tot[0];
            stamps[blockIdx.x * 4 + 0] = t_in; stamps[blockIdx.x * 4 + 1] = t_pub; stamps[blockIdx.x * 4 + 2] = t_got; stamps[blockIdx.x * 4 + 3] = tot[0];
            if (timeout) status[0] = 1;
        }
    }
    __syncthreads();
    const double f = 1.0 + 1e-12 * (double) (sh_tot[0] & 0xFF);
    v.x *= f; v.y *= f; v.z *= f;
    out[(size_t) (blockIdx.x * (blockDim.x >> 6) + wib) * 64 + lane] = v;
}

template <int VARIANT, bool SYS, int W, int R>
static void run1(const char* name, int G, int threads, int work, u64* box, u64* bcast, u64* counter, u64* plain, long long* stamps, double4* a, double4* b, unsigned* status, hipStream_t s) {
    if (W * R > 64) return;      // (the replicas are written by one store instruction)
    const int reps = 40;
    std::vector<long long> h((size_t) G * 4);
    std::vector<double> lat_last, lat_first, pub_spread, kernel_us;
    CK(hipMemsetAsync(box, 0xFF, (size_t) 8 * 16 * 256 * 8, s));       // tag 3: not the first sequence number
    CK(hipMemsetAsync(bcast, 0xFF, 16 * 8, s));
    CK(hipMemsetAsync(counter, 0, 8, s));
    CK(hipMemsetAsync(status, 0, 4, s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL((rendezvous<VARIANT, SYS, W, R>), dim3(G), dim3(threads), 0, s, box, bcast, counter, plain, (unsigned) r, stamps, a, b, work, status);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        long long first_in = h[0], last_pub = h[1], first_pub = h[1], last_got = h[2];
        const long long expect = (long long) G * (G + 1) / 2 * 1000;
        bool ok = true;
        for (int g = 0; g < G; g++) {
            first_in = std::min(first_in, h[g * 4]); last_pub = std::max(last_pub, h[g * 4 + 1]); first_pub = std::min(first_pub, h[g * 4 + 1]); last_got = std::max(last_got, h[g * 4 + 2]);
            ok = ok && (h[g * 4 + 3] / 1000 * 1000 == expect || h[g * 4 + 3] - expect < G + 1);
        }
        if (!ok && r == reps - 1) std::printf("  [%s: totals differ from the expected sum]\n", name);
        if (r >= 5) { lat_last.push_back((last_got - last_pub) * 0.01); lat_first.push_back((last_got - first_in) * 0.01); pub_spread.push_back((last_pub - first_pub) * 0.01); kernel_us.push_back(ms * 1e3); }
    }
    unsigned st; CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    std::printf("%-8s G %4d W %2d thr %3d work %4d: last publish -> all hold totals %5.2f us (min %5.2f max %5.2f) | first wave in -> all hold %5.2f us | publish spread %5.2f us | event %5.1f us%s\n",
                name, G, W, threads, work, med(lat_last), lat_last.front(), lat_last.back(), med(lat_first), med(pub_spread), med(kernel_us), st ? "  [TIMEOUT]" : "");
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? std::atoi(argv[1]) : 251, W = argc > 2 ? std::atoi(argv[2]) : 3, threads = argc > 3 ? std::atoi(argv[3]) : 512, work = argc > 4 ? std::atoi(argv[4]) : 200;
    if (W > 16) return 2;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    u64 *box_uc, *bcast_uc, *cnt_uc, *plain_uc, *box, *bcast, *cnt, *plain;
    const size_t bytes = (size_t) 8 * 16 * 256 * 8;
    CK(hipExtMallocWithFlags((void**) &box_uc, bytes, hipDeviceMallocUncached));
    CK(hipExtMallocWithFlags((void**) &bcast_uc, 4096, hipDeviceMallocUncached));
    CK(hipExtMallocWithFlags((void**) &cnt_uc, 4096, hipDeviceMallocUncached));
    CK(hipExtMallocWithFlags((void**) &plain_uc, bytes, hipDeviceMallocUncached));
    CK(hipMalloc(&box, bytes)); CK(hipMalloc(&bcast, 4096)); CK(hipMalloc(&cnt, 4096)); CK(hipMalloc(&plain, bytes));
    long long* stamps; CK(hipMalloc(&stamps, (size_t) G * 4 * 8));
    unsigned* status; CK(hipMalloc(&status, 4));
    const size_t n = (size_t) G * (threads / 64) * 64;
    double4 *a, *b; CK(hipMalloc(&a, n * sizeof(double4))); CK(hipMalloc(&b, n * sizeof(double4)));
    std::vector<double4> h(n, double4{0.5, 0.25, 0.125, 1.0});
    CK(hipMemcpy(a, h.data(), n * sizeof(double4), hipMemcpyHostToDevice));
#define RUNALL(WW) do { \
    run1<V_A2A, true, WW, 1>("a2a-uc-r1", G, threads, work, box_uc, bcast_uc, cnt_uc, plain_uc, stamps, a, b, status, s); \
    run1<V_A2A, true, WW, 2>("a2a-uc-r2", G, threads, work, box_uc, bcast_uc, cnt_uc, plain_uc, stamps, a, b, status, s); \
    run1<V_A2A, true, WW, 4>("a2a-uc-r4", G, threads, work, box_uc, bcast_uc, cnt_uc, plain_uc, stamps, a, b, status, s); \
    run1<V_A2A, true, WW, 8>("a2a-uc-r8", G, threads, work, box_uc, bcast_uc, cnt_uc, plain_uc, stamps, a, b, status, s); \
    run1<V_A2A, false, WW, 1>("a2a-ag-r1", G, threads, work, box, bcast, cnt, plain, stamps, a, b, status, s); \
    run1<V_A2A, false, WW, 8>("a2a-ag-r8", G, threads, work, box, bcast, cnt, plain, stamps, a, b, status, s); \
    run1<V_SPLIT, true, WW, 1>("split-uc-r1", G, threads, work, box_uc, bcast_uc, cnt_uc, plain_uc, stamps, a, b, status, s); \
    run1<V_SPLIT, true, WW, 4>("split-uc-r4", G, threads, work, box_uc, bcast_uc, cnt_uc, plain_uc, stamps, a, b, status, s); \
    run1<V_HOP2, true, WW, 1>("hop2-uc", G, threads, work, box_uc, bcast_uc, cnt_uc, plain_uc, stamps, a, b, status, s); } while (0)
    if (G > 256) return 2;
    if (W == 3) RUNALL(3); else if (W == 7) RUNALL(7); else if (W == 10) RUNALL(10); else return 2;
    return 0;
}
