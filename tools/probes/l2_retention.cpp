// Does an XCD's L2 keep a kernel's lines for the NEXT kernel?  write_k stores a 2 MB array (block b -> its own 8 KB), then three readers:
// read_same (block b reads what block b wrote: same XCD if blocks are dealt round-robin), read_shift (block b reads block b+1's part: the
// neighbouring XCD), read_again (read_same a second time: lines a reader brought in).  Run under
//   rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d out -- ./l2_retention
// and compare hits per kernel.  hipcc --offload-arch=gfx950 -O2 -o l2_retention l2_retention.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void write_k(float4* a, float v) { a[blockIdx.x * 512 + threadIdx.x] = make_float4(v, v, v, v); }
__global__ void read_same(const float4* a, float* out) { float4 x = a[blockIdx.x * 512 + threadIdx.x]; if (x.x == -1.f) out[0] = x.y; }
__global__ void read_shift(const float4* a, float* out) { float4 x = a[((blockIdx.x + 1) % gridDim.x) * 512 + threadIdx.x]; if (x.x == -1.f) out[0] = x.y; }
__global__ void read_again(const float4* a, float* out) { float4 x = a[blockIdx.x * 512 + threadIdx.x]; if (x.x == -1.f) out[0] = x.y; }
int main() {
    const int blocks = 256;
    float4* a; float* out;
    hipMalloc(&a, blocks * 512 * sizeof(float4)); hipMalloc(&out, 4);
    hipStream_t s; hipStreamCreate(&s);
    for (int it = 0; it < 50; it++) {
        write_k<<<blocks, 512, 0, s>>>(a, (float) it);
        read_same<<<blocks, 512, 0, s>>>(a, out);
        read_again<<<blocks, 512, 0, s>>>(a, out);
        read_shift<<<blocks, 512, 0, s>>>(a, out);
    }
    hipStreamSynchronize(s);
    std::printf("L2 RETENTION PROBE DONE\n");
    return 0;
}
