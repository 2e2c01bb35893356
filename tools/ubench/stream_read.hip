// Read-only streaming ceiling of the box: every thread sums 16-byte vectors of a large buffer (grid-stride),
// one atomic per workgroup so nothing is optimised away.  Build: hipcc --offload-arch=gfx950 -O3 -o stream_read stream_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ p, size_t n, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) atomicAdd(out, acc);
}
int main() {
    const size_t bytes = (size_t)1 << 30;  // 1 GiB, larger than the 256 MB infinity cache
    uint4* d; unsigned* o;
    hipMalloc(&d, bytes); hipMalloc(&o, 4); hipMemset(d, 1, bytes); hipMemset(o, 0, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int grid : {1024, 2048, 4096, 8192}) {
        k_read<<<grid, 256>>>(d, bytes / 16, o);
        hipEventRecord(a);
        for (int r = 0; r < 10; r++) k_read<<<grid, 256>>>(d, bytes / 16, o);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("{\"ubench\": \"stream_read\", \"grid\": %d, \"GBps\": %.1f}\n", grid, bytes * 10.0 / (ms * 1e-3) / 1e9);
    }
    return 0;
}
