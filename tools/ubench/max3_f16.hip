// Is v_pk_maximum3_f16 (gfx950) an exact three-way maximum of packed UNSIGNED 16-bit values below 0x7C00, and what does it cost?
// For non-negative, finite binary16 patterns the float order is the integer order of the bit patterns, and the kernels are compiled
// with f16 denormals preserved (amdhsa_float_denorm_mode_16_64 = 3), so the instruction should return one of its inputs bit for bit.
// Part 1 checks it against two v_pk_max_u16 on every (x, y) with x in [0, 0x7C00) and 96 values of y and z (edges, denormal range,
// random); part 2 times a dependent chain of each.  Build: hipcc --offload-arch=gfx950 -O3 max3_f16.hip -o max3_f16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pmax(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b))); }
__device__ __forceinline__ unsigned pmax3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__global__ void check(const unsigned short* ys, int ny, unsigned long long* bad) {
    const unsigned x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= 0x7C00u) return;
    unsigned long long nb = 0;
    for (int i = 0; i < ny; i++)
        for (int j = 0; j < ny; j++) {
            const unsigned a = x | ((unsigned)ys[i] << 16), b = ys[i] | ((unsigned)ys[j] << 16), c = ys[j] | (x << 16);
            if (pmax3(a, b, c) != pmax(pmax(a, b), c)) nb++;
        }
    if (nb) atomicAdd(bad, nb);
}
template <int OP>
__global__ void rate(unsigned* out, unsigned seed, int iters) {
    unsigned a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = (seed * (i + 1) + threadIdx.x) & 0x3FFF3FFFu;
    const unsigned c = seed & 0x3FFF3FFFu;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (OP == 0) a[i] = pmax(pmax(a[i], a[(i + 1) & 15]), c + i);
            if (OP == 1) a[i] = pmax3(a[i], a[(i + 1) & 15], c + i);
        }
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    unsigned short h[96];
    int n = 0;
    for (unsigned v : {0u, 1u, 2u, 0x3FFu, 0x400u, 0x401u, 0x7FFu, 0x800u, 0x3BFFu, 0x3C00u, 0x7BFEu, 0x7BFFu}) h[n++] = (unsigned short)v;
    srand(7);
    while (n < 96) h[n++] = (unsigned short)(rand() % 0x7C00);
    unsigned short* dy; unsigned long long* db; unsigned* dout;
    hipMalloc(&dy, sizeof(h)); hipMalloc(&db, 8); hipMalloc(&dout, 256 * 8 * 256 * 4);
    hipMemcpy(dy, h, sizeof(h), hipMemcpyHostToDevice); hipMemset(db, 0, 8);
    hipLaunchKernelGGL(check, dim3(0x7C00 / 256), dim3(256), 0, 0, dy, 96, db);
    unsigned long long bad = 1;
    hipMemcpy(&bad, db, 8, hipMemcpyDeviceToHost);
    printf("v_pk_maximum3_f16 vs two v_pk_max_u16 on %d x %d x %d packed triples below 0x7C00: %llu differences\n", 0x7C00, 96, 96, bad);
    for (int w : {2, 4}) {
        for (int op = 0; op < 2; op++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int iters = 4000;
            dim3 grid(256 * w), block(256);
            if (op == 0) hipLaunchKernelGGL(rate<0>, grid, block, 0, 0, dout, 12345u, 10); else hipLaunchKernelGGL(rate<1>, grid, block, 0, 0, dout, 12345u, 10);
            hipEventRecord(e0);
            if (op == 0) hipLaunchKernelGGL(rate<0>, grid, block, 0, 0, dout, 12345u, iters); else hipLaunchKernelGGL(rate<1>, grid, block, 0, 0, dout, 12345u, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-28s waves/SIMD=%d  %.3f ms for %d three-way maxima per lane\n", op ? "v_pk_maximum3_f16" : "2 x v_pk_max_u16", w, ms, iters * 16);
        }
    }
    return bad != 0;
}
