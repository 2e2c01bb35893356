// Micro-benchmark: issue rate of the VALU ops the DP kernel is made of (gfx950).  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
#define AS_US2(x) __builtin_bit_cast(us2, x)
#define AS_U32(x) __builtin_bit_cast(unsigned, x)
template <int OP>
__global__ void k(unsigned* out, unsigned seed, int iters, unsigned long long* clk) {
    // shader-clock ticks (s_memtime) against the constant 100 MHz counter (s_memrealtime): the effective clock under this load
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    unsigned a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = seed * (i + 1) + threadIdx.x;
    unsigned c = seed | 0x00010001u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (OP == 0) a[i] = a[i] + c;                                                                 // v_add_u32
            if (OP == 1) a[i] = AS_U32(__builtin_elementwise_sub_sat(AS_US2(a[i]), AS_US2(c)));              // v_pk_sub_u16 clamp
            if (OP == 2) a[i] = AS_U32(__builtin_elementwise_max(AS_US2(a[i]), AS_US2(c + i)));              // v_pk_max_u16
            if (OP == 3) a[i] = AS_U32(AS_US2(a[i]) * AS_US2(c) + AS_US2(c));                                // v_pk_mad_u16
            if (OP == 4) a[i] = __builtin_amdgcn_alignbit(a[i], a[(i + 1) & 15], 16);                       // v_alignbit_b32
            if (OP == 5) a[i] = (a[i] | c) ^ a[(i + 1) & 15];                                                // v_bitop3_b32
            if (OP == 6) a[i] = AS_U32(AS_US2(a[i]) + AS_US2(c));                                            // v_pk_add_u16
            if (OP == 7) a[i] = max(a[i], c + i);                                                            // v_max_u32
            if (OP == 8) a[i] = AS_U32(AS_US2(a[i]) * AS_US2(c + i));                                        // v_pk_mul_lo_u16
            if (OP == 9) a[i] = __builtin_amdgcn_perm(a[i], a[(i + 1) & 15], 0x05010400u + i);              // v_perm_b32
            if (OP == 10) a[i] = AS_U32(__builtin_elementwise_min(AS_US2(a[i]), AS_US2(c + i)));             // v_pk_min_u16
            if (OP == 11) a[i] = (a[i] << 3) | a[(i + 1) & 15];                                              // v_lshl_or_b32
            if (OP == 12) a[i] = a[i] & (c + i);                                                             // v_and_b32
        }
    }
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - t0; clk[1] = wall_clock64() - w0; }
}
template <int OP>
void run(const char* name, unsigned* d, int wavesPerSimd) {
    int iters = 4000;
    dim3 grid(256 * wavesPerSimd), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    static unsigned long long* clk = nullptr;
    if (!clk) hipMalloc(&clk, 16);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, 12345u, 10, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, 12345u, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = hc[1] ? (double)hc[0] / ((double)hc[1] / 100e6) / 1e9 : 0.0;
    double waveinstr = (double)grid.x * 4 * iters * 16;
    double per_simd_cycles = ms * 1e-3 * 2.4e9;  // assuming 2.4 GHz
    double instr_per_simd = waveinstr / 1024.0;
    printf("%-16s waves/SIMD=%d  %.3f ms  %.2f cycles/wave-instr/SIMD (@2.4GHz)  %.2f at the measured %.2f GHz  %.2f Tlane-ops/s\n", name, wavesPerSimd, ms, per_simd_cycles / instr_per_simd,
           ms * 1e-3 * ghz * 1e9 / instr_per_simd, ghz, waveinstr * 64 / (ms * 1e-3) / 1e12);
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4}) {
        run<0>("v_add_u32", d, w); run<1>("v_pk_sub_u16", d, w); run<2>("v_pk_max_u16", d, w); run<3>("v_pk_mad_u16", d, w);
        run<4>("v_alignbit_b32", d, w); run<5>("v_bitop3_b32", d, w); run<6>("v_pk_add_u16", d, w); run<7>("v_max_u32", d, w);
        run<8>("v_pk_mul_lo_u16", d, w); run<9>("v_perm_b32", d, w); run<10>("v_pk_min_u16", d, w); run<11>("v_lshl_or_b32", d, w); run<12>("v_and_b32", d, w);
    }
    return 0;
}
