// Lookups of the streaming filter's DFA table in LDS (kernels_filter.hip).
// ABS = true: the table starts at LDS address 0 (the kernel's only LDS object, dynamic); ABS = false: the table is a static __shared__
// array, whose compile-time address folds into the ds_read's offset field.  Either way a lookup is v_perm + ds_read_u8.
#pragma once
#include "kernels_common.h"

// LDS layout of the table: row s (= state s) starts at byte s * FZB_DFA_STRIDE, entry [s][b] = next state.  The stride is 288 = 72
// dwords, not 256: a ds_read_u8's bank is (address / 4) mod 32, so with 256-byte rows the entries of one byte value sit in the SAME bank
// for every state - and text is mostly lowercase letters, 7 dwords of a row: the lanes of a wave, in 3-4 different states, met in those
// 7 banks (rocprofv3: SQ_LDS_BANK_CONFLICT = 53% of SQ_LDS_IDX_ACTIVE in k1_dfa, 4.2 LDS cycles per lookup instead of 2).  72 mod 32 = 8
// moves every state's letters 8 banks on, so four consecutive states do not meet at all.  Costs one v_lshl_add per lookup.
#ifndef FZB_DFA_STRIDE
#define FZB_DFA_STRIDE 288u
#endif
#define FZB_DFA_LDS_BYTES(rows) (((u32)(rows) + 1u) * FZB_DFA_STRIDE)
// Kernels that use the ABS = true lookups call this first: a table that does not start at LDS address 0 (somebody added a static
// __shared__ variable to the kernel) must stop the kernel, not read the wrong rows.
__device__ __forceinline__ void dfa_require_lds_base0(const u8* lds) {
    if ((u32)(uintptr_t)(const __attribute__((address_space(3))) u8*)lds != 0u) __builtin_trap();
}
__device__ __forceinline__ void dfa_load_lds(u8* lds, const u8* __restrict__ dfa_g, int rows) {  // every thread of the workgroup; sync afterwards
    for (u32 i = threadIdx.x * 4; i < ((u32)rows + 1) * 256; i += blockDim.x * 4) *(u32*)(lds + (i >> 8) * FZB_DFA_STRIDE + (i & 255)) = *(const u32*)(dfa_g + i);
}

template <int K, bool ABS = true>
__device__ __forceinline__ u32 dfa_step(u32 st, u32 w, const u8* dfa) {
    // address = st * 288 + byte K of w: (st << 8) | byte by v_perm_b32 (byte0 <- w.byteK, byte1 <- st.byte0, bytes 2,3 <- 0), + st * 32
    // The table is the kernel's only LDS object and starts at LDS address 0 (see the kernels), so the perm result IS the address.
    // Spelled as an integer-to-LDS-pointer cast because the address of an extern __shared__ array is a link-time symbol: indexing
    // `dfa` costs a v_add of that symbol (of 0) per lookup.
    const u32 addr = __builtin_amdgcn_perm(st, w, 0x0c0c0400u | (u32)K) + st * (FZB_DFA_STRIDE - 256u);  // st * stride + byte
    if (!ABS) return dfa[addr];
    return *(const __attribute__((address_space(3))) u8*)(uintptr_t)addr;
}
// s_waitcnt lgkmcnt(0) (vmcnt / expcnt untouched): ONE wait for a round of interleaved lookups instead of the compiler's one per use
// (lgkmcnt(3), (2), (1), (0)) - on this chip a wait takes an issue slot like any other instruction (issue model)
#define FZB_WAIT_LDS() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_sched_barrier(0); } while (0)  // the barriers keep the next round's v_perm behind the wait
template <bool ABS = true>
__device__ __forceinline__ void dfa_word4(u32 (&st)[4], const u32 (&w)[4], const u8* dfa) {
#pragma unroll
    for (int p = 0; p < 4; p++) st[p] = dfa_step<0, ABS>(st[p], w[p], dfa);
    FZB_WAIT_LDS();
#pragma unroll
    for (int p = 0; p < 4; p++) st[p] = dfa_step<1, ABS>(st[p], w[p], dfa);
    FZB_WAIT_LDS();
#pragma unroll
    for (int p = 0; p < 4; p++) st[p] = dfa_step<2, ABS>(st[p], w[p], dfa);
    FZB_WAIT_LDS();
#pragma unroll
    for (int p = 0; p < 4; p++) st[p] = dfa_step<3, ABS>(st[p], w[p], dfa);
    FZB_WAIT_LDS();
}
template <bool ABS = true>
__device__ __forceinline__ u32 dfa_partial(u32 st, const uint4& q, u32 nbytes, const u8* dfa) {
    const u32 w4[4] = {q.x, q.y, q.z, q.w};
    for (u32 k = 0; k < nbytes; k++) {
        const u32 b = (w4[k >> 2] >> (8 * (k & 3))) & 0xFF;
        st = ABS ? *(const __attribute__((address_space(3))) u8*)(uintptr_t)(st * FZB_DFA_STRIDE + b) : dfa[st * FZB_DFA_STRIDE + b];
    }
    return st;
}

