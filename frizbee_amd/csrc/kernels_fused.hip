// k12_fused: the streaming filter and the scorer of a short-haystack corpus in ONE persistent kernel, wave-autonomous (experiment).
//
// Path: Matcher::match_list's per-haystack loop (src/matcher/mod.rs:170-206) for an ASCII needle with max_typos = 0 over a corpus
// whose haystacks fit half a Smith-Waterman chunk (fzb_dp_short_applies) - the headline configuration.  Separately, the filter
// (k1_dfa) is bound by HBM (~0.9 of the achievable copy rate, little instruction issue) and the scorer (k2b_dp_short) by instruction
// issue (no memory traffic to speak of); run one after the other they add up.  Here every WAVE alternates between the two on its own -
// no barrier after the set-up, no wave ever waits for another one:
//
//   wave loop over tiles of 256 haystacks (4 per lane, dealt out by position):
//     1. the 4 haystacks of a lane through the LDS DFA (as k1_dfa), one accept bit each; ranks inside the tile from the 4 ballots;
//     2. the tile's survivor count is stored (and added to the count of its group of 16 tiles);
//     3. survivors (haystack, staging slot = tile * 256 + rank) go into the wave's own LDS queue;
//     4. whenever 64 (96 in every other workgroup of a CU) are queued every lane scores one: window search, trim, dp_cf.h rows, record
//        (the body of k2b_dp_short), reading its 32 bytes again (L2: the tile was streamed a few microseconds earlier); the record
//        goes to its staging slot.
//   The remainder of a wave's queue is scored when its tiles run out.
//   k_fused_gather then packs the staged tiles into the caller's array in tile order = haystack order (this replaces bitmap +
//   compaction).
//
// STATUS: opt-in (FZB_FUSED=1), not the default.  Measured on the headline configuration (MI355X, 10M x 32 B, 'deadbe'):
//   three kernels  k1_dfa 55.1 + k_compact1 6.2 + k2b_dp_short 42.0            = 101.5 us per query
//   this kernel    k12_fused 91.6 + k_fused_gather 8.0                          =  99.5 - 101.0 us
// rocprofv3 --pmc: SQ_INSTS_VALU 27.8M + SALU 5.4M + LDS 5.8M wave instructions (split pipeline: k1_dfa 10.7M + 4.0M + 5.1M, k2b_dp_short
// 16.9M + 3.1M + 0.5M): the fused kernel issues one instruction per ~4.8 cycles and SIMD over its whole run, the rate k2b_dp_short
// reaches - it is bound by instruction issue, and it executes the sum of both kernels' instructions.  In the split pipeline the
// filter's 22M instructions hide under its memory time (k1_dfa runs at 0.72 of 8 TB/s = 0.92 of the achievable copy rate); fused, they
// compete with the scorer's for the same issue slots, which is why overlapping the two buys so little.  Variants measured on the way
// (profiles/r02_fused_experiments.md): ordering by a decoupled look-back inside the kernel instead of staging + gather 174 us (with
// ~1000 tiles in flight the look-back chains are long); workgroup-wide tiles with barriers 113 us; one shared LDS queue per workgroup
// with compare-and-swap claims 122 us (a wave scoring next to three filtering waves runs its dependent chains alone); next tile's vectors
// requested before the scoring round, 3 workgroups per CU: no change; alternating 64 / 96-entry scoring triggers between the workgroups
// of a CU to shift their cycles: no change; no DP arithmetic at all 74 us, no DFA either 68 us (= streaming + queue + gather).
//
// Results are identical to the three-kernel pipeline by construction (same device functions, same order); tests/test_gpu_fused.py
// checks that.
#include "fzb_internal.h"
#include "kernels_common.h"
#include "dfa_lds.h"
#include "dp_body.h"
#include "dp_cf.h"
#include <algorithm>
#include <cstdlib>

#define FZB_FT 256u             // haystacks per wave tile
#define FZB_FG_SHIFT 4          // tiles per group count: 16 (4096 haystacks)
#define FZB_FUSED_QCAP 512u     // per wave: >= 95 left over + 256 new survivors; power of two
#define FZB_FUSED_DFA_BYTES 18432u  // the DFA table's LDS region: (FZB_MAX_ROWS + 1) * FZB_DFA_STRIDE
static_assert((FZB_MAX_ROWS + 1) * FZB_DFA_STRIDE <= FZB_FUSED_DFA_BYTES, "DFA region");

struct FusedShared {
    CfTables tab;
    u32 q_li[4][FZB_FUSED_QCAP];   // per wave: haystack (relative to `first`)
    u32 q_pos[4][FZB_FUSED_QCAP];  //           staging slot
};

// UL = 16 / 32: every haystack has exactly UL bytes (CorpusDev::uniform_len): a full tile's vectors are addressed from one wave-uniform
// base, and no per-lane length logic is left in the tile loop (the kernel is bound by instruction issue: rocprofv3 SQ_INSTS_*,
// profiles/r02_fused_*).  UL = 0: lengths from the end offsets (or any other uniform length).
template <int SWL, bool UPPER, typename ET, int UL>
__global__ __launch_bounds__(256, 4) void k12_fused(const u8* __restrict__ bytes, const ET* __restrict__ ends, u64 first, u32 count, u32 index_offset,
                                                                 const u8* __restrict__ dfa_g, int rows, u32 min_len, const NeedleDev nd, int wmode, u32 ulen,
                                                                 u16* __restrict__ tile_counts, u32* __restrict__ group_counts, fzb_match_rec* __restrict__ stage, int dbg, u32 phase_div) {
    // static LDS only: every address is a compile-time constant that folds into the ds instruction's offset field
    __shared__ __attribute__((aligned(256))) u8 lds[FZB_FUSED_DFA_BYTES];
    __shared__ FusedShared sh;
    const u8* dfa = lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const u32 wave = __builtin_amdgcn_readfirstlane((u32)tid >> 6);
    dfa_load_lds(lds, dfa_g, rows);
    cf_build_tables<UPPER>(nd, sh.tab);
    __syncthreads();  // the only barrier
    const u32 ntiles = (count + FZB_FT - 1) / FZB_FT;
    u32* const q_li = sh.q_li[wave];
    u32* const q_pos = sh.q_pos[wave];
    u32 q_head = 0, q_count = 0;  // wave-uniform
    // Waves that score at the same time should share a SIMD with waves that filter: the workgroups of a CU alternate between scoring as
    // soon as 64 survivors are queued and keeping 32 more in hand, which shifts their filter / score cycles by half a period.
    const u32 q_trigger = ((blockIdx.x / phase_div) & 1) ? 96u : 64u;
    const bool prio = !(dbg & 4);
    if (prio) __builtin_amdgcn_s_setprio(2);  // filtering (requesting the next vectors) goes first

    auto score_queued = [&](u32 n) {  // the first n entries of this wave's queue, one per lane
        if (prio) __builtin_amdgcn_s_setprio(0);  // scoring takes the issue slots the filtering waves leave
        if ((u32)lane < n) {
            const u32 slot = (q_head + lane) & (FZB_FUSED_QCAP - 1);
            const u32 li = q_li[slot], opos = q_pos[slot];
            u64 s;
            u32 L;
            uint4 q0 = make_uint4(0, 0, 0, 0), q1 = make_uint4(0, 0, 0, 0);
            if (UL != 0) {
                s = (first + li) * (u64)UL;
                L = UL;
                const uint4* vp = (const uint4*)(bytes + s);
                q0 = vp[0];
                if (UL == 32) q1 = vp[1];
            } else {
                haystack_span_u(ends, ulen, first + li, s, L);
                const uint4* vp = (const uint4*)(bytes + s);
                if (L > 0) q0 = vp[0];
                if (SWL > 32 && L > 16) q1 = vp[1];
            }
            u32 ws = 0, we = L;
            if (wmode == 1) cf_window_first_last_regs(nd, q0, q1, ws, we);
            // trim_haystack (matcher/algo.rs:332-338)
            const u32 sp = ws ? ws - 1 : 0;
            const bool include_exact = sp == 0 && we == L;
            const u32 m = we - sp;
            u32 score = 0;
            u32 hb[SWL / 4];
#pragma unroll
            for (int k = 0; k < SWL / 4; k++) hb[k] = 0;
            if (m > 0 && !(dbg & 1)) {
                load_window_regs<SWL / 4>(q0, q1, sp, m, hb);
                score = dp_single_chunk_cf_tab<SWL, UPPER, SWL / 4, CfNoRowHook>(nd, sp == 0, sh.tab, hb, CfNoRowHook{});
            }
            const bool exact = exact_match<SWL / 4>(nd, include_exact, m, hb);
            if (exact) score = (score + nd.exact_bonus) & 0xFFFF;
            fzb_match_rec rec;
            rec.index = index_offset + li;
            rec.score = (u16)score;
            rec.exact = exact ? 1 : 0;
            rec.valid = 0;
            stage[opos] = rec;
        }
        q_head = (q_head + n) & (FZB_FUSED_QCAP - 1);
        q_count -= n;
        if (prio) __builtin_amdgcn_s_setprio(2);
    };
    auto load_tile = [&](u32 t, u32 (&hl)[4], uint4 (&v0)[4], uint4 (&v1)[4]) {
        u64 hs[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const u32 li = t * FZB_FT + p * 64 + lane;
            hs[p] = 0;
            hl[p] = 0;
            if (t < ntiles && li < count) haystack_span_u(ends, ulen, first + li, hs[p], hl[p]);
        }
#pragma unroll
        for (int p = 0; p < 4; p++) {
            v0[p] = make_uint4(0, 0, 0, 0);
            v1[p] = make_uint4(0, 0, 0, 0);
            const uint4* vp = (const uint4*)(bytes + hs[p]);
            if (hl[p] > 0) v0[p] = vp[0];
            if (hl[p] > 16) v1[p] = vp[1];
        }
    };
    // the four waves of a workgroup take four consecutive tiles (one 1024-haystack stretch per workgroup step)
    const u32 tstride = gridDim.x * 4;
    u32 tile = blockIdx.x * 4 + wave;

    for (; tile < ntiles; tile += tstride) {
        // ---- 1. the DFA over this lane's four haystacks (k1_dfa) -----------------------------------------------------------------
        bool ok[4];
        if (UL != 0 && (tile + 1) * FZB_FT <= count) {
            const u8* tb = bytes + (first + (u64)tile * FZB_FT) * UL;  // wave-uniform
            uint4 v0[4], v1[4];
#pragma unroll
            for (int p = 0; p < 4; p++) {
                const uint4* vp = (const uint4*)(tb + (u32)(p * 64 + lane) * UL);
                v0[p] = vp[0];
                if (UL == 32) v1[p] = vp[1];
            }
            u32 st[4] = {0, 0, 0, 0};
            { const u32 w[4] = {v0[0].x, v0[1].x, v0[2].x, v0[3].x}; dfa_word4<false>(st, w, dfa); }
            { const u32 w[4] = {v0[0].y, v0[1].y, v0[2].y, v0[3].y}; dfa_word4<false>(st, w, dfa); }
            { const u32 w[4] = {v0[0].z, v0[1].z, v0[2].z, v0[3].z}; dfa_word4<false>(st, w, dfa); }
            { const u32 w[4] = {v0[0].w, v0[1].w, v0[2].w, v0[3].w}; dfa_word4<false>(st, w, dfa); }
            if (UL == 32) {
                { const u32 w[4] = {v1[0].x, v1[1].x, v1[2].x, v1[3].x}; dfa_word4<false>(st, w, dfa); }
                { const u32 w[4] = {v1[0].y, v1[1].y, v1[2].y, v1[3].y}; dfa_word4<false>(st, w, dfa); }
                { const u32 w[4] = {v1[0].z, v1[1].z, v1[2].z, v1[3].z}; dfa_word4<false>(st, w, dfa); }
                { const u32 w[4] = {v1[0].w, v1[1].w, v1[2].w, v1[3].w}; dfa_word4<false>(st, w, dfa); }
            }
            const bool len_ok = (u32)UL >= min_len;
#pragma unroll
            for (int p = 0; p < 4; p++) ok[p] = len_ok && st[p] == (u32)rows;
        } else {
        u32 hl[4];
        uint4 v0[4], v1[4];
        load_tile(tile, hl, v0, v1);
        u32 st[4] = {0, 0, 0, 0};
        if (dbg & 2) {
#pragma unroll
            for (int p = 0; p < 4; p++) st[p] = ((v0[p].x ^ v1[p].w) % 20u) == 0 ? (u32)rows : 0u;
        } else {
            if (hl[0] >= 16 && hl[1] >= 16 && hl[2] >= 16 && hl[3] >= 16) {
                { const u32 w[4] = {v0[0].x, v0[1].x, v0[2].x, v0[3].x}; dfa_word4<false>(st, w, dfa); }
                { const u32 w[4] = {v0[0].y, v0[1].y, v0[2].y, v0[3].y}; dfa_word4<false>(st, w, dfa); }
                { const u32 w[4] = {v0[0].z, v0[1].z, v0[2].z, v0[3].z}; dfa_word4<false>(st, w, dfa); }
                { const u32 w[4] = {v0[0].w, v0[1].w, v0[2].w, v0[3].w}; dfa_word4<false>(st, w, dfa); }
            } else {
#pragma unroll
                for (int p = 0; p < 4; p++) st[p] = dfa_partial<false>(st[p], v0[p], hl[p] >= 16 ? 16u : hl[p], dfa);
            }
            if (hl[0] >= 32 && hl[1] >= 32 && hl[2] >= 32 && hl[3] >= 32) {
                { const u32 w[4] = {v1[0].x, v1[1].x, v1[2].x, v1[3].x}; dfa_word4<false>(st, w, dfa); }
                { const u32 w[4] = {v1[0].y, v1[1].y, v1[2].y, v1[3].y}; dfa_word4<false>(st, w, dfa); }
                { const u32 w[4] = {v1[0].z, v1[1].z, v1[2].z, v1[3].z}; dfa_word4<false>(st, w, dfa); }
                { const u32 w[4] = {v1[0].w, v1[1].w, v1[2].w, v1[3].w}; dfa_word4<false>(st, w, dfa); }
            } else {
#pragma unroll
                for (int p = 0; p < 4; p++)
                    if (hl[p] > 16) st[p] = dfa_partial<false>(st[p], v1[p], hl[p] >= 32 ? 16u : hl[p] - 16, dfa);
            }
        }
#pragma unroll
        for (int p = 0; p < 4; p++) ok[p] = tile * FZB_FT + p * 64 + lane < count && hl[p] >= min_len && st[p] == (u32)rows;
        }
        // ---- 2. ranks inside the tile (haystack order: p-major, then lane), the tile's count ---------------------------------------
        u64 ball[4];
        u32 total = 0;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            ball[p] = __ballot(ok[p]);
            total += (u32)__popcll(ball[p]);
        }
        if (lane == 0) {
            tile_counts[tile] = (u16)total;
            if (total) atomicAdd(&group_counts[tile >> FZB_FG_SHIFT], total);
        }
        // ---- 3. queue the survivors with their staging slots -----------------------------------------------------------------------
        const u32 q_tail = q_head + q_count;
        u32 before = 0;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            if (ok[p]) {
                const u32 r = before + (u32)__popcll(ball[p] & (((u64)1 << lane) - 1));
                const u32 slot = (q_tail + r) & (FZB_FUSED_QCAP - 1);
                q_li[slot] = tile * FZB_FT + p * 64 + lane;
                q_pos[slot] = tile * FZB_FT + r;
            }
            before += (u32)__popcll(ball[p]);
        }
        q_count += total;
        // ---- 4. score full rounds ------------------------------------------------------------------------------------------------
        if (q_count >= q_trigger)
            while (q_count >= 64) score_queued(64);
    }
    while (q_count) score_queued(q_count < 64 ? q_count : 64u);
}

// Packs the staged tiles: tile t holds tile_counts[t] records at stage[t * 256 ..]; the records of all tiles, in tile order, go to
// out[0 ..] (clamped to the capacity).  A workgroup owns a run of GROUPS of 16 tiles: it reduces the group counts before its run itself
// (as k_compact1 does with its tile counts), scans its own tiles' counts in LDS, and then works per OUTPUT record - the record's tile by
// binary search in the scanned counts - so that the copies are independent of each other and the writes coalesced.
// The group counts are accumulated with atomics, so they must be zero at launch: there are two arrays, used by alternate launches, and
// this kernel clears the OTHER one (nobody reads it now; clearing its own would race with the workgroups still reducing it).
#define FZB_FG_TILES 256u  // tiles per batch of the gather kernel (64 groups)
__global__ __launch_bounds__(256) void k_fused_gather(const u16* __restrict__ tile_counts, const u32* __restrict__ group_counts, u32* __restrict__ group_counts_next,
                                                      const fzb_match_rec* __restrict__ stage, u32 ntiles,
                                                      fzb_match_rec* __restrict__ out, u32 capacity, u32* __restrict__ dev_count, u32* __restrict__ counters) {
    __shared__ u32 red[4];
    __shared__ u32 pre[FZB_FG_TILES + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 ngroups = (ntiles + (1u << FZB_FG_SHIFT) - 1) >> FZB_FG_SHIFT;
    const u32 G = (ngroups + gridDim.x - 1) / gridDim.x;
    const u32 g0 = min(blockIdx.x * G, ngroups), g1 = min(g0 + G, ngroups);
    u32 part = 0;
    {
        const uint4* c4 = (const uint4*)group_counts;
        const u32 n4 = g0 / 4;
        u32 i = tid;
        for (; i + 768 < n4; i += 1024) {
            const uint4 a = c4[i], b = c4[i + 256], c = c4[i + 512], d = c4[i + 768];
            part += (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (d.x + d.y + d.z + d.w);
        }
        for (; i < n4; i += 256) {
            const uint4 a = c4[i];
            part += a.x + a.y + a.z + a.w;
        }
        for (u32 k = 4 * n4 + tid; k < g0; k += 256) part += group_counts[k];
    }
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    u32 base = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    const u32 t0 = g0 << FZB_FG_SHIFT, t1 = min(g1 << FZB_FG_SHIFT, ntiles);
    for (u32 tb = t0; tb < t1; tb += FZB_FG_TILES) {
        const u32 nt = min(FZB_FG_TILES, t1 - tb);
        const u32 c = (u32)tid < nt ? (u32)tile_counts[tb + tid] : 0u;
        u32 incl = c;
        for (int off = 1; off < 64; off <<= 1) {
            const u32 v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        if (lane == 63) red[wave] = incl;
        __syncthreads();
        u32 wb = 0;
        for (int w = 0; w < wave; w++) wb += red[w];
        pre[tid] = wb + incl - c;  // records of this batch before tile tb + tid
        const u32 batch_total = red[0] + red[1] + red[2] + red[3];
        if (tid == 0) pre[FZB_FG_TILES] = batch_total;
        __syncthreads();
        for (u32 j = tid; j < batch_total; j += 256) {
            u32 lo = 0, hi = nt;  // the last tile with pre[tile] <= j
            while (hi - lo > 1) {
                const u32 mid = (lo + hi) >> 1;
                if (pre[mid] <= j) lo = mid; else hi = mid;
            }
            const u32 o = base + j;
            if (o < capacity) out[o] = stage[(size_t)(tb + lo) * FZB_FT + (j - pre[lo])];
        }
        base += batch_total;
        __syncthreads();
    }
    for (u32 g = g0 + tid; g < g1; g += 256) group_counts_next[g] = 0;  // ready for the next launch
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        for (int k = 2; k < 16; k++) counters[k] = 0;
        counters[0] = base;
        counters[1] = base;
        if (dev_count) { dev_count[0] = base < capacity ? base : capacity; dev_count[1] = base; }
    }
}

bool fzb_fused_applies(const CorpusDev& c, const LaunchCfg& lc, const NeedleDev& nd, int wmode) {
    const bool on = getenv("FZB_FUSED") != nullptr;  // opt-in (see STATUS above); read per call: the tests switch between the two pipelines
    return on && lc.filter_mode == 1 && lc.filter_exact && lc.cf_ok && !nd.unicode && (wmode == 1 || wmode == 2) && fzb_dp_short_applies(c, lc.sw_lanes, 2);
}

void fzb_launch_fused(const CorpusDev& c, u64 first, u32 count, u32 index_offset, const u8* dfa, const NeedleDev& nd, int sw_lanes, int wmode, u32 min_len, u16* tile_counts,
                      u32* group_counts, u32* group_counts_next, fzb_match_rec* stage, fzb_match_rec* out, u32 capacity, u32* dev_count, u32* counters, int num_cus, hipStream_t st) {
    bool upper = false;
    for (int r = 0; r < nd.rows; r++) upper = upper || (nd.c[r] >= 'A' && nd.c[r] <= 'Z');
    const u32 ntiles = (count + FZB_FT - 1) / FZB_FT;
    const u32 ul = getenv("FZB_FUSED_NO_UNIFORM") ? 0u : c.uniform_len;  // 16 / 32 select the uniform-length instantiations
    const int dbg = getenv("FZB_FUSED_DBG") ? atoi(getenv("FZB_FUSED_DBG")) : 0;  // experiments: 1 = no DP arithmetic, 2 = no DFA (wrong results)
#define FZB_K12(SWL, U, ET) do { if (ul == 32 && SWL == 64) FZB_K12P(SWL, U, ET, (SWL == 64 ? 32 : 16)); else if (ul == 16) FZB_K12P(SWL, U, ET, 16); else FZB_K12P(SWL, U, ET, 0); } while (0)
#define FZB_K12P(SWL, U, ET, L_)                                                                                                                   \
    do {                                                                                                                                          \
        static int per_cu = 0;                                                                                                                    \
        if (!per_cu && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k12_fused<SWL, U, ET, L_>, 256, 0) != hipSuccess || per_cu < 1)) per_cu = 2; \
        int wgs = per_cu;                                                                                                                         \
        if (const char* e_ = getenv("FZB_FUSED_WGS_PER_CU")) { const int v_ = atoi(e_); if (v_ >= 1 && v_ < per_cu) wgs = v_; }                    \
        const int grid = (int)std::max<u32>(1u, std::min<u32>((u32)(num_cus * wgs), (ntiles + 3) / 4));                                            \
        hipLaunchKernelGGL((k12_fused<SWL, U, ET, L_>), dim3(grid), dim3(256), 0, st, c.bytes, (const ET*)c.ends, first, count, index_offset, dfa, nd.rows, min_len, nd, wmode, \
                           c.uniform_len, tile_counts, group_counts, stage, dbg, (u32)(getenv("FZB_FUSED_NO_STAGGER") ? 0x40000000 : num_cus));                                                                   \
    } while (0)
#define FZB_K12_ET(SWL, U) do { if (c.ends_u64) FZB_K12(SWL, U, u64); else FZB_K12(SWL, U, u32); } while (0)
#define FZB_K12_U(SWL) do { if (upper) FZB_K12_ET(SWL, true); else FZB_K12_ET(SWL, false); } while (0)
    if (sw_lanes == 64) FZB_K12_U(64); else FZB_K12_U(32);
    const u32 ngroups = (ntiles + (1u << FZB_FG_SHIFT) - 1) >> FZB_FG_SHIFT;
    static const int gdiv = getenv("FZB_FUSED_GATHER_GROUPS") ? std::max(1, atoi(getenv("FZB_FUSED_GATHER_GROUPS"))) : 4;  // groups per gather workgroup (tuning knob)
    const int ggrid = (int)std::max<u32>(1u, std::min<u32>((u32)num_cus * 4, (ngroups + gdiv - 1) / gdiv));
    hipLaunchKernelGGL(k_fused_gather, dim3(ggrid), dim3(256), 0, st, tile_counts, group_counts, group_counts_next, stage, ntiles, out, capacity, dev_count, counters);
}
