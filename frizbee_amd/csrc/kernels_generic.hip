// gfx950 kernels, stage 2c: the generic scorer.  One WAVE per haystack, one DP column per lane, so a wave64
// IS the reference's 64-lane score vector (AVX-512/u8 class); narrower classes use the low SWL lanes.
// `shift_right_padded::<L>` is a wave shuffle whose low L lanes are refilled from the previous chunk's row,
// kept per needle row in LDS (the reference's score_matrix column, src/smith_waterman/matrix.rs).
// Handles everything the single-chunk thread-per-haystack kernel does not:
//   * ASCII windows wider than one chunk               src/smith_waterman/algo/ascii.rs:10-158
//   * the unicode scorer (any width)                   src/smith_waterman/algo/unicode.rs:10-273,
//                                                      src/smith_waterman/algo/unicode_gap.rs:110-236
//   * the greedy fallback beyond 1024 bytes            src/smith_waterman/greedy.rs:7-91
// Arithmetic is done per lane in 32-bit registers and wrapped to the emulated lane type (u8 / u16) after
// every add, so it follows the reference's wrapping add / saturating sub literally.
#include "kernels_common.h"

#define GEN_WAVES 4

__device__ __forceinline__ u32 subs(u32 a, u32 b) { return a > b ? a - b : 0; }

// one entry of a previous chunk's vector.  GLOBAL: the vectors of a long needle live in a per-wave global slab that is rewritten
// chunk after chunk by other lanes of the same wave - read around the vector L1 (agent-scope load), after the writer's fence
template <bool GLOBAL>
__device__ __forceinline__ u32 adj_load(const u16* p) {
    if (GLOBAL) return (u32)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (u32)*p;
}
// shift right by k lanes; lanes < k take adj[SWL - k + lane] (adj = previous chunk's vector in LDS / the slab, or nullptr => 0)
template <int SWL, bool GLOBAL = false>
__device__ __forceinline__ u32 shift_pad(u32 v, int k, const u16* adj, int lane) {
    u32 t = __shfl_up(v, k);
    if (lane < k) t = adj ? adj_load<GLOBAL>(&adj[SWL - k + lane]) : 0u;
    return t;
}
// same with the adjacent vector held in registers of the same wave (lane l holds adjv[l])
template <int SWL>
__device__ __forceinline__ u32 shift_pad_reg(u32 v, int k, u32 adjv, int lane) {
    u32 t = __shfl_up(v, k);
    u32 a = __shfl(adjv, (SWL - k + lane) & 63);
    return lane < k ? a : t;
}

// match_greedy (src/smith_waterman/greedy.rs:7-91), run by one lane
// `fwd` (optional): receives the haystack position matched by every needle byte, in needle order
template <typename ND>
__device__ u32 greedy_score(const ND& nd, const u8* __restrict__ h, u32 hlen, bool include_prefix, u32* __restrict__ fwd = nullptr,
                        bool* matched = nullptr) {
    const u32 n = (u32)nd.nbytes;
    if (matched) *matched = false;
    if (n > hlen) return 0;
    u32 score = 0, hi = 0;
    bool dben = false, prev_lower = false, prev_delim = false;
    auto sadd = [](u32 a, u32 b) { u32 s = a + b; return s > 0xFFFFu ? 0xFFFFu : s; };
    for (u32 ni = 0; ni < n; ni++) {
        const u32 nc = nd.c[ni], fc = nd.f[ni];
        const u32 hstart = hi;
        bool found = false;
        while (hi <= hlen - n + ni) {
            const u32 hc = h[hi];
            const bool digit = hc >= '0' && hc <= '9', upper = hc >= 'A' && hc <= 'Z', lower = hc >= 'a' && hc <= 'z';
            const bool delim = hc < 128 && !(lower || upper || digit);
            if (!delim) dben = true;
            if (nc != hc && fc != hc) {
                prev_delim = dben && delim;
                prev_lower = lower;
                hi++;
                continue;
            }
            score = sadd(score, nd.match_score);
            if (fwd) fwd[ni] = hi;
            if (hi != hstart && ni != 0) {
                u32 gl = hi - hstart;
                gl = gl > 0 ? gl - 1 : 0;
                if (gl > 0xFFFF) gl = 0xFFFF;
                u32 ext = (u32)nd.gex * gl;
                if (ext > 0xFFFF) ext = 0xFFFF;
                score = subs(score, sadd((u32)nd.gap_open, ext));
            }
            if (nc == hc) score = sadd(score, nd.matching_case);
            if (upper && prev_lower) score = sadd(score, nd.capitalization);
            if (include_prefix && hi == 0) score = sadd(score, nd.prefix);
            if (prev_delim && !delim) score = sadd(score, nd.delimiter);
            prev_delim = dben && delim;
            prev_lower = lower;
            hi++;
            found = true;
            break;
        }
        if (!found) return 0;
    }
    if (matched) *matched = true;
    return score;
}

// TRACE = the matched-indices form (src/smith_waterman/algo/mod.rs:49-152, alignment_iter.rs:35-181): direct mode only; every
// (row, column) cell is also written to the wave's slot of `trace.cells` (score | match bit << 16, the reference's score_matrix
// and match_masks), and lane 0 then walks the alignment back from the first column of the last row that holds the score,
// writing the matched byte positions (reverse order, as the reference returns them) to trace.pos[opos * stride ..].
struct TraceArgs {
    u32* cells;   // per wave: (rows + 1) x TRACE_W dwords
    u32* pos;     // per output record: `stride` positions
    u32* npos;    // per output record: how many
    u32 stride;
};
// list mode only: run over the list iff lo <= *count < hi (nullptr: always) and read it forwards (entry q at list + 4 q) instead of downwards
struct GateArgs {
    const u32* count;
    u32 lo, hi;
    int forward;
    const u32* alt_list;   // outside [lo, hi): walk this list instead (entry q at alt_list - 4 (q + 1)), *alt_count entries; nullptr: return
    const u32* alt_count;
};
#define TRACE_W (FZB_MAX_HAYSTACK_LEN + 2 * 64)  // columns: the zero chunk + up to 1024 bytes rounded up to a chunk

template <int SWL, bool UNICODE, bool TRACE, typename ND = NeedleDev, bool SLAB = ND::kLong>
__global__ __launch_bounds__(GEN_WAVES * 64) void k2c_generic(const u8* __restrict__ bytes, const EndsAny ends, u64 first, u32 index_offset,
                                                              const u32* __restrict__ items, const u32* __restrict__ win, int wmode,
                                                              const u32* __restrict__ list, const u32* __restrict__ n_list_ptr, const ND nd,
                                                              fzb_match_rec* __restrict__ out, u32 capacity, u32* __restrict__ dev_count, u32* __restrict__ counters,
                                                              const TraceArgs trace, u16* __restrict__ long_adj, const GateArgs gate) {
    const u32* lst = list;  // the list this launch walks, its length, its direction
    const u32* nptr = n_list_ptr;
    int fwd = gate.forward;
    if (gate.count) {  // (uniform) this launch serves the list only when its length is in [lo, hi): another kernel takes it otherwise -
        const u32 gn = *gate.count;  // and this launch then walks the alternative list, if it was given one (read downwards)
        if (gn < gate.lo || gn >= gate.hi) {
            if (!gate.alt_list) return;
            lst = gate.alt_list;
            nptr = gate.alt_count;
            fwd = 0;
        }
    }
    // per wave: previous chunk's row / match mask (ASCII) or pending mask (unicode), one vector per needle row - in LDS (sized by the
    // needle's actual rows: dynamic shared memory; sized for the 63 rows the by-value needle can have it was 64 KB per workgroup, two
    // workgroups = eight waves per CU whatever the needle), or - SLAB: a long needle whose vectors do not fit LDS - in the wave's slab of
    // `long_adj` ((rows + 1) x SWL x 2 entries, global memory).  A long needle of up to 127 rows at 32 lanes still fits LDS (round 4).
    constexpr bool LONG = SLAB;
    extern __shared__ __attribute__((aligned(16))) u16 s_adj[];
    const int lane = lane_id();
    const int wv = threadIdx.x >> 6;
    u16* const adj_row_base = LONG ? long_adj + (size_t)(blockIdx.x * GEN_WAVES + wv) * 2 * (size_t)(nd.rows + 1) * SWL : s_adj + (size_t)wv * 2 * (size_t)(nd.rows + 1) * SWL;
    u16* const adj_aux_base = adj_row_base + (size_t)(nd.rows + 1) * SWL;
    const u32 nlist = *nptr;
    if (!lst && dev_count && blockIdx.x == 0 && threadIdx.x == 0) { dev_count[0] = nlist < capacity ? nlist : capacity; dev_count[1] = nlist; }
    const u32 LM = (u32)nd.lane_mask;
    const u32 rows = (u32)nd.rows;
    const u32 Mc = nd.match_plus_mismatch & LM, X = nd.mismatch & LM, gex = nd.gex & LM, gopm = nd.gopm & LM;
    const u32 caseb = nd.matching_case & LM, capb = nd.capitalization & LM, delimb = nd.delimiter & LM, prefixb = nd.prefix & LM;
    const bool active = lane < SWL;
    u32* const cells = TRACE ? trace.cells + (size_t)(blockIdx.x * GEN_WAVES + wv) * (size_t)(rows + 1) * TRACE_W : nullptr;

    for (u32 q = blockIdx.x * GEN_WAVES + wv; q < nlist; q += gridDim.x * GEN_WAVES) {
        // list entries: (output position, window start, window end, local haystack index) queued by the single-chunk kernel
        // list mode: the queue grows downwards from `lst` (the end of the chunk's queue slice): entry q is at lst - 4 (q + 1)
        const u32* le = lst ? (fwd ? lst + 4 * (size_t)q : lst - 4 * (size_t)(q + 1)) : nullptr;
        const u32 j = lst ? le[0] : q;          // rank inside this chunk (direct mode) / absolute output position (list mode)
        const u32 opos = j;
        if (opos >= capacity) continue;
        const u32 li = lst ? le[3] : (items ? items[j] : j);
        u64 s;
        u32 L;
        haystack_span(ends, first + li, s, L);
        const u8* hay = bytes + s;
        u32 ws, we;
        if (lst) { ws = le[1]; we = le[2]; }
        else if (wmode == 2) { ws = 0; we = L; }
        else if (TRACE && wmode == 1) {
            // 0-typo ASCII window in its lane-free form (src/prefilter/algo/ascii.rs:6-72): first occurrence of the first needle
            // byte, one past the last occurrence of the last one (either case); the haystack passed the exact filter
            ws = 0xFFFFFFFFu;
            we = 0;
            for (u32 b = 0; b < L; b += 64) {
                const u32 p = b + lane;
                const u32 hb = p < L ? hay[p] : 0u;
                const u64 mf = __ballot(p < L && (hb == nd.c[0] || hb == nd.f[0]));
                const u64 ml = __ballot(p < L && (hb == nd.c[rows - 1] || hb == nd.f[rows - 1]));
                if (mf && ws == 0xFFFFFFFFu) ws = b + (u32)__builtin_ctzll(mf);
                if (ml) we = b + 64u - (u32)__builtin_clzll(ml);
            }
            if (ws == 0xFFFFFFFFu) ws = 0;
        } else { ws = win[2 * j]; we = win[2 * j + 1]; }
        const u32 sp = ws ? ws - 1 : 0;
        const bool include_exact = sp == 0 && we == L;
        const bool include_prefix = sp == 0;
        const u32 m = we - sp;
        const u8* th = hay + sp;  // trimmed haystack
        u32 score = 0;
        u32 npos = 0;  // TRACE: positions written for this record (lane 0)
        u32* const posv = TRACE ? trace.pos + (size_t)opos * trace.stride : nullptr;
        if (m > FZB_MAX_HAYSTACK_LEN) {
            if (lane == 0) {
                if (TRACE) {
                    // match_greedy's positions, shifted by the trim offset and reversed (algo/mod.rs:55-72); None => no positions
                    u32* fwd = cells;
                    bool matched = false;
                    score = greedy_score(nd, th, m, include_prefix, fwd, &matched);
                    const u32 n = (u32)nd.nbytes;
                    if (matched)
                        for (u32 k = 0; k < n; k++) posv[k] = fwd[n - 1 - k] + sp;
                    npos = matched ? n : 0;
                } else {
                    score = greedy_score(nd, th, m, include_prefix);
                }
            }
            score = __shfl(score, 0);
        } else if (m > 0 && rows > 0) {
            const u32 nchunks = (m + SWL - 1) / SWL;
            u32 maxs = 0;
            bool prev_last_lower = false, prev_last_delim = false;  // previous chunk's last lane (ascii.rs:55-56, 78, 95)
            u32 prev_cont = 0, prev_sm = 0;                         // unicode: previous chunk's continuation-gex / scalar-start vectors
            // the chunk's bytes are requested one chunk ahead: a window of ten chunks is otherwise a chain of ten exposed load latencies (the
            // longest window of a short queue IS the kernel's duration: 385 windows of the Arabic-shaped list, 15 us)
            u32 nx0 = (active && (u32)lane < m) ? th[lane] : 0, nx1 = 0, nx2 = 0, nx3 = 0;
            if (UNICODE) {
                nx1 = (active && lane + 1u < m) ? th[lane + 1] : 0;
                nx2 = (active && lane + 2u < m) ? th[lane + 2] : 0;
                nx3 = (active && lane + 3u < m) ? th[lane + 3] : 0;
            }
            for (u32 ch = 0; ch < nchunks; ch++) {
                const u32 base = ch * SWL;
                const u32 pos = base + lane;
                const u32 b0 = nx0;
                const u32 pb1 = nx1, pb2 = nx2, pb3 = nx3;
                if (ch + 1 < nchunks) {
                    const u32 np = pos + SWL;
                    nx0 = (active && np < m) ? th[np] : 0;
                    if (UNICODE) {
                        nx1 = (active && np + 1 < m) ? th[np + 1] : 0;
                        nx2 = (active && np + 2 < m) ? th[np + 2] : 0;
                        nx3 = (active && np + 3 < m) ? th[np + 3] : 0;
                    }
                }
                const bool lower = b0 >= 'a' && b0 <= 'z', upper = b0 >= 'A' && b0 <= 'Z', digit = b0 >= '0' && b0 <= '9';
                const bool delim = !(lower || upper || digit || b0 > 127);
                bool pl = __shfl_up((int)lower, 1), pd = __shfl_up((int)delim, 1);
                if (lane == 0) { pl = prev_last_lower; pd = prev_last_delim; }
                const u32 cap = (upper && pl) ? capb : 0, dl = (pd && !delim) ? delimb : 0;
                u32 bonus = (dl + cap) & LM;
                bonus = (bonus + ((ch == 0 && lane == 0 && include_prefix) ? prefixb : 0)) & LM;
                bonus = (bonus + Mc) & LM;
                prev_last_lower = __shfl((int)lower, SWL - 1);
                prev_last_delim = __shfl((int)delim, SWL - 1);

                // unicode per-chunk vectors (unicode.rs:81-89, 244-273)
                u32 b1 = 0, b2 = 0, b3 = 0;
                bool sstart = false;
                u32 contgex = 0, smask = 0;
                u32 cont_k[6], sm_k[6];
                if (UNICODE) {
                    b1 = pb1;
                    b2 = pb2;
                    b3 = pb3;
                    const bool valid = active && pos < m;
                    const bool cont = b0 > 0x7f && b0 < 0xc0 && valid;
                    sstart = !cont && valid;
                    contgex = cont ? gex : 0;
                    smask = sstart ? LM : 0;
                    // the per-step (continuation count, scalar-start-crossed) vectors do not depend on the needle
                    // row: unicode_gap.rs:141-166 `prepare_next_unicode_gap_step`, evaluated once per chunk
                    u32 c = contgex, ac = prev_cont, sm = smask, asm_ = prev_sm;
                    int k = 0;
                    for (int sh = 1; sh < SWL; sh *= 2, k++) {
                        cont_k[k] = c;
                        sm_k[k] = sm;
                        if (sh * 2 < SWL) {
                            const u32 sc = shift_pad_reg<SWL>(c, sh, ac, lane);
                            c = (c + sc) & LM;
                            u32 sac = __shfl_up(ac, sh);
                            if (lane < sh) sac = 0;
                            ac = (ac + sac) & LM;
                            const u32 ssm = shift_pad_reg<SWL>(sm, sh, asm_, lane);
                            sm = max(sm, ssm);
                            u32 sasm = __shfl_up(asm_, sh);
                            if (lane < sh) sasm = 0;
                            asm_ = max(asm_, sasm);
                        }
                    }
                }

                u32 prev_row = 0, up_mm = 0;
                u32 carry_last = 0;  // S(row-1, previous chunk)[SWL-1], captured before that vector is overwritten
                u32 row = 0;
                for (u32 r = 1; r <= rows; r++) {
                    bool mm, ex;
                    if (UNICODE) {
                        const u32 cl = nd.ulen[r - 1];
                        const u8* uc = nd.uc[r - 1];
                        const u8* uf = nd.uf[r - 1];
                        const u32 lastb = cl == 1 ? b0 : cl == 2 ? b1 : cl == 3 ? b2 : b3;
                        bool e = sstart && lastb == uc[cl - 1];
                        bool fl = sstart && lastb == uf[cl - 1];
                        if (cl > 1) { e = e && b0 == uc[0]; fl = fl && b0 == uf[0]; }
                        if (cl > 2) { e = e && b1 == uc[1]; fl = fl && b1 == uf[1]; }
                        if (cl > 3) { e = e && b2 == uc[2]; fl = fl && b2 == uf[2]; }
                        ex = e;
                        mm = e || fl;
                    } else {
                        ex = b0 == nd.c[r - 1];
                        mm = ex || b0 == nd.f[r - 1];
                    }
                    // diagonal (ascii.rs:118-127 / unicode.rs:165-175)
                    u32 dsrc = __shfl_up(prev_row, 1);
                    if (lane == 0) dsrc = carry_last;
                    u32 diag = (dsrc + (mm ? bonus : 0)) & LM;
                    diag = subs(diag, X);
                    diag = (diag + (ex ? caseb : 0)) & LM;
                    // up (ascii.rs:130-133 / unicode.rs:178-182)
                    u32 up = subs(subs(prev_row, gex), up_mm ? gopm : 0);
                    if (UNICODE) {
                        if (!sstart) { diag = 0; up = 0; }
                    }
                    row = max(diag, up);
                    const u16* adj = ch ? adj_row_base + (size_t)r * SWL : nullptr;
                    const u16* adja = ch ? adj_aux_base + (size_t)r * SWL : nullptr;
                    // capture S(r, prev chunk)[SWL-1] for the next row's diagonal before overwriting
                    const u32 next_carry = ch ? adj_load<LONG>(&adj[SWL - 1]) : 0u;
                    u32 aux;  // ASCII: this row's match mask (0 / LM); unicode: pending gap-open mask
                    if (!UNICODE) {
                        // propagate_horizontal_gaps (ascii_gap.rs:11-105)
                        const u32 mmv = mm ? LM : 0;
                        u32 kg = gex;
                        for (int sh = 1; sh < SWL; sh *= 2) {
                            const u32 srow = shift_pad<SWL, LONG>(row, sh, adj, lane);
                            const u32 smm = shift_pad<SWL, LONG>(mmv, sh, adja, lane);
                            const u32 pen = (kg + (gopm & smm)) & LM;
                            row = max(row, subs(srow, pen));
                            kg = (kg + kg) & LM;
                        }
                        aux = mmv;
                    } else {
                        // propagate_horizontal_unicode_gaps (unicode_gap.rs:110-236)
                        u32 pending = mm ? LM : 0;
                        u32 tot = gex;
                        int k = 0;
                        for (int sh = 1; sh < SWL; sh *= 2, k++) {
                            const u32 srow = shift_pad<SWL, LONG>(row, sh, adj, lane);
                            const u32 spend = shift_pad<SWL, LONG>(pending, sh, adja, lane);
                            const u32 sgex = subs(tot, cont_k[k]);
                            const u32 crossed = spend & sm_k[k];
                            const u32 pen = (sgex + (gopm & crossed)) & LM;
                            row = max(row, subs(srow, pen));
                            pending = max(pending, subs(spend, sm_k[k]));
                            tot = (tot + tot) & LM;
                        }
                        aux = pending;
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (active && ch + 1 < nchunks) {
                        adj_row_base[(size_t)r * SWL + lane] = (u16)row;
                        adj_aux_base[(size_t)r * SWL + lane] = (u16)aux;
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (TRACE) {
                        if (active) cells[(size_t)r * TRACE_W + base + SWL + lane] = row | (mm ? 0x10000u : 0u);
                    }
                    carry_last = next_carry;
                    prev_row = row;
                    up_mm = mm ? 1 : 0;
                }
                if (active) maxs = max(maxs, row);
                if (UNICODE) { prev_cont = contgex; prev_sm = smask; }
                if (LONG) {  // this chunk's vectors are read by other lanes in the next chunk: make the slab writes visible first
                    __threadfence();
                    __builtin_amdgcn_wave_barrier();
                }
            }
            // horizontal max
            for (int off = 32; off > 0; off >>= 1) maxs = max(maxs, (u32)__shfl_xor(maxs, off));
            score = maxs;
            if (TRACE && score != 0) {
                // first column of the last row holding the score (alignment_iter.rs:52-66).  The slot is reused from haystack to
                // haystack, so every read of it goes around the vector L1 (agent-scope loads) after a fence.
                __threadfence();
                __builtin_amdgcn_wave_barrier();
                u32 col = 0xFFFFFFFFu;
                for (u32 ch = 0; ch < nchunks && col == 0xFFFFFFFFu; ch++) {
                    const u32 v = active ? (__hip_atomic_load(&cells[(size_t)rows * TRACE_W + (ch + 1) * SWL + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFFFFu) : 0u;
                    const u64 hit = __ballot(active && v == score);
                    if (hit) col = (ch + 1) * SWL + (u32)__builtin_ctzll(hit);
                }
                if (lane == 0 && col != 0xFFFFFFFFu) {
                    auto cell = [&](u32 r, u32 c) -> u32 {
                        if (r == 0 || c < (u32)SWL) return 0u;  // row 0 and the zero chunk
                        return __hip_atomic_load(&cells[(size_t)r * TRACE_W + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    };
                    const int mt = nd.max_typos;
                    u32 r = rows, typos = 0, sc = score, prev = 0xFFFFFFFFu;
                    for (;;) {
                        if (r == 0) break;
                        if (mt >= 0 && typos > (u32)mt) break;
                        if (col < (u32)SWL || sc == 0) break;  // at the left edge (only moves up remain) or lost the alignment
                        const u32 hidx = col - SWL;
                        if (UNICODE && hidx < m && (th[hidx] & 0xC0) == 0x80) {  // continuation byte: walk left
                            col--;
                            sc = cell(r, col) & 0xFFFFu;
                            continue;
                        }
                        if (cell(r, col) >> 16) {
                            const u32 p = hidx + sp;
                            if (UNICODE) {
                                if (prev != p) {
                                    for (int off = (int)nd.ulen[r - 1] - 1; off >= 0; off--)
                                        if (npos < trace.stride) posv[npos++] = p + (u32)off;
                                    prev = p;
                                }
                            } else if (npos < trace.stride) {
                                posv[npos++] = p;
                            }
                            r--;
                            col--;
                            sc = cell(r, col) & 0xFFFFu;
                            continue;
                        }
                        const u32 dg = cell(r - 1, col - 1) & 0xFFFFu, lf = cell(r, col - 1) & 0xFFFFu, upv = cell(r - 1, col) & 0xFFFFu;
                        if (dg >= lf && dg >= upv) { r--; col--; typos++; sc = dg; }
                        else if (lf >= upv) { col--; sc = lf; }
                        else { typos++; r--; sc = upv; }
                    }
                }
            }
        }
        if (lane == 0) {
            bool exact = include_exact && m == (u32)nd.nbytes;
            if (exact)
                for (u32 k = 0; k < m; k++) exact = exact && th[k] == nd.raw[k];
            if (exact) score = (score + nd.exact_bonus) & 0xFFFF;
            fzb_match_rec rec;
            rec.index = index_offset + li;
            rec.score = (u16)score;
            rec.exact = exact ? 1 : 0;
            rec.valid = 0;
            out[opos] = rec;
            if (TRACE) trace.npos[opos] = npos;
        }
    }
}

// dynamic LDS of the by-value-needle form: per wave two vectors (row, match / pending mask) of sw_lanes u16 per needle row (+ the zero row)
static size_t generic_lds_bytes(const NeedleDev& nd, int sw_lanes) { return (size_t)GEN_WAVES * 2 * (size_t)(nd.rows + 1) * (size_t)sw_lanes * sizeof(u16); }

void fzb_launch_generic(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win, int wmode, const u32* list, const u32* n_list_ptr,
                        const NeedleDev& nd, int sw_lanes, int unicode, fzb_match_rec* out, u32 capacity, u32* dev_count, u32* counters, int grid, hipStream_t st,
                        int list_forward, u32 only_below, const u32* alt_list, const u32* alt_count) {
    const TraceArgs none{nullptr, nullptr, nullptr, 0};
    const GateArgs gate{only_below ? n_list_ptr : nullptr, 0u, only_below, list_forward, only_below ? alt_list : nullptr, alt_count};
    const size_t lds = generic_lds_bytes(nd, sw_lanes);
#define FZB_K2C(SWL, U, ET) hipLaunchKernelGGL((k2c_generic<SWL, U, false>), dim3(grid), dim3(GEN_WAVES * 64), lds, st, c.bytes, EndsAny{c.ends, c.ends_u64}, first, index_offset, items, win, wmode, list, n_list_ptr, nd, out, capacity, dev_count, counters, none, (u16*)nullptr, gate)
#define FZB_K2C_ET(SWL, U) FZB_K2C(SWL, U, )
#define FZB_K2C_U(SWL) do { if (unicode) FZB_K2C_ET(SWL, true); else FZB_K2C_ET(SWL, false); } while (0)
    switch (sw_lanes) {
        case 64: FZB_K2C_U(64); break;
        case 32: FZB_K2C_U(32); break;
        case 16: FZB_K2C_U(16); break;
        default: FZB_K2C_U(8); break;
    }
#undef FZB_K2C
}

// dwords of trace scratch the traced launch needs for `grid` blocks
size_t fzb_trace_scratch_words(const NeedleDev& nd, int grid) { return (size_t)grid * GEN_WAVES * (size_t)(nd.rows + 1) * TRACE_W; }

// the matched-indices form: direct mode over (items, win); record j at out[j], its positions at pos[j * stride ..], npos[j] of them
void fzb_launch_generic_trace(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win, int wmode, const u32* n_items_ptr, const NeedleDev& nd,
                              int sw_lanes, int unicode, fzb_match_rec* out, u32 capacity, u32* dev_count, u32* counters, u32* cells, u32* pos, u32* npos, u32 stride,
                              int grid, hipStream_t st) {
    const TraceArgs tr{cells, pos, npos, stride};
    const size_t lds = generic_lds_bytes(nd, sw_lanes);
#define FZB_K2C(SWL, U, ET) hipLaunchKernelGGL((k2c_generic<SWL, U, true>), dim3(grid), dim3(GEN_WAVES * 64), lds, st, c.bytes, EndsAny{c.ends, c.ends_u64}, first, index_offset, items, win, wmode, (const u32*)nullptr, n_items_ptr, nd, out, capacity, dev_count, counters, tr, (u16*)nullptr, GateArgs{nullptr, 0u, 0u, 0, nullptr, nullptr})
    switch (sw_lanes) {
        case 64: FZB_K2C_U(64); break;
        case 32: FZB_K2C_U(32); break;
        case 16: FZB_K2C_U(16); break;
        default: FZB_K2C_U(8); break;
    }
#undef FZB_K2C
}

// ---- long needles (NeedleLongDev): the same kernel, needle arrays and the per-row previous-chunk vectors in global memory.  Direct mode
// only (the generic scorer is the ONLY scorer of a long needle: every window, any width, greedy beyond 1024 bytes) ---------------------
size_t fzb_generic_long_adj_bytes(const NeedleLongDev& nd, int sw_lanes, int grid) { return (size_t)grid * GEN_WAVES * 2 * (size_t)(nd.rows + 1) * (size_t)sw_lanes * sizeof(u16); }
size_t fzb_trace_scratch_words_long(const NeedleLongDev& nd, int grid) { return (size_t)grid * GEN_WAVES * (size_t)(nd.rows + 1) * TRACE_W; }

// list (optional): the launch walks a queue of (output position, window start, window end, haystack) entries, *n_items_ptr of them, read
// upwards - what k2d_dp_long leaves for it (windows beyond 1024 bytes); the records' counters are then the queueing kernel's business
void fzb_launch_generic_long(const CorpusDev& c, u64 first, u32 index_offset, const u32* items, const u32* win, int wmode, const u32* n_items_ptr, const NeedleLongDev& nd,
                             int sw_lanes, fzb_match_rec* out, u32 capacity, u32* dev_count, u32* counters, u16* adj, const u32* cells, u32* pos, u32* npos, u32 stride, int grid,
                             hipStream_t st, const u32* list) {
    const TraceArgs tr{(u32*)cells, pos, npos, stride};
    const bool trace = cells != nullptr;
    // the previous-chunk vectors in LDS when they fit (60 KB per four-wave workgroup: up to 127 rows at 32 lanes), in the slab otherwise
    const size_t lds = (size_t)GEN_WAVES * 2 * (size_t)(nd.rows + 1) * (size_t)sw_lanes * sizeof(u16);
    const bool in_lds = lds <= (size_t)60 * 1024;
#define FZB_K2C_LS(SWL, U, T, ET, SLAB) hipLaunchKernelGGL((k2c_generic<SWL, U, T, NeedleLongDev, SLAB>), dim3(grid), dim3(GEN_WAVES * 64), SLAB ? 0 : lds, st, c.bytes, EndsAny{c.ends, c.ends_u64}, first, index_offset, items, win, wmode, list, n_items_ptr, nd, out, capacity, dev_count, counters, tr, adj, GateArgs{nullptr, 0u, 0u, 1, nullptr, nullptr})
#define FZB_K2C_L(SWL, U, T, ET) do { if (in_lds) FZB_K2C_LS(SWL, U, T, ET, false); else FZB_K2C_LS(SWL, U, T, ET, true); } while (0)
#define FZB_K2C_L_ET(SWL, U, T) FZB_K2C_L(SWL, U, T, )
#define FZB_K2C_L_T(SWL, U) do { if (trace) FZB_K2C_L_ET(SWL, U, true); else FZB_K2C_L_ET(SWL, U, false); } while (0)
#define FZB_K2C_L_U(SWL) do { if (nd.unicode) FZB_K2C_L_T(SWL, true); else FZB_K2C_L_T(SWL, false); } while (0)
    switch (sw_lanes) {
        case 64: FZB_K2C_L_U(64); break;
        case 32: FZB_K2C_L_U(32); break;
        case 16: FZB_K2C_L_U(16); break;
        default: FZB_K2C_L_U(8); break;
    }
#undef FZB_K2C_L
#undef FZB_K2C_LS
}
