// TEST INFRASTRUCTURE: compiles the DEVICE arithmetic of the Smith-Waterman bodies (frizbee_amd/csrc/dp_body.h, dp_cf.h, dp_cfm.h,
// dp_unicode.h) for the host through tests/kernel_host/shim, one "thread" at a time, so that tests can fuzz the exact code the GPU runs
// against the oracle without a GPU.  Built by tests/kernel_host_lib.py with ROCm's clang++ (-x c++); never part of the product.
#include <vector>
#include <thread>
#include "dp_cfm.h"
#include "dp_quad.h"
#include "dp_unicode.h"

static u16 sadd16(u32 a, u32 b) { return (u16)(a + b > 0xFFFF ? 0xFFFF : a + b); }
static u16 ssub16(u32 a, u32 b) { return (u16)(a > b ? a - b : 0); }

// the NeedleDev fields the ASCII scorers read, filled the way fzb_matcher_create fills them (frizbee_amd/csrc/host.hip)
static void fill_needle(NeedleDev& nd, const u8* needle, int n, int case_sensitive, const u16* sc) {
    memset(&nd, 0, sizeof(nd));
    nd.rows = n;
    nd.nbytes = n;
    nd.lane_mask = 0xFFFF;
    nd.match_plus_mismatch = sadd16(sc[0], sc[1]);
    nd.mismatch = sc[1];
    nd.gex = sc[3];
    nd.gopm = ssub16(sc[2], sc[3]);
    nd.prefix = sc[4];
    nd.capitalization = sc[5];
    nd.matching_case = sc[6];
    nd.exact_bonus = sc[7];
    nd.delimiter = sc[8];
    nd.match_score = sc[0];
    nd.gap_open = sc[2];
    for (int i = 0; i < n; i++) {
        const u8 c = needle[i];
        nd.raw[i] = c;
        nd.c[i] = c;
        nd.f[i] = case_sensitive ? c : (c >= 'a' && c <= 'z') ? (u8)(c - 32) : (c >= 'A' && c <= 'Z') ? (u8)(c + 32) : c;
    }
}

template <int SWL>
static int run(const NeedleDev& nd, const u8* hay, u32 m, int include_prefix, int form, int real, const u8* cls) {
    u32 hb[SWL / 4];
    u8 buf[SWL + 8] = {0};
    memcpy(buf, hay, m);
    load_window_mem<SWL / 4>(buf, m, hb);
    bool upper = false;
    for (int r = 0; r < nd.rows; r++) upper = upper || (nd.c[r] >= 'A' && nd.c[r] <= 'Z');
    constexpr int NW = SWL / 2;
#define FORMS(U)                                                                                                           \
    if (form == 0) return (int)dp_single_chunk<SWL, true, U>(nd, m, include_prefix, cls, hb);                              \
    if (form == 1) return (int)dp_single_chunk<SWL, false, U>(nd, m, include_prefix, cls, hb);                             \
    if (form == 2) return (int)dp_single_chunk<SWL, true, U, (SWL >= 16 ? SWL / 4 : SWL / 2)>(nd, m, include_prefix, cls, hb); \
    if (form == 4) {  /* table set-up (short-haystack kernel): SWL/2 computed lanes */                                    \
        CfTables tab;                                                                                                      \
        for (unsigned t = 0; t < 16; t++) { threadIdx.x = t; cf_build_tables<U>(nd, tab); }                                \
        threadIdx.x = 0;                                                                                                   \
        if (SWL >= 16) return (int)dp_single_chunk_cf_tab<SWL, U, (SWL >= 16 ? SWL / 4 : SWL / 2)>(nd, include_prefix, tab, hb); \
    }                                                                                                                      \
    if (form == 3) {                                                                                                       \
        if (real == NW) return (int)dp_single_chunk_cf<SWL, U, NW>(nd, include_prefix, cls, hb);                           \
        if (real == NW / 2) return (int)dp_single_chunk_cf<SWL, U, NW / 2>(nd, include_prefix, cls, hb);                   \
        if (NW >= 4 && real == 3 * NW / 4) return (int)dp_single_chunk_cf<SWL, U, (NW >= 4 ? 3 * NW / 4 : NW)>(nd, include_prefix, cls, hb); \
        if (NW >= 4 && real == NW / 4) return (int)dp_single_chunk_cf<SWL, U, (NW >= 4 ? NW / 4 : NW)>(nd, include_prefix, cls, hb); \
    }
    if (upper) { FORMS(true) } else { FORMS(false) }
#undef FORMS
    return -2;
}

// multi-chunk windows (swl < m <= 1024): form 5 = dp_multi_chunk (first form, biased scan), 6 = dp_multi_chunk_t (dp_cfm.h), 7 / 8 = the same
// with the last chunk's padding lanes in closed form (needle without NUL)
template <int SWL>
static int run_multi(const NeedleDev& nd, const u8* hay, u32 m, int include_prefix, int form, const u8* cls) {
    static u32 scratch[(FZB_MAX_ROWS + 1) * (SWL / 2) + 64];
    std::vector<u8> buf(m + 96, 0);
    memcpy(buf.data(), hay, m);
    bool upper = false;
    for (int r = 0; r < nd.rows; r++) upper = upper || (nd.c[r] >= 'A' && nd.c[r] <= 'Z');
    if (form == 5) return (int)dp_multi_chunk<SWL, true>(nd, buf.data(), m, include_prefix, cls, scratch, 1, 0);
    CfTables tab;
    for (unsigned t = 0; t < 16; t++) { threadIdx.x = t; if (upper) cf_build_tables<true>(nd, tab); else cf_build_tables<false>(nd, tab); }
    threadIdx.x = 0;
    u32 rp = (u32)(SWL / 2);  // the global slab's row pitch; forms 16 / 17 / 18 = 6 / 7 / 8 with the LDS layout's pitch (fzb_park_lds_dwords)
    if (form >= 16) { rp = (nd.lane_mask == 0xFF ? SWL / 8 : SWL / 4) + 1; form -= 10; }
    if (form == 7 || form == 8) {  // the last chunk's NUL lanes in closed form: dp_multi_chunk_tc with the narrowest class that holds the tail (7) / one class wider (8)
        const u32 tail = m - ((m + SWL - 1) / SWL - 1) * SWL;  // bytes in the last chunk, 1 ..= SWL
        int cls = (int)((tail - 1) / (SWL / 4)) + (form == 8 ? 1 : 0);
        if (cls > 3) cls = 3;
        return upper ? (int)dp_multi_chunk_tc<SWL, true>(nd, buf.data(), m, include_prefix, tab, scratch, 1, 0, rp, (u32)cls) : (int)dp_multi_chunk_tc<SWL, false>(nd, buf.data(), m, include_prefix, tab, scratch, 1, 0, rp, (u32)cls);
    }
    return upper ? (int)dp_multi_chunk_t<SWL, true>(nd, buf.data(), m, include_prefix, tab, scratch, 1, 0, rp) : (int)dp_multi_chunk_t<SWL, false>(nd, buf.data(), m, include_prefix, tab, scratch, 1, 0, rp);
}

// the unicode single-chunk scorer (dp_unicode.h): rows are needle scalars (bytes, flipped bytes, UTF-8 length per row)
template <int SWL>
static int run_unicode(const NeedleDev& nd, const u8* hay, u32 m, int include_prefix, int real, const u8* cls, int form) {
    std::vector<u8> buf(m + 96, 0);
    memcpy(buf.data(), hay, m);
    constexpr int NW = SWL / 2;
    if (form == 1 || form == 2) {  // the biased-throughout form (dp_unicode_single_chunk_t): 1 = as the kernel chooses, 2 = general steps forced
        constexpr int HR = SWL >= 16 ? SWL / 4 : SWL / 2;
        if (real == NW) {
            const bool utf8 = form == 1 && !unicode_has_cont_run4<SWL>(buf.data(), m);
            return utf8 ? (int)dp_unicode_single_chunk_t<SWL, NW, true>(nd, buf.data(), m, include_prefix, cls) : (int)dp_unicode_single_chunk_t<SWL, NW, false>(nd, buf.data(), m, include_prefix, cls);
        }
        if (SWL >= 16 && real == NW / 2) {
            const bool utf8 = form == 1 && !unicode_has_cont_run4<(SWL >= 16 ? SWL / 2 : SWL)>(buf.data(), m);
            return utf8 ? (int)dp_unicode_single_chunk_t<SWL, HR, true>(nd, buf.data(), m, include_prefix, cls) : (int)dp_unicode_single_chunk_t<SWL, HR, false>(nd, buf.data(), m, include_prefix, cls);
        }
        return -2;
    }
    if (real == NW) return (int)dp_unicode_single_chunk<SWL>(nd, buf.data(), m, include_prefix, cls);
    if (SWL >= 16 && real == NW / 2) return (int)dp_unicode_single_chunk<SWL, (SWL >= 16 ? SWL / 4 : SWL / 2)>(nd, buf.data(), m, include_prefix, cls);
    return -2;
}

// a unicode window wider than one chunk (swl < m <= 1024) through dp_unicode_multi_chunk (thread-per-haystack, rows parked per chunk)
// form 0 = dp_unicode_multi_chunk (first form); 1 = dp_unicode_multi_chunk_t with the UTF-8 shortcut where the window allows it (as the
// kernel chooses), 2 = its general steps
template <int SWL>
static int run_unicode_multi(const NeedleDev& nd, const u8* hay, u32 m, int include_prefix, const u8* cls, int form) {
    static u32 scratch[(FZB_MAX_ROWS + 1) * (SWL / 2) + 64];
    std::vector<u8> buf(m + 96, 0);
    memcpy(buf.data(), hay, m);
    if (form == 0) return (int)dp_unicode_multi_chunk<SWL>(nd, buf.data(), m, include_prefix != 0, cls, scratch, 1, 0);
    const bool utf8 = form == 1 && !unicode_window_has_cont_run4(buf.data(), m);
    return utf8 ? (int)dp_unicode_multi_chunk_t<SWL, true>(nd, buf.data(), m, include_prefix != 0, cls, scratch, 1, 0)
                : (int)dp_unicode_multi_chunk_t<SWL, false>(nd, buf.data(), m, include_prefix != 0, cls, scratch, 1, 0);
}

// the same two forms for a LONG needle (NeedleLongDev: the rows behind pointers, any number of them - k2d_dp_long): form 5 = dp_multi_chunk, 6 = dp_multi_chunk_t
template <int SWL>
static int run_multi_long(const NeedleLongDev& nd, bool upper, const u8* hay, u32 m, int include_prefix, int form, const u8* cls) {
    std::vector<u32> scratch((size_t)(nd.rows + 1) * (SWL / 2) + 64, 0xDEADBEEFu);
    std::vector<u8> buf(m + 96, 0);
    memcpy(buf.data(), hay, m);
    if (form == 5) return (int)dp_multi_chunk<SWL, true, NeedleLongDev>(nd, buf.data(), m, include_prefix, cls, scratch.data(), 1, 0);
    CfTables tab;
    for (unsigned t = 0; t < 16; t++) { threadIdx.x = t; if (upper) cf_build_tables<true, NeedleLongDev>(nd, tab); else cf_build_tables<false, NeedleLongDev>(nd, tab); }
    threadIdx.x = 0;
    const u32 rp = (u32)(SWL / 2);
    std::vector<u16> cf(nd.rows);  // what k2d_dp_long stages in LDS
    for (int r = 0; r < nd.rows; r++) cf[r] = (u16)(nd.c[r] | (nd.f[r] << 8));
    NeedleLongRows nr;
    static_cast<NeedleLongDev&>(nr) = nd;
    nr.cf = cf.data();
    return upper ? (int)dp_multi_chunk_t<SWL, true, SWL / 2, NeedleLongRows>(nr, buf.data(), m, include_prefix, tab, scratch.data(), 1, 0, rp)
                 : (int)dp_multi_chunk_t<SWL, false, SWL / 2, NeedleLongRows>(nr, buf.data(), m, include_prefix, tab, scratch.data(), 1, 0, rp);
}

// dp_quad.h: the window scored by four host threads in lockstep, one per quad lane (shim: fzb_host_update_dpp).  form 0 = row by row in registers
// (dp_quad_rows, MAXC chunks), 1 = chunk by chunk with the LDS layout of the parked rows, 2 = chunk by chunk with the slab layout (one slot,
// requested a row ahead).  Every lane returns the score; they must agree.
template <int SWL, typename ND>
static int run_quad(const ND& nd, bool upper, const u8* hay, u32 m, int include_prefix, int form) {
    constexpr int MAXC = 1024 / SWL;
    std::vector<u8> buf(m + 96, 0);
    memcpy(buf.data(), hay, m);
    CfTables tab;
    for (unsigned t = 0; t < 16; t++) { threadIdx.x = t; if (upper) cf_build_tables<true, ND>(nd, tab); else cf_build_tables<false, ND>(nd, tab); }
    threadIdx.x = 0;
    std::vector<u32> park((size_t)(nd.rows + 1) * QuadPark<SWL>::WORDS * 32 + 64, 0xDEADBEEFu);
    u32 res[4] = {0, 0, 0, 0};
    auto lane_fn = [&](int lane) {
        fzb_quad_lane = lane;
        threadIdx.x = (unsigned)lane << 2;
        u32 r;
        if (form == 0) r = upper ? dp_quad_rows<SWL, true, MAXC>(nd, buf.data(), m, include_prefix, tab) : dp_quad_rows<SWL, false, MAXC>(nd, buf.data(), m, include_prefix, tab);
        else if (form == 1) r = upper ? dp_quad_window<SWL, true, false>(nd, buf.data(), m, include_prefix, tab, park.data(), 0u) : dp_quad_window<SWL, false, false>(nd, buf.data(), m, include_prefix, tab, park.data(), 0u);
        else r = upper ? dp_quad_window<SWL, true, true>(nd, buf.data(), m, include_prefix, tab, park.data(), 1u) : dp_quad_window<SWL, false, true>(nd, buf.data(), m, include_prefix, tab, park.data(), 1u);
        res[lane] = r;
    };
    std::thread t1(lane_fn, 1), t2(lane_fn, 2), t3(lane_fn, 3);
    lane_fn(0);
    t1.join(); t2.join(); t3.join();
    fzb_quad_lane = 0;
    threadIdx.x = 0;
    if (res[0] != res[1] || res[0] != res[2] || res[0] != res[3]) return -3;
    return (int)res[0];
}

extern "C" {
// form: 0 = dp_single_chunk biased, 1 = literal (unbiased) scan, 2 = its padded-half form, 3 = dp_single_chunk_cf with `real` dwords,
// 4 = dp_single_chunk_cf_tab (LDS-table set-up, swl/4 dwords).
// Returns the score of `hay[0..m)` as ONE chunk of `swl` lanes (no exact bonus), or < 0 on a bad argument.
int kh_dp_single(const u8* needle, int n, int case_sensitive, const u16* scoring, const u8* hay, int m, int include_prefix, int swl, int form, int real) {
    if (n < 1 || n > FZB_MAX_ROWS || m < 1 || m > swl) return -1;
    NeedleDev nd;
    fill_needle(nd, needle, n, case_sensitive, scoring);
    static u8 cls[256];
    build_cls_table(cls);
    switch (swl) {
        case 64: return run<64>(nd, hay, (u32)m, include_prefix, form, real, cls);
        case 32: return run<32>(nd, hay, (u32)m, include_prefix, form, real, cls);
        case 16: return run<16>(nd, hay, (u32)m, include_prefix, form, real, cls);
        case 8: return run<8>(nd, hay, (u32)m, include_prefix, form, real, cls);
    }
    return -1;
}

int kh_dp_unicode(const u8* uc, const u8* uf, const u8* ulen, int rows, const u16* sc, const u8* hay, int m, int include_prefix, int swl, int real, int form) {
    if (rows < 1 || rows > FZB_MAX_ROWS || m < 0 || m > swl) return -1;
    NeedleDev nd;
    const u8 dummy[1] = {0};
    fill_needle(nd, dummy, 0, 1, sc);
    nd.rows = rows;
    nd.unicode = 1;
    for (int r = 0; r < rows; r++) {
        memcpy(nd.uc[r], uc + 4 * r, 4);
        memcpy(nd.uf[r], uf + 4 * r, 4);
        nd.ulen[r] = ulen[r];
    }
    static u8 cls[256];
    build_cls_table(cls);
    switch (swl) {
        case 64: return run_unicode<64>(nd, hay, (u32)m, include_prefix, real, cls, form);
        case 32: return run_unicode<32>(nd, hay, (u32)m, include_prefix, real, cls, form);
        case 16: return run_unicode<16>(nd, hay, (u32)m, include_prefix, real, cls, form);
        case 8: return run_unicode<8>(nd, hay, (u32)m, include_prefix, real, cls, form);
    }
    return -1;
}

// a haystack of at most 32 bytes through the short-corpus kernel's register path (dp_unicode.h): window from the two vectors
// (unicode_window_regs), window bytes shifted out of them (load_window_regs), set-up with the UTF-8 decision, rows.  out = {ws, we of the
// register search, ws, we of unicode_window_first_last, score of the register path, score of the memory path over the same window}
int kh_unicode_regs(const u8* uc, const u8* uf, const u8* ulen, int rows, const u16* sc, const u8* hay, int len, int swl, unsigned* out) {
    if (rows < 1 || rows > FZB_MAX_ROWS || len < 1 || len > 32 || len > swl / 2) return -1;
    NeedleDev nd;
    const u8 dummy[1] = {0};
    fill_needle(nd, dummy, 0, 1, sc);
    nd.rows = rows;
    nd.unicode = 1;
    for (int r = 0; r < rows; r++) {
        memcpy(nd.uc[r], uc + 4 * r, 4);
        memcpy(nd.uf[r], uf + 4 * r, 4);
        nd.ulen[r] = ulen[r];
    }
    static u8 cls[256];
    build_cls_table(cls);
    std::vector<u32> buf(64, 0);
    memcpy(buf.data(), hay, len);
    for (int i = len; i < 32; i++) ((u8*)buf.data())[i] = (u8)(0x41 + i);  // what follows a haystack inside its two vectors is not always zero
    if (len % 16) for (int i = len; i < (len + 15) / 16 * 16; i++) ((u8*)buf.data())[i] = 0;  // ... but its own 16-byte padding is
    uint4 q0, q1;
    memcpy(&q0, buf.data(), 16);
    memcpy(&q1, (const u8*)buf.data() + 16, 16);
    u32 ws = 0, we = 0, ws2 = 0, we2 = 0;
    unicode_window_regs(nd, q0, q1, (u32)len, ws, we);
    std::vector<u32> clean(64, 0);
    memcpy(clean.data(), hay, len);
    unicode_window_first_last(nd, (const u8*)clean.data(), (u32)len, ws2, we2);
    out[0] = ws; out[1] = we; out[2] = ws2; out[3] = we2;
    const u32 sp = ws ? ws - 1 : 0, m = we - sp;
    auto ident = [](bool b) { return b; };
    u32 s_regs = 0, s_mem = 0;
#define RUN_REGS(SWL)                                                                                                  \
    do {                                                                                                               \
        constexpr int HR = SWL >= 16 ? SWL / 4 : SWL / 2, RB = (HR + 1) / 2;                                           \
        u32 hw[RB], hb[RB + 1];                                                                                        \
        load_window_regs<RB>(q0, q1, sp, m, hw);                                                                       \
        for (int k = 0; k < RB; k++) hb[k] = hw[k];                                                                    \
        hb[RB] = 0;                                                                                                    \
        s_regs = dp_unicode_single_chunk_tr<SWL, HR>(nd, hb, m, sp == 0, cls, ident);                                  \
        const bool utf8 = !unicode_has_cont_run4<SWL>((const u8*)clean.data() + sp, m);                                \
        s_mem = utf8 ? dp_unicode_single_chunk_t<SWL, HR, true>(nd, (const u8*)clean.data() + sp, m, sp == 0, cls)    \
                     : dp_unicode_single_chunk_t<SWL, HR, false>(nd, (const u8*)clean.data() + sp, m, sp == 0, cls);  \
    } while (0)
    if (m >= 1 && we >= ws && m <= (u32)swl / 2) {
        if (swl == 64) RUN_REGS(64);
        else if (swl == 32) RUN_REGS(32);
        else if (swl == 16) RUN_REGS(16);
        else return -1;
    }
#undef RUN_REGS
    out[4] = s_regs; out[5] = s_mem;
    return 0;
}

// the 0-typo unicode window the unicode scorer computes itself (dp_unicode.h, unicode_window_first_last)
int kh_unicode_window(const u8* uc, const u8* uf, const u8* ulen, int rows, const u8* hay, int len, unsigned* out) {
    if (rows < 1 || rows > FZB_MAX_ROWS || len < 0) return -1;
    NeedleDev nd;
    memset(&nd, 0, sizeof(nd));
    nd.rows = rows;
    nd.unicode = 1;
    for (int r = 0; r < rows; r++) {
        memcpy(nd.uc[r], uc + 4 * r, 4);
        memcpy(nd.uf[r], uf + 4 * r, 4);
        nd.ulen[r] = ulen[r];
    }
    std::vector<u32> buf((len + 160 + 3) / 4 + 4, 0);  // 16-byte aligned start, the haystack's own padding zero as in the device layout ...
    memcpy(buf.data(), hay, len);
    for (int i = (len + 15) / 16 * 16; i < len + 96; i++) ((u8*)buf.data())[i] = hay[i % (len ? len : 1)];  // ... and behind it the next haystack (here: the same bytes again)
    u32 ws = 0, we = 0;
    unicode_window_first_last(nd, (const u8*)buf.data(), (u32)len, ws, we);
    out[0] = ws;
    out[1] = we;
    return 0;
}

int kh_dp_multi(const u8* needle, int n, int case_sensitive, int is_u8, const u16* scoring, const u8* hay, int m, int include_prefix, int swl, int form) {
    if (n < 1 || n > FZB_MAX_ROWS || m <= swl || m > 1024) return -1;
    NeedleDev nd;
    fill_needle(nd, needle, n, case_sensitive, scoring);
    nd.lane_mask = is_u8 ? 0xFF : 0xFFFF;
    static u8 cls[256];
    build_cls_table(cls);
    switch (swl) {
        case 64: return run_multi<64>(nd, hay, (u32)m, include_prefix, form, cls);
        case 32: return run_multi<32>(nd, hay, (u32)m, include_prefix, form, cls);
        case 16: return run_multi<16>(nd, hay, (u32)m, include_prefix, form, cls);
        case 8: return run_multi<8>(nd, hay, (u32)m, include_prefix, form, cls);
    }
    return -1;
}

int kh_dp_multi_long(const u8* needle, int n, int case_sensitive, int is_u8, const u16* sc, const u8* hay, int m, int include_prefix, int swl, int form) {
    if (n < 1 || m < 1 || m > 1024) return -1;
    std::vector<u8> c(n + 4), f(n + 4);
    bool upper = false;
    for (int i = 0; i < n; i++) {
        c[i] = needle[i];
        f[i] = case_sensitive ? c[i] : (c[i] >= 'a' && c[i] <= 'z') ? (u8)(c[i] - 32) : (c[i] >= 'A' && c[i] <= 'Z') ? (u8)(c[i] + 32) : c[i];
        upper = upper || (c[i] >= 'A' && c[i] <= 'Z');
    }
    NeedleLongDev nd;
    memset(&nd, 0, sizeof(nd));
    nd.rows = n; nd.nbytes = n; nd.lane_mask = is_u8 ? 0xFF : 0xFFFF;
    nd.match_plus_mismatch = sadd16(sc[0], sc[1]); nd.mismatch = sc[1]; nd.gex = sc[3]; nd.gopm = ssub16(sc[2], sc[3]);
    nd.prefix = sc[4]; nd.capitalization = sc[5]; nd.matching_case = sc[6]; nd.exact_bonus = sc[7]; nd.delimiter = sc[8];
    nd.match_score = sc[0]; nd.gap_open = sc[2];
    nd.raw = c.data(); nd.c = c.data(); nd.f = f.data();
    static u8 cls[256];
    build_cls_table(cls);
    switch (swl) {
        case 64: return run_multi_long<64>(nd, upper, hay, (u32)m, include_prefix, form, cls);
        case 32: return run_multi_long<32>(nd, upper, hay, (u32)m, include_prefix, form, cls);
        case 16: return run_multi_long<16>(nd, upper, hay, (u32)m, include_prefix, form, cls);
        case 8: return run_multi_long<8>(nd, upper, hay, (u32)m, include_prefix, form, cls);
    }
    return -1;
}

// four lanes per window: a NeedleDev needle (n <= 63) or, with `long_needle`, the same through NeedleLongRows (any n)
int kh_dp_quad(const u8* needle, int n, int case_sensitive, int is_u8, const u16* sc, const u8* hay, int m, int include_prefix, int swl, int form, int long_needle) {
    if (n < 1 || m < 1 || m > 1024 || (swl != 64 && swl != 32) || (!long_needle && n > FZB_MAX_ROWS)) return -1;
    bool upper = false;
    for (int i = 0; i < n; i++) upper = upper || (needle[i] >= 'A' && needle[i] <= 'Z');
    if (!long_needle) {
        NeedleDev nd;
        fill_needle(nd, needle, n, case_sensitive, sc);
        nd.lane_mask = is_u8 ? 0xFF : 0xFFFF;
        return swl == 64 ? run_quad<64>(nd, upper, hay, (u32)m, include_prefix, form) : run_quad<32>(nd, upper, hay, (u32)m, include_prefix, form);
    }
    std::vector<u8> c(n + 4), f(n + 4);
    std::vector<u16> cf(n);
    for (int i = 0; i < n; i++) {
        c[i] = needle[i];
        f[i] = case_sensitive ? c[i] : (c[i] >= 'a' && c[i] <= 'z') ? (u8)(c[i] - 32) : (c[i] >= 'A' && c[i] <= 'Z') ? (u8)(c[i] + 32) : c[i];
        cf[i] = (u16)(c[i] | (f[i] << 8));
    }
    NeedleLongRows nd;
    memset(&nd, 0, sizeof(nd));
    nd.rows = n; nd.nbytes = n; nd.lane_mask = is_u8 ? 0xFF : 0xFFFF;
    nd.match_plus_mismatch = sadd16(sc[0], sc[1]); nd.mismatch = sc[1]; nd.gex = sc[3]; nd.gopm = ssub16(sc[2], sc[3]);
    nd.prefix = sc[4]; nd.capitalization = sc[5]; nd.matching_case = sc[6]; nd.exact_bonus = sc[7]; nd.delimiter = sc[8];
    nd.match_score = sc[0]; nd.gap_open = sc[2];
    nd.raw = c.data(); nd.c = c.data(); nd.f = f.data(); nd.cf = cf.data();
    return swl == 64 ? run_quad<64>(nd, upper, hay, (u32)m, include_prefix, form) : run_quad<32>(nd, upper, hay, (u32)m, include_prefix, form);
}

int kh_dp_unicode_multi(const u8* uc, const u8* uf, const u8* ulen, int rows, int is_u8, const u16* sc, const u8* hay, int m, int include_prefix, int swl, int form) {
    if (rows < 1 || rows > FZB_MAX_ROWS || m <= swl || m > 1024) return -1;
    NeedleDev nd;
    const u8 dummy[1] = {0};
    fill_needle(nd, dummy, 0, 1, sc);
    nd.rows = rows;
    nd.unicode = 1;
    nd.lane_mask = is_u8 ? 0xFF : 0xFFFF;
    for (int r = 0; r < rows; r++) {
        memcpy(nd.uc[r], uc + 4 * r, 4);
        memcpy(nd.uf[r], uf + 4 * r, 4);
        nd.ulen[r] = ulen[r];
    }
    static u8 cls[256];
    build_cls_table(cls);
    switch (swl) {
        case 64: return run_unicode_multi<64>(nd, hay, (u32)m, include_prefix, cls, form);
        case 32: return run_unicode_multi<32>(nd, hay, (u32)m, include_prefix, cls, form);
        case 16: return run_unicode_multi<16>(nd, hay, (u32)m, include_prefix, cls, form);
        case 8: return run_unicode_multi<8>(nd, hay, (u32)m, include_prefix, cls, form);
    }
    return -1;
}

// the 0-typo window of a haystack of at most 32 bytes: the short kernel's merged-flag-word search against dp_body.h's (both forms
// of src/prefilter/algo/ascii.rs:6-72); out = {ws, we} of each
int kh_window(const u8* needle, int n, int case_sensitive, const u8* hay, int len, unsigned* out) {
    if (n < 1 || n > FZB_MAX_ROWS || len < 0 || len > 4096) return -1;
    static const u16 sc[9] = {12, 6, 5, 1, 12, 4, 4, 8, 4};
    NeedleDev nd;
    fill_needle(nd, needle, n, case_sensitive, sc);
    if (len > 32) {  // any length: the general search (window_first_last) only; out[2..3] repeat it.  Device layout: the haystack's own
        // padding to 16 is zero, behind it lies the next haystack (here: the same bytes again)
        std::vector<u32> buf((len + 160 + 3) / 4 + 4, 0);
        memcpy(buf.data(), hay, len);
        for (int i = (len + 15) / 16 * 16; i < len + 96; i++) ((u8*)buf.data())[i] = hay[i % len];
        window_first_last(nd, (const u8*)buf.data(), (u32)len, out[0], out[1]);
        out[2] = out[0];
        out[3] = out[1];
        return 0;
    }
    u32 w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    memcpy(w, hay, (size_t)len);
    const uint4 q0 = make_uint4(w[0], w[1], w[2], w[3]), q1 = make_uint4(w[4], w[5], w[6], w[7]);
    window_first_last_regs(nd, q0, q1, (u32)len, out[0], out[1]);
    cf_window_first_last_regs(nd, q0, q1, out[2], out[3]);
    return 0;
}

// the typo prefilter's window of an accepted haystack (<= 32 bytes) in the lane-free form the short kernel computes: out = {ws, we}
int kh_window_typos(const u8* needle, int n, int case_sensitive, int max_typos, const u8* hay, int len, unsigned* out) {
    if (n < 1 || n > FZB_MAX_ROWS || len < 0 || len > 32) return -1;
    static const u16 sc[9] = {12, 6, 5, 1, 12, 4, 4, 8, 4};
    NeedleDev nd;
    fill_needle(nd, needle, n, case_sensitive, sc);
    nd.max_typos = max_typos;
    static u8 fl[256];
    cf_build_typo_table(nd, fl);
    u32 w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    memcpy(w, hay, (size_t)len);
    cf_window_typos_regs(fl, make_uint4(w[0], w[1], w[2], w[3]), make_uint4(w[4], w[5], w[6], w[7]), (u32)len, out[0], out[1]);
    return 0;
}

// batch form for fuzzing: `count` windows packed back to back with byte lengths lens[i]; scores out
int kh_dp_batch(const u8* needle, int n, int case_sensitive, const u16* scoring, const u8* hays, const int* lens, int count, const u8* include_prefix, int swl, int form,
                int real, int* out) {
    size_t off = 0;
    for (int i = 0; i < count; i++) {
        out[i] = kh_dp_single(needle, n, case_sensitive, scoring, hays + off, lens[i], include_prefix[i], swl, form, real);
        off += (size_t)lens[i];
    }
    return 0;
}
}
