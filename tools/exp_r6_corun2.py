"""Round 6, co-run part 2: the kernel trace of part 1 (profiles/r06_corun.txt) shows that the two kernels never shared a CU - k1_dfa at 8 workgroups per CU
holds every wave slot (8 waves per SIMD), k2b_dp_short at 4 waves per SIMD x 128 VGPRs holds every register - so whichever was dispatched first ran
alone.  Here both grids are sized to SHARE a CU (filter w workgroups per CU = w waves per SIMD x 48 VGPRs, scorer d workgroups of two waves per CU = d/2
waves per SIMD x 128 VGPRs) and the pair is timed again; then the C2 step as K staggered sub-ranges with those grids."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, synth, frizbee_amd as F
dev = torch.device("cuda", 0)
TRACE = os.environ.get("EXP_TRACE") == "1"
hip = C.CDLL("libamdhip64.so")
def stream_plain():
    s = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(s), C.c_uint(1)) == 0; return s.value
def stream_prio(p):
    s = C.c_void_p(); assert hip.hipStreamCreateWithPriority(C.byref(s), C.c_uint(1), C.c_int(p)) == 0; return s.value
def event():
    e = C.c_void_p(); assert hip.hipEventCreateWithFlags(C.byref(e), C.c_uint(2)) == 0; return e.value
def timed(fn, iters=40):
    if TRACE: iters = 3
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return round(sorted(ts)[len(ts) // 2] * 1e6, 1)
def knobs(**kw):
    for k in ("FZB_DFA_WGS", "FZB_DP_WGS_PER_CU"): os.environ.pop(k, None)
    for k, v in kw.items():
        if v: os.environ[k] = str(v)
    F.lib().fzb_debug_reload_knobs()
cfg = F.Config(max_typos=0, pf_lanes=64, sw_lanes=64)
def corpus32(n, full, partial, seed=12345):
    flat = torch.zeros(n * 32 + 256, dtype=torch.uint8, device=dev)
    flat[: n * 32].view(n, 32).copy_(synth.make_rows(b"deadbe", n, 32, seed=seed, device=dev, full=full, partial=partial))
    ends = (torch.arange(1, n + 1, dtype=torch.int64, device=dev) * 32).to(torch.int32)
    return F.Corpus.from_device(flat.data_ptr(), ends.data_ptr(), n, flat.numel(), keep=(flat, ends), max_len=32, uniform_len=32)
nA, nB = 10_000_000, 500_000
cA, cB, cC = corpus32(nA, 0.0, 0.25), corpus32(nB, 1.0, 0.0), corpus32(nA, 0.05, 0.20)
mA, mB = F.Matcher("deadbe", cfg), F.Matcher("deadbe", cfg)
outA = torch.zeros(nA * 8 + 64, dtype=torch.uint8, device=dev); cntA = torch.zeros(4, dtype=torch.int32, device=dev)
outB = torch.zeros(nA * 8 + 64, dtype=torch.uint8, device=dev); cntB = torch.zeros(4, dtype=torch.int32, device=dev)
def qA(s): mA.match_list_device(cA, outA.data_ptr(), nA, cntA.data_ptr(), stream=s)
def qB(s): mB.match_list_device(cB, outB.data_ptr(), nA, cntB.data_ptr(), stream=s)
s1, s2 = stream_plain(), stream_plain()
lo, hi = C.c_int(), C.c_int(); hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))
sl, sh = stream_prio(lo.value), stream_prio(hi.value)
LEAN = os.environ.get("EXP_LEAN") == "1"  # the filter's low-instruction instantiation (uniform 32-byte list, 256-byte table pitch: 7.5 M instead of 10.7 M VALU instructions per launch)
if LEAN: os.environ["FZB_DFA_UNI32"] = "1"; os.environ["FZB_DFA_STRIDE256"] = "1"
for w, d in ((8, 0), (4, 4), (4, 3), (4, 2), (5, 3), (5, 2), (6, 2), (3, 4), (3, 5), (2, 6)):
    knobs(FZB_DFA_WGS=w, FZB_DP_WGS_PER_CU=d)
    print(json.dumps(dict(exp="pair with shared grids" + (", lean filter" if LEAN else ""), filter_wgs_per_cu=w, scorer_wgs_per_cu=d or 8, A_alone_us=timed(lambda: qA(s1)), B_alone_us=timed(lambda: qB(s2)),
                          A_then_B_us=timed(lambda: (qA(s1), qB(s2))), B_then_A_us=timed(lambda: (qB(s2), qA(s1))), B_hi_then_A_lo_us=timed(lambda: (qB(sh), qA(sl))))), flush=True)

def staggered(name, corpus, n, needle, cfgq, ks, streams=2):
    out = torch.zeros(n * 8 + 64, dtype=torch.uint8, device=dev)
    ss = [stream_plain() for i in range(streams)]
    res = dict(exp=name, streams=streams)
    for k in ks:
        ms = [F.Matcher(needle, cfgq) for _ in range(k)]
        cnts = [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(k)]
        evs = [event() for _ in range(k)]
        per = -(-(n // k) // 1024) * 1024
        for gated in ((False,) if k == 1 else (False, True)):
            for i, m in enumerate(ms):
                F.lib().fzb_debug_set_gate(m.h, evs[i - 1] if (gated and i > 0) else None, evs[i] if gated else None)
            def run():
                for i, m in enumerate(ms):
                    first = i * per
                    if first >= n: break
                    cnt = min(per, n - first)
                    m.match_list_device(corpus, out.data_ptr() + first * 8, cnt, cnts[i].data_ptr(), stream=ss[i % streams], first=first, count=cnt, index_offset=first)
            res[f"k{k}{'_staggered' if gated else ''}_us"] = timed(run, 30)
        del ms
    print(json.dumps(res), flush=True)
for w, d in ((8, 0), (4, 4), (4, 3), (5, 3), (4, 2), (3, 4)):
    knobs(FZB_DFA_WGS=w, FZB_DP_WGS_PER_CU=d)
    staggered(f"C2 as K sub-ranges on 2 streams, filter {w} / scorer {d or 8} workgroups per CU", cC, nA, "deadbe", cfg, ks=(1, 2, 3, 4, 6))
knobs()
