"""Round 6: why do the streaming filter (HBM-bound) and the scorers (VALU-issue-bound) not run side by side, and what does?

Part 1 (C2 pieces, as tools/exp_overlap.py): A = a filter-only query (10 M x 32 B, nothing survives), B = a scorer-only query (0.5 M haystacks, all
match).  Alone, together on two plain streams, with B on a HIGH-PRIORITY stream (hipStreamCreateWithPriority), and on a CU-masked stream pair
(hipExtStreamCreateWithCUMask: A on the first `k` CUs of every XCD-interleaved numbering, B on the rest).
Part 2 (whole queries): ONE query cut into K tile-aligned sub-ranges whose pipelines are STAGGERED - sub-range k+1's filter waits for sub-range
k's filter (fzb_debug_set_gate), so that it runs beside sub-range k's compaction + scorers instead of beside another filter.  C2 and the C4 shard.
Run it plain for the times, under `rocprofv3 --kernel-trace` (EXP_TRACE=1: few iterations) for per-dispatch start / end.
"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, synth, frizbee_amd as F
dev = torch.device("cuda", 0)
TRACE = os.environ.get("EXP_TRACE") == "1"
PARTS = os.environ.get("EXP_PARTS", "1,2").split(",")
hip = C.CDLL("libamdhip64.so")
def hipchk(rc, what):
    if rc != 0: raise RuntimeError(f"{what}: hip error {rc}")
def stream_plain():
    s = C.c_void_p(); hipchk(hip.hipStreamCreateWithFlags(C.byref(s), C.c_uint(1)), "hipStreamCreateWithFlags"); return s.value
def stream_prio(p):
    s = C.c_void_p(); hipchk(hip.hipStreamCreateWithPriority(C.byref(s), C.c_uint(1), C.c_int(p)), "hipStreamCreateWithPriority"); return s.value
def stream_mask(words):
    arr = (C.c_uint32 * len(words))(*words); s = C.c_void_p()
    hipchk(hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(len(words)), arr), "hipExtStreamCreateWithCUMask"); return s.value
def event():
    e = C.c_void_p(); hipchk(hip.hipEventCreateWithFlags(C.byref(e), C.c_uint(2)), "hipEventCreateWithFlags"); return e.value  # 2 = hipEventDisableTiming
lo, hi = C.c_int(), C.c_int()
hipchk(hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi)), "hipDeviceGetStreamPriorityRange")
print(json.dumps(dict(stream_priority_range=dict(least=lo.value, greatest=hi.value), cus=torch.cuda.get_device_properties(0).multi_processor_count)), flush=True)

def timed(fn, iters=40):
    if TRACE: iters = 3
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return round(sorted(ts)[len(ts) // 2] * 1e6, 1)

cfg = F.Config(max_typos=0, pf_lanes=64, sw_lanes=64)
def corpus32(n, full, partial, seed=12345):
    flat = torch.zeros(n * 32 + 256, dtype=torch.uint8, device=dev)
    flat[: n * 32].view(n, 32).copy_(synth.make_rows(b"deadbe", n, 32, seed=seed, device=dev, full=full, partial=partial))
    ends = (torch.arange(1, n + 1, dtype=torch.int64, device=dev) * 32).to(torch.int32)
    return F.Corpus.from_device(flat.data_ptr(), ends.data_ptr(), n, flat.numel(), keep=(flat, ends), max_len=32, uniform_len=32)

def mask_pair(n_a, total=256):
    """CU masks for a stream pair: bit i = CU i of the runtime's numbering.  A takes n_a/total of every 32-CU word (so both masks spread over all XCDs)."""
    per = 32 * n_a // total
    wa = [(1 << per) - 1] * (total // 32); wb = [0xFFFFFFFF ^ ((1 << per) - 1)] * (total // 32)
    return wa, wb

if "1" in PARTS:
    nA, nB = 10_000_000, 500_000
    cA, cB = corpus32(nA, 0.0, 0.25), corpus32(nB, 1.0, 0.0)
    mA, mB = F.Matcher("deadbe", cfg), F.Matcher("deadbe", cfg)
    outA = torch.zeros(nA * 8 + 64, dtype=torch.uint8, device=dev); cntA = torch.zeros(4, dtype=torch.int32, device=dev)
    outB = torch.zeros(nA * 8 + 64, dtype=torch.uint8, device=dev); cntB = torch.zeros(4, dtype=torch.int32, device=dev)
    def qA(s): mA.match_list_device(cA, outA.data_ptr(), nA, cntA.data_ptr(), stream=s)
    def qB(s): mB.match_list_device(cB, outB.data_ptr(), nA, cntB.data_ptr(), stream=s)
    s1, s2 = stream_plain(), stream_plain()
    base = dict(exp="C2 pieces on two plain streams", A_alone_us=timed(lambda: qA(s1)), B_alone_us=timed(lambda: qB(s2)), A_then_B_us=timed(lambda: (qA(s1), qB(s2))), B_then_A_us=timed(lambda: (qB(s2), qA(s1))))
    print(json.dumps(base), flush=True)
    sh = stream_prio(hi.value); sl = stream_prio(lo.value)
    print(json.dumps(dict(exp="B (scorer) on the high-priority stream, A (filter) on the low-priority one", A_then_B_us=timed(lambda: (qA(sl), qB(sh))), B_then_A_us=timed(lambda: (qB(sh), qA(sl))))), flush=True)
    print(json.dumps(dict(exp="A (filter) on the high-priority stream, B (scorer) on the low-priority one", A_then_B_us=timed(lambda: (qA(sh), qB(sl))), B_then_A_us=timed(lambda: (qB(sl), qA(sh))))), flush=True)
    for wgs in (4, 2):
        os.environ["FZB_DFA_WGS"] = str(wgs); F.lib().fzb_debug_reload_knobs()
        print(json.dumps(dict(exp=f"filter at {wgs} workgroups per CU, scorer on the high-priority stream", A_alone_us=timed(lambda: qA(sl)), A_then_B_us=timed(lambda: (qA(sl), qB(sh))), B_then_A_us=timed(lambda: (qB(sh), qA(sl))))), flush=True)
    os.environ.pop("FZB_DFA_WGS"); F.lib().fzb_debug_reload_knobs()
    for n_a in (64, 96, 128, 160, 192):
        try:
            wa, wb = mask_pair(n_a)
            sa, sb = stream_mask(wa), stream_mask(wb)
        except Exception as e:
            print(json.dumps(dict(exp="CU-masked stream pair", error=str(e))), flush=True); break
        print(json.dumps(dict(exp=f"CU masks: filter on {n_a} CUs, scorer on {256 - n_a}", A_alone_us=timed(lambda: qA(sa)), B_alone_us=timed(lambda: qB(sb)), A_then_B_us=timed(lambda: (qA(sa), qB(sb))), B_then_A_us=timed(lambda: (qB(sb), qA(sa))))), flush=True)
    del cA, cB, mA, mB, outA, outB

def staggered(name, corpus, n, needle, cfgq, ks=(1, 2, 3, 4, 6, 8), streams=2, prio=False):
    """One query over `corpus` as K tile-aligned sub-ranges, each its own matcher (workspace) and output slice, alternating over `streams` streams;
    sub-range k+1's filter waits for sub-range k's."""
    out = torch.zeros(n * 8 + 64, dtype=torch.uint8, device=dev)
    ss = [stream_prio(hi.value if (prio and i % 2) else lo.value) if prio else stream_plain() for i in range(streams)]
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
        res[f"k{k}_records"] = int(sum(int(c[0].item()) for c in cnts))
        del ms
    print(json.dumps(res), flush=True)

if "2" in PARTS:
    nC = 10_000_000
    cC = corpus32(nC, 0.05, 0.20)
    staggered("C2 as K staggered sub-ranges, 2 plain streams", cC, nC, "deadbe", cfg)
    staggered("C2 as K staggered sub-ranges, 3 plain streams", cC, nC, "deadbe", cfg, ks=(3, 4, 6), streams=3)
    staggered("C2, max_typos 2 (C3), 2 plain streams", cC, nC, "deadbe", F.Config(max_typos=2, pf_lanes=64, sw_lanes=64), ks=(1, 2, 4))
    del cC
    n4 = int(os.environ.get("EXP_N4", 12_500_000))
    data, ends = synth.ragged_corpus(b"deadbeef", n4, device=dev)
    cp = F.Corpus(packed=(data, ends))
    staggered("C4 shard as K staggered sub-ranges, 2 plain streams", cp, n4, "deadbeef", cfg)
    staggered("C4 shard as K staggered sub-ranges, 3 plain streams", cp, n4, "deadbeef", cfg, ks=(3, 4, 6, 8), streams=3)
    del cp
    data5, ends5 = None, None
