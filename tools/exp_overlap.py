"""Can the streaming filter (HBM-bound) and the short scorer (VALU-issue-bound) of the C2 pipeline run side by side on one GPU?
Two queries on two streams: A = the 10 M x 32 list with nothing to score (5 % Full removed: full=0), B = a 0.5 M-item list where every
haystack matches (the scorer's 0.5 M windows of the C2 step, almost no filter work).  Times A alone, B alone, and both enqueued back to
back; if the pair takes about max(A, B) the two stages overlap and a software-pipelined step (filter of chunk k+1 beside the scorer of
chunk k) is worth building; if it takes A + B it is not.  Also: the C2 step itself as 2 / 4 sub-ranges alternating between two streams."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, synth, frizbee_amd as F
dev = torch.device("cuda", 0)
def corpus(n, full, partial, seed=12345):
    flat = torch.zeros(n * 32 + 256, dtype=torch.uint8, device=dev)
    flat[: n * 32].view(n, 32).copy_(synth.make_rows(b"deadbe", n, 32, seed=seed, device=dev, full=full, partial=partial))
    ends = (torch.arange(1, n + 1, dtype=torch.int64, device=dev) * 32).to(torch.int32)
    return F.Corpus.from_device(flat.data_ptr(), ends.data_ptr(), n, flat.numel(), keep=(flat, ends), max_len=32, uniform_len=32)
cfg = F.Config(max_typos=0, pf_lanes=64, sw_lanes=64)
nA, nB = 10_000_000, 500_000
cA, cB, cC = corpus(nA, 0.0, 0.25), corpus(nB, 1.0, 0.0), corpus(nA, 0.05, 0.20)
mA, mB = F.Matcher("deadbe", cfg), F.Matcher("deadbe", cfg)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
outA = torch.zeros(nA * 8 + 64, dtype=torch.uint8, device=dev); cntA = torch.zeros(4, dtype=torch.int32, device=dev)
outB = torch.zeros(nA * 8 + 64, dtype=torch.uint8, device=dev); cntB = torch.zeros(4, dtype=torch.int32, device=dev)
def qA(): mA.match_list_device(cA, outA.data_ptr(), nA, cntA.data_ptr(), stream=sA.cuda_stream)
def qB(): mB.match_list_device(cB, outB.data_ptr(), nA, cntB.data_ptr(), stream=sB.cuda_stream)
def timed(fn, iters=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e6
for wgs in (8, 6, 5, 4, 3, 2):  # the filter's resident workgroups per CU: at 8 it holds every wave slot of the chip
    os.environ["FZB_DFA_WGS"] = str(wgs); F.lib().fzb_debug_reload_knobs()
    print(json.dumps(dict(filter_wgs_per_cu=wgs, A_alone_us=timed(qA), B_alone_us=timed(qB), A_then_B_us=timed(lambda: (qA(), qB())), B_then_A_us=timed(lambda: (qB(), qA())))), flush=True)
os.environ.pop("FZB_DFA_WGS"); F.lib().fzb_debug_reload_knobs()
res = dict(A_alone_us=timed(qA), B_alone_us=timed(qB), A_then_B_us=timed(lambda: (qA(), qB())), B_then_A_us=timed(lambda: (qB(), qA())),
           survivors_A=int(cntA[0].item()), survivors_B=int(cntB[0].item()))
print(json.dumps(res), flush=True)
# the C2 step as sub-ranges on two streams (each its own matcher and output slice; tile-aligned cuts)
m2 = [F.Matcher("deadbe", cfg) for _ in range(4)]
cnts = [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(4)]
def whole(): mA.match_list_device(cC, outA.data_ptr(), nA, cntA.data_ptr(), stream=sA.cuda_stream)
def parts(k):
    per = (nA // k) // 1024 * 1024
    def run():
        for i in range(k):
            first = i * per; cnt = per if i + 1 < k else nA - first
            st = (sA, sB)[i % 2]
            m2[i].match_list_device(cC, outB.data_ptr() + first * 8, cnt, cnts[i].data_ptr(), stream=st.cuda_stream, first=first, count=cnt, index_offset=first)
    return run
print(json.dumps(dict(c2_whole_us=timed(whole), c2_two_halves_two_streams_us=timed(parts(2)), c2_four_quarters_two_streams_us=timed(parts(4)),
                      c2_uneven_3_us=timed(parts(3)))), flush=True)
