"""Round 6, co-run part 3 (run under `rocprofv3 --kernel-trace`): which side of the pair suffers, and is it the memory stream or the filter's own
LDS / VALU work that the scorer collides with?  R = a plain streaming read of the filter's 320 MB (torch elementwise compare, 320 MB read + 80 MB written: HBM only, short-lived workgroups),
A = the filter-only query at 4 workgroups per CU, B = the scorer-only query at 4 workgroups per CU.  Pairs: A || B, R || B."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, synth, frizbee_amd as F
dev = torch.device("cuda", 0)
os.environ["FZB_DFA_WGS"] = "4"; os.environ["FZB_DP_WGS_PER_CU"] = "4"
cfg = F.Config(max_typos=0, pf_lanes=64, sw_lanes=64)
def corpus32(n, full, partial, seed=12345):
    flat = torch.zeros(n * 32 + 256, dtype=torch.uint8, device=dev)
    flat[: n * 32].view(n, 32).copy_(synth.make_rows(b"deadbe", n, 32, seed=seed, device=dev, full=full, partial=partial))
    ends = (torch.arange(1, n + 1, dtype=torch.int64, device=dev) * 32).to(torch.int32)
    return F.Corpus.from_device(flat.data_ptr(), ends.data_ptr(), n, flat.numel(), keep=(flat, ends), max_len=32, uniform_len=32), flat
nA, nB = 10_000_000, 500_000
(cA, flatA), (cB, _) = corpus32(nA, 0.0, 0.25), corpus32(nB, 1.0, 0.0)
mA, mB = F.Matcher("deadbe", cfg), F.Matcher("deadbe", cfg)
outA = torch.zeros(nA * 8 + 64, dtype=torch.uint8, device=dev); cntA = torch.zeros(4, dtype=torch.int32, device=dev)
outB = torch.zeros(nA * 8 + 64, dtype=torch.uint8, device=dev); cntB = torch.zeros(4, dtype=torch.int32, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
words = flatA[: nA * 32].view(torch.int32)
flags = torch.zeros(words.numel(), dtype=torch.bool, device=dev)
def qA(): mA.match_list_device(cA, outA.data_ptr(), nA, cntA.data_ptr(), stream=s1.cuda_stream)
def qB(): mB.match_list_device(cB, outB.data_ptr(), nA, cntB.data_ptr(), stream=s2.cuda_stream)
def qR():
    with torch.cuda.stream(s1): torch.gt(words, 0, out=flags)
def timed(fn, iters=12):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return round(sorted(ts)[len(ts) // 2] * 1e6, 1)
print(json.dumps(dict(A_alone_us=timed(qA), B_alone_us=timed(qB), R_alone_us=timed(qR), A_then_B_us=timed(lambda: (qA(), qB())), B_then_A_us=timed(lambda: (qB(), qA())),
                      R_then_B_us=timed(lambda: (qR(), qB())), B_then_R_us=timed(lambda: (qB(), qR())))), flush=True)
