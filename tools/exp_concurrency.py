"""Experiment: do two independent pipelines (HBM-bound filter + VALU-bound DP) overlap when issued on two HIP streams?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, synth, frizbee_amd as F
dev = torch.device("cuda", 0)
n, L = 10_000_000, 32
def mk(seed):
    flat = torch.zeros(n * L + 256, dtype=torch.uint8, device=dev)
    flat[: n * L].view(n, L).copy_(synth.make_rows(b"deadbe", n, L, seed=seed, device=dev))
    ends = (torch.arange(1, n + 1, dtype=torch.int64, device=dev) * L).to(torch.int32)
    return F.Corpus.from_device(flat.data_ptr(), ends.data_ptr(), n, flat.numel(), keep=(flat, ends), max_len=L)
cs = [mk(1), mk(2)]
ms = [F.Matcher("deadbe", F.Config(pf_lanes=64, sw_lanes=64)) for _ in range(2)]
outs = [torch.zeros(n * 8 + 64, dtype=torch.uint8, device=dev) for _ in range(2)]
cnts = [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
def run(conc, reps=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for i in range(2):
            st = streams[i] if conc else streams[0]
            ms[i].match_list_device(cs[i], outs[i].data_ptr(), n, cnts[i].data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for _ in range(3): run(True, 3); run(False, 3)
print("sequential (1 stream) ms per pair:", run(False))
print("concurrent (2 streams) ms per pair:", run(True))
print("sequential again:", run(False))
