"""What a needle change costs on a resident corpus (the C2 list): fzb_matcher_set_pattern alone, the first fzb_match_list behind it (the tables travel
to the device with it), and the query after that.  Usage on a GPU box: python tools/bench_set_pattern.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch, synth, frizbee_amd as F
dev = torch.device("cuda", 0)
n = 10_000_000
rows, ends = synth.fixed_corpus(b"deadbe", n, 32, device=dev)
data = rows.cpu().numpy().reshape(-1)
cp = F.Corpus(packed=(data, ends))
for typos in (0, 1):
    m = F.Matcher("deadbe", F.Config(max_typos=typos, pf_lanes=64, sw_lanes=64))
    m.match_list(cp)
    needles = ("dead", "deadb", "deadbe") * 6
    tsp, tml, tss = [], [], []
    for nd in needles:
        t0 = time.perf_counter(); m.set_pattern(nd); t1 = time.perf_counter(); r = m.match_list(cp, copy=False); t2 = time.perf_counter()
        r = m.match_list(cp, copy=False); t3 = time.perf_counter()
        tsp.append(t1 - t0); tml.append(t2 - t1); tss.append(t3 - t2)
    med = lambda a: sorted(a)[len(a) // 2] * 1e6
    print(f"typos={typos}: set_pattern {med(tsp):.1f} us, first match_list after it {med(tml):.1f} us, next match_list {med(tss):.1f} us", flush=True)
