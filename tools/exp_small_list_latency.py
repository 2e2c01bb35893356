import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, synth, frizbee_amd as F
dev = torch.device("cuda", 0)
def probe(name, needle, cfg, cp, n, steps=200):
    m = F.Matcher(needle, cfg)
    out = torch.zeros(n * 8 + 64, dtype=torch.uint8, device=dev); cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    for _ in range(5): m.match_list_device(cp, out.data_ptr(), n, cnt.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): m.match_list_device(cp, out.data_ptr(), n, cnt.data_ptr())
    t_enq = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / steps
    # one query at a time (latency of a lone query: enqueue + GPU + sync)
    lat = []
    for _ in range(50):
        t1 = time.perf_counter(); m.match_list_device(cp, out.data_ptr(), n, cnt.data_ptr()); torch.cuda.synchronize(); lat.append(time.perf_counter() - t1)
    m.set_profiling(True)
    for _ in range(20): m.match_list_device(cp, out.data_ptr(), n, cnt.data_ptr())
    torch.cuda.synchronize(); tm = m.last_timings_ms()
    print(json.dumps(dict(config=name, enqueue_us=t_enq * 1e6, step_us=t_all * 1e6, lone_query_us=sorted(lat)[len(lat) // 2] * 1e6, gpu_pipeline_us=tm["total"] * 1e3)), flush=True)
for npaths in (20_000, 100_000, 300_000):
    data, ends = synth.paths_corpus(b"linux", npaths, device=dev)
    cp = F.Corpus(packed=(data, ends))
    probe(f"paths {npaths//1000}k typos0", "linux", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64), cp, npaths)
    probe(f"paths {npaths//1000}k 1 typo", "linux", F.Config(max_typos=1, pf_lanes=64, sw_lanes=64), cp, npaths)
data, ends = synth.arabic_corpus()
cp = F.Corpus(packed=(data, ends))
probe("arabic default", "إن", F.Config(pf_lanes=64, sw_lanes=64), cp, len(ends))
