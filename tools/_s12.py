import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, synth, frizbee_amd as F
dev = torch.device("cuda", 0)
def stages(name, needle, cfg, cp, n):
    m = F.Matcher(needle, cfg)
    out = torch.zeros(n * 8 + 64, dtype=torch.uint8, device=dev); cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    for _ in range(3): m.match_list_device(cp, out.data_ptr(), n, cnt.data_ptr())
    m.set_profiling(True)
    for _ in range(10): m.match_list_device(cp, out.data_ptr(), n, cnt.data_ptr())
    torch.cuda.synchronize()
    st = m.last_stage_timings_ms(); c = m.last_counters()
    print(json.dumps(dict(config=name, **{k: round(v * 1e3, 1) for k, v in st.items() if k != "calls"}, **c)), flush=True)
data, ends = synth.paths_corpus(b"linux", 1_406_941, device=dev); cp = F.Corpus(packed=(data, ends))
for mt in (1, 2, 3): stages(f"paths 1.4M {mt} typo(s)", "linux", F.Config(max_typos=mt, pf_lanes=64, sw_lanes=64), cp, 1_406_941)
del cp
data, ends = synth.arabic_corpus(); cp = F.Corpus(packed=(data, ends))
stages("arabic 1 typo", "إن", F.Config(max_typos=1, pf_lanes=64, sw_lanes=64), cp, len(ends))
stages("arabic 2 typos (إنما)", "إنما", F.Config(max_typos=2, pf_lanes=64, sw_lanes=64), cp, len(ends))
