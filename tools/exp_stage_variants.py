"""Where the view filter's staging time goes: the C4 shard's pipeline with parts of the staging switched off (FZB_STAGE_DBG bits; results
meaningless for bits != 0) and with the handoff off altogether.  One corpus, knobs re-read between the runs."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, synth, frizbee_amd as F
dev = torch.device("cuda", 0)
n4 = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
data, ends = synth.ragged_corpus(b"deadbeef", n4, device=dev)
cp = F.Corpus(packed=(data, ends))
out = torch.zeros(n4 * 8 + 64, dtype=torch.uint8, device=dev); cnt = torch.zeros(4, dtype=torch.int32, device=dev)
VARIANTS = [{"FZB_NO_HANDOFF": "1"}, {}] + [{"FZB_STAGE_DBG": v} for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "2", "4", "3", "7"])]
for env in VARIANTS:
    for k in ("FZB_NO_HANDOFF", "FZB_STAGE_DBG"):
        os.environ.pop(k, None)
    os.environ.update(env)
    F.lib().fzb_debug_reload_knobs()
    m = F.Matcher("deadbeef", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64))
    for _ in range(3): m.match_list_device(cp, out.data_ptr(), n4, cnt.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): m.match_list_device(cp, out.data_ptr(), n4, cnt.data_ptr())
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10
    m.set_profiling(True)
    for _ in range(10): m.match_list_device(cp, out.data_ptr(), n4, cnt.data_ptr())
    torch.cuda.synchronize(); st = m.last_stage_timings_ms()
    print(json.dumps(dict(env=env, ms_per_step=round(wall * 1e3, 4), filter=round(st["filter"], 4), scorers=round(st["scorers"], 4), matches=int(cnt[0].item()))), flush=True)
    del m
