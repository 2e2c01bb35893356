"""Round-5 experiments on the C4 shard (12.5 M ragged haystacks of 8..128 bytes, needle 'deadbeef'): what a step costs under the knobs that
move work between the HBM-bound view filter and the issue-bound scorers, and whether two C4 queries in flight overlap (filter of one beside
the scorers of the other) - the measurement a software-pipelined step would have to beat.  One JSON object per line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, synth, frizbee_amd as F
dev = torch.device("cuda", 0)
n4 = int(os.environ.get("EXP_N", 12_500_000))
data, ends = synth.ragged_corpus(b"deadbeef", n4, device=dev)
cp = F.Corpus(packed=(data, ends))
cfg = F.Config(max_typos=0, pf_lanes=64, sw_lanes=64)


def timed(m, out, cnt, steps=10, stream=0):
    for _ in range(3): m.match_list_device(cp, out.data_ptr(), n4, cnt.data_ptr(), stream=stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): m.match_list_device(cp, out.data_ptr(), n4, cnt.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def one(label, env):
    for k, v in env.items(): os.environ[k] = v
    F.lib().fzb_debug_reload_knobs()
    try:
        m = F.Matcher("deadbeef", cfg)
        out = torch.zeros(n4 * 8 + 64, dtype=torch.uint8, device=dev); cnt = torch.zeros(4, dtype=torch.int32, device=dev)
        ms = timed(m, out, cnt)
        m.set_profiling(True)
        for _ in range(10): m.match_list_device(cp, out.data_ptr(), n4, cnt.data_ptr())
        torch.cuda.synchronize(); st = m.last_stage_timings_ms()
        print(json.dumps(dict(exp=label, env=env, ms_per_step=ms, stages_ms={k: st[k] for k in ("filter", "compaction_and_window", "scorers", "total")}, matches=int(cnt[0].item()))), flush=True)
        del m, out, cnt
    finally:
        for k in env: os.environ.pop(k, None)
        F.lib().fzb_debug_reload_knobs()


which = sys.argv[1:] or ["knobs", "two"]
if "knobs" in which:
    one("default", {})
    one("handoff on (round 4's default)", {"FZB_HANDOFF": "1"})
    for w in ("4", "5", "8"): one(f"view filter at {w} workgroups per CU", {"FZB_VIEW_WGS": w})
    one("parked rows in the global slab (round 4's PMC profile predates the LDS parking)", {"FZB_PARK_LDS_KB": "0"})
    one("four scorer launches on two streams instead of k2_classes_all", {"FZB_SMALL_LIST": "0"})
    one("default again", {})
if "two" in which:
    ms_ = [F.Matcher("deadbeef", cfg) for _ in range(2)]
    outs = [torch.zeros(n4 * 8 + 64, dtype=torch.uint8, device=dev) for _ in range(2)]
    cnts = [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(2)]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]

    def pair(conc, reps=10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            for i in range(2):
                ms_[i].match_list_device(cp, outs[i].data_ptr(), n4, cnts[i].data_ptr(), stream=(streams[i] if conc else streams[0]).cuda_stream)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    for _ in range(2): pair(True, 3); pair(False, 3)
    print(json.dumps(dict(exp="two C4 queries: one stream vs two streams (ms per PAIR)", sequential=pair(False), concurrent=pair(True), sequential_again=pair(False))), flush=True)
    for w in ("4", "5"):
        os.environ["FZB_VIEW_WGS"] = w; F.lib().fzb_debug_reload_knobs()
        print(json.dumps(dict(exp=f"two C4 queries, view filter at {w} workgroups per CU (ms per PAIR)", sequential=pair(False), concurrent=pair(True))), flush=True)
    os.environ.pop("FZB_VIEW_WGS", None); F.lib().fzb_debug_reload_knobs()
