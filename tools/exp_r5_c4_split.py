"""Would ONE C4 query run faster as two (or four) half-range pipelines on two streams?  The machinery exists as the sharded query with all
shards on one device (pull form: per-shard clone + stream, one concatenation kernel, one sort, one D2H): ordered records on the host per call,
against fzb_match_list on the unsharded list (same sort, same D2H)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, synth, frizbee_amd as F
dev = torch.device("cuda", 0)
n4 = int(os.environ.get("EXP_N", 12_500_000))
data, ends = synth.ragged_corpus(b"deadbeef", n4, device=dev)
cfg = F.Config(max_typos=0, pf_lanes=64, sw_lanes=64)
def med(fn, k=15):
    fn(); fn()
    ts = []
    for _ in range(k):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3, r
cp = F.Corpus(packed=(data, ends))
m = F.Matcher("deadbeef", cfg)
t_un, r_un = med(lambda: m.match_list(cp, copy=False))
want = r_un.tobytes()
print(json.dumps(dict(exp="C4 shard, unsharded fzb_match_list (ordered records on the host)", ms=t_un, records=int(len(r_un)))), flush=True)
del cp
for k in (2, 3, 4):
    for by_bytes in (False, True):
        sc = F.ShardedCorpus(packed=(data, ends), ndev=k, oversubscribe=True, by_bytes=by_bytes)
        mk = F.Matcher("deadbeef", cfg)
        t_k, r_k = med(lambda: mk.match_list_parallel_sharded(sc, copy=False))
        print(json.dumps(dict(exp=f"C4 shard as {k} shards on this one GPU ({'byte' if by_bytes else 'count'}-balanced), pull form", ms=t_k, vs_unsharded=t_k / t_un, equal=bool(r_k.tobytes() == want))), flush=True)
        del sc, mk
