"""Where the per-step cost of bench.py's N > 1 path comes from, measured with ONE rank on one GPU (RCCL process group of size 1):
the same loop as bench.py's step() - wait, pipeline, post - with the CPU-side enqueue time separated from the total, and the pipeline /
the exchange alone.  Usage: python tools/exp_dist_overhead.py [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
import torch, torch.distributed as dist
import synth, frizbee_amd as F
from frizbee_amd.distributed import ShardExchange

K = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n = 10_000_000
flat = torch.zeros(n * 32 + 256, dtype=torch.uint8, device=dev)
flat[: n * 32].view(n, 32).copy_(synth.make_rows(b"deadbe", n, 32, seed=12345, device=dev))
ends = (torch.arange(1, n + 1, dtype=torch.int64, device=dev) * 32).to(torch.int32)
corpus = F.Corpus.from_device(flat.data_ptr(), ends.data_ptr(), n, flat.numel(), ends_are_u64=False, keep=(flat, ends), max_len=32, uniform_len=32)
m = F.Matcher("deadbe", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64))
out = torch.zeros(n * 8 + 64, dtype=torch.uint8, device=dev)
cnt = torch.zeros(4, dtype=torch.int32, device=dev)
side = torch.cuda.Stream(dev)
torch.cuda.synchronize(dev)
torch.cuda.set_stream(side)
stream = side.cuda_stream
m.match_list_device(corpus, out.data_ptr(), n, cnt.data_ptr(), stream=stream)
torch.cuda.synchronize(dev)
ex = ShardExchange(ShardExchange.plan(int(cnt[0].item()), device=dev), dev)
for s_ in range(8):
    ex.wait(s_ & 1)
    m.match_list_device(corpus, ex.records_ptr(s_ & 1), ex.cap, ex.count_ptr(s_ & 1), stream=stream)
    ex.post(s_ & 1)
ex.collect(0); ex.collect(1)


def timed(fn, k=K):
    for i in range(5): fn(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(k): fn(i)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    return t_enq / k * 1e6, (time.perf_counter() - t0) / k * 1e6


def step_full(i):
    s = i & 1
    ex.wait(s)
    m.match_list_device(corpus, ex.records_ptr(s), ex.cap, ex.count_ptr(s), stream=stream)
    ex.post(s)


def step_pipeline(i):
    s = i & 1
    m.match_list_device(corpus, ex.records_ptr(s), ex.cap, ex.count_ptr(s), stream=stream)


def step_exchange(i):
    s = i & 1
    ex.wait(s)
    ex.post(s)


evs = [torch.cuda.Event(enable_timing=False) for _ in range(2)]
xs = torch.cuda.Stream(dev)


def step_pipeline_event(i):  # the pipeline + ONE event record on the scoring stream per step (what torch's process group does at a collective)
    s = i & 1
    m.match_list_device(corpus, ex.records_ptr(s), ex.cap, ex.count_ptr(s), stream=stream)
    evs[s].record(side)


def step_full_side_stream(i):  # the exchange issued from a second stream that waits for the scoring stream through our own event
    s = i & 1
    ex.wait(s)
    m.match_list_device(corpus, ex.records_ptr(s), ex.cap, ex.count_ptr(s), stream=stream)
    evs[s].record(side)
    with torch.cuda.stream(xs):
        xs.wait_event(evs[s])
        ex.post(s)


res = {}
for name, fn in (("pipeline_only", step_pipeline), ("pipeline_plus_event_record", step_pipeline_event), ("exchange_only", step_exchange), ("pipeline_and_exchange", step_full),
                 ("pipeline_and_exchange_from_side_stream", step_full_side_stream)):
    enq, tot = timed(fn)
    res[name] = {"cpu_enqueue_us_per_step": round(enq, 1), "total_us_per_step": round(tot, 1)}
    ex.collect(0); ex.collect(1)
print(json.dumps(res))

# ---- the ORDERED step (bench.py's e2e_sorted_merge): where its time goes, phase by phase (host clock, a synchronisation after each phase) ----
def phases(k=20):
    acc = {"pipeline": 0.0, "post_gather": 0.0, "wait_gather": 0.0, "merge_sort_d2h": 0.0}
    for i in range(k + 3):
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        m.match_list_device(corpus, ex.records_ptr(0), ex.cap, ex.count_ptr(0), stream=stream)
        torch.cuda.synchronize(dev); t1 = time.perf_counter()
        ex.post(0)
        t2 = time.perf_counter()
        ex.wait(0); torch.cuda.synchronize(dev); t3 = time.perf_counter()
        r = ex.collect_merged(0, m, stream=stream)
        t4 = time.perf_counter()
        if i >= 3:
            for key, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                acc[key] += v / k * 1e6
    return {k_: round(v, 1) for k_, v in acc.items()}, len(r)

ph, nrec = phases()
def ordered(i):
    m.match_list_device(corpus, ex.records_ptr(0), ex.cap, ex.count_ptr(0), stream=stream)
    ex.post(0)
    ex.collect_merged(0, m, stream=stream)
for i in range(3): ordered(i)
torch.cuda.synchronize(dev); t0 = time.perf_counter()
for i in range(20): ordered(i)
t_ord = (time.perf_counter() - t0) / 20 * 1e6
def single(i):
    return m.match_list(corpus, copy=False)
torch.cuda.set_stream(torch.cuda.default_stream(dev))
for i in range(3): single(i)
t0 = time.perf_counter()
for i in range(20): single(i)
t_single = (time.perf_counter() - t0) / 20 * 1e6
print(json.dumps({"ordered_step_us": round(t_ord, 1), "phases_us_with_a_sync_after_each": ph, "records": nrec, "single_gpu_match_list_us": round(t_single, 1)}))
dist.barrier()
dist.destroy_process_group()
