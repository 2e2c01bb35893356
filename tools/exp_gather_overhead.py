import os, sys, time, json
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy as np, torch, torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import frizbee_amd as F, synth
from frizbee_amd.distributed import ShardExchange
n = 10_000_000
flat = torch.zeros(n * 32 + 256, dtype=torch.uint8, device=dev)
flat[: n * 32].view(n, 32).copy_(synth.make_rows(b"deadbe", n, 32, seed=12345, device=dev))
ends = (torch.arange(1, n + 1, dtype=torch.int64, device=dev) * 32).to(torch.int32)
cp = F.Corpus.from_device(flat.data_ptr(), ends.data_ptr(), n, flat.numel(), keep=(flat, ends), max_len=32)
m = F.Matcher("deadbe", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64))
side = torch.cuda.Stream(dev); torch.cuda.set_stream(side); st = side.cuda_stream
ex = ShardExchange(628517, dev)
def run(mode, K=200):
    torch.cuda.synchronize(); t0 = time.perf_counter(); th = 0.0
    for i in range(K):
        slot = i & 1
        if mode >= 1: ex.wait(slot)
        m.match_list_device(cp, ex.records_ptr(slot), ex.cap, ex.count_ptr(slot), stream=st)
        if mode >= 2:
            a = time.perf_counter(); ex.post(slot); th += time.perf_counter() - a
    if mode >= 2: ex.wait(0); ex.wait(1)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e6, th / K * 1e6
for mode in (0, 1, 2, 0, 2):
    print(mode, run(mode))
# variant: gather only count-sized prefix? all_gather_into_tensor single op
buf = torch.zeros(8 + 628517 * 8, dtype=torch.uint8, device=dev); outb = torch.zeros_like(buf)
def run2(K=200):
    torch.cuda.synchronize(); t0 = time.perf_counter(); th = 0
    for i in range(K):
        m.match_list_device(cp, buf.data_ptr() + 8, 628517, buf.data_ptr(), stream=st)
        a = time.perf_counter(); w = dist.all_gather_into_tensor(outb, buf, async_op=True); th += time.perf_counter() - a
    w.wait(); torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e6, th / K * 1e6
print("all_gather_into_tensor", run2())
dist.destroy_process_group()
