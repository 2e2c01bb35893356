"""The long-needle row of bench.py alone (80-byte needle vs 1 M haystacks of 100..200 bytes) + a 200-byte needle (slab form)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, synth, frizbee_amd as F
dev = torch.device("cuda", 0)
for nlen, lo, hi in ((80, 100, 200), (200, 250, 400)):
    nl = 1_000_000 if nlen == 80 else 200_000
    needle = bytes((b"abcdefghijklmnopqrstuvwxyz0123456789_-" * 8)[:nlen])
    g = torch.Generator(device=dev); g.manual_seed(99)
    lens = torch.randint(lo, hi + 1, (nl,), generator=g, device=dev)
    rows = synth.make_rows(needle, nl, hi, lengths=lens, seed=4242, device=dev, chunk=1 << 17)
    mask = torch.arange(hi, device=dev)[None, :] < lens[:, None]
    d, e = rows[mask].cpu().numpy(), np.cumsum(lens.cpu().numpy().astype(np.uint64), dtype=np.uint64)
    del rows, mask
    cp = F.Corpus(packed=(d, e))
    m = F.Matcher(needle.decode(), F.Config(max_typos=0, pf_lanes=64, sw_lanes=32))
    out = torch.zeros(nl * 8 + 64, dtype=torch.uint8, device=dev); cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    for _ in range(2): m.match_list_device(cp, out.data_ptr(), nl, cnt.data_ptr())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): m.match_list_device(cp, out.data_ptr(), nl, cnt.data_ptr())
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 3
    print(json.dumps(dict(needle_bytes=nlen, haystacks=nl, ms_per_step=round(wall * 1e3, 3), matches=int(cnt[0].item()))), flush=True)
    del cp, m
