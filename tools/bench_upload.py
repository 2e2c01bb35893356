#!/usr/bin/env python3
"""fzb_corpus_upload timing (the cold path: raw bytes + offsets from pageable host memory into the device layout), per host-to-device
mode.  Each mode runs in a child process (the mode is read once per process): FZB_UPLOAD_MODE=direct|register|staged, FZB_UPLOAD_THREADS.
    python tools/bench_upload.py            # C2 list (10 M x 32 B) and one C4 shard (12.5 M ragged)
"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))


def child(which):
    import numpy as np, torch, synth, frizbee_amd as F
    dev = torch.device("cuda", 0)
    if which == "C2":
        n = 10_000_000
        rows, ends = synth.fixed_corpus(b"deadbe", n, 32, device=dev)
        data = rows.cpu().numpy().reshape(-1).copy()
    else:
        n = 12_500_000
        data, ends = synth.ragged_corpus(b"deadbeef", n, device=dev)
        data = np.ascontiguousarray(data); ends = np.ascontiguousarray(ends)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); cp = F.Corpus(packed=(data, ends)); ts.append(time.perf_counter() - t0)
        del cp
    nbytes = data.nbytes + ends.nbytes
    print(json.dumps(dict(list=which, mode=os.environ.get("FZB_UPLOAD_MODE", "direct"), threads=os.environ.get("FZB_UPLOAD_THREADS", "default"), haystacks=n, host_bytes=nbytes,
                          first_ms=ts[0] * 1e3, best_ms=min(ts[1:]) * 1e3, median_ms=sorted(ts[1:])[1] * 1e3, GBps_best=nbytes / min(ts[1:]) / 1e9)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for which in ("C2", "C4"):
            for mode, thr in (("direct", ""), ("register", ""), ("staged", "6"), ("staged", "12")):
                env = {**os.environ, "FZB_UPLOAD_MODE": mode}
                if thr: env["FZB_UPLOAD_THREADS"] = thr
                subprocess.run([sys.executable, os.path.abspath(__file__), "--child", which], env=env, timeout=600)
