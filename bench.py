#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X frizbee backend (BASELINE.json `metric`).

A "step" is one pass of the hot path (filter -> lane-exact prefilter -> Smith-Waterman -> index-ordered Match
records in HBM) over one synthetic haystack list that is already resident in HBM.  Workload at every N:
BASELINE.json configs[1] - needle "deadbe" (6 chars) vs 10,000,000 x 32-byte ASCII haystacks per GPU, max_typos=0,
seed 12345, reference "Partial Match" mix (5 % full / 20 % partial / 75 % none).  N > 1 is weak scaling: each rank
owns a contiguous 10M-item shard of a 10M*N list (global index offset), scores it with no data-path collective, then
each step's per-shard match list is gathered to rank 0 by RCCL (frizbee_amd/distributed.py ShardExchange: fixed-capacity
buffers, asynchronous and double-buffered, so step i's gather overlaps step i+1's kernels; every gather has completed
when the closing barrier + synchronize returns).

Prints ONE JSON line on rank 0 (see the driver contract): value = haystacks scored per second, whole job.
  roofline     : dominant HBM-bound kernel = the streaming filter (k1_dfa); achieved = algorithmic bytes per launch
                 (sum len + 4 B end offset per haystack + 1 bit decision) / its average duration over the timed steps,
                 measured with HIP events recorded on the launch stream by the library (fzb_last_timings).
  cpu_baseline : the CPU oracle (a C++ port of the reference, oracle/) running match_list_parallel on all host cores over
                 a bounded sample of the same list, rank 0 / N=1 only.  A reported baseline, not the optimisation target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

NEEDLE = b"deadbe"
HAY_LEN = 32
PER_GPU = 10_000_000
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def cpu_baseline(rows_dev, n_sample, max_typos):
    """Oracle (C++ restatement of the reference CPU path) timed on the host cores.  Checker code, used here only as the
    reported baseline.  Built for this host: with AVX-512 BW/VL/VBMI its lane vectors are zmm registers at the widths of
    the reference's AVX-512 backend (64 x u8 / 32 x u16, 64-lane prefilter), otherwise the portable lane loops."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    native = True
    try:
        O.build(native=True, force=True)  # -march=native on the GPU box's host
        simd = O.simd_kind(True)
    except Exception:
        native, simd = False, O.simd_kind(False)
    cores = os.cpu_count() or 1
    data = np.concatenate([rows_dev[:n_sample].reshape(-1).cpu().numpy(), np.zeros(64, np.uint8)])
    ends = np.arange(1, n_sample + 1, dtype=np.uint64) * np.uint64(HAY_LEN)
    m = O.Matcher(NEEDLE.decode(), lanes=(64, 64, 32), native=native, max_typos=max_typos)
    # The worker loop of match_list_parallel (src/matcher/parallel.rs:43-64: per-call worker threads, 2048-item chunks off an
    # atomic counter) WITHOUT its per-thread sort and single-threaded k-way merge - the same scope as one GPU step, which
    # leaves index-ordered records and does not sort either.  Workers are spawned per call, so the best thread count is
    # not necessarily every hardware thread: try a few, report the fastest.
    best, best_threads, reps = float("inf"), cores, 0
    for threads in sorted({cores, max(cores // 2, 1), max(cores // 4, 1)}, reverse=True):
        m.score_count_unordered(data, ends, threads)  # warm-up
        t_end = time.time() + 6.0
        k = 0
        while k < 3 or (time.time() < t_end and k < 400):
            t0 = time.perf_counter()
            m.score_count_unordered(data, ends, threads)
            dt = time.perf_counter() - t0
            if dt < best:
                best, best_threads = dt, threads
            k += 1
        reps += k
    cores = best_threads
    return {"value": n_sample / best, "unit": "haystacks/s", "cores": cores, "kind": "port", "simd": simd,
            "sample": f"first {n_sample} of the {PER_GPU} len-{HAY_LEN} haystacks, match_list_parallel's scoring loop without the ordering step, {cores} threads (fastest of all / half / quarter of the {os.cpu_count()} hardware threads), best of {reps} runs; "
                      f"C++ restatement of the reference ({'AVX-512 lane vectors, 64 x u8' if simd == 'avx512' else 'portable lane loops'}, "
                      "g++ -O3 -march=native), not the Rust binary"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--per-gpu", type=int, default=PER_GPU)
    ap.add_argument("--max-typos", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    # FZB_BENCH_FORCE_DIST=1 runs the N > 1 code path (process group, exchange) with a single rank: a self-test of that
    # path on a 1-GPU box, not a benchmark configuration
    use_dist = world > 1 or os.environ.get("FZB_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import frizbee_amd as F
    import synth
    from frizbee_amd.distributed import ShardExchange, merge_shard_runs

    n = args.per_gpu
    # ---- synthetic shard, generated directly in HBM (padded-16 layout == back-to-back 32-byte rows) ----
    flat = torch.zeros(n * HAY_LEN + 256, dtype=torch.uint8, device=dev)
    rows = flat[: n * HAY_LEN].view(n, HAY_LEN)
    rows.copy_(synth.make_rows(NEEDLE, n, HAY_LEN, seed=12345 + rank, device=dev))
    ends = (torch.arange(1, n + 1, dtype=torch.int64, device=dev) * HAY_LEN).to(torch.int32)
    corpus = F.Corpus.from_device(flat.data_ptr(), ends.data_ptr(), n, flat.numel(), ends_are_u64=False, keep=(flat, ends), max_len=HAY_LEN)
    cfg = F.Config(max_typos=args.max_typos, pf_lanes=64, sw_lanes=64)  # bit-exact against the AVX-512 (VBMI) reference backend
    m = F.Matcher(NEEDLE.decode(), cfg)
    out = torch.zeros(n * 8 + 64, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    # a real (non-null) stream: everything the library launches (and its profiling events) goes onto this stream
    side = torch.cuda.Stream(dev)
    torch.cuda.synchronize(dev)
    torch.cuda.set_stream(side)
    stream = side.cuda_stream
    index_offset = rank * n

    ex = None
    if use_dist:
        # set-up (untimed): one synchronous pass sizes the exchange buffers every rank agrees on
        m.match_list_device(corpus, out.data_ptr(), n, cnt.data_ptr(), stream=stream, index_offset=index_offset)
        torch.cuda.synchronize(dev)
        ex = ShardExchange(ShardExchange.plan(int(cnt[0].item()), device=dev), dev)
    step_no = [0]

    def step():
        if ex is None:
            m.match_list_device(corpus, out.data_ptr(), n, cnt.data_ptr(), stream=stream, index_offset=index_offset)
            return
        # N > 1: count + records are written straight into this slot's exchange buffer, then gathered to rank 0 by RCCL
        # on its own stream while the next step's kernels run (double-buffered; no host synchronisation in the loop)
        slot = step_no[0] & 1
        step_no[0] += 1
        ex.wait(slot)
        m.match_list_device(corpus, ex.records_ptr(slot), ex.cap, ex.count_ptr(slot), stream=stream, index_offset=index_offset)
        ex.post(slot)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    # Kernel-level timing: the SAME K steps again, immediately after, with HIP events recorded by the library around the
    # filter kernel on the launch stream (kept out of the timed region above so the events cost nothing there;
    # rocprofv3 --kernel-trace of this command covers both passes).
    m.set_profiling(True)
    for _ in range(args.steps):
        step()
    fence()
    tm = m.last_timings_ms()
    m.set_profiling(False)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    gathered = None
    if ex is None:
        n_matches = int(cnt[0].item())
    else:
        last = (step_no[0] - 1) & 1
        runs = ex.collect(last)       # root: the per-shard runs of the last step; everyone: drains the exchange
        ex.collect(last ^ 1)
        n_matches = int(ex.send[last][:4].cpu().numpy().view(np.uint32)[0])
        if rank == 0:
            merged = merge_shard_runs(runs, F.SortStrategy.ScoreThenIndexAsc)  # the reference's combine step, once, outside the timed loop
            gathered = {"matches_all_shards": int(sum(len(r) for r in runs)), "merged_len": int(len(merged)), "exchange_capacity_records": ex.cap}
    counters = m.last_counters()
    if rank == 0:
        total = n * world
        # algorithmic bytes of one filter launch (this rank's shard): payload once + u32 end offset + 1 decision bit per haystack
        filt_bytes = n * HAY_LEN + 4 * n + n / 8
        filt_s = tm["filter"] * 1e-3
        achieved = filt_bytes / filt_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "latest_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("k1_filter_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "haystacks scored/sec (whole node) + achieved HBM GB/s, 6-char needle vs 10M len-32 haystacks",
            "value": total / (elapsed / args.steps),
            "unit": "haystacks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": f"needle 'deadbe' (6 chars) vs {n:,} x {HAY_LEN}-byte ASCII haystacks per GPU, max_typos={args.max_typos}, "
                                   "mix 5% full / 20% partial / 75% none, seed 12345 (BASELINE.json configs[1])",
                       "haystacks_per_gpu": n, "haystack_len": HAY_LEN, "max_typos": args.max_typos,
                       "emulated_reference_backend": "AVX-512 (prefilter 64 lanes, Smith-Waterman 64 x u8)",
                       "sharding": f"contiguous index ranges over {world} GPU(s); per step an asynchronous, double-buffered RCCL gather of the match records to rank 0" if world > 1 else "single GPU",
                       "matches_per_shard": n_matches, "filter_survivors": counters["filter_survivors"], "exchange": gathered},
            "roofline": {"bound": "hbm", "kernel": "k1_dfa", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "bytes_per_launch": filt_bytes, "avg_kernel_ms": tm["filter"], "launches_averaged": tm["calls"]},
            "device_pipeline_ms": tm["total"],
            "pipeline_algorithmic_GBps": (n * HAY_LEN + 4 * n + 8 * n_matches) / (tm["total"] * 1e-3) / 1e9,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(rows, min(n, 10_000_000), args.max_typos)
            res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
