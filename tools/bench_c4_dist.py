#!/usr/bin/env python3
"""BASELINE.json configs[3] end to end: needle 'deadbeef' (8 chars) vs 100,000,000 mixed-length (8..128 byte) haystacks, max_typos=0,
sharded over the GPUs of one node.  Not the driver's bench (bench.py is): the runner for this configuration, 1 process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29541 tools/bench_c4_dist.py
    python tools/bench_c4_dist.py --gpus 8                    # the same: starts the 8 ranks itself (fails loudly with fewer GPUs)
    python tools/bench_c4_dist.py --total 12500000            # one GPU, one shard's worth

The list is defined globally (lengths: one seeded vector all ranks derive identically), cut into BYTE-balanced contiguous index ranges
(frizbee_amd.distributed.shard_ranges_by_bytes; reference shape: src/matcher/parallel.rs:35-87 - contiguous chunks with a global index
offset, per-run sort, k-way merge).  Each rank builds only its own shard, scores it with `index_offset` = its first index, sorts its run
on the device; the sorted runs are gathered to rank 0 (RCCL) and k-way merged there.  Timed per step: scoring only (records in HBM) and
end to end (ordered merged list on rank 0's host).  Checks: shard ranges partition the list, merged length == sum of the shard counts,
merged order == (score desc, index asc), every record's index lies in its shard's range, and the first --check items of every shard
equal the CPU oracle's records."""
import argparse, json, os, sys, time
import numpy as np, torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--total", type=int, default=100_000_000)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--check", type=int, default=200_000)
    ap.add_argument("--gpus", type=int, default=0, help="ranks to start when not launched under torch.distributed.run (0 = WORLD_SIZE or 1)")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X (no CPU fallback exists for the product path)")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:  # plain `python tools/bench_c4_dist.py --gpus N`: start the N ranks ourselves
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible on this node")
        import socket
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__), *sys.argv[1:]]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execvpe(cmd[0], cmd, os.environ)
    json_out = os.fdopen(os.dup(1), "w"); sys.stdout.flush(); os.dup2(2, 1)  # RCCL's banner (C stdio on fd 1) goes to stderr, the JSON line to the real stdout
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    if args.gpus and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or os.environ.get("FZB_C4_FORCE_DIST") == "1"  # the env var runs the exchange path with one rank (self-test)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import frizbee_amd as F, synth
    from frizbee_amd.distributed import ShardExchange, shard_ranges_by_bytes

    n = args.total
    g = torch.Generator(); g.manual_seed(12346)
    lengths = torch.randint(8, 129, (n,), generator=g)                      # the global list's lengths, identical on every rank
    ends_all = np.cumsum(lengths.numpy().astype(np.uint64), dtype=np.uint64)
    ranges = shard_ranges_by_bytes(ends_all, world)
    lo, hi = ranges[rank]
    cnt_items = hi - lo
    # this rank's haystacks (content seeded per shard; lengths from the global vector)
    rows = synth.make_rows(b"deadbeef", cnt_items, 128, lengths=lengths[lo:hi].to(dev), seed=12345 + rank, device=dev)
    mask = torch.arange(128, device=dev)[None, :] < lengths[lo:hi].to(dev)[:, None]
    packed = rows[mask].cpu().numpy(); del rows, mask
    ends = ends_all[lo:hi] - (ends_all[lo - 1] if lo else np.uint64(0))
    corpus = F.Corpus(packed=(packed, ends))
    m = F.Matcher("deadbeef", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64))
    out = torch.zeros(cnt_items * 8 + 64, dtype=torch.uint8, device=dev); cnt = torch.zeros(4, dtype=torch.int32, device=dev)

    def fence():
        if use_dist: dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(2): m.match_list_device(corpus, out.data_ptr(), cnt_items, cnt.data_ptr(), index_offset=lo)
    fence(); t0 = time.perf_counter()
    for _ in range(args.steps): m.match_list_device(corpus, out.data_ptr(), cnt_items, cnt.data_ptr(), index_offset=lo)
    fence(); t_score = (time.perf_counter() - t0) / args.steps
    k_local = int(cnt[0].item())
    # correctness of this shard's head against the CPU oracle (checker), index order
    import oracle_lib as O
    c = min(args.check, cnt_items)
    got = out[: k_local * 8].cpu().numpy().view(F.MATCH_DTYPE)
    want = O.Matcher("deadbeef", lanes=(64, 64, 32), max_typos=0, sort="IndexAsc").match_packed(np.concatenate([packed[: int(ends[c - 1])], np.zeros(64, np.uint8)]), ends[:c])
    want["index"] += np.uint32(lo)
    head = got[got["index"] < lo + c]
    shard_ok = head.tolist() == want.tolist() and bool(((got["index"] >= lo) & (got["index"] < hi)).all())
    # end to end: device sort per rank, gather of the sorted runs, k-way merge on the root
    res = {}
    if use_dist:
        ex = ShardExchange(ShardExchange.plan(k_local, margin=1.05, device=dev), dev)
        fence(); t0 = time.perf_counter()
        for _ in range(max(3, args.steps // 2)):
            m.match_list_sorted_device(corpus, ex.records_ptr(0), ex.cap, ex.count_ptr(0))
            ex.post(0)
            runs = ex.collect(0)
            if rank == 0:
                for r_, run_ in enumerate(runs): run_["index"] += np.uint32(ranges[r_][0])  # the sorted form numbers a shard from 0
                merged = F.k_merge_matches(F.SortStrategy.ScoreThenIndexAsc, runs)
        fence(); t_e2e = (time.perf_counter() - t0) / max(3, args.steps // 2)
    else:
        m2 = F.Matcher("deadbeef", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64))
        m2.match_list(corpus, copy=False)
        t0 = time.perf_counter()
        for _ in range(max(3, args.steps // 2)): merged = m2.match_list(corpus, copy=False)
        t_e2e = (time.perf_counter() - t0) / max(3, args.steps // 2)
        runs = [merged]
    oks = torch.tensor([int(shard_ok), k_local], dtype=torch.int64, device=dev)
    if use_dist:
        allv = [torch.zeros_like(oks) for _ in range(world)]; dist.all_gather(allv, oks)
    else:
        allv = [oks]
    t = torch.tensor([t_score, t_e2e], dtype=torch.float64, device=dev)
    if use_dist: dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        total_matches = int(sum(int(v[1]) for v in allv))
        key = (0xFFFF - merged["score"].astype(np.int64)) * (1 << 32) + merged["index"].astype(np.int64)
        ordered = bool((np.diff(key) > 0).all()) if len(key) > 1 else True
        sum_len = int(ends_all[-1])
        res = {"config": "C4: 'deadbeef' vs mixed-length 8..128 haystacks, max_typos=0", "haystacks": n, "n_gpus": world, "ranks_seen": (dist.get_world_size() if use_dist else 1), "bytes": sum_len,
               "shards": [{"range": list(r), "bytes": int(ends_all[r[1] - 1]) - (int(ends_all[r[0] - 1]) if r[0] else 0)} for r in ranges],
               "scoring_ms_per_step": float(t[0]) * 1e3, "haystacks_per_s_scoring": n / float(t[0]),
               "roofline_step_frac": (sum_len + 4 * n + 8 * total_matches) / float(t[0]) / 8e12 / world,
               "e2e_ordered_list_on_rank0_ms": float(t[1]) * 1e3, "matches": total_matches, "merged_len": int(len(merged)),
               "checks": {"every_shard_head_equals_oracle_and_indices_in_range": all(int(v[0]) == 1 for v in allv), "merged_len_equals_sum_of_shard_counts": int(len(merged)) == total_matches,
                          "merged_order_is_score_desc_then_index_asc": ordered, "oracle_items_per_shard": c}}
    if use_dist:
        dist.barrier(); dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res), file=json_out, flush=True)


if __name__ == "__main__":
    main()
