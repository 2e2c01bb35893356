#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X frizbee backend (BASELINE.json `metric`).

A "step" is one pass of the hot path (streaming filter -> compaction -> [lane-exact prefilter] -> Smith-Waterman -> index-ordered
Match records in HBM) over one synthetic haystack list that is already resident in HBM.  Workload at every N:
BASELINE.json configs[1] - needle "deadbe" (6 chars) vs 10,000,000 x 32-byte ASCII haystacks per GPU, max_typos=0,
seed 12345, reference "Partial Match" mix (5 % full / 20 % partial / 75 % none).  N > 1 is weak scaling: each rank
owns a contiguous 10M-item shard of a 10M*N list (global index offset), scores it with no data-path collective, then
each step's per-shard match list is gathered to rank 0 by RCCL (frizbee_amd/distributed.py ShardExchange: fixed-capacity
buffers, asynchronous and double-buffered, so step i's gather overlaps step i+1's kernels; every gather has completed
when the closing barrier + synchronize returns).

Prints ONE JSON line on rank 0 (see the driver contract): `value` = haystacks scored per second, whole job.  Beside it:
  roofline        the dominant kernel = the streaming filter k1_dfa, HBM-bound: achieved = algorithmic bytes per launch
                  (sum len + 1 decision bit per haystack; the list has uniform length, so no end offsets are read) / its average duration over the profiled steps, measured
                  with HIP events recorded on the launch stream by the library (fzb_last_stage_timings).  `traffic` = HBM bytes per launch from the
                  FETCH_SIZE / WRITE_SIZE counters, collected by two rocprofv3 --pmc child runs of this script (`traffic_source` says so, or
                  names the stored profile if that failed or --no-live-traffic was given).
  roofline_step   the whole step against the same roofline: bytes the step needs (sum len + 8 M; SURVEY 8(d) adds 4 N of end offsets, which a
                  uniform-length list does not read - `frac_with_survey_bytes` uses that formula) / ms_per_step / 8 TB/s.
  stages          HIP-event averages per stage; the scorer is VALU-issue-bound, its `issue_frac` = wave-instructions of the committed
                  SQ profile x 4 cycles / (SIMDs x duration x clock) - stored counters, labelled as such.
  e2e             what a caller of `Matcher::match_list` gets: pipeline + device reverse/radix sort + D2H of the records (median).
  e2e_cold        the same from a list in pageable HOST memory: fzb_corpus_upload + a fresh matcher + first query + D2H (pack + H2D +
                  kernel + D2H + sort); never `value`.
  configs         the other BASELINE.json configurations on this GPU (C3 typos=2, one C4 shard, C5 unicode) and a realistic ragged list (the
                  Chromium-paths shape of the reference's own real-data benchmark): ms, step roofline, matches.
  check           the first 1,000,000 haystacks of the bench list scored by the CPU oracle (checker) outside every timed region and
                  compared record for record with what the GPU produced.
  cpu_baseline    the CPU oracle's match_list_parallel scoring loop (a C++ port of the reference with AVX-512 lane vectors) on the host
                  cores, rank 0 / N=1 only, medians; `linear_bound` = single-thread rate x physical cores, the most a per-call-threaded
                  CPU implementation could reach.  A reported baseline, not the optimisation target.
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
PUBLISHED_PER_THREAD = 1.15e8  # /root/reference BENCHMARKS.md:123 - match_list, len 32, partial match, one thread of a Ryzen 9950X3D


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


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
    hw_threads = os.cpu_count() or 1
    try:  # physical cores = distinct (package, core) pairs
        cores = set()
        for c in os.listdir("/sys/devices/system/cpu"):
            tp = f"/sys/devices/system/cpu/{c}/topology"
            if c.startswith("cpu") and c[3:].isdigit() and os.path.exists(tp + "/core_id"):
                cores.add((open(tp + "/physical_package_id").read().strip(), open(tp + "/core_id").read().strip()))
        phys = len(cores) or hw_threads
    except Exception:
        phys = hw_threads
    data = np.concatenate([rows_dev[:n_sample].reshape(-1).cpu().numpy(), np.zeros(64, np.uint8)])
    ends = np.arange(1, n_sample + 1, dtype=np.uint64) * np.uint64(HAY_LEN)
    m = O.Matcher(NEEDLE.decode(), lanes=(64, 64, 32), native=native, max_typos=max_typos)

    # The worker loop of match_list_parallel (src/matcher/parallel.rs:43-64: per-call worker threads, 2048-item chunks off an
    # atomic counter) WITHOUT its per-thread sort and single-threaded k-way merge - the same scope as one GPU step, which
    # leaves index-ordered records and does not sort either.  Workers are spawned per call (as the reference does), so on a
    # 3 ms job the best thread count is not every hardware thread: a few counts are tried, medians reported.
    def timed(threads, budget_s, min_reps):
        m.score_count_unordered(data, ends, threads)  # warm-up
        ts, t_end = [], time.time() + budget_s
        while len(ts) < min_reps or (time.time() < t_end and len(ts) < 200):
            t0 = time.perf_counter()
            m.score_count_unordered(data, ends, threads)
            ts.append(time.perf_counter() - t0)
        return _median(ts), len(ts)

    n1 = min(n_sample, 2_000_000)  # single thread: a 2M-item prefix is ~20-40 ms per run
    d1, e1 = np.concatenate([data[: n1 * HAY_LEN], np.zeros(64, np.uint8)]), ends[:n1]
    m.score_count_unordered(d1, e1, 1)
    t1s = []
    for _ in range(7):
        t0 = time.perf_counter()
        m.score_count_unordered(d1, e1, 1)
        t1s.append(time.perf_counter() - t0)
    single = n1 / _median(t1s)
    by_threads = {}
    for threads in sorted({hw_threads, phys, max(phys // 2, 1), max(phys // 4, 1)}, reverse=True):
        med, reps = timed(threads, 4.0, 5)
        by_threads[threads] = {"median_ms": med * 1e3, "haystacks_per_s": n_sample / med, "runs": reps}
    best_threads = max(by_threads, key=lambda t: by_threads[t]["haystacks_per_s"])
    best = by_threads[best_threads]["haystacks_per_s"]
    return {"value": best, "unit": "haystacks/s", "cores": best_threads, "kind": "port", "simd": simd,
            "sample": f"all {n_sample} len-{HAY_LEN} haystacks of the bench list, match_list_parallel's scoring loop without the ordering step, median per thread count "
                      f"(workers spawned per call like src/matcher/parallel.rs:43-64); C++ restatement of the reference "
                      f"({'AVX-512 lane vectors, 64 x u8' if simd == 'avx512' else 'portable lane loops'}, g++ -O3 -march=native), not the Rust binary (no cargo on the box: profiles/r02_box_probe.txt)",
            "by_threads": {str(k): v for k, v in by_threads.items()},
            "single_thread": {"haystacks_per_s": single, "sample": f"first {n1} haystacks, median of 7"},
            "physical_cores": phys, "hardware_threads": hw_threads,
            "linear_bound": {"haystacks_per_s": single * phys, "what": "single-thread rate x physical cores (perfect scaling, no spawn cost)"},
            "published_bound": {"haystacks_per_s": PUBLISHED_PER_THREAD * phys,
                                "what": "the reference's published 1.15e8 haystacks/s/thread (BENCHMARKS.md:123, Ryzen 9950X3D, same length and mix) x physical cores"}}


def _pmc_child(counters, kernel_substr, extra=()):
    """One child run of this script under `rocprofv3 --pmc <counters>` (counters only, no trace domains); returns {counter: (average per
    dispatch of the kernels whose name contains kernel_substr, dispatches)}."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    tmp = tempfile.mkdtemp(prefix="fzb_pmc_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", tmp, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--fast", "--steps", "3", "--warmup", "1", *extra]
        subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, timeout=240, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        acc = {}
        for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] in counters and kernel_substr in r["Kernel_Name"]:
                    t = acc.setdefault(r["Counter_Name"], [0.0, 0])
                    t[0] += float(r["Counter_Value"])
                    t[1] += 1
        return {k: (v[0] / v[1], v[1]) for k, v in acc.items() if v[1]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def live_counters():
    """Measured NOW, by child runs of this script under rocprofv3 --pmc (separate passes, counters only - as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes):
      * HBM bytes per k1_dfa launch = (FETCH_SIZE x 2 + WRITE_SIZE) KiB: both counters are KiB per dispatch, and on gfx950 FETCH_SIZE counts
        64 B per 128-B request for wide coalesced 16 B/lane reads, so it is doubled (the reduction of tools/pmc_traffic.py);
      * wave-instructions per scorer launch (SQ_INSTS_VALU, SQ_INSTS_SALU; SQ_ACTIVE_INST_ANY = quad-cycles of instruction issue, all kinds).
    Returns (dict, None) or (None, reason): a profiler problem must not cost the bench line."""
    import shutil

    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not on PATH"
    try:
        f = _pmc_child(["FETCH_SIZE"], "k1_dfa")
        w = _pmc_child(["WRITE_SIZE"], "k1_dfa")
        if "FETCH_SIZE" not in f or "WRITE_SIZE" not in w:
            return None, "no FETCH_SIZE / WRITE_SIZE rows for k1_dfa in the child runs' output"
        out = {"traffic": (f["FETCH_SIZE"][0] * 2 + w["WRITE_SIZE"][0]) * 1024,
               "traffic_source": f"MEASURED in this run: child runs `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) of `bench.py --fast --steps 3 --warmup 1`, "
                                 f"{f['FETCH_SIZE'][1]} / {w['WRITE_SIZE'][1]} k1_dfa dispatches averaged; FETCH_SIZE doubled (gfx950: 64 B counted per 128-B request)"}
        q = _pmc_child(["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_ANY"], "k2b_dp")
        if "SQ_INSTS_VALU" in q:
            out["scorer"] = {k: v[0] for k, v in q.items()}
            out["scorer"]["dispatches"] = q["SQ_INSTS_VALU"][1]
        # the All Match mix is integer-VALU-bound - its scorer's wave-instructions, from a child run over that list
        qa = _pmc_child(["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_ANY"], "k2b_dp", extra=("--mix", "all"))
        if "SQ_INSTS_VALU" in qa:
            out["scorer_all_match"] = {k: v[0] for k, v in qa.items()}
            out["scorer_all_match"]["dispatches"] = qa["SQ_INSTS_VALU"][1]
        return out, None
    except Exception as e:
        return None, f"{type(e).__name__}: {e}"


def stored_json(name):
    p = os.path.join(ROOT, "profiles", name)
    try:
        return json.load(open(p))
    except Exception:
        return None


def oracle_check(F, m_gpu, corpus, rows, n_check, max_typos, dev):
    """First n_check haystacks: GPU records (index order) vs the CPU oracle's, outside every timed region."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import zlib
    import oracle_lib as O

    out = torch.zeros(n_check * 8 + 64, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    m_gpu.match_list_device(corpus, out.data_ptr(), n_check, cnt.data_ptr(), count=n_check)
    torch.cuda.synchronize(dev)
    k = int(cnt[0].item())
    got = out[: k * 8].cpu().numpy().view(F.MATCH_DTYPE)
    data = np.concatenate([rows[:n_check].reshape(-1).cpu().numpy(), np.zeros(64, np.uint8)])
    ends = np.arange(1, n_check + 1, dtype=np.uint64) * np.uint64(HAY_LEN)
    want = O.Matcher(NEEDLE.decode(), lanes=(64, 64, 32), max_typos=max_typos, sort="IndexAsc").match_packed(data, ends)
    same = len(got) == len(want) and bool((got["index"] == want["index"]).all() and (got["score"] == want["score"]).all() and (got["exact"] == want["exact"]).all())
    return {"items": n_check, "records_gpu": int(len(got)), "records_oracle": int(len(want)), "records_equal": same,
            "crc32_gpu": zlib.crc32(np.ascontiguousarray(got).tobytes()), "crc32_oracle": zlib.crc32(np.ascontiguousarray(want).tobytes()),
            "what": "HIP path vs oracle/ (portable C++ restatement of the reference, emulating its AVX-512 backend) on the first items of the bench list, untimed"}


def step_roofline(sum_len, n, matches, ms, ends_read=True):
    """sum len + 4 N (u32 end offsets) + 8 M.  A corpus of uniform-length haystacks needs no offsets (the kernels compute the
    spans): the bytes then are sum len + 8 M, and the figure with the survey's formula is given beside it for comparison across rounds."""
    b = sum_len + (4 * n if ends_read else 0) + 8 * matches
    r = {"bytes": b, "GBps": b / (ms * 1e-3) / 1e9, "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "end_offsets_read": bool(ends_read)}
    if not ends_read:
        r["frac_with_survey_bytes"] = (sum_len + 4 * n + 8 * matches) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    return r


def sharded_block(F, needle, cfg, host_bytes, host_ends, unsharded_ms, unsharded_result):
    out = {"what": "fzb_match_list_parallel_sharded over the C2 list cut into k shards that share this ONE device (FZB_SHARD_OVERSUBSCRIBE): per-shard worker thread, "
                   "matcher clone and stream; runs copied device to device into one list, ordered once, one D2H.  Ordered records on the host, per call",
           "unsharded_match_list_ms": unsharded_ms}
    want = unsharded_result.tobytes()
    for mode in ("pull", "pull_workers", "copy"):
        # pull: the shards share the root device - enqueued by the calling thread, ONE kernel concatenates the runs (no host round trip
        #       before the final list); pull_workers: the same through the per-shard worker threads (what shards on other devices use).
        # copy: what shards on other devices take - count to the host, hipMemcpyPeerAsync to the run's place (forced on this one GPU)
        if mode == "copy":
            os.environ["FZB_SHARD_GATHER"] = "copy"
        if mode == "pull_workers":
            os.environ["FZB_SHARD_INLINE"] = "0"
        F.lib().fzb_debug_reload_knobs()
        try:
            for k in (1, 2, 8):
                sc = F.ShardedCorpus(packed=(host_bytes, host_ends), ndev=k, oversubscribe=True)
                m = F.Matcher(needle, cfg)
                got = m.match_list_parallel_sharded(sc, copy=False)
                same = got.tobytes() == want
                ts = []
                for _ in range(15):
                    t0 = time.perf_counter()
                    got = m.match_list_parallel_sharded(sc, copy=False)
                    ts.append(time.perf_counter() - t0)
                out[f"shards_{k}_{mode}"] = {"ms_median": _median(ts) * 1e3, "ms_min": min(ts) * 1e3, "vs_unsharded": _median(ts) * 1e3 / unsharded_ms, "equals_match_list": bool(same)}
                del m, sc, got
        finally:
            os.environ.pop("FZB_SHARD_GATHER", None)
            os.environ.pop("FZB_SHARD_INLINE", None)
            F.lib().fzb_debug_reload_knobs()
    return out


def other_configs(F, synth, dev, steps):
    """C3 / C4-shard / C5 of BASELINE.json on this GPU: device pipeline time per step (corpus resident), same definitions as the headline."""
    res = {}

    def run(name, needle, cfg, corpus, n, sum_len, ends_read=True, steps=steps):
        m = F.Matcher(needle, cfg)
        out = torch.zeros(n * 8 + 64, dtype=torch.uint8, device=dev)
        cnt = torch.zeros(4, dtype=torch.int32, device=dev)
        for _ in range(3):
            m.match_list_device(corpus, out.data_ptr(), n, cnt.data_ptr())
        torch.cuda.synchronize(dev)
        # (the best of three repetitions of the K-step loop: one host-side hiccup - a page fault, a scheduler tick - inside a loop of three 1.4 ms steps
        # once read 11 ms per step while the stage events of the same calls read 1.44; the headline's own loop is timed once, as the contract says)
        ms = float("inf")
        for _rep in range(3):
            t0 = time.perf_counter()
            for _ in range(steps):
                m.match_list_device(corpus, out.data_ptr(), n, cnt.data_ptr())
            torch.cuda.synchronize(dev)
            ms = min(ms, (time.perf_counter() - t0) / steps * 1e3)
        m.set_profiling(True)
        for _ in range(steps):
            m.match_list_device(corpus, out.data_ptr(), n, cnt.data_ptr())
        torch.cuda.synchronize(dev)
        try:
            st = m.last_stage_timings_ms()
        except F.FrizbeeError:  # (the long-needle pipeline records no stage events)
            st = {k: None for k in ("filter", "compaction_and_window", "scorers", "total")}
        matches = int(cnt[0].item())
        res[name] = {"haystacks": n, "ms_per_step": ms, "haystacks_per_s": n / (ms * 1e-3), "matches": matches, "roofline_step": step_roofline(sum_len, n, matches, ms, ends_read),
                     "stages_ms": {k: st[k] for k in ("filter", "compaction_and_window", "scorers", "total")}, **{k: v for k, v in m.last_counters().items()}}
        del m, out, cnt

    n = PER_GPU
    flat = torch.zeros(n * HAY_LEN + 256, dtype=torch.uint8, device=dev)
    flat[: n * HAY_LEN].view(n, HAY_LEN).copy_(synth.make_rows(NEEDLE, n, HAY_LEN, seed=12345, device=dev))
    ends = (torch.arange(1, n + 1, dtype=torch.int64, device=dev) * HAY_LEN).to(torch.int32)
    cp = F.Corpus.from_device(flat.data_ptr(), ends.data_ptr(), n, flat.numel(), keep=(flat, ends), max_len=HAY_LEN, uniform_len=HAY_LEN)
    run("C3 10M x 32 B, 'deadbe', max_typos=2", "deadbe", F.Config(max_typos=2, pf_lanes=64, sw_lanes=64), cp, n, n * HAY_LEN, ends_read=False)
    del cp
    # the other two mixes of the reference's generator on the same shape (benches/lib.rs:60-64) - All Match (every haystack
    # scored: integer-VALU-bound, the HBM fraction means nothing there; `scorer_issue` is added from the live counters) and No Match (the filter alone)
    for label, full, partial in (("All Match", 1.0, 0.0), ("No Match", 0.0, 0.0)):
        flat[: n * HAY_LEN].view(n, HAY_LEN).copy_(synth.make_rows(NEEDLE, n, HAY_LEN, seed=12345, device=dev, full=full, partial=partial))
        cp = F.Corpus.from_device(flat.data_ptr(), ends.data_ptr(), n, flat.numel(), keep=(flat, ends), max_len=HAY_LEN, uniform_len=HAY_LEN)
        key = f"C2 shape, mix {label} (benches/lib.rs:60-64): 10M x 32 B, 'deadbe', max_typos=0"
        run(key, "deadbe", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64), cp, n, n * HAY_LEN, ends_read=False, steps=5)
        res[key]["bound"] = "integer VALU issue (every haystack is scored)" if full == 1.0 else "HBM (the streaming filter alone)"
        del cp
    del flat, ends
    n4 = 12_500_000
    data, e4 = synth.ragged_corpus(b"deadbeef", n4, device=dev)
    cp = F.Corpus(packed=(data, e4))
    run("C4 one GPU's shard: 12.5M ragged 8..128 B, 'deadbeef', max_typos=0", "deadbeef", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64), cp, n4, int(e4[-1]))
    del cp
    # ... and BASELINE config 4 WHOLE on this one GPU: 100 M haystacks (the shard's list eight times; 6.8 GB of bytes, 7.6 GB in the padded
    # layout: 64-bit end offsets, the filter view above the 4 GiB line), resident in a fraction of the 288 GB.  Parity at this size:
    # tests/test_gpu_full_size.py::test_c4_whole_hundred_million_ragged_on_one_gpu
    rep = 8
    data8 = np.tile(data, rep)
    e8 = (e4[None, :] + (np.arange(rep, dtype=np.uint64) * e4[-1])[:, None]).reshape(-1)
    t0 = time.perf_counter()
    cp = F.Corpus(packed=(data8, e8))
    up_ms = (time.perf_counter() - t0) * 1e3
    name8 = "C4 whole: 100M ragged 8..128 B on ONE GPU (the 12.5M-item shard x 8), 'deadbeef', max_typos=0"
    run(name8, "deadbeef", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64), cp, n4 * rep, int(e8[-1]), steps=5)
    res[name8]["upload_ms"] = up_ms
    res[name8]["vs_shard_step"] = res[name8]["ms_per_step"] / res["C4 one GPU's shard: 12.5M ragged 8..128 B, 'deadbeef', max_typos=0"]["ms_per_step"]
    del cp, data, e4, data8, e8
    # the shape users see: the reference's real-data benchmark (BENCHMARKS.md:52-65, Chromium file paths: 1 406 941 items, median 67
    # characters, needle "linux", 8 % matching) from its synthetic generator's method - a ragged list, none of the len-32 fast paths
    npaths = 1_406_941
    dp, ep = synth.paths_corpus(b"linux", npaths, device=dev)
    cp = F.Corpus(packed=(dp, ep))
    name = "paths: 1,406,941 items, lengths ~ Normal(67, 17), 'linux', 8% full / 20% partial (the Chromium shape of BENCHMARKS.md:52-65, synthetic)"
    run(name, "linux", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64), cp, npaths, int(ep[-1]))
    mq = F.Matcher("linux", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64))
    mq.match_list(cp, copy=False)
    tq = []
    for _ in range(15):
        t0 = time.perf_counter()
        rq = mq.match_list(cp, copy=False)
        tq.append(time.perf_counter() - t0)
    res[name]["match_list_ms_median"] = _median(tq) * 1e3  # what a caller sees: pipeline + device sort + D2H of the ordered records
    res[name]["reference_published_ms"] = {"sequential": 22.36, "parallel_x8": 3.48, "where": "BENCHMARKS.md:62-65, Ryzen 9950X3D, the real Chromium list"}
    # the other columns of that table on the same list (All Scores = max_typos None; typo budgets 1 / 2 / 3): device pipeline per query
    cols = {}
    for label, mt, ref in (("all_scores", None, (84.64, 13.81)), ("typos_1", 1, (60.76, 9.50)), ("typos_2", 2, (99.15, 15.58)), ("typos_3", 3, (142.39, 20.29))):
        key = f"paths list, column {label}"
        run(key, "linux", F.Config(max_typos=mt, pf_lanes=64, sw_lanes=64), cp, npaths, int(ep[-1]), steps=5)
        cols[label] = {"ms_per_step": res[key]["ms_per_step"], "matches": res[key]["matches"], "reference_published_ms": {"sequential": ref[0], "parallel_x8": ref[1]}}
        del res[key]
    res[name]["other_columns"] = cols
    # repositories of ordinary size: the same shape at 100 k / 300 k items (their few thousand multi-chunk windows take four lanes each: dp_quad.h)
    small = {}
    for nsmall in (100_000, 300_000):
        dps, eps = synth.paths_corpus(b"linux", nsmall, device=dev)
        cps = F.Corpus(packed=(dps, eps))
        key = f"paths-shaped list of {nsmall // 1000} k items"
        run(key, "linux", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64), cps, nsmall, int(eps[-1]))
        small[f"{nsmall // 1000}k"] = {"ms_per_step": res[key]["ms_per_step"], "matches": res[key]["matches"], "multi_chunk_scored": res[key].get("multi_chunk_scored")}
        del res[key], cps, dps, eps
    res[name]["smaller_lists"] = small
    del cp, dp, ep, mq, rq
    # the reference's UTF-8 benchmark shape (BENCHMARKS.md "Arabic": 285 587 sentences, median 37 bytes, a needle of two Arabic letters), synthetic
    da, ea = synth.arabic_corpus()
    cp = F.Corpus(packed=(da, ea))
    namea = "arabic-shaped: 285,587 UTF-8 sentences (median 37 B, lognormal, up to 600 B), two-letter needle (the Arabic shape of BENCHMARKS.md, synthetic)"
    run(namea, "إن", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64), cp, int(len(ea)), int(ea[-1]), steps=5)
    cols = {}
    for label, mt, ref in (("all_scores", None, (15.13, 2.65)), ("typos_1", 1, (11.46, 1.97))):
        key = f"arabic list, column {label}"
        run(key, "إن", F.Config(max_typos=mt, pf_lanes=64, sw_lanes=64), cp, int(len(ea)), int(ea[-1]), steps=5)
        cols[label] = {"ms_per_step": res[key]["ms_per_step"], "matches": res[key]["matches"], "reference_published_ms": {"sequential": ref[0], "parallel_x8": ref[1]}}
        del res[key]
    res[namea]["other_columns"] = cols
    res[namea]["reference_published_ms"] = {"sequential": 2.60, "parallel_x8": 0.481, "where": "BENCHMARKS.md:59-95, Ryzen 9950X3D, the real list"}
    del cp, da, ea
    # a LONG needle (beyond the 64 bytes / 63 rows the by-value kernels take): the lane-exact prefilter as first stage,
    # the wave-per-haystack scorer as the only scorer, per-wave slabs in global memory.  Correct for every accepted length
    # (tests/test_gpu_long_needles.py); this row puts a number on "nothing here is tuned"
    nl = 1_000_000
    long_needle = bytes((b"abcdefghijklmnopqrstuvwxyz0123456789_-" * 3)[:80])
    gl = torch.Generator(device=dev)
    gl.manual_seed(99)
    lens_l = torch.randint(100, 201, (nl,), generator=gl, device=dev)
    rows_l = synth.make_rows(long_needle, nl, 200, lengths=lens_l, seed=4242, device=dev, chunk=1 << 18)
    mask_l = torch.arange(200, device=dev)[None, :] < lens_l[:, None]
    dl, el = rows_l[mask_l].cpu().numpy(), np.cumsum(lens_l.cpu().numpy().astype(np.uint64), dtype=np.uint64)
    del rows_l, mask_l
    cp = F.Corpus(packed=(dl, el))
    run("long needle: 80 bytes vs 1M haystacks of 100..200 B (5% contain it), max_typos=0", long_needle.decode(), F.Config(max_typos=0, pf_lanes=64, sw_lanes=32), cp, nl, int(el[-1]), steps=3)
    del cp, dl, el
    n5, reps = 2_000_000, 5
    d5, _ = synth.utf8_corpus(n5, HAY_LEN)
    d5 = np.tile(d5, reps)
    e5 = np.arange(1, n5 * reps + 1, dtype=np.uint64) * np.uint64(HAY_LEN)
    cp = F.Corpus(packed=(d5, e5))
    run("C5 10M x 32 B UTF-8 (2M distinct x 5), 4-scalar Arabic needle, max_typos=0", "إنما", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64), cp, n5 * reps, n5 * reps * HAY_LEN, ends_read=False)  # uploaded list of uniform length: detected by fzb_corpus_upload
    # ... and the same list with a typo budget: superset filter (scalar-level LCS) -> the reference's chunked multi-path scan at its exact lane width for every survivor
    run("C5 shape with max_typos=1: 10M x 32 B UTF-8, 4-scalar Arabic needle", "إنما", F.Config(max_typos=1, pf_lanes=64, sw_lanes=64), cp, n5 * reps, n5 * reps * HAY_LEN, ends_read=False, steps=5)
    del cp
    return res


def c4_sharded_block(F, synth, dist, dev, rank, world, total, steps, stream):
    """BASELINE.json configs[3] at N ranks: needle 'deadbeef' (8 chars) vs `total` mixed-length (8..128 byte) haystacks, max_typos=0, cut into
    BYTE-balanced contiguous index ranges (frizbee_amd.distributed.shard_ranges_by_bytes; the reference's shape: contiguous chunks with a
    global index offset, src/matcher/parallel.rs:35-87), one range per rank.  The list is defined globally - one seeded length vector every
    rank derives identically - and each rank builds only its own shard.  Timed like the headline: K steps of the rank's pipeline writing
    into the exchange buffer + the asynchronous double-buffered gather to rank 0, barrier + synchronize on both sides, max over ranks.
    Beside it the ORDERED query (`ShardExchange.ordered_query`: gather, one device-side concatenation / radix sort on the root, one D2H,
    grow-and-retry if a shard outgrows the exchange).  Checks (untimed): the ranges partition the list; every shard's first items equal the
    CPU oracle's records and every record's index lies in its shard's range; merged length == sum of the shard counts; merged order ==
    (score desc, index asc).  Every rank calls this (collectives inside); rank 0 returns the row."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from frizbee_amd.distributed import ShardExchange, shard_ranges_by_bytes

    g = torch.Generator()
    g.manual_seed(12346)
    lengths = torch.randint(8, 129, (total,), generator=g)  # the global list's lengths, identical on every rank
    ends_all = np.cumsum(lengths.numpy().astype(np.uint64), dtype=np.uint64)
    ranges = shard_ranges_by_bytes(ends_all, world)
    lo, hi = ranges[rank]
    n_loc = hi - lo
    parts, chunk = [], 1 << 22  # this rank's haystacks, content seeded per shard and chunk (bounded temporaries: 4 M rows of 128 B at a time)
    for ci, a in enumerate(range(lo, hi, chunk)):
        b = min(a + chunk, hi)
        L = lengths[a:b].to(dev)
        rows_c = synth.make_rows(b"deadbeef", b - a, 128, lengths=L, seed=12345 + 7919 * rank + ci, device=dev)
        parts.append(rows_c[torch.arange(128, device=dev)[None, :] < L[:, None]].cpu().numpy())
        del rows_c, L
    packed = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    del parts
    ends = ends_all[lo:hi] - (ends_all[lo - 1] if lo else np.uint64(0))
    t0 = time.perf_counter()
    corpus = F.Corpus(packed=(packed, ends))
    upload_ms = (time.perf_counter() - t0) * 1e3
    m = F.Matcher("deadbeef", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64))
    out = torch.zeros(n_loc * 8 + 64, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(4, dtype=torch.int32, device=dev)

    def fence():
        dist.barrier()
        torch.cuda.synchronize(dev)

    m.match_list_device(corpus, out.data_ptr(), n_loc, cnt.data_ptr(), stream=stream, index_offset=lo)
    torch.cuda.synchronize(dev)
    k_local = int(cnt[0].item())
    # this shard's head against the CPU oracle (checker), index order, global indices
    c = min(200_000, n_loc)
    got = out[: k_local * 8].cpu().numpy().view(F.MATCH_DTYPE)
    if c:
        want = O.Matcher("deadbeef", lanes=(64, 64, 32), max_typos=0, sort="IndexAsc").match_packed(np.concatenate([packed[: int(ends[c - 1])], np.zeros(64, np.uint8)]), ends[:c])
        want["index"] += np.uint32(lo)
        shard_ok = got[got["index"] < lo + c].tolist() == want.tolist() and bool(((got["index"] >= lo) & (got["index"] < hi)).all())
    else:
        shard_ok = len(got) == 0
    ex = ShardExchange(ShardExchange.plan(k_local, margin=1.05, device=None if dist.get_backend() != "nccl" else dev), dev)  # (one repeated query: see the headline loop)
    step_no = [0]

    def step():
        slot = step_no[0] & 1
        step_no[0] += 1
        ex.wait(slot)
        m.match_list_device(corpus, ex.records_ptr(slot), ex.cap, ex.count_ptr(slot), stream=stream, index_offset=lo)
        ex.post(slot)

    for _ in range(4):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    t_score = (time.perf_counter() - t0) / steps
    ex.collect(0)
    ex.collect(1)
    merged = None
    for _ in range(2):
        merged = ex.ordered_query(lambda rp, cap, cp: m.match_list_device(corpus, rp, cap, cp, stream=stream, index_offset=lo), m, stream=stream)
    k_e2e = max(3, min(10, steps))
    fence()
    t0 = time.perf_counter()
    for _ in range(k_e2e):
        merged = ex.ordered_query(lambda rp, cap, cp: m.match_list_device(corpus, rp, cap, cp, stream=stream, index_offset=lo), m, stream=stream)
    fence()
    t_e2e = (time.perf_counter() - t0) / k_e2e
    ctl = torch.device("cpu") if dist.get_backend() != "nccl" else dev
    oks = torch.tensor([int(shard_ok), k_local], dtype=torch.int64, device=ctl)
    allv = [torch.zeros_like(oks) for _ in range(world)]
    dist.all_gather(allv, oks)
    t = torch.tensor([t_score, t_e2e], dtype=torch.float64, device=ctl)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    row = None
    if rank == 0:
        total_matches = int(sum(int(v[1]) for v in allv))
        key = (0xFFFF - merged["score"].astype(np.int64)) * (1 << 32) + merged["index"].astype(np.int64)
        sum_len = int(ends_all[-1]) if total else 0
        bytes_ = sum_len + 4 * total + 8 * total_matches
        row = {"haystacks": total, "n_gpus": world, "bytes": sum_len,
               "shards": [{"range": [int(a), int(b)], "bytes": (int(ends_all[b - 1]) if b else 0) - (int(ends_all[a - 1]) if a else 0)} for a, b in ranges],
               "ms_per_step": float(t[0]) * 1e3, "haystacks_per_s": total / float(t[0]), "matches": total_matches,
               "roofline_step": {"bytes": bytes_, "GBps": bytes_ / float(t[0]) / 1e9, "frac_of_all_gpus_hbm": bytes_ / float(t[0]) / 1e9 / (HBM_PEAK_GBS * world),
                                 "what": "SURVEY 8(d) bytes of the WHOLE list (sum len + 4 N + 8 M) / step time / (8 TB/s x ranks)"},
               "e2e_ordered_list_on_rank0_ms": float(t[1]) * 1e3, "exchange_capacity_records": ex.cap, "exchange_grown": ex.grown, "upload_ms_rank0": upload_ms,
               "checks": {"ranges_partition_the_list": bool(ranges[0][0] == 0 and ranges[-1][1] == total and all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))),
                          "every_shard_head_equals_oracle_and_indices_in_range": all(int(v[0]) == 1 for v in allv), "oracle_items_per_shard": c,
                          "merged_len_equals_sum_of_shard_counts": int(len(merged)) == total_matches,
                          "merged_order_is_score_desc_then_index_asc": bool((np.diff(key) > 0).all()) if len(key) > 1 else True},
               "what": "BASELINE.json configs[3]: 'deadbeef' vs mixed-length 8..128 B haystacks, max_typos=0, byte-balanced contiguous shards, one per rank; "
                       "timed like the headline (pipeline + asynchronous double-buffered gather to rank 0, max over ranks)"}
    del corpus, m, out, cnt, ex
    return row


def cabi_rccl_block(F, dist, dev, rank, world, m, corpus, index_offset, merged_ref, k):
    """`match_list_parallel` with one process per GPU through the C ABI alone: fzb_shard_comm (its own RCCL communicator, created from an id that
    rank 0 draws and torch.distributed merely carries to the others) + fzb_match_list_parallel_rccl per step - run lengths all-gathered, exactly the
    records moved to rank 0 in one RCCL group, one device-side ordering, one copy to the host."""
    from frizbee_amd.distributed import RcclShardComm
    comm = RcclShardComm(rank, world)
    got = None
    for _ in range(2):  # (untimed: exchange buffers, the root's sort buffers and the pinned result are allocated on first use)
        got = comm.match_list_parallel(m, corpus, index_offset, copy=False)
    dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(k):
        got = comm.match_list_parallel(m, corpus, index_offset, copy=False)
    dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    sent, received = comm.last_exchange_bytes()
    row = {"ms_per_step": dt / k * 1e3, "steps": k, "merged_len": int(len(got)) if rank == 0 else None,
           "equals_host_merge": bool(got.tobytes() == merged_ref.tobytes()) if rank == 0 else None,
           "record_bytes_into_root_per_step": received if rank == 0 else None,
           "what": "fzb_match_list_parallel_rccl on every rank, every step (synchronous): pipeline into the communicator's buffer, ncclAllGather of the run lengths (the one "
                   "host synchronisation before the result), ONE RCCL group of sends / receives of exactly the records to rank 0, concatenation + stable radix sort on rank 0's "
                   "device, one D2H.  torch.distributed only carried the 128-byte communicator id"}
    del got
    comm.close()
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--per-gpu", type=int, default=PER_GPU)
    ap.add_argument("--max-typos", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the C3 / C4-shard / C5 block")
    ap.add_argument("--no-sharded", action="store_true", help="skip the sharded-on-one-device block (fzb_match_list_parallel_sharded vs fzb_match_list)")
    ap.add_argument("--no-check", action="store_true", help="skip the 1M-item oracle comparison")
    ap.add_argument("--no-two-in-flight", action="store_true", help="skip the two-streams throughput figure")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not spawn the two rocprofv3 --pmc child runs; report the stored traffic figure")
    ap.add_argument("--fast", action="store_true", help="= --no-cpu-baseline --no-configs --no-check --no-two-in-flight (profiling runs)")
    ap.add_argument("--mix", choices=("partial", "all", "none"), default="partial",
                    help="haystack classes of the list (benches/lib.rs:60-64): partial = 5%% full / 20%% partial / 75%% none (the headline), all = All Match, none = No Match (profiling child runs)")
    ap.add_argument("--c4-total", type=int, default=-1,
                    help="N > 1: haystacks of the BASELINE configs[3] row (100M ragged, byte-balanced shards); -1 = 100,000,000 when more than one rank runs, 0 = skip the row")
    args = ap.parse_args()
    # FZB_BENCH_BACKEND=gloo: the rank-count REHEARSAL of the N > 1 path on a box with fewer GPUs than ranks - ranks share the visible GPU(s),
    # the exchange moves CPU tensors (ShardExchange stages them), everything else is the code an RCCL run executes.  Labelled in the line.
    backend = os.environ.get("FZB_BENCH_BACKEND", "nccl")
    rehearsal = backend != "nccl"
    if args.fast:
        args.no_cpu_baseline = args.no_configs = args.no_check = args.no_two_in_flight = args.no_live_traffic = args.no_sharded = True

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if torch.cuda.device_count() < args.gpus and not rehearsal:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible on this node - refusing to run fewer ranks than asked for")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # invoked as plain `python bench.py --gpus N`: start the N ranks ourselves, exactly the way the driver would
        # (one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1), and hand over to them
        import socket

        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__), *sys.argv[1:]]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execvpe(cmd[0], cmd, os.environ)
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio on fd 1 (seen after the JSON line when stdout is a
    # file: it is flushed at exit), so the line gets its own duplicate of the original stdout and fd 1 itself is pointed at stderr.
    json_out = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    # FZB_BENCH_FORCE_DIST=1 runs the N > 1 code path (process group, exchange) with a single rank: a self-test of that
    # path on a 1-GPU box, not a benchmark configuration
    use_dist = world > 1 or os.environ.get("FZB_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if rehearsal:
            dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    ctl_dev = torch.device("cpu") if rehearsal else dev  # where the small control tensors of the collectives live

    import frizbee_amd as F
    import synth
    from frizbee_amd.distributed import ShardExchange, merge_shard_runs

    n = args.per_gpu
    # ---- synthetic shard, generated directly in HBM (padded-16 layout == back-to-back 32-byte rows) ----
    flat = torch.zeros(n * HAY_LEN + 256, dtype=torch.uint8, device=dev)
    rows = flat[: n * HAY_LEN].view(n, HAY_LEN)
    mix_full, mix_partial = {"partial": (0.05, 0.20), "all": (1.0, 0.0), "none": (0.0, 0.0)}[args.mix]
    rows.copy_(synth.make_rows(NEEDLE, n, HAY_LEN, seed=12345 + rank, device=dev, full=mix_full, partial=mix_partial))
    ends = (torch.arange(1, n + 1, dtype=torch.int64, device=dev) * HAY_LEN).to(torch.int32)
    # every haystack has exactly HAY_LEN bytes: declared to the library (an uploaded list is detected), whose hot kernels then compute the
    # spans instead of reading the end offsets
    corpus = F.Corpus.from_device(flat.data_ptr(), ends.data_ptr(), n, flat.numel(), ends_are_u64=False, keep=(flat, ends), max_len=HAY_LEN, uniform_len=HAY_LEN)
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
        # (the timed loop repeats ONE query without looking at the counts, so its exchange is sized from this first count; a caller whose queries
        # differ uses ShardExchange.ordered_query - below, `e2e_sorted_merge` - which re-sizes the exchange and repeats the query when a shard outgrows it)
        # (margin 1.05: the gather moves the WHOLE fixed-size buffer every step - 8 B x capacity per rank into rank 0 over its xGMI links - so slack in
        # the capacity is bandwidth of the timed loop at N = 8; the query is the same every step, its count does not move)
        ex = ShardExchange(ShardExchange.plan(int(cnt[0].item()), margin=1.05, device=ctl_dev), dev)
        # (still set-up: RCCL builds its channels and rings lazily on the first few collectives of a communicator - several
        # milliseconds each - so a handful of exchanges is run here, before the W warm-up steps of the contract)
        for s_ in range(8):
            ex.wait(s_ & 1)
            m.match_list_device(corpus, ex.records_ptr(s_ & 1), ex.cap, ex.count_ptr(s_ & 1), stream=stream, index_offset=index_offset)
            ex.post(s_ & 1)
        ex.collect(0)
        ex.collect(1)
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
        ex.post(slot, stream=stream)

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
    # stages on the launch stream (kept out of the timed region above so the events cost nothing there;
    # rocprofv3 --kernel-trace of this command covers both passes).
    m.set_profiling(True)
    for _ in range(args.steps):
        step()
    fence()
    st = m.last_stage_timings_ms()
    m.set_profiling(False)
    counters = m.last_counters()  # (of a full step: the untimed checks further down query sub-ranges)
    attribution = None
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # ---- what the exchange costs, so that a scaling loss at N > 1 is attributable (untimed here; same K steps, same fences) -------------
        # (1) the loop WITHOUT the exchange: the pipeline still writes into the exchange buffers, nothing is posted
        ex.wait(0)
        ex.wait(1)
        fence()
        t0 = time.perf_counter()
        for i_ in range(args.steps):
            m.match_list_device(corpus, ex.records_ptr(i_ & 1), ex.cap, ex.count_ptr(i_ & 1), stream=stream, index_offset=index_offset)
        fence()
        t_noex = time.perf_counter() - t0
        # (2) the exchange ALONE: K posts of the (already filled) buffers, double-buffered like the loop, no pipeline in between
        t0 = time.perf_counter()
        for i_ in range(args.steps):
            ex.post(i_ & 1)
        ex.wait(0)
        ex.wait(1)
        fence()
        t_ex = time.perf_counter() - t0
        t = torch.tensor([t_noex, t_ex], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_noex, t_ex = float(t[0].item()), float(t[1].item())
        attribution = {"exchange_bytes_per_rank_per_step": ex.bytes_per_rank(), "exchange_bytes_into_root_per_step": ex.bytes_per_rank() * (world - 1),
                       "transport": ex.transport + ((" (nothing travels with one rank: the root's run is in place)" if ex.transport == "p2p" else " (one rank: RCCL copies the root's own buffer, device to device)") if world == 1 else ""),
                       "ms_per_step_without_exchange": t_noex / args.steps * 1e3, "value_without_exchange": n * world / (t_noex / args.steps),
                       "exchange_alone_ms_per_step": t_ex / args.steps * 1e3,
                       "what": "same K steps, max over ranks: the loop with nothing posted; the posts alone (double-buffered, buffers already filled).  "
                               "ms_per_step - ms_per_step_without_exchange = what the asynchronous exchange costs the timed loop; exchange_alone >= ms_per_step_without_exchange "
                               "means the loop is bound by the transport, not by the kernels"}
    ms_per_step = elapsed / args.steps * 1e3
    gathered = None
    e2e_multi = None
    if ex is None:
        n_matches = int(cnt[0].item())
    else:
        last = (step_no[0] - 1) & 1
        runs = ex.collect(last)       # root: the per-shard runs of the last step; everyone: drains the exchange
        ex.collect(last ^ 1)
        n_matches = int(ex.send[last][:4].cpu().numpy().view(np.uint32)[0])
        if rank == 0:
            merged = merge_shard_runs(runs, F.SortStrategy.ScoreThenIndexAsc)  # host form of the combine (the device form is timed below)
            gathered = {"matches_all_shards": int(sum(len(r) for r in runs)), "merged_len": int(len(merged)), "exchange_capacity_records": ex.cap,
                        "capacity_over_records": ex.cap / max(1, max(len(r) for r in runs)), **(attribution or {})}
        # ---- the end-to-end mode (reported beside `value`, never instead of it): every step ends with the ORDERED list on rank 0's
        # host - what match_list_parallel returns (parallel.rs:66-87).  Each rank runs the pipeline unsorted with global indices, a
        # synchronous gather puts the runs into the root's HBM, and the root orders the whole list ONCE on the device (rank order is
        # ascending index order: concatenation + reverse / stable radix sort, fzb_merge_shard_runs) and makes one copy to the host.
        # Round 3 sorted per rank and k-merged on the root's host: 4.1 ms per step with one rank.
        k_e2e = max(3, min(10, args.steps))
        merged2 = None

        def run_into(records_ptr, capacity, count_ptr):
            m.match_list_device(corpus, records_ptr, capacity, count_ptr, stream=stream, index_offset=index_offset)

        # the first ordered query starts from an exchange that is deliberately too small: ordered_query must re-size it and repeat the query
        ex_small = ShardExchange(4096, dev)
        merged_grow = ex_small.ordered_query(run_into, m, stream=stream)
        grew = ex_small.grown
        del ex_small
        for _ in range(2):  # (untimed: the root's staging / sort buffers and the pinned result buffer are allocated on first use)
            merged2 = ex.ordered_query(run_into, m, stream=stream)
        fence()
        t0 = time.perf_counter()
        for _ in range(k_e2e):
            merged2 = ex.ordered_query(run_into, m, stream=stream)
        fence()
        e2e_multi = {"ms_per_step": (time.perf_counter() - t0) / k_e2e * 1e3, "steps": k_e2e,
                     "what": "ShardExchange.ordered_query, every step: per-rank pipeline (index order, global indices) + gather into rank 0's memory + ONE concatenation / stable radix sort "
                             "(on the device with RCCL; on the host in the gloo rehearsal) + one D2H + an 8-byte broadcast that tells every rank the exchange held every run (grow-and-retry otherwise)",
                     "merged_len": int(len(merged2)) if rank == 0 else None,
                     "equals_host_merge": bool(merged2.tobytes() == merged.tobytes()) if rank == 0 else None,
                     "grow_and_retry": {"exchange_started_at_records": 4096, "times_grown": grew,
                                        "result_equals": bool(merged_grow.tobytes() == merged.tobytes()) if rank == 0 else None}}
        # every rank: the first items of ITS shard against the CPU oracle (checker, untimed), records in index order with global indices
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        n_chk = min(n, 1_000_000)
        m.match_list_device(corpus, out.data_ptr(), n, cnt.data_ptr(), stream=stream, index_offset=index_offset, count=n_chk)
        torch.cuda.synchronize(dev)
        got_s = out[: int(cnt[0].item()) * 8].cpu().numpy().view(F.MATCH_DTYPE)
        want_s = O.Matcher(NEEDLE.decode(), lanes=(64, 64, 32), max_typos=args.max_typos, sort="IndexAsc").match_packed(
            np.concatenate([rows[:n_chk].reshape(-1).cpu().numpy(), np.zeros(64, np.uint8)]), np.arange(1, n_chk + 1, dtype=np.uint64) * np.uint64(HAY_LEN))
        want_s["index"] += np.uint32(index_offset)
        okt = torch.tensor([int(got_s.tolist() == want_s.tolist())], dtype=torch.int64, device=ctl_dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        e2e_multi["every_shard_head_equals_oracle"] = {"items_per_shard": n_chk, "equal_on_every_rank": bool(int(okt.item()) == 1)}
        if n * world <= 4_000_000:
            # small totals (the rehearsal): the WHOLE merged list against the oracle's ordered list - rank 0 regenerates every shard's rows (seeded)
            if rank == 0:
                allrows = torch.cat([synth.make_rows(NEEDLE, n, HAY_LEN, seed=12345 + r_, device=dev) for r_ in range(world)]).reshape(-1).cpu().numpy()
                want_all = O.Matcher(NEEDLE.decode(), lanes=(64, 64, 32), max_typos=args.max_typos, sort="ScoreThenIndexAsc").match_packed(
                    np.concatenate([allrows, np.zeros(64, np.uint8)]), np.arange(1, n * world + 1, dtype=np.uint64) * np.uint64(HAY_LEN))
                e2e_multi["merged_equals_oracle_list"] = bool(merged2.tolist() == want_all.tolist())
        c4_total = args.c4_total if args.c4_total >= 0 else (100_000_000 if world > 1 else 0)
        c4_row = c4_sharded_block(F, synth, dist, dev, rank, world, c4_total, max(3, min(10, args.steps)), stream) if c4_total else None
    if rank == 0:
        total = n * world
        # algorithmic bytes of one filter launch (this rank's shard): payload once + 1 decision bit per haystack (no end offsets: uniform length)
        filt_bytes = n * HAY_LEN + n / 8
        achieved = filt_bytes / (st["filter"] * 1e-3) / 1e9
        tr = stored_json("latest_traffic.json") or {}
        sq = stored_json("latest_sq.json") or {}
        scorer = {"kernel": sq.get("scorer_kernel"), "bound": "valu-issue", "avg_ms": st["scorers"]}
        if sq.get("scorer_valu_wave_instructions_per_launch"):
            clk = sq.get("shader_clock_GHz", 2.25)
            simds = 4 * 256
            scorer.update({"valu_wave_instructions_per_launch": sq["scorer_valu_wave_instructions_per_launch"],
                           "all_wave_instructions_per_launch": sq.get("scorer_all_wave_instructions_per_launch"),
                           "issue_frac": sq["scorer_valu_wave_instructions_per_launch"] * 4 / (simds * st["scorers"] * 1e-3 * clk * 1e9),
                           "assumes": f"4 cycles per wave64 VALU instruction, {simds} SIMDs, {clk} GHz (in-kernel s_memtime / s_memrealtime, tools/exp_dp_timing.py)",
                           "counters_source": sq.get("source")})
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
                                   + ("mix 5% full / 20% partial / 75% none, seed 12345 (BASELINE.json configs[1])" if args.mix == "partial" else f"mix {args.mix.upper()} MATCH (not the headline mix)"),
                       "haystacks_per_gpu": n, "haystack_len": HAY_LEN, "max_typos": args.max_typos,
                       "emulated_reference_backend": "AVX-512 (prefilter 64 lanes, Smith-Waterman 64 x u8)",
                       "sharding": f"contiguous index ranges over {world} GPU(s); per step an asynchronous, double-buffered RCCL gather of the match records to rank 0" if world > 1 else "single GPU",
                       "ranks_seen": (dist.get_world_size() if use_dist else 1),
                       "backend": ((f"{backend} - REHEARSAL of the N > 1 path: {world} rank(s) share {torch.cuda.device_count()} GPU(s), the exchange moves CPU tensors; not a scaling measurement"
                                    if rehearsal else "nccl (RCCL)") if use_dist else "none (single process)"),
                       "matches_per_shard": n_matches, "filter_survivors": counters["filter_survivors"], "exchange": gathered},
            "roofline": {"bound": "hbm", "kernel": "k1_dfa", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": tr.get("k1_filter_hbm_bytes_per_launch"),
                         "traffic_source": "STORED, not measured in this run: profiles/latest_traffic.json (" + str(tr.get("source")) + ")",
                         "traffic_stored": tr.get("k1_filter_hbm_bytes_per_launch"),
                         "bytes_per_launch": filt_bytes, "avg_kernel_ms": st["filter"], "launches_averaged": st["calls"]},
            "roofline_step": step_roofline(n * HAY_LEN, n, n_matches, ms_per_step, ends_read=False),
            "stages": {"filter_ms": st["filter"], "compaction_ms": st["compaction_and_window"], "scorer_ms": st["scorers"], "device_pipeline_ms": st["total"], "scorer": scorer,
                       "what": "HIP events on the launch stream around each stage, averaged over the profiled steps (the K steps after the timed K)"},
        }
        if e2e_multi:
            res["e2e_sorted_merge"] = e2e_multi
            if c4_row is not None:
                res.setdefault("configs", {})[f"C4 sharded over {world} GPU(s): {c4_row['haystacks']:,} ragged 8..128 B, 'deadbeef', max_typos=0 (BASELINE.json configs[3])"] = c4_row
        if world == 1:
            # what `Matcher::match_list` hands a caller: ordered records in host memory (pipeline + device sort + D2H), per call
            torch.cuda.set_stream(torch.cuda.default_stream(dev))
            m2 = F.Matcher(NEEDLE.decode(), cfg)
            m2.match_list(corpus, copy=False)
            ts = []
            for _ in range(15):
                t0 = time.perf_counter()
                r = m2.match_list(corpus, copy=False)
                ts.append(time.perf_counter() - t0)
            res["e2e"] = {"match_list_ms_median": _median(ts) * 1e3, "match_list_ms_min": min(ts) * 1e3, "records": int(len(r)), "haystacks_per_s": n / _median(ts),
                          "what": "fzb_match_list = Matcher::match_list (src/matcher/mod.rs:212-222): pipeline + device reverse/radix sort + D2H of the ordered records, corpus resident"}
            # ... and what the same call costs COLD, from a list in pageable host memory: upload (raw bytes + offsets at link speed, device
            # layout built by kernels) + a fresh matcher's first query + D2H.  Never `value`: the timed region above starts with the list in HBM.
            host_bytes = rows.cpu().numpy().reshape(-1)
            host_ends = np.arange(1, n + 1, dtype=np.uint64) * np.uint64(HAY_LEN)
            cold = []
            for _ in range(4):
                t0 = time.perf_counter()
                cpc = F.Corpus(packed=(host_bytes, host_ends))
                t1 = time.perf_counter()
                mc = F.Matcher(NEEDLE.decode(), cfg)
                rc_ = mc.match_list(cpc, copy=False)
                t2 = time.perf_counter()
                cold.append((t2 - t0, t1 - t0, t2 - t1))
                del cpc, mc, rc_
            cold_sorted = sorted(cold[1:])
            res["e2e_cold"] = {"ms_first": cold[0][0] * 1e3, "ms_median": cold_sorted[len(cold_sorted) // 2][0] * 1e3, "upload_ms_median": _median([c_[1] for c_ in cold[1:]]) * 1e3,
                               "first_query_ms_median": _median([c_[2] for c_ in cold[1:]]) * 1e3, "host_bytes": int(host_bytes.nbytes + host_ends.nbytes),
                               "upload_GBps": (host_bytes.nbytes + host_ends.nbytes) / _median([c_[1] for c_ in cold[1:]]) / 1e9,
                               "haystacks_per_s": n / cold_sorted[len(cold_sorted) // 2][0],
                               "what": "fzb_corpus_upload (pageable host memory -> HBM, padded-16 layout built on the device) + fzb_matcher_create + first fzb_match_list (workspace allocation, pipeline, device sort, D2H); "
                                       "ms_first also pays the first-touch of the process"}
            if not args.no_sharded:
                # The multi-device form behind the C ABI (fzb_match_list_parallel_sharded), as far as ONE GPU can show it: the same list
                # cut into 8 shards that share this device (persistent worker thread + matcher clone + stream per shard, runs gathered
                # device to device, ordered once on the root), against fzb_match_list on the unsharded list.  Same records, same order.
                try:
                    res["sharded"] = sharded_block(F, NEEDLE.decode(), cfg, host_bytes, host_ends, res["e2e"]["match_list_ms_median"], r)
                except Exception as ex_:  # reported, never fatal for the headline
                    res["sharded"] = {"error": repr(ex_)}
            del host_bytes, host_ends
            if not args.no_two_in_flight:
                # two independent queries in flight on two streams (two matchers = two workspaces): the HBM-bound filter of one overlaps the
                # issue-bound scorer of the other.  Reported beside `value`, never as it: bench steps are sequential single queries.
                m3 = F.Matcher(NEEDLE.decode(), cfg)
                out3 = torch.zeros(n * 8 + 64, dtype=torch.uint8, device=dev)
                cnt3 = torch.zeros(4, dtype=torch.int32, device=dev)
                sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
                pairs = max(10, args.steps // 2)

                def pair_loop(k):
                    for _ in range(k):
                        m.match_list_device(corpus, out.data_ptr(), n, cnt.data_ptr(), stream=sa.cuda_stream)
                        m3.match_list_device(corpus, out3.data_ptr(), n, cnt3.data_ptr(), stream=sb.cuda_stream)

                pair_loop(3)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                pair_loop(pairs)
                torch.cuda.synchronize(dev)
                tp = (time.perf_counter() - t0) / (2 * pairs)
                res["two_queries_in_flight"] = {"ms_per_query": tp * 1e3, "haystacks_per_s": n / tp, "queries": 2 * pairs,
                                                "what": "the same query issued alternately on two HIP streams through two matchers; throughput of independent queries, not the latency of one"}
                del m3, out3, cnt3
            if not args.no_check:
                res["check"] = oracle_check(F, m2, corpus, rows, min(n, 1_000_000), args.max_typos, dev)
            del m2
            if not args.no_configs:
                res.setdefault("configs", {}).update(other_configs(F, synth, dev, 10))
            if not args.no_live_traffic:
                torch.cuda.synchronize(dev)
                lc_, why = live_counters()
                if lc_ is not None:
                    res["roofline"]["traffic"] = lc_["traffic"]
                    res["roofline"]["traffic_source"] = lc_["traffic_source"]
                    if "scorer" in lc_ and st["scorers"] > 0:
                        clk = (stored_json("latest_sq.json") or {}).get("shader_clock_GHz", 2.25)
                        sc_ = res["stages"]["scorer"]
                        sc_.update({"valu_wave_instructions_per_launch": lc_["scorer"]["SQ_INSTS_VALU"], "salu_wave_instructions_per_launch": lc_["scorer"].get("SQ_INSTS_SALU"),
                                    "issue_quad_cycles_all_instructions_per_launch": lc_["scorer"].get("SQ_ACTIVE_INST_ANY"),
                                    "issue_frac": lc_["scorer"]["SQ_INSTS_VALU"] * 4 / (4 * 256 * st["scorers"] * 1e-3 * clk * 1e9),
                                    "issue_frac_all_instructions": (lc_["scorer"]["SQ_ACTIVE_INST_ANY"] * 4 / (4 * 256 * st["scorers"] * 1e-3 * clk * 1e9)) if "SQ_ACTIVE_INST_ANY" in lc_["scorer"] else None,
                                    "counters_source": f"MEASURED in this run: child run `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY` of `bench.py --fast --steps 3 --warmup 1`, {lc_['scorer']['dispatches']} scorer dispatches averaged"})
                    if "scorer_all_match" in lc_ and "configs" in res:
                        clk = (stored_json("latest_sq.json") or {}).get("shader_clock_GHz", 2.25)
                        for key, row in res["configs"].items():
                            if "mix All Match" in key and row["stages_ms"].get("scorers"):
                                t_sc = row["stages_ms"]["scorers"] * 1e-3
                                a_ = lc_["scorer_all_match"]
                                row["scorer_issue"] = {"valu_wave_instructions_per_launch": a_["SQ_INSTS_VALU"], "salu_wave_instructions_per_launch": a_.get("SQ_INSTS_SALU"),
                                                       "valu_issue_frac": a_["SQ_INSTS_VALU"] * 4 / (4 * 256 * t_sc * clk * 1e9),
                                                       "issue_frac_all_instructions": (a_["SQ_ACTIVE_INST_ANY"] * 4 / (4 * 256 * t_sc * clk * 1e9)) if "SQ_ACTIVE_INST_ANY" in a_ else None,
                                                       "assumes": f"4 cycles per wave64 VALU instruction, 1024 SIMDs, {clk} GHz",
                                                       "counters_source": f"MEASURED in this run: child run `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY` of `bench.py --fast --mix all --steps 3 --warmup 1`, {a_['dispatches']} scorer dispatches averaged"}
                else:
                    res["roofline"]["traffic_source"] += "; live collection failed: " + why
            if not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(rows, min(n, 10_000_000), args.max_typos)
                res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
                bound = max(res["cpu_baseline"]["value"], res["cpu_baseline"]["linear_bound"]["haystacks_per_s"], res["cpu_baseline"]["published_bound"]["haystacks_per_s"])
                res["gpu_over_cpu_bound"] = {"ratio": res["value"] / bound,
                                             "against": "the largest of: measured port, single-thread x physical cores, published per-thread x physical cores"}
    if use_dist and not rehearsal and os.environ.get("FZB_BENCH_CABI_RCCL", "1") != "0":
        # ---- the ordered query through the C ABI ALONE (fzb_match_list_parallel_rccl, csrc/host_rccl.hip: RCCL opened by the library itself, what a
        # Rust host with one process per GPU binds) - untimed for `value`, reported beside e2e_sorted_merge.  It has never run on more than one GPU, so it
        # runs LAST, after everything else of the line exists, under a watchdog: if it has not come back in time every rank leaves and rank 0 prints the
        # line without it, saying so.
        import threading

        def bail():
            if rank == 0:
                res["e2e_cabi_rccl"] = {"error": "fzb_match_list_parallel_rccl did not return within 120 s (watchdog): everything else in this line was measured before it started"}
                print(json.dumps(res), file=json_out, flush=True)
            os._exit(0)

        dog = threading.Timer(120.0, bail)
        dog.daemon = True
        dog.start()
        try:
            cabi = cabi_rccl_block(F, dist, dev, rank, world, m, corpus, index_offset, merged if rank == 0 else None, max(3, min(10, args.steps)))
        except Exception as ex_:  # reported, never fatal for the line
            cabi = {"error": repr(ex_)}
        dog.cancel()
        if rank == 0:
            res["e2e_cabi_rccl"] = cabi
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res), file=json_out, flush=True)


if __name__ == "__main__":
    main()
