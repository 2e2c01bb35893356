"""Device-pipeline timing of the other BASELINE.json configurations (not the headline bench line):
C3 = len-32 / max_typos=2, C4-shard = ragged 8..128 (one GPU's 12.5M-item shard of the 100M list), C5 = UTF-8 len-32.
Prints one JSON object per config with the per-kernel-stage event timings measured by the library."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, synth, frizbee_amd as F
dev = torch.device("cuda", 0)
def run(name, needle, cfg, corpus, n, steps=10):
    m = F.Matcher(needle, cfg)
    out = torch.zeros(n * 8 + 64, dtype=torch.uint8, device=dev); cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    for _ in range(2): m.match_list_device(corpus, out.data_ptr(), n, cnt.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): m.match_list_device(corpus, out.data_ptr(), n, cnt.data_ptr())
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / steps
    m.set_profiling(True)
    for _ in range(steps): m.match_list_device(corpus, out.data_ptr(), n, cnt.data_ptr())
    torch.cuda.synchronize(); tm = m.last_timings_ms(); c = m.last_counters()
    print(json.dumps(dict(config=name, haystacks=n, ms_per_step=wall * 1e3, haystacks_per_s=n / wall, filter_ms=tm["filter"], pipeline_ms=tm["total"], matches=int(cnt[0].item()), **c)), flush=True)
which = sys.argv[1:] or ["C2", "C3", "C4", "C5"]
n = 10_000_000
if "C2" in which or "C3" in which:
    flat = torch.zeros(n * 32 + 256, dtype=torch.uint8, device=dev)
    flat[: n * 32].view(n, 32).copy_(synth.make_rows(b"deadbe", n, 32, seed=12345, device=dev))
    ends = (torch.arange(1, n + 1, dtype=torch.int64, device=dev) * 32).to(torch.int32)
    cp = F.Corpus.from_device(flat.data_ptr(), ends.data_ptr(), n, flat.numel(), keep=(flat, ends), max_len=32, uniform_len=32)
    if "C2" in which: run("C2 len32 typos0", "deadbe", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64), cp, n)
    if "C3" in which: run("C3 len32 typos2", "deadbe", F.Config(max_typos=2, pf_lanes=64, sw_lanes=64), cp, n)
    if "C2none" in which: run("C2 len32 typosNone(all scored)", "deadbe", F.Config(max_typos=None, pf_lanes=64, sw_lanes=64), cp, n, steps=3)
    del cp, flat, ends
for mixname, full, partial in (("MIXALL", 1.0, 0.0), ("MIXNONE", 0.0, 0.0)):
    # the other two mixes of the reference's benchmark (benches/lib.rs:60-64): All Match (VALU-bound ceiling), No Match (pure filter)
    if mixname in which:
        flat = torch.zeros(n * 32 + 256, dtype=torch.uint8, device=dev)
        flat[: n * 32].view(n, 32).copy_(synth.make_rows(b"deadbe", n, 32, seed=12345, device=dev, full=full, partial=partial))
        ends = (torch.arange(1, n + 1, dtype=torch.int64, device=dev) * 32).to(torch.int32)
        cp = F.Corpus.from_device(flat.data_ptr(), ends.data_ptr(), n, flat.numel(), keep=(flat, ends), max_len=32, uniform_len=32)
        run(f"C2 list with mix full={full} partial={partial}", "deadbe", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64), cp, n, steps=5)
        del cp, flat, ends
if "MULTI" in which:
    # three fuzzy patterns composed on the device (two narrowing intersections + one negation) on the C2 list
    flat = torch.zeros(n * 32 + 256, dtype=torch.uint8, device=dev)
    flat[: n * 32].view(n, 32).copy_(synth.make_rows(b"deadbe", n, 32, seed=12345, device=dev))
    ends = (torch.arange(1, n + 1, dtype=torch.int64, device=dev) * 32).to(torch.int32)
    cp = F.Corpus.from_device(flat.data_ptr(), ends.data_ptr(), n, flat.numel(), keep=(flat, ends), max_len=32, uniform_len=32)
    mm = F.MultiMatcher([F.Pattern("dead"), F.Pattern("be"), F.Pattern("x", negated=True)], F.Config(max_typos=0, pf_lanes=64))
    out = torch.zeros(n * 8 + 64, dtype=torch.uint8, device=dev); cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    for _ in range(2): mm.match_list_device(cp, out.data_ptr(), n, cnt.data_ptr())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): mm.match_list_device(cp, out.data_ptr(), n, cnt.data_ptr())
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10
    print(json.dumps(dict(config="C2 list, patterns dead + be + !x", haystacks=n, ms_per_step=wall * 1e3, haystacks_per_s=n / wall, matches=int(cnt[0].item()))), flush=True)
    del cp, flat, ends
if "LITERAL" in which:
    # the literal modes on the C2 list (device pipeline: accept pass -> compaction -> scoring pass)
    flat = torch.zeros(n * 32 + 256, dtype=torch.uint8, device=dev)
    flat[: n * 32].view(n, 32).copy_(synth.make_rows(b"deadbe", n, 32, seed=12345, device=dev))
    ends = (torch.arange(1, n + 1, dtype=torch.int64, device=dev) * 32).to(torch.int32)
    cp = F.Corpus.from_device(flat.data_ptr(), ends.data_ptr(), n, flat.numel(), keep=(flat, ends), max_len=32, uniform_len=32)
    out = torch.zeros(n * 8 + 64, dtype=torch.uint8, device=dev); cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    for needle, matching in (("de", "Substring"), ("deadbe", "Substring"), ("d", "Prefix"), ("e", "Suffix")):
        m = F.Matcher(needle, F.Config(matching=F.Matching[matching]))
        for _ in range(2): m.match_list_device(cp, out.data_ptr(), n, cnt.data_ptr())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): m.match_list_device(cp, out.data_ptr(), n, cnt.data_ptr())
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10
        print(json.dumps(dict(config=f"C2 list, literal {matching} {needle!r}", haystacks=n, ms_per_step=wall * 1e3, haystacks_per_s=n / wall, matches=int(cnt[0].item()))), flush=True)
    del cp, flat, ends
if "E2E" in which:
    # end to end from host memory: pack + upload once (fzb_corpus_upload), then the ordered query the caller sees
    # (fzb_match_list: pipeline + device sort + D2H of the records)
    rows, ends = synth.fixed_corpus(b"deadbe", n, 32, device=dev)
    data = rows.cpu().numpy().reshape(-1)
    t0 = time.perf_counter(); cp = F.Corpus(packed=(data, ends)); t_up = time.perf_counter() - t0
    m = F.Matcher("deadbe", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64))
    m.match_list(cp)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); r = m.match_list(cp, copy=False); ts.append(time.perf_counter() - t0)
    # re-query after a needle change (Matcher::set_pattern): tables re-uploaded, workspace kept
    tq = []
    for needle in ("dead", "deadb", "deadbe") * 4:
        t0 = time.perf_counter(); m.set_pattern(needle); r2 = m.match_list(cp, copy=False); tq.append(time.perf_counter() - t0)
    m.set_pattern("deadbe")
    print(json.dumps(dict(config="C2 end to end", haystacks=n, set_pattern_plus_match_list_ms_median=sorted(tq)[len(tq) // 2] * 1e3, corpus_upload_ms=t_up * 1e3, upload_GBps=data.nbytes / t_up / 1e9,
                          match_list_ms_best=min(ts) * 1e3, match_list_ms_median=sorted(ts)[len(ts) // 2] * 1e3, matches=int(len(r)),
                          haystacks_per_s_query=n / min(ts))), flush=True)
    del cp
if "C4" in which:
    n4 = int(os.environ.get("FZB_N4", 12_500_000 if "C4small" not in which else 2_000_000))
    data, ends = synth.ragged_corpus(b"deadbeef", n4, device=dev)
    cp = F.Corpus(packed=(data, ends))
    run("C4 shard ragged 8..128 typos0", "deadbeef", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64), cp, n4)
    del cp
if "PATHS" in which:
    npaths = 1_406_941
    data, ends = synth.paths_corpus(b"linux", npaths, device=dev)
    cp = F.Corpus(packed=(data, ends))
    run("paths-shaped 1.4M ~Normal(67,17) 'linux' typos0", "linux", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64), cp, npaths)
    del cp
if "C5" in which:
    n5 = 2_000_000
    data, ends = synth.utf8_corpus(n5, 32)
    reps = 5
    data = np.tile(data, reps); ends = np.arange(1, n5 * reps + 1, dtype=np.uint64) * np.uint64(32)
    cp = F.Corpus(packed=(data, ends))
    run("C5 utf8 len32 typos0 (2M distinct x5)", "إنما", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64), cp, n5 * reps)
    if "C5T" in which:
        run("C5 utf8 len32 typos1 (2M distinct x5)", "إنما", F.Config(max_typos=1, pf_lanes=64, sw_lanes=64), cp, n5 * reps)
        run("C5 utf8 len32 typos2 (2M distinct x5)", "إنما", F.Config(max_typos=2, pf_lanes=64, sw_lanes=64), cp, n5 * reps)
if "INDICES" in which:
    # matched byte positions for the top K of a sorted match_list over the resident C2 list
    # (host call -> selection upload -> item pipeline -> traced scorer + on-device traceback -> copies -> host ordering)
    flat = torch.zeros(n * 32 + 256, dtype=torch.uint8, device=dev)
    flat[: n * 32].view(n, 32).copy_(synth.make_rows(b"deadbe", n, 32, seed=12345, device=dev))
    ends = (torch.arange(1, n + 1, dtype=torch.int64, device=dev) * 32).to(torch.int32)
    cp = F.Corpus.from_device(flat.data_ptr(), ends.data_ptr(), n, flat.numel(), keep=(flat, ends), max_len=32, uniform_len=32)
    for typos in (0, 1):
        m = F.Matcher("deadbe", F.Config(max_typos=typos, pf_lanes=64, sw_lanes=64))
        top = m.match_list(cp)
        for k in (100, 1000, 10000, 100000):
            sel = top["index"][:k].astype(np.uint32)
            m.match_list_indices(cp, sel)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); r = m.match_list_indices(cp, sel); ts.append(time.perf_counter() - t0)
            # the C call alone (without building Python objects)
            import ctypes as C
            out, nn, pos = C.c_void_p(), C.c_size_t(), C.c_void_p()
            tc = []
            for _ in range(5):
                t0 = time.perf_counter()
                F.lib().fzb_match_list_indices(m.h, cp.h, sel.ctypes.data, len(sel), C.byref(out), C.byref(nn), C.byref(pos))
                tc.append(time.perf_counter() - t0)
                F.lib().fzb_match_indices_free(out, pos)
            print(json.dumps(dict(config=f"C2 list, positions for the top {k} of match_list, max_typos={typos}", selection=k, records=len(r),
                                  c_call_ms=sorted(tc)[2] * 1e3, python_call_ms=sorted(ts)[2] * 1e3, haystacks_per_s=k / sorted(tc)[2])), flush=True)
    del cp, flat, ends
if "PATHSSMALL" in which:
    # repositories of ordinary size (the 1.4 M-item row is Chromium): the same shape at 100 k / 300 k items
    for npaths in (100_000, 300_000):
        data, ends = synth.paths_corpus(b"linux", npaths, device=dev)
        cp = F.Corpus(packed=(data, ends))
        run(f"paths-shaped {npaths // 1000}k 'linux' typos0", "linux", F.Config(max_typos=0, pf_lanes=64, sw_lanes=64), cp, npaths)
        run(f"paths-shaped {npaths // 1000}k 'linux' 1 typo", "linux", F.Config(max_typos=1, pf_lanes=64, sw_lanes=64), cp, npaths, steps=5)
        del cp
if "PATHSVAR" in which:
    # the other columns of the reference's Chromium table (BENCHMARKS.md:59-65): All Scores (max_typos None), 1 / 2 / 3 typos
    npaths = 1_406_941
    data, ends = synth.paths_corpus(b"linux", npaths, device=dev)
    cp = F.Corpus(packed=(data, ends))
    for label, mt in (("All Scores (max_typos None)", None), ("1 typo", 1), ("2 typos", 2), ("3 typos", 3)):
        run(f"paths-shaped 1.4M 'linux' {label}", "linux", F.Config(max_typos=mt, pf_lanes=64, sw_lanes=64), cp, npaths, steps=5)
    del cp
if "ARABIC" in which or "ARABICDEF" in which or "ARABICALL" in which:
    # the shape of the reference's UTF-8 benchmark (BENCHMARKS.md "Arabic": 285 587 sentences, needle of two Arabic letters)
    t0 = time.perf_counter(); data, ends = synth.arabic_corpus(); tgen = time.perf_counter() - t0
    cp = F.Corpus(packed=(data, ends))
    lens = np.diff(np.concatenate([[0], ends.astype(np.int64)]))
    print(json.dumps(dict(config="arabic-shaped list", items=int(len(ends)), total_bytes=int(ends[-1]), median_len=float(np.median(lens)), mean_len=float(lens.mean()), std_len=float(lens.std()), gen_s=tgen)), flush=True)
    for label, mt in (("typos0", 0), ("All Scores (max_typos None)", None), ("1 typo", 1)):
        if "ARABICDEF" in which and mt != 0: continue  # (profiling the default column alone)
        if "ARABICALL" in which and mt is not None: continue
        run(f"arabic-shaped 285k {label}", "إن", F.Config(max_typos=mt, pf_lanes=64, sw_lanes=64), cp, int(len(ends)), steps=5)
    del cp
if "LONG" in which:
    # bench.py's long-needle row: 80 bytes vs 1 M haystacks of 100..200 bytes, 5 % contain it (streaming DFA, then one thread per window)
    nl = 1_000_000
    long_needle = bytes((b"abcdefghijklmnopqrstuvwxyz0123456789_-" * 3)[:80])
    gl = torch.Generator(device=dev); gl.manual_seed(99)
    lens_l = torch.randint(100, 201, (nl,), generator=gl, device=dev)
    rows_l = synth.make_rows(long_needle, nl, 200, lengths=lens_l, seed=4242, device=dev, chunk=1 << 18)
    mask_l = torch.arange(200, device=dev)[None, :] < lens_l[:, None]
    dl, el = rows_l[mask_l].cpu().numpy(), np.cumsum(lens_l.cpu().numpy().astype(np.uint64), dtype=np.uint64)
    del rows_l, mask_l
    cp = F.Corpus(packed=(dl, el))
    run("long needle 80 B vs 1M x 100..200 B typos0", long_needle.decode(), F.Config(max_typos=0, pf_lanes=64, sw_lanes=32), cp, nl, steps=5)
    run("long needle 80 B, 64-lane score chunks asked for", long_needle.decode(), F.Config(max_typos=0, pf_lanes=64, sw_lanes=64), cp, nl, steps=5)
    del cp
