"""Round-5 A/B timings (device pipeline per query, ms): the window kernel's PRE form (occurrence masks laid out ahead by the whole workgroup)
against round 4's one-pass form (FZB_WINDOW_NO_PRE=1) on the reference's two real-data shapes with typo budgets, and the long-needle
scorers (one thread per window + streaming DFA first stage, against the wave-per-haystack kernel alone: FZB_LONG_GENERIC_ONLY=1).
One JSON object per line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, synth, frizbee_amd as F
dev = torch.device("cuda", 0)


def timed(label, needle, cfg, cp, n, env, steps=10):
    for k, v in env.items(): os.environ[k] = v
    F.lib().fzb_debug_reload_knobs()
    try:
        m = F.Matcher(needle, cfg)
        out = torch.zeros(n * 8 + 64, dtype=torch.uint8, device=dev); cnt = torch.zeros(4, dtype=torch.int32, device=dev)
        for _ in range(3): m.match_list_device(cp, out.data_ptr(), n, cnt.data_ptr())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): m.match_list_device(cp, out.data_ptr(), n, cnt.data_ptr())
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / steps * 1e3
        m.set_profiling(True)
        for _ in range(steps): m.match_list_device(cp, out.data_ptr(), n, cnt.data_ptr())
        torch.cuda.synchronize(); st = m.last_stage_timings_ms()
        print(json.dumps(dict(exp=label, env=env, ms_per_step=ms, stages_ms={k: st[k] for k in ("filter", "compaction_and_window", "scorers", "total")}, matches=int(cnt[0].item()), **m.last_counters())), flush=True)
    finally:
        for k in env: os.environ.pop(k, None)
        F.lib().fzb_debug_reload_knobs()


which = sys.argv[1:] or ["arabic", "paths", "long"]
if "arabic1" in which:  # the 1-typo column alone (measurement runs under FZB_WINDOW_DBG)
    da, ea = synth.arabic_corpus()
    cp = F.Corpus(packed=(da, ea))
    timed("arabic-shaped 285k, max_typos=1", "إن", F.Config(max_typos=1, pf_lanes=64, sw_lanes=64), cp, int(len(ea)), {}, steps=5)
    del cp
if "arabic" in which:
    da, ea = synth.arabic_corpus()
    cp = F.Corpus(packed=(da, ea))
    for mt in (1, 2):
        for env in ({}, {"FZB_WINDOW_WHOLE_TILES": "1"}, {"FZB_WINDOW_NO_PRE": "1"}):
            timed(f"arabic-shaped 285k, max_typos={mt}", "إن" if mt == 1 else "إنما", F.Config(max_typos=mt, pf_lanes=64, sw_lanes=64), cp, int(len(ea)), env, steps=5)
    del cp
if "paths" in which:
    dp, ep = synth.paths_corpus(b"linux", 1_406_941, device=dev)
    cp = F.Corpus(packed=(dp, ep))
    for mt in (1, 2, 3):
        for env in ({}, {"FZB_WINDOW_WHOLE_TILES": "1"}, {"FZB_WINDOW_NO_PRE": "1"}):
            timed(f"paths-shaped 1.4M 'linux', max_typos={mt}", "linux", F.Config(max_typos=mt, pf_lanes=64, sw_lanes=64), cp, 1_406_941, env, steps=5)
    del cp
if "long" in which:
    nl = 1_000_000
    long_needle = bytes((b"abcdefghijklmnopqrstuvwxyz0123456789_-" * 3)[:80])
    gl = torch.Generator(device=dev); gl.manual_seed(99)
    lens_l = torch.randint(100, 201, (nl,), generator=gl, device=dev)
    rows_l = synth.make_rows(long_needle, nl, 200, lengths=lens_l, seed=4242, device=dev, chunk=1 << 18)
    mask_l = torch.arange(200, device=dev)[None, :] < lens_l[:, None]
    dl, el = rows_l[mask_l].cpu().numpy(), np.cumsum(lens_l.cpu().numpy().astype(np.uint64), dtype=np.uint64)
    del rows_l, mask_l
    cp = F.Corpus(packed=(dl, el))
    for mt in (0, 1, None):
        for env in ({}, {"FZB_LONG_GENERIC_ONLY": "1"}):
            timed(f"80-byte needle vs 1M haystacks of 100..200 B, max_typos={mt}", long_needle.decode(), F.Config(max_typos=mt, pf_lanes=64, sw_lanes=32), cp, nl, env, steps=3)
    del cp
