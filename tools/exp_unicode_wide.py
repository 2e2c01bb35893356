"""Unicode windows wider than one chunk: the wave-per-haystack kernel (k2c_generic, LDS sized by the needle's rows since round 4) at several
grid sizes against the thread-per-haystack multi-chunk scorer (k2u_dp_unicode_multi, FZB_UNICODE_MULTI=1), on the Arabic-shaped list
(285 587 sentences, 45 k windows beyond 64 bytes under All Scores) and on the same list eight times over (360 k such windows)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, synth, frizbee_amd as F
dev = torch.device("cuda", 0)
data, ends = synth.arabic_corpus()
lists = {"arabic x1": (data, ends)}
rep = 8
lists["arabic x8"] = (np.tile(data, rep), (ends[None, :] + (np.arange(rep, dtype=np.uint64) * ends[-1])[:, None]).reshape(-1))
for name, (d, e) in lists.items():
    n = len(e)
    cp = F.Corpus(packed=(d, e))
    out = torch.zeros(n * 8 + 64, dtype=torch.uint8, device=dev); cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    for env in ({"FZB_GENERIC_WGS": "2"}, {"FZB_GENERIC_WGS": "4"}, {}, {"FZB_GENERIC_WGS": "8"}, {"FZB_GENERIC_WGS": "12"}, {"FZB_UNICODE_MULTI": "1"}):
        for k in ("FZB_GENERIC_WGS", "FZB_UNICODE_MULTI"):
            os.environ.pop(k, None)
        os.environ.update(env)
        F.lib().fzb_debug_reload_knobs()
        for label, mt in (("All Scores", None), ("typos0", 0)):
            m = F.Matcher("إن", F.Config(max_typos=mt, pf_lanes=64, sw_lanes=64))
            for _ in range(2): m.match_list_device(cp, out.data_ptr(), n, cnt.data_ptr())
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): m.match_list_device(cp, out.data_ptr(), n, cnt.data_ptr())
            torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 5
            print(json.dumps(dict(list=name, env=env, config=label, ms_per_step=round(wall * 1e3, 4), **m.last_counters())), flush=True)
            del m
    del cp
