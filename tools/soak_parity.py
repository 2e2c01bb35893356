"""Parity soak on a GPU box: random needles, scorings, typo budgets, lane widths and list shapes - fresh seeds every run - HIP path against the
oracle until the time budget is spent (tests/test_gpu_fuzz_isa.py's generators and checker, which raise with the first differing record and its
haystack).  Usage: python tools/soak_parity.py [seconds=600] [seed=time | reuse]   Prints one line per 50 lists and a summary; exit code 1 on a difference."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_fuzz_isa as T

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "reuse" else int(time.time())
REUSE = "reuse" in sys.argv[2:]  # one long-lived matcher per lane width, re-targeted by set_config / set_pattern: the workspace and table-upload paths of a session
if REUSE:
    import frizbee_amd as F, oracle_lib as O
    _live = {}
    def _check_reuse(needle, data, ends, lanes, tag, **cfg):
        pf, sw8, sw16 = T.LANE_TRIPLES[lanes]
        om = O.Matcher(needle, lanes=(pf, sw8, sw16), sort="IndexAsc", **cfg)
        fc = F.Config(max_typos=cfg.get("max_typos", 0), scoring=F.Scoring(*cfg.get("scoring", T.DEFAULT)), pf_lanes=pf,
                      unicode=F.UnicodeMatching[cfg.get("unicode", "Smart")], casing=F.CaseMatching[cfg.get("casing", "Smart")])
        fm = _live.get(lanes)
        if fm is None:
            fm = _live[lanes] = F.Matcher(needle, fc)
        else:
            fm.set_config(fc)
            fm.set_pattern(needle)
        want = om.match_packed(data, ends)
        got = fm.match_list_into(F.Corpus(packed=(data, ends)))
        if got.tolist() != want.tolist():
            bad = next((i for i in range(min(len(got), len(want))) if got[i].tolist() != want[i].tolist()), min(len(got), len(want)))
            raise AssertionError((tag, "records", len(got), len(want), "first difference at", bad, got[bad:bad + 1].tolist(), want[bad:bad + 1].tolist()))
        return len(ends)
    T.check = _check_reuse
rng = np.random.default_rng(seed)
print(f"soak seed {seed}, {budget:.0f} s", flush=True)
ALPHA = b"abcdefABCDEF_-/ .019xyzXYZ"
POOLS = [
    ("short", np.arange(0, 33)),
    ("chunk", np.arange(0, 65)),
    ("ragged", np.concatenate([np.arange(1, 130), [191, 192, 193, 255, 256, 257, 300]])),
    ("wide", np.array([60, 64, 65, 100, 128, 129, 200, 256, 257, 400, 512, 513, 700, 1000, 1023, 1024, 1025, 1100])),
    ("paths", np.clip(np.round(np.abs(np.random.default_rng(1).normal(67, 17, 400))), 1, 160).astype(np.int64)),
]
t0 = time.time(); lists = 0; items = 0
while time.time() - t0 < budget:
    lanes = int(rng.choice([64, 64, 64, 32, 16]))
    if rng.random() < 0.3:  # a unicode list: needle of 1..6 scalars out of a small multi-width alphabet
        chars = "éÉaAЖж中文字إنما😀ñ_ b"
        un = int(rng.integers(1, 7))
        needle_u = "".join(chars[int(x)] for x in rng.integers(0, len(chars), un))
        typos = [0, 0, 1, 2, None][int(rng.integers(0, 5))]
        if typos not in (0, None) and un <= typos: typos = 0
        sc = T.SCORINGS[int(rng.integers(0, len(T.SCORINGS)))][0] if rng.random() < 0.4 else T.DEFAULT
        n = int(rng.choice([300, 4000, 9000, 40000]))
        max_chars = int(rng.choice([8, 14, 30, 60, 120, 300]))
        if max_chars >= 120: n = min(n, 4000)
        data, ends = T.make_unicode_list(rng, needle_u, n, max_chars)
        tag = (seed, lists, lanes, sc, needle_u, typos, "unicode", n, max_chars)
        try:
            T.check(needle_u, data, ends, lanes, tag, max_typos=typos, scoring=sc)
        except AssertionError as e:
            print("DIFFERENCE", e, flush=True); sys.exit(1)
        except Exception as e:
            if "too long" in str(e) or "overflow" in str(e): continue
            print("ERROR", tag, repr(e), flush=True); sys.exit(2)
        lists += 1; items += n
        if lists % 50 == 0: print(f"{lists} lists, {items} haystacks, {time.time() - t0:.0f} s", flush=True)
        continue
    nn = int(rng.choice([1, 2, 3, 5, 6, 8, 12, 16, 20, 32, 40, 63, 64, 65, 80, 130], p=[.06, .08, .08, .1, .12, .12, .08, .06, .05, .05, .03, .03, .03, .03, .05, .03]))
    needle = bytes(rng.choice(np.frombuffer(ALPHA, np.uint8), nn))
    if rng.random() < 0.5: needle = needle.lower()
    if rng.random() < 0.3: sc = T.SCORINGS[int(rng.integers(0, len(T.SCORINGS)))][0]
    elif rng.random() < 0.5: sc = T.DEFAULT
    else:
        sc = [int(rng.integers(0, 40)), int(rng.integers(0, 20)), int(rng.integers(0, 20)), int(rng.integers(0, 6)), int(rng.integers(0, 30)), int(rng.integers(0, 12)),
              int(rng.integers(0, 12)), int(rng.integers(0, 20)), int(rng.integers(0, 12))]
    typos = [0, 0, 0, 1, 2, 3, None][int(rng.integers(0, 7))]
    if typos not in (0, None) and nn <= typos: typos = 0
    name, pool = POOLS[int(rng.integers(0, len(POOLS)))]
    if nn >= 40 and name == "short": name, pool = POOLS[2]
    n = int(rng.choice([300, 3000, 26000, 26000, 70000, 150000]))
    if typos is None or nn >= 40: n = min(n, 26000)
    if name == "wide": n = min(n, 8000)
    casing = ["Smart", "Smart", "Respect", "Ignore"][int(rng.integers(0, 4))]
    data, ends = T.make_list(rng, needle, n, pool)
    tag = (seed, lists, lanes, sc, needle, typos, name, n, casing)
    try:
        T.check(needle, data, ends, lanes, tag, max_typos=typos, scoring=sc, casing=casing)
    except AssertionError as e:
        print("DIFFERENCE", e, flush=True); sys.exit(1)
    except Exception as e:  # a configuration the API refuses (needle too long for the scoring ...) must be refused by both sides: check() builds the oracle first
        msg = str(e)
        if "too long" in msg or "overflow" in msg: continue
        print("ERROR", tag, repr(e), flush=True); sys.exit(2)
    lists += 1; items += n
    if lists % 50 == 0: print(f"{lists} lists, {items} haystacks, {time.time() - t0:.0f} s", flush=True)
print(f"soak ok: {lists} lists, {items} haystacks in {time.time() - t0:.0f} s, seed {seed}", flush=True)
