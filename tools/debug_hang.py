import sys, os, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
UNI = ["é", "ن", "다", "😀", "न", " ", "_", "/", "a", "b", "c", "A", "B", "É", "0", "إ", "م"]
def cases(pf, max_typos):
    rng = np.random.default_rng(31337 + (max_typos or 9) + pf)
    for it in range(10):
        asz = int(rng.integers(3, len(UNI)))
        needle = "".join(UNI[int(x)] for x in rng.integers(0, asz, int(rng.integers(1, 7))))
        if needle.isascii(): needle += "é"
        hs = []
        for _ in range(250):
            n = int(rng.integers(0, 60)) if rng.random() > 0.2 else int(rng.choice([0, 1, 7, 8, 15, 16, 31, 32, 33]))
            chars = [UNI[int(x)] for x in rng.integers(0, asz, n)]
            if n > len(needle) and rng.random() < 0.5:
                pos = np.sort(rng.choice(n, len(needle), replace=False))
                for p, c in zip(pos, needle):
                    if rng.random() < 0.9: chars[p] = c
            hs.append("".join(chars))
        yield it, needle, ["Smart", "Ignore", "Respect"][it % 3], hs
if len(sys.argv) > 1 and sys.argv[1] == "one":
    import frizbee_amd as F
    d = json.loads(sys.stdin.read())
    m = F.Matcher(d["needle"], F.Config(max_typos=d["k"], casing=F.CaseMatching[d["casing"]], pf_lanes=d["pf"], sw_lanes=d["pf"]))
    r = m.match_list(d["hs"])
    print(len(r))
    sys.exit(0)
pf, k = 64, 2
for it, needle, casing, hs in cases(pf, k):
    def run(sub):
        try:
            p = subprocess.run([sys.executable, __file__, "one"], input=json.dumps(dict(needle=needle, k=k, casing=casing, pf=pf, hs=sub)), capture_output=True, text=True, timeout=20)
            return p.returncode == 0, p.stdout.strip() + p.stderr.strip()[-300:]
        except subprocess.TimeoutExpired:
            return False, "TIMEOUT"
    ok, out = run(hs)
    print(it, repr(needle), casing, ok, out, flush=True)
    if not ok:
        lo = hs
        # bisect to a single haystack
        while len(lo) > 1:
            a, b = lo[: len(lo) // 2], lo[len(lo) // 2 :]
            oka, _ = run(a)
            lo = b if oka else a
        print("CULPRIT", repr(needle), casing, repr(lo[0]), lo[0].encode(), flush=True)
        break
