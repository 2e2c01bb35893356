"""Deterministic synthetic haystack lists for tests and bench.py.

Method after the reference's benchmark generator (/root/reference/benches/match_list/generate.rs:48-129,
mix constants /root/reference/benches/lib.rs:63): every item is one of
  Full    - all needle chars, in order, at random positions; filler = random [A-Za-z0-9]
  Partial - a random strict subset (0..n-1 chars) of the needle, in order; filler excludes the needle's letters (both cases)
  None    - filler excluding the needle's letters
Default mix = the reference's "Partial Match" set: 5 % Full, 20 % Partial, 75 % None; seed 12345
(/root/reference/benches/match_list/mod.rs:17).  Written with torch ops so the 10M-item bench set can be
generated directly in HBM; tests run the same code on CPU.  (PRNG differs from Rust's StdRng: the data is
statistically, not byte-wise, the reference's.)
"""
import numpy as np
import torch

ALNUM = b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789"


def _alphabets(needle: bytes, device):
    needle_l = set(bytes([c]).lower()[0] for c in needle) | set(bytes([c]).upper()[0] for c in needle)
    filt = bytes(c for c in ALNUM if c not in needle_l)
    return (torch.tensor(list(ALNUM), dtype=torch.uint8, device=device), torch.tensor(list(filt), dtype=torch.uint8, device=device))


def make_rows(needle: bytes, n: int, width: int, lengths=None, seed=12345, device="cpu", full=0.05, partial=0.20, chunk=1 << 20):
    """Returns a (n, width) uint8 tensor; row i holds a haystack of lengths[i] bytes (default: width), zero padded."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    nn = len(needle)
    alnum, filt = _alphabets(needle, device)
    needle_t = torch.tensor(list(needle), dtype=torch.uint8, device=device)
    out = torch.empty((n, width), dtype=torch.uint8, device=device)
    ar = torch.arange(width, device=device)
    for lo in range(0, n, chunk):
        hi = min(lo + chunk, n)
        m = hi - lo
        L = torch.full((m,), width, dtype=torch.int64, device=device) if lengths is None else lengths[lo:hi].to(torch.int64)
        u = torch.rand(m, generator=g, device=device)
        is_partial = u < partial
        is_full = (~is_partial) & (u < partial + full)
        # number of needle chars embedded: Full -> n (capped by length), Partial -> uniform in [0, min(L, n)), None -> 0
        kmax = torch.minimum(L, torch.full_like(L, nn))
        kpart = (torch.rand(m, generator=g, device=device) * kmax).floor().to(torch.int64).clamp_(max=nn - 1 if nn else 0)
        k = torch.where(is_full, kmax, torch.where(is_partial, kpart, torch.zeros_like(L)))
        # which needle indices (sorted random subset of size k; Full takes all)
        rn = torch.rand((m, nn), generator=g, device=device)
        rank_n = rn.argsort(dim=1).argsort(dim=1)
        sel = rank_n < k[:, None]
        sel = torch.where(is_full[:, None] & (torch.arange(nn, device=device)[None, :] < k[:, None]), torch.ones_like(sel), torch.where(is_full[:, None], torch.zeros_like(sel), sel))
        order = (~sel).to(torch.int8).argsort(dim=1, stable=True)  # selected indices first, in needle order
        chars_sorted = needle_t[order]
        # where in the haystack (sorted random positions < L)
        rp = torch.rand((m, width), generator=g, device=device)
        rp = torch.where(ar[None, :] < L[:, None], rp, torch.full_like(rp, 2.0))
        rank_p = rp.argsort(dim=1).argsort(dim=1)
        pos = rank_p < k[:, None]
        slot = (pos.cumsum(dim=1) - 1).clamp_(min=0)
        slot = slot.clamp_(max=max(nn - 1, 0))
        emb = torch.gather(chars_sorted, 1, slot) if nn else torch.zeros((m, width), dtype=torch.uint8, device=device)
        # filler
        fa = alnum[(torch.rand((m, width), generator=g, device=device) * len(alnum)).long().clamp_(max=len(alnum) - 1)]
        if len(filt):
            ff = filt[(torch.rand((m, width), generator=g, device=device) * len(filt)).long().clamp_(max=len(filt) - 1)]
        else:
            ff = fa
        fill = torch.where(is_full[:, None], fa, ff)
        row = torch.where(pos, emb, fill)
        row = torch.where(ar[None, :] < L[:, None], row, torch.zeros_like(row))
        out[lo:hi] = row
    return out


def fixed_corpus(needle: bytes, n: int, length: int = 32, **kw):
    """n haystacks of exactly `length` bytes.  Returns (rows (n,length) uint8 tensor, ends uint64 numpy) in the UPLOAD format
    (packed back to back).  For length % 16 == 0 this is also the device padded-16 layout."""
    rows = make_rows(needle, n, length, **kw)
    ends = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(length))
    return rows, ends


def ragged_corpus(needle: bytes, n: int, lo: int = 8, hi: int = 128, seed=12345, device="cpu", **kw):
    """n haystacks with lengths uniform in [lo, hi].  Returns (packed uint8 numpy, ends uint64 numpy) in the upload format."""
    g = torch.Generator(device=device)
    g.manual_seed(seed + 1)
    lengths = torch.randint(lo, hi + 1, (n,), generator=g, device=device)
    rows = make_rows(needle, n, hi, lengths=lengths, seed=seed, device=device, **kw)
    mask = torch.arange(hi, device=device)[None, :] < lengths[:, None]
    packed = rows[mask].cpu().numpy()
    ends = np.cumsum(lengths.cpu().numpy().astype(np.uint64), dtype=np.uint64)
    return packed, ends


def paths_corpus(needle: bytes = b"linux", n: int = 1_406_941, median: float = 67.0, std: float = 17.0, full=0.08, partial=0.20, seed=12345, device="cpu", width=160):
    """The shape of the reference's real-data benchmark (BENCHMARKS.md:52-65: all file paths of the Chromium repository - 1 406 941 items,
    median 67 characters, needle "linux", 8 % matching) produced with its synthetic generator's method (benches/match_list/generate.rs:
    lengths ~ round(|Normal(median, std)|) >= 1, Full / Partial / None classes, alphanumeric filler).  `partial` and `std` are not published
    for that list ("unknown"): 20 % and 17 are this repo's choice.  Lengths are capped at `width` (5.5 sigma).  Returns the upload format."""
    g = torch.Generator(device=device)
    g.manual_seed(seed + 2)
    lengths = torch.randn(n, generator=g, device=device).mul_(std).add_(median).round_().abs_().clamp_(1, width).to(torch.int64)
    rows = make_rows(needle, n, width, lengths=lengths, seed=seed, device=device, full=full, partial=partial)
    mask = torch.arange(width, device=device)[None, :] < lengths[:, None]
    packed = rows[mask].cpu().numpy()
    ends = np.cumsum(lengths.cpu().numpy().astype(np.uint64), dtype=np.uint64)
    return packed, ends


def utf8_corpus(n: int, length: int = 32, seed=12345, needle="إنما", full=0.05, partial=0.20):
    """n valid-UTF-8 haystacks of exactly `length` bytes built from 2-byte Arabic scalars + ASCII space/punct (BASELINE config 5).
    Host-side (numpy) generator; returns (packed uint8 numpy, ends uint64 numpy)."""
    rng = np.random.default_rng(seed)
    needle_chars = list(needle)
    arabic = [chr(c) for c in range(0x0621, 0x064B) if chr(c) not in needle_chars]
    ascii_fill = list(" .,-_/:")
    out = bytearray()
    u = rng.random(n)
    for i in range(n):
        is_partial = u[i] < partial
        is_full = (not is_partial) and u[i] < partial + full
        units = []
        budget = length
        if is_full:
            emb = list(needle_chars)
        elif is_partial:
            k = int(rng.integers(0, len(needle_chars)))
            idx = sorted(rng.choice(len(needle_chars), size=k, replace=False).tolist())
            emb = [needle_chars[j] for j in idx]
        else:
            emb = []
        budget -= sum(len(c.encode()) for c in emb)
        fill = []
        while budget > 0:
            if budget >= 2 and rng.random() < 0.7:
                c = arabic[int(rng.integers(len(arabic)))]
                if is_full and rng.random() < 0.2:
                    c = needle_chars[int(rng.integers(len(needle_chars)))]
                fill.append(c)
                budget -= 2
            else:
                fill.append(ascii_fill[int(rng.integers(len(ascii_fill)))])
                budget -= 1
        # join_randomly: keep both orders
        total = len(emb) + len(fill)
        take = np.zeros(total, dtype=bool)
        if emb:
            take[np.sort(rng.choice(total, size=len(emb), replace=False))] = True
        ei = fi = 0
        for t in take:
            if t:
                units.append(emb[ei]); ei += 1
            else:
                units.append(fill[fi]); fi += 1
        b = "".join(units).encode()
        assert len(b) == length, (len(b), length)
        out += b
    return np.frombuffer(bytes(out), dtype=np.uint8).copy(), (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(length))


def arabic_corpus(n: int = 285_587, needle="إن", median=37.0, mean=43.18, full=0.07934, partial=0.59514, seed=12345, max_bytes=600):
    """The shape of the reference's UTF-8 benchmark (BENCHMARKS.md: "Arabic" - 285 587 sentences, needle "إن", 7.9 % matching, 59.5 % partial,
    median 37 bytes / 21 chars, mean 43.18 bytes, 2 bytes per character) produced synthetically: byte lengths ~ LogNormal with that median and
    mean (sigma^2 = 2 ln(mean / median); its standard deviation comes out at 26 bytes where the real set has 33), Arabic letters and spaces,
    Full / Partial / None classes as in the reference's generator.  Host-side (numpy); returns (packed uint8 numpy, ends uint64 numpy)."""
    rng = np.random.default_rng(seed + 7)
    sigma = float(np.sqrt(2.0 * np.log(mean / median)))
    lens = np.clip(np.round(rng.lognormal(np.log(median), sigma, n)), 2, max_bytes).astype(np.int64)
    needle_chars = list(needle)
    arabic = [chr(c) for c in range(0x0621, 0x064B) if chr(c) not in needle_chars]
    u = rng.random(n)
    out = bytearray()
    ends = np.empty(n, dtype=np.uint64)
    for i in range(n):
        is_partial = u[i] < partial
        is_full = (not is_partial) and u[i] < partial + full
        emb = list(needle_chars) if is_full else ([needle_chars[int(rng.integers(len(needle_chars)))]] if is_partial and rng.random() < 0.5 else [])
        budget = int(lens[i]) - 2 * len(emb)
        fill = []
        while budget > 0:
            if budget >= 2 and rng.random() < 0.85:
                fill.append(arabic[int(rng.integers(len(arabic)))]); budget -= 2
            else:
                fill.append(" "); budget -= 1
        total = len(emb) + len(fill)
        take = np.zeros(total, dtype=bool)
        if emb:
            take[np.sort(rng.choice(total, size=len(emb), replace=False))] = True
        ei = fi = 0
        units = []
        for t in take:
            if t:
                units.append(emb[ei]); ei += 1
            else:
                units.append(fill[fi]); fi += 1
        out += "".join(units).encode()
        ends[i] = len(out)
    return np.frombuffer(bytes(out), dtype=np.uint8).copy(), ends
