#!/usr/bin/env python3
"""HBM traffic of the streaming filter from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes).  Both counters are in KiB per dispatch; on gfx950 FETCH_SIZE counts 64 B per 128-B
request for wide coalesced 16 B/lane reads, so it is doubled (same guide).  Usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, sys


def avg(d, counter, kernel):
    tot = n = 0
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]:
                tot += float(r["Counter_Value"]); n += 1
    return (tot / n if n else None), n


fetch, nf = avg(sys.argv[1], "FETCH_SIZE", "k1_dfa")
write, nw = avg(sys.argv[2], "WRITE_SIZE", "k1_dfa")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 5 --warmup 2 --fast, MI355X",
       "kernel": "k1_dfa", "FETCH_SIZE_KB_avg_per_dispatch_raw": fetch, "WRITE_SIZE_KB_avg_per_dispatch_raw": write, "dispatches": [nf, nw],
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced 16 B/lane reads -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as is",
       "k1_filter_hbm_bytes_per_launch": (fetch * 2 + write) * 1024 if fetch is not None and write is not None else None,
       "algorithmic_bytes_per_launch": 10_000_000 * 32 + 10_000_000 / 8,
       "note": "uniform-length list: no end offsets read (round 1: + 4 B per haystack)"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
