#!/bin/bash
# Per-kernel times of the ragged configurations (C4 shard, paths-shaped list): rocprofv3 --kernel-trace --stats of tools/bench_configs.py.
# usage: tools/prof_ragged.sh OUTDIR [configs...]   (run on the GPU box; OUTDIR under gpurun_out/)
out=$1; shift
cfgs=${@:-C4 PATHS}
mkdir -p "$out"
export TMPDIR=/tmp
root=$(pwd)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/prof" -o p -- python "$root/tools/bench_configs.py" $cfgs ) > "$out/prof.log" 2>&1
f=$(find "$out/prof" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$out/kernel_stats.csv" && python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:18]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.2f} total_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
grep -h '"config"' "$out/prof.log" | cut -c1-300
