#!/bin/bash
# HBM-side traffic of every kernel of the C4-shard pipeline: FETCH_SIZE and WRITE_SIZE in separate counter-only rocprofv3 --pmc passes
# (MI355X_MICROARCH.md: they do not fit one pass; on gfx950 FETCH_SIZE counts 64 B per 128-byte request of a wide coalesced read - doubled by
# the reader of this file, not here).
# Usage: bash tools/pmc_c4_traffic.sh <outdir>
OUT=$(realpath -m $1); mkdir -p $OUT; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=$OUT/$c
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d $d -o p -- python $OLDPWD/tools/bench_configs.py C4 > $d.log 2>&1)
  echo "## $c (KiB per launch, raw counter)"
  python tools/pmc_summary.py $d | grep -E "k1_cdfa|k_compact1|k2w_classify|k2_classes_all"
  rm -rf $d
done
