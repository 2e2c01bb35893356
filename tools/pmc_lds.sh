#!/bin/bash
# LDS / issue counters of the bench's kernels (one rocprofv3 --pmc pass per set; counters only).  Usage: bash tools/pmc_lds.sh <outdir> [env...]
OUT=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  (cd /tmp && env "$@" timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/p$i -o p -- python $OLDPWD/bench.py --steps 5 --warmup 2 --fast --no-check > $OUT/p$i.log 2>&1)
  python tools/pmc_summary.py $OUT/p$i
  rm -rf $OUT/p$i
done
