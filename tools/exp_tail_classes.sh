#!/bin/bash
# C4-shard (+ paths-shaped list) timing with / without the multi-chunk tail classes (each in its own process: the knobs are read once)
OUT=${1:-gpurun_out/tail}; mkdir -p $OUT
run() { echo "== $*" | tee -a $OUT/variants.log; env "$@" python tools/bench_configs.py C4 PATHS 2>&1 | grep -v amdgpu.ids | tee -a $OUT/variants.log; }
run FZB_NO_TAIL_CLASSES=1
run FZB_X=1
