#!/bin/bash
# one-off timing batch (each variant in its own process: the knobs are read once)
OUT=${1:-gpurun_out/batch}; mkdir -p $OUT
run() { cfgs=$1; shift; echo "== $cfgs $*" | tee -a $OUT/variants.log; env "$@" timeout 300 python tools/bench_configs.py $cfgs 2>&1 | grep -v amdgpu.ids | cut -c1-420 | tee -a $OUT/variants.log; }
run "C4" FZB_SMALL_LIST=0
run "C4" FZB_X=1
run "C4" FZB_SMALL_LIST=0
run "C4" FZB_X=1
