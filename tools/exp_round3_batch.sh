#!/bin/bash
# one-off timing batch (each variant in its own process: the knobs are read once)
OUT=${1:-gpurun_out/batch}; mkdir -p $OUT
run() { cfgs=$1; shift; echo "== $cfgs $*" | tee -a $OUT/variants.log; env "$@" timeout 300 python tools/bench_configs.py $cfgs 2>&1 | grep -v amdgpu.ids | cut -c1-420 | tee -a $OUT/variants.log; }
run "C5" FZB_X=1
run "C5" FZB_K2U_WAVES=3
run "C4 PATHS" FZB_X=1
run "C4 PATHS" FZB_CLASSIFY_PER=1
run "C4 PATHS" FZB_CLASSIFY_PER=4
