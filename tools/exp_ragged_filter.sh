#!/bin/bash
# C4-shard (+ paths-shaped list) timing of the ragged filter variants (each in its own process: the knobs are read once)
OUT=${1:-gpurun_out/ragged}; mkdir -p $OUT
run() { echo "== $*" | tee -a $OUT/variants.log; env "$@" python tools/bench_configs.py C4 PATHS 2>&1 | grep -v amdgpu.ids | tee -a $OUT/variants.log; }
run FZB_FILTER_VIEW=0 FZB_NO_CDFA=1   # canonical layout, byte automaton (burst form): round 2's filter
run FZB_FILTER_VIEW=0                 # canonical layout, class-composite automaton
run FZB_FILTER_VIEW=1                 # interleaved length-sorted view + class-composite automaton (the default)
run FZB_FILTER_VIEW=1 FZB_CDFA_NODFA=1   # the view's loads alone (results meaningless)
