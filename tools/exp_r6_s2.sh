#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6s2; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
EXP_LEAN=1 timeout 900 python tools/exp_r6_corun2.py > $OUT/corun2_lean.log 2> $OUT/corun2.err; echo "corun2 rc=$?"
cat $OUT/corun2_lean.log; tail -3 $OUT/corun2.err
