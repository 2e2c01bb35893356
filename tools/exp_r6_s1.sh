#!/bin/bash
# round 6, GPU session 1: box probe, co-run experiments (times + kernel trace), baseline bench line
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6s1; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
bash tools/gpu_session.sh r6s1 probe > /dev/null 2>&1
timeout 900 python tools/exp_r6_corun.py > $OUT/corun.log 2> $OUT/corun.err; echo "corun rc=$?"
(cd /tmp && EXP_TRACE=1 EXP_PARTS=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace1 -o t -- python $GRAFT_REPO_ROOT/tools/exp_r6_corun.py > $OUT/trace1.log 2>&1)
f=$(find $OUT/trace1 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f $OUT/corun_part1_kernel_trace.csv; rm -rf $OUT/trace1
(cd /tmp && EXP_TRACE=1 EXP_PARTS=2 EXP_N4=12500000 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace2 -o t -- python $GRAFT_REPO_ROOT/tools/exp_r6_corun.py > $OUT/trace2.log 2>&1)
f=$(find $OUT/trace2 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f $OUT/corun_part2_kernel_trace.csv; rm -rf $OUT/trace2
bash tools/gpu_session.sh r6s1 benchfast > /dev/null 2>&1
cat $OUT/corun.log; tail -3 $OUT/corun.err; ls -la $OUT
