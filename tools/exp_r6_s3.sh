#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6s3; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace3 -o t -- python $GRAFT_REPO_ROOT/tools/exp_r6_corun3.py > $OUT/corun3.log 2>&1)
f=$(find $OUT/trace3 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && cp $f $OUT/corun_part3_kernel_trace.csv; rm -rf $OUT/trace3
timeout 300 python tools/exp_r6_corun3.py > $OUT/corun3_plain.log 2>&1
tail -3 $OUT/corun3.log; tail -2 $OUT/corun3_plain.log
