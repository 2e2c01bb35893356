#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6s10; mkdir -p $OUT; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_coop.py tests/test_gpu_knobs.py tests/test_gpu_ragged_view.py -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
for v in default 0 32768 131072 100000000; do
  if [ $v = default ]; then unset FZB_COOP_BELOW; else export FZB_COOP_BELOW=$v; fi
  echo "== FZB_COOP_BELOW=$v"
  timeout 600 python tools/bench_configs.py PATHS PATHSVAR PATHSSMALL C4small 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    if 'ms_per_step' in j: print('  %-70s %.4f ms  multi=%s' % (j['config'][:70], j['ms_per_step'], j.get('multi_chunk_scored')))
"
done
unset FZB_COOP_BELOW
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $GRAFT_REPO_ROOT/tools/bench_configs.py PATHS > $OUT/prof.log 2>&1); f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/prof_summary.py $f $OUT/paths_kernel_stats.csv | head -8; rm -rf $OUT/prof
