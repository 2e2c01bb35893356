cd $GRAFT_REPO_ROOT; O=gpurun_out/r6s59; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-300 $O/bench.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/profcfg -o p -- python $GRAFT_REPO_ROOT/tools/bench_configs.py C3 C4 C5 > $GRAFT_REPO_ROOT/$O/profcfg.log 2>&1); f=$(find $O/profcfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/prof_summary.py $f $O/cfg_kernel_stats.csv; rm -rf $O/profcfg
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/profp -o p -- python $GRAFT_REPO_ROOT/tools/bench_configs.py PATHS PATHSSMALL > $GRAFT_REPO_ROOT/$O/profp.log 2>&1); f=$(find $O/profp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/prof_summary.py $f $O/paths_kernel_stats.csv; rm -rf $O/profp
head -12 $O/cfg_kernel_stats.csv; head -12 $O/paths_kernel_stats.csv
python tools/_exp_enqueue.py 2>&1 | grep -v amdgpu | tee $O/enqueue.txt
