cd $GRAFT_REPO_ROOT; O=gpurun_out/r6s53; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
