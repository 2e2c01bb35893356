#!/bin/bash
# One gpurun call's worth of work on the MI355X box: GPU tests, the bench line, rocprofv3 kernel stats, the other configs, micro-benchmarks.
# Usage (from the repo root on the GPU box): bash tools/gpu_session.sh <tag> [steps...]   steps default: test bench prof cfg ubench probe
TAG=${1:-run}; shift
STEPS=${@:-test bench prof cfg ubench probe}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
[ -z "$GRAFT_REPO_ROOT" ] && OUT=$(pwd)/gpurun_out/$TAG
mkdir -p $OUT
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for s in $STEPS; do
  t0=$(date +%s)
  case $s in
    test) timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -5 $OUT/pytest.log;;
    testfull) timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -15 $OUT/pytest.log;;
    bench) timeout 600 python bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt; cat $OUT/bench.json;;
    benchfast) timeout 600 python bench.py --steps 50 --warmup 5 --fast > $OUT/bench_fast.json 2> $OUT/bench.err; echo "bench rc=$?" | tee -a $OUT/summary.txt; cat $OUT/bench_fast.json;;
    prof) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $OLDPWD/bench.py --steps 20 --warmup 3 --fast > $OUT/prof.log 2>&1); f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then python tools/prof_summary.py $f $OUT/kernel_stats.csv; cp $f $OUT/kernel_stats_raw.csv; else tail -20 $OUT/prof.log; fi; rm -rf $OUT/prof;;
    profcfg) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/profcfg -o p -- python $OLDPWD/tools/bench_configs.py C3 C4 C5 > $OUT/profcfg.log 2>&1); f=$(find $OUT/profcfg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/prof_summary.py $f $OUT/cfg_kernel_stats.csv; rm -rf $OUT/profcfg;;
    cfg) timeout 900 python tools/bench_configs.py C3 C4 C5 > $OUT/cfg.log 2>&1; cat $OUT/cfg.log;;
    pmcsq) for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE GRBM_COUNT"; do n=$(echo $c | tr ' ' '_'); (cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$n -o p -- python $OLDPWD/bench.py --steps 5 --warmup 2 --fast > $OUT/pmc_$n.log 2>&1); done; for d in $OUT/pmc_*/; do python tools/pmc_summary.py $d; done > $OUT/pmc_sq.txt 2>&1; cat $OUT/pmc_sq.txt; python tools/pmc_sq_json.py $OUT/pmc_sq.txt $OUT/sq.json; rm -rf $OUT/pmc_*/;;
    pmctraffic) for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python $OLDPWD/bench.py --steps 5 --warmup 2 --fast > $OUT/pmc_$c.log 2>&1); done; python tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/traffic.json; rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE;;
    pmcc4) bash tools/pmc_c4_traffic.sh $OUT/pmc_c4 > $OUT/pmc_c4_traffic.txt 2>&1; cat $OUT/pmc_c4_traffic.txt; rm -rf $OUT/pmc_c4;;
    ubench) timeout 120 tools/ubench/valu_rate > $OUT/valu_rate.txt 2>&1; cat $OUT/valu_rate.txt;;
    probe) { echo "== cargo/rustc probe on the GPU box"; which cargo rustc 2>&1; cargo --version 2>&1; rustc --version 2>&1; ls ~/.cargo 2>&1 | head -3; echo "== cpu"; lscpu | egrep "Model name|^CPU\(s\)|Socket|Thread|Core|MHz|Flags" | cut -c1-400; echo "== gpu"; rocm-smi --showclocks 2>&1 | head -30; } > $OUT/probe.txt 2>&1; head -40 $OUT/probe.txt;;
    *) if [ -f "$s" ]; then timeout 900 python $s > $OUT/$(basename $s .py).log 2>&1; tail -30 $OUT/$(basename $s .py).log; else echo "unknown step $s"; fi;;
  esac
  echo "step $s took $(( $(date +%s) - t0 )) s" | tee -a $OUT/summary.txt
done
