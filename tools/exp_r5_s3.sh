#!/bin/bash
# round-5 session 3: k1_dfa variants on the headline (bench.py --fast under the knobs), the overlap experiment with the lighter filter,
# kernel trace of the Arabic-shaped typo query, long-needle A/B.  Usage: bash tools/exp_r5_s3.sh <outdir>
OUT=$(realpath -m $1); mkdir -p $OUT; export TMPDIR=/tmp
for v in "" "FZB_DFA_STRIDE256=1" "FZB_DFA_GENERAL=1" "FZB_DFA_STRIDE256=1 FZB_DFA_WGS=6" ""; do
  echo "== bench.py --fast $v"
  env $v python bench.py --fast --steps 50 --warmup 5 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step']*1e3,1), 'us step; filter', round(j['stages']['filter_ms']*1e3,1), 'compaction', round(j['stages']['compaction_ms']*1e3,1), 'scorer', round(j['stages']['scorer_ms']*1e3,1), 'roofline frac', round(j['roofline']['frac'],3))"
done
echo "== tools/exp_overlap.py (default table pitch)"; python tools/exp_overlap.py 2>/dev/null | tail -3
echo "== tools/exp_overlap.py FZB_DFA_STRIDE256=1"; FZB_DFA_STRIDE256=1 python tools/exp_overlap.py 2>/dev/null | tail -9
echo "== SQ_INSTS_VALU / SQ_INSTS_LDS of k1_dfa per launch: default, then FZB_DFA_STRIDE256=1"
for v in "" "FZB_DFA_STRIDE256=1"; do
  (cd /tmp && env $v timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -o p -- python $OLDPWD/bench.py --fast --steps 5 --warmup 2 > $OUT/pmc_sq.log 2>&1)
  python tools/pmc_summary.py $OUT/pmc_sq | grep -E "k1_dfa|k2b_dp_short"; rm -rf $OUT/pmc_sq
done
echo "== kernel trace: arabic-shaped list, typo budgets (tools/exp_r5_ab.py arabic)"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_ar -o p -- python $OLDPWD/tools/exp_r5_ab.py arabic > $OUT/prof_ar.log 2>&1)
f=$(find $OUT/prof_ar -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python tools/prof_summary.py $f $OUT/arabic_typos_kernel_stats.csv && cat $OUT/arabic_typos_kernel_stats.csv; rm -rf $OUT/prof_ar
echo "== long needle A/B"; python tools/exp_r5_ab.py long 2>/dev/null | cut -c1-330
