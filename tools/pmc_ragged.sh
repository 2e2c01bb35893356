#!/bin/bash
# Memory-pipeline counters of the ragged filter (C4 shard), one rocprofv3 --pmc pass per set (counters only, no trace domains).
# Usage: [PMC_RAGGED_SETS=n] bash tools/pmc_ragged.sh <outdir>
OUT=$(realpath -m $1); mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA|TCP|TD|TCC)_[A-Z0-9_]+" | sort -u | tr "\n" " " > $OUT/available.txt
i=0
for c in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE" "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  [ -n "$PMC_RAGGED_SETS" ] && [ $i -gt $PMC_RAGGED_SETS ] && break   # PMC_RAGGED_SETS=n: only the first n counter sets
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/p$i -o p -- python $OLDPWD/tools/bench_configs.py C4 > $OUT/p$i.log 2>&1)
  python tools/pmc_summary.py $OUT/p$i | grep -E "ragged|k1_dfa|k1_cdfa"
  rm -rf $OUT/p$i
done
