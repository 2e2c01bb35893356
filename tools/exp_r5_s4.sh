#!/bin/bash
# round-5 session 4: where the window kernel's PRE form spends its time on the Arabic-shaped 1-typo query (FZB_WINDOW_DBG, measurement only),
# two C2 queries in flight under the k1_dfa variants.  Usage: bash tools/exp_r5_s4.sh <outdir>
OUT=$(realpath -m $1); mkdir -p $OUT; export TMPDIR=/tmp
for d in 0 1 2 3 4 6; do
  echo "== arabic 1 typo, FZB_WINDOW_DBG=$d (1: no masks ahead, 2: no walk, 4: walk of the first chunk only, no unaligned end scan)"
  FZB_WINDOW_DBG=$d python tools/exp_r5_ab.py arabic1 2>/dev/null | cut -c1-260
done
echo "== two C2 queries in flight (tools/exp_concurrency.py): default / FZB_DFA_STRIDE256=1 / FZB_DFA_GENERAL=1"
for v in "" "FZB_DFA_STRIDE256=1" "FZB_DFA_GENERAL=1"; do echo "-- $v"; env $v python tools/exp_concurrency.py 2>/dev/null | tail -3; done
