cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for g in 4 3 2 0; do
  if [ $g = 0 ]; then export FZB_NO_FUSED_CLASSIFY=1; echo "--- two launches"; else unset FZB_NO_FUSED_CLASSIFY; export FZB_EXP_FG=$g; echo "--- fused, grid = $g per CU"; fi
  python tools/bench_configs.py C4 PATHS PATHSSMALL 2>&1 | grep -v amdgpu | grep -v "1 typo" | cut -c12-50,88-130
done
done
