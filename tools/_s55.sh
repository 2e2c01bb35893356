cd $GRAFT_REPO_ROOT
for f in 0 1; do
  if [ $f = 1 ]; then export FZB_EXP_FLIP=1; fi
  python tools/bench_configs.py C2 C3 2>&1 | grep -v amdgpu | cut -c1-260
done
