cd $GRAFT_REPO_ROOT; O=gpurun_out/r6s56; mkdir -p $O
timeout 1000 python tools/soak_parity.py 900 2>&1 | grep -v amdgpu | tail -2 | tee $O/soak.txt
timeout 500 python tools/soak_parity.py 420 reuse 2>&1 | grep -v amdgpu | tail -1 | tee -a $O/soak.txt
