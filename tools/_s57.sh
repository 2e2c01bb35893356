cd $GRAFT_REPO_ROOT
python tools/_exp_enqueue.py 2>&1 | grep -v amdgpu
