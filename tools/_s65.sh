cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_cpp_facade.py -m gpu -q 2>&1 | tail -3
timeout 60 tests/cpp/rccl_ranks 1 100000 2>&1 | grep -v amdgpu | tail -4
timeout 60 tests/cpp/rccl_ranks 2 1000 2>&1 | grep -v amdgpu | tail -4
