cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03_s27; timeout 600 python -m pytest tests/test_sparse.py -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r03_s27/fail.log; cat gpurun_out/r03_s27/fail.log
