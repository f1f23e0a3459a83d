cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh r06n > gpurun_out/r06n_refresh.log 2>&1
tail -14 gpurun_out/r06n_refresh.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06n/gpu_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r06n/gpu_tests.log | tail -2
