cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh r06j > gpurun_out/r06j_refresh.log 2>&1
tail -30 gpurun_out/r06j_refresh.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06j/gpu_tests.log 2>&1
tail -3 gpurun_out/r06j/gpu_tests.log
