cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_main_py.py -m gpu -q -k "fb6k or cwqflags or normpos or d200eps" -s 2>&1 | grep -v "^$" | tail -40
