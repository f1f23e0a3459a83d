#!/bin/bash
# round 5, GPU call 15: the final tree - profiles refresh (PMC traffic, default bench, kernel stats, other workloads),
# then the whole GPU suite and smoke
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/refresh_profiles.sh r05j 2>&1 | tail -25
O=gpurun_out/r05j
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
grep nr_throttled /sys/fs/cgroup/cpu.stat
