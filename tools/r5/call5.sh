#!/bin/bash
# round 5, GPU call 5: the whole GPU suite on the pruned tree, then the profile refresh (bench, rocprof stats, PMC traffic,
# other workloads), the dense layer's PMC picture and the RCCL path at world size 1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5e
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest_all rc=$?"
tail -3 $O/pytest_all.log
bash tools/refresh_profiles.sh r5e 2>&1 | tail -20
bash tools/pmc_dense.sh r5e C2 > $O/pmc_dense_stdout.log 2>&1
GNNRAG_FORCE_DIST=1 timeout 400 python bench.py --gpus 1 --no-cpu-baseline --no-e2e --spread-steps 0 > $O/bench_force_dist.log 2>&1; echo "force_dist rc=$?"
tail -1 $O/bench_force_dist.log | cut -c1-400
