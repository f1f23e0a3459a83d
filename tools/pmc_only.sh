#!/bin/bash
# The two PMC passes per workload that profiles/pmc_traffic*.json come from (part of tools/refresh_profiles.sh), alone.
# Usage (GPU box): bash tools/pmc_only.sh <tag>
set -u
TAG=${1:-rXX}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export GNNRAG_COMMIT=${GNNRAG_COMMIT:-$(cat $R/.commit_stamp 2>/dev/null || echo unknown)}
cd /tmp && export TMPDIR=/tmp
for W in C2 C5; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$W -o $C -- python $R/tools/prof_ops.py --workload $W --reps 4 --ops agg,aggfd,fr,updfd > $OUT/pmc_${W}_$C.log 2>&1
  done
  ( cd $R && python tools/make_pmc_traffic.py $(find $OUT/pmc_$W -name 'FETCH_SIZE_results.db' | head -1) $(find $OUT/pmc_$W -name 'WRITE_SIZE_results.db' | head -1) $OUT/pmc_traffic_$W.json $W > $OUT/make_pmc_$W.log 2>&1 )
done
find $OUT -name '*.db' -delete
cat $OUT/make_pmc_C2.log $OUT/make_pmc_C5.log
