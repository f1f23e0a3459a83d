#!/bin/bash
# round 5, GPU call 11: section timers of a steady-state forward (no profiler), with / without gc, with / without per-seam sync
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for K in sections sections_sync; do
  timeout 300 bash tools/profile_forward.sh d200 64 $K GNNRAG_PROFILE_WARM=4 GNNRAG_PROFILE_CALLS=8 2>&1 | grep -v "INFO\|it/s\|it\]" | tail -14
done
timeout 300 bash tools/profile_forward.sh d200 64 sections GNNRAG_PROFILE_WARM=4 GNNRAG_PROFILE_CALLS=8 GNNRAG_PROFILE_GC=0 2>&1 | grep -v "INFO\|it/s\|it\]" | tail -14
timeout 300 bash tools/profile_forward.sh d200 16 sections 2>&1 | grep -v "INFO\|it/s\|it\]" | tail -14
timeout 300 bash tools/profile_forward.sh d50 1 sections 2>&1 | grep -v "INFO\|it/s\|it\]" | tail -14
