#!/bin/bash
# round 5, GPU call 10: host profile of a steady-state forward at d200 (batch 64: 12 calls in the run; batch 16: 43)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 bash tools/profile_forward.sh d200 64 torch GNNRAG_PROFILE_WARM=4 GNNRAG_PROFILE_CALLS=8 2>&1 | tail -60
timeout 300 bash tools/profile_forward.sh d200 64 cprofile GNNRAG_PROFILE_WARM=4 GNNRAG_PROFILE_CALLS=8 2>&1 | tail -60
timeout 300 bash tools/profile_forward.sh d200 64 torchcuda GNNRAG_PROFILE_WARM=4 GNNRAG_PROFILE_CALLS=8 2>&1 | tail -5
timeout 300 bash tools/profile_forward.sh d200 16 cprofile 2>&1 | tail -3
