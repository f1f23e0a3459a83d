#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
timeout 300 bash tools/profile_forward.sh d200 16 sections GNNRAG_PROFILE_WARM=8 GNNRAG_PROFILE_CALLS=30 2>&1 | grep "per call, ms\|steady"| cut -c1-700
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
OMP_NUM_THREADS=4 MKL_NUM_THREADS=4 timeout 300 bash tools/profile_forward.sh d200 16 sections GNNRAG_PROFILE_WARM=8 GNNRAG_PROFILE_CALLS=30 OMP_NUM_THREADS=4 2>&1 | grep "per call, ms\|steady" | cut -c1-700
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
timeout 300 bash tools/profile_forward.sh d200 16 sections GNNRAG_PROFILE_WARM=8 GNNRAG_PROFILE_CALLS=30 OMP_WAIT_POLICY=PASSIVE GOMP_SPINCOUNT=0 2>&1 | grep "per call, ms\|steady" | cut -c1-700
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
timeout 300 bash tools/profile_forward.sh d200 16 sections GNNRAG_PROFILE_WARM=8 GNNRAG_PROFILE_CALLS=30 GPU_MAX_HW_QUEUES=2 HSA_ENABLE_INTERRUPT=0 2>&1 | grep "per call, ms\|steady" | cut -c1-700
