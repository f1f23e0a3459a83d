#!/bin/bash
mkdir -p gpurun_out/r4c7
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_backward.py 2>&1 | tail -12 | tee gpurun_out/r4c7/pytest_bwd.txt
python tools/time_backward.py --workload C2 2>&1 | tail -3 | tee gpurun_out/r4c7/time_backward_C2.txt
