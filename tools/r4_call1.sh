#!/bin/bash
# round 4, GPU call 1: new parity tests, MFMA micro-probe, PMC of the dense kernels, noslp A/B
mkdir -p gpurun_out/r4c1
./tools/probe/mfma_probe > gpurun_out/r4c1/mfma_probe.txt 2>&1; cat gpurun_out/r4c1/mfma_probe.txt
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_hub_rows.py "tests/test_gpu_baseline_shapes.py::test_c5_full_batch_dense_hub_form_against_oracle_slices" "tests/test_gpu_baseline_shapes.py::test_c4_full_batch_against_oracle_slices" tests/test_abi_and_host.py 2>&1 | tail -15 | tee gpurun_out/r4c1/pytest_new.txt
rocprofv3 -L > gpurun_out/r4c1/counters.txt 2>&1
bash tools/pmc_dense.sh r4c1 C2 2>&1 | tail -120
GNNRAG_TUNE_GEMM=1 python tools/tune_variants.py --run default noslp default noslp 2>&1 | tee gpurun_out/r4c1/tune.txt
