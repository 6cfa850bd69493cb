#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gemm_gpu.py tests/test_model_gpu.py tests/test_kernels_gpu.py -x -q > gpurun_out/r04_run5_tests.log 2>&1; tail -3 gpurun_out/r04_run5_tests.log
bash tools/ab_bench.sh "--gelu-dg 0" "--gelu-dg 1" 3 > gpurun_out/r04_run5_ab.log 2>&1; cat gpurun_out/r04_run5_ab.log
