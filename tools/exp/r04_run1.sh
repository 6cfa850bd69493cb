#!/bin/bash
# round 4, run 1: gemm_p8d against gemm_p8 (results + times), the GEMM / model tests, interleaved step A/B
mkdir -p gpurun_out
python tools/p8d_check.py > gpurun_out/p8d_check.log 2>&1; echo "p8d_check rc=$?" >> gpurun_out/p8d_check.log
tail -40 gpurun_out/p8d_check.log
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_model_gpu.py -x -q > gpurun_out/r04_run1_tests.log 2>&1; tail -5 gpurun_out/r04_run1_tests.log
bash tools/ab_bench.sh "--opt gemm_p8d=0" "--opt gemm_p8d=1" 3 > gpurun_out/r04_run1_ab.log 2>&1; cat gpurun_out/r04_run1_ab.log
