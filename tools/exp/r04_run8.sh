#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm_gpu.py -x -q 2>&1 | tail -2
bash tools/ab_lib.sh "" mem_amd/exp/plainst.so 3 > gpurun_out/r04_run8_ab.log 2>&1; cat gpurun_out/r04_run8_ab.log
