#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/r05_gpu_tests_mid.txt
