#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "attn or window or rel" 2>&1 | tail -3 > gpurun_out/r14_tests.txt
: > gpurun_out/r14_time.txt
for i in 1 2; do
for v in mem_amd/exp/e17.so ""; do
MEMHIP_LIB=$v python tools/attn16_time.py 2>&1 | tail -1 >> gpurun_out/r14_time.txt
done; done
cat gpurun_out/r14_tests.txt gpurun_out/r14_time.txt
