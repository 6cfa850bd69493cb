#!/bin/bash
# round 4, run 12: attn16 forward with aligned ds_read_b64 bias reads
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "attn16 or attn" 2>&1 | tail -3 > gpurun_out/r12_tests.txt
for i in 1 2; do
MEMHIP_LIB=mem_amd/exp/prev2.so python tools/attn16_time.py 2>&1 | tail -1 >> gpurun_out/r12_time.txt
python tools/attn16_time.py 2>&1 | tail -1 >> gpurun_out/r12_time.txt
done
MEMHIP_LIB=mem_amd/exp/timing.so python tools/attn16_sections.py > gpurun_out/r12_sections.txt 2>&1
cat gpurun_out/r12_tests.txt gpurun_out/r12_time.txt gpurun_out/r12_sections.txt
