#!/bin/bash
cd /root/repo
python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "attn" 2>&1 | tail -2
timeout 300 python tools/stress_attn16.py 3 40 10 2>&1 | tail -2
for i in 1 2; do
MEMHIP_LIB=mem_amd/exp/prev4.so python tools/attn16_time.py 2>&1 | tail -1
python tools/attn16_time.py 2>&1 | tail -1
done
