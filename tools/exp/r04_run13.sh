#!/bin/bash
# round 4, run 13: attn16 forward marginal costs (pieces removed)
cd /root/repo; mkdir -p gpurun_out; : > gpurun_out/r13_time.txt
for i in 1 2; do
for v in prev2 e18; do
MEMHIP_LIB=mem_amd/exp/$v.so python tools/attn16_time.py 2>&1 | tail -1 >> gpurun_out/r13_time.txt
done; done
cat gpurun_out/r13_time.txt
