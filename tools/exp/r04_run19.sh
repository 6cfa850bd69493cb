#!/bin/bash
cd /root/repo
V=$1
MEMHIP_LIB=mem_amd/exp/$V.so python -m pytest tests/test_gemm_gpu.py -q -x 2>&1 | tail -1
for i in 1 2; do
python tools/bench_gemm.py 2>&1 | grep "^w_\|^qkv\|^fc2" | sed 's/^/base /'
MEMHIP_LIB=mem_amd/exp/$V.so python tools/bench_gemm.py 2>&1 | grep "^w_\|^qkv\|^fc2" | sed "s/^/$V /"
done
bash tools/ab_lib.sh "" mem_amd/exp/$V.so 3
