#!/bin/bash
cd /root/repo
for V in "$@"; do MEMHIP_LIB=mem_amd/exp/$V.so python -m pytest tests/test_gemm_gpu.py -q -x 2>&1 | tail -1; done
for i in 1 2; do
python tools/bench_gemm.py 2>&1 | grep "^qkv\|^fc1\|^fc2\|^proj\|^sq8k" | sed 's/^/base /'
for V in "$@"; do MEMHIP_LIB=mem_amd/exp/$V.so python tools/bench_gemm.py 2>&1 | grep "^qkv\|^fc1\|^fc2\|^proj\|^sq8k" | sed "s/^/$V /"; done
done
