#!/bin/bash
cd /root/repo
V=${1:-barmid}
MEMHIP_LIB=mem_amd/exp/$V.so python -m pytest tests/test_gemm_gpu.py -q -x 2>&1 | tail -2
for i in 1 2; do
python tools/bench_gemm.py 2>&1 | grep -v "^w_\|amdgpu" | sed 's/^/base /'
MEMHIP_LIB=mem_amd/exp/$V.so python tools/bench_gemm.py 2>&1 | grep -v "^w_\|amdgpu" | sed "s/^/$V /"
done
