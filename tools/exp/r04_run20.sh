#!/bin/bash
cd /root/repo
for i in 1 2; do
for V in "" "$@"; do MEMHIP_LIB=${V:+mem_amd/exp/$V.so} python tools/bench_gemm.py 2>&1 | grep "^qkv\|^fc1\|^fc2\|^proj\|^sq8k" | sed "s/^/${V:-base} /"; done
done
