#!/bin/bash
cd /root/repo
python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "attn" 2>&1 | tail -2
python -m pytest tests -q -x -m gpu -k "c5 or long or config5" 2>&1 | tail -2
for i in 1 2; do
MEMHIP_LIB=mem_amd/exp/prev3.so python tools/bench_attn_stream.py 16 2>&1 | tail -1
python tools/bench_attn_stream.py 16 2>&1 | tail -1
done
