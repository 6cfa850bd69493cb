#!/bin/bash
mkdir -p gpurun_out
python tools/p8d_skew.py > gpurun_out/p8d_skew.log 2>&1; cat gpurun_out/p8d_skew.log
MEMHIP_LIB=mem_amd/exp/p8dstamp.so python - > gpurun_out/p8d_stamps_skew.log 2>&1 <<'PY'
import sys; sys.argv=['tools/p8d_stamps.py']; __file__='tools/p8d_stamps.py'
from mem_amd import _lib; _lib.set_option('gemm_stagger', -2400)
src=open('tools/p8d_stamps.py').read().split('M = 256 * 192')[0]
exec(src)
M = 256*192
for epi in ("bias","gelu"): run(M, 3072, 768, epi)
PY
cat gpurun_out/p8d_stamps_skew.log
