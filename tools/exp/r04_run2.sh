#!/bin/bash
mkdir -p gpurun_out
MEMHIP_LIB=mem_amd/exp/p8dstamp.so python tools/p8d_stamps.py > gpurun_out/p8d_stamps.log 2>&1
cat gpurun_out/p8d_stamps.log
MEMHIP_LIB=mem_amd/exp/stamp.so python -c "
import sys; sys.argv=['tools/p8_stamps.py']
from mem_amd import _lib; _lib.set_option('gemm_p8d',0)
exec(open('tools/p8_stamps.py').read())" > gpurun_out/p8_stamps.log 2>&1
cat gpurun_out/p8_stamps.log
