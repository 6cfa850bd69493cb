#!/bin/bash
# weight-gradient GEMM: what do the slab write-out (one fp32 256x256 slab per workgroup) and the reduction pass cost?
cd /root/repo; mkdir -p gpurun_out
for V in ship tnnostore ship tnnostore; do echo "== $V"; if [ $V = ship ]; then L=""; else L="mem_amd/exp/$V.so"; fi; MEMHIP_LIB=$L python tools/bench_gemm.py 2>&1 | grep "^w_"; done | tee gpurun_out/r05_tn_nostore.txt
