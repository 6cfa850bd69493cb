#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for V in ship win1 win2 win3; do echo "== $V"; if [ $V = ship ]; then L=""; else L="mem_amd/exp/$V.so"; fi; MEMHIP_LIB=$L timeout 300 python tools/attn_win_check.py fwd time 2>&1 | grep "^mode"; done | tee gpurun_out/r05_attn_win_fwd_exp.txt
