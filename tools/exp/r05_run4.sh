#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for V in ship winprio0 ship winprio0; do echo "== $V"; if [ $V = ship ]; then L=""; else L="mem_amd/exp/$V.so"; fi; WIN_MODES=1 MEMHIP_LIB=$L timeout 300 python tools/attn_win_check.py all time 2>&1 | grep "^mode"; done | tee gpurun_out/r05_attn_win_prio.txt
