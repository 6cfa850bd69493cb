#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for V in ship winlate ship winlate; do echo "== $V"; if [ $V = ship ]; then L=""; else L="mem_amd/exp/$V.so"; fi; WIN_MODES=1 MEMHIP_LIB=$L timeout 200 python tools/attn_win_check.py fwd time 2>&1 | grep "^mode 1: fwd"; done
