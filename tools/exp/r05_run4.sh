#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
WIN_MODES=0,1 timeout 300 python tools/attn_win_check.py fwd time 2>&1 | grep -v "mode 1 lse" | tail -6
MEMHIP_LIB=mem_amd/exp/winstamp.so timeout 300 python tools/attn_win_stamps.py
