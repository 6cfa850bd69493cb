#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
WIN_MODES=0,1,3 timeout 300 python tools/attn_win_check.py fwd time 2>&1 | grep -v "lse:\|float64" | tail -9 | tee gpurun_out/r05_attn_win_fwd6.txt
