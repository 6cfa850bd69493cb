#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
WIN_MODES=0,1 timeout 600 python tools/attn_win_check.py all time 2>&1 | grep -v "vs float64\|lse:" | tee gpurun_out/r05_attn_win_all2.txt
