#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 600 python tools/attn_win_check.py fwd time 2>&1 | tee gpurun_out/r05_attn_win_fwd.txt
