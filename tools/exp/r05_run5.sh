#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
for m in 1 0; do echo "== mode $m"; tools/prof_attn_win.sh r05_attnwin_pmc $m; done 2>&1 | tee gpurun_out/r05_attn_win_pmc.txt
