#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
WIN_MODES=0,1 timeout 600 python tools/attn_win_check.py all time 2>&1 | grep -v "vs float64\|lse:" | tail -9
python tools/stress_attn_win.py 5 12 4 2>&1 | tail -3
for o in attn_win=1 attn_win=0 attn_win=1; do echo "== $o"; MEMHIP_OPTS=$o python tools/bench_vitl.py 64 4 2>&1 | tail -1 | cut -c1-200; done
