#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r05_attnwin_trace
WIN_MODES=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_attnwin_trace -- python tools/attn_win_check.py all time > gpurun_out/r05_attnwin_trace.log 2>&1
f=$(ls gpurun_out/r05_attnwin_trace/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-200
cp $f gpurun_out/r05_attnwin_kernel_stats.csv; rm -rf gpurun_out/r05_attnwin_trace
