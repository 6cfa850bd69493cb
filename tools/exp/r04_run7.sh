#!/bin/bash
mkdir -p gpurun_out
tools/prof.sh r04a_seq --no-side-stream --steps 10 --warmup 3
rm -rf gpurun_out/r04a_seq
python tools/prof_summary.py gpurun_out/r04a_seq_kernel_stats.csv 13 2>/dev/null | head -60
