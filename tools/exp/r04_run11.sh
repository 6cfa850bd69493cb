#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_vitl64 -- python tools/bench_vitl.py 64 3 > gpurun_out/r04_vitl64.log 2>&1
f=$(ls gpurun_out/r04_vitl64/*/*kernel_stats.csv | head -1); cp $f gpurun_out/r04_vitl64_kernel_stats.csv; rm -rf gpurun_out/r04_vitl64
tail -1 gpurun_out/r04_vitl64.log
python tools/prof_summary.py gpurun_out/r04_vitl64_kernel_stats.csv 5 16
