#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_vitl64 -- python tools/bench_vitl.py 64 4 > gpurun_out/r05_vitl64.log 2>&1
cp $(ls gpurun_out/r05_vitl64/*/*kernel_stats.csv | head -1) gpurun_out/r05_vitl64_kernel_stats.csv; rm -rf gpurun_out/r05_vitl64
tail -1 gpurun_out/r05_vitl64.log
