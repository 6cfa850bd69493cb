#!/bin/bash
mkdir -p gpurun_out
for b in 16 32 64; do python tools/bench_vitl.py $b 5 2>&1 | tail -1; done | tee gpurun_out/r04_vitl_batch.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_vitl -- python tools/bench_vitl.py 16 5 > gpurun_out/r04_vitl.log 2>&1
f=$(ls gpurun_out/r04_vitl/*/*kernel_stats.csv | head -1); cp $f gpurun_out/r04_vitl_kernel_stats.csv; rm -rf gpurun_out/r04_vitl
python tools/prof_summary.py gpurun_out/r04_vitl_kernel_stats.csv 7 24
