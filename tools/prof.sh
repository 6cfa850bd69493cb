#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- python bench.py --no-cpu-baseline --no-gemm-timer --no-tokenizer-figure --no-raster-figure --no-config4-figure --no-config5-figure --no-entrypoint-figure "$@" > gpurun_out/$tag.log 2>&1
tail -1 gpurun_out/$tag.log | cut -c1-300
f=$(ls gpurun_out/$tag/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/${tag}_kernel_stats.csv
