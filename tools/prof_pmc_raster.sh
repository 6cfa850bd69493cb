#!/bin/bash
# usage (GPU box, repo root): tools/prof_pmc_raster.sh <tag>  -- HBM traffic of the rasterizer kernels at BASELINE configs[3]
# size (tools/raster_bench.py, B=64 x 1 M events), FETCH_SIZE / WRITE_SIZE in separate passes.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/${tag}_$c -- python tools/raster_bench.py > gpurun_out/${tag}_$c.log 2>&1
  f=$(ls gpurun_out/${tag}_$c/*/*counter_collection.csv | head -1)
  python tools/pmc_summary.py $f $c > gpurun_out/${tag}_$c.json
  rm -rf gpurun_out/${tag}_$c
done
python tools/pmc_combine.py gpurun_out/${tag}_FETCH_SIZE.json gpurun_out/${tag}_WRITE_SIZE.json gpurun_out/${tag}_traffic.json | grep -A5 raster
