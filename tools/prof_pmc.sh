#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_pmc.sh <tag>
# HBM traffic of every kernel of the bench step: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (they do not
# fit one pass on gfx950), counters only next to --kernel-trace (MI355X_MICROARCH.md, rocprofv3 PMC slots).
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/${tag}_$c -- python bench.py --no-cpu-baseline --no-gemm-timer --no-tokenizer-figure --no-raster-figure --no-config4-figure --no-config5-figure --no-entrypoint-figure --steps 2 --warmup 1 > gpurun_out/${tag}_$c.log 2>&1
  f=$(ls gpurun_out/${tag}_$c/*/*counter_collection.csv | head -1)
  python tools/pmc_summary.py $f $c > gpurun_out/${tag}_$c.json
  tail -c 600 gpurun_out/${tag}_$c.json
done
