#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof_mfma.sh <tag>
# MFMA-pipe utilisation of every kernel of the bench step from the SQ counters (own pass, counters only next to
# --kernel-trace): SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs) per kernel family.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/${tag}_$c -- python bench.py --no-cpu-baseline --no-gemm-timer --no-tokenizer-figure --no-raster-figure --no-config4-figure --no-config5-figure --no-entrypoint-figure --no-side-stream --steps 2 --warmup 1 "$@" > gpurun_out/${tag}_$c.log 2>&1
  f=$(ls gpurun_out/${tag}_$c/*/*counter_collection.csv | head -1)
  python tools/pmc_summary.py $f $c split > gpurun_out/${tag}_$c.json
  tail -c 300 gpurun_out/${tag}_$c.json
  rm -rf gpurun_out/${tag}_$c
done
