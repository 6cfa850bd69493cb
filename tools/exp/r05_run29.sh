#!/bin/bash
# round 5, call 29: NT GEMM launches with 2 / 4 workgroups per CU instead of the persistent one (finer dispatch granularity beside the
# weight-gradient stream), inside the step
cd /root/repo; mkdir -p gpurun_out
F="--steps 40 --warmup 10 --no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-entrypoint-figure --no-config5-figure --no-config4-figure"
for rep in 1 2; do
  for V in default wg2 wg4; do
    if [ $V = default ]; then L=""; else L="mem_amd/exp/$V.so"; fi
    MEMHIP_LIB=$L python bench.py $F > gpurun_out/r05_wgm_$rep.json 2> gpurun_out/r05_wgm_$rep.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r05_wgm_$rep.json").read().strip().splitlines()[-1])
print("$V rep $rep ms_per_step", d["ms_per_step"], "p50", d.get("ms_per_step_p50"))
PY
  done
done
