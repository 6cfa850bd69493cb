#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
F="--steps 40 --warmup 10 --no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-entrypoint-figure --no-config5-figure --no-config4-figure"
for rep in 1 2 3; do
  for V in 1 85 70; do
    MEMHIP_LIB=mem_amd/exp/gfill.so python bench.py $F --opt tn_group=$V > gpurun_out/r05_gf_${V}_$rep.json 2> gpurun_out/r05_gf_${V}_$rep.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r05_gf_${V}_$rep.json").read().strip().splitlines()[-1])
print("group fill $V rep $rep ms_per_step", d["ms_per_step"], "p50", d.get("ms_per_step_p50"))
PY
  done
done 2>&1 | tee gpurun_out/r05_group_fill_ab.txt
