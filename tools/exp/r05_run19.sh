#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
F="--steps 40 --warmup 10 --no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-entrypoint-figure --no-config5-figure --no-config4-figure"
for rep in 1 2 3; do
  for V in g1 g2s7 g2s3; do
    if [ $V = g1 ]; then X="--wgrad-group 1"; O=""; fi
    if [ $V = g2s7 ]; then X="--wgrad-group 2"; O=""; fi
    if [ $V = g2s3 ]; then X="--wgrad-group 2"; O="--opt tn_group=2"; fi
    python bench.py $F $X $O > gpurun_out/r05_g_${V}_$rep.json 2> gpurun_out/r05_g_${V}_$rep.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r05_g_${V}_$rep.json").read().strip().splitlines()[-1])
print("$V rep $rep ms_per_step", d["ms_per_step"], "p50", d.get("ms_per_step_p50"))
PY
  done
done
