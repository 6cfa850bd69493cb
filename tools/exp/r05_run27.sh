#!/bin/bash
# round 5, call 27: ln_bwd_grid (workgroups of the LayerNorm-backward kernels) re-swept inside the step
cd /root/repo; mkdir -p gpurun_out
F="--steps 40 --warmup 10 --no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-entrypoint-figure --no-config5-figure --no-config4-figure"
for rep in 1 2; do
  for V in 2048 3072 4096 6144 8192; do
    python bench.py $F --opt ln_bwd_grid=$V > gpurun_out/r05_lng_${V}_$rep.json 2> gpurun_out/r05_lng_${V}_$rep.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r05_lng_${V}_$rep.json").read().strip().splitlines()[-1])
print("ln_bwd_grid $V rep $rep ms_per_step", d["ms_per_step"], "p50", d.get("ms_per_step_p50"))
PY
  done
done
