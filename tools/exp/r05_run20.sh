#!/bin/bash
# round 5, call 20: the weight-gradient stream plans for fewer CUs (--side-cus: fewer, longer row slices, fewer slabs; the main stream
# fills what it leaves idle), interleaved; wgrad_group 1
cd /root/repo; mkdir -p gpurun_out
F="--steps 40 --warmup 10 --no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-entrypoint-figure --no-config5-figure --no-config4-figure"
for rep in 1 2 3; do
  for V in 0 192 144 112; do
    python bench.py $F --side-cus $V > gpurun_out/r05_sc_${V}_$rep.json 2> gpurun_out/r05_sc_${V}_$rep.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r05_sc_${V}_$rep.json").read().strip().splitlines()[-1])
print("side_cus $V rep $rep ms_per_step", d["ms_per_step"], "p50", d.get("ms_per_step_p50"))
PY
  done
done 2>&1 | tee gpurun_out/r05_side_cus_ab.txt
