#!/bin/bash
# round 5, call 22: engine.wgrad_group 3 (the four weight gradients of a block as ONE launch, MLP gradient buffers alternating per layer)
# against 2, interleaved; engine test first
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_model_gpu.py -q -x -k "grouped_weight or work_skipping or tail_rows" 2>&1 | tail -2
F="--steps 40 --warmup 10 --no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-entrypoint-figure --no-config5-figure --no-config4-figure"
for rep in 1 2 3; do
  for V in 2 3; do
    python bench.py $F --wgrad-group $V > gpurun_out/r05_g4_${V}_$rep.json 2> gpurun_out/r05_g4_${V}_$rep.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r05_g4_${V}_$rep.json").read().strip().splitlines()[-1])
print("wgrad_group $V rep $rep ms_per_step", d["ms_per_step"], "p50", d.get("ms_per_step_p50"), "loss", d["config"]["last_loss"])
PY
  done
done 2>&1 | tee gpurun_out/r05_group4_ab.txt
