#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_gemm_gpu.py -q -x 2>&1 | tail -3
F="--steps 40 --warmup 10 --no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-entrypoint-figure --no-config5-figure --no-config4-figure --no-gemm-timer"
for rep in 1 2 3; do
  for V in 1 0; do
    python bench.py $F --opt gemm_p8_pair=$V > gpurun_out/r05_pair_$V.json 2> gpurun_out/r05_pair_$V.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r05_pair_$V.json").read().strip().splitlines()[-1])
print("pair=$V rep $rep", d["ms_per_step"], d.get("ms_per_step_p50"))
PY
  done
done 2>&1 | tee gpurun_out/r05_pair_ab.txt
