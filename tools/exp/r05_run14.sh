#!/bin/bash
# round 5, call 14: grouped proj + qkv weight gradient -- API test, engine tests, interleaved step A/B (--no-wgrad-group)
cd /root/repo; mkdir -p gpurun_out
python -m pytest tests/test_gemm_gpu.py -q -x -k "tn_group or tn_weight" 2>&1 | tail -3
python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py tests/test_ddp_gpu.py -q -x 2>&1 | tail -4
F="--steps 40 --warmup 10 --no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-entrypoint-figure --no-config5-figure --no-config4-figure"
for rep in 1 2 3; do
  for V in group single; do
    if [ $V = single ]; then X="--no-wgrad-group"; else X=""; fi
    python bench.py $F $X > gpurun_out/r05_wg_${V}_$rep.json 2> gpurun_out/r05_wg_${V}_$rep.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r05_wg_${V}_$rep.json").read().strip().splitlines()[-1])
print("$V $rep ms_per_step", d["ms_per_step"], "p50", d.get("ms_per_step_p50"))
PY
  done
done 2>&1 | tee gpurun_out/r05_wgrad_group_ab.txt
