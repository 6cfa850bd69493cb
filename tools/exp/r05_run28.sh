#!/bin/bash
# round 5, call 28: options tuned on kernels alone, re-swept inside the two-stream step (ln_bwd_grid = 2048 is the default now)
cd /root/repo; mkdir -p gpurun_out
F="--steps 40 --warmup 10 --no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-entrypoint-figure --no-config5-figure --no-config4-figure"
for rep in 1 2; do
  for V in "ln_bwd_grid=2048" "attn16_stagger=0" "attn16_stagger=20000" "attn16_stagger=80000" "attn16_stagger_fwd=20000" "gemm_stagger=1000" "gemm_stagger=4000"; do
    python bench.py $F --opt $V > gpurun_out/r05_sw_$rep.json 2> gpurun_out/r05_sw_$rep.err
    python - <<PY
import json
d=json.loads(open("gpurun_out/r05_sw_$rep.json").read().strip().splitlines()[-1])
print("$V rep $rep ms_per_step", d["ms_per_step"], "p50", d.get("ms_per_step_p50"))
PY
  done
done 2>&1 | tee gpurun_out/r05_option_sweep.txt
