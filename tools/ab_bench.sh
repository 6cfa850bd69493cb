#!/bin/bash
# tools/ab_bench.sh "<args A>" "<args B>" [reps]: interleaved A/B runs of bench.py on ONE box (boxes differ by ~10 %)
A="$1"; B="$2"; R=${3:-2}
common="--no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-config4-figure --no-config5-figure --no-entrypoint-figure --no-gemm-timer --steps 20 --warmup 5"
for i in $(seq $R); do
  for v in A B; do
    if [ $v = A ]; then args="$A"; else args="$B"; fi
    python bench.py $common $args 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$args', d['ms_per_step'], d['ms_per_step_p50'])"
  done
done
