#!/bin/bash
# tools/ab_lib.sh <libA.so|""> <libB.so|""> [reps] [bench args]: interleaved runs of bench.py with two builds of libmemhip.so ("" = the
# shipped one) on ONE box
A="$1"; B="$2"; R=${3:-2}; shift 3
common="--no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-config4-figure --no-config5-figure --no-entrypoint-figure --no-gemm-timer --steps 20 --warmup 5"
for i in $(seq $R); do
  for v in A B; do
    if [ $v = A ]; then lib="$A"; else lib="$B"; fi
    MEMHIP_LIB="$lib" python bench.py $common "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '${lib:-shipped}', d['ms_per_step'], d['ms_per_step_p50'])"
  done
done
