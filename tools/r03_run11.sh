#!/bin/bash
# A/B of a library option inside the step: tools/r03_run11.sh "--opt attn16_stagger_fwd=20000" ""
for i in 1 2 3; do
for cfg in "$1" "$2"; do
  MEMHIP_BENCH_STEP_TIMES=1 python bench.py --no-cpu-baseline --no-tokenizer-figure --no-raster-figure --no-config4-figure --no-entrypoint-figure --no-gemm-timer --steps 60 --warmup 5 $cfg 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if 'per-step ms' in l and 'host' not in l:
        v=json.loads(l.split('ms:')[1]); s=sorted(v); print('cfg[$cfg] mean %.2f p50 %.2f max %.2f'%(sum(v)/len(v), s[len(s)//2], s[-1]))
"
done; done
